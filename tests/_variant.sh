#!/bin/bash
# development helper (GPU box): rebuild strip.o with extra defines and relink.  usage: bash tests/_variant.sh "-DPHMRF_MULTI_WPE=2"
cd "$GRAFT_REPO_ROOT/phylo_hmrf_amd/csrc"
mkdir -p .obj
for f in api kernels moves graph init coarse; do
  [ -f .obj/$f.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -fno-honor-nans -mno-amdgpu-ieee -c -o .obj/$f.o $f.hip &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -fno-honor-nans -mno-amdgpu-ieee $1 -c -o .obj/strip.o strip.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libphmrf.so .obj/*.o
