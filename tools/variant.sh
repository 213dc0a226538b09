#!/bin/bash
# development helper: build variants/libphmrf_NAME.so with -DPHMRF_DEV (the environment knobs) and extra defines for strip.hip (the product library is left alone;
# load a variant with PHMRF_LIB=variants/libphmrf_NAME.so).  usage: bash tools/variant.sh NAME "-DPHMRF_PHASE_CLOCK" [file]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=$2; FILE=${3:-strip}
cd "$ROOT/phylo_hmrf_amd/csrc"
make -s -j8 >/dev/null
mkdir -p "$ROOT/variants" .obj/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -fno-honor-nans -mno-amdgpu-ieee -DPHMRF_DEV $DEFS -c -o .obj/var/$FILE.$NAME.o $FILE.hip
OBJS=""
for f in api kernels moves strip graph init coarse tile c2f maxflow; do
  if [ "$f" = "$FILE" ]; then OBJS="$OBJS .obj/var/$FILE.$NAME.o"; else OBJS="$OBJS .obj/dev/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/variants/libphmrf_$NAME.so" $OBJS
echo "built variants/libphmrf_$NAME.so"
