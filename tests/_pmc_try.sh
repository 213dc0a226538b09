#!/bin/bash
# development helper (GPU box): does `bench.py --workload small` finish under rocprofv3 --pmc with variants/libphmrf_$1.so ?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$1" ] && cp variants/libphmrf_$1.so phylo_hmrf_amd/libphmrf.so
timeout -k 10 ${2:-60} rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pm_small -- python3 bench.py --workload small --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/pm_small.out 2> gpurun_out/pm_small.err
echo "variant=$1 rc=$?"
rm -rf gpurun_out/pm_small
