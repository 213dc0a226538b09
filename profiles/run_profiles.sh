#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: refreshes the round's bench line, kernel stats and PMC traffic.
# usage: bash profiles/run_profiles.sh TAG      -> gpurun_out/TAG_*  (copy what is to be judged into profiles/)
TAG=${1:-r2_x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.err
echo stats done; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 bench.py --steps 2 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/${TAG}_pmc_fetch.err
echo fetch done; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 bench.py --steps 2 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/${TAG}_pmc_write.err
python3 profiles/aggregate_pmc.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_by_kernel.json $O/${TAG}_pmc_traffic.json > /dev/null
find $O/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
# the raw traces are large: keep the summaries only
rm -rf $O/${TAG}_stats $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
tail -1 $O/${TAG}_bench.json
