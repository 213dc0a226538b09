#!/bin/bash
# GPU box: (1) the driver's command as a same-box baseline, (2) the benched regime under rocprofv3 -- a kernel trace with
# the 14 blocks in flight, then a counter pass -- aggregated by profiles/aggregate_regime.py.   bash tools/job_r6_regime.sh TAG
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit > $O/${TAG}_base.json 2> $O/${TAG}_base.err || exit 1
python3 -c "
import json;d=json.loads(open('$O/${TAG}_base.json').read().strip().splitlines()[-1]);r=d['roofline']
print('base: ms/step %.2f estep %.2f mstep %.2f cold %.0f | frac %.4f full %s mop %s'%(d['ms_per_step'],d['estep_ms'],d['mstep_ms'],d['cold_first_iteration_ms'],r['frac'],r['full_sweep']['own_ms'],r['mop_up']['own_ms']))"
CMD="bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance --no-kernel-timing"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_kt -- python3 $CMD > $O/${TAG}_regime_bench.json 2> $O/${TAG}_kt.err || exit 1
echo "kernel trace done"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc -- python3 $CMD > $O/${TAG}_regime_bench_pmc.json 2> $O/${TAG}_pmc.err || exit 1
echo "pmc done"
python3 profiles/aggregate_regime.py $O/${TAG}_kt $O/${TAG}_pmc $O/${TAG}_regime_bench.json $O/${TAG}_regime.json > $O/${TAG}_regime.txt 2>&1
head -c 6000 $O/${TAG}_regime.txt
rm -rf $O/${TAG}_kt $O/${TAG}_pmc
