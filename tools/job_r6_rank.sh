#!/bin/bash
# GPU box: one emulated rank of an 8-rank run under a kernel trace: where does its E-step go?  bash tools/job_r6_rank.sh RANK
R=${1:-3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
CMD="bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-kernel-timing --emulate-world 8 --emulate-rank $R"
python3 $CMD > $O/r6_rank${R}_plain.json 2> $O/r6_rank${R}_plain.err || exit 1
python3 -c "
import json;d=json.loads(open('$O/r6_rank${R}_plain.json').read().strip().splitlines()[-1]);e=d['emulated']
print('rank %d plain: %.2f M nodes, %d blocks + tiles %s: E %.2f ms M %.2f ms step %.2f ms'%(e['rank'],e['nodes']/1e6,e['whole_blocks'],e['tiles'],d['estep_ms'],d['mstep_ms'],d['ms_per_step']))"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_rank -- python3 $CMD > $O/r6_rank${R}_kt.json 2> $O/kt_rank.err || exit 1
python3 profiles/aggregate_regime.py $O/kt_rank $O/none $O/r6_rank${R}_kt.json $O/r6_rank${R}_regime.json ${2:-4} 6 > $O/r6_rank${R}_regime.txt 2>&1
head -c 3500 $O/r6_rank${R}_regime.txt
rm -rf $O/kt_rank
