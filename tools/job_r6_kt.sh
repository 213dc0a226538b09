#!/bin/bash
# GPU box: kernel trace of the warm solve of the chr1-sized block -> per-launch durations in launch order
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O/ws_empty
export PHMRF_TRACE_PERT=0.05
[ -n "$1" ] && [ "$1" != product ] && export PHMRF_LIB=$1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_ws -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/kt_ws.err || exit 1
python3 profiles/warm_solve_aggregate.py $O/kt_ws $O/ws_empty $O/r6_warm_kt.json > /dev/null
python3 -c "
import json
d=json.load(open('$O/r6_warm_kt.json'))
print(' '.join('%s:%.0f'%(k.replace('_kernel','').replace('__amd_rocclr_',''),u) for k,u in d['launch_order_us']))
print({k:v for k,v in d['warm_solve_kernels_us'].items()})
"
rm -rf $O/kt_ws
