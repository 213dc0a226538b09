#!/bin/bash
# quick: kernel-trace only
TAG=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export PHMRF_TRACE_PERT=0.05
rocprofv3 --kernel-trace --output-format csv -d $O/ws_kt -- python3 tools/trace.py 20 4980 1000 > $O/${TAG}_warm_trace.out 2> $O/${TAG}_warm_trace.err
mkdir -p $O/ws_empty
python3 profiles/warm_solve_aggregate.py $O/ws_kt $O/ws_empty $O/${TAG}_warm_solve.json > /dev/null
rm -rf $O/ws_kt
python3 -c "
import json
d=json.load(open('$O/${TAG}_warm_solve.json'))
k=d['warm_solve_kernels_us']
print('$TAG total_us', round(sum(v['total_us'] for v in k.values())), {a:round(v['total_us']) for a,v in k.items() if 'strip' in a})
print([x[1] for x in d['launch_order_us'] if 'multi' in x[0]])
"
tail -2 $O/${TAG}_warm_trace.out | cut -c1-150
