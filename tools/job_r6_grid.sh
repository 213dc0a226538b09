#!/bin/bash
# GPU box: per-launch durations of the warm chr1 solve with the mop-up launches' grid capped (development library)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O/ws_empty
export PHMRF_TRACE_PERT=0.05
export PHMRF_LIB=phylo_hmrf_amd/libphmrf_dev.so
for g in 0 2048 4096 8192 16384; do
  export PHMRF_MOPUP_GRID=$g
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_ws -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/kt_ws.err || exit 1
  python3 profiles/warm_solve_aggregate.py $O/kt_ws $O/ws_empty $O/r6_warm_kt_g$g.json > /dev/null
  python3 -c "
import json
d=json.load(open('$O/r6_warm_kt_g$g.json'))
seq=[(k,u) for k,u in d['launch_order_us'] if 'strip_cols' in k or 'fusion' in k]
print('grid $g:', ' '.join('%s:%.0f'%(k.replace('_kernel','').replace('_cols',''),u) for k,u in seq), '| total strip %.0f fusion %.0f'%(sum(u for k,u in seq if 'strip' in k), sum(u for k,u in seq if 'fusion' in k)))
"
  rm -rf $O/kt_ws
done
