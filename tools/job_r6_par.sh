#!/bin/bash
# GPU box: four waves per strip in the late rounds -- the A/B test, then the warm chr1 solve launch by launch per threshold
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O/ws_empty
timeout -k 10 900 python3 -m pytest tests/test_gpu_estep.py -x -q -m gpu -k "seed_masks or deterministic or strip_multi or strip_passes" > $O/r6_par_tests.log 2>&1; rc=$?
tail -4 $O/r6_par_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
export PHMRF_TRACE_PERT=0.05
export PHMRF_LIB=phylo_hmrf_amd/libphmrf_dev.so
for g in 0 1024 3072 8192 100000; do
  export PHMRF_PAR_DIRTY=$g
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_ws -- python3 tools/trace.py 20 ${PAR_N:-4980} 1000 > /dev/null 2> $O/kt_ws.err || exit 1
  python3 profiles/warm_solve_aggregate.py $O/kt_ws $O/ws_empty $O/r6_par_kt_$g.json > /dev/null
  python3 -c "
import json
d=json.load(open('$O/r6_par_kt_$g.json'))
seq=[(k,u) for k,u in d['launch_order_us'] if 'strip_cols' in k]
print('par_dirty $g:', ' '.join('%s:%.0f'%(k.replace('_kernel','').replace('_cols',''),u) for k,u in seq), '| total strip %.0f'%(sum(u for k,u in seq)))
"
  rm -rf $O/kt_ws
done
