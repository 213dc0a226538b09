#!/bin/bash
# SQ counters of the warm part of tools/trace.py on the chr1 block; usage: warm_solve_pmc.sh TAG "COUNTER ..." ["COUNTER ..." ...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export PHMRF_TRACE_PERT=0.05
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/wp_$i -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/${TAG}_pmc$i.err
  mkdir -p $O/wp_empty
  python3 profiles/warm_solve_aggregate.py $O/wp_empty $O/wp_$i $O/${TAG}_pmc$i.json > /dev/null 2>&1
  python3 -c "
import json
d=json.load(open('$O/${TAG}_pmc$i.json')).get('warm_solve_sq',{})
for k in ('strip_cols_kernel','fusion_cols_kernel','strip_kernel'):
    if k in d: print(k, {a:(v if '/' in a or 'per' in a else int(v)) for a,v in d[k].items()})
"
  rm -rf $O/wp_$i
done
