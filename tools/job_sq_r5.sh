#!/bin/bash
# GPU box: SQ counters of the strip kernels in the warm solve of the chr1 block, per development build: bash tools/job_sq_r5.sh name ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O/ws_empty
export PHMRF_TRACE_PERT=0.05
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU")
for v in "$@"; do
  if [ "$v" = "product" ]; then export PHMRF_LIB=""; else export PHMRF_LIB="variants/libphmrf_$v.so"; fi
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/wp_$i -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/sq_${v}_$i.err
    python3 profiles/warm_solve_aggregate.py $O/ws_empty $O/wp_$i $O/sq_${v}_$i.json > /dev/null 2>&1
    python3 -c "
import json
d=json.load(open('$O/sq_${v}_$i.json')).get('warm_solve_sq',{})
for k,x in d.items():
    if 'strip_cols' in k: print('$v', k[:22], {a:(round(b,3) if b<10 else int(b)) for a,b in x.items()})
"
    rm -rf $O/wp_$i
  done
done
