cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export PHMRF_TRACE_PERT=0.05
run() { # tag, counters
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/ws_pm -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/$1.err
  mkdir -p $O/ws_empty
  python3 profiles/warm_solve_aggregate.py $O/ws_empty $O/ws_pm $O/$1.json > /dev/null
  rm -rf $O/ws_pm
  python3 - <<PY
import json
d=json.load(open("$O/$1.json")).get("warm_solve_sq",{})
for k,v in d.items():
    if "strip" in k: print("$1", k, {a:(round(b,4) if b<10 else int(b)) for a,b in v.items()})
PY
}
run r3_cols_ic "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU"
PHMRF_MULTI_V=1 run r3_old_ic "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU"
PHMRF_MULTI_V=1 run r3_old_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
