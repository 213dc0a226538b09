cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
TAG=${1:-r3}
export PHMRF_TRACE_PERT=0.05
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/ws_pm -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/${TAG}_warm_pmc.err
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/ws_pm2 -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/${TAG}_warm_pmc2.err
mkdir -p $O/ws_empty
python3 profiles/warm_solve_aggregate.py $O/ws_empty $O/ws_pm $O/${TAG}_sq1.json > /dev/null
python3 profiles/warm_solve_aggregate.py $O/ws_empty $O/ws_pm2 $O/${TAG}_sq2.json > /dev/null
rm -rf $O/ws_pm $O/ws_pm2
python3 - <<PY
import json
for f in ("$O/${TAG}_sq1.json","$O/${TAG}_sq2.json"):
    d=json.load(open(f)).get("warm_solve_sq",{})
    for k,v in d.items():
        if "strip" in k: print(k, {a:(round(b,4) if b<10 else int(b)) for a,b in v.items()})
PY
