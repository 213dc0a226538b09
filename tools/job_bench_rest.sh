cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for w in cfg2 cfg4 cfg3-chr1; do python bench.py --workload $w --no-cpu-baseline > $O/r3_$w.json 2>> $O/r3_rest.err; done
python3 - <<'PY'
import json
for f in ("r3_cfg2","r3_cfg4","r3_cfg3-chr1"):
    d=json.load(open("gpurun_out/%s.json"%f))
    print(f, "value %.3e ms/step %.1f estep %.1f mstep %.1f" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"]))
PY
