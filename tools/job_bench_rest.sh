cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for w in cfg3-chr1 cfg2 cfg4; do for rep in a b; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/r3s_${w}_$rep.json 2>> $O/r3_rest.err; done; done
python3 - <<'PY'
import json
for w in ("cfg3-chr1","cfg2","cfg4"):
  for rep in "ab":
    d=json.load(open("gpurun_out/r3s_%s_%s.json"%(w,rep)))
    print(w, rep, "value %.3e ms/step %.2f estep %.2f mstep %.2f" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"]))
PY
