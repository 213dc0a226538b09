cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r3_d_bench_steps20_warmup5.json 2> $O/r3_d.err
python bench.py --no-cpu-baseline > $O/r3_d_bench_default.json 2>> $O/r3_d.err
for w in cfg2 cfg4 cfg3-chr1; do python bench.py --workload $w --no-cpu-baseline > $O/r3_d_$w.json 2>> $O/r3_d.err; done
python3 - <<'PY'
import json
for f in ("r3_d_bench_steps20_warmup5","r3_d_bench_default","r3_d_cfg2","r3_d_cfg4","r3_d_cfg3-chr1"):
    d=json.load(open("gpurun_out/%s.json"%f))
    r=d["roofline"]
    print(f, "value %.3e ms/step %.1f estep %.1f mstep %.1f" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"]), "| roofline", r["kernel"], r["frac"], "isolated", r["isolated"], "traffic", r["traffic"])
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["vectorised"]["value"])
PY
