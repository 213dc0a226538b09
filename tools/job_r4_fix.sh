cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py > $O/r4_bench.json 2> $O/r4_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_stats -- python3 bench.py --no-cpu-baseline --no-fit > $O/r4_bench_under_rocprof.json 2> $O/r4_rocprof.err
find $O/r4_stats -name "*kernel_stats.csv" -exec cp {} $O/r4_kernel_stats.csv \;
rm -rf $O/r4_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_stats -- python3 bench.py --no-cpu-baseline --no-fit --block-threads 1 --mstep-workers 1 > $O/r4_bench_under_rocprof_serial.json 2>> $O/r4_rocprof.err
find $O/r4_stats -name "*kernel_stats.csv" -exec cp {} $O/r4_kernel_stats_serial.csv \;
rm -rf $O/r4_stats
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --emulate-world 4 --emulate-rank 2 > $O/r4_emu4_r2.json 2>/dev/null
python3 - <<'PY'
import json, csv
for f in ("r4_bench", "r4_bench_under_rocprof", "r4_bench_under_rocprof_serial", "r4_emu4_r2"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.3e  ms/step %.1f (E %.1f + M %.1f)" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"]), "| %s frac %.3f avg_launch_us %.1f iso_frac %s launches %d bytes/launch %d" % (r["kernel"], r["frac"], r["avg_launch_us"], r.get("kernel_frac_isolated"), r["launches"], r["algorithmic_bytes_per_launch"]))
for f in ("r4_kernel_stats", "r4_kernel_stats_serial"):
    rows = list(csv.DictReader(open("gpurun_out/%s.csv" % f)))
    for r in rows[:6]:
        print(f, r["Name"][:60], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
