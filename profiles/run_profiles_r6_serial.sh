#!/bin/bash
# Round 6: the pair tests/test_roofline_profiles.py reads -- ONE run of bench.py with one block in flight under
# rocprofv3 --kernel-trace --stats (GPU box, through gpurun):  bash profiles/run_profiles_r6_serial.sh
#   profiles/r6_kernel_stats_serial.csv            the profiler's per-kernel summary of that run
#   profiles/r6_bench_under_rocprof_serial.json    the JSON line of that run
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_stats -- python3 bench.py --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance --block-threads 1 --mstep-workers 1 --steps 8 --warmup 2 > $O/r6_bench_under_rocprof_serial.json 2> $O/r6_rocprof_serial.err
find $O/r6_stats -name "*kernel_stats.csv" -exec cp {} $O/r6_kernel_stats_serial.csv \;
rm -rf $O/r6_stats
python3 - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/r6_bench_under_rocprof_serial.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.3e ms/step %.1f | %s frac %.4f achieved %.1f GB/s own avg %.1f us, full %s, mop %s" % (d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["achieved"], r["avg_launch_us"], r["full_sweep"], r["mop_up"]))
for row in list(csv.DictReader(open("gpurun_out/r6_kernel_stats_serial.csv")))[:8]:
    print(row["Name"][:70], row["Calls"], "avg_us %.1f" % (float(row["AverageNs"]) / 1e3), row["Percentage"])
PY
