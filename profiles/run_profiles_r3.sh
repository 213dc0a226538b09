#!/bin/bash
# Round 3: everything under profiles/r3_* from ONE build (GPU box, through gpurun): bash profiles/run_profiles_r3.sh GIT_REV
#   r3_bench.json, r3_bench_steps20_warmup5.json        the bench lines (default command; the driver's command)
#   r3_kernel_stats.csv, r3_bench_under_rocprof.json    rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline`
#                                                       (12 blocks in flight: a kernel's duration includes the share of the
#                                                       GPU that other streams took) and the JSON line of that profiled run
#   r3_kernel_stats_serial.csv, r3_bench_under_rocprof_serial.json   the same with --block-threads 1 --mstep-workers 1: one
#                                                       stream in flight, so the stats' AverageNs of strip_cols_kernel is
#                                                       what roofline.avg_launch_us / roofline.isolated measure live
#   pmc_by_kernel.json                                  FETCH_SIZE / WRITE_SIZE per kernel name (run_pmc_by_kernel.sh)
#   r3_warm_solve.json                                  one warm solve of the chr1 block, launch by launch + SQ counters
REV=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
# the counters first: bench.py fills roofline.traffic from profiles/pmc_by_kernel.json when its source_hash is this build's
bash profiles/run_pmc_by_kernel.sh $REV > $O/r3_pmc.out 2>&1
[ -s $O/pmc_by_kernel.json ] && cp $O/pmc_by_kernel.json profiles/pmc_by_kernel.json
python3 bench.py > $O/r3_bench.json 2> $O/r3_bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r3_bench_steps20_warmup5.json 2>> $O/r3_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_stats -- python3 bench.py --no-cpu-baseline > $O/r3_bench_under_rocprof.json 2> $O/r3_rocprof.err
find $O/r3_stats -name "*kernel_stats.csv" -exec cp {} $O/r3_kernel_stats.csv \;
rm -rf $O/r3_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_stats -- python3 bench.py --no-cpu-baseline --block-threads 1 --mstep-workers 1 > $O/r3_bench_under_rocprof_serial.json 2>> $O/r3_rocprof.err
find $O/r3_stats -name "*kernel_stats.csv" -exec cp {} $O/r3_kernel_stats_serial.csv \;
rm -rf $O/r3_stats
bash profiles/warm_solve_profile.sh r3 > $O/r3_warm.out 2>&1
python3 - <<'PY'
import json, csv
for f in ("r3_bench", "r3_bench_steps20_warmup5", "r3_bench_under_rocprof", "r3_bench_under_rocprof_serial"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "value %.3e  ms/step %.1f (E %.1f + M %.1f)" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"]),
              "| %s frac %.3f avg_launch_us %.1f isolated %s" % (r["kernel"], r["frac"], r["avg_launch_us"], r["isolated"]))
    except Exception as e:
        print(f, "FAILED", e)
for f in ("r3_kernel_stats", "r3_kernel_stats_serial"):
    try:
        rows = list(csv.DictReader(open("gpurun_out/%s.csv" % f)))
        for r in rows[:6]:
            print(f, r["Name"][:60], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
    except Exception as e:
        print(f, "FAILED", e)
PY
