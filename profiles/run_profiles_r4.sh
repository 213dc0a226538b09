#!/bin/bash
# Round 4: everything under profiles/r4_* from ONE build (GPU box, through gpurun): bash profiles/run_profiles_r4.sh GIT_REV
#   r4_bench_steps20_warmup5{,_b,_c}.json               the driver's command, three runs (with the `fit` object)
#   r4_bench.json                                       the default command
#   r4_kernel_stats.csv, r4_bench_under_rocprof.json    rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-fit`
#                                                       (14 blocks in flight) and the JSON line of that profiled run
#   r4_kernel_stats_serial.csv, r4_bench_under_rocprof_serial.json   the same with --block-threads 1 --mstep-workers 1: one
#                                                       stream in flight, a launch's duration is the kernel's own
#   pmc_by_kernel.json                                  FETCH_SIZE / WRITE_SIZE per kernel name (run_pmc_by_kernel.sh)
#   r4_emu8_r{0..7}.json                                one-GPU rehearsal of every rank of an 8-GPU run of cfg3 (row tiles,
#                                                       dealt M-step): bench.py --emulate-world 8 --emulate-rank r
#   r4_cfg2.json r4_cfg4.json r4_cfg3-chr1.json r4_cfg5-chr1.json   the other workloads
#   r4_warm_solve.json                                  one warm solve of the chr1 block, launch by launch
REV=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
# the counters first: bench.py fills roofline.traffic from profiles/pmc_by_kernel.json when its source_hash is this build's
bash profiles/run_pmc_by_kernel.sh $REV > $O/r4_pmc.out 2>&1
[ -s $O/pmc_by_kernel.json ] && cp $O/pmc_by_kernel.json profiles/pmc_by_kernel.json
echo "pmc done"
python3 bench.py > $O/r4_bench.json 2> $O/r4_bench.err
for t in "" _b _c; do python3 bench.py --steps 20 --warmup 5 > $O/r4_bench_steps20_warmup5$t.json 2>> $O/r4_bench.err; done
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_stats -- python3 bench.py --no-cpu-baseline --no-fit > $O/r4_bench_under_rocprof.json 2> $O/r4_rocprof.err
find $O/r4_stats -name "*kernel_stats.csv" -exec cp {} $O/r4_kernel_stats.csv \;
rm -rf $O/r4_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_stats -- python3 bench.py --no-cpu-baseline --no-fit --block-threads 1 --mstep-workers 1 > $O/r4_bench_under_rocprof_serial.json 2>> $O/r4_rocprof.err
find $O/r4_stats -name "*kernel_stats.csv" -exec cp {} $O/r4_kernel_stats_serial.csv \;
rm -rf $O/r4_stats
echo "rocprof done"
bash tools/job_emulate_only.sh r4 8 > $O/r4_emu8.log 2>&1
echo "emulate done"
for w in cfg3-chr1 cfg2 cfg4; do python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/r4_$w.json 2>> $O/r4_rest.err; done
python3 bench.py --workload cfg5-chr1 --steps 3 --warmup 2 --no-cpu-baseline --no-fit > $O/r4_cfg5-chr1.json 2>> $O/r4_rest.err
echo "rest done"
bash profiles/warm_solve_profile.sh r4 > $O/r4_warm.out 2>&1
python3 - <<'PY'
import json, csv
for f in ("r4_bench", "r4_bench_steps20_warmup5", "r4_bench_steps20_warmup5_b", "r4_bench_steps20_warmup5_c", "r4_bench_under_rocprof",
          "r4_bench_under_rocprof_serial", "r4_cfg3-chr1", "r4_cfg2", "r4_cfg4", "r4_cfg5-chr1"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        ft = d.get("fit") or {}
        print(f, "value %.3e  ms/step %.1f (E %.1f + M %.1f) cold %.0f" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"], d.get("cold_first_iteration_ms") or -1),
              "| %s frac %.3f avg_launch_us %.1f iso_frac %s | fit %s it %.3e" % (r["kernel"], r["frac"], r["avg_launch_us"], r.get("kernel_frac_isolated"), ft.get("iterations"), ft.get("value") or 0))
    except Exception as e:
        print(f, "FAILED", e)
for f in ("r4_kernel_stats", "r4_kernel_stats_serial"):
    try:
        rows = list(csv.DictReader(open("gpurun_out/%s.csv" % f)))
        for r in rows[:8]:
            print(f, r["Name"][:60], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
    except Exception as e:
        print(f, "FAILED", e)
print(open("gpurun_out/r4_emu8.log").read())
PY
