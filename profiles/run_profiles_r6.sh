#!/bin/bash
# Round 6: everything under profiles/r6_* from ONE build (GPU box, through gpurun; three calls, each inside gpurun's limit):
#   bash profiles/run_profiles_r6.sh a GIT_REV   pmc_by_kernel.json (FETCH_SIZE / WRITE_SIZE per kernel name) + the driver's command x 3
#                                                + the default command
#   bash profiles/run_profiles_r6.sh b           rocprofv3 --kernel-trace --stats of bench.py with 14 blocks in flight and with one
#                                                (the pair tests/test_roofline_profiles.py reads) + one warm solve launch by launch
#   bash profiles/run_profiles_r6.sh c           the 8-rank rehearsal, the other workloads, the lockstep rounds' exchange
PART=${1:-a}
REV=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
summ() {
python3 - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f).read().strip().splitlines() if l.startswith("{")][-1])
        r = d["roofline"]
        ft, fr = d.get("fit") or {}, d.get("fit_reference_start") or {}
        print(f, "value %.3e  ms/step %.1f (E %.1f + M %.1f) median %.1f cold %.0f" % (d["value"], d["ms_per_step"], d["estep_ms"], d["mstep_ms"], d.get("ms_per_step_median") or -1, d.get("cold_first_iteration_ms") or -1),
              "| %s frac %.4f (full %.4f at %.0f us, mop-up %.4f at %.0f us) class %.3f | fit %s it %.3e / ref-start %s it %.3e" % (
                  r["kernel"], r["frac"] or 0, (r["full_sweep"]["GBps"] or 0) / r["peak"], r["full_sweep"]["avg_launch_us"], (r["mop_up"]["GBps"] or 0) / r["peak"], r["mop_up"]["avg_launch_us"],
                  r["class_throughput"]["frac"], ft.get("iterations"), ft.get("value") or 0, fr.get("iterations"), fr.get("value") or 0))
    except Exception as e:
        print(f, "FAILED", e)
PY
}
if [ "$PART" = "a" ]; then
  # the counters first: bench.py fills roofline.traffic from profiles/pmc_by_kernel.json when its source_hash is this build's
  bash profiles/run_pmc_by_kernel.sh $REV > $O/r6_pmc.out 2>&1
  [ -s $O/pmc_by_kernel.json ] && cp $O/pmc_by_kernel.json profiles/pmc_by_kernel.json
  echo "pmc done"
  python3 bench.py > $O/r6_bench.json 2> $O/r6_bench.err
  for t in "" _b _c; do python3 bench.py --steps 20 --warmup 5 > $O/r6_bench_steps20_warmup5$t.json 2>> $O/r6_bench.err; done
  summ r6_bench r6_bench_steps20_warmup5 r6_bench_steps20_warmup5_b r6_bench_steps20_warmup5_c
fi
if [ "$PART" = "b" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_stats -- python3 bench.py --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance > $O/r6_bench_under_rocprof.json 2> $O/r6_rocprof.err
  find $O/r6_stats -name "*kernel_stats.csv" -exec cp {} $O/r6_kernel_stats.csv \;
  rm -rf $O/r6_stats
  bash profiles/run_profiles_r6_serial.sh
  bash profiles/warm_solve_profile.sh r6 > $O/r6_warm.out 2>&1
  summ r6_bench_under_rocprof r6_bench_under_rocprof_serial
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6_warm_solve.json"))
k = d["warm_solve_kernels_us"]
print("warm solve of the chr1 block: kernels %.0f us" % sum(v["total_us"] for v in k.values()), {a: round(v["total_us"]) for a, v in sorted(k.items(), key=lambda kv: -kv[1]["total_us"])[:12]})
PY
fi
if [ "$PART" = "c" ]; then
  bash tools/job_emulate_only.sh r6 8 > $O/r6_emu8.log 2>&1
  cat $O/r6_emu8.log
  for w in cfg3-chr1 cfg2 cfg4; do python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/r6_$w.json 2>> $O/r6_rest.err; done
  python3 bench.py --workload cfg5-chr1 --steps 3 --warmup 2 --no-cpu-baseline --no-fit --no-through-fit > $O/r6_cfg5-chr1.json 2>> $O/r6_rest.err
  summ r6_cfg3-chr1 r6_cfg2 r6_cfg4 r6_cfg5-chr1
  bash tools/job_tile_exchange.sh r6
fi
