#!/usr/bin/env python3
"""The BENCHED regime (bench.py's default: 26 blocks, 14 in flight) under rocprofv3, two passes of the same command
    python3 bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance --no-kernel-timing

  KT_DIR   rocprofv3 --kernel-trace                      (concurrent streams as benched: where each stream's time goes)
  PMC_DIR  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
           (the counter pass serialises the kernels: its counts are per kernel, its clock is not the benched one)

usage: aggregate_regime.py KT_DIR PMC_DIR BENCH_JSON OUT.json [blocks_per_iteration=26] [first_steady_iteration=6]

An EM iteration is found by counting emission_kernel dispatches (one per block and E-step; the set-up runs one round of
them too, so iteration 0 is the set-up and 1 the cold first E-step).  Steady iterations: first_steady_iteration onwards.
"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from aggregate_pmc import short

SIMDS = 256 * 4          # MI355X: 256 CUs x 4 SIMDs
XCDS = 8


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def family(k):
    if k.startswith("__amd_rocclr_copy"):
        return "copy"
    if k.startswith("__amd_rocclr_fill"):
        return "fill"
    if not any(k.startswith(p) for p in ("strip", "fusion", "propose", "cc_", "comp_", "energy", "emission", "posterior",
                                         "coars", "icm", "chain", "choose", "unary", "dirty", "tile", "put_", "argmax",
                                         "round_", "seed")):
        return "other (torch / rocBLAS)"
    return k


def iteration_index(sorted_rows, per_iter):
    """iteration number of every dispatch (rows sorted by dispatch order): emission dispatches counted so far"""
    seen, out = 0, []
    for r in sorted_rows:
        if short(r["Kernel_Name"]).startswith("emission_kernel"):
            seen += 1
        out.append((seen - 1) // per_iter if seen else -1)
    return out


def main():
    kt_dir, pmc_dir, bench_json, out_path = sys.argv[1:5]
    per_iter = int(sys.argv[5]) if len(sys.argv) > 5 else 26
    first = int(sys.argv[6]) if len(sys.argv) > 6 else 6
    out = {"command": "python3 bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance --no-kernel-timing"}
    bench = None
    try:
        bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
        out["bench_under_kernel_trace"] = {k: bench.get(k) for k in ("ms_per_step", "estep_ms", "mstep_ms", "value", "block_threads")}
        out["build"] = bench.get("build")
    except Exception as e:          # noqa
        out["bench_under_kernel_trace"] = "unreadable: %s" % e

    # ---- pass 1: the concurrent regime, per stream ------------------------------------------------------------------
    kt = rows(kt_dir, "*kernel_trace.csv")
    kt.sort(key=lambda r: int(r["Start_Timestamp"]))
    if kt:
        it = iteration_index(kt, per_iter)
        steady = [(r, i) for r, i in zip(kt, it) if first <= i < first + 4]      # (the 4 timed steps)
        iters = sorted({i for _, i in steady})
        n_it = max(len(iters), 1)
        t_lo = min(int(r["Start_Timestamp"]) for r, _ in steady)
        t_hi = max(int(r["End_Timestamp"]) for r, _ in steady)
        skey = "Stream_Id" if "Stream_Id" in kt[0] else "Queue_Id"
        streams, fam = {}, {}
        ivs = []
        for r, _ in steady:
            k = family(short(r["Kernel_Name"]))
            s0, s1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            d = (s1 - s0) / 1e6
            st = streams.setdefault(r.get(skey, "?"), {"kernel_ms": 0.0, "copy_fill_ms": 0.0, "dispatches": 0, "first": s0, "last": s1})
            st["copy_fill_ms" if k in ("copy", "fill") else "kernel_ms"] += d
            st["dispatches"] += 1
            st["first"], st["last"] = min(st["first"], s0), max(st["last"], s1)
            e = fam.setdefault(k, {"dispatches": 0, "ms": 0.0})
            e["dispatches"] += 1
            e["ms"] += d
            ivs.append((s0, s1))
        ivs.sort()
        busy, cs, ce = 0, ivs[0][0], ivs[0][1]
        for s0, s1 in ivs[1:]:
            if s0 > ce:
                busy += ce - cs
                cs, ce = s0, s1
            elif s1 > ce:
                ce = s1
        busy += ce - cs
        tot_ms = sum(e["ms"] for e in fam.values())
        out["concurrent"] = {
            "steady_iterations": n_it, "window_ms_per_iteration": round((t_hi - t_lo) / 1e6 / n_it, 2),
            "gpu_busy_ms_per_iteration (>= 1 kernel running)": round(busy / 1e6 / n_it, 2),
            "summed_kernel_ms_per_iteration": round(tot_ms / n_it, 2),
            "copy_fill_share_of_summed_time": round(sum(fam.get(k, {"ms": 0})["ms"] for k in ("copy", "fill")) / max(tot_ms, 1e-9), 4),
            "copy_fill_dispatches_per_iteration": round(sum(fam.get(k, {"dispatches": 0})["dispatches"] for k in ("copy", "fill")) / n_it, 1),
            "mean_dispatches_in_flight_while_busy": round(tot_ms * 1e6 / max(busy, 1), 2),
            "per_family": {k: {"dispatches_per_iteration": round(v["dispatches"] / n_it, 1), "ms_per_iteration": round(v["ms"] / n_it, 3),
                               "avg_us": round(1e3 * v["ms"] / v["dispatches"], 1)}
                           for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])},
            "per_stream (%s)" % skey: {s: {"in_kernel_ms": round(v["kernel_ms"] / n_it, 2), "in_copy_fill_ms": round(v["copy_fill_ms"] / n_it, 2),
                                          "dispatches": round(v["dispatches"] / n_it, 1)}
                                      for s, v in sorted(streams.items(), key=lambda kv: -kv[1]["kernel_ms"])},
        }

    # ---- pass 2: the counters (serialised kernels) ------------------------------------------------------------------
    pm = rows(pmc_dir, "*counter_collection.csv")
    if pm:
        by_d = {}
        for r in pm:
            e = by_d.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"]})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ds = [dict(v, Dispatch_Id=k) for k, v in sorted(by_d.items())]
        it = iteration_index(ds, per_iter)
        fam, tot = {}, {}
        iters = set()
        for r, i in zip(ds, it):
            if i < first or i >= first + 4:
                continue
            iters.add(i)
            k = family(short(r["Kernel_Name"]))
            e = fam.setdefault(k, {"dispatches": 0})
            e["dispatches"] += 1
            for c, v in r.items():
                if c in ("Kernel_Name", "Dispatch_Id"):
                    continue
                e[c] = e.get(c, 0.0) + v
                tot[c] = tot.get(c, 0.0) + v
        n_it = max(len(iters), 1)
        res = {}
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
            busy = v.get("GRBM_GUI_ACTIVE", 0.0) / XCDS        # (the counter is summed over the 8 XCDs)
            res[k] = {"dispatches_per_iteration": round(v["dispatches"] / n_it, 1),
                      "SQ_INSTS_VALU_per_iteration": int(v.get("SQ_INSTS_VALU", 0) / n_it),
                      "SQ_WAVES_per_iteration": int(v.get("SQ_WAVES", 0) / n_it),
                      "active_clocks_per_iteration (GRBM_GUI_ACTIVE / 8)": int(busy / n_it),
                      "valu_pipe_busy_own (INSTS_VALU x 4 / (active clocks x 1024 SIMDs))":
                          round(v.get("SQ_INSTS_VALU", 0) * 4.0 / max(busy * SIMDS, 1), 4)}
        out["counters"] = {"steady_iterations": n_it, "per_family": res,
                           "SQ_INSTS_VALU_per_iteration": int(tot.get("SQ_INSTS_VALU", 0) / n_it)}
        # whole-E-step VALU-pipe utilisation in the BENCHED regime: the instructions of one iteration (a property of the
        # work, not of the pass's timing) over the SIMD-clocks of the E-step as benched
        if bench and bench.get("estep_ms"):
            ghz = float(os.environ.get("PHMRF_GPU_GHZ", "2.4"))
            clocks = bench["estep_ms"] * 1e-3 * ghz * 1e9
            out["estep_valu_pipe_busy_as_benched"] = {
                "value": round(tot.get("SQ_INSTS_VALU", 0) / n_it * 4.0 / (clocks * SIMDS), 4),
                "estep_ms": bench["estep_ms"], "assumed_GHz": ghz,
                "formula": "SQ_INSTS_VALU per EM iteration x 4 / (E-step seconds of the kernel-trace pass x clock x 1024 SIMDs)"}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("concurrent", "counters")}, indent=1))
    if "concurrent" in out:
        c = dict(out["concurrent"])
        c.pop("per_stream (%s)" % ("Stream_Id" if kt and "Stream_Id" in kt[0] else "Queue_Id"), None)
        print(json.dumps(c, indent=1)[:4000])
    if "counters" in out:
        print(json.dumps(out["counters"], indent=1)[:5000])


if __name__ == "__main__":
    main()
