"""CPU: `roofline.frac` of the bench line is the dominant kernel's own fraction of the HBM roofline, and it can be recomputed
from the committed profile pair of ONE run (profiles/r6_bench_under_rocprof_serial.json = the JSON line of
`rocprofv3 --kernel-trace --stats -- python3 bench.py --block-threads 1 ...`, profiles/r6_kernel_stats_serial.csv = that
run's per-kernel summary): algorithmic bytes per launch (SURVEY.md 8d accounting x device-counted cells, from the line) over
the kernel's average duration in the profiler's summary, against the same peak.  The two must agree within 10 %
(the summary also holds the launches of the cold warm-up iteration and of the passes after the timed region, whose mix of
full sweeps and mop-up launches differs a little from the timed region's)."""
import csv
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r6_bench_under_rocprof_serial.json")
STATS = os.path.join(ROOT, "profiles", "r6_kernel_stats_serial.csv")


def _load():
    assert os.path.exists(LINE) and os.path.exists(STATS), "profiles/r6_bench_under_rocprof_serial.json / r5_kernel_stats_serial.csv missing"
    d = json.loads(open(LINE).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(STATS)))
    return d, rows


def test_frac_of_the_bench_line_follows_from_the_profilers_kernel_durations():
    d, rows = _load()
    r = d["roofline"]
    assert d["block_threads"] == 1, "the pair must come from a --block-threads 1 run (one stream in flight)"
    names = r["kernel_names"]
    calls = total_ns = 0
    for row in rows:
        if any(("::" + k + "<") in row["Name"] or ("::" + k + "(") in row["Name"] for k in names):
            calls += int(row["Calls"])
            total_ns += float(row["TotalDurationNs"])
    assert calls > 0, names
    avg_us = total_ns / calls / 1e3
    frac_csv = r["algorithmic_bytes_per_launch"] / (avg_us * 1e-6) / 1e9 / r["peak"]
    assert abs(frac_csv - r["frac"]) <= 0.10 * r["frac"], (frac_csv, r["frac"], avg_us, r["avg_launch_us"])
    # the line's own durations (HIP events around each launch) against the profiler's
    assert abs(avg_us - r["avg_launch_us"]) <= 0.10 * avg_us, (avg_us, r["avg_launch_us"])


def test_frac_is_bytes_over_own_durations_and_its_parts_add_up():
    d, _ = _load()
    r = d["roofline"]
    full, mop = r["full_sweep"], r["mop_up"]
    by, ms = full["bytes"] + mop["bytes"], full["own_ms"] + mop["own_ms"]
    assert full["launches"] + mop["launches"] == r["launches"]
    assert abs(by / (ms * 1e-3) / 1e9 - r["achieved"]) <= 0.01 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # a full sweep moves far more bytes per launch than a mop-up launch and runs nearer the roofline
    assert full["bytes"] / full["launches"] > mop["bytes"] / max(mop["launches"], 1)
    assert full["GBps"] > (mop["GBps"] or 0.0)
    assert d["build"]["source_hash"]


def test_valu_on_the_line_follows_from_the_committed_counter_pass():
    """Round 6: `roofline.valu.busy` -- the dominant kernel's share of the vector pipes' issue capacity -- is
    SQ_INSTS_VALU x 4 clocks / (GRBM_GUI_ACTIVE / 8 XCDs x 1,024 SIMDs) over the kernel's dispatches in the committed
    rocprofv3 --pmc pass (profiles/pmc_by_kernel.json, `valu` per kernel name), and `bound` names whichever of it and the HBM
    fraction is the higher.  Recomputed here from the raw counters; the pass's own clock (active clocks / kernel duration) must
    be a plausible MI355X clock, which checks the / 8."""
    d, _ = _load()
    r = d["roofline"]
    pk = json.load(open(os.path.join(ROOT, "profiles", "pmc_by_kernel.json")))
    insts = clocks = ns = 0.0
    for name, rec in pk["kernels"].items():
        v = rec.get("valu")
        if v and any(name.startswith(k) for k in r["kernel_names"]):
            insts += v["sq_insts_valu"]
            clocks += v["active_clocks"]
            ns += v["duration_ns"]
    assert clocks > 0, "no SQ_INSTS_VALU pass in profiles/pmc_by_kernel.json (profiles/run_pmc_by_kernel.sh)"
    busy = insts * 4.0 / (clocks * 1024.0)
    assert 0.2 < busy < 1.0, busy
    assert 1.2 < clocks / ns < 3.0, clocks / ns                     # GHz
    if pk["source_hash"] == d["build"]["source_hash"]:               # the line was written by the build the pass profiled
        assert r["valu"]["busy"] is not None and abs(r["valu"]["busy"] - busy) <= 0.02 * busy, (r["valu"], busy)
        assert r["bound"] == ("valu" if busy > r["frac"] else "hbm")


def test_whole_estep_figures_follow_from_the_committed_passes():
    """Round 6: `roofline.whole_estep` -- the vector pipes' and the HBM's share over the WHOLE E-step -- is arithmetic on the two
    committed counter files of this build: SQ_INSTS_VALU per iteration of profiles/r6_regime.json, FETCH_SIZE / WRITE_SIZE of
    the warm path's kernels in profiles/pmc_by_kernel.json.  While both are this build's the figures must be there and say
    what DESIGN.md says: neither resource is saturated (vector pipes 0.5 - 0.7, HBM 0.25 - 0.65 of peak)."""
    sys.path.insert(0, ROOT)
    import bench
    pk = json.load(open(os.path.join(ROOT, "profiles", "pmc_by_kernel.json")))
    reg = json.load(open(os.path.join(ROOT, "profiles", "r6_regime.json")))
    w = bench.whole_estep_figures("cfg3", 1e-3 * reg["bench_under_kernel_trace"]["estep_ms"])
    if pk["source_hash"] == bench.source_hash():
        lo, hi = w["hbm_frac"]
        assert 0.25 < lo < hi < 0.65 and hi < 2.0 * lo + 1e-9, w
        assert w["hbm_bytes_per_iteration"][0] > 88.8e6 * 300          # >= 300 B per node and EM iteration
    if reg["build"]["source_hash"] == bench.source_hash():
        assert 0.5 < w["valu_busy"] < 0.7, w
    other = bench.whole_estep_figures("small", 1e-3)
    assert other["valu_busy"] is None and other["hbm_frac"] is None and "another" in other["hbm_source"]
