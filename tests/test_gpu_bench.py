"""GPU: the bench.py contract (one JSON line, last on stdout, with the keys the driver and the judge read)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "40"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_prints_one_json_line_with_the_contract_keys():
    d = _run([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["nodes_per_rank"] == [300 * 301 // 2 + 200 * 260] and d["config"]["units_per_rank"] == [2]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # bound names whichever of the two figures is the higher for the dominant kernel; achieved / peak / frac stay the HBM ones.
    # (valu.busy comes from a committed counter pass of cfg3 with this build's source hash: on this small workload it is None
    #  and says why, so the bound is the HBM one)
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "busy" in r["valu"] and "source" in r["valu"] and (r["valu"]["busy"] is not None or r["bound"] == "hbm")
    # frac is the KERNEL's fraction: bytes over the launches' own durations, full sweeps and mop-up launches apart
    full, mop = r["full_sweep"], r["mop_up"]
    assert r["own_durations_from"] and full["launches"] > 0 and full["bytes"] > 0 and full["own_ms"] > 0
    assert full["launches"] + mop["launches"] == r["launches"] and mop["launches"] >= 0
    assert abs((full["bytes"] + mop["bytes"]) / max(r["launches"], 1) - r["algorithmic_bytes_per_launch"]) <= 2
    assert abs(r["achieved"] - (full["bytes"] + mop["bytes"]) / ((full["own_ms"] + mop["own_ms"]) * 1e-3) / 1e9) <= 0.02 * r["achieved"] + 0.2
    # the class's time per step is measured on one time line: it cannot exceed the step
    ct = r["class_throughput"]
    assert 0 < ct["busy_ms_per_step"] <= d["ms_per_step"] and abs(ct["frac"] - ct["achieved"] / r["peak"]) < 1e-3
    for k in d["kernels"].values():
        assert k["busy_ms"] <= k["ms"] + 1e-3 and k["busy_ms"] <= d["ms_per_step"] * d["steps"] + 1e-3
    # every figure in the roofline object is measured in this run or says where it comes from
    assert "traffic_source" in r and (r["traffic"] is None or r["traffic"] > 0)
    # (round 6) the whole E-step's figures are the committed cfg3 passes' or absent with the reason -- never another workload's
    we = r["whole_estep"]
    assert we["valu_busy"] is None and we["hbm_frac"] is None and we["valu_source"] and we["hbm_source"]
    # the start of an E-step's labelling is on the line where the driver keeps it, and the whole fit is there under both rules
    assert "warm_start=best" in d["config"]["workload"]
    assert d["fit"]["warm_start"] == "best" and d["fit_reference_start"]["warm_start"] == "local"
    for f in (d["fit"], d["fit_reference_start"]):
        assert f["iterations"] >= 1 and f["value"] > 0 and len(f["estep_ms"]) == f["iterations"]
        assert (f["steady_ms_per_iteration"] is None) == (f["iterations"] <= 5)      # the steady cost under each start rule
    assert d["build"]["source_hash"]
    # (round 6) the stopping tolerance is on the line, and so is the number at the tolerance rounds 3 - 5 benched
    assert d["config"]["mrf_solver"]["energy_tol_ppb"] == 10000
    old = d["at_tol_1000ppb"]
    assert old["energy_tol_ppb"] == 1000 and old["steps"] == d["steps"] and old["ms_per_step"] > 0 and old["value"] > 0
    # (round 6) the same iterations through the product's own loop, and every rank's own clocks
    fs = d["fit_surface"]
    assert fs["iterations_timed"] == 2 and fs["warmup"] == 1 and fs["ms_per_step"] > 0 and len(fs["ms_per_step_by_iteration"]) == 3
    assert abs(fs["ratio_to_ms_per_step"] - fs["ms_per_step"] / d["ms_per_step"]) < 1e-3 and len(fs["cost1"]) == 3
    pr = d["per_rank"]
    assert len(pr["estep_ms"]) == 1 and abs(pr["estep_ms"][0] - d["estep_ms"]) < 1e-2 and pr["nodes"] == d["config"]["nodes_per_rank"]
    if r["kernel"] in ("strip", "fusion"):
        lim = d["roofline_limiter"]
        # device-counted work: (strip, label) pairs, cells swept once per strip visit, one unary entry per cell and label
        assert lim["bound"].startswith("instruction issue") and 0 < lim["lds"]["frac"] < 1 and lim["units"] > 0 and lim["dp_steps"] > 0
        assert 0 < lim["swept_cells"] <= lim["label_cells"] <= 315 * lim["units"]
        assert lim["single_proposal_cells"] > 0 and lim["proposal_nodes"] > 0
        assert set(lim) <= {"bound", "kernel", "lds", "units", "single_proposal_cells", "swept_cells", "label_cells", "dp_steps",
                            "proposal_nodes", "note"}          # no figures from other builds or other workloads
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "vectorised", "all_cores_upper_bound"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["vectorised"]["value"] >= c["value"] * 0.5


def test_bench_rccl_path_single_rank():
    """PHMRF_FORCE_DIST=1: the N > 1 code path (nccl process group = RCCL, all-reduce of the statistics, broadcast of the
    M-step result, barrier, max-over-ranks timing) with one rank; the line must match the plain single-GPU run's shape."""
    d = _run(["--no-cpu-baseline"], env={"PHMRF_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29519",
                                         "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["units_per_rank"] == [2]
    assert "cpu_baseline" not in d


def test_bench_weak_scaling_mode_replicates_the_workload():
    d = _run(["--no-cpu-baseline", "--scaling", "weak"])
    assert d["scaling"] == "weak" and d["config"]["units_per_rank"] == [2]


def test_bench_ab_env_runs_every_estep_under_both_values_from_the_same_labellings():
    """--ab-env (development library): both runs of an E-step start from the same labellings and parameters, so with a knob
    that leaves the labellings alone (the strip scan) they end within a solve's own spread of each other -- a block's
    expansion cut advances from one solve to the next (b->geom_phase), which is why the order of the two runs alternates --
    where two EM iterations apart differ in cost1 by percents."""
    dev = os.path.join(ROOT, "phylo_hmrf_amd", "libphmrf_dev.so")
    d = _run(["--no-cpu-baseline", "--no-fit", "--no-through-fit", "--no-kernel-timing", "--ab-env", "PHMRF_SCAN=0,1", "--ab-steps", "3"],
             env={"PHMRF_LIB": dev})
    ab = d["ab"]
    assert ab["env"] == "PHMRF_SCAN" and ab["values"] == ["0", "1"] and ab["steps"] == 3
    assert len(ab["estep_ms"]["0"]) == 3 and len(ab["estep_ms"]["1"]) == 3 and min(ab["estep_ms"]["0"] + ab["estep_ms"]["1"]) > 0
    for x, y in zip(ab["cost1"]["0"], ab["cost1"]["1"]):
        assert abs(x - y) <= 1e-3 * abs(x), (x, y)
