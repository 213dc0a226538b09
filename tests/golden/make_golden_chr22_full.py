#!/usr/bin/env python3
"""BASELINE config 1 at its real block size on real data: the reference's own fit_accumulate_test (K=20, --miter 5, gco
swap through pygco's quantisation) on the FULL chr22 synteny block of example_input (683 bins, 233,586 nodes), run from
the reference itself in the build container (companion of make_golden_example.py, whose helpers and stated deviations --
chr22 only, hg38 a deterministic surrogate, smoothing filter off -- it shares).

    python tests/golden/make_golden_chr22_full.py        # needs /root/reference and `make -C oracle ref`; ~10 min

Writes tests/golden/example_chr22_full.npz: X (float32: the observations are rounded to float32 BEFORE the reference
sees them, so the fixture holds exactly what the reference was fed), len_vec, and per EM iteration the E-step inputs
(means_, _covars_, the warm-start labels) and what the reference made of them (labels as uint8, E_float, the cost_vec
row).  The edge list is NOT stored: tests rebuild it from X with the edge builder that tests/golden/grid_edges.npz pins
bit-exact on the reference's own (oracle.ref_numpy.grid_edges), which is also what this script hands the reference.
"""
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
import make_golden_example as MGE  # noqa: E402
from oracle import gco_ref, ref_numpy  # noqa: E402


def main():
    if not gco_ref.available():
        raise SystemExit("build oracle/_ref first: make -C oracle ref")
    t00 = time.time()
    mod, tmp = MG.import_reference()
    import utility
    tabs = MGE.read_species_tables()
    dfull, flfull = MGE.write_input_dir(tabs, "22", MGE.SYNTENY22)
    sizes = dfull + "/hg38.chrom.sizes"
    mvf = MG.quiet(utility.quantile_contact_vec, [22], MGE.RES, sizes, flfull, MGE.SPECIES)
    x_maxf = float(np.median(mvf[:, 6]))
    samples, len_vec, elv = MG.quiet(utility.load_data_chromosome2, [22], x_maxf, 0, MGE.RES, 8, 2, 0.0, 0, sizes, flfull,
                                     MGE.SPECIES, dfull, "golden")
    n, H, W = int(len_vec[0][0]), int(len_vec[0][3]), int(len_vec[0][4])
    assert samples.shape[0] == n == H * (H + 1) // 2 and H == W, (samples.shape, len_vec)
    # the fixture stores float32: round first, rebuild the edge list from the rounded observations with the pinned builder
    # (and check that builder once more against the reference's own on the unrounded ones)
    e_chk = ref_numpy.grid_edges(samples, H, W, True, 8)
    assert np.array_equal(e_chk, np.asarray(elv[0], dtype=np.float64)), "pinned edge builder != reference's on this block"
    X = np.float64(np.float32(samples))
    edges = ref_numpy.grid_edges(X, H, W, True, 8)
    print("block: %d bins, %d nodes, %d edges; loader %.0f s" % (H, n, edges.shape[0], time.time() - t00), flush=True)

    K, beta, beta1, m_iter = 20, 1.0, 0.5, 5
    np.random.seed(22)                                    # the reference draws from the global NumPy state
    m = MG.quiet(mod.phyloHMRF, n_components=K, run_id=0, n_samples=n, n_features=4, observation=X,
                 edge_list=MG.TREE4, len_vec=len_vec, type_id=1, branch_list=[0, 32, 20, 6, 6, 6, 12],
                 edge_list_1=[edges], cons_param=1.0, beta=beta, beta1=beta1, initial_mode=0, initial_weight=0.3,
                 initial_weight1=0.1, initial_magnitude=1.0, learning_rate=0.001, estimate_type=3, max_iter=100,
                 n_iter=5000, tol=1e-7)
    lik_single = m._ou_lik_varied_single

    def lik_single_guarded(params, X_):                   # (see make_golden_example.py: NumPy 2 raises where NumPy 1 returned NaN)
        try:
            v = lik_single(params, X_)
        except np.linalg.LinAlgError:
            return 1e10
        return v if np.isfinite(v) else 1e10

    m._ou_lik_varied_single = lik_single_guarded
    trace = []
    state = {"local": None}
    orig_mstep, orig_init = m._do_mstep, m._init

    def init_hook(X_, lengths=None):
        orig_init(X_, lengths=lengths)
        state["local"] = np.asarray(m.labels_local).copy()

    def mstep_hook(stats):
        lp = m._compute_log_likelihood(X)
        eid, w = m.edge_idList_undirected_vec[0], m.edge_weightList_undirected_vec[0]
        lab = np.int64(m.labels)
        trace.append(dict(means=m.means_.copy(), covars=m._covars_.copy(), init=np.int64(state["local"]), labels=lab,
                          efloat=np.array(ref_numpy.mrf_energy(lab, lp, eid, w, beta)),
                          efloat_init=np.array(ref_numpy.mrf_energy(np.int64(state["local"]), lp, eid, w, beta))))
        print("  iteration %d recorded (%.0f s)" % (len(trace) - 1, time.time() - t00), flush=True)
        state["local"] = np.asarray(m.labels_local).copy()
        orig_mstep(stats)

    m._init, m._do_mstep = init_hook, mstep_hook
    res = MG.quiet(m.fit_accumulate_test, X, len_vec, 0.001, "golden", m_iter)
    cost_vec = res[5]
    assert len(trace) == m_iter
    out = dict(X=np.float32(X), len_vec=np.asarray(len_vec, dtype=np.int64), K=K, beta=beta, beta1=beta1, m_iter=m_iter,
               cost_vec=cost_vec, n_edges=edges.shape[0])
    for key in trace[0]:
        arr = np.stack([t[key] for t in trace])
        out["it_" + key] = np.uint8(arr) if key in ("init", "labels") else arr
    np.savez_compressed(os.path.join(HERE, "example_chr22_full.npz"), **out)
    os.chdir(ROOT)
    for d in (dfull, tmp):
        shutil.rmtree(d, ignore_errors=True)
    print("example_chr22_full.npz written (%.0f s)" % (time.time() - t00))


if __name__ == "__main__":
    main()
