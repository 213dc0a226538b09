"""CPU: pin the compiled reference gco (oracle/_ref) on the reference's only known-answer material,
gco_source/example.cpp:276-338 (printed energies 250 -> 44 and 250 -> 244), and replay the committed
golden labelling fixtures.  Skipped when oracle/_ref was not built (e.g. no /root/reference)."""
import os

import numpy as np
import pytest

from oracle import gco_ref, ref_numpy as R

pytestmark = pytest.mark.skipif(not gco_ref.available(), reason="oracle/_ref/libgco_ref.so not built")
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _example_problem():
    W, H, K = 10, 5, 7
    n = W * H
    data = np.full((n, K), 10, dtype=np.intc)
    data[:25, 0] = 0
    data[25:, 5] = 0
    l = np.arange(K)
    smooth = np.minimum((l[:, None] - l[None, :]) ** 2, 4).astype(np.intc)
    return W, H, data, smooth


def test_example_cpp_general_graph_known_answer():
    W, H, data, smooth = _example_problem()
    edges = [(x + y * W, x - 1 + y * W) for y in range(H) for x in range(1, W)]
    edges += [(x + y * W, x + (y - 1) * W) for y in range(1, H) for x in range(W)]
    lab, e = gco_ref.cut_general_graph_int(np.array(edges), np.ones(len(edges)), data, smooth, n_iter=2,
                                           algorithm="expansion", return_energy=True)
    assert (e["before"], e["after"]) == (250, 44)


def test_example_cpp_spatially_varying_known_answer():
    W, H, data, smooth = _example_problem()
    edges, w = [], []
    for y in range(H):
        for x in range(1, W):
            p1, p2 = x - 1 + y * W, x + y * W
            edges.append((p1, p2)); w.append(p1 + p2)
    for y in range(1, H):
        for x in range(W):
            p1, p2 = x + (y - 1) * W, x + y * W
            edges.append((p1, p2)); w.append(p1 * p2)
    lab, e = gco_ref.cut_general_graph_int(np.array(edges), np.array(w), data, smooth, n_iter=2,
                                           algorithm="expansion", return_energy=True)
    assert (e["before"], e["after"]) == (250, 244)


@pytest.mark.parametrize("tag", ["chain", "diag", "offdiag"])
def test_replay_golden_labellings(tag):
    g = np.load(os.path.join(G, "gco_%s.npz" % tag))
    K = int(g["K"])
    w, eid = R.edge_weights_from_distance(g["edges"], 0.5)
    V = R.potts_matrix(K, float(g["beta"]))
    for alg in ("swap", "expansion"):
        for q in ("pygco", "fine"):
            lab, e = gco_ref.cut_general_graph(eid, w, -g["logprob"], V, n_iter=5000, algorithm=alg,
                                               init_labels=g["init"], quant=q, return_energy=True)
            assert np.array_equal(lab, g["labels_%s_%s" % (alg, q)])
            assert e["after"] == int(g["eint_%s_%s" % (alg, q)][1])


def test_knn_graph_fixture_is_what_the_reference_gco_gives():
    """tests/golden/knn_gco_energies.json (the general-graph parity cases of tests/test_gpu_estep.py) against the compiled
    reference: the inputs regenerate from the seeds (e_init, e_argmax) and gco's swap on the smallest case reproduces the
    recorded energies under both quantisations; gco's own EXPANSION on it -- the move maxflow.hip makes on the device --
    lands within 1e-3 of its swap (neither dominates; recorded here so that the GPU's result has both to be held against)."""
    import json
    import sys
    sys.path.insert(0, G)
    import make_golden_knn_gco as mk
    rec = json.load(open(os.path.join(G, "knn_gco_energies.json")))["cases"]
    case = [c for c in mk.CASES if c[1] == 8000][0]
    r = [c for c in rec if c["seed"] == case[0]][0]
    n, eid, w, lp, init = mk.case_inputs(*case)
    assert n == r["n"] and eid.shape[0] == r["edges"]
    np.testing.assert_allclose(R.mrf_energy(init, lp, eid, w, 1.0)[0], r["e_init"], rtol=1e-12)
    np.testing.assert_allclose(R.mrf_energy(np.argmax(lp, axis=1), lp, eid, w, 1.0)[0], r["e_argmax"], rtol=1e-12)
    V = R.potts_matrix(lp.shape[1], 1.0)
    for q in ("pygco", "fine"):
        lab = gco_ref.cut_general_graph(eid, w, -lp, V, n_iter=5000, algorithm="swap", init_labels=init, quant=q)
        np.testing.assert_allclose(R.mrf_energy(lab, lp, eid, w, 1.0)[0], r["e_" + q], rtol=1e-12)
    lab = gco_ref.cut_general_graph(eid, w, -lp, V, n_iter=5000, algorithm="expansion", init_labels=init, quant="fine")
    e_exp = R.mrf_energy(lab, lp, eid, w, 1.0)[0]
    assert abs(e_exp - r["e_fine"]) <= 1e-3 * abs(r["e_fine"]), (e_exp, r["e_fine"])
