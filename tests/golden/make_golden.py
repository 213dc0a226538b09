#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (build container only).

The reference (/root/reference/phylo_hmrf.py, base.py) is Python 2.  This script converts a
scratch copy with lib2to3 in a temp dir, injects shims for the third-party APIs it imports
that no longer exist (scipy.misc.logsumexp, sklearn 0.18 gmm helpers, sklearn.base._pprint,
sklearn.externals.joblib, medpy, skimage) and a ``pygco`` module backed by the reference's own
gco-v3.0 (oracle/_ref/libgco_ref.so via oracle/gco_ref.py), imports it, and records
input/output vectors of the reference's own functions.  Neither the converted source nor
bytecode is written into the repository: only the .npz data below.

    python tests/golden/make_golden.py          # needs /root/reference and `make -C oracle ref`

Fixtures:
  G1 tree_tables.npz      leaf_vec / parent_list / pair_list / A2 of example_input/edge.1.txt and a
                          15-edge 8-leaf tree
  G2 ou_params.npz        params_vec[K,3B+2] -> means_, _covars_  (_ou_param_varied_constraint)
  G3 emission.npz         (X, means_, _covars_) -> logprob via phyloHMRF._compute_log_likelihood
                          [sklearn-0.18 density supplied by the oracle restatement; cross-checked
                          here against scipy.stats.multivariate_normal.logpdf]
  G4 posteriors_*.npz     (labels, logprob, edges, w) -> posteriors, 4 costs, stats
                          (_compute_posteriors_graph, stats lines phylo_hmrf.py:311-314),
                          estimate_type 0 and 3, with an isolated node
  G5 gco_*.npz            (unary, edges, w, V, init) -> labels, energies: swap(5000) and expansion
                          under pygco and fine quantisation, via the reference's predict() path
  G6 em_trace.npz         8 iterations of fit_accumulate_test with _do_mstep replaced by a fixed
                          parameter schedule: cost_vec, iter ids, t_labels
  G7 mstep_objective.npz  (params, stats, n, lambda_0) -> _ou_lik_varied_constraint value, V, leaf means
"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("PHMRF_REFERENCE", "/root/reference")

from oracle import gco_ref, ref_numpy  # noqa: E402


def import_reference():
    tmp = tempfile.mkdtemp(prefix="phmrf_ref_")
    for f in ("phylo_hmrf.py", "base.py", "utility.py"):
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.check_call([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", tmp],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import scipy.special
    import sklearn.base
    import sklearn.mixture
    import joblib
    misc = types.ModuleType("scipy.misc")
    misc.logsumexp = scipy.special.logsumexp
    sys.modules["scipy.misc"] = misc
    import scipy
    scipy.misc = misc
    sklearn.base._pprint = lambda params, offset=0, printer=repr: repr(params)
    sklearn.mixture.sample_gaussian = lambda *a, **k: None
    sklearn.mixture.log_multivariate_normal_density = (
        lambda X, means, covars, covariance_type="diag":
        ref_numpy.log_multivariate_normal_density_full(X, means, covars))
    sklearn.mixture.distribute_covar_matrix_to_match_covariance_type = (
        lambda tied_cv, covariance_type, n_components: np.tile(tied_cv, (n_components, 1, 1)))
    sklearn.mixture._validate_covars = lambda *a, **k: None
    import sklearn.externals as ext
    ext.joblib = joblib
    sys.modules["sklearn.externals.joblib"] = joblib
    pygco = types.ModuleType("pygco")
    pygco.cut_general_graph = gco_ref.cut_general_graph
    sys.modules["pygco"] = pygco
    for name, attrs in (("medpy", {}), ("medpy.filter", {}),
                        ("medpy.filter.smoothing", {"anisotropic_diffusion": None}),
                        ("skimage", {}),
                        ("skimage.restoration", {"denoise_tv_chambolle": None, "denoise_bilateral": None})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    sys.path.insert(0, tmp)
    os.chdir(tmp)  # the constructor writes base_mtx_*, ou_A1.txt, ou_A2.txt into CWD (:805-807, :914-917)
    mod = importlib.import_module("phylo_hmrf")
    return mod, tmp


def quiet(fn, *a, **k):
    """The reference prints heavily; silence stdout around a call."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_model(mod, X, edge_list_tree, len_vec, edge_list_vec, K, beta=1.0, beta1=0.5, estimate_type=3):
    S = X.shape[1]
    B = int(np.max(edge_list_tree))
    return quiet(mod.phyloHMRF, n_components=K, run_id=0, n_samples=X.shape[0], n_features=S,
                 observation=X, edge_list=edge_list_tree, len_vec=len_vec, type_id=1,
                 branch_list=[1.0] * B, edge_list_1=edge_list_vec, cons_param=1.0, beta=beta, beta1=beta1,
                 initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0,
                 learning_rate=0.001, estimate_type=estimate_type, max_iter=100, n_iter=5000, tol=1e-7)


def synth_block(rng, N, S, K, diagonal=True):
    """Small labelled block with Gaussian features (not the bench generator; just data)."""
    n = N * (N + 1) // 2 if diagonal else N * N
    mu = rng.uniform(0.5, 4.0, size=(K, S))
    lab = rng.integers(0, K, size=n)
    X = np.abs(mu[lab] + 0.4 * rng.standard_normal((n, S))) + 0.05
    edges = ref_numpy.grid_edges(X, N, N, diagonal, 8)
    return X, edges, lab


TREE4 = [[0, 1], [1, 2], [1, 3], [3, 4], [4, 5], [4, 6], [3, 7]]          # example_input/edge.1.txt
TREE8 = [[0, 1], [0, 2], [1, 3], [1, 4], [2, 5], [2, 6], [3, 7], [3, 8], [4, 9], [4, 10],
         [5, 11], [5, 12], [6, 13], [6, 14]]                               # balanced, 8 leaves


def main():
    if not gco_ref.available():
        raise SystemExit("build oracle/_ref first: make -C oracle ref")
    mod, tmp = import_reference()
    rng = np.random.default_rng(20191003)
    out = {}

    # ---------------- G1 / G2: tree tables and OU -> (mean, cov) --------------------------------
    g1, g2 = {}, {}
    for tag, tree in (("t4", TREE4), ("t8", TREE8)):
        S = 4 if tag == "t4" else 8
        K = 5
        X = rng.uniform(0.1, 3.0, size=(6, S))
        edges = np.array([[0, 1, 0.1], [1, 2, 0.2], [2, 3, 0.3], [3, 4, 0.1], [4, 5, 0.2]])
        len_vec = [[6, 0, 6, 3, 3, 0, 0, 0, 1, 1]]
        m = make_model(mod, X, tree, len_vec, [edges], K)
        g1[tag + "_edge_list"] = np.array(tree)
        g1[tag + "_leaf_vec"] = np.asarray(m.leaf_vec)
        g1[tag + "_parent_list"] = np.array([-1 if isinstance(p, list) else int(p) for p in m.parent_list])
        g1[tag + "_pair_list"] = np.array(m.pair_list)
        g1[tag + "_A2"] = np.asarray(m.A2)
        g1[tag + "_leaf_list_keys"] = np.array(sorted(m.leaf_list.keys()))
        g1[tag + "_leaf_list_vals"] = np.array([m.leaf_list[k] for k in sorted(m.leaf_list.keys())])
        g1[tag + "_n_params"] = np.array(m.n_params)
        B = m.branch_dim
        P = np.zeros((K, m.n_params))
        P[:, 0] = rng.uniform(0.1, 1.0, K)
        P[:, 1:1 + B] = rng.uniform(0.05, 2.0, (K, B))
        P[:, 1 + B:1 + 2 * B] = rng.uniform(0.05, 2.0, (K, B))
        P[:, 1 + 2 * B:] = rng.uniform(0.0, 4.0, (K, B + 1))
        P[1, 1] = 1e-9          # beta <= 1e-7 edge case (:999-1001)
        P[2, 1:1 + B] = 5e-8
        m.means_ = np.zeros((K, S))
        m._covars_ = np.zeros((K, S, S))
        quiet(m._ou_param_varied_constraint, P)
        g2[tag + "_params"] = P
        g2[tag + "_means"] = m.means_.copy()
        g2[tag + "_covars"] = m._covars_.copy()
        if tag == "t4":
            m4, P4 = m, P
    np.savez_compressed(os.path.join(HERE, "tree_tables.npz"), **g1)
    np.savez_compressed(os.path.join(HERE, "ou_params.npz"), **g2)

    # ---------------- G3: emission ------------------------------------------------------------
    from scipy.stats import multivariate_normal
    g3 = {}
    for tag, S, K in (("s4", 4, 20), ("s8", 8, 30)):
        n = 1500
        A = rng.standard_normal((K, S, S))
        cov = np.einsum("kij,klj->kil", A, A) * 0.2 + 2e-3 * np.eye(S)
        cov[0] = np.outer(np.ones(S), np.ones(S)) * 0.7 + 2e-3 * np.eye(S)      # near-singular: jitter only
        cov[1] = np.diag(rng.uniform(0.01, 2.0, S))
        mu = rng.uniform(0.0, 4.0, (K, S))
        X = np.abs(mu[rng.integers(0, K, n)] + 0.5 * rng.standard_normal((n, S)))
        X[:3] = 0.0
        m = types.SimpleNamespace(means_=mu, _covars_=cov, covariance_type="full")
        lp = mod.phyloHMRF._compute_log_likelihood(m, X)
        chk = np.stack([multivariate_normal(mu[k], cov[k]).logpdf(X) for k in range(K)], axis=1)
        assert np.allclose(lp, chk, rtol=1e-9, atol=1e-9), np.abs(lp - chk).max()
        g3[tag + "_X"], g3[tag + "_means"], g3[tag + "_covars"], g3[tag + "_logprob"] = X, mu, cov, lp
    np.savez_compressed(os.path.join(HERE, "emission.npz"), **g3)

    # ---------------- G4: posteriors, costs, stats --------------------------------------------
    for et in (0, 3):
        K, S, N = 6, 4, 12
        X, edges, lab_true = synth_block(rng, N, S, K, diagonal=True)
        n = X.shape[0]
        # isolate the last node: drop its edges
        keep = (edges[:, 0] != n - 1) & (edges[:, 1] != n - 1)
        edges = edges[keep]
        len_vec = [[n, 0, n, N, N, 0, 0, 0, 1, 22]]
        m = make_model(mod, X, TREE4, len_vec, [edges], K, beta=1.0, beta1=0.5, estimate_type=et)
        mu = rng.uniform(0.5, 4.0, (K, S))
        A = rng.standard_normal((K, S, S))
        cov = np.einsum("kij,klj->kil", A, A) * 0.3 + 0.3 * np.eye(S)   # wide: no softmax underflow rows
        m.means_, m._covars_ = mu, cov
        logprob = m._compute_log_likelihood(X)
        labels = lab_true.copy()
        flip = rng.random(n) < 0.2
        labels[flip] = rng.integers(0, K, flip.sum())
        post, pc, pcn, uc, c1 = quiet(m._compute_posteriors_graph, X, labels, logprob, 0)
        assert np.all(np.isfinite(post))
        stats = {"post": post.sum(axis=0), "obs": np.dot(post.T, X),
                 "obsobsT": np.einsum("ij,ik,il->jkl", post, X, X)}          # phylo_hmrf.py:311-314
        np.savez_compressed(os.path.join(HERE, "posteriors_et%d.npz" % et), X=X, edges=edges, labels=labels,
                            logprob=logprob, beta=1.0, beta1=0.5, estimate_type=et, posteriors=post,
                            pairwise_cost=pc, pairwise_cost_normalize=pcn, unary_cost=uc, cost1=c1,
                            w=m.edge_weightList_undirected_vec[0], **stats)

    # ---------------- G5: gco through the reference's predict() -------------------------------
    cases = {
        "chain": dict(N=1, S=4, K=5),
        "diag": dict(N=11, S=4, K=6),      # upper-tri block, n = 66
        "offdiag": dict(N=0, S=4, K=8),    # 40 x 50 full block
    }
    for tag, c in cases.items():
        K, S = c["K"], c["S"]
        if tag == "chain":
            n = 60
            X = np.abs(rng.standard_normal((n, S))) + 0.1
            edges = np.stack([np.arange(n - 1), np.arange(1, n), rng.uniform(0.0, 2.0, n - 1)], axis=1).astype(float)
            Hh, Ww, typ = 1, n, 0
        elif tag == "diag":
            X, edges, _ = synth_block(rng, c["N"], S, K, diagonal=True)
            n = X.shape[0]
            Hh, Ww, typ = c["N"], c["N"], 1
        else:
            Hh, Ww, typ = 40, 50, 0
            n = Hh * Ww
            mu0 = rng.uniform(0.5, 4.0, (K, S))
            lab0 = (np.arange(n) // Ww // 10 * 5 + (np.arange(n) % Ww) // 10) % K
            X = np.abs(mu0[lab0] + 0.6 * rng.standard_normal((n, S))) + 0.05
            edges = ref_numpy.grid_edges(X, Hh, Ww, False, 8)
        len_vec = [[n, 0, n, Hh, Ww, 0, 0, 0, typ, 22]]
        m = make_model(mod, X, TREE4, len_vec, [edges], K, beta=1.0, beta1=0.5, estimate_type=3)
        mu = rng.uniform(0.5, 4.0, (K, S))
        A = rng.standard_normal((K, S, S))
        m.means_, m._covars_ = mu, np.einsum("kij,klj->kil", A, A) * 0.3 + 0.3 * np.eye(S)
        init = rng.integers(0, K, n).astype(np.int64)
        m.labels_local = init.copy()
        m.labels = init.copy()
        labels_ref, logprob = quiet(m.predict, X, 0)          # the reference's own E-step labelling call
        w = m.edge_weightList_undirected_vec[0]
        eid = m.edge_idList_undirected_vec[0]
        V = m.edge_potential
        rec = dict(X=X, edges=edges, w=w, logprob=logprob, init=init, beta=1.0, K=K, labels_swap_pygco=labels_ref)
        for alg in ("swap", "expansion"):
            for q in ("pygco", "fine"):
                lab, e = gco_ref.cut_general_graph(eid, w, -logprob, V, n_iter=5000, algorithm=alg,
                                                   init_labels=init, quant=q, return_energy=True)
                rec["labels_%s_%s" % (alg, q)] = lab
                rec["eint_%s_%s" % (alg, q)] = np.array([e["before"], e["after"], e["data"], e["smooth"]])
                rec["efloat_%s_%s" % (alg, q)] = np.array(ref_numpy.mrf_energy(lab, logprob, eid, w, 1.0))
        assert np.array_equal(rec["labels_swap_pygco"], labels_ref)
        rec["efloat_init"] = np.array(ref_numpy.mrf_energy(init, logprob, eid, w, 1.0))
        np.savez_compressed(os.path.join(HERE, "gco_%s.npz" % tag), **rec)

    # ---------------- G6: EM driver bookkeeping trace -----------------------------------------
    K, S = 4, 4
    blocks = []
    for N in (9, 7):
        blocks.append(synth_block(rng, N, S, K, diagonal=True))
    X = np.concatenate([b[0] for b in blocks])
    len_vec, off = [], 0
    for i, (xb, eb, _) in enumerate(blocks):
        N = (9, 7)[i]
        len_vec.append([xb.shape[0], off, off + xb.shape[0], N, N, 0, 0, i, 1, 22])
        off += xb.shape[0]
    m = make_model(mod, X, TREE4, len_vec, [b[1] for b in blocks], K, beta=1.0, beta1=0.5, estimate_type=3)
    B = m.branch_dim
    n_it = 8
    sched = np.zeros((n_it + 1, K, m.n_params))
    sched[:, :, 0] = rng.uniform(0.2, 0.8, (n_it + 1, K))
    sched[:, :, 1:1 + 2 * B] = rng.uniform(0.2, 1.5, (n_it + 1, K, 2 * B))
    sched[:, :, 1 + 2 * B:] = rng.uniform(0.3, 3.5, (n_it + 1, K, B + 1))
    init_label = np.concatenate([b[2] for b in blocks]).astype(np.int64)
    state = {"it": 0}

    def fake_init(X_, lengths=None):                         # replaces k-means + SLSQP init (:205-264)
        m.startprob_ = np.full(K, 1.0 / K)
        m.transmat_ = np.full((K, K), 1.0 / K)
        m.params_vec1 = sched[0].copy()
        m.init_ou_params = sched[0].copy()
        m.means_ = np.zeros((K, S))
        m._covars_ = np.zeros((K, S, S))
        m._ou_param_varied_constraint(sched[0])
        m._covars_ = m._covars_ + 1e-3 * np.eye(S)           # EM-time covariances carry 2e-3 (:1522-1524)
        m.labels = init_label.copy()
        m.labels_local = init_label.copy()

    def fake_mstep(stats):                                   # fixed parameter schedule instead of SLSQP
        state["it"] += 1
        m.params_vec1 = sched[state["it"]].copy()
        m._ou_param_varied_constraint(m.params_vec1)
        m._covars_ = m._covars_ + 1e-3 * np.eye(S)

    m._init = fake_init
    m._do_mstep = fake_mstep
    res = quiet(m.fit_accumulate_test, X, len_vec, 1e-3, "golden", n_it)
    params_vec, params_vec1, params_vecList, it1, it2, cost_vec, t_labels = res
    np.savez_compressed(os.path.join(HERE, "em_trace.npz"), X=X, sched=sched, init_label=init_label,
                        len_vec=np.array(len_vec), edges0=blocks[0][1], edges1=blocks[1][1],
                        params_vec=params_vec, params_vec1=params_vec1, params_vecList=params_vecList,
                        iter_id1=it1, iter_id2=it2, cost_vec=cost_vec, t_labels=t_labels,
                        final_means=m.means_, final_covars=m._covars_, beta=1.0, beta1=0.5, threshold=1e-3,
                        m_iter=n_it)
    # ---------------- G7: M-step objective (_ou_lik_varied_constraint, phylo_hmrf.py:1038-1138) -------------
    g7 = {}
    for tag, tree, S in (("t4", TREE4, 4), ("t8", TREE8, 8)):
        K = 3
        Xs = np.abs(rng.standard_normal((400, S))) + rng.uniform(0, 3, S)
        edges = np.stack([np.arange(399), np.arange(1, 400), rng.uniform(0, 1, 399)], 1)
        mm = make_model(mod, Xs, tree, [[400, 0, 400, 1, 400, 0, 0, 0, 0, 1]], [edges], K)
        gam = rng.random((400, K))
        gam /= gam.sum(1, keepdims=True)
        mm.stats = {"post": gam.sum(0), "obs": gam.T @ Xs, "obs*obs.T": np.einsum("ij,ik,il->jkl", gam, Xs, Xs)}
        mm.n_samples = 5000
        mm.init_ou_params = rng.uniform(0.1, 2.0, (K, mm.n_params))
        P = rng.uniform(0.05, 3.0, (6, mm.n_params))
        P[1, 1] = 1e-9                                   # beta <= 1e-7 branch
        lik = np.zeros((6, K))
        V = np.zeros((6, K, S, S))
        mu = np.zeros((6, K, S))
        for i in range(6):
            for c in range(K):
                lik[i, c] = quiet(mm._ou_lik_varied_constraint, P[i], c)
                V[i, c] = mm.cv_mtx
                mu[i, c] = mm.values[mm.leaf_vec, 0]
        g7.update({tag + "_params": P, tag + "_lik": lik, tag + "_V": V, tag + "_mu": mu, tag + "_post": mm.stats["post"],
                   tag + "_obs": mm.stats["obs"], tag + "_obsobsT": mm.stats["obs*obs.T"], tag + "_n_samples": 5000,
                   tag + "_lambda_0": 1.0, tag + "_check": np.array([mm._check_params(P[i]) for i in range(6)])})
    np.savez_compressed(os.path.join(HERE, "mstep_objective.npz"), **g7)
    os.chdir(ROOT)
    shutil.rmtree(tmp, ignore_errors=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
