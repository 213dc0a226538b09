"""CPU: host M-step (phylo_hmrf_amd/tree.py, mstep.py) against fixtures recorded from the reference's
_ou_param_varied_constraint / _ou_lik_varied_constraint / _check_params, plus a finite-difference check of the
analytic gradient and an optimiser sanity check (the optimum is no worse than the start)."""
import os

import numpy as np
import pytest

from phylo_hmrf_amd import mstep
from phylo_hmrf_amd.tree import PhyloTree

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["t4", "t8"])
def test_tree_and_ou_map_match_reference(tag):
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "ou_params.npz"))
    t = PhyloTree(g1[tag + "_edge_list"])
    assert np.array_equal(t.leaf_vec, g1[tag + "_leaf_vec"])
    assert np.array_equal(t.A2, g1[tag + "_A2"])
    pl = g1[tag + "_pair_list"]
    assert np.array_equal(t.leaf_vec[t.pair_a], pl[:, 0]) and np.array_equal(t.leaf_vec[t.pair_b], pl[:, 1])
    assert np.array_equal(t.pair_anc, pl[:, 2])
    assert t.n_params == int(g1[tag + "_n_params"])
    means, covars = t.mean_cov(g[tag + "_params"])
    np.testing.assert_allclose(means, g[tag + "_means"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(covars, g[tag + "_covars"], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("tag", ["t4", "t8"])
def test_objective_matches_reference(tag):
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1[tag + "_edge_list"])
    P = g[tag + "_params"]
    for c in range(g[tag + "_post"].shape[0]):
        obj = mstep.OUObjective(t, g[tag + "_post"][c], g[tag + "_obs"][c], g[tag + "_obsobsT"][c],
                                int(g[tag + "_n_samples"]), float(g[tag + "_lambda_0"]))
        for i in range(P.shape[0]):
            f = obj.value(P[i])
            np.testing.assert_allclose(f, g[tag + "_lik"][i, c], rtol=1e-11)
            np.testing.assert_allclose(obj.last_V, g[tag + "_V"][i, c], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(obj.last_mean, g[tag + "_mu"][i, c], rtol=1e-12)
    assert [mstep.check_params(t, P[i]) for i in range(P.shape[0])] == list(g[tag + "_check"])
    bad = P[0].copy()
    bad[2] = 101.0
    assert mstep.check_params(t, bad) == -1
    bad[3] = np.nan
    assert mstep.check_params(t, bad) == -2


@pytest.mark.parametrize("tag", ["t4", "t8"])
def test_analytic_gradient(tag):
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1[tag + "_edge_list"])
    obj = mstep.OUObjective(t, g[tag + "_post"][0], g[tag + "_obs"][0], g[tag + "_obsobsT"][0], 5000, 1.0)
    for p in g[tag + "_params"][[0, 2, 5]]:
        f, gr = obj.value_and_grad(p)
        num = np.zeros_like(p)
        for i in range(p.shape[0]):
            d = np.zeros_like(p)
            d[i] = 1e-6
            num[i] = (obj.value(p + d) - obj.value(p - d)) / 2e-6
        np.testing.assert_allclose(gr, num, rtol=2e-5, atol=2e-7)


def test_do_mstep_improves_the_objective_and_respects_the_box():
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1["t4_edge_list"])
    stats = {"post": g["t4_post"], "obs": g["t4_obs"], "obs*obs.T": g["t4_obsobsT"]}
    K = stats["post"].shape[0]
    cur = g["t4_params"][:K]
    rng = np.random.default_rng(1)
    params, means, covars, lik = mstep.do_mstep(t, stats, cur, cur, 5000, 1.0, 0, 0.3, 0.1, 1.0, rng, workers=1)
    assert params.shape == cur.shape and np.all(params >= 0) and np.all(params <= 100)
    for c in range(K):
        obj = mstep.OUObjective(t, stats["post"][c], stats["obs"][c], stats["obs*obs.T"][c], 5000, 1.0)
        assert lik[c] <= obj.value(np.clip(cur[c], 1e-16, 100)) + 1e-9
        # EM-time covariance = V + min_covar*I (phylo_hmrf.py:1524)
        obj.value(params[c])
        np.testing.assert_allclose(covars[c], obj.last_V + 1e-3 * np.eye(4))
        np.testing.assert_allclose(means[c], obj.last_mean)
    # workers > 1: all states in one library call on host threads (phmrf_ou_mstep) -- the same numbers as the in-process
    # Python loop over the states, to the last bit, whatever the thread count
    ref = mstep.do_mstep(t, stats, cur, cur, 5000, 1.0, 1, 0.3, 0.1, 1.0, np.random.default_rng(1), workers=1)
    for w in (2, 3, 16):
        got = mstep.do_mstep(t, stats, cur, cur, 5000, 1.0, 1, 0.3, 0.1, 1.0, np.random.default_rng(1), workers=w)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
    # ... and the fork pool behind it (PHMRF_MSTEP_NATIVE=0; used when SciPy's SLSQP entry point is not available)
    mstep.NATIVE_MSTEP[0] = False
    try:
        got = mstep.do_mstep(t, stats, cur, cur, 5000, 1.0, 1, 0.3, 0.1, 1.0, np.random.default_rng(1), workers=2)
    finally:
        mstep.NATIVE_MSTEP[0] = True
        mstep.close_pool()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tag", ["t4", "t8"])
def test_native_objective_matches_numpy(tag):
    """libphmrf_host.so (include/phmrf_host.h) against the NumPy objective it restates: value, gradient, V, mu."""
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1[tag + "_edge_list"])
    rng = np.random.default_rng(7)
    for c in range(g[tag + "_post"].shape[0]):
        args = (t, g[tag + "_post"][c], g[tag + "_obs"][c], g[tag + "_obsobsT"][c], 5000, 1.0)
        nat, ref = mstep.OUObjective(*args), mstep.OUObjective(*args, native=False)
        pts = list(g[tag + "_params"]) + [rng.uniform(1e-3, 3.0, t.n_params) for _ in range(20)]
        pts.append(np.full(t.n_params, 1e-16))            # lower corner of the box: beta <= 1e-7 branch
        for p in pts:
            f1, g1_ = nat.value_and_grad(p)
            f0, g0 = ref.value_and_grad(p)
            np.testing.assert_allclose(f1, f0, rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(g1_, g0, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(nat.last_V, ref.last_V, rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(nat.last_mean, ref.last_mean, rtol=1e-13, atol=1e-15)


def test_native_objective_hands_ill_conditioned_cases_to_numpy():
    """V singular even after 10 x min_covar (min_covar = 0 here): status 2 -> the pseudo-inverse branch runs in NumPy."""
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1["t4_edge_list"])
    args = (t, g["t4_post"][0], g["t4_obs"][0], g["t4_obsobsT"][0], 5000, 1.0)
    nat, ref = mstep.OUObjective(*args, min_covar=0.0), mstep.OUObjective(*args, min_covar=0.0, native=False)
    p = np.full(t.n_params, 1e-16)                        # every variance 0 -> V = 0
    assert nat._native_value_and_grad(p, True) is None
    f1, _ = nat.value_and_grad(p, want_grad=False)
    f0, _ = ref.value_and_grad(p, want_grad=False)
    assert np.isfinite(f0) and f1 == f0


def test_lean_slsqp_driver_equals_scipy_minimize():
    """mstep._slsqp_lean drives scipy's SLSQP core without the generic wrappers: same iterates, same optimum."""
    from scipy.optimize import minimize
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1["t4_edge_list"])
    rng = np.random.default_rng(0)
    for c in range(g["t4_post"].shape[0]):
        obj = mstep.OUObjective(t, g["t4_post"][c], g["t4_obs"][c], g["t4_obsobsT"][c], 5000, 1.0)
        x0 = rng.uniform(0.01, 1.0, t.n_params)
        lean = mstep._slsqp_lean(obj.value_and_grad, x0, mstep.LOWER, mstep.UPPER, acc=1e-6, maxiter=200)
        if lean is None:
            pytest.skip("scipy.optimize._slsqp.slsqp is not importable in this SciPy: the driver falls back to minimize()")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            res = minimize(obj.value_and_grad, x0, jac=True, method="SLSQP", bounds=[(mstep.LOWER, mstep.UPPER)] * t.n_params,
                           tol=1e-6, options={"maxiter": 200})
        assert lean[1] == res.status
        np.testing.assert_allclose(lean[0], res.x, rtol=1e-12, atol=1e-14)


def test_native_slsqp_loop_reproduces_the_python_loop():
    """phmrf_ou_slsqp (libphmrf_host.so: SciPy's Fortran SLSQP core called through its address, the objective evaluated in
    place) against the Python loop around the same core: identical iterates, hence identical results and exit modes."""
    from phylo_hmrf_amd import synthetic
    from phylo_hmrf_amd.mstep import LOWER, UPPER, OUObjective, _slsqp_lean, _slsqp_native, slsqp_entry
    from phylo_hmrf_amd.tree import PhyloTree
    if slsqp_entry() is None:
        pytest.skip("this SciPy does not expose the SLSQP core's address")
    for S, K in ((4, 6), (8, 4)):
        tree = PhyloTree(synthetic.tree_for(S))
        rng = np.random.default_rng(S)
        P = synthetic.sample_ou_params(rng, tree, K)
        mu, cv = tree.mean_cov(P)
        cv = cv + 1e-3 * np.eye(S)
        n = 20000
        lab = rng.integers(0, K, n)
        L = np.linalg.cholesky(cv)
        X = np.maximum(mu[lab] + np.einsum("nij,nj->ni", L[lab], rng.standard_normal((n, S))), 0)
        for c in range(K):
            Xc = X[lab == c]
            obj = OUObjective(tree, float(Xc.shape[0]), Xc.sum(0), Xc.T @ Xc, n, 1.0)
            x0 = 0.4 * P[c] + 0.6 * rng.random(tree.n_params)
            a = _slsqp_native(obj, x0, LOWER, UPPER)
            b = _slsqp_lean(obj.value_and_grad, np.clip(x0, LOWER, UPPER), LOWER, UPPER, acc=1e-6, maxiter=200)
            assert a is not None and b is not None
            assert a[1] == b[1]
            assert np.array_equal(a[0], b[0]), float(np.abs(a[0] - b[0]).max())


def test_native_mstep_flags_an_ill_conditioned_state_and_leaves_the_others_alone():
    """phmrf_ou_mstep: a state whose covariance is ill-conditioned (every variance 0 with min_covar = 0) comes back with
    status 2 for the caller's Python loop; the other state of the same call is fitted as usual."""
    import ctypes
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "mstep_objective.npz"))
    t = PhyloTree(g1["t4_edge_list"])
    entry = mstep.slsqp_entry()
    if entry is None:
        pytest.skip("this SciPy does not expose the SLSQP entry point")
    nt = mstep.NativeTree(t)
    K, P, S = 2, t.n_params, t.n_features
    post = np.ascontiguousarray(g["t4_post"][:2], dtype=np.float64)
    obs = np.ascontiguousarray(g["t4_obs"][:2], dtype=np.float64)
    oo = np.ascontiguousarray(g["t4_obsobsT"][:2], dtype=np.float64)
    cur = np.clip(g["t4_params"][:2], mstep.LOWER, mstep.UPPER)
    guesses = np.ascontiguousarray(np.stack([cur[0:1], np.full((1, P), 1e-16)]))          # [K, 1, P]
    params, lik, mean, V = np.zeros((K, P)), np.zeros(K), np.zeros((K, S)), np.zeros((K, S, S))
    status = np.full(K, -1, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    L = mstep.host_lib()
    L.phmrf_ou_mstep.restype = ctypes.c_int
    L.phmrf_ou_mstep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double,
                                 ctypes.c_double, dp, ctypes.c_int, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                 ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, ctypes.POINTER(ctypes.c_int)]
    st = L.phmrf_ou_mstep(ctypes.cast(ctypes.byref(nt.tables), ctypes.c_void_p), entry, K, post.ctypes.data_as(dp),
                          obs.ctypes.data_as(dp), oo.ctypes.data_as(dp), 5000.0, 1.0 / np.sqrt(5000.0), 0.0,
                          guesses.ctypes.data_as(dp), 1, cur.ctypes.data_as(dp), mstep.LOWER, mstep.UPPER, 1e-6, 200, 2,
                          params.ctypes.data_as(dp), lik.ctypes.data_as(dp), mean.ctypes.data_as(dp), V.ctypes.data_as(dp),
                          status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert st == 0 and status[1] == 2
    if status[0] == 0:                                   # (min_covar = 0 may make state 0 ill-conditioned on the way, too)
        obj = mstep.OUObjective(t, post[0], obs[0], oo[0], 5000, 1.0, min_covar=0.0)
        assert lik[0] == obj.value(params[0]) and np.array_equal(mean[0], obj.last_mean) and np.array_equal(V[0], obj.last_V)
        assert lik[0] <= obj.value(cur[0]) + 1e-9
    # bad arguments are refused
    assert L.phmrf_ou_mstep(None, entry, K, post.ctypes.data_as(dp), obs.ctypes.data_as(dp), oo.ctypes.data_as(dp), 5000.0,
                            0.01, 0.0, guesses.ctypes.data_as(dp), 1, cur.ctypes.data_as(dp), mstep.LOWER, mstep.UPPER, 1e-6,
                            200, 2, params.ctypes.data_as(dp), lik.ctypes.data_as(dp), mean.ctypes.data_as(dp),
                            V.ctypes.data_as(dp), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 1


def test_states_dealt_to_ranks_give_the_single_rank_result_bit_for_bit():
    """do_mstep(states=...): every state draws its restarts from a generator of its own (seeded from one draw of the fit's
    generator and the state's index), so fitting the states in two shares -- as two ranks do, each adding zeros for the
    other's rows -- gives exactly the arrays of fitting them all in one call (phyloHMRF._do_mstep, bench.py mstep_all)."""
    from phylo_hmrf_amd import mstep
    from phylo_hmrf_amd import synthetic
    from phylo_hmrf_amd.tree import PhyloTree
    tree = PhyloTree(synthetic.tree_for(4))
    K, S = 6, 4
    rng = np.random.default_rng(5)
    A = rng.standard_normal((K, S, S))
    post = rng.uniform(500, 3000, K)
    mu = rng.uniform(0.5, 3.0, (K, S))
    cov = np.einsum("kij,klj->kil", A, A) * 0.2 + 0.05 * np.eye(S)
    stats = {"post": post, "obs": post[:, None] * mu,
             "obs*obs.T": post[:, None, None] * (cov + mu[:, :, None] * mu[:, None, :])}
    cur = rng.uniform(0.1, 1.5, (K, tree.n_params))
    init = rng.uniform(0.1, 1.5, (K, tree.n_params))
    args = (tree, stats, cur, init, float(post.sum()), 1.0, 0, 0.3, 0.1, 1.0)
    full = mstep.do_mstep(*args, np.random.default_rng(9), workers=3)
    g0, g1 = np.random.default_rng(9), np.random.default_rng(9)
    a = mstep.do_mstep(*args, g0, workers=2, states=[0, 2, 4])
    b = mstep.do_mstep(*args, g1, workers=1, states=[1, 3, 5])
    for x, y, z in zip(full, a, b):
        assert np.array_equal(x, y + z)
    # each call advanced its generator by the one draw: the next M-step is in step on every rank
    assert g0.integers(0, 2 ** 62) == g1.integers(0, 2 ** 62)
