"""Host M-step: per state, fit the 3B+2 Ornstein-Uhlenbeck parameters to the sufficient statistics.

Semantics of the reference (phylo_hmrf.py:1500-1528 `_do_mstep`, :1327-1403 `_ou_optimize2(_unit)`,
:1038-1138 `_ou_lik_varied_constraint`, :1405-1425 `_check_params`):

    minimise  f(p) = N_c * log(det V + 1e-16) / n + tr(V^-1 S_w) / n + lambda_0 / sqrt(n) * |p|^2
    V   = cov_OU(p) + min_covar * I
    S_w = obs*obs.T - obs mu^T - mu obs^T + mu mu^T N_c            (N_c = post[c], mu = leaf means)
    s.t. 1e-16 <= p_i <= 100   for every parameter (:1365-1366), SciPy SLSQP, tol 1e-6,
    start = a1 * init_ou_params + a2 * current + (1 - a1 - a2) * random     (:1378-1380)

What differs from the reference is speed only (once the E-step takes milliseconds the K
SLSQP runs cap the EM rate): the objective comes with its ANALYTIC gradient (reverse sweep over the tree), the
box is passed as `bounds`, the K independent states are solved in a fork pool, and value + gradient are evaluated by
the native host helper `libphmrf_host.so` (include/phmrf_host.h, csrc/ou_host.cpp; ~70x faster per evaluation),
and SciPy's SLSQP core is driven directly (`_slsqp_lean`: same routine and calling sequence as `minimize`, without
its generic wrappers).
The NumPy implementation below stays as the fall-back for an ill-conditioned V (pseudo-inverse branch of the
reference) and as the oracle of the native one (tests/test_mstep.py).
"""
import ctypes
import atexit
import multiprocessing as mp
import os
import warnings

import numpy as np
from scipy.optimize import minimize

_HOST_LIB = None


class _TreeTables(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("S", ctypes.c_int32), ("n_pairs", ctypes.c_int32), ("n_order", ctypes.c_int32),
                ("parent", ctypes.POINTER(ctypes.c_int32)), ("order", ctypes.POINTER(ctypes.c_int32)),
                ("leaf_vec", ctypes.POINTER(ctypes.c_int32)), ("pair_a", ctypes.POINTER(ctypes.c_int32)),
                ("pair_b", ctypes.POINTER(ctypes.c_int32)), ("pair_anc", ctypes.POINTER(ctypes.c_int32)),
                ("A2", ctypes.POINTER(ctypes.c_double))]


def host_lib():
    """libphmrf_host.so (built by csrc/Makefile / __graft_entry__.build()); raises if it is missing."""
    global _HOST_LIB
    if _HOST_LIB is None:
        # PHMRF_HOST_LIB: a development build, e.g. the AddressSanitizer one (make -C phylo_hmrf_amd/csrc asan)
        path = os.environ.get("PHMRF_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libphmrf_host.so")
        if not os.path.exists(path):
            raise RuntimeError("libphmrf_host.so is missing: run `make -C phylo_hmrf_amd/csrc` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        L.phmrf_ou_objective.argtypes = [ctypes.POINTER(_TreeTables), dp, ctypes.c_double, dp, dp, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_double, dp, dp, dp, dp]
        L.phmrf_ou_objective.restype = ctypes.c_int
        L.phmrf_host_version.restype = ctypes.c_int
        ipt = ctypes.POINTER(ctypes.c_int)
        L.phmrf_ou_slsqp.argtypes = [ctypes.POINTER(_TreeTables), ctypes.c_void_p, ctypes.c_double, dp, dp, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int, dp, ipt, ipt]
        L.phmrf_ou_slsqp.restype = ctypes.c_int
        _HOST_LIB = L
    return _HOST_LIB


_SLSQP_ENTRY = [None, False]        # [address of SciPy's Fortran SLSQP core, looked up already]


def slsqp_entry():
    """The address of SciPy's own SLSQP routine (the Fortran core behind scipy.optimize.minimize(method='SLSQP')), from
    the f2py object's `_cpointer` capsule; None if this SciPy does not expose it (the callers then keep the Python loop).
    Tied to the calling sequence of SciPy 1.12 - 1.15 (checked against the f2py signature string)."""
    if not _SLSQP_ENTRY[1]:
        _SLSQP_ENTRY[1] = True
        try:
            from scipy.optimize._slsqp import slsqp
            sig = (slsqp.__doc__ or "").split("\n")[0].replace(" ", "")
            want = ("slsqp(m,meq,x,xl,xu,f,c,g,a,acc,iter,mode,w,jw,alpha,f0,gs,h1,h2,h3,h4,t,t0,tol,iexact,incons,ireset,"
                    "itermx,line,n1,n2,n3,[la,n,l_w,l_jw])")
            ver = tuple(int(x) for x in _scipy_version().split(".")[:2] if x.isdigit())
            if sig == want and (1, 12) <= ver <= (1, 15):        # (the calling sequence this was written and tested for)
                api = ctypes.pythonapi
                api.PyCapsule_GetName.restype = ctypes.c_char_p
                api.PyCapsule_GetName.argtypes = [ctypes.py_object]
                api.PyCapsule_GetPointer.restype = ctypes.c_void_p
                api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
                cap = slsqp._cpointer
                _SLSQP_ENTRY[0] = api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap))
        except Exception:
            _SLSQP_ENTRY[0] = None
    return _SLSQP_ENTRY[0]


class NativeTree(object):
    """The tree tables in the layout of `phmrf_tree_tables` (keeps the arrays alive)."""

    def __init__(self, tree):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self._arr = [i32(tree.parent), i32(tree.order), i32(tree.leaf_vec), i32(tree.pair_a), i32(tree.pair_b),
                     i32(tree.pair_anc), np.ascontiguousarray(tree.A2, dtype=np.float64)]
        ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        a = self._arr
        self.tables = _TreeTables(tree.node_num, tree.n_features, len(tree.pair_a), len(tree.order), ip(a[0]), ip(a[1]),
                                  ip(a[2]), ip(a[3]), ip(a[4]), ip(a[5]),
                                  a[6].ctypes.data_as(ctypes.POINTER(ctypes.c_double)))

SMALL_EPS = 1e-16          # phylo_hmrf.py:49
LOWER, UPPER = SMALL_EPS, 100.0


def check_params(tree, params):
    """`_check_params` (phylo_hmrf.py:1405-1425): 1 ok, -1 out of bounds, -2 out of bounds with NaN."""
    _, beta, lam, theta = tree.split(params)
    ok1 = (beta >= 0) & (beta <= 100) & (lam >= 0) & (lam <= 100)
    ok2 = (theta >= -100) & (theta <= 100)
    if ok1.sum() < tree.branch_dim or ok2.sum() < tree.branch_dim + 1:
        return -2 if np.isnan(np.asarray(params)[1:]).any() else -1
    return 1


def _well_conditioned(V):
    """np.linalg.cond(V) < 1/eps (phylo_hmrf.py:1109) for the symmetric V, via its eigenvalues."""
    ev = np.abs(np.linalg.eigvalsh(V))
    lo = ev.min()
    return bool(np.isfinite(ev).all() and lo > 0 and ev.max() / lo < 1.0 / np.finfo(float).eps)


class OUObjective(object):
    """f(p) and its gradient for one state.  post: N_c; obs: [S]; obsobsT: [S,S]."""

    def __init__(self, tree, post, obs, obsobsT, n_samples, lambda_0, min_covar=1e-3, native=True):
        self.native = NativeTree(tree) if native else None
        self._nbuf = None
        self.t = tree
        self.post = float(post)
        self.obs = np.asarray(obs, dtype=np.float64)
        self.oo = np.asarray(obsobsT, dtype=np.float64)
        self.n = float(n_samples)
        self.reg = float(lambda_0) / np.sqrt(self.n)                                  # :1100-1102, :1113
        self.min_covar = min_covar
        self.last_V = None
        self.last_mean = None

    def _model(self, p):
        t = self.t
        mean, var, e, ratio = t.node_moments(p)
        _, beta, lam, theta = t.split(p)
        beta_full = np.concatenate([[0.0], beta])
        s1 = t.A2 @ beta_full
        ex = np.exp(-s1)
        S = t.n_features
        cov = np.zeros((S, S))
        s2 = var[t.pair_anc] * ex
        cov[t.pair_a, t.pair_b] = s2
        cov[t.pair_b, t.pair_a] = s2
        cov[np.arange(S), np.arange(S)] = var[t.leaf_vec]
        return mean, var, e, ratio, ex, s2, cov

    def value(self, p):
        return self.value_and_grad(p, want_grad=False)[0]

    def value_and_grad(self, p, want_grad=True):
        if self.native is not None:
            out = self._native_value_and_grad(p, want_grad)
            if out is not None:
                return out
        return self._numpy_value_and_grad(p, want_grad)

    def _native_value_and_grad(self, p, want_grad):
        """libphmrf_host.so; None when V stays ill-conditioned (the NumPy path then takes the pseudo-inverse)."""
        nb = self._nbuf
        if nb is None:                      # buffers and pointers once per objective: the call itself is ~5 us
            t = self.t
            S = t.n_features
            dp = ctypes.POINTER(ctypes.c_double)
            nb = self._nbuf = {
                "fn": host_lib().phmrf_ou_objective, "tab": ctypes.byref(self.native.tables),
                "p": np.zeros(t.n_params), "g": np.zeros(t.n_params), "V": np.zeros((S, S)), "mu": np.zeros(S),
                "obs": np.ascontiguousarray(self.obs), "oo": np.ascontiguousarray(self.oo), "f": ctypes.c_double(0.0)}
            for k in ("p", "g", "V", "mu", "obs", "oo"):
                nb[k + "_ptr"] = nb[k].ctypes.data_as(dp)
            nb["f_ref"] = ctypes.byref(nb["f"])
        nb["p"][:] = p
        st = nb["fn"](nb["tab"], nb["p_ptr"], self.post, nb["obs_ptr"], nb["oo_ptr"], self.n, self.reg, self.min_covar,
                      nb["f_ref"], nb["g_ptr"] if want_grad else None, nb["V_ptr"], nb["mu_ptr"])
        if st == 2:
            return None
        if st != 0:
            raise RuntimeError("phmrf_ou_objective failed with status %d" % st)
        self.last_V, self.last_mean = nb["V"].copy(), nb["mu"].copy()
        return nb["f"].value, (nb["g"].copy() if want_grad else None)

    def _numpy_value_and_grad(self, p, want_grad=True):
        t = self.t
        p = np.asarray(p, dtype=np.float64)
        S = t.n_features
        mean, var, e, ratio, ex, s2, cov = self._model(p)
        V = cov + self.min_covar * np.eye(S)                                          # :1090
        mu = mean[t.leaf_vec]
        # ill-conditioned V: add min_covar*I up to 10 times, then fall back to the pseudo-inverse (:1108-1133)
        cnt = 0
        well = _well_conditioned(V)
        while not well and cnt < 10:
            V = V + self.min_covar * np.eye(S)
            cnt += 1
            well = _well_conditioned(V)
        Vi = np.linalg.inv(V) if well else np.linalg.pinv(V)
        om = np.outer(self.obs, mu)
        Sw = self.oo - om - om.T + np.outer(mu, mu) * self.post                       # :1093-1097
        detV = np.linalg.det(V)
        f = self.post * np.log(detV + SMALL_EPS) / self.n + np.sum(Vi * Sw) / self.n + self.reg * float(p @ p)
        self.last_V, self.last_mean = V, mu                                           # self.cv_mtx / self.values (:1135-1136)
        if not want_grad:
            return f, None
        # ---- reverse sweep ------------------------------------------------------------------------------
        G = (self.post * (detV / (detV + SMALL_EPS)) * Vi - Vi @ Sw @ Vi) / self.n    # df/dV (symmetric)
        gmu = 2.0 * (Vi @ (self.post * mu - self.obs)) / self.n                       # df/dmu
        N, B = t.node_num, t.branch_dim
        gvar = np.zeros(N)
        gmean = np.zeros(N)
        gbeta_full = np.zeros(N)
        ge = np.zeros(N)
        gratio = np.zeros(N)
        gtheta = np.zeros(N)
        gvar[t.leaf_vec] += np.diag(G)
        gmean[t.leaf_vec] += gmu
        gpair = 2.0 * G[t.pair_a, t.pair_b]                                           # V_ab appears twice
        np.add.at(gvar, t.pair_anc, gpair * ex)
        gbeta_full += t.A2.T @ (-gpair * s2)
        for i in reversed(t.order):
            pa = t.parent[i]
            gvar[pa] += gvar[i] * e[i] ** 2
            gratio[i] += gvar[i] * (1.0 - e[i] ** 2)
            ge[i] += gvar[i] * (2.0 * e[i] * (var[pa] - ratio[i]))
            gmean[pa] += gmean[i] * e[i]
            gtheta[i] += gmean[i] * (1.0 - e[i])
            ge[i] += gmean[i] * (mean[pa] - p[1 + 2 * B + i])
        _, beta, lam, theta = t.split(p)
        gb = gbeta_full[1:] - ge[1:] * e[1:]
        ok = beta > 1e-07
        safe_beta = np.where(ok, beta, 1.0)
        glam = np.where(ok, gratio[1:] / (2.0 * safe_beta), 0.0)
        gb = gb + np.where(ok, -gratio[1:] * lam / (2.0 * safe_beta ** 2), 0.0)
        roots = np.flatnonzero(t.parent < 0)
        gtheta[roots] += gmean[roots]
        g = np.concatenate([[gvar[roots].sum()], gb, glam, gtheta]) + 2.0 * self.reg * p
        return f, g


class OUObjectiveSingle(OUObjective):
    """`_ou_lik_varied_single` (phylo_hmrf.py:1246-1325): log det V + tr(V^-1 S) on one cluster's sample moments,
    no ridge term; used by the initialisation (:1427-1498)."""

    def __init__(self, tree, X, min_covar=1e-3):
        X = np.asarray(X, dtype=np.float64)
        n = X.shape[0]
        OUObjective.__init__(self, tree, 1.0, X.mean(axis=0), X.T @ X / n, 1.0, 0.0, min_covar)

    @classmethod
    def from_moments(cls, tree, mean, second_moment, min_covar=1e-3):
        """The same objective from the cluster's sample mean and X^T X / n -- all `_ou_lik_varied_single` takes from the
        observations (:1290-1293) -- e.g. as accumulated on the device (phmrf_kmeans_moments)."""
        obj = cls.__new__(cls)
        OUObjective.__init__(obj, tree, 1.0, np.asarray(mean, dtype=np.float64), np.asarray(second_moment, dtype=np.float64),
                             1.0, 0.0, min_covar)
        return obj


def _slsqp_lean(fun_and_grad, x0, lower, upper, acc=1e-6, maxiter=200):
    """SciPy's SLSQP core (scipy.optimize._slsqp.slsqp, the routine `minimize(method="SLSQP")` drives) for a problem
    with bounds only, driven without the generic wrappers (ScalarFunction, bound clipping, constraint plumbing), which
    cost more than the native objective itself.  Same algorithm, same calling sequence as scipy 1.15
    `_minimize_slsqp`; returns (x, exit_mode) or None when that private entry point is not available."""
    try:
        from scipy.optimize._slsqp import slsqp
    except Exception:
        return None
    n = x0.shape[0]
    n1 = n + 1
    m = meq = 0
    la = 1
    mineq = m - meq + n1 + n1
    len_w = ((3 * n1 + m) * (n1 + 1) + (n1 - meq + 1) * (mineq + 2) + 2 * mineq + (n1 + mineq) * (n1 - meq) + 2 * meq + n1
             + ((n + 1) * n) // 2 + 2 * m + 3 * n + 3 * n1 + 1)
    w = np.zeros(len_w)
    jw = np.zeros(mineq)
    x = np.clip(np.asarray(x0, dtype=float), lower, upper)
    xl = np.full(n, lower, dtype=float)
    xu = np.full(n, upper, dtype=float)
    mode = np.array(0, int)
    acc_a = np.array(acc, float)
    majiter = np.array(maxiter, int)
    st_f = [np.array(0, float) for _ in range(10)]      # alpha f0 gs h1 h2 h3 h4 t t0 tol
    st_i = [np.array(0, int) for _ in range(8)]         # iexact incons ireset itermx line n1 n2 n3
    c = np.zeros(la)
    a = np.zeros((la, n1))

    def evaluate(xx):                                    # gh11403: SLSQP may leave the box by an ulp or two
        return fun_and_grad(np.clip(xx, lower, upper))

    fx, gr = evaluate(x)
    g = np.append(gr, 0.0)
    while True:
        slsqp(m, meq, x, xl, xu, fx, c, g, a, acc_a, majiter, mode, w, jw, *st_f, *st_i)
        if mode == 1:
            fx, gr = evaluate(x)
        elif mode == -1:
            _, gr = evaluate(x)
            g = np.append(gr, 0.0)
        if abs(mode) != 1:
            break
    return x, int(mode)


def _slsqp_native(obj, x0, lower, upper, acc=1e-6, maxiter=200):
    """The same SLSQP run with the loop in libphmrf_host.so (phmrf_ou_slsqp): SciPy's Fortran core called through its
    address, the objective evaluated in place -- identical iterates, no interpreter in the loop.  -> (x, exit_mode), or
    None when the entry point is unavailable, the objective is not the native one, or an evaluation met an
    ill-conditioned covariance (the Python loop then repeats the state and takes the pseudo-inverse path)."""
    entry = slsqp_entry()
    if entry is None or obj.native is None:
        return None
    dp = ctypes.POINTER(ctypes.c_double)
    n = obj.t.n_params
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    obs, oo = np.ascontiguousarray(obj.obs), np.ascontiguousarray(obj.oo)
    x = np.zeros(n)
    mode, n_eval = ctypes.c_int(0), ctypes.c_int(0)
    st = host_lib().phmrf_ou_slsqp(ctypes.byref(obj.native.tables), entry, obj.post, obs.ctypes.data_as(dp),
                                   oo.ctypes.data_as(dp), obj.n, obj.reg, obj.min_covar, x0.ctypes.data_as(dp), lower, upper,
                                   acc, int(maxiter), x.ctypes.data_as(dp), ctypes.byref(mode), ctypes.byref(n_eval))
    if st != 0:
        return None
    return x, int(mode.value)


_LEAN_OK = [True]      # cleared (per process) the first time the private SLSQP core rejects our call
NATIVE_LOOP = [os.environ.get("PHMRF_MSTEP_NATIVE_LOOP", "1") != "0"]      # the SLSQP loop in libphmrf_host.so (default)


def _solve_state(args):
    tree, post, obs, oo, n_samples, lambda_0, guesses, init_params = args
    obj = OUObjective(tree, post, obs, oo, n_samples, lambda_0)
    bounds = [(LOWER, UPPER)] * tree.n_params
    flag, flag1, params1 = 0, -1, init_params
    for guess in guesses:                                                             # retry loop (:1332-1342)
        x0 = np.clip(guess, LOWER, UPPER)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                lean = _slsqp_native(obj, x0, LOWER, UPPER, acc=1e-6, maxiter=200) if NATIVE_LOOP[0] else None
                if lean is None and _LEAN_OK[0]:
                    try:
                        lean = _slsqp_lean(obj.value_and_grad, x0, LOWER, UPPER, acc=1e-6, maxiter=200)
                    except (TypeError, ValueError, AttributeError, SystemError) as err:
                        # the private SciPy entry point does not have the signature this was written for (SciPy 1.15):
                        # use the public wrapper from now on instead of failing every retry silently
                        _LEAN_OK[0] = False
                        warnings.warn("scipy.optimize._slsqp.slsqp is not callable as expected (%s: %s); the M-step "
                                      "uses scipy.optimize.minimize from here on" % (type(err).__name__, err))
                if lean is not None:
                    params1 = lean[0]
                else:
                    res = minimize(obj.value_and_grad, x0, jac=True, method="SLSQP", bounds=bounds, tol=1e-6,
                                   options={"disp": False, "maxiter": 200})
                    params1 = res.x
        except Exception:                                                             # (:1386-1392)
            continue
        flag = check_params(tree, params1)
        flag1 = flag
        if flag > 0:
            break
    if not (flag > 0 and flag1 > 0):                                                  # (:1346-1349)
        params1 = np.array(init_params, dtype=np.float64)
    lik = obj.value(params1)
    return params1, lik, obj.last_mean.copy(), obj.last_V.copy()


NATIVE_MSTEP = [os.environ.get("PHMRF_MSTEP_NATIVE", "1") != "0"]      # all K states in ONE library call, on host threads


def _mstep_native(tree, tasks, n_samples, lambda_0, n_threads, min_covar=1e-3):
    """All states of an M-step in one call of libphmrf_host.so (`phmrf_ou_mstep`): what `_solve_state` does per state --
    SLSQP from the start points in turn, `_check_params`, fall-back to the initial parameters, the objective at the
    result -- on host threads, no interpreter, no pickling, no worker processes (2.5 - 3 ms instead of 4.4 - 6.2 ms per EM
    iteration at K = 20; a fork pool costs ~0.1 ms per task to feed and a sleeping worker's wake-up on top).  A state whose
    covariance turns ill-conditioned comes back flagged and is repeated by `_solve_state` (pseudo-inverse path).
    -> the list `_solve_state` would return, or None when the native loop is unavailable."""
    entry = slsqp_entry() if NATIVE_LOOP[0] else None
    if entry is None or not _native_self_check():
        return None
    nt = NativeTree(tree)
    K, P, S = len(tasks), tree.n_params, tree.n_features
    R = len(tasks[0][6])
    post = np.ascontiguousarray([float(t[1]) for t in tasks], dtype=np.float64)
    obs = np.ascontiguousarray([np.asarray(t[2], dtype=np.float64) for t in tasks])
    oo = np.ascontiguousarray([np.asarray(t[3], dtype=np.float64) for t in tasks])
    guesses = np.ascontiguousarray([[np.asarray(g, dtype=np.float64) for g in t[6]] for t in tasks])
    init = np.ascontiguousarray([np.asarray(t[7], dtype=np.float64) for t in tasks])
    assert obs.shape == (K, S) and oo.shape == (K, S, S) and guesses.shape == (K, R, P) and init.shape == (K, P)
    params, lik = np.zeros((K, P)), np.zeros(K)
    mean, V = np.zeros((K, S)), np.zeros((K, S, S))
    status = np.zeros(K, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    L = host_lib()
    if not getattr(L, "_mstep_ready", False):
        L.phmrf_ou_mstep.restype = ctypes.c_int
        L.phmrf_ou_mstep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, dp, dp, dp, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, dp, ctypes.c_int, dp, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp,
                                     ctypes.POINTER(ctypes.c_int)]
        L._mstep_ready = True
    reg = float(lambda_0) / np.sqrt(float(n_samples))
    st = L.phmrf_ou_mstep(ctypes.cast(ctypes.byref(nt.tables), ctypes.c_void_p), entry, K, post.ctypes.data_as(dp),
                          obs.ctypes.data_as(dp), oo.ctypes.data_as(dp), float(n_samples), reg, float(min_covar),
                          guesses.ctypes.data_as(dp), R, init.ctypes.data_as(dp), LOWER, UPPER, 1e-6, 200, int(n_threads),
                          params.ctypes.data_as(dp), lik.ctypes.data_as(dp), mean.ctypes.data_as(dp), V.ctypes.data_as(dp),
                          status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if st != 0:
        return None
    out = []
    for c in range(K):
        if status[c] != 0:
            out.append(_solve_state(tasks[c]))
        else:
            out.append((params[c].copy(), float(lik[c]), mean[c].copy(), V[c].copy()))
    return out


_POOL = None
_POOL_SIZE = 0
_WARNED_SERIAL = False


def _pool(workers):
    global _POOL, _POOL_SIZE
    if workers <= 1:
        return None
    if _POOL is None:
        # Forked ONCE.  The callers that own a GPU create the pool before anything initialises HIP (phyloHMRF.__init__
        # before its first block, bench.py and the CLI before importing torch): a fork() of a process whose HIP / RCCL
        # runtime and block threads are live is not safe.  (A fork server would lift that ordering, but it starts by
        # exec-ing an interpreter, which rocprofv3 wraps and breaks, and which the GPU pool forbids once HIP is up.)
        # An existing pool serves every later request, whatever its size (fewer tasks than workers leave some idle,
        # more tasks queue): it is never re-forked; close_pool() ends it.
        from . import _lib
        if _lib._lib is not None:
            # HIP is (or may be) up in this process already: no fork from here.  The states are then fitted one after
            # the other in this process (same results, K x the time); callers that want the pool create it first
            # (phylo_hmrf.py, bench.py: mstep._pool(K) before anything touches the GPU)
            global _WARNED_SERIAL
            if not _WARNED_SERIAL:
                warnings.warn("M-step worker pool requested after the GPU library was loaded: fitting the states "
                              "serially (fork after HIP initialisation is not safe)", RuntimeWarning)
                _WARNED_SERIAL = True
            return None
        _POOL = mp.get_context("fork").Pool(workers)
        _POOL_SIZE = workers
        atexit.register(close_pool)
    return _POOL


def close_pool():
    global _POOL
    if _POOL is not None:
        _POOL.terminate()
        _POOL.join()
        _POOL = None


_NATIVE_CHECKED = [None]      # None: not checked yet; True / False: the self-check's verdict


def _native_self_check():
    """Once per process: one small fixed state fitted by the native loop (SciPy's Fortran SLSQP core called through its raw
    address from libphmrf_host.so) and by the Python loop over the same routine (`_slsqp_lean`).  The f2py signature string
    and the SciPy version gate in `slsqp_entry` say the calling sequence SHOULD match; this says it DOES in this build of
    SciPy (integer width, argument order, no hidden state): the iterates must agree to the last bit, otherwise the native
    path is switched off for the process and the M-step runs the Python loop."""
    if _NATIVE_CHECKED[0] is None:
        ok = False
        try:
            from .tree import PhyloTree
            tree = PhyloTree([[0, 1], [1, 2], [1, 3], [3, 4], [4, 5], [4, 6], [3, 7]])
            rng = np.random.default_rng(12345)
            S = tree.n_features
            A = rng.standard_normal((S, S))
            mu = rng.uniform(0.5, 3.0, S)
            post = 1000.0
            obj = OUObjective(tree, post, post * mu, post * (A @ A.T * 0.2 + 0.05 * np.eye(S) + np.outer(mu, mu)), 5000.0, 1.0)
            x0 = np.clip(rng.uniform(0.1, 1.5, tree.n_params), LOWER, UPPER)
            nat = _slsqp_native(obj, x0, LOWER, UPPER, acc=1e-6, maxiter=200)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = _slsqp_lean(obj.value_and_grad, x0, LOWER, UPPER, acc=1e-6, maxiter=200)
            ok = nat is not None and ref is not None and nat[1] == ref[1] and np.array_equal(nat[0], ref[0])
            if nat is not None and ref is not None and not ok:
                warnings.warn("the native SLSQP loop does not reproduce scipy.optimize._slsqp.slsqp on the self-check problem "
                              "(SciPy %s): the M-step uses the Python loop" % _scipy_version(), RuntimeWarning)
        except Exception:
            ok = False
        _NATIVE_CHECKED[0] = ok
    return _NATIVE_CHECKED[0]


def _scipy_version():
    try:
        import scipy
        return scipy.__version__
    except Exception:
        return "?"


def native_available():
    """the M-step's native path (all states in one call of libphmrf_host.so on host threads) can be used in this process"""
    try:
        return bool(NATIVE_MSTEP[0] and NATIVE_LOOP[0] and slsqp_entry() is not None and host_lib() is not None
                    and _native_self_check())
    except Exception:
        return False


def do_mstep(tree, stats, params_cur, init_ou_params, n_samples, lambda_0, initial_mode, w1, w1a, w2, rng,
             min_covar=1e-3, workers=None, retries=3, states=None, base=None):
    """`_do_mstep` (phylo_hmrf.py:1500-1528) for all K states, or for the states listed in `states` (the others' rows of
    the result stay zero: with several ranks every rank fits its share and an all-reduce puts the rows together).
    The reference draws the random parts of all states' start points from one stream in state order (:1371-1380).  Here
    they come, laid out by state, from a generator seeded with ONE draw of `rng` per call: the result of a state does not
    depend on which other states are fitted in the same call, nor by whom -- as long as every rank calls this once per
    M-step with generators in the same state, each advances by exactly that one draw.
    -> params[K,3B+2], means[K,S], covars[K,S,S] (= V + min_covar*I as at :1524), lik[K]."""
    K = params_cur.shape[0]
    P = tree.n_params
    N = tree.node_num
    if base is None:        # (a caller that fits the states in several calls draws it once and hands it to each)
        base = int(rng.integers(0, 2 ** 62))
    todo = list(range(K)) if states is None else [int(c) for c in states]
    # the random parts of ALL states' start points from one generator seeded by `base`, laid out by state: what state c
    # gets does not depend on which states this call fits (every rank draws the same K x retries x P block: 1,380 doubles
    # at K = 20 -- cheaper than a generator per state)
    block = np.random.default_rng(base)
    ra = block.random((K, retries, P))
    rb = block.random((K, retries, P - N)) if initial_mode == 1 else None
    tasks = []
    for c in todo:
        guesses = []
        for q in range(retries):
            if initial_mode == 1:                                                     # (:1371-1376)
                r = 2.0 * ra[c, q] - 1.0
                r[0:P - N] = rb[c, q]
                r = w2 * r
            else:
                r = w2 * ra[c, q]
            guesses.append(w1 * init_ou_params[c] + w1a * params_cur[c] + (1.0 - w1 - w1a) * r)   # (:1378-1380)
        tasks.append((tree, stats["post"][c], stats["obs"][c], stats["obs*obs.T"][c], n_samples, lambda_0, guesses,
                      init_ou_params[c]))
    if workers is None:
        workers = min(max(len(tasks), 1), os.cpu_count() or 1)
    out = None
    if tasks and NATIVE_MSTEP[0]:
        out = _mstep_native(tree, tasks, n_samples, lambda_0, max(1, min(workers, len(tasks))))
    if out is None:
        pool = _pool(workers) if len(tasks) > 1 else None
        out = pool.map(_solve_state, tasks) if pool is not None else [_solve_state(t) for t in tasks]
    S = tree.n_features
    params = np.zeros((K, P))
    means = np.zeros((K, S))
    covars = np.zeros((K, S, S))
    lik = np.zeros(K)
    for c, (p, l, mu, V) in zip(todo, out):
        params[c], lik[c], means[c] = p, l, mu
        covars[c] = V + min_covar * np.eye(S)                                         # (:1524)
    return params, means, covars, lik


def ou_init_guess(tree, mean_values, w2, rng):
    """`_ou_init_guess` (phylo_hmrf.py:1453-1480): random start with node means propagated up from the leaves."""
    P, N = tree.n_params, tree.node_num
    guess = w2 * rng.random(P)
    mv = np.zeros(N)
    flag = np.zeros(N)
    mv[tree.leaf_vec] = mean_values
    flag[tree.leaf_vec] = 2
    for j in range(N - 1, 0, -1):
        p = tree.parent[j]
        if flag[p] == 0:
            mv[p] = mv[j]
            flag[p] += 1
        elif flag[p] == 1:
            mv[p] = 0.5 * mv[p] + 0.5 * mv[j]
            flag[p] += 1
    guess[P - N:P] = mv
    return guess


def _init_state(args):
    tree, X, guesses = args
    # (X: the cluster's rows, or the pair (mean, X^T X / n) of its sample moments)
    obj = OUObjectiveSingle.from_moments(tree, X[0], X[1]) if isinstance(X, tuple) else OUObjectiveSingle(tree, X)
    bounds = [(LOWER, UPPER)] * tree.n_params
    params1, flag = guesses[-1], -1
    for guess in guesses[:-1]:
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                res = minimize(obj.value_and_grad, np.clip(guess, LOWER, UPPER), jac=True, method="SLSQP",
                               bounds=bounds, tol=1e-6, options={"disp": False, "maxiter": 200})
        except Exception:
            continue
        params1 = res.x
        flag = check_params(tree, params1)
        if flag > 0:
            break
    if flag <= 0:
        params1 = guesses[-1]                                                          # (:1446-1448)
    return params1, obj.value(params1)


def _init_states(tree, tasks, workers):
    """the per-cluster fits of the initialisation: natively on host threads when libphmrf_host.so's SLSQP loop is available
    (`_ou_lik_varied_single` is the M-step's objective with N_c = n = 1 and no ridge term, so phmrf_ou_mstep fits it: the
    start points in turn, `_check_params`, the last guess as the fall-back) -- no worker processes, hence no fork after the
    GPU runtime is up; otherwise in the fork pool / serially with scipy.optimize.minimize"""
    if tasks and native_available():
        nat = []
        for tree_, Xc, guesses in tasks:
            obj = OUObjectiveSingle.from_moments(tree_, Xc[0], Xc[1]) if isinstance(Xc, tuple) else OUObjectiveSingle(tree_, Xc)
            nat.append((tree_, 1.0, obj.obs, obj.oo, 1.0, 0.0, list(guesses[:-1]), guesses[-1]))
        out = _mstep_native(tree, nat, 1.0, 0.0, max(1, min(workers, len(nat))))
        if out is not None:
            return [(p, l) for p, l, _, _ in out]
    pool = _pool(workers)
    return pool.map(_init_state, tasks) if pool is not None else [_init_state(t) for t in tasks]


def init_ou_params(tree, X, init_label, means, params_default, w2, rng, workers=None, max_per_cluster=200000):
    """`_init_ou_param` (phylo_hmrf.py:184-203): per k-means cluster, fit the OU parameters to that cluster."""
    K = params_default.shape[0]
    out = params_default.copy()
    tasks, idx = [], []
    for c in range(K):
        b = np.flatnonzero(init_label == c)
        if b.shape[0] == 0:
            continue                                                                  # "empty cluster!" (:191-192)
        if b.shape[0] > max_per_cluster:
            b = rng.choice(b, max_per_cluster, replace=False)
        guesses = [ou_init_guess(tree, means[c], w2, rng) for _ in range(4)]
        tasks.append((tree, X[b], guesses))
        idx.append(c)
    if workers is None:
        workers = min(max(len(tasks), 1), os.cpu_count() or 1)
    res = _init_states(tree, tasks, workers)
    for c, (p, _) in zip(idx, res):
        out[c] = p
    return out


def init_ou_params_moments(tree, counts, sums, outer, means, params_default, w2, rng, workers=None):
    """`_init_ou_param` (phylo_hmrf.py:184-203) from per-cluster sufficient statistics instead of the clusters' rows:
    counts[K], sums[K,S] = sum of x, outer[K,S,S] = sum of x x^T (phmrf_kmeans_moments, summed over blocks and ranks).
    The objective of the per-cluster fit (:1246-1325) only ever uses the cluster's mean and X^T X / n, so this is the same
    fit on ALL of a cluster's nodes without a host pass over them (init_ou_params subsamples clusters above 200,000 rows)."""
    K = params_default.shape[0]
    out = params_default.copy()
    tasks, idx = [], []
    for c in range(K):
        if not counts[c] > 0:
            continue                                                                  # "empty cluster!" (:191-192)
        guesses = [ou_init_guess(tree, means[c], w2, rng) for _ in range(4)]
        tasks.append((tree, (sums[c] / counts[c], outer[c] / counts[c]), guesses))
        idx.append(c)
    if workers is None:
        workers = min(max(len(tasks), 1), os.cpu_count() or 1)
    res = _init_states(tree, tasks, workers)
    for c, (p, _) in zip(idx, res):
        out[c] = p
    return out

