"""TEST INFRASTRUCTURE ONLY: ctypes front-end to oracle/libphmrf_oracle.so (oracle/estep_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("PHMRF_ORACLE_LIB") or os.path.join(_HERE, "libphmrf_oracle.so")     # (env: the sanitizer build)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", _HERE, "port"])
        L = ctypes.CDLL(_PATH)
        L.oracle_swap.restype = ctypes.c_int64
        L.oracle_energy.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def emission(X, means, covars):
    X = np.ascontiguousarray(X, dtype=np.float64)
    means = np.ascontiguousarray(means, dtype=np.float64)
    covars = np.ascontiguousarray(covars, dtype=np.float64)
    n, S = X.shape
    K = means.shape[0]
    out = np.empty((n, K))
    rc = lib().oracle_emission(_p(X, ctypes.c_double), ctypes.c_int64(n), S, K, _p(means, ctypes.c_double),
                               _p(covars, ctypes.c_double), _p(out, ctypes.c_double))
    if rc:
        raise ValueError("'covars' must be symmetric, positive-definite")
    return out


def swap_int(edges, w_int, unary_int, smooth_int, init_labels, max_cycles=5000):
    unary_int = np.ascontiguousarray(unary_int, dtype=np.intc)
    smooth_int = np.ascontiguousarray(smooth_int, dtype=np.intc)
    n, K = unary_int.shape
    s1 = np.ascontiguousarray(np.asarray(edges)[:, 0], dtype=np.intc)
    s2 = np.ascontiguousarray(np.asarray(edges)[:, 1], dtype=np.intc)
    w_int = np.ascontiguousarray(w_int, dtype=np.intc)
    lab = np.ascontiguousarray(init_labels, dtype=np.intc).copy()
    cyc = ctypes.c_int(0)
    i = ctypes.c_int
    e = lib().oracle_swap(ctypes.c_int64(n), K, _p(unary_int, i), _p(smooth_int, i), ctypes.c_int64(len(s1)), _p(s1, i),
                          _p(s2, i), _p(w_int, i), _p(lab, i), int(max_cycles), ctypes.byref(cyc))
    if e == -1:
        raise ValueError("non-submodular pair term")
    return lab, int(e), cyc.value


def posterior_stats(X, logprob, edges, w, labels, beta, estimate_type):
    X = np.ascontiguousarray(X, dtype=np.float64)
    logprob = np.ascontiguousarray(logprob, dtype=np.float64)
    edges = np.ascontiguousarray(np.asarray(edges)[:, :2], dtype=np.int64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.intc)
    n, S = X.shape
    K = logprob.shape[1]
    stats = np.zeros(K * (1 + S + S * S))
    costs = np.zeros(4)
    post = np.empty((n, K))
    d = ctypes.c_double
    lib().oracle_posterior_stats(_p(X, d), _p(logprob, d), ctypes.c_int64(n), S, K, ctypes.c_int64(len(w)),
                                 _p(edges, ctypes.c_int64), _p(w, d), _p(labels, ctypes.c_int), d(beta),
                                 int(estimate_type), _p(stats, d), _p(costs, d), _p(post, d))
    return stats, costs, post


def energy(logprob, edges, w, labels, beta):
    logprob = np.ascontiguousarray(logprob, dtype=np.float64)
    edges = np.ascontiguousarray(np.asarray(edges)[:, :2], dtype=np.int64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.intc)
    n, K = logprob.shape
    return lib().oracle_energy(_p(logprob, ctypes.c_double), ctypes.c_int64(n), K, ctypes.c_int64(len(w)),
                               _p(edges, ctypes.c_int64), _p(w, ctypes.c_double), _p(labels, ctypes.c_int),
                               ctypes.c_double(beta))
