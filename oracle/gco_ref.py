"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

ctypes front-end to ``oracle/_ref/libgco_ref.so`` (the reference's own gco-v3.0,
compiled in place by ``oracle/Makefile``), shaped like the call the reference makes:

    pygco.cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost,
                            n_iter=5000, algorithm='swap', init_labels=...,
                            down_weight_factor=None)          # phylo_hmrf.py:496-498

``pygco`` itself is NOT vendored in the reference (install pointer README.md:84, no
version pinned), so its float->int conversion is restated here from its published
source (yujiali/pygco ``pygco.py``: ``_UNARY_FLOAT_PRECISION = 100000``,
``_PAIRWISE_FLOAT_PRECISION = 1000``, ``_SMOOTH_COST_PRECISION = 100``; with
``down_weight_factor=None`` both unary costs and edge weights are first divided by
``max(|unary|.max(), |w|.max() * V.max()) + 1e-8``; conversions truncate via
``astype(np.intc)``).  PARITY UNPINNED for that conversion: the reference holds no
test that fixes it.  What IS pinned is gco itself: ``tests/test_oracle_gco.py`` replays
gco_source/example.cpp:276-338 through this wrapper and checks the known energies
(250 -> 44 and 250 -> 244).

Two quantisations are offered:
  quant='pygco'  the reference's effective behaviour (coarse edge weights),
  quant='fine'   scale so the largest term is just under GCO_MAX_ENERGYTERM (1e7,
                 GCoptimization.h:139-143): the strongest swap optimum gco can give.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libgco_ref.so")

_UNARY_FLOAT_PRECISION = 100000
_PAIRWISE_FLOAT_PRECISION = 1000
_SMOOTH_COST_PRECISION = 100
_SMALL_CONSTANT = 1e-8

_lib = None


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libgco_ref.so missing: run `make -C oracle ref` "
                               "in a container where /root/reference is mounted")
        L = ctypes.CDLL(_LIB_PATH)
        ip = ctypes.POINTER(ctypes.c_int)
        llp = ctypes.POINTER(ctypes.c_longlong)
        L.gcoref_last_error.restype = ctypes.c_char_p
        L.gcoref_create_general_graph.argtypes = [ctypes.c_int, ctypes.c_int, ip]
        L.gcoref_destroy.argtypes = [ctypes.c_int]
        L.gcoref_set_data_cost.argtypes = [ctypes.c_int, ip]
        L.gcoref_set_smooth_cost.argtypes = [ctypes.c_int, ip]
        L.gcoref_set_all_neighbors.argtypes = [ctypes.c_int, ip, ip, ip, ctypes.c_int]
        L.gcoref_set_labels.argtypes = [ctypes.c_int, ip, ctypes.c_int]
        L.gcoref_get_labels.argtypes = [ctypes.c_int, ip, ctypes.c_int]
        L.gcoref_swap.argtypes = [ctypes.c_int, ctypes.c_int, llp]
        L.gcoref_expansion.argtypes = [ctypes.c_int, ctypes.c_int, llp]
        L.gcoref_alpha_beta_swap.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.gcoref_energy.argtypes = [ctypes.c_int, llp, llp, llp]
        _lib = L
    return _lib


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _check(rc):
    if rc != 0:
        raise RuntimeError("gco: " + lib().gcoref_last_error().decode())


def cut_general_graph_int(edges, w_int, unary_int, smooth_int, n_iter=-1, algorithm="expansion",
                          init_labels=None, return_energy=False):
    """Integer-exact entry: everything is already ``intc``.  Returns labels (int32)."""
    L = lib()
    unary_int = np.ascontiguousarray(unary_int, dtype=np.intc)
    smooth_int = np.ascontiguousarray(smooth_int, dtype=np.intc)
    n, K = unary_int.shape
    edges = np.asarray(edges)
    s1 = np.ascontiguousarray(edges[:, 0], dtype=np.intc)
    s2 = np.ascontiguousarray(edges[:, 1], dtype=np.intc)
    w_int = np.ascontiguousarray(w_int, dtype=np.intc)
    h = ctypes.c_int(0)
    _check(L.gcoref_create_general_graph(n, K, ctypes.byref(h)))
    try:
        _check(L.gcoref_set_data_cost(h, _ip(unary_int)))
        _check(L.gcoref_set_all_neighbors(h, _ip(s1), _ip(s2), _ip(w_int), len(s1)))
        _check(L.gcoref_set_smooth_cost(h, _ip(smooth_int)))
        if init_labels is not None:
            il = np.ascontiguousarray(np.asarray(init_labels), dtype=np.intc)
            _check(L.gcoref_set_labels(h, _ip(il), n))
        e0 = [ctypes.c_longlong(0) for _ in range(3)]
        _check(L.gcoref_energy(h, *[ctypes.byref(x) for x in e0]))
        e = ctypes.c_longlong(0)
        if algorithm == "expansion":
            _check(L.gcoref_expansion(h, n_iter, ctypes.byref(e)))
        elif algorithm == "swap":
            _check(L.gcoref_swap(h, n_iter, ctypes.byref(e)))
        else:
            raise ValueError(algorithm)
        labels = np.zeros(n, dtype=np.intc)
        _check(L.gcoref_get_labels(h, _ip(labels), n))
        e1 = [ctypes.c_longlong(0) for _ in range(3)]
        _check(L.gcoref_energy(h, *[ctypes.byref(x) for x in e1]))
    finally:
        L.gcoref_destroy(h)
    if return_energy:
        return labels, dict(before=e0[0].value, after=e1[0].value, data=e1[1].value, smooth=e1[2].value)
    return labels


def quantise(edge_weights, unary_cost, pairwise_cost, quant="pygco", down_weight_factor=None):
    """float (unary, w, V) -> intc triple under the chosen quantisation."""
    unary_cost = np.asarray(unary_cost, dtype=np.float64)
    edge_weights = np.asarray(edge_weights, dtype=np.float64)
    pairwise_cost = np.asarray(pairwise_cost, dtype=np.float64)
    if quant == "pygco":
        if down_weight_factor is None:
            down_weight_factor = max(np.abs(unary_cost).max(),
                                     np.abs(edge_weights).max() * pairwise_cost.max()) + _SMALL_CONSTANT
        u = (unary_cost / down_weight_factor * _UNARY_FLOAT_PRECISION).astype(np.intc)
        w = (edge_weights / down_weight_factor * _PAIRWISE_FLOAT_PRECISION).astype(np.intc)
        v = (pairwise_cost * _SMOOTH_COST_PRECISION).astype(np.intc)
        return u, w, v
    if quant == "fine":
        # One common scale s for unary and for w*V so the minimiser is that of the float
        # energy up to rounding; V is folded into w (Potts: V in {0, beta}) with V_int in {0,1}.
        beta = pairwise_cost.max()
        vmask = (pairwise_cost > 0).astype(np.intc)
        big = max(np.abs(unary_cost).max(), np.abs(edge_weights).max() * beta)
        s = 0.9 * 1e7 / big
        u = np.rint(unary_cost * s).astype(np.intc)
        w = np.rint(edge_weights * beta * s).astype(np.intc)
        return u, w, vmask
    raise ValueError(quant)


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm="expansion",
                      init_labels=None, down_weight_factor=None, quant="pygco", return_energy=False):
    """Restatement of ``pygco.cut_general_graph`` over the compiled reference gco."""
    u, w, v = quantise(edge_weights, unary_cost, pairwise_cost, quant, down_weight_factor)
    return cut_general_graph_int(edges, w, u, v, n_iter=n_iter, algorithm=algorithm,
                                 init_labels=init_labels, return_energy=return_energy)
