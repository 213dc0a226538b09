"""CPU: libphmrf.so loads without a GPU and exports every symbol include/phmrf.h declares; the ctypes table in
phylo_hmrf_amd/_lib.py covers exactly that set.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phmrf.h")
LIB = os.path.join(ROOT, "phylo_hmrf_amd", "libphmrf.so")


def declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"PHMRF_API\s+[\w\s\*]+?\b(phmrf_\w+)\s*\(", txt)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ("phmrf_emission", "phmrf_mrf_solve", "phmrf_posterior_stats", "phmrf_mrf_energy",
                 "phmrf_block_create", "phmrf_block_set_graph", "phmrf_block_destroy"):
        assert must in syms


@pytest.mark.skipif(not os.path.exists(LIB), reason="libphmrf.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


@pytest.mark.skipif(not os.path.exists(LIB), reason="libphmrf.so not built")
def test_ctypes_table_matches_header():
    from phylo_hmrf_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_symbols()
    L = _lib.load()
    assert L.phmrf_version() >= 100
    assert L.phmrf_status_string(0).decode() == "ok"
    assert L.phmrf_status_string(6).decode().startswith("covariance")


@pytest.mark.skipif(not os.path.exists(LIB), reason="libphmrf.so not built")
def test_product_path_fails_loudly_without_a_gpu():
    from phylo_hmrf_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        from phylo_hmrf_amd import Block
        Block(10, 4, 5)


DEV_KNOBS = ("PHMRF_PEEL_SWEEPS", "PHMRF_STRIP_DEBUG", "PHMRF_CHAIN_DEBUG", "PHMRF_COARSE_NO_GATE", "PHMRF_COARSE_NO_STAMP_GATE",
             "PHMRF_COARSE_BATCH", "PHMRF_NO_PIN_LOOK", "PHMRF_ENERGY_FULL", "PHMRF_ENERGY_CHECK", "PHMRF_CC_ROWS", "PHMRF_POST_TB",
             "PHMRF_POST_GRID", "PHMRF_CHILD_COUNT", "PHMRF_MULTI_V", "PHMRF_FUSION_V", "PHMRF_NO_XCD_MAP")


@pytest.mark.skipif(not os.path.exists(LIB), reason="libphmrf.so not built")
def test_product_library_has_no_development_knobs():
    """The product library cannot be talked into another labelling through the environment: the names of the development
    knobs (timing experiments that cut kernels short, slow paths for A/B runs, the deleted round-2 kernels' switches) do not
    occur in its binary at all -- the getenv calls are compiled only with -DPHMRF_DEV (common.h, PHMRF_DEV_ENV), into
    libphmrf_dev.so, which the tests that A/B a shortcut load through PHMRF_LIB.  What the product does read:
    PHMRF_DETERMINISTIC (a feature of the boundary, include/phmrf.h) and PHMRF_SOLVE_TRACE (prints, changes no result)."""
    blob = open(LIB, "rb").read()
    present = [k for k in DEV_KNOBS if k.encode() in blob]
    assert not present, present
    assert b"PHMRF_DETERMINISTIC" in blob and b"PHMRF_SOLVE_TRACE" in blob
    dev = os.path.join(ROOT, "phylo_hmrf_amd", "libphmrf_dev.so")
    if os.path.exists(dev):
        dblob = open(dev, "rb").read()
        for k in ("PHMRF_PEEL_SWEEPS", "PHMRF_COARSE_BATCH", "PHMRF_NO_PIN_LOOK", "PHMRF_COARSE_NO_STAMP_GATE", "PHMRF_NO_XCD_MAP"):
            assert k.encode() in dblob, k
        L = ctypes.CDLL(dev)
        missing = [s for s in declared_symbols() if not hasattr(L, s)]
        assert not missing, missing
    src = os.path.join(ROOT, "phylo_hmrf_amd", "csrc")
    n_getenv = sum(open(os.path.join(src, f)).read().count("getenv(") for f in os.listdir(src) if f.endswith(".hip"))
    assert n_getenv <= 4, n_getenv


def test_emission_pack_is_host_only_and_matches_oracle():
    """phmrf_emission_pack is pure host code (Cholesky, inverse factor, log-det): check it without a GPU."""
    if not os.path.exists(LIB):
        pytest.skip("libphmrf.so not built")
    import numpy as np
    from phylo_hmrf_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    S, K = 4, 3
    A = rng.standard_normal((K, S, S))
    cov = np.einsum("kij,klj->kil", A, A) + 2e-3 * np.eye(S)
    mu = rng.uniform(0, 3, (K, S))
    nf = ctypes.c_int64(0)
    assert L.phmrf_emission_pack_size(S, K, ctypes.byref(nf)) == 0
    PS = S + S * (S + 1) // 2 + 1
    assert nf.value == K * PS
    out = np.zeros(nf.value, dtype=np.float32)
    assert L.phmrf_emission_pack(S, K, _lib.ptr_d(mu), _lib.ptr_d(cov), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) == 0
    for k in range(K):
        p = out[k * PS:(k + 1) * PS]
        Lc = np.linalg.cholesky(cov[k])
        Li = np.linalg.inv(Lc)
        np.testing.assert_allclose(p[:S], mu[k], rtol=1e-6)
        np.testing.assert_allclose(p[S:-1], Li[np.tril_indices(S)], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(p[-1], -0.5 * (S * np.log(2 * np.pi) + 2 * np.log(np.diag(Lc)).sum()), rtol=1e-6)
    bad = cov.copy()
    bad[1] = -np.eye(S)
    assert L.phmrf_emission_pack(S, K, _lib.ptr_d(mu), _lib.ptr_d(bad), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) == 6


HOST_HEADER = os.path.join(ROOT, "include", "phmrf_host.h")
HOST_LIB = os.path.join(ROOT, "phylo_hmrf_amd", "libphmrf_host.so")


@pytest.mark.skipif(not os.path.exists(HOST_LIB), reason="libphmrf_host.so not built (run __graft_entry__.build())")
def test_host_library_exports_every_declared_symbol():
    syms = sorted(set(re.findall(r"PHMRF_HOST_API\s+[\w\s\*]+?\b(phmrf_\w+)\s*\(", open(HOST_HEADER).read())))
    assert syms == ["phmrf_bilateral", "phmrf_host_version", "phmrf_median_fill", "phmrf_ou_mstep", "phmrf_ou_objective",
                    "phmrf_ou_slsqp"]
    L = ctypes.CDLL(HOST_LIB)
    assert not [s for s in syms if not hasattr(L, s)]
    L.phmrf_host_version.restype = ctypes.c_int
    assert L.phmrf_host_version() >= 1
