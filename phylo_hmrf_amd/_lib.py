"""ctypes binding of libphmrf.so (include/phmrf.h).  No CPU fallback: if the HIP library is missing or
no GPU is visible, the product path raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHMRF_LIB: development builds (tools/variant.sh writes them under variants/) are loaded from their own path; the
# product library is never overwritten
LIB_PATH = os.environ.get("PHMRF_LIB") or os.path.join(_HERE, "libphmrf.so")

OK = 0
ABI_VERSION = 124             # include/phmrf.h PHMRF_VERSION: checked against the library in load()
NUM_KERNEL_CLASSES = 10
KERNEL_CLASSES = ("emission", "icm", "chain", "component", "energy", "posterior_stats", "strip", "propose", "coarse", "fusion")


class PhmrfError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "libphmrf status %d: %s" % (status, message))
        self.status = status


class SolveOpts(ctypes.Structure):
    _fields_ = [("max_rounds", ctypes.c_int), ("use_chains", ctypes.c_int), ("use_components", ctypes.c_int),
                ("init_mode", ctypes.c_int), ("use_strips", ctypes.c_int), ("use_expansion", ctypes.c_int),
                ("min_changed", ctypes.c_int), ("use_coarse", ctypes.c_int), ("energy_tol_ppb", ctypes.c_int),
                ("coarse_start", ctypes.c_int)]


class SolveResult(ctypes.Structure):
    _fields_ = [("energy", ctypes.c_double), ("energy_unary", ctypes.c_double), ("energy_pair", ctypes.c_double),
                ("energy_init", ctypes.c_double), ("rounds", ctypes.c_int), ("converged", ctypes.c_int),
                ("changed", ctypes.c_int64)]


_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double
_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "phmrf_version": [],
    "phmrf_last_error": [],
    "phmrf_status_string": [_i],
    "phmrf_device_count": [ctypes.POINTER(_i)],
    "phmrf_set_device": [_i],
    "phmrf_block_create": [_i64, _i, _i, ctypes.POINTER(_vp)],
    "phmrf_block_destroy": [_vp],
    "phmrf_block_set_stream": [_vp, _vp],
    "phmrf_block_sync": [_vp],
    "phmrf_block_set_observations": [_vp, _dp],
    "phmrf_block_set_observations_dev": [_vp, _vp],
    "phmrf_block_set_graph": [_vp, _i64, _lp, _dp],
    "phmrf_block_set_grid": [_vp, _i, _i, _i, _i],
    "phmrf_block_build_grid_graph": [_vp, _i, _i, _i, _i, _d],
    "phmrf_block_get_adjacency": [_vp, ctypes.POINTER(_i), _ip, _fp],
    "phmrf_block_set_labels": [_vp, _ip],
    "phmrf_block_get_labels": [_vp, _ip],
    "phmrf_block_save_labels": [_vp, _i],
    "phmrf_block_restore_labels": [_vp, _i],
    "phmrf_block_get_saved_labels": [_vp, _i, _ip],
    "phmrf_block_warm_start": [_vp, _d, _i, _i, _dp, _dp, ctypes.POINTER(_i)],
    "phmrf_emission": [_vp, _dp, _dp],
    "phmrf_block_get_logprob": [_vp, _dp],
    "phmrf_block_set_logprob": [_vp, _dp],
    "phmrf_emission_pack_size": [_i, _i, _lp],
    "phmrf_emission_pack": [_i, _i, _dp, _dp, _fp],
    "phmrf_emission_dev": [_vp, _i64, _i, _i, _vp, _vp, _vp],
    "phmrf_mrf_solve": [_vp, _d, ctypes.POINTER(SolveOpts), ctypes.POINTER(SolveResult)],
    "phmrf_mrf_solve_group": [ctypes.POINTER(ctypes.c_void_p), _i, _d, ctypes.POINTER(SolveOpts)],
    "phmrf_mrf_solve_begin": [_vp, _d, ctypes.POINTER(SolveOpts), _i],
    "phmrf_mrf_solve_round_launch": [_vp],
    "phmrf_mrf_solve_round_collect": [_vp, ctypes.POINTER(ctypes.c_uint64), _dp],
    "phmrf_mrf_solve_round_decide": [_vp, ctypes.POINTER(ctypes.c_uint64), _dp, ctypes.POINTER(_i)],
    "phmrf_mrf_solve_end": [_vp, ctypes.POINTER(SolveResult)],
    "phmrf_block_set_tile": [_vp, _i, _i, _i64],
    "phmrf_block_tile_pins": [_vp, _i, _i],
    "phmrf_block_tile_get_boundary": [_vp, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8)],
    "phmrf_block_tile_put_halo": [_vp, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8)],
    "phmrf_mrf_icm_sweep": [_vp, _d, _lp],
    "phmrf_mrf_chain_sweep": [_vp, _d, _i, _lp],
    "phmrf_mrf_component_pass": [_vp, _d, _lp],
    "phmrf_mrf_graph_expansion": [_vp, _d, _i, _lp],
    "phmrf_block_prepare_components": [_vp],
    "phmrf_mrf_strip_pass": [_vp, _d, _i, _i, _i, _i, _lp],
    "phmrf_mrf_strip_multi_pass": [_vp, _d, _i, _i, _i, ctypes.c_uint64, _lp],
    "phmrf_mrf_energy": [_vp, _d, _dp, _dp, _dp],
    "phmrf_mrf_coarse_pass": [_vp, _d, _i, _i, _i, _i, _i, _lp],
    "phmrf_block_coarse_problem": [_vp, _d, _i, _i, _i, _lp, _fp, _fp],
    "phmrf_posterior_stats": [_vp, _d, _i, _dp, _dp, _dp],
    "phmrf_posterior_stats_dev": [_vp, _d, _i, _vp],
    "phmrf_kmeans_step": [_vp, _dp, _i, _dp],
    "phmrf_kmeans_moments": [_vp, _dp, _i, _dp],
    "phmrf_block_enable_timing": [_vp, _i],
    "phmrf_block_set_timing_classes": [_vp, ctypes.c_uint32],
    "phmrf_block_get_timing": [_vp, _i, _dp, _lp],
    "phmrf_block_get_timing_first": [_vp, _i, _dp, _lp],
    "phmrf_block_reset_timing": [_vp],
    "phmrf_block_get_work": [_vp, _lp],
    "phmrf_block_get_work_first": [_vp, _lp],
    "phmrf_block_get_work_ex": [_vp, _i, _i, _lp],
    "phmrf_time_base_reset": [],
    "phmrf_block_get_intervals": [_vp, _i, _dp, _i64, _lp],
}
_RESTYPES = {"phmrf_last_error": ctypes.c_char_p, "phmrf_status_string": ctypes.c_char_p}


def load():
    """Load libphmrf.so and declare every entry point.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C phylo_hmrf_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        try:
            # PyTorch ships its own libamdhip64; two HIP runtimes in one process do not both see the GPU.  Importing
            # torch first makes the loader resolve libphmrf.so against the runtime torch uses (plumbing only).
            import torch  # noqa: F401
        except Exception:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        got = L.phmrf_version()
        if got != ABI_VERSION:
            raise ImportError("%s has ABI version %d, this binding expects %d (include/phmrf.h PHMRF_VERSION): rebuild it "
                              "with `make -C phylo_hmrf_amd/csrc`" % (LIB_PATH, got, ABI_VERSION))
        _lib = L
    return _lib


def check(status):
    if status != OK:
        raise PhmrfError(status, load().phmrf_last_error().decode())


def device_count():
    c = ctypes.c_int(0)
    st = load().phmrf_device_count(ctypes.byref(c))
    return c.value if st == OK else 0


def require_gpu():
    if device_count() < 1:
        raise RuntimeError("phylo_hmrf_amd needs an MI355X (gfx950) GPU: no HIP device is visible and there is "
                           "no CPU fallback")


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr_d(a):
    return a.ctypes.data_as(_dp)


def ptr_i32(a):
    return a.ctypes.data_as(_ip)


def ptr_i64(a):
    return a.ctypes.data_as(_lp)
