"""phylo_hmrf_amd -- MI355X-native E-step of Phylo-HMRF (emission likelihood, MRF labelling, posterior /
sufficient statistics) behind the reference's fit/predict surface.  The compute path is libphmrf.so
(hand-written HIP for gfx950, C ABI in include/phmrf.h); there is no CPU fallback."""
from . import _lib
from ._lib import PhmrfError, device_count, require_gpu
from .block import Block, pack_stats, unpack_stats

__all__ = ["Block", "PhmrfError", "device_count", "require_gpu", "pack_stats", "unpack_stats"]
