"""phylo_hmrf_amd -- MI355X-native E-step of Phylo-HMRF (emission likelihood, MRF labelling, posterior /
sufficient statistics) behind the reference's fit/predict surface.  The compute path is libphmrf.so
(hand-written HIP for gfx950, C ABI in include/phmrf.h); there is no CPU fallback."""
import os as _os

# Every block owns a HIP stream and a dozen of them are in flight (concurrent.py); the HIP runtime multiplexes streams on
# 4 hardware queues unless told otherwise, and launches that share a queue run one after the other.  16 queues: -4 % on
# the whole-genome EM iteration (tools/job_queues.sh).  Read by the runtime when it initialises, i.e. at the first HIP
# call of the process; a value the user has set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import _lib
from ._lib import PhmrfError, device_count, require_gpu
from .block import Block, pack_stats, unpack_stats

__all__ = ["Block", "PhmrfError", "device_count", "require_gpu", "pack_stats", "unpack_stats"]
