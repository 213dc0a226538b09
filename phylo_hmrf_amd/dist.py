"""Sharding of syntenic blocks over ranks (one process per GPU) and the one collective of the EM loop.

Blocks are independent MRFs (reference: one OS process per block, base.py:357-362; edge lists are block-local,
utility.py:509-529), so the data path has NO collective.  The only exchange per EM iteration replaces the parent's
reduction loop (base.py:384-394, :571-580): one all-reduce(sum) of
    [ post K | obs K*S | obs*obs.T K*S*S | 4 cost numerators | node count ]
(K*(1+S+S*S)+5 doubles: 3.4 KB at K=20,S=4) -- latency-bound, far from the xGMI per-link limit.  With backend
"nccl" this is RCCL on the GPUs; the CPU tests run it over gloo.
"""
import os

import numpy as np


def env_rank_world():
    """(rank, world size, this rank's device) from the launcher's environment.  PHMRF_ONE_GPU=1 (tests): every rank on
    device 0 -- the sharded path on one card, over gloo (RCCL refuses two ranks on one device)"""
    local = 0 if os.environ.get("PHMRF_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), local


def lpt_assign(sizes, world):
    """Longest-processing-time-first assignment of blocks to ranks -> owner[region] (deterministic)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.zeros(len(sizes), dtype=np.int64)
    for r in order:
        k = int(np.argmin(load))
        owner[r] = k
        load[k] += sizes[r]
    return owner


class Reducer(object):
    """all-reduce(sum) of a small float64 vector across ranks; identity when world == 1."""

    def __init__(self, world=1, device=None):
        self.world = world
        self.device = device
        self._buf = None
        if world > 1:
            import torch
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised before a multi-rank fit")
            if self.device is None and dist.get_backend() == "nccl":      # RCCL needs device tensors
                self.device = torch.device("cuda", torch.cuda.current_device())

    def allreduce(self, vec):
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        if self.world == 1:
            return vec
        import torch
        import torch.distributed as dist
        if self._buf is None or self._buf.numel() != vec.size:
            self._buf = torch.zeros(vec.size, dtype=torch.float64, device=self.device or "cpu")
        self._buf.copy_(torch.from_numpy(vec))
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
        return self._buf.cpu().numpy().copy()

    def allreduce_bytes(self, arr):
        """all-reduce(sum) of a uint8 array (label gathers: each position is written by exactly one rank)"""
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        if self.world == 1:
            return arr
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(arr.copy()).to(self.device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def broadcast(self, vec, src=0):
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        if self.world == 1:
            return vec
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(vec.copy()).to(self.device or "cpu")
        dist.broadcast(t, src=src)
        return t.cpu().numpy()
