"""Concurrent E-step over independent blocks.

The reference runs the E-step of every syntenic block in its own forked process (base.py:357-362).  Here every block
owns a HIP stream (`phmrf_block_create`), and a small pool of host threads drives several blocks at once: the library
calls release the GIL (ctypes), the per-round host synchronisations of one solve overlap with the kernels of the
others, and the launch-latency-bound kernels of small blocks run next to those of large ones.
"""
from concurrent.futures import ThreadPoolExecutor


class BlockRunner(object):
    def __init__(self, n_threads, device=None):
        """device: HIP device the worker threads select (None: leave it alone -- CPU test doubles)."""
        self.n_threads = max(1, int(n_threads))
        self.device = None if device is None else int(device)
        self._pool = None
        self._aux = None
        if self.n_threads > 1:
            self._pool = ThreadPoolExecutor(max_workers=self.n_threads, initializer=self._init_thread)

    def _init_thread(self):
        if self.device is None:
            return
        from . import _lib
        _lib.check(_lib.load().phmrf_set_device(self.device))      # the current device is per host thread

    def map(self, fn, items):
        """fn(item) for every item (in order of submission when sequential); exceptions propagate."""
        if self._pool is None:
            return [fn(it) for it in items]
        return list(self._pool.map(fn, items))

    def start(self, fn, items):
        """submit fn(item) for every item and return at once; .results() waits (exceptions propagate there)"""
        items = list(items)
        if self._pool is None:
            return _Deferred(lambda: [fn(it) for it in items])
        futures = [self._pool.submit(fn, it) for it in items]
        return _Deferred(lambda: [f.result() for f in futures])

    def start_one(self, fn):
        """fn() on ONE helper thread of its own (created at the first call), returning at once; .results() -> [fn()].
        For a caller that drives all whole blocks from one thread (block_threads=0) and has row tiles to run meanwhile."""
        if self._aux is None:
            self._aux = ThreadPoolExecutor(max_workers=1, initializer=self._init_thread)
        future = self._aux.submit(fn)
        return _Deferred(lambda: [future.result()])

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self._aux is not None:
            self._aux.shutdown(wait=True)
            self._aux = None


class _Deferred(object):
    def __init__(self, wait):
        self._wait = wait

    def results(self):
        return self._wait()
