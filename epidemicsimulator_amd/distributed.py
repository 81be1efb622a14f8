"""Output-Area sharded runs: one process per GPU, `torch.distributed` for the two small SUM
all-reduces a time step needs (backend "nccl" = RCCL over xGMI on the GPU node; "gloo" in tests).

The reference has no distributed path (README.md:24 lists it as future work); the sharding follows
its only data-parallel axis, the Output Areas (sim/src/simulator.rs:167).  Citizens live on the shard
of their home area; the infected counts of buildings / school rooms whose members live on several
shards, the census and the vaccination liveness bits travel in the exchange buffers
(include/esim.h: esim_step_begin / esim_step_exposures / esim_step_finish).
"""
import ctypes as C

import numpy as np

from . import _lib
from .simulator import RECORD_DTYPE


class ShardedSimulator:
    def __init__(self, whole_population, rank, world_size, params=None, device_index=0, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world_size = rank, world_size
        self.lib = _lib.load()
        self.params = params if params is not None else _lib.default_params()
        self.params.device = device_index
        self.cuts = whole_population.even_cuts(world_size)
        self.population = whole_population.shard(self.cuts, rank)
        self.n_citizens_global = whole_population.n_citizens
        self._ctx = C.c_void_p()
        torch.cuda.set_device(device_index)
        _lib.check(self.lib.esim_create(C.byref(self.params), C.byref(self._ctx)))
        ps = self.population.as_struct()
        if world_size == 1:          # a single shard is an ordinary (unsharded) population
            ps.n_citizens_global = ps.n_citizens
        _lib.check(self.lib.esim_upload_population(self._ctx, C.byref(ps)), self._ctx)
        self.sharded = world_size > 1
        self._steps = 0
        if self.sharded:
            # collectives are ordered against this stream; the library enqueues its kernels on it too
            self.stream = torch.cuda.Stream(device=device_index)
            _lib.check(self.lib.esim_set_stream(self._ctx, C.c_void_p(self.stream.cuda_stream)), self._ctx)
            self.xbuf = []
            for which in (0, 1):
                n = C.c_size_t(0)
                ptr = C.c_void_p()
                _lib.check(self.lib.esim_exchange_buffer(self._ctx, which, C.byref(ptr), C.byref(n)), self._ctx)
                t = torch.zeros(n.value, dtype=torch.int32, device="cuda:%d" % device_index)
                _lib.check(self.lib.esim_set_exchange_buffer(self._ctx, which, C.c_void_p(t.data_ptr())), self._ctx)
                self.xbuf.append(t)
            torch.cuda.synchronize()

    def run(self, n_steps):
        """n_steps time steps; returns nothing (read the records afterwards) so no host sync is forced."""
        lib, ctx = self.lib, self._ctx
        if not self.sharded:
            buf = (_lib.StepResult * max(1, n_steps))()
            n_done = C.c_uint32(0)
            _lib.check(lib.esim_run(ctx, n_steps, 0, buf, C.byref(n_done)), ctx)
            self._steps += n_steps
            return
        torch, dist = self.torch, self.dist
        with torch.cuda.stream(self.stream):
            for _ in range(n_steps):
                _lib.check(lib.esim_step_begin(ctx), ctx)
                dist.all_reduce(self.xbuf[0], group=self.group)
                _lib.check(lib.esim_step_exposures(ctx), ctx)
                dist.all_reduce(self.xbuf[1], group=self.group)
                _lib.check(lib.esim_step_finish(ctx, None), ctx)
        self._steps += n_steps

    def synchronize(self):
        _lib.check(self.lib.esim_synchronize(self._ctx), self._ctx)

    def records(self, first_step=1, n=None):
        n = self._steps - first_step + 1 if n is None else n
        buf = (_lib.StepResult * max(1, n))()
        _lib.check(self.lib.esim_read_records(self._ctx, first_step, n, buf), self._ctx)
        return np.frombuffer(buf, dtype=RECORD_DTYPE, count=n).copy()

    def reset(self):
        _lib.check(self.lib.esim_reset(self._ctx), self._ctx)
        self._steps = 0

    def download_state(self):
        n = self.population.n_citizens
        out = {"status": np.zeros(n, np.uint8), "timer": np.zeros(n, np.uint16),
               "current_building": np.zeros(n, np.uint32), "on_bus": np.zeros(n, np.uint8),
               "eligible": np.zeros(n, np.uint8)}
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        _lib.check(self.lib.esim_download_state(
            self._ctx, p(out["status"], C.c_uint8), p(out["timer"], C.c_uint16),
            p(out["current_building"], C.c_uint32), p(out["on_bus"], C.c_uint8),
            p(out["eligible"], C.c_uint8)), self._ctx)
        return out

    def enable_kernel_timing(self, stride):
        _lib.check(self.lib.esim_enable_kernel_timing(self._ctx, int(stride)), self._ctx)

    def kernel_timings(self):
        ms = (C.c_double * 3)()
        n = C.c_uint32(0)
        _lib.check(self.lib.esim_kernel_timings(self._ctx, ms, C.byref(n)), self._ctx)
        return {"k_infected_ms": ms[0], "k_expose_ms": ms[1], "k_finish_ms": ms[2], "launches": n.value}

    def close(self):
        if self._ctx:
            self.lib.esim_destroy(self._ctx)
            self._ctx = C.c_void_p()
