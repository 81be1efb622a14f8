"""Output-Area sharded runs: one process per GPU, the exchange between the shards owned by libesim.

The reference has no distributed path (README.md:24 lists it as future work); the sharding follows its only data-parallel
axis, the Output Areas (sim/src/simulator.rs:167).  Citizens live on the shard of their home area; a building or school room
whose members live on several shards is "shared", and the Infected standing in it are what crosses between the shards -- the
commuter exchange of SURVEY.md 8(e).  Every step runs in three device phases around two small SUM all-reduces (include/esim.h,
esim_run_sharded): A = census + Infected counts of the shared buildings / rooms, B = exposure totals, eligible count and the
liveness bits of the step's vaccination candidates -- or, wherever a chunk of steps can run on every shard, one round of
exchanges per chunk.  A device-side error on any shard reaches every rank in those same collectives: all ranks
raise the same EsimError from run() together; a peer that died shows as ESIM_ETIMEDOUT after the deadline.  The collectives are issued by the library itself: over RCCL (xGMI) with
its own communicator, enqueued on the context's stream between its kernels; or, for tests that put several ranks on one GPU,
through a callback into torch.distributed's gloo backend.  torch.distributed is otherwise only the launcher's rendezvous (it
carries the RCCL unique id from rank 0 to the others)."""
import ctypes as C

import numpy as np

from . import _lib
from .population import Population
from .simulator import RECORD_DTYPE


def clean_cuts(pop, n_shards, slack=0.1):
    """Area boundaries for `n_shards` shards of a whole population: among the boundaries that keep every shard within
    `slack` of an even share of the citizens, the one the fewest citizens commute across (home area on one side, work
    building on the other)."""
    ah = pop.building_area[pop.home_building].astype(np.int64)
    aw = pop.building_area[pop.work_building].astype(np.int64)
    lo, hi = np.minimum(ah, aw), np.maximum(ah, aw)
    diff = np.zeros(pop.n_areas + 2, np.int64)
    m = lo != hi
    np.add.at(diff, lo[m] + 1, 1)
    np.add.at(diff, hi[m] + 1, -1)
    crossings = np.cumsum(diff)[: pop.n_areas + 1]          # crossings[b]: commuters across the boundary before area b
    per_area = np.bincount(ah, minlength=pop.n_areas)
    cum = np.concatenate([[0], np.cumsum(per_area)])
    cuts = [0]
    for k in range(1, n_shards):
        target = pop.n_citizens * k / n_shards
        cand = np.arange(cuts[-1] + 1, pop.n_areas)
        if cand.size == 0:
            cuts.append(pop.n_areas)
            continue
        off = np.abs(cum[cand] - target)
        near = cand[off <= max(off.min(), slack * pop.n_citizens / n_shards)]
        cost = crossings[near] * float(pop.n_citizens) + np.abs(cum[near] - target)
        cuts.append(int(near[np.argmin(cost)]))
    cuts.append(pop.n_areas)
    return np.maximum.accumulate(np.asarray(cuts, np.uint32))


def exchange_unique_id(lib, dist, rank, group=None):
    """Rank 0 makes the RCCL unique id (esim_comm_unique_id) and broadcasts it TOGETHER WITH whether that worked: a rank 0 that
    raised before the broadcast would leave the other ranks inside it (and the caller's next collective mismatched).  Every
    rank returns the 128 bytes or raises the same EsimError -- after the broadcast, so the ranks stay in step."""
    uid = (C.c_uint8 * 128)()
    status, text = 0, ""
    if rank == 0:
        status = int(lib.esim_comm_unique_id(uid, 128))
        if status != 0:
            msg = lib.esim_last_error(None)
            text = msg.decode() if msg else ""
    box = [(status, text, bytes(uid))]
    dist.broadcast_object_list(box, src=0, group=group)
    status, text, raw = box[0]
    if status != 0:
        raise _lib.EsimError(status, "rank 0 could not make the RCCL unique id (%s)" % text)
    return raw


class ShardedSimulator:
    """One rank of a sharded run.  Give either `whole_population` (cut here with `cuts`, default clean_cuts) or
    `shard_population` (already this rank's shard).  transport: "rccl" (the library's own communicator; the unique id
    travels over `group`) or "callback" (every exchange is a torch.distributed all_reduce on `group`, e.g. gloo)."""

    def __init__(self, whole_population=None, rank=0, world_size=1, params=None, device_index=0, group=None,
                 shard_population=None, cuts=None, transport="rccl", timeout_s=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world_size = rank, world_size
        self.lib = _lib.load()
        self.params = params if params is not None else _lib.default_params()
        self.params.device = device_index
        if shard_population is not None:
            self.population = shard_population
        else:
            self.cuts = clean_cuts(whole_population, world_size) if cuts is None else np.asarray(cuts, np.uint32)
            self.population = whole_population.shard(self.cuts, rank)
        self.n_citizens_global = self.population.n_citizens_global
        self._ctx = C.c_void_p()
        torch.cuda.set_device(device_index)
        _lib.check(self.lib.esim_create(C.byref(self.params), C.byref(self._ctx)))
        ps = self.population.as_struct()
        _lib.check(self.lib.esim_upload_population(self._ctx, C.byref(ps)), self._ctx)
        self.sharded = world_size > 1
        self.transport = transport if self.sharded else "none"
        self._steps = 0
        if timeout_s is not None:          # deadline of the library's waits inside a sharded run (default 60 s)
            _lib.check(self.lib.esim_comm_set_timeout(self._ctx, float(timeout_s)), self._ctx)
        if self.sharded and transport == "rccl":
            raw = exchange_unique_id(self.lib, dist, rank, group)      # every rank raises together when rank 0 could not make it
            buf = (C.c_uint8 * 128).from_buffer_copy(raw)
            _lib.check(self.lib.esim_comm_init_rccl(self._ctx, buf, 128, rank, world_size), self._ctx)
        elif self.sharded:
            def allreduce(_user, _which, host_ptr, n):
                try:
                    a = np.ctypeslib.as_array(C.cast(host_ptr, C.POINTER(C.c_int32)), shape=(n,))
                    dist.all_reduce(torch.from_numpy(a), group=self.group)      # in place, on the library's staging buffer
                    return 0
                except Exception:          # never unwind into the library
                    return 1
            self._cb = _lib.ALLREDUCE_FN(allreduce)              # (kept alive with the object)
            _lib.check(self.lib.esim_comm_init_callback(self._ctx, self._cb, None, rank, world_size), self._ctx)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------
    def run(self, n_steps):
        """n_steps time steps (read the records afterwards).  Every rank calls it with the same n_steps."""
        n_done = C.c_uint32(0)
        if self.sharded:
            _lib.check(self.lib.esim_run_sharded(self._ctx, n_steps, C.byref(n_done)), self._ctx)
        else:
            buf = (_lib.StepResult * max(1, n_steps))()
            _lib.check(self.lib.esim_run(self._ctx, n_steps, 0, buf, C.byref(n_done)), self._ctx)
        self._steps += n_done.value

    def synchronize(self):
        _lib.check(self.lib.esim_synchronize(self._ctx), self._ctx)

    def records(self, first_step=1, n=None):
        """Whole-population records (the same on every rank)."""
        n = self._steps - first_step + 1 if n is None else n
        buf = (_lib.StepResult * max(1, n))()
        _lib.check(self.lib.esim_read_records(self._ctx, first_step, n, buf), self._ctx)
        return np.frombuffer(buf, dtype=RECORD_DTYPE, count=n).copy()

    local_records = records

    def collectives(self):
        n = C.c_uint64(0)
        _lib.check(self.lib.esim_comm_stats(self._ctx, C.byref(n)), self._ctx)
        return n.value

    def shard_stats(self):
        """Steps run as time-parallel chunks (one round of exchanges per chunk) / as coupled steps (two exchanges per step)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.esim_shard_stats(self._ctx, C.byref(a), C.byref(b)), self._ctx)
        return {"chunk_steps": a.value, "coupled_steps": b.value}

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
        ms = C.c_double(0)
        n = C.c_uint32(0)
        _lib.check(self.lib.esim_kernel_timings(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return {"multi_kernel_step_ms": ms.value, "steps_timed": n.value}

    def set_small_step_limit(self, max_infected):
        _lib.check(self.lib.esim_set_small_step_limit(self._ctx, int(max_infected)), self._ctx)

    def set_pipeline(self, level):
        """0: sequential steps only; 1: chunks as one kernel per step; 2: time-parallel chunks; 3 (default): also under a vaccination programme."""
        _lib.check(self.lib.esim_set_pipeline(self._ctx, int(level)), self._ctx)

    def chunk_timing(self):
        ms, ns, nc = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.esim_chunk_timing(self._ctx, C.byref(ms), C.byref(ns), C.byref(nc)), self._ctx)
        return {"chunk_ms": ms.value, "steps": ns.value, "chunks": nc.value}

    def vax_chunk_stats(self):
        ns, nc = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.esim_vax_chunk_stats(self._ctx, C.byref(ns), C.byref(nc)), self._ctx)
        return {"steps": ns.value, "cuts": nc.value}

    def pipeline_timing(self):
        ms, nt, nr = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.esim_pipeline_timing(self._ctx, C.byref(ms), C.byref(nt), C.byref(nr)), self._ctx)
        return {"k_pipe_ms": ms.value, "steps_timed": nt.value, "steps": nr.value}

    def small_kernel_timing(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        _lib.check(self.lib.esim_small_kernel_timing(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return {"k_small_ms": ms.value, "steps": n.value}

    def close(self):
        if self._ctx:
            self.lib.esim_destroy(self._ctx)
            self._ctx = C.c_void_p()
