"""Output-Area sharded runs: one process per GPU, `torch.distributed` for the collectives
(backend "nccl" = RCCL over xGMI on the GPU node; "gloo" in tests).

The reference has no distributed path (README.md:24 lists it as future work); the sharding follows
its only data-parallel axis, the Output Areas (sim/src/simulator.rs:167).  Citizens live on the shard
of their home area.  Two modes (include/esim.h):

* coupled -- every step is split in three device phases around two small SUM all-reduces: A = census +
  infected counts of buildings / school rooms whose members live on several shards (the commuter
  exchange), B = exposure totals, eligible count and vaccination-candidate liveness bits.
* decoupled -- when the shards share no building and no vaccination programme runs, the only thing a
  shard needs from the others is the global Infected count for the intervention thresholds, and that is
  known exposed_time + 1 steps ahead (a citizen exposed now is not Infected before).  Shards exchange the
  vector of their next <= 96 Infected counts once per batch and run the batch without any collective.
  The host reads the reduced vector, finds the step at which vaccination would trigger, and switches to
  the coupled mode from that step on.
"""
import ctypes as C

import numpy as np

from . import _lib
from .population import Population
from .simulator import RECORD_DTYPE

_SUMMED = ("susceptible", "exposed", "infected", "recovered", "vaccinated", "exposures_building",
           "exposures_bus", "n_riders")


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


class ShardedSimulator:
    """One rank of a sharded run.  Give either `whole_population` (cut here with `cuts`, default
    clean_cuts) or `shard_population` (already this rank's shard, e.g. Population.synthetic_shard)."""

    def __init__(self, whole_population=None, rank=0, world_size=1, params=None, device_index=0, group=None,
                 shard_population=None, cuts=None, decoupled=True):
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
        self._steps = 0
        self._local_ranges = []          # [first, last] step ranges whose records hold this shard's census only
        self.coupled_steps = 0
        self.free_steps = 0
        self.mode_free = bool(decoupled) and self.sharded and self.population.n_shared_buildings == 0 \
            and self.population.n_shared_rooms == 0
        if self.sharded:
            # every rank must take the same branch: decoupled only if NO rank shares a building
            flag = torch.tensor([0 if self.mode_free else 1], dtype=torch.int32, device="cuda:%d" % device_index)
            dist.all_reduce(flag, group=group)
            self.mode_free = int(flag.item()) == 0
            # collectives are ordered against this stream; the library enqueues its kernels on it too
            self.stream = torch.cuda.Stream(device=device_index)
            _lib.check(self.lib.esim_set_stream(self._ctx, C.c_void_p(self.stream.cuda_stream)), self._ctx)
            self.xbuf = []
            for which in (0, 1, 2):
                n = C.c_size_t(0)
                ptr = C.c_void_p()
                _lib.check(self.lib.esim_exchange_buffer(self._ctx, which, C.byref(ptr), C.byref(n)), self._ctx)
                t = torch.zeros(max(1, n.value), dtype=torch.int32, device="cuda:%d" % device_index)
                _lib.check(self.lib.esim_set_exchange_buffer(self._ctx, which, C.c_void_p(t.data_ptr())), self._ctx)
                self.xbuf.append(t)
            self.free_batch = int(self.xbuf[2].numel()) - 1     # the last word counts shards that need the per-step form
            self.burst_max = 16                                 # chunks kept in flight between two host waits
            self._burst, self._backoff, self._sync_left = 1, 0, 0
            torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------
    def run(self, n_steps):
        """n_steps time steps (read the records afterwards)."""
        lib, ctx = self.lib, self._ctx
        if not self.sharded:
            buf = (_lib.StepResult * max(1, n_steps))()
            n_done = C.c_uint32(0)
            _lib.check(lib.esim_run(ctx, n_steps, 0, buf, C.byref(n_done)), ctx)
            self._steps += n_steps
            return
        torch, dist = self.torch, self.dist
        done = 0
        with torch.cuda.stream(self.stream):
            while done < n_steps:
                if self.mode_free and self.burst_max > 0 and self._sync_left == 0:
                    # several whole chunks in flight: census ahead, then { all-reduce; chunk + next census ahead } x burst, one wait at the end.
                    # A chunk that cannot run this way is a no-op on every rank; the chunk-by-chunk form below then
                    # takes it and the burst length starts again from one.
                    want = n_steps - done
                    chunks = min(-(-want // self.free_batch), self._burst)
                    _lib.check(lib.esim_free_begin(ctx, want), ctx)
                    _lib.check(lib.esim_future_infected(ctx), ctx)
                    for _ in range(chunks):                       # (a chunk's last kernel leaves the next census ahead in F)
                        dist.all_reduce(self.xbuf[2], group=self.group)
                        _lib.check(lib.esim_free_enqueue(ctx), ctx)
                    got = C.c_uint32(0)
                    _lib.check(lib.esim_free_collect(ctx, C.byref(got)), ctx)
                    n_free = got.value
                    if n_free > 0:
                        self._local_ranges.append((self._steps + 1, self._steps + n_free))
                        self._steps += n_free
                        self.free_steps += n_free
                        done += n_free
                    if n_free >= min(want, chunks * self.free_batch):
                        self._burst = min(self.burst_max, self._burst * 2)
                        self._backoff = 0
                    else:
                        # the next chunk goes one at a time; chunks that keep failing are probed less and less often
                        self._backoff = 0 if n_free else min(64, max(1, 2 * self._backoff))
                        self._sync_left = max(1, self._backoff)
                        self._burst = 1
                    continue
                if self.mode_free:
                    self._sync_left = max(0, self._sync_left - 1)
                    _lib.check(lib.esim_future_infected(ctx), ctx)
                    dist.all_reduce(self.xbuf[2], group=self.group)
                    want = min(self.free_batch, n_steps - done)
                    got = C.c_uint32(0)
                    _lib.check(lib.esim_run_free(ctx, want, C.byref(got)), ctx)   # one host sync per chunk inside
                    n_free = got.value
                    if n_free > 0:
                        self._local_ranges.append((self._steps + 1, self._steps + n_free))
                        self._steps += n_free
                        self.free_steps += n_free
                        done += n_free
                    if n_free < want:
                        self.mode_free = False        # the programme starts in the next step: coupled from here on
                    continue
                _lib.check(lib.esim_step_begin(ctx), ctx)
                dist.all_reduce(self.xbuf[0], group=self.group)
                _lib.check(lib.esim_step_exposures(ctx), ctx)
                dist.all_reduce(self.xbuf[1], group=self.group)
                _lib.check(lib.esim_step_finish(ctx, None), ctx)
                self._steps += 1
                self.coupled_steps += 1
                done += 1

    def synchronize(self):
        _lib.check(self.lib.esim_synchronize(self._ctx), self._ctx)

    def local_records(self, first_step=1, n=None):
        n = self._steps - first_step + 1 if n is None else n
        buf = (_lib.StepResult * max(1, n))()
        _lib.check(self.lib.esim_read_records(self._ctx, first_step, n, buf), self._ctx)
        return np.frombuffer(buf, dtype=RECORD_DTYPE, count=n).copy()

    def records(self, first_step=1, n=None):
        """Whole-population records: steps run decoupled hold per-shard counts and are summed over the ranks
        here (one all-reduce for the whole range); coupled steps are already global."""
        rec = self.local_records(first_step, n)
        if not self.sharded or not self._local_ranges:
            return rec
        torch, dist = self.torch, self.dist
        steps = rec["time_step"].astype(np.int64)
        local = np.zeros(len(rec), bool)
        for a, b in self._local_ranges:
            local |= (steps >= a) & (steps <= b)
        vals = np.stack([np.where(local, rec[f], 0).astype(np.int64) for f in _SUMMED])
        t = torch.from_numpy(vals).to("cuda:%d" % self.params.device)
        dist.all_reduce(t, group=self.group)
        tot = t.cpu().numpy()
        for i, f in enumerate(_SUMMED):
            rec[f] = np.where(local, tot[i], rec[f]).astype(np.uint32)
        rec["disease_exists"] = np.where(local, (rec["susceptible"] + rec["exposed"] + rec["infected"]) != 0,
                                         rec["disease_exists"]).astype(np.uint32)
        return rec

    def reset(self, decoupled=True):
        _lib.check(self.lib.esim_reset(self._ctx), self._ctx)
        self._steps = 0
        self._local_ranges = []
        self.coupled_steps = self.free_steps = 0
        self._burst, self._backoff, self._sync_left = 1, 0, 0
        if self.sharded:
            free = bool(decoupled) and self.population.n_shared_buildings == 0 and self.population.n_shared_rooms == 0
            flag = self.torch.tensor([0 if free else 1], dtype=self.torch.int32, device="cuda:%d" % self.params.device)
            self.dist.all_reduce(flag, group=self.group)
            self.mode_free = int(flag.item()) == 0

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
        """0: sequential steps only; 1: chunks as one kernel per step; 2 (default): time-parallel chunks."""
        _lib.check(self.lib.esim_set_pipeline(self._ctx, int(level)), self._ctx)

    def chunk_timing(self):
        ms, ns, nc = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.esim_chunk_timing(self._ctx, C.byref(ms), C.byref(ns), C.byref(nc)), self._ctx)
        return {"chunk_ms": ms.value, "steps": ns.value, "chunks": nc.value}

    def vax_chunk_stats(self):
        """Steps run as chunks under a vaccination programme and how many of those chunks were cut short."""
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
