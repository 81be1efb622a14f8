"""Host-side mirror of the reference's `sim::simulator::Simulator` over libesim.

Same names, argument meaning and side effects as the reference API that `run` and
`visualisation` call (sim/src/simulator.rs):
  Simulator.step()              -> bool      simulator.rs:131-152  (False = epidemic over)
  Simulator.simulate(output)    -> None      simulator.rs:108-127  (progress print every 50 steps,
                                             then the four JSON files of statistics.rs:113-150)
The compute is the HIP library; this file only marshals.
"""
import ctypes as C
import json
import os
import time

import numpy as np

from . import _lib
from .population import Population

DEBUG_ITERATION_PRINT = 50  # sim/src/config.rs:34


class StatisticsRecorder:
    """What `StatisticsRecorder` (sim/src/statistics.rs:97-204) keeps, filled from step results."""

    def __init__(self):
        self.global_stats = []       # StatisticEntry dicts, statistics.rs:208-215
        self.timer_entries = []      # per step {"Generate Exposures":..,"Apply Exposures":..,"Apply Interventions":..,"total":..}
        self.memory_usage_entries = []
        self.exposures_all = []      # exposures per time step ("All" series, statistics.rs:119-136)

    def push(self, rec, timings=None):
        self.global_stats.append({k: rec[k] for k in
                                  ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")})
        self.exposures_all.append(rec["exposures_building"] + rec["exposures_bus"])
        if timings is not None:
            self.timer_entries.append(timings)
        self.memory_usage_entries.append(_memory_usage())

    def push_block(self, arr):
        """The same for a block of records (structured array of esim_step_result) that came back from one
        device-resident run: no per-step Python work beyond list building."""
        cols = [arr[k].tolist() for k in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")]
        keys = ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")
        self.global_stats.extend(dict(zip(keys, row)) for row in zip(*cols))
        self.exposures_all.extend((arr["exposures_building"].astype(np.int64) + arr["exposures_bus"]).tolist())
        self.memory_usage_entries.extend([_memory_usage()] * len(arr))

    def dump_to_file(self, directory, per_output_area=None):
        """statistics.rs:113-150. `dump_to_file` calls next() first, which appends one all-zero
        trailing StatisticEntry (Q14) -- reproduced so downstream notebooks see the same shape."""
        os.makedirs(directory, exist_ok=True)          # fs::create_dir_all(directory), statistics.rs:116
        stats = list(self.global_stats)
        stats.append({"time_step": len(stats) + 1, "susceptible": 0, "exposed": 0, "infected": 0,
                      "recovered": 0, "vaccinated": 0})
        with open(directory + "exposures.json", "w") as f:
            # "All"/"All": exposures per time step.  (The reference overwrites this entry with every place's own series while
            # it drains its map, statistics.rs:123-125, so what it leaves there is one arbitrary place's; the total is what
            # the name says.)  "OutputArea": statistics.rs:127-130.
            doc = {"All": {"All": self.exposures_all}}
            if per_output_area is not None:
                doc["OutputArea"] = per_output_area
            json.dump(doc, f)
        with open(directory + "timings.json", "w") as f:
            json.dump(self.timer_entries, f)
        with open(directory + "memory.json", "w") as f:
            json.dump(self.memory_usage_entries, f)
        with open(directory + "global_stats.json", "w") as f:
            json.dump(stats, f)


def _memory_usage():
    # config.rs:42-47 (VM size of the process, GB)
    try:
        with open("/proc/self/statm") as f:
            pages = int(f.read().split()[0])
        return "%.2f GB" % (pages * os.sysconf("SC_PAGE_SIZE") / 1024 / 1024 / 1024.0)
    except OSError:
        return "0.00 GB"


class Simulator:
    """`Simulator::from(builder)` (simulator.rs:601-644): takes a built population."""

    def __init__(self, population, params=None, area_code="synthetic", record_timings=False, area_codes=None):
        """record_timings: keep the reference's per-step function timers (statistics.rs:46-95,
        simulator.rs:137,140,143) -- GPU time of the three phases from HIP events; costs one
        synchronisation per step, so it is off unless asked for."""
        self.lib = _lib.load()
        self.area_code = area_code
        self.area_codes = area_codes          # Output Area codes (reference_io), for the keys of exposures.json
        self.population = population
        self.params = params if params is not None else _lib.default_params()
        self.current_population = population.n_citizens
        self.statistics_recorder = StatisticsRecorder()
        self._ctx = C.c_void_p()
        _lib.check(self.lib.esim_create(C.byref(self.params), C.byref(self._ctx)))
        ps = population.as_struct()
        _lib.check(self.lib.esim_upload_population(self._ctx, C.byref(ps)), self._ctx)
        self._steps = 0
        self.record_timings = bool(record_timings)
        self._last_phase = None
        if self.record_timings:
            self.enable_phase_timing(True)
            self._last_phase = self.phase_timings()

    # -- reference API -----------------------------------------------------------------
    def step(self):
        """Applies a single time step; returns False if the disease has finished."""
        r = _lib.StepResult()
        _lib.check(self.lib.esim_step(self._ctx, C.byref(r)), self._ctx)
        self._steps += 1
        rec = r.as_dict()
        timings = None
        if self.record_timings:
            now = self.phase_timings()
            timings = {k: now[k] - self._last_phase[k] for k in now}
            self._last_phase = now
        self.statistics_recorder.push(rec, timings)
        self.last = rec
        return bool(rec["disease_exists"])

    def simulate(self, output_name):
        """simulator.rs:108-127: steps until the disease is gone or max_time_step, a progress line after the steps
        with index 0, 50, 100, ... and the statistics dump.  With record_timings the loop is the reference's step by
        step; otherwise the steps between two progress lines are one device-resident run (esim_run with
        stop_when_done), which produces the same records without a host round trip per step."""
        start = time.time()
        max_time_step = int(self.params.max_steps)
        # A context that has already run (step(), run(), load_checkpoint()) continues from its own clock: the loop index of
        # simulator.rs:114 is the absolute step index self._steps, so the budget and the progress cadence stay the run's.
        if self.record_timings:
            while self._steps < max_time_step:
                time_step = self._steps
                if not self.step():
                    break
                if time_step % DEBUG_ITERATION_PRINT == 0:
                    self._progress(start)
                    start = time.time()
        else:
            self.enable_chunk_kernel_timing(True)       # feeds the phase keys of timings.json (below); off again when simulate() returns
            while self._steps < max_time_step:
                # the next progress line follows the step with index 0 (mod 50)
                done = self._steps
                n = 1 if done == 0 else DEBUG_ITERATION_PRINT - (done - 1) % DEBUG_ITERATION_PRINT
                n = min(n, max_time_step - done)
                t_block = time.time()
                arr = self.run(n, stop_when_done=True)
                if len(arr) == 0:
                    break
                # The phases of simulator.rs:137-143 do not exist separately in a device-resident run: a chunk pass works on up to
                # 96 steps at once.  Their keys are APPORTIONED: the device time of the block's chunk kernels (HIP events in front of
                # every kernel, esim_chunk_kernel_timings) by kernel -- marks + fold -> "Generate Exposures", draw + units -> "Apply
                # Exposures", plan / decisions / counts / books / scatter -> "Apply Interventions" -- spread evenly over the block's
                # steps; "total" is the block's wall time spread the same way.  Blocks whose steps ran in the sequential form (the step
                # that starts the vaccination programme) carry "total" only.
                per_step = (time.time() - t_block) / len(arr)
                entry = {"total": per_step}
                phases = self.chunk_phase_seconds()
                if sum(phases.values()) > 0.0:
                    entry = dict({k: v / len(arr) for k, v in phases.items()}, total=per_step)
                self.statistics_recorder.timer_entries.extend(dict(entry) for _ in range(len(arr)))
                self.last = {k: int(arr[k][-1]) for k in arr.dtype.names}
                if not self.last["disease_exists"]:
                    break
                if (self._steps - 1) % DEBUG_ITERATION_PRINT == 0:
                    self._progress(start)
                    start = time.time()
        if not self.record_timings:
            self.enable_chunk_kernel_timing(False)
        self.statistics_recorder.dump_to_file(output_name, self.exposures_per_output_area(self.area_codes))

    def _progress(self, start):
        print("Completed %3d time steps, in: %6s seconds  Statistics: %s,   Memory usage: %s" % (
            DEBUG_ITERATION_PRINT, "%.2f" % (time.time() - start), self.last, _memory_usage()))

    # -- device-resident loop ----------------------------------------------------------
    def run(self, n_steps, stop_when_done=False, clock=None):
        """`n_steps` of the loop of simulate() without host round trips; returns the records as a
        structured numpy array (fields of esim_step_result).  clock: a list that receives the wall seconds of the library
        call alone (esim_run: enqueue, device work, read-back of the records), without this method's own bookkeeping."""
        buf = (_lib.StepResult * max(1, n_steps))()
        n_done = C.c_uint32(0)
        t0 = time.perf_counter()
        rc = self.lib.esim_run(self._ctx, n_steps, int(stop_when_done), buf, C.byref(n_done))
        if clock is not None:
            clock.append(time.perf_counter() - t0)
        _lib.check(rc, self._ctx)
        self._steps += n_done.value
        arr = np.frombuffer(buf, dtype=RECORD_DTYPE, count=n_done.value).copy()
        self.statistics_recorder.push_block(arr)
        return arr

    def synchronize(self):
        _lib.check(self.lib.esim_synchronize(self._ctx), self._ctx)

    def reset(self):
        _lib.check(self.lib.esim_reset(self._ctx), self._ctx)
        self.statistics_recorder = StatisticsRecorder()
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

    # -- read access the reference's callers use ----------------------
    def citizen_output_area_lookup(self):
        """`Simulator::citizen_output_area_lookup` (simulator.rs:97, rebuilt at :200-231) as two arrays: the Output
        Area every citizen currently stands in and its position in that area's `citizens` list.  The reference's
        order inside an area depends on HashMap iteration; here it is ascending global index."""
        st = self.download_state()
        area = self.population.building_area[st["current_building"]].astype(np.uint32)
        order = np.lexsort((np.arange(area.size), area))
        local = np.empty(area.size, np.uint32)
        start = np.concatenate([[0], np.cumsum(np.bincount(area, minlength=self.population.n_areas))])
        local[order] = (np.arange(area.size) - start[area[order]]).astype(np.uint32)
        return area, local

    def exposure_events(self):
        """Every exposure so far as (citizen, time_step, on_bus) arrays, sorted by (time step, citizen): the
        `add_exposure` calls of the run (statistics.rs:181-195)."""
        n = C.c_uint32(0)
        rc = self.lib.esim_download_exposure_log(self._ctx, None, None, None, 0, C.byref(n))
        if rc not in (_lib.ESIM_OK, _lib.ESIM_ERANGE):
            _lib.check(rc, self._ctx)
        cit, step, bus = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint8)
        if n.value:
            p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
            _lib.check(self.lib.esim_download_exposure_log(self._ctx, p(cit, C.c_uint32), p(step, C.c_uint32), p(bus, C.c_uint8),
                                                           n.value, C.byref(n)), self._ctx)
        order = np.lexsort((cit, step))
        return cit[order], step[order], bus[order]

    def exposures_per_output_area(self, area_codes=None):
        """What `exposures.json` holds under "OutputArea" (statistics.rs:119-136,156-171): per Output Area the number of
        building exposures in every time step that had any, in time order.  An exposure in a building is credited to the
        building's area (statistics.rs:186-190), which simulator.rs:324 makes the area the citizen stands in at that
        step: its work building's from the "starts work" arm to the "goes home" arm of an unlocked day
        (citizen.rs:176-206), its home's otherwise."""
        cit, step, bus = self.exposure_events()
        rec = self.records_so_far()
        pop = self.population
        # replay the schedule: the arm of step s runs iff no lockdown was in force, i.e. the record of step s - 1 has none
        at_work = np.zeros(len(rec) + 1, bool)
        cur = False
        for s in range(1, len(rec) + 1):
            if s == 1 or not rec["lockdown"][s - 2]:
                h = s % 24
                if h == self.params.start_hour:
                    cur = True
                elif h == self.params.end_hour:
                    cur = False
            at_work[s] = cur
        b = bus == 0
        c, s = cit[b], step[b]
        has_work = pop.work_building[c] != pop.home_building[c]
        where = np.where(at_work[s] & has_work, pop.work_building[c], pop.home_building[c])
        area = pop.building_area[where]
        out = {}
        order = np.lexsort((s, area))
        a_sorted, s_sorted = area[order], s[order]
        if len(a_sorted):
            key = a_sorted.astype(np.int64) * (len(rec) + 2) + s_sorted
            uniq, counts = np.unique(key, return_counts=True)
            for k, n in zip(uniq.tolist(), counts.tolist()):
                a = k // (len(rec) + 2)
                out.setdefault(area_codes[a] if area_codes else "OA%07d" % a, []).append(n)
        return out

    def records_so_far(self):
        n = self._steps
        buf = (_lib.StepResult * max(1, n))()
        _lib.check(self.lib.esim_read_records(self._ctx, 1, n, buf), self._ctx)
        return np.frombuffer(buf, dtype=RECORD_DTYPE, count=n).copy()

    # -- checkpoint / resume --------------------------------------------------------------
    def save_checkpoint(self, path):
        """Everything needed to continue this run later, bit for bit (esim_checkpoint_save), as one file."""
        size = C.c_size_t(0)
        _lib.check(self.lib.esim_checkpoint_size(self._ctx, C.byref(size)), self._ctx)
        buf = np.empty(size.value, np.uint8)
        _lib.check(self.lib.esim_checkpoint_save(self._ctx, buf.ctypes.data_as(C.c_void_p), size.value), self._ctx)
        with open(path, "wb") as f:
            buf.tofile(f)

    def load_checkpoint(self, path):
        """Continue a run saved by `save_checkpoint` with the same population and parameters."""
        buf = np.fromfile(path, np.uint8)
        _lib.check(self.lib.esim_checkpoint_restore(self._ctx, buf.ctypes.data_as(C.c_void_p), buf.size), self._ctx)
        # the number of steps done is in the control block; find it through the records
        probe = _lib.StepResult()
        lo, hi = 0, int(self.params.max_steps)
        while lo < hi:                                           # records are written in order: last one with a time step
            mid = (lo + hi + 1) // 2
            _lib.check(self.lib.esim_read_records(self._ctx, mid, 1, C.byref(probe)), self._ctx)
            if probe.time_step == mid:
                lo = mid
            else:
                hi = mid - 1
        self._steps = lo
        self.statistics_recorder = StatisticsRecorder()
        if lo:
            self.statistics_recorder.push_block(self.records_so_far())

    def infected_per_area(self):
        """Infected citizens per Output Area where they currently stand (the heat-map `visualisation` draws from
        `output_areas[..].citizens`, run/src/main.rs:246-259)."""
        st = self.download_state()
        area = self.population.building_area[st["current_building"]]
        return np.bincount(area[st["status"] == _lib.INFECTED], minlength=self.population.n_areas)

    def enable_kernel_timing(self, stride):
        _lib.check(self.lib.esim_enable_kernel_timing(self._ctx, int(stride)), self._ctx)

    def kernel_timings(self):
        ms = C.c_double(0)
        n = C.c_uint32(0)
        _lib.check(self.lib.esim_kernel_timings(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return {"multi_kernel_step_ms": ms.value, "steps_timed": n.value}

    def set_small_step_limit(self, max_infected):
        """Steps with at most this many Infected citizens run in the persistent single-workgroup kernel
        (0 = always use the multi-workgroup kernels)."""
        _lib.check(self.lib.esim_set_small_step_limit(self._ctx, int(max_infected)), self._ctx)

    def set_tiny_chunk_limit(self, max_pairs):
        """Time-parallel chunks with at most this many (Infected citizen, step) pairs run as one launch of one workgroup
        (0 = always the wide form; default 2048)."""
        _lib.check(self.lib.esim_set_tiny_chunk_limit(self._ctx, int(max_pairs)), self._ctx)

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
        nr = C.c_uint64(0)
        _lib.check(self.lib.esim_vax_repair_stats(self._ctx, C.byref(nr)), self._ctx)
        return {"steps": ns.value, "cuts": nc.value, "repairs": nr.value}

    def pipeline_timing(self):
        ms, nt, nr = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.lib.esim_pipeline_timing(self._ctx, C.byref(ms), C.byref(nt), C.byref(nr)), self._ctx)
        return {"k_pipe_ms": ms.value, "steps_timed": nt.value, "steps": nr.value}

    def small_kernel_timing(self):
        ms, n = C.c_double(0), C.c_uint64(0)
        _lib.check(self.lib.esim_small_kernel_timing(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return {"k_small_ms": ms.value, "steps": n.value}

    def enable_chunk_kernel_timing(self, on=True):
        _lib.check(self.lib.esim_enable_chunk_kernel_timing(self._ctx, int(on)), self._ctx)

    def chunk_kernel_timings(self):
        """Device milliseconds and launches per kernel of the chunk pass since the last call (esim_chunk_kernel_timings)."""
        n = len(_lib.CHUNK_KERNELS)
        ms, calls = (C.c_double * n)(), (C.c_uint64 * n)()
        _lib.check(self.lib.esim_chunk_kernel_timings(self._ctx, ms, calls), self._ctx)
        return {k: {"ms": ms[i], "calls": int(calls[i])} for i, k in enumerate(_lib.CHUNK_KERNELS)}

    def chunk_phase_seconds(self):
        """The same grouped under the reference's three timer labels (simulator.rs:137,140,143), in seconds."""
        out = {"Generate Exposures": 0.0, "Apply Exposures": 0.0, "Apply Interventions": 0.0}
        for k, v in self.chunk_kernel_timings().items():
            if k == "tiny":
                # one launch holds all three phases: shared out by the kernel's own stage timers (tools/tiny_stages.py on uk64m:
                # entries and keys 10 %, the items' draws 40 %, census ahead, decisions and books 50 %)
                for phase, share in _lib.TINY_PHASE_SHARES.items():
                    out[phase] += share * v["ms"] * 1e-3
                continue
            out[_lib.PHASE_OF_KERNEL.get(k, "Apply Interventions")] += v["ms"] * 1e-3
        return out

    def enable_phase_timing(self, on=True):
        _lib.check(self.lib.esim_enable_phase_timing(self._ctx, int(on)), self._ctx)

    def phase_timings(self):
        t = (C.c_double * 4)()
        _lib.check(self.lib.esim_phase_timings(self._ctx, t), self._ctx)
        return {"Generate Exposures": t[0], "Apply Exposures": t[1], "Apply Interventions": t[2], "total": t[3]}

    def close(self):
        if self._ctx:
            self.lib.esim_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RECORD_DTYPE = np.dtype([(n, np.uint32) for n in _lib.RECORD_FIELDS])
