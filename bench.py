#!/usr/bin/env python3
"""bench.py -- citizen-timesteps/sec of the per-timestep Citizen update loop on MI355X.

A "step" is one time step (Simulator::step, sim/src/simulator.rs:131) over the whole synthetic population.
Workload: BASELINE.json configs[4], the 64 M-citizen synthetic UK (preset `uk64m`: 10 seeds, interventions on -- the
lockdown starts around step 3000 and the vaccination programme around step 3240 of its 5000 steps).  It fits one GPU, so
N = 1 runs all of it.  For N > 1 the SAME world is sharded by Output Areas over the N GPUs (strong scaling, the default):
every rank builds the world, keeps the band of the map that is its shard (esim_synth_create_shard) and runs
esim_run_sharded, the library exchanging the commuter counts and the census itself over its own RCCL communicator.
`--scaling weak` keeps the per-GPU work fixed instead (a world of N x the preset).  Population build and upload are outside
the timed region; inputs are resident in HBM when it starts.

The timed region is steps 1..K (after W untimed warm-up steps and a reset).  Whatever K is, the line also carries the FULL
5000-step run of the workload and a run that spends most of its steps under a vaccination programme (`york`, BASELINE.json
configs[1]), each checked against the records the CPU oracle produced offline (tests/golden/), with wall and device time
per step -- they cost milliseconds.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL_BYTES_PER_CITIZEN_STEP = 26.0     # state R+W 4, flags 2, home/work/room ids 12, two count gathers 8
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
N_SIMD = 1024.0                         # 256 CUs x 4 SIMDs
CENSUS = ("susceptible", "exposed", "infected", "recovered", "vaccinated")
# the arithmetic the path computes in -- not a precision claim (RNG contract)
DTYPE = "u32 citizen word; u32 uniform vs ceil(q*2^32) (reference: f64 uniform; tolerance 2^-32 per draw)"


def golden_check(preset, rec, seed):
    """Records of steps 1..len(rec) against the preset's offline oracle run: every record of the first 100 steps, every 50th after."""
    path = os.path.join(ROOT, "tests", "golden", "oracle_%s_5000.json" % preset)
    if not os.path.exists(path):
        return None
    gold = json.load(open(path))
    if gold["seed"] != seed:
        return None
    compared = 0
    for want in gold.get("first_records", []) + gold["records"]:
        t = want["time_step"]
        if t > len(rec):
            continue
        for f in CENSUS:
            if int(rec[t - 1][f]) != want[f]:
                raise SystemExit("bench: %s step %d %s = %d, the offline CPU oracle has %d (%s)" % (preset, t, f, int(rec[t - 1][f]), want[f], path))
        compared += 1
    return {"file": os.path.relpath(path, ROOT), "records_compared": compared, "match": True}


def timed_run(sim, steps, events_in_a_second_run=False):
    """One esim_run of `steps` steps from time step 0: wall time bracketed by synchronisation, device time of the chunk passes
    from HIP events on the context's stream (esim_chunk_timing), how the steps were executed.
    events_in_a_second_run: the wall time is taken on a call WITHOUT the library's HIP events (a burst of one-launch chunks then
    hands its records over through host-visible mirrors and a flag the host polls, instead of two copies and a stream wait), the
    device figures on a second, identical call with them."""
    clock = []
    if events_in_a_second_run:
        sim.reset()
        sim.enable_kernel_timing(0)
        sim.synchronize()
        rec0 = sim.run(steps, clock=clock)   # esim_run returns with the K records on the host
    sim.reset()
    sim.enable_kernel_timing(16)
    sim.synchronize()
    clock2 = []
    rec = sim.run(steps, clock=clock2)       # esim_run returns with the stream drained and the records on the host
    if events_in_a_second_run:
        assert all((rec0[k] == rec[k]).all() for k in rec.dtype.names), "the two identical calls differ"
    wall = clock[0] if events_in_a_second_run else clock2[0]
    kc, kv, kp, ks, km = sim.chunk_timing(), sim.vax_chunk_stats(), sim.pipeline_timing(), sim.small_kernel_timing(), sim.kernel_timings()
    seq_steps = steps - kc["steps"] - kp["steps"]
    device_ms = kc["chunk_ms"] + kp["steps"] * kp["k_pipe_ms"] + ks["k_small_ms"] + (seq_steps - ks["steps"]) * km["multi_kernel_step_ms"]
    return rec, {"steps": steps, "wall_us_per_step": wall / steps * 1e6, "device_us_per_step": device_ms / steps * 1e3,
                 "chunk_steps": kc["steps"], "chunk_passes_device_ms": kc["chunk_ms"], "steps_under_vaccination_in_chunks": kv["steps"],
                 "chunks_cut": kv["cuts"], "k_pipe_steps": kp["steps"], "sequential_steps": seq_steps,
                 "steps_with_vaccination_active": int((rec["vaccination_active"] > 0).sum()), "steps_in_lockdown": int((rec["lockdown"] > 0).sum()),
                 "peak_infected": int(rec["infected"].max()),
                 # what the sparse pass actually works on: one (Infected citizen, step) makes the draws of that citizen's buildings
                 # in that step.  The epidemic is one random realisation -- its size moves the wall time, not the code's speed.
                 "infected_citizen_steps": int(rec["infected"].astype("int64").sum()),
                 "exposures": int(rec["exposures_building"].astype("int64").sum() + rec["exposures_bus"].astype("int64").sum()),
                 "device_ns_per_infected_citizen_step": device_ms * 1e6 / max(1, int(rec["infected"].astype("int64").sum())),
                 "final_record": {k: int(rec[k][-1]) for k in ("time_step",) + CENSUS}}


def cpu_baseline(pop, params, seconds, threads):
    """The reference-shaped CPU path (oracle/esim_refshape.cpp: area-parallel array-of-structs citizens, per-area hash maps,
    mutex-guarded lookup table -- the structure of sim/src/simulator.rs:87-103,167-260) on this host, on a bounded sample of the
    workload: the first 1/16 of its Output Areas (a band of the map, cut with esim_shard_population and run as a population of
    its own), as many steps as fit in about `seconds`.  Returns the figure and (population, records) for the equality check."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _oracle
    cuts = np.asarray([0, max(1, pop.n_areas // 16), pop.n_areas], np.uint32)
    part = pop.shard(cuts, 0)
    sample = type(pop)(home_building=part.home_building, work_building=part.work_building, room=part.room, flags=part.flags,
                       age=part.age, occupation=part.occupation, building_area=part.building_area, building_type=part.building_type,
                       room_building=part.room_building, seeds=part.seeds if part.n_seeds else np.asarray([0], np.uint32), n_areas=part.n_areas)
    t0 = time.perf_counter()
    rsh = _oracle.ReferenceShaped(sample, _oracle.params_from_esim(params), threads)
    build = time.perf_counter() - t0
    recs, spent, steps = [], 0.0, 0
    while spent < seconds and steps < 480:
        t0 = time.perf_counter()
        recs.append(rsh.run(8))
        spent += time.perf_counter() - t0
        steps += 8
    rsh.close()
    rec = np.concatenate(recs)
    return {"value": sample.n_citizens * steps / spent, "unit": "citizen-timesteps/s", "cores": threads, "kind": "port", "shape": "reference-shaped (area-parallel array-of-structs citizens, per-area hash maps, mutex-guarded lookup: oracle/esim_refshape.cpp)",
            "sample": "Output Areas [0, %d) of the same world (%d citizens, a band of the map cut with esim_shard_population and run as a "
                      "population of its own), steps 1..%d, oracle/esim_refshape.cpp on %d threads: %.1f s (+ %.1f s to build its structures)"
                      % (int(cuts[1]), sample.n_citizens, steps, threads, spent, build),
            "reference_published": "4.36e6 citizen-timesteps/s (3.46 M citizens, 32-core Xeon 6138, rayon 40 threads) and 8.90e6 (York, workstation): "
                                   "the reference's own runs, different hardware, real census population (BASELINE.md)"}, sample, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--preset", default="uk64m", help="synthetic population preset (default: the benchmark workload)")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"), help="N > 1: shard ONE world (strong) or a world N times the preset (weak)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time of the reference-shaped baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0 = what this process may use, at most 16)")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "callback"), help="N > 1: the library's own RCCL communicator, or its callback transport into torch.distributed (rehearsal on one GPU)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend of the rendezvous (default gloo: it only carries the unique id, barriers and the timing maximum)")
    ap.add_argument("--no-extra-runs", action="store_true", help="skip the full-length and vaccination-regime runs")
    ap.add_argument("--comm-timeout", type=float, default=120.0, help="N > 1: deadline (s) of the library's waits inside a sharded run; on expiry the communicator is aborted and the line carries value null")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (this pool's driver supports dmabuf IPC only: RCCL needs it)
    import numpy as np
    import torch
    import torch.distributed as dist
    from epidemicsimulator_amd import Population, Simulator, _lib
    from epidemicsimulator_amd.distributed import ShardedSimulator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("ESIM_BENCH_SAME_DEVICE"):      # rehearsal of the multi-rank path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    steps, warmup = args.steps, min(args.warmup, args.steps)
    spec = _lib.SynthSpec()
    _lib.check(_lib.load().esim_synth_preset(args.preset.encode(), __import__("ctypes").byref(spec)))

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend or "gloo"          # the rendezvous only carries the RCCL unique id, the barriers and the timing maximum
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        mult = world if args.scaling == "weak" else 1
        n_total, n_areas = spec.n_citizens * mult, spec.n_areas * mult
        pop = Population.synthetic_shard(rank, world, args.preset, n_citizens=n_total, n_areas=n_areas)
        params = _lib.default_params(max_steps=max(steps, warmup, 1))
        def null_line(reason):
            """The contract's line without a number: a run that could not be set up or did not finish is not measured."""
            return {"metric": "citizen-timesteps/sec", "value": None, "unit": "citizen-timesteps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                    "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                    "config": {"workload": "%s sharded over %d GPUs" % (args.preset, world), "transport": args.transport}, "error": reason}

        sim, note = None, None
        try:
            sim = ShardedSimulator(None, rank, world, params, device_index=local_rank, shard_population=pop, transport=args.transport,
                                   timeout_s=args.comm_timeout)
        except Exception as ex:            # (librccl not loadable, communicator set-up failed, shards that do not belong together)
            note = "%s: %s" % (type(ex).__name__, ex)
        failed = torch.tensor([0 if sim is not None else 1], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(failed)
        if int(failed.item()) > 0:
            # No silent change of transport: a number measured over the host-staged callback transport is not the RCCL number.
            # (`--transport callback` asks for that transport explicitly: rehearsals with several ranks on one GPU.)
            if rank == 0:
                print(json.dumps(null_line("the sharded context could not be set up over transport '%s' on %d of %d ranks (%s)"
                                           % (args.transport, int(failed.item()), world, note or "another rank failed"))))
            if sim is not None:
                sim.close()
            dist.destroy_process_group()
            sys.exit(2)
        transport_used, transport_note = args.transport, None

        def fence():
            sim.synchronize()
            torch.cuda.synchronize()
            dist.barrier()

        try:
            sim.run(warmup)
            fence()
            sim.reset()
            fence()
            t0 = time.perf_counter()
            sim.run(steps)
            fence()
            elapsed = time.perf_counter() - t0
        except Exception as ex:
            # A device-side error reaches every rank in the same collective (all raise together, same code); a peer that died shows
            # as ESIM_ETIMEDOUT after the deadline.  Either way this run measured nothing: say so and leave with an error, without
            # entering another collective that a missing peer would never complete.
            print(json.dumps(dict(null_line("rank %d: %s: %s" % (rank, type(ex).__name__, ex)), rank=rank)), flush=True)
            os._exit(3)
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        shared = torch.tensor([pop.n_citizens, pop.n_shared_buildings, pop.n_shared_rooms], dtype=torch.int64, device=dev)
        gathered = [torch.zeros_like(shared) for _ in range(world)]
        dist.all_gather(gathered, shared)
        rec = sim.records(1, steps)
        if rank == 0:
            out = {
                "metric": "citizen-timesteps/sec", "value": n_total * steps / elapsed, "unit": "citizen-timesteps/s",
                "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": DTYPE,
                "data": "synthetic",
                "config": {"workload": "%s%s: %d citizens, %d Output Areas, %d seeds, %d steps, interventions on; Output Areas sharded in bands of the "
                                       "map over %d GPUs; time-parallel chunks with one round of exchanges per chunk, coupled steps (two exchanges per step) where a chunk cannot run; the library's exchange goes over %s"
                                       % (args.preset, " x %d" % mult if mult > 1 else "", n_total, n_areas, spec.n_seeds, steps, world,
                                          "its own RCCL communicator" if transport_used == "rccl" else "its callback transport (%s)" % backend),
                           "citizens_per_gpu": [int(g[0]) for g in gathered], "shared_buildings": int(gathered[0][1]), "shared_rooms": int(gathered[0][2]),
                           "collectives": sim.collectives(), "chunk_steps": sim.shard_stats()["chunk_steps"],
                           "coupled_steps": sim.shard_stats()["coupled_steps"], "seed": int(params.seed), "transport": transport_used, "transport_note": transport_note},
                "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                             "note": "per-step exchange form: bound by kernel boundaries and collective latency, not by HBM (DESIGN.md 7); the one-GPU line "
                                     "carries the counter-based figure"},
                "final_record": {k: int(rec[k][-1]) for k in ("time_step",) + CENSUS},
            }
            if args.scaling == "strong":
                g = golden_check(args.preset, rec, int(params.seed))
                if g:
                    out["golden_check"] = g
            print(json.dumps(out))
        sim.close()
        dist.destroy_process_group()
        return

    # ---- one GPU: the whole workload on one context ----------------------------------------------------------------
    pop = Population.synthetic(args.preset)
    params = _lib.default_params(max_steps=5000)
    sim = Simulator(pop, params)
    sim.enable_kernel_timing(0)                       # (the warm-up takes the same path as the timed call: no HIP events of the library's)
    sim.run(warmup)                                   # W untimed warm-up steps, then back to time step 0
    rec, info = timed_run(sim, steps, events_in_a_second_run=True)
    elapsed = info["wall_us_per_step"] * steps * 1e-6
    out = {
        "metric": "citizen-timesteps/sec", "value": pop.n_citizens * steps / elapsed, "unit": "citizen-timesteps/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "%s: %d citizens, %d Output Areas, %d seeds, steps 1..%d of 5000, interventions on" % (args.preset, pop.n_citizens, pop.n_areas, spec.n_seeds, steps),
                   "seed": int(params.seed), "timed_region": info},
        "final_record": info["final_record"],
    }
    g = golden_check(args.preset, rec, int(params.seed))
    if g:
        out["golden_check"] = g
    full = info
    if not args.no_extra_runs:
        if steps != 5000:
            rec5, full = timed_run(sim, 5000)
            full["golden_check"] = golden_check(args.preset, rec5, int(params.seed))
            full["value"] = pop.n_citizens * 5000 / (full["wall_us_per_step"] * 5000 * 1e-6)
        out["full_run"] = dict(full, workload="%s, all 5000 steps" % args.preset)
    # ---- roofline of the dominant kernel of the chunk pass.  Its average launch duration is measured HERE: one more full run
    # with a HIP event in front of every chunk kernel on the context's stream (esim_chunk_kernel_timings).  What a launch moves
    # and issues comes from the rocprofv3 passes of the same command kept under profiles/ (current_<preset>.json, written by
    # profiles/summarize_r03.py): HBM-side bytes request-size exact (TCC_EA0_RDREQ_32B/_64B/_128B, WRREQ/_64B: on gfx950 every
    # read request is 128 B, whatever the access width -- profiles/*_tcc_calibration.md), SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES,
    # and the counting build's tallies (draws, Philox blocks, useful bytes).  The pass is sparse: it never touches most of
    # SURVEY 8(d)'s 26 B x citizens (kept as `model_bytes` for context), so it is priced against the two roofs it can hit --
    # HBM lines and vector issue -- and `bound` names the nearer one.
    rf = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
    ppath = os.path.join(ROOT, "profiles", "current_%s.json" % args.preset)
    if os.path.exists(ppath) and not args.no_extra_runs:
        prof = json.load(open(ppath))
        sim.enable_chunk_kernel_timing(True)
        _, kinfo = timed_run(sim, 5000)
        kt = sim.chunk_kernel_timings()
        sim.enable_chunk_kernel_timing(False)
        dev_ms = sum(v["ms"] for v in kt.values())
        dom = max(kt, key=lambda k: kt[k]["ms"])
        pk = prof["kernels"].get("k_chunk_" + dom) or prof["kernels"].get("k_" + dom)
        if pk and kt[dom]["calls"] and "hbm_bytes_exact" in pk:      # (a round-3 profile summary: profiles/summarize_r03.py)
            dur_s = kt[dom]["ms"] * 1e-3 / kt[dom]["calls"]                      # live average launch duration
            hbm_bytes = pk["hbm_bytes_exact"] / pk["calls"]
            valu_cyc = pk["SQ_ACTIVE_INST_VALU"] * 4.0 / pk["calls"]             # SIMD-cycles with a VALU instruction active, per launch
            wave_cyc = pk["SQ_WAVE_CYCLES"] * 4.0 / pk["calls"]
            hbm = {"achieved_GBs": hbm_bytes / dur_s / 1e9, "peak_GBs": HBM_PEAK_GBS, "frac": hbm_bytes / dur_s / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": hbm_bytes,
                   "read_requests_32B_64B_128B_per_launch": [x / pk["calls"] for x in pk["read_requests_32B_64B_128B"]]}
            valu = {"busy_simd_cycles_per_launch": valu_cyc, "frac_at_2.4GHz": valu_cyc / (N_SIMD * 2.4e9 * dur_s), "frac_at_2.1GHz": valu_cyc / (N_SIMD * 2.1e9 * dur_s),
                    "resident_wavefronts_per_simd_at_2.4GHz": wave_cyc / (N_SIMD * 2.4e9 * dur_s), "wavefronts_per_simd_that_fit": 5.0,   # (96 registers: tests/test_kernel_resources.py; the grid is 16 per SIMD over a launch)
                    "note": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x clock x launch duration); the chip holds 2.1-2.4 GHz under this load"}
            near_valu = valu["frac_at_2.4GHz"] >= hbm["frac"]
            rf = {"bound": "valu" if near_valu else "hbm",
                  "achieved": valu_cyc / dur_s / 1e9 if near_valu else hbm["achieved_GBs"], "peak": N_SIMD * 2.4 if near_valu else HBM_PEAK_GBS,
                  "unit": "G SIMD-cycles/s with a vector instruction active (peak: 1024 SIMDs x 2.4 GHz)" if near_valu else "GB/s",
                  "frac": valu["frac_at_2.4GHz"] if near_valu else hbm["frac"], "traffic": hbm_bytes,
                  "kernel": "k_chunk_" + dom, "share_of_chunk_pass_device_time": kt[dom]["ms"] / dev_ms, "launches": kt[dom]["calls"], "avg_launch_ms": dur_s * 1e3,
                  "avg_launch_ms_in_the_profile": pk["avg_us"] * 1e-3, "hbm": hbm, "valu": valu,
                  "counters_from": os.path.relpath(ppath, ROOT) + " <- " + prof["tag"],
                  "model_bytes": MODEL_BYTES_PER_CITIZEN_STEP * pop.n_citizens, "model_note": "SURVEY 8(d): 26 B x citizens per time step -- what a DENSE pass would move; "
                  "the sparse pass moves %.4f of it per step" % (prof["hbm_bytes_exact_per_step"] / (MODEL_BYTES_PER_CITIZEN_STEP * pop.n_citizens)),
                  "device_ms_per_kernel": {k: round(v["ms"], 3) for k, v in kt.items() if v["calls"]}}
            w = prof.get("work")
            if w:
                run_s = dev_ms * 1e-3
                rf.update({"draws_per_s": w["counts"]["draws"] / run_s, "philox_blocks_per_s": w["counts"]["philox_blocks"] / run_s,
                           "member_slot_pairs_per_s": w["counts"]["pairs"] / run_s, "useful_bytes_per_run": w["useful_bytes"],
                           "useful_over_fetched": w["useful_bytes"] / (prof["hbm_bytes_exact_per_step"] * prof["steps"]),
                           "useful_bytes_model": w["useful_bytes_model"]})
    out["roofline"] = rf
    if not args.no_extra_runs and args.preset != "york":
        # a run that spends most of its steps under a vaccination programme: BASELINE.json configs[1]
        ypop = Population.synthetic("york")
        ysim = Simulator(ypop, _lib.default_params(max_steps=5000))
        ysim.run(24)
        yrec, yinfo = timed_run(ysim, 5000)
        yinfo["golden_check"] = golden_check("york", yrec, int(params.seed))
        yinfo["value"] = ypop.n_citizens * 5000 / (yinfo["wall_us_per_step"] * 5000 * 1e-6)
        out["vaccination_run"] = dict(yinfo, workload="york: %d citizens, 5000 steps, vaccination programme from step %d"
                                                      % (ypop.n_citizens, int(np.argmax(yrec["vaccination_active"] > 0)) + 1))
        ysim.close()
    if args.cpu_seconds > 0:
        n_thr = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))      # the GPU box gives one GPU's job 16 CPUs
        cb, sample, cpu_rec = cpu_baseline(pop, params, args.cpu_seconds, n_thr)
        # the same sample on the GPU: the records must be identical
        ssim = Simulator(sample, _lib.default_params(max_steps=len(cpu_rec)))
        srec = ssim.run(len(cpu_rec))
        ssim.close()
        for f in CENSUS + ("exposures_building", "exposures_bus"):
            if not (srec[f] == cpu_rec[f]).all():
                raise SystemExit("bench: GPU records of the baseline's sample differ from the CPU's in field %s" % f)
        cb["records_match_gpu"] = True
        out["cpu_baseline"] = cb
    print(json.dumps(out))
    sim.close()


if __name__ == "__main__":
    main()
