#!/usr/bin/env python3
"""bench.py -- citizen-timesteps/sec of the per-timestep Citizen update loop on MI355X.

A "step" is one time step (Simulator::step, sim/src/simulator.rs:131) over the whole synthetic
population.  Workload: BASELINE.json configs[4], the 64 M-citizen synthetic UK (preset `uk64m`, 10 seeds,
interventions enabled) -- it fits one GPU, so N=1 runs all of it.  For N>1 the per-GPU work is kept fixed
(weak scaling): a world of N x 64 M citizens / N x 290 000 Output Areas, Output Areas sharded by whole school
catchments, one process per GPU; every rank generates only its own shard.  Shards that share no building
exchange one 96-entry SUM all-reduce per 96 steps (decoupled mode) until a vaccination
programme starts, then two small all-reduces per step.  Population build and upload are outside the timed
region; inputs are resident in HBM when it starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CITIZEN_STEP = 26.0      # state R+W 4, flags 2, home/work/room ids 12, two count gathers 8
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(pop, params, steps, threads):
    """The CPU oracle (oracle/esim_oracle.c) timed on this host on the first `steps` time steps of the same population, its
    per-citizen pass on `threads` host threads (the part the reference runs under rayon).  A reported baseline, not the
    thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(params))
    threads = orc.set_threads(threads)
    t0 = time.perf_counter()
    rec = orc.run(steps)
    dt = time.perf_counter() - t0
    orc.close()
    return {"value": pop.n_citizens * len(rec) / dt, "unit": "citizen-timesteps/s", "cores": threads, "kind": "port",
            "sample": "first %d time steps of the same %d-citizen population, oracle/esim_oracle.c with its per-citizen pass on %d "
                      "thread(s) (bus sorting, exposures and interventions serial), %.1f s" % (len(rec), pop.n_citizens, threads, dt)}, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--preset", default="uk64m", help="synthetic population preset (default: the benchmark workload)")
    ap.add_argument("--cpu-steps", type=int, default=48, help="time steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0 = what this process may use, at most 16)")
    ap.add_argument("--timing-stride", type=int, default=16, help="bracket the per-citizen kernels with HIP events every n-th step")
    ap.add_argument("--small-limit", type=int, default=None, help="override the persistent-kernel hand-over threshold (Infected citizens)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from epidemicsimulator_amd import Population, _lib
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    steps, warmup = args.steps, min(args.warmup, args.steps)
    # weak scaling: the world is `world` times the preset, this rank generates its own shard of it
    spec = _lib.SynthSpec()
    _lib.check(_lib.load().esim_synth_preset(args.preset.encode(), __import__("ctypes").byref(spec)))
    n_total, n_areas = spec.n_citizens * world, spec.n_areas * world
    pop = Population.synthetic_shard(rank, world, args.preset, n_citizens=n_total, n_areas=n_areas)
    params = _lib.default_params(max_steps=max(steps, warmup, 1))
    sim = ShardedSimulator(None, rank, world, params, device_index=local_rank, shard_population=pop)
    if args.small_limit is not None:
        sim.set_small_step_limit(args.small_limit)

    def fence():
        sim.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # W untimed warm-up steps, then back to time step 0 so the timed region is steps 1..K of the workload
    sim.run(warmup)
    fence()
    sim.reset()
    sim.enable_kernel_timing(args.timing_stride)
    fence()
    t0 = time.perf_counter()
    sim.run(steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kt = sim.kernel_timings()
    ks = sim.small_kernel_timing()
    kp = sim.pipeline_timing()
    kc = sim.chunk_timing()
    rec = sim.records(1, steps)

    if rank == 0:
        n_local = sim.population.n_citizens
        value = n_total * steps / elapsed
        # One time step = one pass of the hot path.  While no vaccination programme runs, steps are processed in
        # chunks of <= 96 whose inputs are known ahead: normally ALL steps of a chunk are
        # drawn by one pass of four to six kernels (k_chunk_marks, k_chunk_draw, k_chunk_units, k_chunk_books, and
        # k_chunk_count / k_chunk_scatter while many are Infected; a burst of chunks is bracketed by one HIP event pair
        # on the context's stream); when a chunk does not fit that form it runs as one k_pipe launch per step (sampled
        # event pairs).  Steps that can vaccinate run sequentially:
        # k_small (persistent, timed per launch) or k_infected + k_expose + k_finish (sampled).  Together they
        # carry SURVEY.md 8(d)'s 26 algorithmic bytes per citizen-timestep.
        pipe_steps = kp["steps"]
        seq_steps = steps - pipe_steps - kc["steps"]
        big_steps = seq_steps - ks["steps"]
        parts = {"time-parallel chunk pass (k_chunk_marks, k_chunk_draw, k_chunk_units, k_chunk_books [+ k_chunk_count, "
                 "k_chunk_scatter while many are Infected])": kc["chunk_ms"],
                 "k_pipe": pipe_steps * kp["k_pipe_ms"], "k_small": ks["k_small_ms"],
                 "k_infected+k_expose+k_finish": big_steps * kt["multi_kernel_step_ms"]}
        step_ms = sum(parts.values()) / steps
        dom = max(parts, key=parts.get)
        algo_bytes = ALGO_BYTES_PER_CITIZEN_STEP * n_local
        achieved = algo_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("workload") == args.preset and tj.get("n_gpus") == world:
                traffic = tj.get("step_bytes_per_launch")
        out = {
            "metric": "citizen-timesteps/sec", "value": value, "unit": "citizen-timesteps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 citizen word, u64 Philox thresholds",
            "data": "synthetic",
            "config": {"workload": "%s x %d: %d citizens, %d Output Areas, %d seeds, %d steps, interventions on; "
                                   "Output Areas sharded by school catchment over %d GPU(s)"
                                   % (args.preset, world, n_total, n_areas, spec.n_seeds, steps, world),
                       "citizens_per_gpu": n_local, "seed": int(params.seed),
                       "decoupled_steps": sim.free_steps, "coupled_steps": sim.coupled_steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dom,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": step_ms,
                         "time_parallel_chunks": {"steps": kc["steps"], "chunks": kc["chunks"], "total_ms": kc["chunk_ms"],
                                                  "ms_per_chunk": kc["chunk_ms"] / kc["chunks"] if kc["chunks"] else None},
                         "k_pipe": {"steps": kp["steps"], "steps_timed": kp["steps_timed"], "ms_per_launch": kp["k_pipe_ms"]},
                         "k_small": {"steps": ks["steps"], "total_ms": ks["k_small_ms"],
                                     "ms_per_step": ks["k_small_ms"] / ks["steps"] if ks["steps"] else None},
                         "multi_kernel_steps": {"steps": big_steps, "steps_timed": kt["steps_timed"],
                                                "ms_per_step": kt["multi_kernel_step_ms"]},
                         "note": "avg_launch_ms = device time of the pass divided by the time steps it covers (a chunk pass "
                                 "covers up to 96 steps); algorithmic bytes = 26 B x citizens of the shard per time step. "
                                 "frac >> 1 because the pass never touches most of the model's bytes: only infected citizens "
                                 "and the members of the buildings they stand in are visited, and the remaining work is "
                                 "bound by memory latency and kernel boundaries, not by HBM (DESIGN.md sections 3 and 5)",
                         "wall_algorithmic_GBs": ALGO_BYTES_PER_CITIZEN_STEP * n_total * steps / elapsed / 1e9},
            "final_record": {k: int(rec[k][-1]) for k in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")},
        }
        if world == 1:
            # the whole run against records the CPU oracle produced offline for this preset (tests/golden/make_preset_golden.py):
            # every 50th step up to the steps run here
            gpath = os.path.join(ROOT, "tests", "golden", "oracle_%s_5000.json" % args.preset)
            if os.path.exists(gpath):
                gold = json.load(open(gpath))
                if gold["seed"] == int(params.seed):
                    compared = 0
                    for want in gold["records"]:
                        if want["time_step"] > steps:
                            break
                        got = rec[want["time_step"] - 1]
                        for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated"):
                            if int(got[f]) != want[f]:
                                raise SystemExit("bench: step %d %s = %d, the offline CPU oracle has %d (%s)"
                                                 % (want["time_step"], f, int(got[f]), want[f], gpath))
                        compared += 1
                    out["golden_check"] = {"file": os.path.relpath(gpath, ROOT), "records_compared": compared, "match": True}
        if world == 1 and args.cpu_steps > 0:
            n_thr = args.cpu_threads or min(16, len(os.sched_getaffinity(0)))      # the GPU box gives one GPU's job 16 CPUs
            cb, orc_rec = cpu_baseline(pop, params, min(args.cpu_steps, steps), n_thr)
            out["cpu_baseline"] = cb
            for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated"):
                if not (orc_rec[f] == rec[f][:len(orc_rec)]).all():
                    raise SystemExit("bench: GPU records differ from the CPU oracle in field %s" % f)
            out["cpu_baseline"]["records_match_gpu"] = True
        print(json.dumps(out))
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
