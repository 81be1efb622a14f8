"""One rank of a sharded run (spawned by test_sharded_gpu.py / usable by hand):
   RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment; all ranks use cuda:<LOCAL_RANK or 0>.
Compares this rank's records and its slice of the per-citizen state with the whole-population oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    import _oracle
    from epidemicsimulator_amd import Population, _lib
    from epidemicsimulator_amd.distributed import ShardedSimulator

    cfg = json.loads(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    for k, v in cfg.get("env_by_rank", {}).get(str(rank), {}).items():      # (e.g. ESIM_HASH_LOG2 on ONE rank: shards that disagree)
        os.environ[k] = str(v)
    import datetime
    dist.init_process_group(cfg["backend"], rank=rank, world_size=world, timeout=datetime.timedelta(seconds=cfg.get("pg_timeout", 300)))
    if cfg.get("mode") == "mismatch":
        return mismatch(cfg, rank, world, dist)
    if cfg.get("mode") == "inject":
        return inject(cfg, rank, world, dist)
    pop = Population.synthetic(cfg.get("preset", "york"), **cfg["spec"])
    ep = _lib.default_params(**cfg["params"])
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    transport = cfg.get("transport", "callback")   # several ranks on one GPU: the library's exchange goes through gloo
    if cfg.get("cuts") == "even":                  # cuts through school catchments
        sim = ShardedSimulator(pop, rank, world, ep, device_index=dev, cuts=pop.even_cuts(world), transport=transport)
    elif cfg.get("cuts") == "generated":           # esim_synth_create_shard: the world generated and cut inside the library
        sim = ShardedSimulator(None, rank, world, ep, device_index=dev, transport=transport,
                               shard_population=Population.synthetic_shard(rank, world, cfg.get("preset", "york"), **cfg["spec"]))
    else:                                          # the least-crossed cuts near an even split
        sim = ShardedSimulator(pop, rank, world, ep, device_index=dev, transport=transport)
    if cfg.get("expect_shared", True) and world > 1:
        assert sim.population.n_shared_buildings > 0 and sim.population.n_shared_rooms > 0
    if cfg.get("pipeline") is not None:
        sim.set_pipeline(cfg["pipeline"])          # 0: coupled steps only
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    done = 0
    while done < cfg["steps"]:
        n = min(cfg["chunk"], cfg["steps"] - done)
        sim.run(n)
        sim.synchronize()
        got = sim.records(done + 1, n)
        want = orc.run(n)
        for f in _lib.RECORD_FIELDS:
            if f == "reserved":
                continue
            if not (got[f] == want[f]).all():
                i = int(np.argmax(got[f] != want[f]))
                raise AssertionError("rank %d: %s differs at step %d: %d vs oracle %d"
                                     % (rank, f, done + i + 1, int(got[f][i]), int(want[f][i])))
        st, ost = sim.download_state(), orc.state()
        lo, hi = sim.population.citizen_id_base, sim.population.citizen_id_base + sim.population.n_citizens
        for k in ("status", "timer", "on_bus", "eligible"):
            assert (st[k] == ost[k][lo:hi]).all(), (rank, k, done)
        done += n
    if cfg.get("expect"):
        last = got[-1]
        for k, v in cfg["expect"].items():
            assert int(last[k]) >= v, (k, int(last[k]), v)
    n_coll = sim.collectives()
    st = sim.shard_stats()
    if cfg.get("pipeline") is None and world > 1:
        assert st["chunk_steps"] > 0, st                      # the default form draws chunks wherever it can
    if cfg.get("expect_coupled"):
        assert st["coupled_steps"] >= cfg["expect_coupled"], st
    if cfg.get("expect_repairs"):
        import ctypes as C
        nr = C.c_uint64(0)
        _lib.check(sim.lib.esim_vax_repair_stats(sim._ctx, C.byref(nr)), sim._ctx)
        assert nr.value >= cfg["expect_repairs"], ("plans repaired", nr.value)
    dist.barrier()
    sim.close()
    dist.destroy_process_group()
    print("rank %d ok: %d steps (%d in chunks, %d coupled), %d collectives, %d local citizens, %d shared buildings, %d shared rooms"
          % (rank, cfg["steps"], st["chunk_steps"], st["coupled_steps"], n_coll, hi - lo, sim.population.n_shared_buildings, sim.population.n_shared_rooms))


def _small(cfg, rank, world, shard_of=None):
    from epidemicsimulator_amd import Population, _lib
    from epidemicsimulator_amd.distributed import ShardedSimulator
    pop = Population.synthetic("york", **cfg["spec"])
    ep = _lib.default_params(**cfg["params"])
    shard = pop.shard(pop.even_cuts(world), rank if shard_of is None else shard_of)
    return ShardedSimulator(None, rank, world, ep, device_index=0, shard_population=shard, transport="callback")


def mismatch(cfg, rank, world, dist):
    """Rank 1 uploads shard 0 again: the set-up's layout check must refuse the communicator on every rank."""
    from epidemicsimulator_amd import _lib
    try:
        _small(cfg, rank, world, shard_of=0)
    except _lib.EsimError as ex:
        assert ex.code == -1 and "shard" in str(ex), ex
        print("rank %d ok: refused (%s)" % (rank, str(ex)[:120]))
        return
    raise AssertionError("rank %d: a communicator over shards that do not belong together was accepted" % rank)


def inject(cfg, rank, world, dist):
    """One rank raises a device-side ESIM_ERANGE in the middle of the run: EVERY rank must return that code from the same
    esim_run_sharded call, within seconds -- nobody is left inside a collective."""
    import time
    from epidemicsimulator_amd import _lib
    sim = _small(cfg, rank, world)
    if cfg.get("pipeline") is not None:
        sim.set_pipeline(cfg["pipeline"])
    sim.run(cfg["steps"])                               # a healthy stretch first
    if rank == cfg["bad_rank"]:
        _lib.check(sim.lib.esim_debug_inject_error(sim._ctx, -5), sim._ctx)
    t0 = time.time()
    try:
        sim.run(cfg["steps"])
    except _lib.EsimError as ex:
        assert ex.code == -5, ex
        assert time.time() - t0 < 30.0
        print("rank %d ok: run() raised ERANGE after %.2f s (%s)" % (rank, time.time() - t0, "own error" if rank == cfg["bad_rank"] else "a peer's"))
        sim.close()
        return
    raise AssertionError("rank %d: run() returned although rank %d is in an error state" % (rank, cfg["bad_rank"]))


if __name__ == "__main__":
    main()
