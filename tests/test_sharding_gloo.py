"""world_size-2 gloo test (CPU) of the host-side sharding contract the multi-GPU path relies on:
every rank cuts its own shard, the shared tables line up across ranks, and a SUM all-reduce of the
per-shard infected counts over the shared slots reproduces the whole-population counts."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, errq):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch
        import torch.distributed as dist
        from epidemicsimulator_amd import Population, _lib
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pop = Population.synthetic("york", n_citizens=15000, n_areas=30, citizens_per_school=3000)
        cuts = pop.even_cuts(world)
        sh = pop.shard(cuts, rank)
        # (1) the layout of the exchange buffer is identical on every rank
        sizes = torch.tensor([sh.n_shared_buildings, sh.n_shared_rooms, sh.n_citizens_global], dtype=torch.int64)
        gathered = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(gathered, sizes)
        assert all((g == sizes).all() for g in gathered)
        tot = torch.tensor([sh.n_citizens, sh.n_seeds], dtype=torch.int64)
        dist.all_reduce(tot)
        assert tot.tolist() == [pop.n_citizens, pop.n_seeds]
        # (2) a synthetic "who is infected and at work" state, decided per GLOBAL citizen id
        rng = np.random.default_rng(5)
        infected = rng.random(pop.n_citizens) < 0.05
        lo = sh.citizen_id_base
        mine = infected[lo:lo + sh.n_citizens]
        has_work = sh.home_building != sh.work_building
        cnt_b = np.bincount(sh.work_building[mine & has_work], minlength=sh.n_buildings).astype(np.int32)
        is_school = sh.building_type[sh.work_building] == _lib.SCHOOL
        cnt_r = np.bincount(sh.room[mine & has_work & is_school], minlength=max(1, sh.n_rooms)).astype(np.int32)
        # pack -> all-reduce -> unpack, exactly what k_pack_a / k_unpack_a do
        xb = np.where(sh.shared_building_local >= 0, cnt_b[np.maximum(sh.shared_building_local, 0)], 0).astype(np.int32)
        xr = np.where(sh.shared_room_local >= 0, cnt_r[np.maximum(sh.shared_room_local, 0)], 0).astype(np.int32)
        x = torch.from_numpy(np.concatenate([xb, xr]))
        dist.all_reduce(x)
        x = x.numpy()
        sel = sh.shared_building_local >= 0
        cnt_b[sh.shared_building_local[sel]] = x[:sh.n_shared_buildings][sel]
        sel = sh.shared_room_local >= 0
        cnt_r[sh.shared_room_local[sel]] = x[sh.n_shared_buildings:][sel]
        # whole-population truth, looked up through this shard's citizens
        pw = pop.home_building != pop.work_building
        g_b = np.bincount(pop.work_building[infected & pw], minlength=pop.n_buildings)
        g_sch = pop.building_type[pop.work_building] == _lib.SCHOOL
        g_r = np.bincount(pop.room[infected & pw & g_sch], minlength=pop.n_rooms)
        want_b = g_b[pop.work_building[lo:lo + sh.n_citizens]]
        assert (cnt_b[sh.work_building] == want_b).all()
        m = has_work & is_school
        assert (cnt_r[sh.room[m]] == g_r[pop.room[lo:lo + sh.n_citizens][m]]).all()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        errq.put("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        raise


def test_shared_slot_exchange_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    errq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, errq)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    assert not errs, "\n".join(errs)
    assert all(p.exitcode == 0 for p in procs)


def _uid_worker(rank, world, port, errq):
    """The RCCL unique id cannot be made on rank 0 (as when librccl does not load there): every rank must raise, promptly and
    in step -- none may be left inside the broadcast (the old code raised on rank 0 BEFORE the broadcast)."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import datetime
        import torch.distributed as dist
        from epidemicsimulator_amd import _lib
        from epidemicsimulator_amd.distributed import exchange_unique_id
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

        class FakeLib:                                   # rccl().ok == false on rank 0
            def esim_comm_unique_id(self, buf, cap):
                return -2 if rank == 0 else 0
            def esim_last_error(self, ctx):
                return b"librccl could not be loaded"

        try:
            exchange_unique_id(FakeLib(), dist, rank)
            raise AssertionError("rank %d: no error although rank 0 could not make the id" % rank)
        except _lib.EsimError as ex:
            assert ex.code == -2 and "librccl" in str(ex)
        # the ranks are still in step: the next collective matches
        import torch
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        assert int(t.item()) == world * (world + 1) // 2

        class GoodLib(FakeLib):
            def esim_comm_unique_id(self, buf, cap):
                for i in range(128):
                    buf[i] = (i * 7 + 1) & 0xFF
                return 0
        raw = exchange_unique_id(GoodLib(), dist, rank)
        assert raw == bytes(((i * 7 + 1) & 0xFF) for i in range(128))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        errq.put("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        raise


def test_unique_id_failure_on_rank_0_reaches_every_rank_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    errq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uid_worker, args=(r, 2, port, errq)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    assert not errs, "\n".join(errs)
    assert all(p.exitcode == 0 for p in procs)
