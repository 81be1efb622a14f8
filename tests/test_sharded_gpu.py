"""The Output-Area sharded path (esim_run_sharded: three device phases per step around two SUM all-reduces that the library
issues itself) against the whole-population oracle.  Ranks share the one GPU of the test box, so the library's exchange goes
through its callback transport into gloo; on the 8-GPU node bench.py runs the same loop over the library's own RCCL
communicator, whose call path test_rccl_communicator_of_one_rank exercises here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, cfg, timeout=420):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_worker.py"), json.dumps(cfg)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    return outs


AGGRESSIVE = dict(exposure_chance=0.004, vaccination_rate=40, vaccination_threshold=0.02, lockdown_threshold=0.03,
                  mask_pt_threshold=0.005, mask_everywhere_threshold=0.01, seed=77, max_steps=700)


def test_two_shards_match_oracle():
    # cuts through school catchments => shared buildings / rooms: the commuter exchange in every step
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=360, chunk=120, expect=dict(vaccinated=1))
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_two_and_three_shards_all_gather_exchange_match_oracle():
    # the commuter exchange of the chunk form is an all-to-all of owner-addressed segments by default (a record goes to the shards
    # that have members in its building); ESIM_XS_MODE=gather keeps round 2's all-gather of every shard's records to every shard
    # for the A/B -- same records either way
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=360, chunk=120, expect=dict(vaccinated=1),
               env_by_rank={str(r): {"ESIM_XS_MODE": "gather"} for r in range(3)})
    assert all("ok" in o for o in launch(2, cfg))
    cfg3 = dict(cfg, spec=dict(n_citizens=9000, n_areas=13, citizens_per_school=3000, n_seeds=16), params=dict(AGGRESSIVE, seed=9), steps=240)
    assert all("ok" in o for o in launch(3, cfg3))


def test_two_shards_coupled_steps_only_match_oracle():
    # the form every step can take: three device phases around two exchanges per step (pipeline level 0)
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=360, chunk=120, expect=dict(vaccinated=1), pipeline=0)
    outs = launch(2, cfg)
    assert all("ok" in o and "(0 in chunks" in o for o in outs)


def test_three_uneven_shards_match_oracle():
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=9000, n_areas=13, citizens_per_school=3000, n_seeds=16),
               params=dict(AGGRESSIVE, seed=9), steps=240, chunk=120)
    outs = launch(3, cfg)
    assert all("ok" in o for o in outs)


def test_least_crossed_cuts_through_the_vaccination_programme_match_oracle():
    cfg = dict(backend="gloo", cuts="clean", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=600, chunk=150, expect=dict(vaccinated=1000))
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_three_generated_shards_match_oracle():
    # every rank asks the library for its own shard (esim_synth_create_shard); the oracle runs the whole world
    cfg = dict(backend="gloo", cuts="generated", spec=dict(n_citizens=15000, n_areas=48, citizens_per_school=2500, n_seeds=16),
               params=dict(AGGRESSIVE, seed=21), steps=500, chunk=250)
    outs = launch(3, cfg)
    assert all("ok" in o for o in outs)


def test_shards_without_any_infected_citizen_match_oracle():
    # one seed in a world of three shards: two of them stay all-susceptible for a long time
    cfg = dict(backend="gloo", cuts="generated", spec=dict(n_citizens=15000, n_areas=48, citizens_per_school=2500, n_seeds=1),
               params=dict(exposure_chance=0.004, seed=31, max_steps=500), steps=480, chunk=240)
    outs = launch(3, cfg)
    assert all("ok" in o for o in outs)


def test_few_eligible_candidates_need_more_than_one_batch():
    # at the trigger (step 107) only 1225 of 12 000 citizens are still Susceptible, i.e. eligible: 600 vaccinations a step need
    # ~5900 candidates, more than one batch of 4096 -- the shards exchange the liveness of a window of 32 768 candidates
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=300),
               params=dict(exposure_chance=0.3, vaccination_rate=600, vaccination_threshold=0.4, lockdown_threshold=0.9, mask_pt_threshold=0.9,
                           mask_everywhere_threshold=0.95, seed=3, max_steps=700), steps=240, chunk=120, expect=dict(vaccinated=1000))
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


@pytest.mark.parametrize("world", (2, 3))
def test_sharded_plans_are_repaired_after_bus_exposures(world):
    # many bus exposures under a programme that vaccinates a tenth of the world per step: citizens the plan vaccinates later are
    # exposed on a bus first, in every chunk.  ESIM_VAX_REPAIR=2: the shards agree on the step to walk the plan again from (buffer
    # L), exchange the candidates' liveness as it truly stood and walk the same steps again -- records and states as the oracle's
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=9000, n_areas=30, citizens_per_school=2500, n_seeds=40),
               params=dict(exposure_chance=0.02, vaccination_rate=60, vaccination_threshold=0.01, lockdown_threshold=0.9, mask_pt_threshold=0.9,
                           mask_everywhere_threshold=0.95, bus_capacity=60, seed=5, max_steps=700), steps=480, chunk=160,
               expect=dict(vaccinated=1000), expect_repairs=2, env_by_rank={str(r): {"ESIM_VAX_REPAIR": "2"} for r in range(world)})
    outs = launch(world, cfg)
    assert all("ok" in o for o in outs)


def test_a_rank_whose_chunks_never_fit_takes_every_rank_to_coupled_steps():
    # ESIM_HASH_LOG2=4 on rank 1 only: its chunks can never take the one-pass form, rank 0's can.  The "cannot" word of buffer F
    # is summed, so both ranks skip the chunk together and back off to coupled steps -- same records as the oracle
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=240, chunk=120, env_by_rank={"1": {"ESIM_HASH_LOG2": 4}}, pipeline=3, expect_coupled=200)
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_commuter_segment_grows_with_the_need():
    # the all-gathered segment of Infected commuters starts at 2 records per rank here (ESIM_XS_CAP): every chunk that needs
    # more is a no-op on all ranks, the segment doubles (the need is gathered too, so all ranks agree) and the chunk runs again
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=360, chunk=120, expect=dict(vaccinated=1),
               env_by_rank={"0": {"ESIM_XS_CAP": 2}, "1": {"ESIM_XS_CAP": 2}})
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_one_ranks_device_error_makes_every_rank_return_it():
    # failure semantics of a sharded run (the reference bubbles a failed step up to main, run/src/main.rs:306-308): no rank is
    # left inside a collective when another one fails -- in the chunk form and in the coupled-step form
    for pipeline in (3, 0):
        cfg = dict(backend="gloo", mode="inject", bad_rank=1, pipeline=pipeline, steps=120, pg_timeout=60,
                   spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16), params=AGGRESSIVE)
        outs = launch(2, cfg, timeout=180)
        assert all("raised ERANGE" in o for o in outs), outs


def test_shards_that_do_not_belong_together_are_refused():
    cfg = dict(backend="gloo", mode="mismatch", pg_timeout=60,
               spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16), params=AGGRESSIVE)
    outs = launch(2, cfg, timeout=180)
    assert all("refused" in o for o in outs), outs


def test_syn3m5_two_shards_match_oracle():
    # BASELINE.json configs[3]: synthetic 3.5 M citizens, Output-Area sharded with even cuts (here 2 ranks on the test GPU):
    # commuters to schools across the cut, every step coupled
    cfg = dict(backend="gloo", cuts="even", preset="syn3m5", spec=dict(), params=dict(max_steps=400), steps=360, chunk=180)
    outs = launch(2, cfg, timeout=600)
    assert all("ok" in o for o in outs)


def test_rccl_communicator_of_one_rank():
    # the library's own RCCL transport on the one GPU there is: unique id, communicator, ncclAllReduce enqueued on the context's
    # stream between the kernels of every step (sums over one rank change nothing) -- same records as esim_run
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.dirname(HERE))
    from epidemicsimulator_amd import Population, Simulator, _lib
    pop = Population.synthetic("york", n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16)
    ep = _lib.default_params(**AGGRESSIVE)
    ref = Simulator(pop, ep)
    want = ref.run(300)
    ref.close()
    sim = Simulator(pop, ep)
    uid = (C.c_uint8 * 128)()
    _lib.check(sim.lib.esim_comm_unique_id(uid, 128))
    _lib.check(sim.lib.esim_comm_init_rccl(sim._ctx, uid, 128, 0, 1), sim._ctx)
    n_done = C.c_uint32(0)
    _lib.check(sim.lib.esim_run_sharded(sim._ctx, 300, C.byref(n_done)), sim._ctx)
    assert n_done.value == 300
    sim._steps = 300
    got = sim.records_so_far()
    n_coll = C.c_uint64(0)
    _lib.check(sim.lib.esim_comm_stats(sim._ctx, C.byref(n_coll)), sim._ctx)
    assert n_coll.value > 0
    for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated", "exposures_building", "exposures_bus", "vaccinated_now"):
        assert (got[f] == want[f]).all(), f
    sim.close()


def test_bench_two_ranks_on_one_gpu():
    # bench.py's N > 1 path end to end (strong scaling: ONE york world sharded over two ranks), the two ranks sharing the test
    # GPU and the library's exchange going through its callback transport into gloo -- the driver's N-GPU run uses the same loop
    # over the library's RCCL communicator
    port = free_port()
    env = dict(os.environ, ESIM_BENCH_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1300", "--warmup", "24",
           "--preset", "york", "--transport", "callback"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 1300 and d["scaling"] == "strong" and d["value"] > 0
    assert sum(d["config"]["citizens_per_gpu"]) == 197603 and d["config"]["shared_buildings"] > 0
    assert d["config"]["collectives"] > 0 and d["config"]["chunk_steps"] > 0
    assert d["golden_check"]["match"] and d["golden_check"]["records_compared"] >= 100 + 24      # first 100 records + every 50th up to 1300
    fr = d["final_record"]
    assert fr["time_step"] == 1300 and fr["vaccinated"] > 0          # the programme starts at step 1099
    assert fr["susceptible"] + fr["exposed"] + fr["infected"] + fr["recovered"] + fr["vaccinated"] == 197603
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])


def test_bench_one_gpu_line():
    # the one-GPU line on the york preset: the contract's keys, the golden check of the timed region, the full 5000-step run, a
    # CPU baseline sample whose records equal the GPU's
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--steps", "20", "--warmup", "5", "--preset", "york", "--cpu-seconds", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["steps"] == 20 and d["golden_check"]["records_compared"] == 20
    assert d["full_run"]["steps"] == 5000 and d["full_run"]["golden_check"]["records_compared"] == 200
    assert d["full_run"]["steps_with_vaccination_active"] > 3800 and d["full_run"]["sequential_steps"] < 50
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["shape"].startswith("reference-shaped") and d["cpu_baseline"]["records_match_gpu"]
    assert d["dtype"].startswith("u32 citizen word; u32 uniform")
