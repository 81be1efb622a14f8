"""The Output-Area sharded path (split-phase steps + SUM all-reduce of the exchange buffers) against the
whole-population oracle.  Ranks share the one GPU of the test box and reduce over gloo; on the 8-GPU
node bench.py runs the same code over nccl (RCCL)."""
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


def test_two_shards_coupled_match_oracle():
    # cuts through school catchments => shared buildings/rooms => coupled steps with two all-reduces
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=360, chunk=120)
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_three_uneven_shards_coupled_match_oracle():
    cfg = dict(backend="gloo", cuts="even", spec=dict(n_citizens=9000, n_areas=13, citizens_per_school=3000, n_seeds=16),
               params=dict(AGGRESSIVE, seed=9), steps=240, chunk=120)
    outs = launch(3, cfg)
    assert all("ok" in o for o in outs)


def test_two_shards_decoupled_then_coupled_match_oracle():
    # commuter-free cuts: decoupled 96-step batches (one all-reduce each) until the vaccination trigger,
    # coupled steps afterwards; records of the decoupled part are summed over the ranks
    cfg = dict(backend="gloo", cuts="clean", expect_both_modes=True,
               spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=600, chunk=150)
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_two_shards_decoupled_chunk_by_chunk_match_oracle():
    # the same without bursts: one host wait per chunk (esim_run_free)
    cfg = dict(backend="gloo", cuts="clean", expect_both_modes=True, burst_max=0,
               spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=AGGRESSIVE, steps=600, chunk=150)
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_two_shards_one_without_room_for_one_pass_chunks_match_oracle():
    # rank 1's hash map is too small for any one-pass chunk: buffer F's last word tells rank 0, speculative chunks are
    # no-ops on both, and both fall back to the per-step form of the chunk in lockstep
    cfg = dict(backend="gloo", cuts="clean", expect_both_modes=True, tiny_hash_ranks=[1],
               spec=dict(n_citizens=12000, n_areas=40, citizens_per_school=2500, n_seeds=16),
               params=dict(AGGRESSIVE, seed=5), steps=450, chunk=150)
    outs = launch(2, cfg)
    assert all("ok" in o for o in outs)


def test_three_generated_shards_match_oracle():
    # every rank generates only its own shard (esim_synth_create_shard); the oracle runs the whole world
    cfg = dict(backend="gloo", cuts="generated", spec=dict(n_citizens=15000, n_areas=48, citizens_per_school=2500, n_seeds=16),
               params=dict(AGGRESSIVE, seed=21), steps=500, chunk=250)
    outs = launch(3, cfg)
    assert all("ok" in o for o in outs)


def test_shards_without_any_infected_citizen_match_oracle():
    # one seed in a world of three shards: two of them stay all-susceptible for the whole run (the normal case of the
    # weak-scaling benchmark), their chunk passes have nothing to mark or draw
    cfg = dict(backend="gloo", cuts="generated", spec=dict(n_citizens=15000, n_areas=48, citizens_per_school=2500, n_seeds=1),
               params=dict(exposure_chance=0.004, seed=31, max_steps=500), steps=480, chunk=240)
    outs = launch(3, cfg)
    assert all("ok" in o for o in outs)


def test_syn3m5_two_generated_shards_match_oracle():
    # BASELINE.json configs[3]: synthetic 3.5 M citizens, Output-Area sharded (here 2 ranks on the test GPU)
    cfg = dict(backend="gloo", cuts="generated", spec=dict(n_citizens=3457142, n_areas=15669, citizens_per_school=20600, n_seeds=10),
               params=dict(max_steps=800), steps=720, chunk=360)
    outs = launch(2, cfg, timeout=500)
    assert all("ok" in o for o in outs)


def test_bench_two_ranks_on_one_gpu():
    # bench.py's N > 1 path end to end (weak scaling: every rank generates its own shard of a world twice the preset),
    # two ranks sharing the test GPU and reducing over gloo -- the driver's N-GPU run uses the same code over nccl
    port = free_port()
    env = dict(os.environ, ESIM_BENCH_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "600", "--warmup", "24",
           "--preset", "york", "--backend", "gloo", "--cpu-steps", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 600 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["decoupled_steps"] == 600 and d["config"]["coupled_steps"] == 0
    fr = d["final_record"]
    assert fr["time_step"] == 600
    assert fr["susceptible"] + fr["exposed"] + fr["infected"] + fr["recovered"] + fr["vaccinated"] == 2 * 197603
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
