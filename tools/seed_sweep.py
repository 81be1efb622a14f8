"""A preset under Philox seeds other than the default, GPU (default form: time-parallel chunks) against the CPU oracle run here:
records of every step and the full per-citizen state at the end of each block.  Aggressive parameters bring the interventions
forward so that a short run covers them.   python tools/seed_sweep.py preset steps seed [seed ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle
import test_parity_gpu as T
from epidemicsimulator_amd import Population, Simulator, _lib
preset, steps, seeds = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
pop = Population.synthetic(preset)
for seed in seeds:
    t0 = time.time()
    ep = _lib.default_params(max_steps=steps, seed=seed, exposure_chance=0.0011, vaccination_threshold=0.002, lockdown_threshold=0.004,
                             mask_pt_threshold=0.0005, mask_everywhere_threshold=0.001)
    sim = Simulator(pop, ep)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    orc.set_threads(min(8, os.cpu_count() or 1))
    done = 0
    while done < steps:
        n = min(500, steps - done)
        a, b = sim.run(n), orc.run(n)
        T.assert_same_records(a, b)
        T.assert_same_state(sim, orc)
        done += n
        print("  %s seed %d: %d steps equal so far, infected %d, %.0f s" % (preset, seed, done, int(a["infected"][-1]), time.time() - t0), flush=True)
    print("%s seed %d: %d steps equal (records + state every 500); peak infected %d, vaccinated %d, lockdown steps %d, %s; %.0f s"
          % (preset, seed, steps, int(a["infected"].max()), int(a["vaccinated"][-1]), int((a["lockdown"] > 0).sum()), sim.vax_chunk_stats(), time.time() - t0), flush=True)
    sim.close()
