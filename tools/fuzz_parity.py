"""Extended parity fuzz (GPU): random populations and parameter sets beyond the ones the test suite pins, every execution
form against the oracle.  python tools/fuzz_parity.py [first_seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as T                                  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 30)
BIG = "--big" in sys.argv           # larger worlds, longer runs, odd block lengths
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    if BIG:
        pop = T.random_population(seed, n=int(rng.choice([4000, 9000, 20000])), n_areas=int(rng.choice([2, 9, 40])),
                                  n_buildings=int(rng.choice([60, 500, 3000])), n_schools=int(rng.choice([1, 4, 12])),
                                  rooms_per_school=int(rng.choice([1, 6, 20])))
    else:
        pop = T.random_population(seed, n=int(rng.choice([300, 700, 2500])), n_areas=int(rng.choice([1, 5, 12])),
                                  n_buildings=int(rng.choice([40, 90, 400])), n_schools=int(rng.choice([1, 3])),
                                  rooms_per_school=int(rng.choice([1, 4, 9])))
    params = dict(exposure_chance=float(rng.choice([0.0005, 0.002, 0.01, 0.05])), seed=int(rng.integers(1, 1 << 40)),
                  vaccination_rate=int(rng.choice([3, 25, 400, 5000])), vaccination_threshold=float(rng.choice([0.02, 0.08, 0.3, 2.0])),
                  lockdown_threshold=float(rng.choice([0.01, 0.05, 0.15, 0.9])), mask_pt_threshold=float(rng.choice([0.005, 0.02])),
                  mask_everywhere_threshold=float(rng.choice([0.04, 0.2])), bus_capacity=int(rng.choice([2, 3, 20, 64])),
                  exposed_time=int(rng.choice([1, 5, 30, 96])), infected_time=int(rng.choice([3, 17, 100, 336])),
                  start_hour=int(rng.choice([9, 6, 1])), end_hour=int(rng.choice([17, 20, 23])))
    steps = int(rng.choice([600, 1000, 1500] if BIG else [200, 500, 900]))
    T.run_both(pop, steps, check_state_every=int(rng.choice([17, 50, 97, 125, 300, 1000] if BIG else [50, 125, 300])),
               small_limits=("pmap", "vax", "wide", "tinymax", "tp", "pipe", None) if BIG else T.SMALL_LIMITS, **params)
    print("seed %d ok (%d citizens, %d steps, %s)" % (seed, pop.n_citizens, steps, {k: params[k] for k in ("exposure_chance", "exposed_time", "infected_time", "bus_capacity")}), flush=True)
print("all %d ok in %.0f s" % (count, time.time() - t0))
