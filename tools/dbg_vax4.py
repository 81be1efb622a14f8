import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_parity_gpu as T
import _oracle
from epidemicsimulator_amd import Simulator, _lib
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pop = T.random_population(seed)
rng = np.random.default_rng(1000 + seed)
params = dict(exposure_chance=float(rng.choice([0.002, 0.01, 0.05])), seed=int(rng.integers(1, 1 << 40)),
              vaccination_rate=int(rng.choice([3, 25, 400])), vaccination_threshold=float(rng.choice([0.02, 0.08, 0.3])),
              lockdown_threshold=float(rng.choice([0.05, 0.15, 0.9])), mask_pt_threshold=0.02,
              mask_everywhere_threshold=float(rng.choice([0.04, 0.2])), bus_capacity=int(rng.choice([3, 20])),
              exposed_time=int(rng.choice([5, 96])), infected_time=int(rng.choice([17, 336])),
              start_hour=int(rng.choice([9, 6])), end_hour=int(rng.choice([17, 20])))
print(params, pop.n_citizens)
ep = _lib.default_params(**params)
orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
sim = Simulator(pop, ep); sim.set_pipeline(3)
done = 0
while done < 500:
    g, o = sim.run(125), orc.run(125); done += 125
    bad = [f for f in T.FIELDS if not (g[f] == o[f]).all()]
    if bad:
        i = min(int(np.argmax(g[f] != o[f])) for f in bad)
        print("block ending", done, "fields", bad, "first at step", done - 125 + i + 1)
        for k in range(max(0, i - 3), min(125, i + 3)):
            print("  step", done - 125 + k + 1, {f: (int(g[f][k]), int(o[f][k])) for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated", "exposures_building", "exposures_bus", "vaccinated_now", "eligible_count", "lockdown", "n_riders")})
        sg, so = sim.download_state(), orc.state()
        d = np.nonzero((sg["status"] != so["status"]) | (sg["timer"] != so["timer"]) | (sg["eligible"] != so["eligible"]))[0]
        print("  differ:", [(int(c), int(sg["status"][c]), int(sg["timer"][c]), int(sg["eligible"][c]), int(so["status"][c]), int(so["timer"][c]), int(so["eligible"][c]), int(pop.flags[c])) for c in d[:8]])
        break
print("vax", sim.vax_chunk_stats())
