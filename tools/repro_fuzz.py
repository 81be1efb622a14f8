"""One seed of tools/fuzz_parity.py, every execution form on its own, with the first mismatch of each (diagnostics)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as T
import _oracle
from epidemicsimulator_amd import Simulator, _lib

seed = int(sys.argv[1])
block = int(sys.argv[2]) if len(sys.argv) > 2 else None
rng = np.random.default_rng(7000 + seed)
pop = T.random_population(seed, n=int(rng.choice([300, 700, 2500])), n_areas=int(rng.choice([1, 5, 12])),
                          n_buildings=int(rng.choice([40, 90, 400])), n_schools=int(rng.choice([1, 3])),
                          rooms_per_school=int(rng.choice([1, 4, 9])))
params = dict(exposure_chance=float(rng.choice([0.0005, 0.002, 0.01, 0.05])), seed=int(rng.integers(1, 1 << 40)),
              vaccination_rate=int(rng.choice([3, 25, 400, 5000])), vaccination_threshold=float(rng.choice([0.02, 0.08, 0.3, 2.0])),
              lockdown_threshold=float(rng.choice([0.01, 0.05, 0.15, 0.9])), mask_pt_threshold=float(rng.choice([0.005, 0.02])),
              mask_everywhere_threshold=float(rng.choice([0.04, 0.2])), bus_capacity=int(rng.choice([2, 3, 20, 64])),
              exposed_time=int(rng.choice([1, 5, 30, 96])), infected_time=int(rng.choice([3, 17, 100, 336])),
              start_hour=int(rng.choice([9, 6, 1])), end_hour=int(rng.choice([17, 20, 23])))
steps = int(rng.choice([200, 500, 900])); cse = int(rng.choice([50, 125, 300]))
if block: cse = block
print(params, steps, cse)
ep = _lib.default_params(**params)
orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
want = orc.run(steps)
for name, level, lim in (("vax", 3, None), ("tp", 2, None), ("pipe", 1, None), ("seq", 0, None), ("seq-multi", 0, 0), ("seq-small", 0, 1 << 30)):
    sim = Simulator(pop, ep)
    sim.set_pipeline(level)
    if lim is not None: sim.set_small_step_limit(lim)
    got = []
    done = 0
    while done < steps:
        n = min(cse, steps - done); got.append(sim.run(n)); done += n
    got = np.concatenate(got)
    bad = [f for f in T.FIELDS if not (got[f] == want[f]).all()]
    if bad:
        f = bad[0]; i = int(np.argmax(got[f] != want[f]))
        print(name, "MISMATCH fields", bad, "first", f, "at step", i + 1, "gpu", got[f][i], "oracle", want[f][i],
              "| infected there", want["infected"][i], "lockdown", want["lockdown"][max(0, i - 1)], "bus exp gpu/orc", got["exposures_bus"][i], want["exposures_bus"][i],
              "bld exp gpu/orc", got["exposures_building"][i], want["exposures_building"][i])
    else:
        print(name, "ok")
    sim.close()

# ---- the one-pass form again, chunk by chunk, down to the first citizen that differs
if "--trace" in sys.argv:
    for attempt in range(6):
        sim = Simulator(pop, ep); sim.set_pipeline(3)
        orc2 = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
        done = 0
        prev_o = orc2.state()
        found = False
        while done < steps and not found:
            n = min(cse, steps - done)
            g = sim.run(n); o = orc2.run(n)
            done += n
            sg, so = sim.download_state(), orc2.state()
            bad = np.nonzero((sg["status"] != so["status"]) | (sg["timer"] != so["timer"]))[0]
            if len(bad):
                found = True
                print("attempt", attempt, "block ending at step", done, "citizens that differ:", bad[:10].tolist())
                for c in bad[:4]:
                    print("  citizen", int(c), "gpu", int(sg["status"][c]), int(sg["timer"][c]), "oracle", int(so["status"][c]), int(so["timer"][c]),
                          "home", int(pop.home_building[c]), "work", int(pop.work_building[c]), "room", int(pop.room[c]), "flags", int(pop.flags[c]),
                          "home type", int(pop.building_type[pop.home_building[c]]), "work type", int(pop.building_type[pop.work_building[c]]),
                          "areas", int(pop.building_area[pop.home_building[c]]), int(pop.building_area[pop.work_building[c]]))
                    hb = pop.home_building[c]
                    print("    residents of its home:", int((pop.home_building == hb).sum()), "workers there:", int(((pop.work_building == hb) & (pop.home_building != hb)).sum()),
                          "| infected residents now (oracle):", int(((pop.home_building == hb) & (so["status"] == 2)).sum()))
                i = int(np.argmax(g["susceptible"] != o["susceptible"])) if (g["susceptible"] != o["susceptible"]).any() else -1
                print("  first record mismatch in block at offset", i, "lockdown", o["lockdown"][:].tolist()[:n], "infected", int(o["infected"][0]), "->", int(o["infected"][-1]))
        print("attempt", attempt, "clean" if not found else "")
        sim.close()
