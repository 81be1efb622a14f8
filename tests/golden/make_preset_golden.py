"""Golden records of the benchmark workloads at full size: the CPU oracle on a synthetic preset for 5000 steps (uk64m: about
20 minutes and 6 GB on one core).  Writes tests/golden/oracle_<preset>_5000.json: every 50th record and the last.
  python tests/golden/make_uk64m_golden.py [preset]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle
from epidemicsimulator_amd import Population, _lib

t0 = time.time()
preset = sys.argv[1] if len(sys.argv) > 1 else "uk64m"
pop = Population.synthetic(preset)
ep = _lib.default_params(max_steps=5000)
orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
print("population %.0f s" % (time.time() - t0), flush=True)
rows = []
done = 0
while done < 5000:
    r = orc.run(50)
    done += 50
    rows.append({k: int(r[k][-1]) for k in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated",
                                            "exposures_building", "exposures_bus", "lockdown", "mask_status")})
    rows[-1]["exposures_building_block"] = int(r["exposures_building"].sum())
    rows[-1]["exposures_bus_block"] = int(r["exposures_bus"].sum())
    if done % 500 == 0:
        print(done, rows[-1], "%.0f s" % (time.time() - t0), flush=True)
json.dump({"preset": preset, "seed": int(ep.seed), "steps": 5000, "every": 50, "records": rows},
          open(os.path.join(ROOT, "tests", "golden", "oracle_%s_5000.json" % preset), "w"), indent=0)
print("done %.0f s" % (time.time() - t0))
