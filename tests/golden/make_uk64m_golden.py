"""Golden records of the benchmark workload at full size: the CPU oracle on the 64 M-citizen `uk64m` preset for 5000 steps
(about 20 minutes and 6 GB on one core).  Writes tests/golden/oracle_uk64m_5000.json: every 50th record and the last.
  python tests/golden/make_uk64m_golden.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle
from epidemicsimulator_amd import Population, _lib

t0 = time.time()
pop = Population.synthetic("uk64m")
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
json.dump({"preset": "uk64m", "seed": int(ep.seed), "steps": 5000, "every": 50, "records": rows},
          open(os.path.join(ROOT, "tests", "golden", "oracle_uk64m_5000.json"), "w"), indent=0)
print("done %.0f s" % (time.time() - t0))
