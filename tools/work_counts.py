"""What the chunk pass WORKS ON in a full run of a preset, counted exactly by the counting build (make -C epidemicsimulator_amd/csrc
count -> libesim_count.so; this script selects it): Infected log entries marked, hash keys, items, member words staged, (member,
four-step slot) pairs looked at and drawn, Philox blocks, Bernoulli draws, hits.  Writes profiles/<tag>_work_<preset>.json -- the
sparse pass's own unit counts, which bench.py and profiles/summarize_r03.py price its roofline in.
  python tools/work_counts.py tag [preset] [steps]"""
import ctypes as C
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ESIM_LIB"] = os.path.join(ROOT, "epidemicsimulator_amd", "libesim_count.so")
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib  # noqa: E402

NAMES = ["entries", "keys", "claims", "records", "direct_counter_atomics", "folded_records", "items", "members", "members_through_index",
         "pairs", "pairs_active", "philox_blocks", "draws", "units", "route_pairs", "riders_ranked", "bus_draws", "hits"]
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
preset = sys.argv[2] if len(sys.argv) > 2 else "uk64m"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
pop = Population.synthetic(preset)
sim = Simulator(pop, _lib.default_params(max_steps=5000))
fn = sim.lib.esim_work_counters
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32]
buf = (C.c_uint64 * len(NAMES))()
rec = sim.run(steps)
_lib.check(fn(sim._ctx, buf, len(NAMES)), sim._ctx)
w = {k: int(v) for k, v in zip(NAMES, buf)}
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_%s_5000.json" % preset)))
ok = all(int(rec[g["time_step"] - 1][f]) == g[f] for g in gold["records"] if g["time_step"] <= steps for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated"))
# Useful bytes of the sparse formulation: what it has to read and write given WHAT it visits (not how the cache lines fall).
#   per Infected entry marked: log id 4 + citizen word 4 + home / work / room / route ids 16
#   per key: hash key 8 (look) ; per claim: key 8 + item record 32 + slot index 4 ; per record left: 4 + the slot's counter 4
#   per member staged: citizen word 4 (+ 4 through an index list); per hit: the atomicMin 4 + exposure list 4 + log 4 + count 4
#   per rider ranked: id 4 + word 4
useful = (w["entries"] * 24 + w["keys"] * 8 + w["claims"] * 44 + w["records"] * 8 + w["direct_counter_atomics"] * 4 + w["folded_records"] * 4 +
          w["members"] * 4 + w["members_through_index"] * 4 + w["hits"] * 16 + w["riders_ranked"] * 8)
exposures = int(rec["exposures_building"].astype("int64").sum() + rec["exposures_bus"].astype("int64").sum())
out = {"tag": tag, "preset": preset, "steps": steps, "golden_match": bool(ok), "counts": w, "useful_bytes": useful,
       "infected_citizen_steps": int(rec["infected"].astype("int64").sum()), "exposures": exposures,
       "useful_bytes_model": "entry 24 B, key 8, claim 44, record 8, counter atomic 4, folded record 4, member word 4 (+4 via index), hit 16, rider 8"}
path = os.path.join(ROOT, "profiles", "%s_work_%s.json" % (tag, preset))
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
