"""Seed ensemble of the CPU oracle on the `york` preset with the parameters of the reference's recorded v1.7.1 run, against
the facts tests/golden/reference_york_v171_envelope.json holds for that run (calibration aid for popgen.cpp's one free
input, the dwellings OSM tags per Output Area).   python tools/envelope.py [n_seeds] [steps] [spec_field=value ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _envelope

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    over = {}
    for a in sys.argv[3:]:
        k, v = a.split("=")
        over[k] = float(v) if "." in v else int(v)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_york_v171_envelope.json")))
    keys = ("peak_infected", "peak_exposed", "first_vaccinated_record", "recovered_decreases", "exposures_total", "areas_with_exposures",
            "exposures_share_top25_areas")
    print("reference", ref["first_step_over"], {k: ref.get(k) for k in keys})
    for f in _envelope.ensemble(n, steps, **over):
        print(f["k"], f["first_step_over"], {k: (round(f[k], 3) if isinstance(f[k], float) else f[k]) for k in keys}, "lock@h", f["lockdown_first_hour"])
