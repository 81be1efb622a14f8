"""Generates the fixtures under tests/golden/ (run here, in the build container).

1. oracle_small_world.json   -- output of the CPU oracle (oracle/esim_oracle.c) on a small seeded world;
                                freezes the oracle so later edits cannot drift silently.
2. reference_york_v171_envelope.json -- summary facts read from the reference's own recorded output
   /root/reference/statistics_results/v1.7.1/1946157112TYPE299/global_stats.json (a data file of the
   reference; only derived numbers are kept).  The reference itself cannot be built or run here
   (Rust toolchain and input data absent), so this is the only reference output available.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

import _oracle  # noqa: E402
from epidemicsimulator_amd import Population  # noqa: E402


def small_world():
    spec = dict(n_citizens=6000, n_areas=24, citizens_per_school=1500, n_seeds=12)
    params = dict(exposure_chance=0.004, vaccination_rate=40, vaccination_threshold=0.02,
                  lockdown_threshold=0.03, mask_pt_threshold=0.005, mask_everywhere_threshold=0.01, seed=77)
    steps = 1200
    pop = Population.synthetic("york", **spec)
    rec = _oracle.Oracle(pop, _oracle.default_params(**params)).run(steps)
    keep = ["susceptible", "exposed", "infected", "recovered", "vaccinated", "exposures_building",
            "exposures_bus", "lockdown", "mask_status", "vaccinated_now", "eligible_count", "n_riders"]
    out = {"spec": spec, "params": params, "steps": steps, "records": {k: rec[k].tolist() for k in keep}}
    with open(os.path.join(HERE, "oracle_small_world.json"), "w") as f:
        json.dump(out, f)
    print("small world: vaccinated", int(rec["vaccinated"][-1]), "bus exposures", int(rec["exposures_bus"].sum()),
          "lockdown steps", int(rec["lockdown"].sum()))


def reference_envelope():
    path = "/root/reference/statistics_results/v1.7.1/1946157112TYPE299/global_stats.json"
    if not os.path.exists(path):
        print("reference not mounted; keeping the committed envelope")
        return
    stats = json.load(open(path))
    stats = [s for s in stats if s["susceptible"] + s["exposed"] + s["infected"] + s["recovered"] + s["vaccinated"] > 0]
    n = stats[0]["susceptible"] + stats[0]["exposed"] + stats[0]["infected"] + stats[0]["recovered"] + stats[0]["vaccinated"]
    x = np.array([s["infected"] / n for s in stats])
    first = {str(th): int(stats[int(np.argmax(x > th))]["time_step"]) for th in (0.001, 0.0022, 0.0034, 0.005)}
    rec = np.array([s["recovered"] for s in stats])
    vac = np.array([s["vaccinated"] for s in stats])
    out = {"source": "statistics_results/v1.7.1/1946157112TYPE299/global_stats.json", "n_citizens": int(n),
           "n_records": len(stats), "first_step_over": first,
           "peak_infected": int(max(s["infected"] for s in stats)), "peak_exposed": int(max(s["exposed"] for s in stats)),
           "first_vaccinated_record": int(stats[int(np.argmax(vac > 0))]["time_step"]),
           "recovered_decreases": bool((np.diff(rec) < 0).any()),
           "seed_infected_first_record": int(stats[0]["infected"])}
    # the same run's exposures.json: exposures per Output Area (statistics.rs:119-136 writes one series per area that had any)
    expo = json.load(open(os.path.join(os.path.dirname(path), "exposures.json")))["OutputArea"]
    per_area = np.sort(np.array([sum(v) for v in expo.values()]))[::-1]
    out["exposures_source"] = "statistics_results/v1.7.1/1946157112TYPE299/exposures.json"
    out["exposures_total"] = int(per_area.sum())
    out["areas_with_exposures"] = int(per_area.size)
    out["exposures_share_top25_areas"] = float(per_area[:25].sum() / per_area.sum())
    out["final_record"] = {k: int(stats[-1][k]) for k in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")}
    # Q10 (simulator.rs:481-553): the record of the step whose end created citizens_eligible_for_vaccine -- its Susceptible count
    # IS the size of the set -- and the Vaccinated census every 250 steps from there on: with `rate` ids chosen per step and never
    # removed, E[V after k batches] = E * (1 - (1 - rate / E)^k), which the test checks on the reference's own curve
    trig = int(np.argmax(vac > 0)) - 1                     # the last record without a Vaccinated citizen: the trigger step
    out["vaccination_trigger_record"] = {k: int(stats[trig][k]) for k in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")}
    out["vaccination_rate_observed"] = int(vac[trig + 1])
    out["vaccinated_series"] = {str(int(stats[i]["time_step"])): int(vac[i]) for i in range(trig + 1, len(stats)) if (i - trig) % 250 == 0 or i == len(stats) - 1}
    with open(os.path.join(HERE, "reference_york_v171_envelope.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    small_world()
    reference_envelope()
