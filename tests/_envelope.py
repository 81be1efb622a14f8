"""The facts of the reference's recorded York run (tests/golden/reference_york_v171_envelope.json) computed from a run of the
CPU oracle, and a seed ensemble of such runs.  Test infrastructure (uses oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

THRESHOLDS = (0.001, 0.0022, 0.0034, 0.005)


def facts(rec, n, exposure_area=None):
    """rec: records with time_step, susceptible, exposed, infected, recovered, vaccinated (arrays); exposure_area: the
    Output Area credited with each exposure (one entry per exposure), or None."""
    inf = np.asarray(rec["infected"], dtype=np.int64)
    step = np.asarray(rec["time_step"], dtype=np.int64)
    out = {"first_step_over": {}}
    for th in THRESHOLDS:
        over = np.nonzero(inf > th * n)[0]
        out["first_step_over"][str(th)] = int(step[over[0]]) if over.size else None
    out["peak_infected"] = int(inf.max())
    out["peak_exposed"] = int(np.asarray(rec["exposed"]).max())
    v = np.nonzero(np.asarray(rec["vaccinated"]) > 0)[0]
    out["first_vaccinated_record"] = int(step[v[0]]) if v.size else None
    out["recovered_decreases"] = bool((np.diff(np.asarray(rec["recovered"], dtype=np.int64)) < 0).any())
    out["seed_infected_first_record"] = int(inf[0])
    # the vaccination curve (Q10): the record of the trigger step -- the last one without a Vaccinated citizen; its Susceptible
    # count is the size of citizens_eligible_for_vaccine, simulator.rs:487-513 -- and the Vaccinated census every 250 steps after
    vac = np.asarray(rec["vaccinated"], dtype=np.int64)
    if v.size and v[0] > 0:
        trig = int(v[0]) - 1
        out["vaccination_trigger_record"] = {"time_step": int(step[trig]), "susceptible": int(np.asarray(rec["susceptible"])[trig])}
        out["vaccination_rate_observed"] = int(vac[trig + 1])
        out["vaccinated_series"] = {str(int(step[i])): int(vac[i]) for i in range(trig + 1, len(step)) if (i - trig) % 250 == 0 or i == len(step) - 1}
        out["final_record"] = {k: int(np.asarray(rec[k])[-1]) for k in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated")}
    if exposure_area is not None:
        per_area = np.bincount(np.asarray(exposure_area, dtype=np.int64))
        per_area = np.sort(per_area[per_area > 0])[::-1]
        out["exposures_total"] = int(per_area.sum())
        out["areas_with_exposures"] = int(per_area.size)
        out["exposures_share_top25_areas"] = float(per_area[:25].sum() / max(1, per_area.sum()))
    return out


def q10_identity(f, rate, n_sigma=3.0):
    """The vaccination curve of a run against what simulator.rs:524-553 implies (Q10): every step `rate` DISTINCT members of
    citizens_eligible_for_vaccine are set Vaccinated, whatever they are, and none is ever removed from the set (only a handful
    leave it, through bus exposures, :447-449) -- so after k batches the Vaccinated census is the number of distinct members
    chosen so far, E * (1 - (1 - rate / E)^k) in expectation with a standard deviation below sqrt(E q (1 - q)), q = (1 - rate /
    E)^k.  A "fixed" vaccination that removed the chosen would give min(rate * k, E): tens of sigmas away within 250 steps.
    Returns the largest deviation in sigmas; raises when a sample is further than n_sigma."""
    import math
    E, t0 = f["vaccination_trigger_record"]["susceptible"], f["vaccination_trigger_record"]["time_step"]
    assert f["vaccination_rate_observed"] == min(rate, E), (f["vaccination_rate_observed"], rate)      # first batch: all new
    worst = 0.0
    for s, got in f["vaccinated_series"].items():
        k = int(s) - t0                                       # batches of steps t0 .. s-1 show in record s (Q13)
        q = (1.0 - rate / E) ** k
        want, sd = E * (1.0 - q), math.sqrt(max(E * q * (1.0 - q), 1.0))
        z = (got - want) / sd
        assert abs(z) <= n_sigma, "Vaccinated census %d at step %s: expected %.0f +- %.0f without removal (%.1f sigma); with removal it would be %d" % (
            got, s, want, sd, z, min(rate * k, E))
        worst = max(worst, abs(z))
    fr = f["final_record"]
    # bookkeeping at the end: nobody outside the set is ever vaccinated, and those of it never chosen are Susceptible or were
    # exposed after the trigger
    assert fr["vaccinated"] <= E and fr["susceptible"] <= E - fr["vaccinated"]
    return worst


def york_run(k, steps=5000, vaccination_rate=85, **spec_overrides):
    """Ensemble member k: the `york` preset generated with population seed k and run under Philox seed k, with the
    vaccination rate of the reference's v1.7.1 run (its records show 85 per step; the current source has 85 * 18)."""
    import _oracle
    from epidemicsimulator_amd import Population, _lib
    pop = Population.synthetic("york", seed=0x5EED2011 + 7919 * k, **spec_overrides)
    ep = _lib.default_params(max_steps=steps, vaccination_rate=vaccination_rate, seed=0x5EED2011 + 104729 * k)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    rec = orc.run(steps)
    step, area = orc.exposures()
    f = facts(rec, pop.n_citizens, area[(step > 0) & (area != 0xFFFFFFFF)])
    f["k"] = k
    lock = np.nonzero(rec["lockdown"])[0]
    f["lockdown_first_hour"] = int((rec["time_step"][lock[0]] + 1) % 24) if lock.size else None
    return f


def _one(args):
    return york_run(args[0], args[1], **args[2])


def ensemble(n, steps=5000, workers=None, **spec_overrides):
    from concurrent.futures import ProcessPoolExecutor
    workers = workers or min(n, os.cpu_count() or 1)
    with ProcessPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(_one, [(k, steps, spec_overrides) for k in range(n)]))
