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
    if exposure_area is not None:
        per_area = np.bincount(np.asarray(exposure_area, dtype=np.int64))
        per_area = np.sort(per_area[per_area > 0])[::-1]
        out["exposures_total"] = int(per_area.sum())
        out["areas_with_exposures"] = int(per_area.size)
        out["exposures_share_top25_areas"] = float(per_area[:25].sum() / max(1, per_area.sum()))
    return out


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
