"""BASELINE.json's configurations at FULL size and length (5000 steps) against records the CPU oracle produced offline
(tests/golden/make_preset_golden.py: the oracle needs minutes to tens of minutes for these, the GPU milliseconds).  Every
50th record and the exposure totals of every block of 50 steps must match; one esim_run call, i.e. the speculative
bursts of time-parallel chunks exactly as bench.py drives them."""
import json
import os

import numpy as np
import pytest

from epidemicsimulator_amd import Population, Simulator, _lib

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("preset", ["york", "yh_census", "syn3m5", "uk64m"])
def test_full_length_run_matches_offline_oracle(preset):
    path = os.path.join(GOLDEN, "oracle_%s_5000.json" % preset)
    if not os.path.exists(path):
        pytest.skip("no golden for %s (tests/golden/make_preset_golden.py %s)" % (preset, preset))
    gold = json.load(open(path))
    ep = _lib.default_params(max_steps=gold["steps"])
    assert int(ep.seed) == gold["seed"]
    sim = Simulator(Population.synthetic(preset), ep)
    rec = sim.run(gold["steps"])
    every = gold["every"]
    for i, want in enumerate(gold["records"]):
        got = rec[(i + 1) * every - 1]
        for f in ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated", "lockdown", "mask_status"):
            assert int(got[f]) == want[f], (preset, f, want["time_step"], int(got[f]), want[f])
        block = rec[i * every:(i + 1) * every]
        assert int(block["exposures_building"].sum()) == want["exposures_building_block"], (preset, want["time_step"])
        assert int(block["exposures_bus"].sum()) == want["exposures_bus_block"], (preset, want["time_step"])
    sim.close()
