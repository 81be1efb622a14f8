"""BASELINE.json's configurations at FULL size and length (5000 steps) against what the CPU oracle produced offline
(tests/golden/make_preset_golden.py: the oracle needs seconds to more than an hour for these, the GPU milliseconds to seconds).
Compared: every record of the first 100 steps, every 50th record after that with the exposure totals of its block of 50,
and the sha256 digest of the FULL per-citizen state (status, timer, position, bus, eligibility of every citizen) after
step 100 and after every 1000th step.  The run is driven as bench.py drives it: esim_run over the speculative bursts of time-parallel chunks."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _oracle  # noqa: E402  (state_digest only: the oracle itself is not run here)
from epidemicsimulator_amd import Population, Simulator, _lib  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CENSUS = ("time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated", "lockdown", "vaccination_active",
          "mask_status", "n_riders", "vaccinated_now", "eligible_count")


@pytest.mark.parametrize("preset", ["york", "yh_census", "syn3m5", "uk64m"])
def test_full_length_run_matches_offline_oracle(preset):
    path = os.path.join(GOLDEN, "oracle_%s_5000.json" % preset)
    if not os.path.exists(path):
        pytest.skip("no golden for %s (tests/golden/make_preset_golden.py %s)" % (preset, preset))
    gold = json.load(open(path))
    ep = _lib.default_params(max_steps=gold["steps"])
    assert int(ep.seed) == gold["seed"]
    sim = Simulator(Population.synthetic(preset), ep)
    rec = sim.run(100)
    for i, want in enumerate(gold["first_records"]):
        for f in CENSUS + ("exposures_building", "exposures_bus"):
            assert int(rec[i][f]) == want[f], (preset, f, want["time_step"], int(rec[i][f]), want[f])
    assert _oracle.state_digest(sim.download_state()) == gold["state_sha256"]["100"], "per-citizen state after step 100"
    # ... and after every step the golden holds a digest for (every 1000th), the run continuing from there
    done = 100
    for at in sorted(int(k) for k in gold["state_sha256"] if int(k) > 100):
        rec = np.concatenate([rec, sim.run(at - done)])
        done = at
        assert _oracle.state_digest(sim.download_state()) == gold["state_sha256"][str(at)], "per-citizen state after step %d" % at
    assert done == gold["steps"]
    every = gold["every"]
    for i, want in enumerate(gold["records"]):
        got = rec[(i + 1) * every - 1]
        for f in CENSUS:
            assert int(got[f]) == want[f], (preset, f, want["time_step"], int(got[f]), want[f])
        block = rec[i * every:(i + 1) * every]
        assert int(block["exposures_building"].sum()) == want["exposures_building_block"], (preset, want["time_step"])
        assert int(block["exposures_bus"].sum()) == want["exposures_bus_block"], (preset, want["time_step"])
    sim.close()
