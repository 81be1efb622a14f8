"""The reference-shaped multithreaded CPU path (oracle/esim_refshape.cpp: area-parallel array-of-structs citizens, per-area
hash maps, mutex-guarded lookup table -- the structure of sim/src/simulator.rs:87-103,167-260) must give the oracle's records,
whatever the thread count.  It is bench.py's cpu_baseline; CPU only."""
import numpy as np

import _oracle
from epidemicsimulator_amd import Population, _lib


def same(a, b):
    for f in a.dtype.names:
        if f == "pad":
            continue
        assert (a[f] == b[f]).all(), (f, int(np.argmax(a[f] != b[f])))


def test_records_equal_the_oracle_with_every_branch_taken():
    pop = Population.synthetic("york", n_citizens=9000, n_areas=30, citizens_per_school=3000, n_seeds=15, p_public_transport=0.4)
    ep = _lib.default_params(exposure_chance=0.004, vaccination_rate=40, vaccination_threshold=0.02, lockdown_threshold=0.03,
                             mask_pt_threshold=0.005, mask_everywhere_threshold=0.01, seed=5)
    prm = _oracle.params_from_esim(ep)
    want = _oracle.Oracle(pop, prm).run(700)
    assert want["vaccinated"][-1] > 0 and want["exposures_bus"].sum() > 0 and want["lockdown"].sum() > 0
    for threads in (1, 4):
        same(_oracle.ReferenceShaped(pop, prm, threads).run(700), want)


def test_york_first_days_equal_the_oracle():
    pop = Population.synthetic("york")
    prm = _oracle.params_from_esim(_lib.default_params())
    same(_oracle.ReferenceShaped(pop, prm, 8).run(120), _oracle.Oracle(pop, prm).run(120))
