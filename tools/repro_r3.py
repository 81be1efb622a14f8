import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import _oracle
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic("york", n_citizens=9000, n_areas=2, citizens_per_school=9000, n_seeds=40, p_public_transport=0.5)
ep = _lib.default_params(exposure_chance=0.02, vaccination_threshold=0.9, lockdown_threshold=0.95, mask_pt_threshold=0.2, mask_everywhere_threshold=0.4, seed=5)
sim = Simulator(pop, ep); sim.set_pipeline(3)
orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
for b in range(7):
    r = sim.run(100); o = orc.run(100)
    print("block", b, len(r), "chunks", sim.chunk_timing(), "pipe", sim.pipeline_timing(), flush=True)
    if len(r) != 100: break
    for f in ("susceptible","exposed","infected","exposures_building","exposures_bus"):
        assert (r[f] == o[f]).all(), (b, f)
