"""A prevalence far beyond the one-pass chunk form's capacity (2 M citizens, up to 39 % Infected at once): the run falls back to
sequential steps and must still equal the oracle, records and state (diagnostics; the oracle takes ~30 s)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import _oracle
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic("york", n_citizens=2000000, n_areas=6400, citizens_per_school=20000, n_seeds=200)
ep = _lib.default_params(exposure_chance=0.004, lockdown_threshold=0.9, vaccination_threshold=0.25, seed=31, max_steps=1500)
sim = Simulator(pop, ep)
sim.enable_kernel_timing(8)
t0 = time.time(); g = sim.run(1200); dt = time.time() - t0
print("gpu 1200 steps %.3fs -> %.3g citizen-steps/s" % (dt, pop.n_citizens * 1200 / dt), sim.kernel_timings(), sim.small_kernel_timing())
i = int(g['infected'].argmax()); print("peak infected", g['infected'][i], "at", g['time_step'][i], "final", g[['susceptible','exposed','infected','recovered','vaccinated']][-1])
orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
t0 = time.time(); o = orc.run(1200); print("oracle %.1fs" % (time.time() - t0))
bad = [f for f in _lib.RECORD_FIELDS if f != 'reserved' and not (g[f] == o[f]).all()]
print("mismatch fields:", bad)
st, ost = sim.download_state(), orc.state()
print("state equal:", all((st[k] == ost[k]).all() for k in st))
# per-phase look at the heavy region
seg = g[600:1200]
print("mean infected in steps 600-1200:", seg['infected'].mean())
