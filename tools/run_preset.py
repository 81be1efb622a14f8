"""Runs a preset for 5000 steps on the GPU (one esim_run call), checks the records against the preset's golden, prints wall
time per step and how the steps were executed.  python tools/run_preset.py preset [pipeline level]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "york"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pop = Population.synthetic(preset)
sim = Simulator(pop, _lib.default_params(max_steps=5000))
sim.set_pipeline(level)
sim.run(96); sim.reset()
sim.enable_kernel_timing(16)
t0 = time.perf_counter(); rec = sim.run(5000); dt = time.perf_counter() - t0
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_%s_5000.json" % preset)))
ok = all(int(rec[w["time_step"] - 1][f]) == w[f] for w in gold["records"] for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated"))
print(preset, "level", level, "us/step %.3f" % (dt / 5000 * 1e6), "golden", ok, "chunks", sim.chunk_timing(), "vax", sim.vax_chunk_stats(),
      "pipe", sim.pipeline_timing(), "small", sim.small_kernel_timing(), "multi", sim.kernel_timings() if hasattr(sim, "kernel_timings") else None)
first_v = int(np.argmax(rec["vaccination_active"] > 0)) + 1 if (rec["vaccination_active"] > 0).any() else None
print("  vaccination from step", first_v, "peak infected", int(rec["infected"].max()), "bus exposures", int(rec["exposures_bus"].sum()), "final", {k: int(rec[k][-1]) for k in ("susceptible", "infected", "vaccinated")})
