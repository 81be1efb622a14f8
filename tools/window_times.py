"""Device time per chunk kernel inside windows of the run: python tools/window_times.py preset first_step:last_step ...
(HIP events in front of every chunk kernel, esim_chunk_kernel_timings; the steps before a window run untimed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib
preset = sys.argv[1]
sim = Simulator(Population.synthetic(preset), _lib.default_params(max_steps=5000))
for w in sys.argv[2:]:
    a, b = (int(x) for x in w.split(":"))
    if a - 1 > sim._steps: sim.run(a - 1 - sim._steps)
    sim.enable_chunk_kernel_timing(True); sim.chunk_kernel_timings()
    rec = sim.run(b - sim._steps)
    kt = sim.chunk_kernel_timings(); sim.enable_chunk_kernel_timing(False)
    print("steps %d..%d (Infected %d, lockdown %d, programme %d): %.2f ms | " % (a, b, int(rec["infected"][-1]), int(rec["lockdown"][-1]), int(rec["vaccination_active"][-1]), sum(v["ms"] for v in kt.values()))
          + " ".join("%s %.3f (%d)" % (k, v["ms"], v["calls"]) for k, v in kt.items() if v["calls"]))
