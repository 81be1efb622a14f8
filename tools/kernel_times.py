"""Device time per kernel of the chunk pass over a full run of a preset (HIP events in front of every chunk kernel).
   python tools/kernel_times.py [preset]      (ESIM_PMAP / ESIM_PMAP_REBUILD / ESIM_DRAW_MULT ... select the variant)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "uk64m"
sim = Simulator(Population.synthetic(preset), _lib.default_params(max_steps=5000))
sim.run(96); sim.reset()
sim.enable_chunk_kernel_timing(True)
t0 = time.perf_counter(); rec = sim.run(5000); dt = time.perf_counter() - t0
kt = sim.chunk_kernel_timings()
tot = sum(v["ms"] for v in kt.values())
print("%s PMAP=%s REBUILD=%s: wall %.1f ms, chunk kernels %.2f ms | " % (preset, os.environ.get("ESIM_PMAP", "1"), os.environ.get("ESIM_PMAP_REBUILD", "4"), dt * 1e3, tot) +
      " ".join("%s %.2f (%d)" % (k, v["ms"], v["calls"]) for k, v in kt.items() if v["calls"]))
