"""Two builds of libesim on the SAME epidemic states: build A runs the preset to the given steps and saves checkpoints (on the
box's /tmp), then A and B each restore every checkpoint and run 96 more steps; prints the device time of those steps.  Used to
compare designs whose RNG contracts differ (their own trajectories are different epidemics and cannot be compared).
  python tools/state_ab.py make|time preset step [step ...]      (ESIM_LIB selects the build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib
mode, preset, steps = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3:]]
pop = Population.synthetic(preset)
sim = Simulator(pop, _lib.default_params(max_steps=5000))
path = lambda s: "/tmp/esim_ckpt_%s_%d.bin" % (preset, s)
if mode == "make":
    for s in steps:
        sim.run(s - sim._steps)
        sim.save_checkpoint(path(s))
    print("saved", steps)
else:
    for s in steps:
        for rep in range(2):
            sim.load_checkpoint(path(s))
            sim.enable_kernel_timing(16)
            a = sim.chunk_timing()
            t0 = time.perf_counter(); rec = sim.run(96); dt = time.perf_counter() - t0
            b = sim.chunk_timing()
            print("%s from %d: 96 steps wall %.1f us, device %.1f us in %d chunks; infected %d -> %d, exposures %d" %
                  (os.path.basename(os.environ.get("ESIM_LIB", "libesim.so")), s, dt * 1e6, (b["chunk_ms"] - a["chunk_ms"]) * 1e3, b["chunks"] - a["chunks"],
                   int(rec["infected"][0]), int(rec["infected"][-1]), int(rec["exposures_building"].sum() + rec["exposures_bus"].sum())))
