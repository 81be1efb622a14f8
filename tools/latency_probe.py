"""Where the wall time of a short esim_run goes (ESIM_TRACE_HOST=1: the library prints its host-side marks): `steps` steps from
time step 0, repeated; with and without the per-16th-step kernel timing the bench switches on.
   python tools/latency_probe.py [preset] [steps]"""
import os, sys, time
os.environ["ESIM_TRACE_HOST"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "uk64m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sim = Simulator(Population.synthetic(preset), _lib.default_params(max_steps=5000))
sim.run(5)
for timing in (0, 16, 0):
    print("kernel timing stride %d" % timing, file=sys.stderr)
    for rep in range(4):
        sim.reset()
        sim.enable_kernel_timing(timing)
        sim.synchronize()
        clock = []
        sim.run(steps, clock=clock)
        print("  python-side wall %.1f us" % (clock[0] * 1e6), file=sys.stderr)
