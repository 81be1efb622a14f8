"""Per-chunk counters of the time-parallel pass on the benchmark workload (diagnostics)."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
from epidemicsimulator_amd import Population, Simulator, _lib
names = ["t", "chunk_ok", "chunk_parallel", "chunk_pairs", "n_items", "items_per_wave", "n_units", "n_route_pairs",
         "n_route_pairs_big", "n_newexp", "log_len", "n_susceptible", "lockdown", "mask", "at_work", "bus_dir"]
pop = Population.synthetic(sys.argv[1] if len(sys.argv) > 1 else "uk64m")
sim = Simulator(pop, _lib.default_params(max_steps=5000))
out = (C.c_uint32 * 16)()
targets = list(range(96, 96 * 13, 96)) + [1920, 2880, 3840, 4800, 4896, 4992]
for target in targets:
    n = target - sim._steps
    t0 = time.perf_counter(); sim.run(n); dt = time.perf_counter() - t0
    _lib.check(sim.lib.esim_debug_counters(sim._ctx, out), sim._ctx)
    d = dict(zip(names, out))
    print("after %4d steps (%.2f us/step): " % (target, dt / n * 1e6), {k: d[k] for k in names[1:11]})
