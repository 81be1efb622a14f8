"""Per-wavefront timers of k_chunk_units (diagnostics build with -DESIM_WAVE_PROFILE -DESIM_PROFILE_UNITS)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ.setdefault("ESIM_LIB", os.path.abspath("epidemicsimulator_amd/libesim_profu.so"))
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic(sys.argv[1] if len(sys.argv) > 1 else "uk64m")
sim = Simulator(pop, _lib.default_params(max_steps=5000))
W = 4096
buf = np.zeros(W * 16, np.uint32)
khz = C.c_int(0)
sim.lib.esim_prof_read.restype = C.c_int
sim.lib.esim_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int)]
targets = [int(x) for x in sys.argv[2:]] or [96, 192, 288, 384, 960, 1920, 3840, 4800]
for target in targets:
    sim.run(target - sim._steps)
    _lib.check(sim.lib.esim_prof_read(sim._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size, C.byref(khz)), sim._ctx)
    r = buf.reshape(W, 16).astype(np.int64)
    us = lambda t: np.asarray(t, float) / (khz.value / 1000.0)
    live = r[:, 2] > 0
    t0 = r[live, 0].min() if live.any() else 0
    b = r[:, 4] > 0
    print("t=%4d units: waves that ran %d, starts within %.1f us, span %.1f us | prologue med %.1f max %.1f | loop med %.1f max %.1f | waves with units %d, "
          "most units %d, longest unit %.1f us, most 64-pair rounds in a wave %d"
          % (target, live.sum(), us(r[live, 0].max() - t0), us(r[live, 2].max() - t0), np.median(us(r[live, 1] - r[live, 0])), us(r[live, 1] - r[live, 0]).max(),
             np.median(us(r[live, 2] - r[live, 1])), us(r[live, 2] - r[live, 1]).max(), b.sum(), r[:, 4].max(), us(r[:, 5].max()), r[:, 7].max()))
    print("        units in all %d, 64-pair rounds in all %d (%.2f per unit) | us per unit %.2f, us per round %.2f"
          % (r[:, 4].sum(), r[:, 7].sum(), r[:, 7].sum() / max(1, r[:, 4].sum()), us(r[live, 2] - r[live, 1]).sum() / max(1, r[:, 4].sum()),
             us(r[live, 2] - r[live, 1]).sum() / max(1, r[:, 7].sum())))
    buf[:] = 0
