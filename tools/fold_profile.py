import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ.setdefault("ESIM_LIB", os.path.abspath("epidemicsimulator_amd/libesim_prof.so"))
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic("uk64m")
sim = Simulator(pop, _lib.default_params(max_steps=5000))
W = 4096
buf = np.zeros(W * 16, np.uint32); khz = C.c_int(0)
sim.lib.esim_prof_read.restype = C.c_int
sim.lib.esim_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int)]
for target in (2880, 3360, 3840, 4800):
    sim.run(target - sim._steps)
    _lib.check(sim.lib.esim_prof_read(sim._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size, C.byref(khz)), sim._ctx)
    r = buf.reshape(W, 16).astype(np.int64)
    nb, tm, rec = r[:, 11], r[:, 12] / (khz.value / 1000.0), r[:, 13]
    print("t=%d big slots %d (per wave med %d max %d) records %d (max per wave %d) | wave time med %.1f max %.1f us | us per slot %.2f" %
          (target, nb.sum(), np.median(nb), nb.max(), rec.sum(), rec.max(), np.median(tm), tm.max(), tm.sum() / max(1, nb.sum())))
