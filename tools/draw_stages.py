"""Per-item stage timers of k_chunk_draw (diagnostics build `make -C epidemicsimulator_amd/csrc prof`): where an item's ~4 us go."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ.setdefault("ESIM_LIB", os.path.abspath("epidemicsimulator_amd/libesim_prof.so"))
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic("uk64m")
sim = Simulator(pop, _lib.default_params(max_steps=5000))
W = 4096; buf = np.zeros(W * 16, np.uint32); khz = C.c_int(0)
sim.lib.esim_prof_read.restype = C.c_int
sim.lib.esim_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int)]
for target in (2880, 3840, 4800):
    sim.run(target - sim._steps)
    _lib.check(sim.lib.esim_prof_read(sim._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size, C.byref(khz)), sim._ctx)
    r = buf.reshape(W, 16).astype(np.int64); k = khz.value / 1000.0
    items = r[:, 4].sum()
    print("t=%d items %d rounds %d | per item (us): decode+counts %.2f, member loads issued %.2f, steps->slots %.2f, residents list %.2f, workers list %.2f | items phase per item %.2f"
          % (target, items, r[:, 6].sum(), *(r[:, c].sum() / k / items for c in (11, 12, 13, 14, 15)), (r[:, 2] - r[:, 1]).sum() / k / items))
