"""Stage timers of k_chunk_tiny (diagnostics build: make -C epidemicsimulator_amd/csrc prof): where the one-launch chunk's time goes.
   python tools/tiny_stages.py [preset] [steps]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ.setdefault("ESIM_LIB", os.path.abspath("epidemicsimulator_amd/libesim_prof.so"))
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic(sys.argv[1] if len(sys.argv) > 1 else "uk64m")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sim = Simulator(pop, _lib.default_params(max_steps=5000))
sim.lib.esim_prof_read.restype = C.c_int
sim.lib.esim_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int)]
big = np.zeros(16384 * 16, np.uint32); khz = C.c_int(0)
names = ["census ahead", "decisions", "ctrl, ring lists, tables", "entries", "distinct keys", "items", "long lists", "big routes", "books"]
for rep in range(3):
    sim.reset(); sim.run(steps)
    _lib.check(sim.lib.esim_prof_read(sim._ctx, big.ctypes.data_as(C.POINTER(C.c_uint32)), big.size, C.byref(khz)), sim._ctx)
    r = big.reshape(16384, 16)[16382].astype(np.int64); k = khz.value / 1000.0
    bk = big.reshape(16384, 16)[16383].astype(np.int64)
    print("tiny chunk of %d steps: total %.1f us | " % (steps, (r[9] - r[0]) / k) + " | ".join("%s %.1f" % (names[i], (r[i + 1] - r[i]) / k) for i in range(9)))
    print("    books: exposures counted %.1f, census + records %.1f, log + clean-up %.1f, rest %.1f, next census %.1f us" % tuple(bk[i] / k for i in (0, 1, 2, 3, 4)))
