"""Per-wavefront timers of the chunk pass (diagnostics build: make -C epidemicsimulator_amd/csrc prof).
   python tools/wave_profile.py [preset]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ.setdefault("ESIM_LIB", os.path.abspath("epidemicsimulator_amd/libesim_prof.so"))
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic(sys.argv[1] if len(sys.argv) > 1 else "uk64m")
sim = Simulator(pop, _lib.default_params(max_steps=5000))
W = 4096
buf = np.zeros(W * 16, np.uint32)
khz = C.c_int(0)
sim.lib.esim_prof_read.restype = C.c_int
sim.lib.esim_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int)]
targets = [int(x) for x in sys.argv[2:]] or [96, 192, 960, 1056, 1920, 2880, 3840, 4800, 4896, 4992]
for target in targets:
    sim.run(target - sim._steps)
    _lib.check(sim.lib.esim_prof_read(sim._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size, C.byref(khz)), sim._ctx)
    r = buf.reshape(W, 16).astype(np.int64)
    us = lambda ticks: np.asarray(ticks, float) / (khz.value / 1000.0)
    t0 = r[:, 0].min()
    pre, items, routes, tot = us(r[:, 1] - r[:, 0]), us(r[:, 2] - r[:, 1]), us(r[:, 3] - r[:, 2]), us(r[:, 3] - r[:, 0])
    busy = r[:, 4] > 0
    print("t=%4d clock %d kHz | draw: starts within %.1f us, span %.1f us | preamble med %.1f max %.1f | items phase med %.1f max %.1f "
          "(waves with items %d, most items %d, longest item %.1f) | routes phase med %.1f max %.1f"
          % (target, khz.value, us(r[:, 0].max() - t0), us(r[:, 3].max() - t0), np.median(pre), pre.max(), np.median(items), items.max(),
             busy.sum(), r[:, 4].max(), us(r[:, 5].max()), np.median(routes), routes.max()))
    print("        draw: items in all %d, 64-pair rounds drawn on the spot %d (%.2f per item); items phase per item %.2f us, per round %.2f us"
          % (r[:, 4].sum(), r[:, 6].sum(), r[:, 6].sum() / max(1, r[:, 4].sum()), items.sum() / max(1, r[:, 4].sum()), items.sum() / max(1, r[:, 6].sum())))
    m0 = r[:, 8].min()
    mt = us(r[:, 9] - r[:, 8])
    print("        marks: starts within %.1f us, span %.1f us | wave med %.1f max %.1f | waves with entries %d, most entries %d"
          % (us(r[:, 8].max() - m0), us(r[:, 9].max() - m0), np.median(mt), mt.max(), (r[:, 10] > 0).sum(), r[:, 10].max()))
    b = r[:, 10] > 0
    if b.any():
        print("        marks stages (waves with entries, median / max us): words and keys %.1f/%.1f | hash claims %.1f/%.1f | item records %.1f/%.1f | "
              "record positions, records %.1f/%.1f | direct counts, route pairs %.1f/%.1f" % tuple(x for i in (11, 12, 13, 14, 15) for x in (np.median(us(r[b, i])), us(r[b, i]).max())))
    big = np.zeros(16384 * 16, np.uint32)
    _lib.check(sim.lib.esim_prof_read(sim._ctx, big.ctypes.data_as(C.POINTER(C.c_uint32)), big.size, C.byref(khz)), sim._ctx)
    bk = big.reshape(16384, 16)[16383].astype(np.int64)
    print("        books (us): count %.1f | finish %.1f | scatter+clean-up %.1f | next (future+decide) %.1f (of which after future %.1f)"
          % tuple(us(bk[i]) for i in (0, 1, 2, 3, 4)))
    print("        fold, first workgroup (us): counts + scan %.1f | prefix into LDS %.1f | searches + stores %.1f" % tuple(us(bk[i]) for i in (8, 9, 10)))
