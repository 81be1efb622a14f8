"""Diagnostics: the vaccination-planned chunk form (pipeline level 3) against level 2 (sequential steps under a programme) on a
preset, block by block, down to the first record field and the first citizens that differ.  python tools/compare_forms.py preset [block]"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from epidemicsimulator_amd import Population, Simulator, _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "syn3m5"
block = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pop = Population.synthetic(preset)
ep = _lib.default_params(max_steps=5000)
a = Simulator(pop, ep); a.set_pipeline(3)
b = Simulator(pop, ep); b.set_pipeline(2)
done = 0
while done < 5000:
    ra, rb = a.run(block), b.run(block)
    done += block
    bad = [f for f in ra.dtype.names if f != "reserved" and not (ra[f] == rb[f]).all()]
    dbg = (C.c_uint32 * 16)(); a.lib.esim_debug_counters(a._ctx, dbg)
    if bad:
        f = bad[0]; i = int(np.argmax(ra[f] != rb[f]))
        print("block ending", done, "fields", bad, "first", f, "at step", done - block + i + 1, "lvl3", int(ra[f][i]), "lvl2", int(rb[f][i]))
        for k in range(max(0, i - 1), min(block, i + 2)):
            print("  step", done - block + k + 1, {g: (int(ra[g][k]), int(rb[g][k])) for g in ("susceptible", "exposed", "infected", "recovered", "vaccinated", "exposures_building", "exposures_bus", "vaccinated_now", "eligible_count")})
        sa, sb = a.download_state(), b.download_state()
        diff = np.nonzero((sa["status"] != sb["status"]) | (sa["timer"] != sb["timer"]) | (sa["eligible"] != sb["eligible"]))[0]
        print("  citizens that differ:", diff[:10].tolist())
        for c in diff[:5]:
            print("   ", int(c), "lvl3", int(sa["status"][c]), int(sa["timer"][c]), int(sa["eligible"][c]), "lvl2", int(sb["status"][c]), int(sb["timer"][c]), int(sb["eligible"][c]),
                  "flags", int(pop.flags[c]), "room", int(pop.room[c]))
        break
    if done % 500 == 0:
        print(done, "ok", {g: int(ra[g][-1]) for g in ("susceptible", "infected", "vaccinated")}, flush=True)
print("vax cuts so far: see debug counters", list(dbg))
