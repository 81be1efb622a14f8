import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from epidemicsimulator_amd import Population, Simulator, _lib
pop = Population.synthetic("syn3m5")
ep = _lib.default_params(max_steps=5000)
X = 3308203
a = Simulator(pop, ep); a.set_pipeline(3)
b = Simulator(pop, ep); b.set_pipeline(2)
for s in (a, b):
    done = 0
    while done < 4224:
        s.run(96); done += 96
sa, sb = a.download_state(), b.download_state()
print("state equal at 4224:", all((sa[k] == sb[k]).all() for k in sa))
ah = pop.building_area[pop.home_building]; aw = pop.building_area[pop.work_building]
riders = np.nonzero((pop.flags & 1).astype(bool) & (ah == ah[X]) & (aw == aw[X]))[0]
print("route of X: home area", int(ah[X]), "work area", int(aw[X]), "riders", len(riders))
rb = b.run(31)            # steps 4225..4255
sb1 = b.download_state()
print("lvl2 after 4255: riders status", dict(zip(*np.unique(sb1["status"][riders], return_counts=True))), "X", int(sb1["status"][X]), int(sb1["eligible"][X]))
inf = riders[sb1["status"][riders] == 2]
print(" infected riders", inf.tolist(), "timers", sb1["timer"][inf].tolist())
r = b.run(1)
sb2 = b.download_state()
print("lvl2 step 4256:", {g: int(r[g][0]) for g in ("exposures_bus", "exposures_building", "vaccinated_now", "eligible_count", "n_riders")})
chg = riders[sb2["status"][riders] != sb1["status"][riders]]
print(" riders whose status changed in 4256 (lvl2):", [(int(c), int(sb1["status"][c]), int(sb2["status"][c])) for c in chg])
ra = a.run(31)
sa1 = a.download_state()
print("lvl3 after run(31) to 4255: state equal to lvl2:", all((sa1[k] == sb1[k]).all() for k in sa1))
r3 = a.run(1)
sa2 = a.download_state()
print("lvl3 step 4256:", {g: int(r3[g][0]) for g in ("exposures_bus", "exposures_building", "vaccinated_now", "eligible_count", "n_riders")})
d = np.nonzero((sa2["status"] != sb2["status"]) | (sa2["eligible"] != sb2["eligible"]))[0]
print("differ after 4256:", d[:10].tolist())
