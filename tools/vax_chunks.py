"""How the steps under a vaccination programme are cut into chunks: from the step before the programme starts, calls of 96 steps;
per call the chunk passes it took, the cuts, the steps each pass committed on average.   python tools/vax_chunks.py [preset] [from_step]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epidemicsimulator_amd import Population, Simulator, _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "uk64m"
start = int(sys.argv[2]) if len(sys.argv) > 2 else 3264
sim = Simulator(Population.synthetic(preset), _lib.default_params(max_steps=5000))
sim.enable_kernel_timing(16)
sim.run(start)
sim.chunk_timing(); sim.vax_chunk_stats()
while sim._steps < 5000:
    n = min(96, 5000 - sim._steps)
    rec = sim.run(n)
    kc, kv = sim.chunk_timing(), sim.vax_chunk_stats()
    print("steps %4d..%4d: chunk passes %2d (%.2f ms), under the programme %3d steps, cuts so far %3d, repairs %3d, Infected %6d, bus exposures %4d, vaccinated now %5d, eligible %d"
          % (sim._steps - n + 1, sim._steps, kc["chunks"], kc["chunk_ms"], kv["steps"], kv["cuts"], kv["repairs"], int(rec["infected"][-1]), int(rec["exposures_bus"].sum()),
             int(rec["vaccinated_now"][-1]), int(rec["eligible_count"][-1])))
if "--tail" in sys.argv:
    sim.reset(); rec = sim.run(5000)
    for a in range(4700, 5000, 20):
        r = rec[a:a + 20]
        print("steps %d..%d lockdown %s riders %s bus exposures %s" % (a + 1, a + 20, "".join(str(int(x)) for x in r["lockdown"]), "".join("1" if x else "0" for x in r["n_riders"]), [int(x) for x in r["exposures_bus"]]))
