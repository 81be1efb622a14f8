import sys, json, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import _oracle
from epidemicsimulator_amd import Population, Simulator, _lib
g = json.load(open('/root/repo/tests/golden/oracle_small_world.json'))
pop = Population.synthetic("york", **g["spec"])
ep=_lib.default_params(**g["params"])
for level in (3,2):
    sim = Simulator(pop, ep); sim.set_pipeline(level)
    rec = sim.run(g["steps"])
    orc=_oracle.Oracle(pop,_oracle.params_from_esim(ep)); want=orc.run(g["steps"])
    bad=False
    for f in rec.dtype.names:
        if f in ("reserved",): continue
        if not (rec[f]==want[f]).all():
            i=int(np.argmax(rec[f]!=want[f])); print('level',level,'field',f,'first mismatch at step',i+1,'gpu',rec[f][i],'orc',want[f][i]); bad=True
    if bad:
        i=min(int(np.argmax(rec[f]!=want[f])) for f in rec.dtype.names if f!='reserved' and (rec[f]!=want[f]).any())
        for k in range(max(0,i-2), i+3):
            print(k+1, {f:(int(rec[f][k]),int(want[f][k])) for f in ("susceptible","exposed","infected","recovered","vaccinated","exposures_building","exposures_bus","vaccinated_now","eligible_count","lockdown","mask_status","n_riders","vaccination_active")})
    else: print('level',level,'all records match; vaccinated', int(rec['vaccinated'][-1]))
    sg,so=sim.download_state(),orc.state()
    for k in sg: print(' state',k,'equal',bool((sg[k]==so[k]).all()))
    dbg=(__import__('ctypes').c_uint32*16)(); sim.lib.esim_debug_counters(sim._ctx, dbg); print(' dbg', list(dbg))
    sim.close()
