import sys; sys.path.insert(0,'.')
import numpy as np
from rapidnet_amd import capi, synth
p=synth.make_problem("barcelona493"); dh,ah=synth.forecast_at(p["forecast"],0)
d=capi.Solver(p["network"],p["tree"],p["config"]); d.initialiseSmpcController(dh,ah)
s=capi.Solver(p["network"],p["tree"],p["config"],structured=True); s.initialiseSmpcController(dh,ah)
d.apgReset(); s.apgReset()
tot=0
for n in (1,9,40,50,100,100,200):
    d.apgIterate(n,history=False); s.apgIterate(n,history=False); tot+=n
    out=[]
    for bid in (capi.BUF_X,capi.BUF_U,capi.BUF_UPD_XI,capi.BUF_UPD_PSI,capi.BUF_DUAL_XI,capi.BUF_RES_PSI):
        a,b=d.get(bid),s.get(bid); out.append(np.abs(a-b).max()/np.abs(a).max())
    print(tot, ["%.1e"%v for v in out])
