# (not collected by pytest: an analysis script on the CPU oracle, kept beside the tests because only tests may use the oracle)
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracedicom_amd import luts, scenarios
from oracle import oracle
es=luts.synth_luts()
ct,_=scenarios.hetero_phantom(256)   # CT resolution does not matter for the BEV-side quantities
scn=scenarios.hetero_ct(es,n=256,angles=[0.0],ct=ct)
d=np.zeros_like(scn.ct); of=oracle.run_field(scn,scn.beams[0],d,keep_layers=True)
W,H,L=of.info["ray_dims"]; S=scn.beams[0].tracerSteps
plan=of.get("layer_plan").reshape(L,8); first=of.info["beam_first_inside"]
rw=of.get("ray_weights").reshape(L,H,W)
fp=of.get("first_passive").reshape(L,H,W)
tx,ty=W//32,H//8; nT=tx*ty; nCU=256
# true cost model: per wave (2 rows x 32 cols) live steps = max over its rays of (first_passive - first) if any ray live else ~small
cost=np.zeros((L,nT,2))
for l in range(L):
    steps=int(plan[l,5])-first
    for t in range(nT):
        y0,x0=(t//tx)*8,(t%tx)*32
        wsum=0.0
        for w in range(4):
            live=rw[l,y0+2*w:y0+2*w+2,x0:x0+32]>=1.0
            wsum += steps*(1.0 if live.any() else 0.35)
        cost[l,t,0]=wsum*1.55; cost[l,t,1]=wsum*1.0
# current placement
afterLast=plan[:,5]
items=sorted(range(2*L),key=lambda p:(-(afterLast[p>>1]*(100 if p&1 else 155)),p))
nB=2*nT*L
load_cur=np.zeros(nCU)
for b in range(nB):
    item=b
    if nB<=6*nCU:
        rr,c,nFull=b//nCU,b%nCU,nB//nCU
        rev = rr<nFull and ((nFull-1-rr)&1)==0
        item=rr*nCU+((nCU-1-c) if rev else c)
    pr=items[item//nT]; l,role=pr>>1,pr&1; t=item%nT
    load_cur[b%nCU]+=cost[l,t,role]
flat=sorted(((cost[l,t,r],l,t,r) for l in range(L) for t in range(nT) for r in range(2)),reverse=True)
load_lpt=np.zeros(nCU)
for c,l,t,r in flat:       # true LPT: always to the least loaded CU
    i=int(np.argmin(load_lpt)); load_lpt[i]+=c
load_snake=np.zeros(nCU)
for i,(c,l,t,r) in enumerate(flat):
    rr,cc=i//nCU,i%nCU
    load_snake[(nCU-1-cc) if rr&1 else cc]+=c
for name,ld in (("current",load_cur),("snake by true cost",load_snake),("LPT",load_lpt)):
    print("%-20s max/mean %.3f  min/mean %.3f"%(name,ld.max()/ld.mean(),ld.min()/ld.mean()))
