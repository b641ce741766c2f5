import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch
import oracle_lib as ol, parity_cases as pc
from cmad_amd.models.device import DeviceEvaluator, build_desc
from cmad_amd.models.history_engine import HistoryEngine
from cmad_amd.synthetic import gauss_point_batch
rng=np.random.default_rng(5)
values=pc._lame_values(rng,"hosford",{"a":4.0})
mat=ol.Material(values,def_type=0,model_kind=ol.SMALL_RATE_EP)
desc,info=build_desc(values,def_type=0,model_kind=1)
ev=DeviceEvaluator(desc,info); eng=HistoryEngine(ev)
B,K=3,4
base=gauss_point_batch(B,seed=5,ndims=3)
gh=np.stack([k*0.7*base for k in range(K+1)])
xi0=np.repeat(mat.init_xi()[:,None],B,axis=1)
xi_hist,sig=eng.primal(gh,xi0)
sbar=np.zeros((K+1,6,B))
g,dx=eng.direct(gh,xi_hist,sbar,want_blocks=True)
t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
prev=None
for k in range(1,K+1):
    d,_=ev.direct_step(t(gh[k]),t(xi_hist[k-1]),t(xi_hist[k]),prev,gradu_prev=t(gh[k-1]))
    prev=d
    dd=d.cpu().numpy()
    err=np.abs(dd-dx[k])
    print("step",k,"max |direct_step| %.3e"%np.abs(dd).max(),"max diff %.3e"%err.max(), "argmax",np.unravel_index(err.argmax(),err.shape))
    if err.max()>1e-8:
        i=np.unravel_index(err.argmax(),err.shape)
        print(" hist",dx[k][:, :, i[2]][:, :7]); print(" step",dd[:, :, i[2]][:, :7])

# oracle reference for dx
nx=mat.nx
KP2O=pc.KP2O
print("---- oracle comparison")
for b in range(B):
    dxp=np.zeros((nx,12))
    for k in range(1,K+1):
        U,Up=gh[k][:,b],gh[k-1][:,b]; x,xp=xi_hist[k][:,b],xi_hist[k-1][:,b]
        A=mat.jacobian(ol.W_XI,x,xp,U,Up); Bm=mat.jacobian(ol.W_XI_PREV,x,xp,U,Up); P=mat.jacobian(ol.W_PARAMS,x,xp,U,Up)[:,KP2O]
        dxp=-np.linalg.solve(A,P+Bm@dxp)
        e=np.abs(dx[k][:,:6,b]-dxp[:,:6]).max()
        print("pt",b,"step",k,"hist vs oracle %.3e"%e, "alpha",x[6], "scale %.3e"%np.abs(dxp[:,:6]).max())
