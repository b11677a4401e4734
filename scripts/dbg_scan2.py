import os, sys, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,"tests")); sys.path.insert(0,os.path.join(R,"oracle"))
os.environ["LQG_NO_DECOUPLE"]="1"; os.environ["LQG_SCAN_DEBUG_STOP"]="1"
import numpy as np, torch
from conftest import load_golden
from gpu_common import system_from_golden, np_
from lqg_amd import _hip, _abi
name=sys.argv[1]
g, actor, dyn = load_golden(name)
s = system_from_golden(actor, dyn, torch.float64)
x = torch.as_tensor(g["x"][:1], dtype=torch.float64, device="cuda")
ln=_hip.Launch(s.actor,s.dynamics,d=x.shape[-1],n_trials=1)
lib=_abi.load()
nb=lib.lqg_scan_workspace_bytes(C.byref(ln.p))
ws=torch.zeros(nb,dtype=torch.uint8,device="cuda")
ll=ln.empty(1)
# n_trials>0 && ll: trial kernel would run on garbage ops; pass via moments entry with no outputs
rc=lib.lqg_conditional_moments_scan(C.byref(ln.p), ln.traj(x,False), _abi.NULL_TRAJ, _abi.NULL_VIEW, C.c_void_p(ws.data_ptr()), nb, ln.stream())
torch.cuda.synchronize()
T=ln.T; b=ln.dims["b"]; y=ln.dims["y"]; m=ln.m
er=max((T+1)*3*b*b, T*3*m*m)
bufs=[ws[i*er*8:(i*er+T*3*b*b)*8].view(torch.float64).cpu().numpy().reshape(T,3,b,b) for i in (0,1)]
# sequential reference
A=actor["A"][0]; F=actor["F"][0]; V=actor["V"][0]; W=actor["W"][0]
P=V@V.T; Ps=[]
for t in range(T):
    Pp=A@P@A.T+V@V.T; G=F@Pp@F.T+W@W.T; K=Pp@F.T@np.linalg.inv(G); P=(np.eye(b)-K@F)@Pp; Ps.append(P)
Ps=np.stack(Ps)
for i,bf in enumerate(bufs):
    e=np.abs(bf[:,1]-Ps).reshape(T,-1).max(1)
    print("buf",i,"C err per t", np.array2string(e[:8],precision=2))
# element check: rebuild elements in numpy and compare with what a level-0 buffer would hold is not available; print result A,J at t=1
Q=V@V.T; Rw=W@W.T; S=F@Q@F.T+Rw; K=Q@F.T@np.linalg.inv(S); Ak=(np.eye(b)-K@F)@A; Ck=(np.eye(b)-K@F)@Q; Jk=A.T@F.T@np.linalg.inv(S)@F@A
print("numpy elem1 A\n",Ak,"\nC\n",Ck,"\nJ\n",Jk)
