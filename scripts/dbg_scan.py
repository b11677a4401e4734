import os, sys, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,"tests")); sys.path.insert(0,os.path.join(R,"oracle"))
os.environ["LQG_NO_DECOUPLE"]="1"
import numpy as np, torch
from conftest import load_golden
from gpu_common import system_from_golden, np_
from lqg_amd import _hip, _abi
al=lambda v:(v+255)//256*256
for name in sys.argv[1:]:
    g, actor, dyn = load_golden(name)
    s = system_from_golden(actor, dyn, torch.float64)
    x = torch.as_tensor(g["x"], dtype=torch.float64, device="cuda")
    n=x.shape[0]
    ln=_hip.Launch(s.actor,s.dynamics,d=x.shape[-1],n_trials=n)
    lib=_abi.load()
    nb=lib.lqg_scan_workspace_bytes(C.byref(ln.p))
    ws=torch.zeros(nb,dtype=torch.uint8,device="cuda")
    Sig=ln.empty(ln.T,ln.m,ln.m); mu=ln.empty(n,ln.T,ln.m)
    rc=lib.lqg_conditional_moments_scan(C.byref(ln.p), ln.traj(x,False), ln.traj(mu), ln.view(Sig), C.c_void_p(ws.data_ptr()), nb, ln.stream())
    torch.cuda.synchronize()
    T=ln.T; b=ln.dims["b"]; u=ln.dims["u"]; y=ln.dims["y"]; m=ln.m
    er=max((T+1)*3*b*b, T*3*m*m)
    l_off=al(2*er*8); k_off=l_off+al(T*u*b*8); fg_off=k_off+al(T*b*y*8)
    Lb=ws[l_off:l_off+T*u*b*8].view(torch.float64).cpu().numpy().reshape(T,u,b)
    Kb=ws[k_off:k_off+T*b*y*8].view(torch.float64).cpu().numpy().reshape(T,b,y)
    FG=ws[fg_off:fg_off+T*2*m*m*8].view(torch.float64).cpu().numpy().reshape(T,2,m,m)
    eL=np.abs(Lb-g["L"]).reshape(T,-1).max(1)/np.abs(g["L"]).max(); eK=np.abs(Kb-g["K"]).reshape(T,-1).max(1)/np.abs(g["K"]).max()
    print(name,"rc",rc,"L err first/last/max", eL[0], eL[-1], eL.max(), "argmax",eL.argmax())
    print("   K err per t", np.array2string(eK[:6],precision=2), "max",eK.max())
    S=np_(Sig); ref=g["Sigma"][0]
    err=np.abs(S-ref).reshape(T,-1).max(1)/np.abs(ref).max()
    print("   Sigma err per t", np.array2string(err[:6],precision=2))
