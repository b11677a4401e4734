"""Feasibility prototype (NumPy, CPU; NOT part of the product or of the tests): time-PARALLEL Riccati and Kalman
covariance sweeps as associative scans (Sarkka & Garcia-Fernandez 2021: elements (A, C, J), combination with one n x n
inverse; for LQR the suffix scan of (A, B R^-1 B', Q) elements closed by (0, 0, Qf)), against the sequential recursions
of lqg/control/lqr.py:16-42 and lqg/belief/kf.py:6-21 on the model zoo.  Measured here (T = 500 / 1000, Hillis-Steele
order, max-norm relative error of every S_t / P_t against the sequential fp64 recursion):
    fp64: 1e-15 .. 3e-14 on every model (bounded, subjective, point-mass; action cost 0.01 .. 10)
    fp32: Riccati 2e-6 .. 9e-6 (sequential fp32: 2e-7 .. 4e-7), Kalman 2e-7 .. 8e-7
i.e. in fp64 the scan reproduces the sequential sweeps to rounding; in fp32 it loses about one digit.  This is the route
past the ~2x that a cooperative (workgroup-per-system) mapping gives in the one-system regime: log2(T)
dependent combines instead of T dependent steps.  The moment recursion of system.py:209-235 is the same construction with
m x m time-varying elements (exact observation of the leading block) and is the expensive part.
"""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, torch
import lqg_amd
import lqg_np as O

def combine(e1, e2):
    """e1 (earlier in scan order) ⊗ e2; each = (A, C, J) arrays [..., n, n]"""
    A1,C1,J1 = e1; A2,C2,J2 = e2
    n = A1.shape[-1]; I = np.eye(n, dtype=A1.dtype)
    M = I + C1 @ J2
    Mi_A1 = np.linalg.solve(M, A1)
    Mi_C1 = np.linalg.solve(M, C1)
    A = A2 @ Mi_A1
    C = A2 @ Mi_C1 @ np.swapaxes(A2,-1,-2) + C2
    N = I + J2 @ C1
    J = np.swapaxes(A1,-1,-2) @ np.linalg.solve(N, J2 @ A1) + J1
    C = 0.5*(C + np.swapaxes(C,-1,-2)); J = 0.5*(J + np.swapaxes(J,-1,-2))
    return A, C, J

def scan(elems):
    """inclusive prefix scan (Hillis-Steele) over axis 0"""
    A,C,J = [e.copy() for e in elems]
    T = A.shape[0]; d = 1
    while d < T:
        a = combine((A[:-d],C[:-d],J[:-d]), (A[d:],C[d:],J[d:]))
        A = np.concatenate([A[:d], a[0]]); C = np.concatenate([C[:d], a[1]]); J = np.concatenate([J[:d], a[2]])
        d *= 2
    return A,C,J

def riccati_scan(Am,Bm,Q,R,Qf,T,dt):
    n = Am.shape[0]
    U = (Bm @ np.linalg.inv(R) @ Bm.T).astype(dt)
    # elements in REVERSED time: index 0 = terminal (0,0,Qf), then steps T-1, T-2, ..., 0 ; scan composes a_{k,k+1} ⊗ (suffix)
    # we need suffix products a_k ⊗ a_{k+1} ⊗ ... ⊗ a_T : reversed-order prefix scan with swapped operands
    A = np.stack([np.zeros((n,n))]+[Am]*T).astype(dt); C = np.stack([np.zeros((n,n))]+[U]*T).astype(dt); J = np.stack([Qf]+[Q]*T).astype(dt)
    # prefix over reversed sequence with combine(new_earlier_in_time, accumulated_later): implement by flipping operand order
    d = 1; Tn = A.shape[0]
    while d < Tn:
        # element k (covering times k..k-?): accumulated later part is at index k-d (closer to terminal), earlier-in-time is index k
        a = combine((A[d:],C[d:],J[d:]), (A[:-d],C[:-d],J[:-d]))
        A = np.concatenate([A[:d], a[0]]); C = np.concatenate([C[:d], a[1]]); J = np.concatenate([J[:d], a[2]])
        d *= 2
    S = J  # S[0] = Qf = S_T ; S[j] = S_{T-j}
    return S[::-1]   # S_0..S_T

def riccati_seq(Am,Bm,Q,R,Qf,T,dt):
    S = Qf.astype(dt); out=[S]
    Am,Bm,Q,R = [x.astype(dt) for x in (Am,Bm,Q,R)]
    for t in range(T):
        H = R + Bm.T@S@Bm; G = Bm.T@S@Am
        L = -np.linalg.solve(H,G)
        S = Q + Am.T@S@Am + L.T@(H@L+G) + G.T@L
        S = 0.5*(S+S.T)
        out.append(S)
    return np.stack(out[::-1])

def kalman_scan(F,H,Q,R,P0,T,dt):
    n=F.shape[0]; I=np.eye(n)
    F,H,Q,R,P0=[x.astype(dt) for x in (F,H,Q,R,P0)]
    S = H@Q@H.T + R; K = Q@H.T@np.linalg.inv(S)
    Ak = (I-K@H)@F; Ck=(I-K@H)@Q; Jk = F.T@H.T@np.linalg.inv(S)@H@F
    P1 = F@P0@F.T+Q; S1 = H@P1@H.T+R; K1 = P1@H.T@np.linalg.inv(S1)
    C1 = P1 - K1@S1@K1.T
    A = np.stack([np.zeros((n,n))]+[Ak]*(T-1)).astype(dt); C=np.stack([C1]+[Ck]*(T-1)).astype(dt); J=np.stack([np.zeros((n,n))]+[Jk]*(T-1)).astype(dt)
    A,C,J = scan((A,C,J))
    return C   # filtered P_{t|t}, t=0..T-1

def kalman_seq(F,H,Q,R,P0,T,dt):
    F,H,Q,R,P=[x.astype(dt) for x in (F,H,Q,R,P0)]
    out=[]
    for t in range(T):
        Pp = F@P@F.T+Q; G = H@Pp@H.T+R; K = Pp@H.T@np.linalg.inv(G); P = (np.eye(F.shape[0])-K@H)@Pp
        out.append(P)
    return np.stack(out)

def rel(a,b): return float(np.abs(a-b).max()/np.abs(b).max())
for name,m in (("bounded",lqg_amd.BoundedActor(T=500,device='cpu',dtype=torch.float64,sigma_target=20.,action_cost=0.05)),
               ("subjective",lqg_amd.SubjectiveActor(T=500,device='cpu',dtype=torch.float64)),
               ("pointmass",lqg_amd.PointMassBoundedActor(T=500,device='cpu',dtype=torch.float64)),
               ("bounded_cost10",lqg_amd.BoundedActor(T=1000,device='cpu',dtype=torch.float64,action_cost=10.,sigma_target=50.)),
               ("bounded_cost.01",lqg_amd.BoundedActor(T=1000,device='cpu',dtype=torch.float64,action_cost=0.01,sigma_target=1.))):
    a=m.actor; g=lambda t: t[0].numpy()
    Am,Bm,Q,R,Fm,V,W = g(a.A),g(a.B),g(a.Q),g(a.R),g(a.F),g(a.V),g(a.W); Qf=a.Qf.numpy(); T=m.T
    ref = riccati_seq(Am,Bm,Q,R,Qf,T,np.float64)
    for dt in (np.float64,np.float32):
        S = riccati_scan(Am,Bm,Q,R,Qf,T,dt)
        Sq = riccati_seq(Am,Bm,Q,R,Qf,T,dt)
        P0=V@V.T
        Pref = kalman_seq(Am,Fm,V@V.T,W@W.T,P0,T,np.float64)
        Ps = kalman_scan(Am,Fm,V@V.T,W@W.T,P0,T,dt); Pq = kalman_seq(Am,Fm,V@V.T,W@W.T,P0,T,dt)
        print(f"{name:16s} {dt.__name__}: riccati scan {rel(S,ref):.2e} (seq-in-dtype {rel(Sq,ref):.2e})  kalman scan {rel(Ps,Pref):.2e} (seq {rel(Pq,Pref):.2e})")
