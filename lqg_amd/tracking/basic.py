"""Tracking-task models — mirrors lqg/tracking/basic.py (TrackingTask :7-40, BoundedActor :43-64,
OptimalActor :67-87, RelativeObservationBoundedActor :90-124): same constructor arguments and defaults,
same matrices; parameters may additionally be 1-D tensors of B candidates."""
from lqg_amd.system import Actor, System
from lqg_amd.tracking import _build as bd


class TrackingTask(System):
    def __init__(self, dim=1, process_noise=1.0, action_variability=0.5, sigma_target=6.0, sigma_cursor=6.0,
                 action_cost=1.0, dt=1.0 / 60.0, T=1000, device=None, dtype=None):
        self.dim = dim
        self.process_noise = process_noise
        device, dtype = bd.resolve(device, dtype, process_noise, action_variability, sigma_target, sigma_cursor,
                                   action_cost)
        (pn, av, st, sc, ac), lead = bd.params(device, dtype, process_noise, action_variability, sigma_target,
                                               sigma_cursor, action_cost)
        d = 2 * dim
        # dynamics model: target random walk, cursor integrates the action        basic.py:21-24
        A = bd.const([[1.0 if i == j else 0.0 for j in range(d)] for i in range(d)], lead, device, dtype)
        B = dt * bd.block_diag(*[bd.const([[0.0], [1.0]], lead, device, dtype)] * dim)
        # observation model                                                      basic.py:27
        F = A
        # noise model                                                            basic.py:30-31
        V = bd.diag([pn, av] * dim)
        W = bd.diag([st, sc] * dim)
        # cost function: squared target-cursor distance, quadratic action cost    basic.py:34-35
        Q = bd.block_diag(*[bd.const([[1.0, -1.0], [-1.0, 1.0]], lead, device, dtype)] * dim)
        R = bd.diag([ac] * dim)
        spec = Actor(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)
        super().__init__(actor=spec, dynamics=spec)
        self._zoo_structure = dict(dim=dim)      # what (besides the class) fixes the sparsity pattern


class BoundedActor(TrackingTask):
    def __init__(self, dim=1, process_noise=1.0, action_variability=0.5, sigma_target=6.0, sigma_cursor=6.0,
                 action_cost=1.0, dt=1.0 / 60, T=1000, device=None, dtype=None):
        super().__init__(dim=dim, process_noise=process_noise, action_variability=action_variability,
                         sigma_target=sigma_target, sigma_cursor=sigma_cursor, action_cost=action_cost, dt=dt, T=T,
                         device=device, dtype=dtype)


class OptimalActor(TrackingTask):
    def __init__(self, dim=1, process_noise=1.0, action_variability=0.5, sigma_target=6.0, sigma_cursor=6.0,
                 dt=1.0 / 60, T=1000, device=None, dtype=None):
        super().__init__(dim=dim, process_noise=process_noise, action_variability=action_variability,
                         sigma_target=sigma_target, sigma_cursor=sigma_cursor, action_cost=1e-3, dt=dt, T=T,
                         device=device, dtype=dtype)


class RelativeObservationBoundedActor(System):
    def __init__(self, dim=1, process_noise=1.0, action_variability=0.5, sigma=6.0, action_cost=1.0, dt=1.0 / 60.0,
                 T=1000, device=None, dtype=None):
        self.dim = dim
        self.process_noise = process_noise
        device, dtype = bd.resolve(device, dtype, process_noise, action_variability, sigma, action_cost)
        (pn, av, sg, ac), lead = bd.params(device, dtype, process_noise, action_variability, sigma, action_cost)
        d = 2 * dim
        A = bd.const([[1.0 if i == j else 0.0 for j in range(d)] for i in range(d)], lead, device, dtype)
        B = dt * bd.block_diag(*[bd.const([[0.0], [1.0]], lead, device, dtype)] * dim)
        # only the target-cursor difference is observed                          basic.py:110
        F = bd.block_diag(*[bd.const([[1.0, -1.0]], lead, device, dtype)] * dim)
        V = bd.diag([pn, av] * dim)
        W = bd.diag([sg] * dim)
        Q = bd.block_diag(*[bd.const([[1.0, -1.0], [-1.0, 1.0]], lead, device, dtype)] * dim)
        R = bd.diag([ac] * dim)
        spec = Actor(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)
        super().__init__(actor=spec, dynamics=spec)
        self._zoo_structure = dict(dim=dim)
