"""SubjectiveActor — mirrors lqg/tracking/subjective.py:7-47: the actor's internal model has a target
velocity state the true dynamics lack (xdim = 2*dim, bdim = 3*dim); observed dims are permuted to the front."""
from itertools import chain

from lqg_amd.system import Actor, Dynamics, System
from lqg_amd.tracking import _build as bd


def swap_dims(d, dim):
    """Index order that moves the two observed dims of each axis in front (subjective.py:7-12)."""
    per = d // dim
    obs = [list(range(per * i, per * i + 2)) for i in range(dim)]
    unobs = [list(range(per * i + 2, per * (i + 1))) for i in range(dim)]
    return list(chain(*(obs + unobs)))


class SubjectiveActor(System):
    def __init__(self, dim=1, process_noise=1., action_cost=1., action_variability=0.5, subj_noise=1.,
                 subj_vel_noise=.5, sigma_target=6., sigma_cursor=6., dt=1. / 60, T=1000, device=None, dtype=None):
        device, dtype = bd.resolve(device, dtype, process_noise, action_cost, action_variability, subj_noise,
                                   subj_vel_noise, sigma_target, sigma_cursor)
        (pn, ac, av, sn, svn, st, sc), lead = bd.params(device, dtype, process_noise, action_cost,
                                                        action_variability, subj_noise, subj_vel_noise,
                                                        sigma_target, sigma_cursor)
        c = lambda rows: bd.const(rows, lead, device, dtype)
        eye = lambda n: c([[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)])
        # true dynamics                                                          subjective.py:18-25
        A = eye(2 * dim)
        B = bd.block_diag(*[c([[0.], [1. * dt]])] * dim)
        F = eye(2 * dim)
        V = bd.block_diag(*[bd.diag([pn, av])] * dim)
        W = bd.block_diag(*[bd.diag([st, sc])] * dim)
        dyn = Dynamics(A=A, B=B, F=F, V=V, W=W, T=T)
        # the actor's subjective model: target position, cursor, target velocity  subjective.py:27-36
        A = bd.block_diag(*[c([[1., 0., dt], [0., 1., 0.], [0., 0., 1.]])] * dim)
        B = bd.block_diag(*[c([[0.], [1. * dt], [0.]])] * dim)
        F = bd.block_diag(*[c([[1., 0., 0.], [0., 1., 0.]])] * dim)
        V = bd.block_diag(*[bd.diag([sn, av, svn])] * dim)
        Q = bd.block_diag(*[c([[1., -1., 0.], [-1., 1., 0.], [0., 0., 0.]])] * dim)
        R = bd.diag([ac] * dim)
        dims = swap_dims(A.shape[-1], dim)                                      # subjective.py:38-44
        A = bd.take(bd.take(A, dims, -2), dims, -1)
        B = bd.take(B, dims, -2)
        V = bd.take(V, dims, -2)
        F = bd.take(F, dims, -1)
        Q = bd.take(bd.take(Q, dims, -2), dims, -1)
        act = Actor(A=A, B=B, F=F, V=V, W=W, Q=Q, R=R, T=T)
        super().__init__(actor=act, dynamics=dyn)
        self._zoo_structure = dict(dim=dim)
