"""CPU: property tests of the block-decoupling host logic (lqg_amd/decouple.py): for random systems assembled from K
independent blocks under random permutations of the state / belief / control / observation indices, the interaction
graph's components are exactly the blocks, observed dims lead each component's state list, and blocks without observed
dims are dropped (they do not enter the likelihood)."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from lqg_amd.decouple import components_from_masks


def _assemble(rng, sizes, d_obs_per_block):
    """sizes: list of (nx, nb, nu, ny) per block.  Returns dims, masks, ground-truth blocks (index sets after permutation)."""
    NX, NB, NU, NY = (sum(s[i] for s in sizes) for i in range(4))
    # observed dims = the first d entries of x AFTER permutation: choose which original x indices are observed
    xo, bo, uo, yo = 0, 0, 0, 0
    blocks, observed = [], []
    for (nx, nb, nu, ny), dob in zip(sizes, d_obs_per_block):
        blocks.append((list(range(xo, xo + nx)), list(range(bo, bo + nb)), list(range(uo, uo + nu)), list(range(yo, yo + ny))))
        observed += list(range(xo, xo + min(dob, nx)))
        xo, bo, uo, yo = xo + nx, bo + nb, uo + nu, yo + ny
    rest = [i for i in range(NX) if i not in observed]
    rng.shuffle(observed)
    rng.shuffle(rest)
    px = observed + rest                                  # new position k holds original index px[k]
    pb, pu, py = (list(rng.permutation(n)) for n in (NB, NU, NY))
    inv = lambda p: {orig: new for new, orig in enumerate(p)}
    ix, ib, iu, iy = inv(px), inv(pb), inv(pu), inv(py)
    m = {k: np.zeros(s, dtype=bool) for k, s in dict(Ad=(NX, NX), Bd=(NX, NU), Fd=(NY, NX), N1=(NX, NX), WWd=(NY, NY),
                                                      Aa=(NB, NB), Ba=(NB, NU), Fa=(NY, NB), VVa=(NB, NB), WWa=(NY, NY),
                                                      Q=(NB, NB), Rr=(NU, NU)).items()}
    truth = []
    for xs, bs, us, ys in blocks:
        X, B, U, Y = [ix[i] for i in xs], [ib[i] for i in bs], [iu[i] for i in us], [iy[i] for i in ys]
        # a connected block: chain links inside every index family plus full coupling matrices
        for fam, name in ((X, "Ad"), (B, "Aa"), (Y, "WWd"), (U, "Rr")):
            for a, b in zip(fam, fam[1:]):
                m[name][a, b] = m[name][b, a] = True
            for a in fam:
                m[name][a, a] = True
        for x in X:
            m["Bd"][x, U[0]] = True
            m["Fd"][Y[0], x] = True
        for b in B:
            m["Ba"][b, U[0]] = True
            m["Fa"][Y[0], b] = True
            m["Q"][b, b] = True
        truth.append((sorted(X), sorted(B), sorted(U), sorted(Y)))
    d = len(observed)
    return dict(x=NX, b=NB, u=NU, y=NY, d=d), m, truth


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(2, 4), st.data())
def test_components_are_exactly_the_independent_blocks(seed, K, data):
    rng = np.random.default_rng(seed)
    sizes = [tuple(data.draw(st.integers(1, 3)) for _ in range(4)) for _ in range(K)]
    dobs = [data.draw(st.integers(0, 2)) for _ in range(K)]
    if sum(min(d, s[0]) for d, s in zip(dobs, sizes)) == 0:
        dobs[0] = 1
    dims, masks, truth = _assemble(rng, sizes, dobs)
    comps = components_from_masks(dims, masks)
    d = dims["d"]
    expect = [t for t in truth if any(i < d for i in t[0])]                   # blocks with at least one observed dim
    assert comps is not None and len(comps) == len(expect)
    got = sorted((sorted(c["xs"]), sorted(c["bs"]), sorted(c["us"]), sorted(c["ys"])) for c in comps)
    assert got == sorted(expect)
    seen = []
    for c in comps:
        obs = [i for i in c["xs"] if i < d]
        assert c["xs"][:len(obs)] == sorted(obs) == c["cols"]                 # observed dims lead, in data order
        seen += c["cols"]
    assert sorted(seen) == sorted(i for t in expect for i in t[0] if i < d)
    assert [c["cols"][0] for c in comps] == sorted(c["cols"][0] for c in comps)   # components ordered by first data column


def test_connected_graph_is_not_split():
    rng = np.random.default_rng(0)
    dims, masks, _ = _assemble(rng, [(2, 3, 1, 2)], [2])
    assert components_from_masks(dims, masks) is None
    dims, masks, _ = _assemble(rng, [(2, 2, 1, 1), (2, 2, 1, 1)], [1, 1])
    masks["Q"][0, dims["b"] - 1] = masks["Q"][dims["b"] - 1, 0] = True       # a cost term couples the two beliefs
    assert components_from_masks(dims, masks) is None


def test_identical_components_are_grouped_from_class_or_data():
    """decouple.identical_groups: both axes of a dim=2 zoo model are one 1-D system (grouped); a plain System with the same
    specs is grouped from the data; changing one axis' parameter un-groups it; an explicit Sigma0 disables merging."""
    import torch
    import lqg_amd
    from lqg_amd import decouple
    m = lqg_amd.SubjectiveActor(dim=2, T=12, device="cpu", dtype=torch.float64)
    parts = m.decoupled(4)
    assert parts is not None and len(parts) == 2
    assert decouple.identical_groups(m, 4, parts) == [[0, 1]]
    twin = lqg_amd.System(actor=m.actor, dynamics=m.dynamics)
    assert decouple.identical_groups(twin, 4, twin.decoupled(4)) == [[0, 1]]
    W0 = m.actor.W[0].clone()
    W0[3, 3] *= 2.0
    W = W0.expand(12, 4, 4)
    odd = lqg_amd.System(actor=m.actor._replace(W=W), dynamics=m.dynamics._replace(W=W))
    assert decouple.identical_groups(odd, 4, odd.decoupled(4)) == [[0], [1]]
    S0 = torch.eye(6, dtype=torch.float64)
    assert decouple.identical_groups(m, 4, parts, Sigma0=S0) == [[0], [1]]
    one = lqg_amd.BoundedActor(dim=1, T=12, device="cpu")
    assert one.decoupled(2) is None and decouple.identical_groups(one, 2, None) == []


def test_replacing_a_spec_voids_the_class_level_structure():
    """A zoo model whose spec is swapped after construction must not keep its class-derived pattern / identical-axes
    decision (they would be silently wrong for the new spec)."""
    import torch
    import lqg_amd
    from lqg_amd import decouple
    m = lqg_amd.SubjectiveActor(dim=2, T=12, device="cpu", dtype=torch.float64)
    assert decouple.identical_groups(m, 4, m.decoupled(4)) == [[0, 1]] and hasattr(m, "_zoo_structure")
    W0 = m.actor.W[0].clone()
    W0[3, 3] *= 2.0
    W = W0.expand(12, 4, 4)
    m.actor = m.actor._replace(W=W)
    m.dynamics = m.dynamics._replace(W=W)
    assert not hasattr(m, "_zoo_structure") and "_lqg_decouple" not in m.__dict__
    assert decouple.identical_groups(m, 4, m.decoupled(4)) == [[0], [1]]
    m64 = lqg_amd.BoundedActor(dim=2, T=5, device="cpu").to(torch.float64)          # a cast keeps the structure
    assert hasattr(m64, "_zoo_structure") and m64.actor.A.dtype == torch.float64


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_time_varying_costs_that_stay_costs_decouple_and_broken_ones_do_not(dtype):
    """Round 6: specs that move in time.  Costs moved by a congruence D_t Q D_t stay positive semi-definite (certified by Gershgorin
    on the unit-diagonal scaling, to the rounding of the storage dtype): the floor of lqr.py:27-28 is provably inactive and the two
    1-D components decouple.  Per-entry jitter breaks Q's zero eigenvalue into a negative one: refused, the joint problem is solved."""
    import bench_m2
    from lqg_amd import decouple
    dev = torch.device("cpu")
    good, _ = bench_m2.m2_system(dev, dtype, 16, 12, psd=True)
    assert decouple.floor_provably_inactive(good) and len(good.decoupled(4)) == 2
    for sub, cols, _ in good.decoupled(4):                       # the split keeps [T][element][system] storage
        assert sub.actor.A.stride(0) == 1 and sub.dynamics.F.stride(0) == 1 and sub.actor.A.shape[1] == 12
    bad, _ = bench_m2.m2_system(dev, dtype, 16, 12, psd=False)
    assert not decouple.floor_provably_inactive(bad) and bad.decoupled(4) is None
    # one step of one system with an indefinite cost is enough to refuse
    worse, _ = bench_m2.m2_system(dev, dtype, 16, 12, psd=True)
    worse.actor.Q[3, 7, 0, 1] -= 1e-2
    worse.actor.Q[3, 7, 1, 0] -= 1e-2
    assert not decouple.floor_provably_inactive(worse)
