"""-m gpu: the per-trial sweep split along time (lqg_amd/csrc/lqg_trial_chunk.hpp: zero-state pass, boundary fix-up, density
pass, chunk sum) against the one-pass k_trial / k_trial_sp and against the golden vectors of the reference's own source.

LQG_TRIAL_CHUNKS=0 is the one-pass sweep, =k forces k chunks; by default the chunk count follows the number of waves the
trials alone would put in flight."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from gpu_common import np_, system_from_golden

pytestmark = pytest.mark.gpu


def _ll(model, x, monkeypatch, chunks, scan="0", specialise=True):
    from lqg_amd.plan import LogLikelihoodPlan
    monkeypatch.setenv("LQG_TRIAL_CHUNKS", str(chunks))
    monkeypatch.setenv("LQG_SCAN", scan)
    monkeypatch.setenv("LQG_FUSE_TRIALS_MAX", "0")            # (few trials would otherwise run as fused (system, trial) pairs)
    if not specialise:
        monkeypatch.setenv("LQG_NO_SPECIALIZE", "1")
    else:
        monkeypatch.delenv("LQG_NO_SPECIALIZE", raising=False)
    return LogLikelihoodPlan(model, x).run().clone()


@pytest.mark.parametrize("specialise", [True, False], ids=["pattern", "generic"])
@pytest.mark.parametrize("model,T,n", [("bounded", 500, 50), ("pointmass", 500, 3000), ("subjective2d", 137, 70),
                                       ("bounded", 16, 5), ("bounded", 19, 300)])
def test_chunked_sweep_equals_one_pass_fp64(model, T, n, specialise, monkeypatch):
    import lqg_amd
    dev = torch.device("cuda")
    if model == "bounded":
        m, d = lqg_amd.BoundedActor(T=T, device=dev, dtype=torch.float64), 2
    elif model == "pointmass":
        m, d = lqg_amd.PointMassBoundedActor(T=T, action_variability=0.5, device=dev, dtype=torch.float64), 2
    else:
        m, d = lqg_amd.SubjectiveActor(dim=2, T=T, device=dev, dtype=torch.float64), 4
    with torch.no_grad():
        x = m.simulate(5, n=n)[..., :d].contiguous()
    ref = _ll(m, x, monkeypatch, 0, specialise=specialise)
    for chunks in (2, 3, 4, 7, 31):
        got = _ll(m, x, monkeypatch, chunks, specialise=specialise)
        assert got.shape == ref.shape
        assert float((got / ref - 1).abs().max()) < 1e-11, (chunks, float((got / ref - 1).abs().max()))
    # several systems at once (candidates of one parameter) and the default chunk rule
    monkeypatch.delenv("LQG_TRIAL_CHUNKS")
    got = _ll(m, x, monkeypatch, 0)
    monkeypatch.delenv("LQG_TRIAL_CHUNKS")
    from lqg_amd.plan import LogLikelihoodPlan
    auto = LogLikelihoodPlan(m, x).run().clone()
    assert float((auto / got - 1).abs().max()) < 1e-11


def test_chunked_sweep_many_systems_and_fp32(monkeypatch):
    import lqg_amd
    dev = torch.device("cuda")
    sig = torch.tensor([4.0, 9.0, 15.0], dtype=torch.float64, device=dev)
    m = lqg_amd.BoundedActor(T=300, sigma_target=sig, device=dev, dtype=torch.float64)
    with torch.no_grad():
        x = m.simulate(2, n=40)[0].contiguous()                  # one data set scored under three candidates
    ref = _ll(m, x, monkeypatch, 0)
    got = _ll(m, x, monkeypatch, 6)
    assert got.shape == (3, 40) and float((got / ref - 1).abs().max()) < 1e-11
    m32 = m.to(torch.float32)
    one = _ll(m32, x.float(), monkeypatch, 0).double()
    chk = _ll(m32, x.float(), monkeypatch, 6).double()
    e_one, e_chk = float((one / ref - 1).abs().max()), float((chk / ref - 1).abs().max())
    assert e_chk < max(2e-6, 2 * e_one), (e_chk, e_one)


@pytest.mark.parametrize("name", ["bounded_T100", "subjective1d_T50", "pointmass_d2_T50", "delay1_bounded_T30"])
def test_chunked_sweep_matches_golden(name, monkeypatch):
    g, actor, dyn = load_golden(name)
    if name not in golden_names():
        pytest.skip("golden not present")
    from lqg_amd.plan import LogLikelihoodPlan
    monkeypatch.setenv("LQG_TRIAL_CHUNKS", "5")
    monkeypatch.setenv("LQG_FUSE_TRIALS_MAX", "0")
    monkeypatch.setenv("LQG_NO_DECOUPLE", "1")
    sys_ = system_from_golden(actor, dyn, torch.float64)
    x = torch.as_tensor(np.concatenate([g["x"]] * 2), dtype=torch.float64, device="cuda")    # > 2 trials: not the fused path
    S0 = torch.as_tensor(g["Sigma0"], dtype=torch.float64, device="cuda") if "Sigma0" in g else None
    ll = np_(LogLikelihoodPlan(sys_, x, Sigma0=S0).run().clone())
    want = np.concatenate([g["ll"]] * 2)
    assert np.abs(ll / want - 1).max() < 1e-10
