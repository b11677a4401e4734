"""CPU, build container only: the committed golden vectors are reproducible from the reference's own source.

Skipped where /root/reference is absent (the GPU box).  Re-runs two cases of oracle/gen_golden.py — the reference's
lqg.control.lqr.backward, lqg.belief.kf.forward, System.conditional_moments / log_likelihood / simulate executed under
oracle/jax_standin.py — and compares with tests/golden/*.npz bit for bit (same NumPy / LAPACK build)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lqg")), reason="reference checkout not present")
def test_golden_vectors_come_from_the_reference_source(tmp_path):
    code = f"""
import sys, os
sys.path.insert(0, {os.path.join(ROOT, 'oracle')!r})
import gen_golden as G
G.OUT = {str(tmp_path)!r}
from lqg.tracking import SubjectiveActor, BoundedActor
G.run_case("subjective1d_T50", SubjectiveActor(dim=1, T=50), n=3, d=2, seed=15)
G.run_case("bounded_T100", BoundedActor(T=100, sigma_target=6.0, sigma_cursor=1.0, action_cost=0.05,
                                        action_variability=0.5), n=3, d=2, seed=12)
"""
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.DEVNULL)
    for name in ("subjective1d_T50", "bounded_T100"):
        new, old = np.load(tmp_path / f"{name}.npz"), np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        assert sorted(new.files) == sorted(old.files)
        for k in old.files:
            assert np.array_equal(new[k], old[k]), (name, k)
