"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Tolerances (fp64): s' and reward <= 1e-9 abs-rel; FD Jacobians <= 2e-6 relative to the largest
entry (one-sided differences with eps = 1e-6 amplify the ~1e-15 step-level disagreement between two
different arithmetic formulations -- 3-D oracle vs planar kernel -- by 1/eps; units whose active
constraint set flips inside a perturbation are compared column-wise with the same bound since both
sides see the same flip)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _engine(env):
    import mbrl_amd
    from mbrl_amd.engine import FdEngine
    return FdEngine(env)


@pytest.mark.parametrize("env", util.ENVS)
def test_step_matches_oracle(env, torch_cuda):
    from oracle import Oracle
    o = Oracle(env)
    e = _engine(env)
    q, v, c = util.on_trajectory(o, env, 512, seed=0)
    qo, vo, ro, so = o.step(q, v, c, nthreads=8)
    qn, vn, r, st = e.step(q, v, c)
    assert util.relerr(qn.cpu().numpy(), qo) < 1e-9
    assert util.relerr(vn.cpu().numpy(), vo) < 1e-8
    assert util.relerr(r.cpu().numpy(), ro) < 1e-8
    assert (st.cpu().numpy() == so.astype(np.int32)).all()


@pytest.mark.parametrize("fd_mode", [0, 1])
@pytest.mark.parametrize("env", util.ENVS)
def test_fd_jacobians_match_oracle(env, fd_mode, torch_cuda):
    from oracle import Oracle
    o = Oracle(env)
    e = _engine(env)
    q, v, c = util.on_trajectory(o, env, 300, seed=1)
    ref = o.fd_jacobians(q, v, c, fd_mode=fd_mode, nthreads=8)
    out = e.fd_jacobians(q, v, c, fd_mode=fd_mode)
    got = {k: t.cpu().numpy() for k, t in out.items()}
    assert util.relerr(got["s_next"], ref["s_next"]) < 1e-9
    assert util.relerr(got["reward"], ref["reward"]) < 1e-8
    for k in ("dsds", "dsda", "drds", "drda"):
        # per-unit scale: largest |entry| of that unit's Jacobian
        a, b = got[k].reshape(len(q), -1), ref[k].reshape(len(q), -1)
        scale = 1.0 + np.abs(b).max(axis=1, keepdims=True)
        err = np.abs(a - b) / scale
        assert err.max() < 2e-6, (k, float(err.max()), int(err.max(axis=1).argmax()))
    assert (got["status"] == ref["status"].astype(np.int32)).all()
