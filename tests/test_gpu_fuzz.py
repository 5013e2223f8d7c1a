"""GPU: the HIP step kernel against the oracle on HOSTILE states -- far outside anything a rollout visits: joints driven
deep past their limits, bodies well inside the floor, large velocities, controls far beyond ctrlrange.  The two
implementations must flag the same units (per-unit status words equal) and agree on the rest; nothing may fault or hang.
(MuJoCo would raise MujocoException on some of these; here they are per-unit status bits.)"""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", util.ENVS)
def test_hostile_states_step_matches_oracle(env):
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    from oracle import Oracle
    o, e = Oracle(env), FdEngine(env)
    rng = np.random.default_rng(42)
    N = 600
    q, v, c = util.reset_like(env, N, 5, o.nq, o.nu)
    q += rng.normal(0, 0.6, q.shape) * (rng.random((N, 1)) < 0.7)          # joints past their limits, bodies in the floor
    v = rng.normal(0, 1.0, v.shape) * rng.choice([0.1, 3.0, 30.0], (N, 1))
    c = rng.normal(0, 1.0, c.shape) * rng.choice([0.5, 5.0, 50.0], (N, 1))
    if env in ("HopperEnv", "Walker2dEnv", "HalfCheetahEnv"):
        q[: N // 3, 1] -= rng.uniform(0, 0.4, N // 3)                       # a third of them pushed into the ground
    qo, vo, ro, so = o.step(q, v, c, nthreads=8)
    qn, vn, r, st = e.step(q, v, c)
    st = st.cpu().numpy()
    same = st == so.astype(np.int32)              # incl. contact capacity: units beyond the fast kernel's 8 slots are re-run
    assert same.all(), (env, int((~same).sum()))
    ok = (so == 0) & (st == 0)
    assert ok.mean() > 0.5, ok.mean()
    scale = 1.0 + np.abs(np.concatenate([qo, vo], 1)[ok]).max(axis=1, keepdims=True)
    err = np.abs(np.concatenate([qn.cpu().numpy() - qo, vn.cpu().numpy() - vo], 1)[ok]) / scale
    assert err.max() < 1e-9, (env, float(err.max()))


@pytest.mark.parametrize("env", ["HalfCheetahEnv", "HopperEnv", "Walker2dEnv"])
def test_hostile_states_linearise_without_fault(env):
    """The FD launches (both schedules) on the same kind of input: equal bits between the schedules, status words
    equal to the oracle's, finite Jacobians wherever no rollout of the unit was flagged."""
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    from oracle import Oracle
    o, e = Oracle(env), FdEngine(env)
    rng = np.random.default_rng(7)
    N = 1500
    q, v, c = util.reset_like(env, N, 6, o.nq, o.nu)
    q += rng.normal(0, 0.5, q.shape) * (rng.random((N, 1)) < 0.6)
    v = rng.normal(0, 1.0, v.shape) * rng.choice([0.1, 3.0, 20.0], (N, 1))
    c = rng.normal(0, 1.0, c.shape) * rng.choice([0.5, 5.0], (N, 1))
    a = e.fd_jacobians(q, v, c, planned=False)
    b = e.fd_jacobians(q, v, c, planned=True)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    ref = o.fd_jacobians(q[:300], v[:300], c[:300], nthreads=8)
    st = a["status"].cpu().numpy()
    assert (st[:300] == ref["status"].astype(np.int32)).all()
    clean = a["status"] == 0
    assert bool(torch.isfinite(a["dsds"][clean]).all()) and bool(torch.isfinite(a["dsda"][clean]).all())
