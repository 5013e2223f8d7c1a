"""GPU: more floor contacts than the fast kernel's eight slots -- a walker or a cheetah lying flat on the ground.  The
fast kernel flags such units MBRL_ST_OVERFLOW; the fix-up launch that follows every rollout launch re-runs them with one
contact slot per floor candidate, so the caller sees no flag and results that agree with the oracle (24 contacts) like
any other unit's."""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def flat_states(env, N, seed):
    """Bodies laid out along the ground: every capsule (nearly) horizontal and within reach of the floor."""
    rng = np.random.default_rng(seed)
    if env == "Walker2dEnv":
        q = np.zeros((N, 9))
        q[:, 1] = rng.uniform(0.03, 0.07, N)                       # torso height (absolute)
        q[:, 2] = rng.choice([-1.0, 1.0], N) * (np.pi / 2) + rng.normal(0, 0.03, N)
        q[:, 3:] = rng.normal(0, 0.04, (N, 6))
    else:                                                          # HalfCheetah: belly down, legs folded flat
        q = np.zeros((N, 9))
        q[:, 1] = rng.uniform(-0.66, -0.62, N)                     # torso centre 0.04 .. 0.08 above the floor
        q[:, 2] = rng.normal(0, 0.02, N)
        q[:, 3] = 1.5 + rng.normal(0, 0.03, N)                     # back thigh horizontal (past its limit)
        q[:, 6] = -1.5 + rng.normal(0, 0.03, N)                    # front thigh horizontal
        q[:, [4, 5, 7, 8]] = rng.normal(0, 0.03, (N, 4))
    q[:, 0] = rng.normal(0, 1.0, N)
    v = rng.normal(0, 0.3, (N, 9))
    c = np.clip(rng.normal(0, 0.5, (N, 6)), -1, 1)
    return q, v, c


@pytest.mark.parametrize("env", ["Walker2dEnv", "HalfCheetahEnv"])
def test_units_with_more_than_eight_contacts_are_fixed_up(env):
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    from oracle import Oracle
    o, e = Oracle(env), FdEngine(env)
    N = 96
    q, v, c = flat_states(env, N, 3)
    # the states do overflow the fast kernel (seen with the fix-up switched off) ...
    os.environ["MBRL_NO_OVERFLOW_REDO"] = "1"
    try:
        raw = e.step(q, v, c)[3].cpu().numpy()
    finally:
        del os.environ["MBRL_NO_OVERFLOW_REDO"]
    assert ((raw & 4) != 0).sum() >= N // 8, int(((raw & 4) != 0).sum())
    # ... and the caller never sees it: env.step
    qo, vo, ro, so = o.step(q, v, c, nthreads=8)
    qn, vn, r, st = e.step(q, v, c)
    assert (so == 0).all() and int(st.abs().sum()) == 0
    scale = 1.0 + np.abs(np.concatenate([qo, vo], 1)).max(axis=1, keepdims=True)
    err = np.abs(np.concatenate([qn.cpu().numpy() - qo, vn.cpu().numpy() - vo], 1)) / scale
    assert err.max() < 1e-9, float(err.max())
    assert np.abs(r.cpu().numpy() - ro).max() < 1e-9 * (1 + np.abs(ro).max())
    # the linearisation, both schedules and both difference modes
    for fd_mode in (0, 1):
        ref = o.fd_jacobians(q, v, c, fd_mode=fd_mode, nthreads=8)
        a = e.fd_jacobians(q, v, c, fd_mode=fd_mode, planned=False)
        b = e.fd_jacobians(q, v, c, fd_mode=fd_mode, planned=True)
        torch.cuda.synchronize()
        for k in a:
            assert torch.equal(a[k], b[k]), (k, fd_mode)
        assert int(a["status"].abs().sum()) == 0 and (ref["status"] == 0).all()
        # the nominal rollout of a linearisation is the step launch's, bit for bit (the reference asserts it,
        # backward.py:371,374) -- also where the two are re-run by different means (in the wave / by the fix-up launch)
        assert torch.equal(a["s_next"], torch.cat([qn, vn], 1)) and torch.equal(a["reward"], r), fd_mode
        for k in ("dsds", "dsda", "drds", "drda"):
            x, y = a[k].cpu().numpy().reshape(N, -1), ref[k].reshape(N, -1)
            rel = np.abs(x - y).max(axis=1) / (1.0 + np.abs(y).max(axis=1))
            # a dozen stiff contact rows at once: "looser where contact is active" (north_star); forward differences
            # multiply the 1e-12 between two formulations by 1e6, central ones by 5e5 (measured worst: 6.4e-7)
            assert rel.max() < (5e-6 if fd_mode == 0 else 2e-6), (k, fd_mode, float(rel.max()))
        assert np.abs(a["s_next"].cpu().numpy() - ref["s_next"]).max() < 1e-8


def test_recorded_plan_of_overflowing_units():
    """mbrl_step_record on flat walkers: the fix-up re-records the units' plan entries, and the planned launch on the
    recorded plan gives the bits of the ordinary schedules."""
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    e = FdEngine("Walker2dEnv")
    N = 64
    q, v, c = (torch.as_tensor(x, device="cuda") for x in flat_states("Walker2dEnv", N, 5))
    ref = e.fd_jacobians(q, v, c, planned=False)
    qn, vn, r, st = e.step(q, v, c)
    plan = e.new_plan(N)
    Q1, V1 = torch.empty_like(q), torch.empty_like(v)
    R, S = torch.empty(N, dtype=torch.float64, device="cuda"), torch.empty(N, dtype=torch.int32, device="cuda")
    e.step_record(q, v, c, Q1, V1, R, S, plan, N, 0)
    assert torch.equal(Q1, qn) and torch.equal(V1, vn) and torch.equal(R, r) and torch.equal(S, st) and int(S.abs().sum()) == 0
    e.plan_finish(plan, N, R)
    out = e.new_outputs(N)
    out["status"].copy_(S)
    e.fd_planned(q, v, c, out, plan=plan)
    torch.cuda.synchronize()
    for k in ("dsds", "dsda", "drds", "drda", "status"):
        assert torch.equal(ref[k], out[k]), k
