"""CPU: the oracle against the golden vectors produced by the reference's own Python
(tests/golden/gen_reference_golden.py) -- pins the FD schedule (eps, column order, layout, test=True
re-step), and the op chain rule / GRAD_WEIGHTS variants (SURVEY.md 8c pin (1), App. C)."""
import glob
import os

import numpy as np
import pytest

from oracle import Oracle

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_*.npz")))


def recurrence(g, weights):
    """numpy restatement of the backward recurrence, episodes visited last-to-first."""
    A, Bm, rs, ra, c = g["dsds"], g["dsda"], g["drds"], g["drda"], g["c"]
    B, H = c.shape
    da = np.zeros_like(g["actions"])
    ds0 = np.zeros((B, A.shape[-1]))
    S = 0.0
    for b in range(B - 1, -1, -1):
        gv = np.zeros(A.shape[-1])
        for t in range(H - 1, -1, -1):
            W = H - t
            wa_r = wa_d = ws_r = 1.0
            if weights == "average":
                wa_r = wa_d = 1.0 / W
            elif weights == "prioritise":
                S += W
                ws_r, wa_r, wa_d = W - 1.0, W / S, 1.0 / S
            da[b, t] = c[b, t] * wa_r * ra[b, t] + wa_d * (gv @ Bm[b, t])
            gv = c[b, t] * ws_r * rs[b, t] + gv @ A[b, t]
        ds0[b] = gv
    return da, ds0


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
def test_oracle_fd_reproduces_reference_calculate_gradients(path):
    g = np.load(path)
    o = Oracle(str(g["env"]))
    B, H = g["c"].shape
    s = g["states"][:, :H].reshape(B * H, -1)
    a = g["actions"].reshape(B * H, -1)
    out = o.fd_jacobians(s[:, :o.nq], s[:, o.nq:], a, eps=1e-6, fd_mode=0, nsteps=int(g["nsteps"]))
    # same step, same schedule, same arithmetic => bit-exact
    assert np.array_equal(out["dsds"].reshape(g["dsds"].shape), g["dsds"])
    assert np.array_equal(out["dsda"].reshape(g["dsda"].shape), g["dsda"])
    assert np.array_equal(out["drds"].reshape(g["drds"].shape), g["drds"])
    assert np.array_equal(out["drda"].reshape(g["drda"].shape), g["drda"])
    if int(g["nsteps"]) == 1:
        assert np.array_equal(out["s_next"].reshape(B, H, -1), g["states"][:, 1:])
        assert np.array_equal(out["reward"].reshape(B, H), g["rewards"])
        # the reference's reward op returns torch.Tensor([r]).double(): fp32-rounded (mujoco.py:38)
        assert np.array_equal(g["rewards"].astype(np.float32).astype(np.float64), g["rewards_op"])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
def test_recurrence_reproduces_reference_autograd(path):
    g = np.load(path)
    da, ds0 = recurrence(g, str(g["weights"]))
    np.testing.assert_allclose(da, g["grad_actions"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ds0, g["grad_s0"], rtol=1e-10, atol=1e-12)


GOLD_CL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refcl_*.npz")))


@pytest.mark.parametrize("path", GOLD_CL, ids=[os.path.basename(p)[6:-4] for p in GOLD_CL])
def test_oracle_closed_loop_fd_reproduces_reference(path):
    """cfg.MODEL.POLICY.NETWORK = True (backward.py:418-419,460-461): state-perturbed rollouts run under the policy's
    action at the perturbed state.  Oracle steps + the fixture's policy, assembled as calculate_gradients does."""
    import util
    g = np.load(path)
    o = Oracle(str(g["env"]))
    q, v, c = g["qpos"], g["qvel"], g["ctrl"]
    K, nq, nv, nu, n = q.shape[0], o.nq, o.nv, o.nu, o.n
    cp = g["ctrl_pert"]
    assert np.abs(util.closed_loop_actions_cpu(g) - cp).max() < 1e-5
    eps = 1e-6
    qn, vn, r, st = o.step(q, v, c)
    assert (st == 0).all()
    assert np.array_equal(np.concatenate([qn, vn], 1), g["s_next"]) and np.array_equal(r, g["reward"])
    for k in range(K):
        A = np.zeros((n, n)); rs = np.zeros(n)
        for j in range(nv + nq):
            qq, vv = q[k].copy(), v[k].copy()
            if j < nv:
                vv[j] += eps
            else:
                qq[j - nv] += eps
            q1, v1, r1, _ = o.step(qq, vv, cp[k, j])
            col = nq + j if j < nv else j - nv
            A[:nq, col] = (q1[0] - qn[k]) / eps
            A[nq:, col] = (v1[0] - vn[k]) / eps
            rs[col] = (r1[0] - r[k]) / eps
        assert np.array_equal(A, g["dsds"][k]) and np.array_equal(rs, g["drds"][k])
    # the ctrl columns are the open-loop ones
    ol = o.fd_jacobians(q, v, c)
    assert np.array_equal(ol["dsda"], g["dsda"]) and np.array_equal(ol["drda"], g["drda"])
