"""Shared helpers for the tests: synthetic input sets and tolerances."""
import numpy as np

ENVS = ("InvertedPendulumEnv", "InvertedDoublePendulumEnv", "SwimmerEnv", "HopperEnv", "Walker2dEnv",
        "HalfCheetahEnv")


def reset_like(env, N, seed, nq, nu):
    """Set S1: reset-like states (mirrors each env's reset_model distribution, widened a little
    for the fixed-init envs so that units differ) + uniform controls."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-0.1, 0.1, (N, nq))
    v = 0.1 * rng.standard_normal((N, nq))
    c = rng.uniform(-1, 1, (N, nu))
    if env in ("HopperEnv", "Walker2dEnv"):
        q[:, 1] += 1.25
        q[:, 3:] = -np.abs(q[:, 3:]) * 0.5   # knee/hip ranges are [-150 deg, 0]
    return q, v, c


def on_trajectory(oracle, env, N, seed, max_steps=20):
    """Set S2: S1 states advanced k ~ U{0..max_steps} env steps under clip(N(0, .5^2), -1, 1) controls
    with the ORACLE's step (contact-rich), then a fresh control."""
    rng = np.random.default_rng(seed + 1000)
    q, v, c = reset_like(env, N, seed, oracle.nq, oracle.nu)
    k = rng.integers(0, max_steps + 1, N)
    for t in range(max_steps):
        live = k > t
        if not live.any():
            break
        a = np.clip(rng.normal(0, 0.5, (N, oracle.nu)), -1, 1)
        qn, vn, _, _ = oracle.step(q[live], v[live], a[live], nthreads=8)
        q[live], v[live] = qn, vn
    c = rng.uniform(-1, 1, (N, oracle.nu))
    return q, v, c


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


def fixture_mlp(g, device="cpu"):
    """The float32 Linear -> LeakyReLU -> Linear policy stored in a closed-loop fixture (refcl_*.npz)."""
    import torch
    w0, b0, w1, b1 = (torch.from_numpy(np.asarray(g[k])) for k in ("w0", "b0", "w1", "b1"))
    net = torch.nn.Sequential(torch.nn.Linear(w0.shape[1], w0.shape[0]), torch.nn.LeakyReLU(),
                              torch.nn.Linear(w1.shape[1], w1.shape[0]))
    with torch.no_grad():
        net[0].weight.copy_(w0); net[0].bias.copy_(b0); net[2].weight.copy_(w1); net[2].bias.copy_(b1)
    return net.to(device)


def closed_loop_actions_cpu(g, eps=1e-6):
    """ctrl_pert [K, nv+nq, nu] as backward.py:415-419,457-461 forms it (one state at a time, CPU float32).  Float32
    matmul rounding differs between CPU models, and eps = 1e-6 amplifies one float32 ulp of an action to a few per
    cent of a Jacobian entry: parity tests therefore use the actions RECORDED in the fixture (g["ctrl_pert"]); this
    re-evaluation only has to agree with them to float32 accuracy."""
    import torch
    net = fixture_mlp(g)
    q, v = g["qpos"], g["qvel"]
    K, nq, nv = q.shape[0], q.shape[1], v.shape[1]
    out = np.zeros((K, nv + nq, g["ctrl"].shape[1]))
    for k in range(K):
        for j in range(nv + nq):
            qq, vv = q[k].copy(), v[k].copy()
            if j < nv:
                vv[j] += eps
            else:
                qq[j - nv] += eps
            out[k, j] = net(torch.from_numpy(np.concatenate((qq, vv))).float()).double().detach().numpy()
    return out
