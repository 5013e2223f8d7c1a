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
