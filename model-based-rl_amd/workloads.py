"""Synthetic input sets for benchmarking.

S1: reset-like states (HalfCheetah: mujoco/envs/half_cheetah.py:45-48 -- init + U(-.1,.1),
    0.1*N(0,1)) with ctrl ~ U(-1,1).
S2: S1 advanced k ~ U{0..H-1} env steps with this library's own HIP step kernel under
    ctrl_t = clip(N(0, 0.5^2), -1, 1) (INITIAL_SD 0.5, configs/half_cheetah.yaml:20), then a
    fresh control: on-trajectory, contact-rich states.
"""
import numpy as np
import torch


def s1(engine, N, seed):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-0.1, 0.1, (N, engine.nq))
    v = 0.1 * rng.standard_normal((N, engine.nv))
    c = rng.uniform(-1.0, 1.0, (N, engine.nu))
    if engine.env in ("HopperEnv", "Walker2dEnv"):
        q[:, 1] += 1.25
        q[:, 3:] = -np.abs(q[:, 3:]) * 0.5
    dev = engine.device
    return tuple(torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (q, v, c))


def s2(engine, N, seed, horizon=40):
    q, v, c = s1(engine, N, seed)
    g = torch.Generator(device="cpu").manual_seed(seed + 12345)
    k = torch.randint(0, horizon, (N,), generator=g).to(engine.device)
    for t in range(horizon - 1):
        a = torch.clamp(0.5 * torch.randn((N, engine.nu), generator=g, dtype=torch.float64), -1, 1).to(engine.device)
        qn, vn, _, st = engine.step(q, v, a)
        live = ((k > t) & (st == 0)).unsqueeze(1)
        q, v = torch.where(live, qn, q), torch.where(live, vn, v)
    return q.contiguous(), v.contiguous(), c
