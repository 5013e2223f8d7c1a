"""Batched replacement of the reference's episode/step loops + loss.backward()
(model/engine/trainer.py:53-77).

The reference walks B episodes x H steps one env.step at a time and lets autograd call the FD
procedure B*H times.  Here: H launches of the step kernel advance all B episodes together, ONE
launch of the FD kernel linearises all B*H stored (s_t, a_t), and one scan kernel runs the backward
recurrence of the op chain.
"""
import torch

from . import lib


def rollout(engine, s0, actions):
    """s0 [B, n], actions [B, H, nu] -> states [B, H+1, n], rewards [B, H], status [B, H]."""
    B, H, _ = actions.shape
    nq = engine.nq
    states = torch.empty((B, H + 1, engine.n), dtype=torch.float64, device=engine.device)
    rewards = torch.empty((B, H), dtype=torch.float64, device=engine.device)
    status = torch.empty((B, H), dtype=torch.int32, device=engine.device)
    states[:, 0] = torch.as_tensor(s0, dtype=torch.float64, device=engine.device)
    for t in range(H):
        s = states[:, t]
        qn, vn, r, st = engine.step(s[:, :nq], s[:, nq:], actions[:, t])
        states[:, t + 1, :nq], states[:, t + 1, nq:] = qn, vn
        rewards[:, t], status[:, t] = r, st
    return states, rewards, status


def action_gradients(engine, states, actions, c, grad_weights="average", eps=1e-6, fd_mode=lib.FD_FORWARD):
    """dL/d actions [B, H, nu] and dL/d s0 [B, n] for L = sum_{b,t} c[b,t] * r[b,t], with the op
    weighting of model/blocks/mujoco.py:46-76.  One FD launch over B*H units + one scan."""
    B, H, nu = actions.shape
    nq = engine.nq
    s = states[:, :H].reshape(B * H, engine.n)
    jac = engine.fd_jacobians(s[:, :nq], s[:, nq:], actions.reshape(B * H, nu), eps=eps, fd_mode=fd_mode)
    da, ds0 = engine.vjp_scan(jac, c, grad_weights)
    return da, ds0, jac
