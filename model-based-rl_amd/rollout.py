"""Batched replacement of the reference's episode/step loops + loss.backward()
(model/engine/trainer.py:53-77).

The reference walks B episodes x H steps one env.step at a time and lets autograd call the FD
procedure B*H times.  Here: H launches of the step kernel advance all B episodes together, ONE
launch of the FD kernel linearises all B*H stored (s_t, a_t), and one scan kernel runs the backward
recurrence of the op chain.
"""
import torch

from . import lib


def _time_major_buffers(engine, s0, H):
    """Q [H+1, B, nq], V [H+1, B, nv], R [H, B], S [H, B]: step t reads slice t and writes slice t+1 in place, so the
    loop allocates and copies nothing."""
    s0 = torch.as_tensor(s0, dtype=torch.float64, device=engine.device)
    B, nq = s0.shape[0], engine.nq
    Q = torch.empty((H + 1, B, nq), dtype=torch.float64, device=engine.device)
    V = torch.empty((H + 1, B, engine.nv), dtype=torch.float64, device=engine.device)
    R = torch.empty((H, B), dtype=torch.float64, device=engine.device)
    S = torch.empty((H, B), dtype=torch.int32, device=engine.device)
    Q[0], V[0] = s0[:, :nq], s0[:, nq:]
    return Q, V, R, S


def _episode_major(Q, V, R, S):
    return torch.cat([Q, V], dim=2).permute(1, 0, 2).contiguous(), R.t().contiguous(), S.t().contiguous()


def rollout(engine, s0, actions):
    """s0 [B, n], actions [B, H, nu] -> states [B, H+1, n], rewards [B, H], status [B, H]."""
    B, H, _ = actions.shape
    Q, V, R, S = _time_major_buffers(engine, s0, H)
    A = torch.as_tensor(actions, dtype=torch.float64, device=engine.device).permute(1, 0, 2).contiguous()   # [H, B, nu]
    for t in range(H):
        engine.step_into(Q[t], V[t], A[t], Q[t + 1], V[t + 1], R[t], S[t])
    return _episode_major(Q, V, R, S)


def action_gradients(engine, states, actions, c, grad_weights="average", eps=1e-6, fd_mode=lib.FD_FORWARD):
    """dL/d actions [B, H, nu] and dL/d s0 [B, n] for L = sum_{b,t} c[b,t] * r[b,t], with the op
    weighting of model/blocks/mujoco.py:46-76.  One FD launch over B*H units + one scan."""
    B, H, nu = actions.shape
    nq = engine.nq
    s = states[:, :H].reshape(B * H, engine.n)
    jac = engine.fd_jacobians(s[:, :nq], s[:, nq:], actions.reshape(B * H, nu), eps=eps, fd_mode=fd_mode)
    da, ds0 = engine.vjp_scan(jac, c, grad_weights)
    return da, ds0, jac


class RolloutRewards(torch.autograd.Function):
    """rewards [B, H] of a stored rollout as a differentiable function of (actions [B, H, nu], s0 [B, n]).

    Forward hands back the rewards the step kernel produced; backward is the whole chain of MjBlock.backward
    calls autograd would issue for the epoch (model/blocks/mujoco.py:43-104) in two launches: ONE FD launch over
    the B*H stored (s_t, a_t) and one scan.  Closed-loop mode passes `policy` (cfg.MODEL.POLICY.NETWORK,
    backward.py:418-419,460-461).  `grad_weights` is cfg.MODEL.POLICY.GRAD_WEIGHTS."""

    @staticmethod
    def forward(ctx, actions, s0, engine, states, rewards, grad_weights, eps, fd_mode, policy, raise_on_status):
        ctx.engine, ctx.gw, ctx.eps, ctx.fd_mode, ctx.policy, ctx.chk = engine, grad_weights, eps, fd_mode, policy, raise_on_status
        ctx.save_for_backward(actions.detach(), states)
        # the reference's reward op rounds through fp32 (torch.Tensor([r]).double(), mujoco.py:38); kept
        return rewards.float().double()

    @staticmethod
    def backward(ctx, c):
        actions, states = ctx.saved_tensors
        eng = ctx.engine
        B, H, nu = actions.shape
        s = states[:, :H].reshape(B * H, eng.n)
        jac = eng.fd_jacobians(s[:, :eng.nq], s[:, eng.nq:], actions.reshape(B * H, nu), eps=ctx.eps, fd_mode=ctx.fd_mode,
                               policy=ctx.policy)
        if ctx.chk and int((jac["status"] != 0).sum()):
            raise lib.MbrlError("FD linearisation flagged %d of %d units (status bits, include/mbrl.h)"
                                % (int((jac["status"] != 0).sum()), B * H))
        da, ds0 = eng.vjp_scan(jac, c.contiguous(), ctx.gw)
        return da, ds0, None, None, None, None, None, None, None, None


def rollout_with_policy(engine, agent, model_policy, s0, horizon):
    """Forward pass of one epoch for all episodes together (trainer.py:57-74 + basic.py:29-46): the policy sees
    the DETACHED state (basic.py:38), the step kernel advances every episode.  Returns states [B, H+1, n],
    actions [B, H, nu] (with their autograd graph to the policy parameters), rewards [B, H], status [B, H]."""
    Q, V, R, S = _time_major_buffers(engine, s0, horizon)
    acts = []
    for t in range(horizon):
        agent.set_step_idx(t)
        a = model_policy(torch.cat([Q[t], V[t]], dim=1).float()).double()
        acts.append(a)
        engine.step_into(Q[t], V[t], a.detach().contiguous(), Q[t + 1], V[t + 1], R[t], S[t])
    states, rewards, status = _episode_major(Q, V, R, S)
    agent.set_step_idx(horizon)
    return states, torch.stack(acts, dim=1), rewards, status
