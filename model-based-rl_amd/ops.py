"""The `MuJoCo` / `Reward` autograd ops (reference: model/blocks/mujoco.py:6-106).

Same factory, same `forward(ctx, state, action)` / `backward(ctx, grad_output) -> (ds, da)`
signatures and the same hidden coupling (the reward op must run right after the dynamics op of
the same step and returns what that op parked on the agent).  state: [..., nq+nv] fp64,
action: [..., nu] fp64 on the agent's GPU; leading batch dims are this build's extension.
"""
import torch
from torch import autograd


def mj_torch_block_factory(agent, mode):
    mj_forward = agent.forward_factory(mode)
    mj_gradients = agent.gradient_factory(mode)
    nq = agent.model.nq

    class MjBlock(autograd.Function):

        @staticmethod
        def forward(ctx, state, action):
            if mode == "dynamics":
                s = state.detach().reshape(agent.batch, -1)
                agent.data.qpos[:] = s[:, :nq]
                agent.data.qvel[:] = s[:, nq:]
                agent.data.ctrl[:] = action.detach().reshape(agent.batch, -1)
                agent.data_snapshot = agent.get_snapshot()
                next_state = mj_forward()
                agent.next_state = next_state
                ctx.data_snapshot = agent.data_snapshot
                return next_state.clone().reshape(state.shape)
            elif mode == "reward":
                ctx.data_snapshot = agent.data_snapshot
                ctx.reward = agent.reward
                ctx.next_state = agent.next_state
                # torch.Tensor([reward]).double() (mujoco.py:38) rounds the value through fp32; kept
                r = torch.as_tensor(agent.reward, dtype=torch.float64, device=state.device).float().double()
                return r.reshape(state.shape[:-1] + (1,))
            else:
                raise TypeError("mode has to be 'dynamics' or 'gradient'")

        @staticmethod
        def backward(ctx, grad_output):
            P = agent.cfg.MODEL.POLICY
            if P.GRAD_WEIGHTS == "prioritise":
                weight = P.MAX_HORIZON_STEPS - ctx.data_snapshot.step_idx.value
            else:
                weight = 1 / (P.MAX_HORIZON_STEPS - ctx.data_snapshot.step_idx.value)

            # 'prioritise': the reference visits episodes one after the other (last to first) and lets
            # agent.running_sum carry over between them (mujoco.py:65); when B episodes advance together the sum
            # episode b has seen is the scalar accumulated over its own steps plus what the later episodes added
            # -- of the WHOLE batch: under sharding (agent.episode_offset, agent.total_batch) the later episodes of the
            # other ranks count too, so that a sharded run weights exactly like a single-process run
            def running_sum():
                B, H = agent.batch, P.MAX_HORIZON_STEPS
                first, total = getattr(agent, "episode_offset", 0), getattr(agent, "total_batch", None) or B
                if total == 1:
                    return agent.running_sum
                later = total - 1 - first - torch.arange(B, dtype=torch.float64, device=grad_output.device)
                return (agent.running_sum + later * (0.5 * H * (H + 1))).reshape(B, 1, 1)

            if mode == "dynamics":
                state_jacobian = agent.dynamics_gradients["state"]
                action_jacobian = agent.dynamics_gradients["action"]
                if P.GRAD_WEIGHTS == "prioritise":
                    action_jacobian = (1.0 / running_sum()) * action_jacobian
                elif P.GRAD_WEIGHTS == "average":
                    action_jacobian = weight * action_jacobian
            elif mode == "reward":
                if P.GRAD_WEIGHTS == "prioritise":
                    agent.running_sum += weight
                # "reward" is always called first (mujoco.py:67-68)
                mj_gradients(ctx.data_snapshot, ctx.next_state, ctx.reward, test=True)
                state_jacobian = agent.reward_gradients["state"]
                action_jacobian = agent.reward_gradients["action"]
                if P.GRAD_WEIGHTS == "prioritise":
                    state_jacobian = (weight - 1) * state_jacobian
                    action_jacobian = (weight / running_sum()) * action_jacobian
                elif P.GRAD_WEIGHTS == "average":
                    action_jacobian = weight * action_jacobian
            else:
                raise TypeError("mode has to be 'dynamics' or 'reward'")

            g = grad_output.unsqueeze(-2)
            ds = torch.matmul(g, state_jacobian).squeeze(-2)
            da = torch.matmul(g, action_jacobian).squeeze(-2)
            return ds, da

    return MjBlock
