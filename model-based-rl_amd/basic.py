"""Config-selected graph  policy -> dynamics op -> reward op  (reference: model/archs/basic.py:10-46)."""
import torch
import torch.nn as nn

from .ops import mj_torch_block_factory


class OpenLoopPolicy(nn.Module):
    """Minimal open-loop action sequence [H, nu] indexed by the agent's step counter -- the mean of the
    reference's trajectory-optimisation strategies (strategies.py:402-408) without their sampling."""

    def __init__(self, cfg, agent):
        super().__init__()
        H, nu = cfg.MODEL.POLICY.MAX_HORIZON_STEPS, agent.model.nu
        self.agent = agent
        self.mean = nn.Parameter(torch.full((H, nu), float(cfg.MODEL.POLICY.INITIAL_ACTION_MEAN), dtype=torch.float64))
        self.action_dim = nu

    def forward(self, state):
        a = self.mean[int(self.agent.get_step_idx())]
        return a.expand(state.shape[:-1] + a.shape) if state.dim() > 1 else a


class Basic(nn.Module):
    def __init__(self, cfg, agent, policy_net=None):
        super().__init__()
        self.cfg = cfg
        self.agent = agent
        self.policy_net = policy_net if policy_net is not None else OpenLoopPolicy(cfg, agent)
        self.agent.unwrapped.policy_net = self.policy_net
        self.dynamics_block = mj_torch_block_factory(agent, 'dynamics').apply
        self.reward_block = mj_torch_block_factory(agent, 'reward').apply

    def forward(self, state):
        # the state reaches the policy detached (basic.py:38): no d a / d s path in the graph
        action = self.policy_net(state.detach().float()).double().to(state.device)
        next_state = self.dynamics_block(state, action)
        reward = self.reward_block(state, action)
        return next_state, reward
