"""The in-repo trajectory / policy strategy that uses the FD gradients, batched on the GPU
(reference: model/blocks/policy/strategies.py:12-467, solver/build.py:4-25,
model/layers/feed_forward.py:25-62).

`VariationalOptimization` keeps the reference's parameters, names and arithmetic -- `mean` ([nu, H] tensor or an
MLP when cfg.MODEL.POLICY.NETWORK), `sd` [nu, H], Normal(mean, sd).rsample() when BATCH_SIZE > 1, the loss
variants R / PR / IR / SIR, the per-parameter-group Adam / SGD of `build_optimizer`, the sd floor of
`standard_optimize` -- but one call of `forward` serves ALL episodes of the batch ([B, n] -> [B, nu]) instead of
one episode.  With batch 1 and a 1-D state it behaves as the reference does.  The configured default strategy of
the reference's yamls (`Perttu`) needs a private module that is absent from the reference tree
(strategies.py:9) and is out of scope; `CMAES` does not use the gradients.
"""
import numpy as np
import torch
from torch import nn
from torch.nn.parameter import Parameter


def build_optimizer(cfg, named_parameters):
    """solver/build.py:4-25, including its carry-over: `lr` is not reset between parameters, so a parameter
    that follows a "bias" / "sd" one inherits that learning rate (no effect when both factors are 1, as in the
    reference's yamls)."""
    params = []
    lr = cfg.SOLVER.BASE_LR
    for key, value in named_parameters:
        if not value.requires_grad:
            continue
        if "bias" in key:
            lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
        if any(x in key for x in ["std", "sd"]):
            lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.STD_LR_FACTOR
        params += [{"params": [value], "lr": lr}]
    if cfg.SOLVER.OPTIMIZER == "sgd":
        return torch.optim.SGD(params, lr)
    return torch.optim.Adam(params, lr, betas=tuple(cfg.SOLVER.ADAM_BETAS))


class FeedForward(nn.Module):
    """Linear / LeakyReLU stack with the reference's layer names (feed_forward.py:25-62), so that a
    `state_dict` saved by either side loads in the other (`sequential.linear_layer_i.*`)."""

    def __init__(self, input_size, middle_sizes, output_size):
        super().__init__()
        sizes = [input_size] + list(middle_sizes) + [output_size]
        layers = []
        for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
            layers.append(("linear_layer_%d" % i, nn.Linear(a, b, bias=True)))
            if i < len(sizes) - 2:
                layers.append(("activation_layer_%d" % i, nn.LeakyReLU()))
        from collections import OrderedDict
        self.sequential = nn.Sequential(OrderedDict(layers))

    def forward(self, x):
        return self.sequential(x)


class VariationalOptimization(nn.Module):
    def __init__(self, cfg, agent, reinforce_loss_weight=1.0, min_reinforce_loss_weight=0.0, min_sd=0, shard=None):
        """`shard` = (lo, hi): the episodes of the global batch this process owns (multi-GPU trainer); default all."""
        super().__init__()
        self.cfg = cfg
        P = cfg.MODEL.POLICY
        self.method = P.METHOD
        self.action_dim, self.state_dim = agent.model.nu, agent.model.nq + agent.model.nv
        self.horizon = P.MAX_HORIZON_STEPS
        self.dim = (self.action_dim, self.horizon)
        self.batch_size = cfg.MODEL.BATCH_SIZE           # global: decides whether actions are sampled (:450)
        self.shard = tuple(shard) if shard is not None else (0, self.batch_size)
        self.local_batch = self.shard[1] - self.shard[0]
        self.device = agent.engine.device
        # exploration noise is drawn for the WHOLE batch on every rank from an identically seeded generator and
        # sliced, so a sharded run samples exactly what a 1-GPU run samples
        self.noise_gen = torch.Generator(device=self.device)
        self.noise_gen.manual_seed(int(cfg.MODEL.RANDOM_SEED) + 104729)
        self.min_sd = min_sd
        self.sd_threshold = 0.0001                       # strategies.py:38
        self.gamma = P.GAMMA
        self.reinforce_loss_weight = reinforce_loss_weight
        self.min_reinforce_loss_weight = min_reinforce_loss_weight
        self.eps = np.finfo(np.float32).eps.item()
        self.loss_functions = {"IR": self.IR_loss, "PR": self.PR_loss, "SIR": self.SIR_loss, "R": self.R_loss}
        self.step_idx = agent.get_step_idx()             # shared by reference (strategies.py:56-57)
        self.episode_idx = agent.get_episode_idx()
        self.agent_batch = agent.batch

        if P.NETWORK:                                    # strategies.py:393-399
            self.mean = FeedForward(self.state_dim, P.LAYERS, self.action_dim).to(self.device)
        else:                                            # strategies.py:402-405, initialise_mean :116-121
            m = np.asarray(np.random.normal(P.INITIAL_ACTION_MEAN, P.INITIAL_ACTION_SD, self.dim), dtype=np.float64)
            self.mean = Parameter(torch.from_numpy(m).to(self.device))
        self.sd = Parameter(torch.full(self.dim, float(P.INITIAL_SD), dtype=torch.float64, device=self.device))
        self.initial_clamped_sd = self.clamp_sd(self.sd.detach().clone())
        # the record of the last epoch's actions lives on the device ([action_dim, horizon, batch], the reference's
        # layout); `clamped_action` copies it to the host when somebody looks (no transfer inside the step loop)
        self._clamped_action = torch.zeros((self.action_dim, self.horizon, self.local_batch), dtype=torch.float64,
                                           device=self.device)
        self._sd_positive, self._sd_version = None, -1
        self.optimizer = build_optimizer(cfg, self.named_parameters())
        self.log_prob = torch.zeros(self.local_batch, self.horizon, dtype=torch.float64, device=self.device)

    # ---- reference helpers (strategies.py:83-114) ----
    def clamp_sd(self, sd):
        return sd

    def clamp_action(self, action):
        return action

    def get_clamped_sd(self):
        return self.clamp_sd(self.sd).detach().cpu().numpy()

    @property
    def clamped_action(self):
        return self._clamped_action.cpu().numpy()

    def get_clamped_action(self):
        return self.clamped_action

    def _sd_is_positive(self, t):
        """strategies.py:441: noise is drawn only where the standard deviation is positive.  sd changes once per
        optimisation step, so the test is made for all t at once when the parameter has changed (one host sync per
        epoch instead of one per env step)."""
        if self._sd_version != self.sd._version:
            self._sd_positive = (self.clamp_sd(self.sd).detach().min(dim=0).values > 0).tolist()
            self._sd_version = self.sd._version
        return self._sd_positive[t]

    def mean_action(self, state):
        """Deterministic action of a closed-loop policy (what the FD re-query of backward.py:418-419 evaluates)."""
        if not self.cfg.MODEL.POLICY.NETWORK:
            raise TypeError("mean_action(state) needs cfg.MODEL.POLICY.NETWORK")
        return self.clamp_action(self.mean(state).double())

    # ---- forward (strategies.py:427-464), all episodes of the agent's batch at once ----
    def forward(self, state):
        t = int(self.step_idx)
        clamped_sd = self.clamp_sd(self.sd[:, t])
        batched = state.dim() > 1
        if self.cfg.MODEL.POLICY.NETWORK:
            mean = self.mean(state).double()
        else:
            mean = self.mean[:, t]
            if batched:
                mean = mean.expand(state.shape[0], -1)
        if not self.training:
            return self.clamp_action(mean)
        dist = torch.distributions.Normal(mean, clamped_sd, validate_args=False) if self._sd_is_positive(t) else None
        if self.batch_size > 1 and dist is not None:
            if batched:      # = dist.rsample(): mean + sd * N(0, 1), with the noise of the global batch sliced to this shard
                z = torch.randn((self.batch_size, self.action_dim), dtype=torch.float64, device=self.device,
                                generator=self.noise_gen)[self.shard[0]:self.shard[1]]
                action = mean + clamped_sd * z[:mean.shape[0]]
            else:
                action = dist.rsample()
        else:
            action = self.clamp_action(mean)
        a = action.detach()
        if batched:
            self._clamped_action[:, t, :a.shape[0]] = a.T
            if dist is not None:
                self.log_prob[:a.shape[0], t] = dist.log_prob(a).sum(dim=-1)
        else:
            e = int(self.episode_idx) - 1
            self._clamped_action[:, t, e] = a
            if dist is not None:
                self.log_prob[e, t] = dist.log_prob(a).sum()
        return action

    def forward_open_loop(self, horizon, batch):
        """All `horizon` forward() calls of one batched training rollout at once, for an open-loop policy (the actions
        do not depend on the states): the same noise draws in the same order, the same arithmetic per element, the same
        side effects (`_clamped_action`, `log_prob`) -- but a dozen kernels per epoch instead of a dozen per env step,
        so the rollout's step launches run back to back.  Returns actions [horizon, batch, action_dim] (with their graph
        to `mean` and `sd`), or None when the per-step path has to be taken (closed-loop policy, evaluation mode, a
        batch of one, a non-positive standard deviation somewhere)."""
        if self.cfg.MODEL.POLICY.NETWORK or not self.training or self.batch_size <= 1:
            return None
        if horizon != self.horizon or batch != self.local_batch:
            return None
        if not all(self._sd_is_positive(t) for t in range(horizon)):
            return None
        z = torch.stack([torch.randn((self.batch_size, self.action_dim), dtype=torch.float64, device=self.device,
                                     generator=self.noise_gen)[self.shard[0]:self.shard[1]][:batch]
                         for _ in range(horizon)])                                      # [H, B, nu], forward()'s draws
        mean = self.mean.T[:, None, :].expand(horizon, batch, self.action_dim)          # mean[:, t] per step
        sd = self.clamp_sd(self.sd).T[:, None, :]                                       # clamp_sd(sd[:, t]) per step
        action = mean + sd * z
        a = action.detach()
        self._clamped_action[:, :, :batch] = a.permute(2, 0, 1)
        dist = torch.distributions.Normal(mean, sd, validate_args=False)
        self.log_prob[:batch, :] = dist.log_prob(a).sum(dim=-1).T
        return action

    # ---- losses (strategies.py:129-280) ----
    def calculate_reinforce_loss(self, batch_loss, stepwise_loss=False):
        returns = torch.empty_like(batch_loss)
        R = torch.zeros(batch_loss.shape[0], dtype=torch.float64, device=batch_loss.device)
        for k in range(1, batch_loss.shape[1] + 1):
            R = -batch_loss[:, -k].detach() + self.gamma * R
            returns[:, -k] = R
        advantages = ((returns - returns.mean(dim=0)) / (returns.std(dim=0) + self.eps)).detach()
        if stepwise_loss:
            return (-advantages * self.log_prob).mean(dim=0)
        return torch.mean(torch.sum(-advantages * self.log_prob, dim=1))

    def calculate_objective_loss(self, batch_loss, stepwise_loss=False):
        batch_loss = torch.where(torch.isnan(batch_loss), torch.zeros_like(batch_loss), batch_loss)   # :191-192
        if stepwise_loss:
            return torch.mean(batch_loss, dim=0)
        return torch.sum(torch.mean(batch_loss, dim=0))                                                # :204

    @staticmethod
    def _objective_stat(batch_loss):
        b = torch.where(torch.isnan(batch_loss), torch.zeros_like(batch_loss), batch_loss)
        return float(torch.mean(torch.sum(b, dim=1)).detach())

    def R_loss(self, batch_loss):
        objective_loss = self.calculate_objective_loss(batch_loss)
        return objective_loss, {"objective_loss": self._objective_stat(batch_loss),
                                "total_loss": float(objective_loss.detach())}

    def PR_loss(self, batch_loss):
        reinforce_loss = self.calculate_reinforce_loss(batch_loss)
        objective_loss = self.calculate_objective_loss(batch_loss)
        loss = objective_loss + self.reinforce_loss_weight * reinforce_loss
        return loss, {"objective_loss": self._objective_stat(batch_loss),
                      "reinforce_loss": float(reinforce_loss.detach()), "total_loss": float(loss.detach())}

    def _mix(self, elementwise):
        clamped_sd = self.clamp_sd(self.sd)
        mix = 0.5 * ((clamped_sd - self.min_sd) / (self.initial_clamped_sd - self.min_sd)).detach()
        if not elementwise:
            mix = mix.mean()
        mix = self.min_reinforce_loss_weight + (1.0 - self.min_reinforce_loss_weight) * mix
        return torch.clamp(mix, 0.0, 1.0)

    def IR_loss(self, batch_loss):
        reinforce_loss = self.calculate_reinforce_loss(batch_loss)
        objective_loss = self.calculate_objective_loss(batch_loss)
        m = self._mix(False)
        loss = (1.0 - m) * objective_loss + m * reinforce_loss
        return loss, {"objective_loss": self._objective_stat(batch_loss),
                      "reinforce_loss": float(reinforce_loss.detach()), "total_loss": float(loss.detach())}

    def SIR_loss(self, batch_loss):
        reinforce_loss = self.calculate_reinforce_loss(batch_loss, stepwise_loss=True)
        objective_loss = self.calculate_objective_loss(batch_loss, stepwise_loss=True)
        m = self._mix(True)                        # [nu, H] against [H]: broadcast, then summed over both (:273-277)
        loss = ((1.0 - m) * objective_loss + m * reinforce_loss).sum()
        return loss, {"objective_loss": self._objective_stat(batch_loss),
                      "reinforce_loss": float(reinforce_loss.sum().detach()), "total_loss": float(loss.detach())}

    # ---- optimise (strategies.py:311-364, 466-467) ----
    def optimize(self, batch_loss, grad_sync=None):
        """`batch_loss` [local_batch, H].  `grad_sync`: optional callable run between backward and the optimiser step
        (the multi-GPU trainer passes the flattened all-reduce of parallel.py; the reference is single-process).
        Sharded over G ranks the loss of a shard is weighted local_batch / BATCH_SIZE so that the summed gradient
        is the 1-GPU gradient for the objective loss `R` (batch mean of per-step losses, :204); the REINFORCE
        baselines of PR / IR / SIR are then per-shard statistics."""
        loss, stats = self.loss_functions.get(self.method, self.R_loss)(batch_loss)
        if self.local_batch != self.batch_size:
            loss = loss * (float(self.local_batch) / float(self.batch_size))
        self.optimizer.zero_grad()
        loss.backward()
        if grad_sync is not None:
            grad_sync([p.grad for p in self.parameters() if p.grad is not None])
        if self.cfg.SOLVER.OPTIMIZER == "sgd":
            nn.utils.clip_grad_value_(self.parameters(), 1)
        self.optimizer.step()
        with torch.no_grad():                      # "Make sure sd is not negative" :357-359
            self.sd.data.clamp_(min=self.sd_threshold)
        self.log_prob = torch.zeros(self.local_batch, self.horizon, dtype=torch.float64, device=self.device)
        return stats


def build_policy(cfg, agent, shard=None):
    """model/blocks/policy/build.py:4-10: class looked up by cfg.MODEL.POLICY.ARCH."""
    arch = cfg.MODEL.POLICY.ARCH
    if arch == "VariationalOptimization":
        return VariationalOptimization(cfg, agent, shard=shard)
    if arch in ("Perttu", "CMAES"):
        raise NotImplementedError("POLICY.ARCH %r is outside the FD path (Perttu needs the reference's absent private "
                                  "`optimizer` module, strategies.py:9); use VariationalOptimization" % arch)
    raise AttributeError("no policy strategy named %r" % (arch,))
