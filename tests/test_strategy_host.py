"""CPU: host logic of the strategy / optimiser mirror (reference: model/blocks/policy/strategies.py:129-467,
solver/build.py:4-25) -- no GPU, no kernel: a stand-in agent provides dimensions and the shared step index."""
import types

import numpy as np
import pytest
import torch

import mbrl_amd
from mbrl_amd import config
from mbrl_amd.agent import Index
from mbrl_amd.strategies import VariationalOptimization, build_optimizer, build_policy, FeedForward


def fake_agent(nq=2, nv=2, nu=1, batch=1):
    step, ep = Index(0), Index(1)
    return types.SimpleNamespace(model=types.SimpleNamespace(nq=nq, nv=nv, nu=nu), batch=batch,
                                 engine=types.SimpleNamespace(device=torch.device("cpu")),
                                 get_step_idx=lambda: step, get_episode_idx=lambda: ep)


def cfg_vo(B=4, H=5, method="R", sd=0.3, network=False):
    cfg = config.cfg_for_env("InvertedPendulumEnv")
    cfg.MODEL.BATCH_SIZE, cfg.MODEL.POLICY.MAX_HORIZON_STEPS = B, H
    cfg.MODEL.POLICY.ARCH, cfg.MODEL.POLICY.METHOD = "VariationalOptimization", method
    cfg.MODEL.POLICY.INITIAL_SD, cfg.MODEL.POLICY.NETWORK, cfg.MODEL.POLICY.LAYERS = sd, network, [8]
    cfg.MODEL.RANDOM_SEED = 3
    return cfg


def test_build_optimizer_groups_and_lr_carry_over():
    cfg = config.get_cfg_defaults()
    cfg.SOLVER.BASE_LR, cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.STD_LR_FACTOR = 0.01, 2, 0.5
    p = lambda: torch.nn.Parameter(torch.zeros(2))
    named = [("mean.linear_layer_0.weight", p()), ("mean.linear_layer_0.bias", p()), ("mean.linear_layer_1.weight", p()),
             ("sd", p())]
    opt = build_optimizer(cfg, named)
    assert isinstance(opt, torch.optim.Adam)
    # solver/build.py:6-19: lr is not reset inside the loop, so the weight AFTER a bias inherits the bias rate
    assert [g["lr"] for g in opt.param_groups] == [0.01, 0.02, 0.02, 0.005]
    cfg.SOLVER.OPTIMIZER = "sgd"
    assert isinstance(build_optimizer(cfg, named), torch.optim.SGD)


def test_R_loss_matches_reference_formula_and_zeroes_nans():
    cfg = cfg_vo()
    pol = VariationalOptimization(cfg, fake_agent())
    bl = torch.tensor(np.random.default_rng(0).standard_normal((4, 5)))
    bl[1, 3:] = float("nan")                       # an episode that ended early (trainer.py:54-55,73-74)
    loss, stats = pol.R_loss(bl.clone())
    z = torch.where(torch.isnan(bl), torch.zeros_like(bl), bl)
    assert float(loss) == pytest.approx(float(z.mean(dim=0).sum()))           # strategies.py:191-204
    assert stats["objective_loss"] == pytest.approx(float(z.sum(dim=1).mean()))  # :224


def test_forward_open_loop_batched_and_unbatched_agree_with_reference_semantics():
    cfg = cfg_vo(B=1, sd=0.3)
    ag = fake_agent()
    pol = VariationalOptimization(cfg, ag)
    ag.get_step_idx().set(2)
    a = pol(torch.zeros(4))                          # BATCH_SIZE 1: action = mean (strategies.py:452-453)
    assert torch.equal(a, pol.mean[:, 2])
    assert pol.clamped_action[:, 2, 0] == pytest.approx(a.detach().numpy())
    pol.eval()
    assert torch.equal(pol(torch.zeros(3, 4)), pol.mean[:, 2].expand(3, -1))   # testing mode: mean (:438-439)


def test_sharded_sampling_reproduces_the_full_batch():
    """Ranks draw the whole batch's noise from identically seeded generators and slice it."""
    cfg = cfg_vo(B=8)
    torch.manual_seed(0); np.random.seed(0)
    full = VariationalOptimization(cfg, fake_agent(batch=8))
    np.random.seed(0)
    lo = VariationalOptimization(cfg, fake_agent(batch=3), shard=(0, 3))
    np.random.seed(0)
    hi = VariationalOptimization(cfg, fake_agent(batch=5), shard=(3, 8))
    for t in range(3):
        for pol, n in ((full, 8), (lo, 3), (hi, 5)):
            pol.step_idx.set(t)
        a = full(torch.zeros(8, 4)); b = lo(torch.zeros(3, 4)); c = hi(torch.zeros(5, 4))
        assert torch.equal(a, torch.cat([b, c]))
    assert torch.equal(full.log_prob[:3, :3], lo.log_prob[:, :3])


def test_sharded_optimize_equals_full_batch_step():
    """Sum over shards of (local_batch / B) * local R-loss gradients == full-batch gradient; one Adam step agrees."""
    cfg = cfg_vo(B=6, H=4)
    pols = []
    for shard in ((0, 6), (0, 2), (2, 6)):
        np.random.seed(1)
        pols.append(VariationalOptimization(cfg, fake_agent(batch=shard[1] - shard[0]), shard=shard))
    full, a, b = pols
    w = torch.tensor(np.random.default_rng(5).standard_normal((6, 4)))

    def batch_loss(pol, rows):
        out = []
        for t in range(4):
            pol.step_idx.set(t)
            act = pol(torch.zeros(rows.stop - rows.start, 4))
            out.append((act ** 2).sum(dim=1) * w[rows, t])        # any differentiable function of the actions
        return torch.stack(out, dim=1)

    full.optimize(batch_loss(full, slice(0, 6)))
    grads = {}

    def collect(name):
        def sync(gl):
            grads[name] = [g.clone() for g in gl]
        return sync
    a.optimize(batch_loss(a, slice(0, 2)), grad_sync=collect("a"))
    b.optimize(batch_loss(b, slice(2, 6)), grad_sync=collect("b"))
    # what the all-reduce would hand every rank
    for ga, gb, p in zip(grads["a"], grads["b"], full.parameters()):
        assert torch.allclose(ga + gb, p.grad, rtol=1e-12, atol=1e-14)


def test_closed_loop_policy_mlp_and_state_dict_names():
    cfg = cfg_vo(network=True)
    pol = VariationalOptimization(cfg, fake_agent())
    keys = set(pol.state_dict().keys())
    assert {"sd", "mean.sequential.linear_layer_0.weight", "mean.sequential.linear_layer_1.bias"} <= keys
    s = torch.randn(7, 4)
    assert pol.mean_action(s.float()).shape == (7, 1) and pol.mean_action(s.float()).dtype == torch.float64
    assert isinstance(pol.mean, FeedForward)


def test_build_policy_names():
    cfg = cfg_vo()
    assert isinstance(build_policy(cfg, fake_agent()), VariationalOptimization)
    cfg.MODEL.POLICY.ARCH = "Perttu"
    with pytest.raises(NotImplementedError):
        build_policy(cfg, fake_agent())
    cfg.MODEL.POLICY.ARCH = "Nope"
    with pytest.raises(AttributeError):
        build_policy(cfg, fake_agent())


def strategy_step(shard, grad_sync):
    """One optimisation step of VariationalOptimization on episodes [lo, hi) of a fixed synthetic batch of 6; returns
    the parameters after the step (used by the gloo world-size-2 test as well)."""
    cfg = cfg_vo(B=6, H=4)
    np.random.seed(1)
    pol = VariationalOptimization(cfg, fake_agent(batch=shard[1] - shard[0]), shard=shard)
    w = torch.tensor(np.random.default_rng(5).standard_normal((6, 4)))
    rows = slice(shard[0], shard[1])
    out = []
    for t in range(4):
        pol.step_idx.set(t)
        act = pol(torch.zeros(shard[1] - shard[0], 4))
        out.append((act ** 2).sum(dim=1) * w[rows, t])
    pol.optimize(torch.stack(out, dim=1), grad_sync=grad_sync)
    return {k: v.detach().clone() for k, v in pol.state_dict().items()}
