"""GPU: the reference's OWN test file, /root/reference/tests/test_gradients.py, restated over the HIP path through the
reference's agent protocol (forward_factory / gradient_factory / get_snapshot / set_snapshot) and its autograd ops.

  test_reward_gradients   (:12-41)   first-order Taylor check of the reward, InvertedPendulum, 5 decimal places
  test_dynamics_gradients (:43-75)   the same for the next state, Swimmer, np.allclose (rtol 1e-5, atol 1e-8)
  test_action_rewards     (:77-149)  100 action scales sigma in linspace(1e-2, 1e1, 100) -- far beyond ctrlrange -- on
                                     Swimmer: two runs from the same initial state give the SAME reward (==), a re-step
                                     from the snapshot gives the same reward (==), and calculate_gradients(test=True)
                                     passes its own bit-exactness asserts (backward.py:371,374)

The reference's first two tests call a gradient signature that no longer exists (test_gradients.py:29,60 vs
backward.py:495); they are restated against the live one, mj_gradients(data_snapshot, next_state, reward, test).  The
reference file holds no numbers: these are self-consistency checks, here as there."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NWARMUP = 5          # test_gradients.py:19,50,102


def _agent(env, batch=1):
    import mbrl_amd  # noqa: F401
    from mbrl_amd import config
    from mbrl_amd.agent import build_agent
    cfg = config.cfg_for_env(env)
    return cfg, build_agent(cfg, batch=batch)


def _sample_action(rng, agent, batch):
    hi = 3.0 if agent.cfg.MUJOCO.ENV == "InvertedPendulumEnv" else 1.0      # action_space = actuator ctrlrange
    a = rng.uniform(-hi, hi, (batch, agent.model.nu))
    return a[0] if batch == 1 else a


def _linearise_after_warmup(env, batch, seed):
    """Warm up 5 steps under random actions, then linearise at (state, action) with the agent's gradient function and
    evaluate f at (state, action) + eps * 1: everything the two Taylor tests need."""
    import torch
    cfg, agent = _agent(env, batch)
    rng = np.random.default_rng(seed)
    fwd, grad = agent.forward_factory("dynamics"), agent.gradient_factory("reward")
    agent.reset()
    for _ in range(NWARMUP):
        agent.step(torch.as_tensor(_sample_action(rng, agent, batch)))
    action = torch.as_tensor(_sample_action(rng, agent, batch), device=agent.engine.device).reshape(batch, -1)
    agent.data.ctrl[:] = action
    snap = agent.get_snapshot()
    s = fwd().reshape(batch, -1).clone()
    r = torch.as_tensor(agent.reward).reshape(batch).clone()
    grad(snap, s if batch > 1 else s[0], agent.reward, test=True)           # its asserts: re-step == forward pass, bit for bit
    nq = agent.model.nq
    eps = 1e-6
    agent.set_snapshot(snap)
    agent.data.qpos[:] = snap.qpos + eps
    agent.data.qvel[:] = snap.qvel + eps
    agent.data.ctrl[:] = snap.ctrl + eps
    s_prime = fwd().reshape(batch, -1).clone()
    r_prime = torch.as_tensor(agent.reward).reshape(batch).clone()
    J = {k: (v if batch > 1 else v.unsqueeze(0)) for k, v in (("dsds", agent.dynamics_gradients["state"]),
                                                             ("dsda", agent.dynamics_gradients["action"]),
                                                             ("drds", agent.reward_gradients["state"]),
                                                             ("drda", agent.reward_gradients["action"]))}
    n, nu = 2 * nq, agent.model.nu
    ones_s = torch.full((n, 1), eps, dtype=torch.float64, device=s.device)
    ones_a = torch.full((nu, 1), eps, dtype=torch.float64, device=s.device)
    s_est = s + (J["dsds"] @ ones_s).squeeze(-1) + (J["dsda"] @ ones_a).squeeze(-1)
    r_est = r + (J["drds"] @ ones_s).reshape(batch) + (J["drda"] @ ones_a).reshape(batch)
    return (x.cpu().numpy() for x in (s_prime, s_est, r_prime, r_est))


@pytest.mark.parametrize("batch", [1, 64])
def test_reward_gradients(batch):
    """test_gradients.py:12-41 (InvertedPendulum): r(x + eps 1) = r(x) + dr/ds eps 1 + dr/da eps 1 to 5 places."""
    _, _, r_prime, r_est = _linearise_after_warmup("InvertedPendulumEnv", batch, seed=0)
    assert np.abs(np.round(r_prime - r_est, 5)).max() == 0.0        # assertAlmostEqual(places=5)


@pytest.mark.parametrize("batch", [1, 64])
def test_dynamics_gradients(batch):
    """test_gradients.py:43-75 (Swimmer): np.allclose(s(x + eps 1), s(x) + ds/ds eps 1 + ds/da eps 1)."""
    s_prime, s_est, _, _ = _linearise_after_warmup("SwimmerEnv", batch, seed=1)
    assert np.allclose(s_prime, s_est)                               # rtol 1e-5, atol 1e-8 (numpy defaults, as there)


@pytest.mark.parametrize("env", ["HalfCheetahEnv", "HopperEnv", "Walker2dEnv", "InvertedDoublePendulumEnv"])
def test_taylor_consistency_other_envs(env):
    """The same first-order check on the envs the reference's tests do not visit.  With contacts a perturbation of
    eps in EVERY input can flip a constraint that no single-input perturbation flips, so a small share of units may
    miss the bound; the reference's own scheme (one-sided differences through a non-smooth step) has that property."""
    s_prime, s_est, r_prime, r_est = _linearise_after_warmup(env, 256, seed=2)
    ok = np.isclose(s_prime, s_est).all(axis=1) & (np.abs(r_prime - r_est) < 5e-6)
    assert ok.mean() >= 0.97, ok.mean()


def _run_reward_test(sigma, batch, seed, through_ops):
    """run_reward_test (test_gradients.py:87-149); `sigma` [batch]: one action scale per episode of the batch."""
    import torch
    cfg, agent = _agent("SwimmerEnv", batch)
    dev = agent.engine.device
    fwd, grad = agent.forward_factory("dynamics"), agent.gradient_factory("reward")
    agent.reset()
    init_qpos, init_qvel = agent.data.qpos.clone(), agent.data.qvel.clone()
    rng = np.random.default_rng(seed)
    action = torch.as_tensor(rng.standard_normal((batch, agent.model.nu)) * np.asarray(sigma).reshape(batch, 1), device=dev)
    for _ in range(NWARMUP):                              # first simulation
        fwd(action)
    agent.data.ctrl[:] = action
    data = agent.get_snapshot()
    fwd(action)
    reward = torch.as_tensor(agent.reward).clone()
    next_state = torch.cat([agent.data.qpos, agent.data.qvel], dim=-1).clone()
    agent.reset()                                          # second simulation, mj_forward() without arguments
    agent.data.qpos[:], agent.data.qvel[:] = init_qpos, init_qvel
    for _ in range(NWARMUP):
        agent.data.ctrl[:] = action
        fwd()
    agent.data.ctrl[:] = action
    fwd()
    assert torch.equal(torch.as_tensor(agent.reward), reward), "Simulations from same initial state diverged"
    agent.set_snapshot(data)                               # re-step from the snapshot
    fwd()
    assert torch.equal(torch.as_tensor(agent.reward), reward), "Simulation from snapshot diverged"
    if not through_ops:
        grad(data, next_state if batch > 1 else next_state[0], reward, test=True)     # asserts inside (backward.py:371,374)
    else:
        # the same through the autograd ops (model/blocks/mujoco.py): MuJoCo op, Reward op, backward -> the Reward
        # op's backward calls mj_gradients(..., test=True) with what the forward pass stored
        from mbrl_amd.ops import mj_torch_block_factory
        dyn, rew = mj_torch_block_factory(agent, "dynamics"), mj_torch_block_factory(agent, "reward")
        agent.set_snapshot(data)
        s = torch.cat([data.qpos, data.qvel], dim=-1).clone().requires_grad_(True)
        a = action.clone().requires_grad_(True)
        if batch == 1:
            s, a = s[0].detach().requires_grad_(True), a[0].detach().requires_grad_(True)
        ns = dyn.apply(s, a)
        r = rew.apply(s, a)
        assert torch.equal(ns.reshape(next_state.shape), next_state)
        r.sum().backward()
        assert bool(torch.isfinite(a.grad).all()) and bool(torch.isfinite(s.grad).all())
    assert int((agent.status != 0).sum()) == 0


SIGMAS = np.linspace(1e-2, 1e1, 100)                       # test_gradients.py:80-81


@pytest.mark.parametrize("through_ops", [False, True])
def test_action_rewards(through_ops):
    """test_gradients.py:77-85: all 100 action scales, one per episode of a 100-episode batch."""
    _run_reward_test(SIGMAS, 100, seed=0, through_ops=through_ops)


@pytest.mark.parametrize("sigma", [SIGMAS[0], SIGMAS[37], SIGMAS[-1]])
def test_action_rewards_unbatched(sigma):
    """...and one episode at a time, exactly as the reference runs it (1-D state and action through the ops)."""
    _run_reward_test([sigma], 1, seed=int(sigma * 100), through_ops=True)
