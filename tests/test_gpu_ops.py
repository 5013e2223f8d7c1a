"""GPU: the host-side mirror of the reference's op protocol (MjBlock ops, agent, batched rollout + scan) over the
HIP path, against the golden vectors produced by the REFERENCE's own MjBlock + calculate_gradients
(tests/golden/gen_reference_golden.py).  Tolerance: the golden Jacobians come from the CPU oracle's step, the
ones here from the HIP kernels (agreement 2e-6 relative, test_gpu_parity.py); gradients are sums of products of
those over H = 5 steps, so 2e-5 relative to the largest gradient entry."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_*.npz")))
IDS = [os.path.basename(p)[4:-4] for p in GOLD]


def _cfg(g):
    import mbrl_amd
    from mbrl_amd import config
    cfg = config.cfg_for_env(str(g["env"]))
    cfg.MODEL.POLICY.GRAD_WEIGHTS = str(g["weights"])
    cfg.MODEL.POLICY.MAX_HORIZON_STEPS = int(g["H"])
    cfg.MODEL.NSTEPS_FOR_BACKWARD = int(g["nsteps"])
    cfg.MODEL.FRAME_SKIP = int(g["frame_skip"])
    return cfg


def _close(a, b, tol=2e-5):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= tol * (1.0 + np.abs(b).max())


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_ops_reproduce_reference_autograd(path):
    """Episode by episode, op by op, exactly as model/archs/basic.py:29-46 + trainer.py:57-77 drive them."""
    import torch
    import mbrl_amd
    from mbrl_amd.agent import build_agent
    from mbrl_amd.ops import mj_torch_block_factory
    g = np.load(path)
    cfg = _cfg(g)
    agent = build_agent(cfg, batch=1, device="cuda:0")
    dyn = mj_torch_block_factory(agent, "dynamics").apply
    rew = mj_torch_block_factory(agent, "reward").apply
    B, H = g["c"].shape
    acts = torch.tensor(g["actions"], dtype=torch.float64, device="cuda:0", requires_grad=True)
    s0 = torch.tensor(g["states"][:, 0], dtype=torch.float64, device="cuda:0", requires_grad=True)
    loss = 0
    for b in range(B):
        agent.set_step_idx(0)
        s = s0[b]
        for t in range(H):
            s_next = dyn(s, acts[b, t])
            r = rew(s, acts[b, t])
            assert r.shape == (1,)
            # forward values: s' bit-comparable to 1e-9 with the oracle-generated trajectory, reward fp32-rounded
            assert _close(s_next.detach().cpu().numpy(), g["states"][b, t + 1], 1e-8)
            assert abs(float(r.detach()) - g["rewards_op"][b, t]) <= 1e-6 * (1 + abs(g["rewards_op"][b, t]))
            s = s_next
            loss = loss + float(g["c"][b, t]) * r.sum()
    agent.running_sum = 0           # trainer.py:76
    loss.backward()
    assert _close(acts.grad.cpu().numpy(), g["grad_actions"])
    assert _close(s0.grad.cpu().numpy(), g["grad_s0"])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_batched_rollout_and_scan_reproduce_reference_autograd(path):
    """The batched replacement: H step launches, ONE FD launch over B*H units, one scan kernel."""
    import torch
    import mbrl_amd
    from mbrl_amd import rollout
    from mbrl_amd.engine import FdEngine
    g = np.load(path)
    if int(g["nsteps"]) != 1:
        pytest.skip("the scan composes one-step Jacobians")
    eng = FdEngine(str(g["env"]), frame_skip=int(g["frame_skip"]))
    acts = torch.tensor(g["actions"], dtype=torch.float64, device="cuda")
    states, rewards, status = rollout.rollout(eng, g["states"][:, 0], acts)
    assert int(status.abs().sum()) == 0
    assert _close(states.cpu().numpy(), g["states"], 1e-8)
    assert _close(rewards.cpu().numpy(), g["rewards"], 1e-8)
    da, ds0, jac = rollout.action_gradients(eng, states, acts, g["c"], grad_weights=str(g["weights"]))
    assert _close(jac["dsds"].cpu().numpy().reshape(g["dsds"].shape), g["dsds"], 2e-6)
    assert _close(da.cpu().numpy(), g["grad_actions"])
    assert _close(ds0.cpu().numpy(), g["grad_s0"])


def test_basic_graph_runs_batched():
    """Basic.forward (policy -> dynamics op -> reward op) with a leading batch dimension."""
    import torch
    import mbrl_amd
    from mbrl_amd import config
    from mbrl_amd.agent import build_agent
    from mbrl_amd.basic import Basic
    cfg = config.cfg_for_env("HalfCheetahEnv")
    cfg.MODEL.POLICY.MAX_HORIZON_STEPS = 3
    agent = build_agent(cfg, batch=16, device="cuda:0")
    model = Basic(cfg, agent)
    s = agent.reset()
    total = 0
    for t in range(3):
        s, r = model(s)
        assert s.shape == (16, 18) and r.shape == (16, 1)
        total = total + r.sum()
    agent.running_sum = 0
    total.backward()
    gr = model.policy_net.mean.grad
    assert gr.shape == (3, 6) and torch.isfinite(gr).all() and float(gr.abs().max()) > 0


GOLD_CL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "refcl_*.npz")))


@pytest.mark.parametrize("path", GOLD_CL, ids=[os.path.basename(p)[6:-4] for p in GOLD_CL])
def test_closed_loop_fd_matches_reference(path):
    """Closed-loop linearisation (cfg.MODEL.POLICY.NETWORK, backward.py:418-419,460-461) through the C ABI's
    mbrl_fd_jacobians_closed_loop against Jacobians the reference's calculate_gradients produced.  The policy's
    actions are the ones the reference's FD loop obtained (recorded in the fixture) and handed over as ctrl_pert:
    with eps = 1e-6 a float32 policy makes the state columns depend on the last bit of the matmul, so the kernel
    is pinned separately from the GEMM.  Tolerance 2e-6 relative to the unit's largest entry, as in test_gpu_parity."""
    import torch
    import util
    import mbrl_amd
    from mbrl_amd.engine import FdEngine
    g = np.load(path)
    eng = FdEngine(str(g["env"]))
    cp = g["ctrl_pert"]
    out = eng.fd_jacobians(g["qpos"], g["qvel"], g["ctrl"], ctrl_pert=cp)
    assert int(out["status"].abs().sum()) == 0
    assert _close(out["s_next"].cpu().numpy(), g["s_next"], 1e-9)
    for k in ("dsds", "dsda", "drds", "drda"):
        a, b = out[k].cpu().numpy().reshape(len(cp), -1), g[k].reshape(len(cp), -1)
        err = np.abs(a - b) / (1.0 + np.abs(b).max(axis=1, keepdims=True))
        assert err.max() < 2e-6, (k, float(err.max()))
    # and the convenience path: policy evaluated on the GPU (a batched GEMM), same layout of ctrl_pert
    net = util.fixture_mlp(g, "cuda")
    cp_gpu = eng.perturbed_policy_actions(eng._in(g["qpos"], eng.nq), eng._in(g["qvel"], eng.nv), net)
    assert cp_gpu.shape == cp.shape
    assert np.abs(cp_gpu.cpu().numpy() - cp).max() < 1e-5       # float32 GEMM vs float32 CPU loop
    out2 = eng.fd_jacobians(g["qpos"], g["qvel"], g["ctrl"], policy=net)
    out3 = eng.fd_jacobians(g["qpos"], g["qvel"], g["ctrl"], ctrl_pert=cp_gpu)
    assert torch.equal(out2["dsds"], out3["dsds"])
    # the perturbed float32 states handed to the policy are those of the plain formula float32(s + eps e_j), bit for bit
    seen = []
    spy = lambda x: (seen.append(x.clone()), net(x))[1]
    q64, v64 = eng._in(g["qpos"], eng.nq), eng._in(g["qvel"], eng.nv)
    s64 = torch.cat([q64, v64], dim=1)
    colj = [eng.nq + j for j in range(eng.nv)] + list(range(eng.nq))
    for mode, signs in ((0, (1.0,)), (1, (1.0, -1.0))):
        seen.clear()
        eng.perturbed_policy_actions(q64, v64, spy, 1e-6, mode)
        want = torch.stack([torch.stack([(s64 + sg * 1e-6 * torch.nn.functional.one_hot(torch.tensor(cj), eng.n).to(s64))
                                         for sg in signs], dim=1) for cj in colj], dim=1)     # [N, n, len(signs), n]
        assert torch.equal(seen[0], want.reshape(-1, eng.n).float())
    # central differences take [N, nv+nq, 2, nu]
    cpc = eng.perturbed_policy_actions(eng._in(g["qpos"], eng.nq), eng._in(g["qvel"], eng.nv), net, fd_mode=1)
    assert cpc.shape == (len(cp), eng.n, 2, eng.nu)
    outc = eng.fd_jacobians(g["qpos"], g["qvel"], g["ctrl"], fd_mode=1, ctrl_pert=cpc)
    assert torch.isfinite(outc["dsds"]).all()
    assert torch.equal(outc["dsda"], eng.fd_jacobians(g["qpos"], g["qvel"], g["ctrl"], fd_mode=1)["dsda"])
