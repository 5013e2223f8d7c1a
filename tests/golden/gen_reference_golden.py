#!/usr/bin/env python3
"""Generate golden vectors with the REFERENCE's own Python (run in the build container only).

Loads, by file path, the reference's op and FD procedure
    /root/reference/model/blocks/mujoco.py      (MjBlock forward/backward, weights)
    /root/reference/mujoco/utils/backward.py    (calculate_gradients)
    /root/reference/mujoco/utils/forward.py     (mj_forward)
    /root/reference/utils/index.py              (Index)
and drives them over a duck-typed agent whose `step` is this repo's CPU oracle
(oracle/mjc_oracle.c through ctypes).  mujoco_py / gym are absent offline; backward.py only
touches `mujoco_py.const.JNT_BALL/JNT_FREE` on the hinge/slide path, so a constants-only module
object stands in for the import.  This pins the FD schedule (eps, column order, layout,
test=True re-step), the autograd chain rule and the GRAD_WEIGHTS variants against reference code,
independently of the physics.  Output: tests/golden/ref_<env>_<weights>_n<nsteps>.npz
(data only: inputs and expected outputs).
"""
import copy
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle import Oracle  # noqa: E402


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


mj_stub = types.ModuleType("mujoco_py")
mj_stub.const = types.SimpleNamespace(JNT_FREE=0, JNT_BALL=1, JNT_SLIDE=2, JNT_HINGE=3)
mj_stub.functions = types.SimpleNamespace()
sys.modules["mujoco_py"] = mj_stub

ref_index = load("ref_index", os.path.join(REF, "utils/index.py"))
ref_backward = load("ref_backward", os.path.join(REF, "mujoco/utils/backward.py"))
ref_forward = load("ref_forward", os.path.join(REF, "mujoco/utils/forward.py"))
ref_block = load("ref_block", os.path.join(REF, "model/blocks/mujoco.py"))
Index = ref_index.Index


class NS(types.SimpleNamespace):
    pass


class FakeAgent:
    """Implements exactly the attributes/methods the reference's op + FD code touch."""

    def __init__(self, env, H, weights, nsteps):
        self.o = Oracle(env)
        o = self.o
        self.cfg = NS(MODEL=NS(NSTEPS_FOR_BACKWARD=nsteps,
                               POLICY=NS(NETWORK=False, GRAD_WEIGHTS=weights, MAX_HORIZON_STEPS=H)))
        self.model = NS(nq=o.nq, nv=o.nv, nu=o.nu, dof_jntid=np.arange(o.nv), jnt_type=np.full(o.nv, 3),
                        jnt_qposadr=np.arange(o.nv), jnt_dofadr=np.arange(o.nv))
        self.data = NS(qpos=np.zeros(o.nq), qvel=np.zeros(o.nv), ctrl=np.zeros(o.nu), time=0.0)
        self._step_idx = Index(0)
        self.running_sum = 0
        self.calls = []

    # wrappers/mj_block.py:14-70
    def forward_factory(self, mode):
        return ref_forward.mj_forward_factory(self, mode)

    def gradient_factory(self, mode):
        return ref_backward.mj_gradients_factory(self, mode)

    def forward_wrapper(self, mode):
        return lambda fn: fn

    def gradient_wrapper(self, mode):
        def deco(fn):
            def w(*a, **k):
                fn(*a, **k)
            return w
        return deco

    # wrappers/etc.py:93-131, 151-153
    def get_snapshot(self):
        return NS(time=self.data.time, qpos=self.data.qpos.copy(), qvel=self.data.qvel.copy(),
                  ctrl=self.data.ctrl.copy(), step_idx=copy.deepcopy(self._step_idx))

    def set_snapshot(self, s):
        self.data.time = s.time
        self.data.qpos[:] = s.qpos
        self.data.qvel[:] = s.qvel
        self.data.ctrl[:] = s.ctrl
        self._step_idx.set(s.step_idx)

    def step(self, action):
        self._step_idx += 1
        a = np.array(action, dtype=np.float64)
        self.data.ctrl[:] = a
        q, v, r, st = self.o.step(self.data.qpos, self.data.qvel, a)
        assert st[0] == 0
        self.data.qpos[:] = q[0]
        self.data.qvel[:] = v[0]
        return np.concatenate([q[0], v[0]]), float(r[0]), False, {}


def run(env, weights, nsteps, H, seed, B=2):
    rng = np.random.default_rng(seed)
    agent = FakeAgent(env, H, weights, nsteps)
    o = agent.o
    dyn = ref_block.mj_torch_block_factory(agent, "dynamics").apply
    rew = ref_block.mj_torch_block_factory(agent, "reward").apply
    orig = ref_backward.calculate_gradients
    rec = []

    def spy(ag, snap, next_state, reward, test=False):
        orig(ag, snap, next_state, reward, test=test)
        rec.append(dict(t=int(snap.step_idx.value), qpos=snap.qpos.copy(), qvel=snap.qvel.copy(),
                        ctrl=snap.ctrl.copy(),
                        A=ag.dynamics_gradients["state"].copy(), B=ag.dynamics_gradients["action"].copy(),
                        rs=ag.reward_gradients["state"].copy(), ra=ag.reward_gradients["action"].copy()))
    ref_backward.calculate_gradients = spy
    q0 = rng.uniform(-0.05, 0.05, (B, o.nq))
    v0 = 0.1 * rng.standard_normal((B, o.nv))
    if env in ("HopperEnv", "Walker2dEnv"):
        q0[:, 1] += 1.25
        q0[:, 3:] = -np.abs(q0[:, 3:])
    acts = torch.tensor(rng.uniform(-0.8, 0.8, (B, H, o.nu)), dtype=torch.float64, requires_grad=True)
    cw = rng.uniform(0.5, 1.5, (B, H))
    s0 = torch.tensor(np.concatenate([q0, v0], 1), dtype=torch.float64, requires_grad=True)
    loss = 0
    states, rewards, rewards_op = np.zeros((B, H + 1, o.n)), np.zeros((B, H)), np.zeros((B, H))
    for b in range(B):
        agent._step_idx.set(0)
        s = s0[b]
        states[b, 0] = s.detach().numpy()
        for t in range(H):
            s_next = dyn(s, acts[b, t])      # basic.py:41
            r = rew(s, acts[b, t])           # basic.py:44 -- the reward op is fed s_t, not s_{t+1}
            s = s_next
            states[b, t + 1] = s.detach().numpy()
            rewards[b, t] = agent.reward     # fp64 env reward (forward.py:36)
            rewards_op[b, t] = float(r.detach().reshape(-1)[0])   # op output: fp32-rounded (mujoco.py:38)
            loss = loss + float(cw[b, t]) * r.sum()
    agent.running_sum = 0       # trainer.py:76
    loss.backward()
    ref_backward.calculate_gradients = orig
    # order of calculate_gradients calls as autograd issued them
    order = np.array([[r["t"]] for r in rec])
    n, nu = o.n, o.nu
    A = np.zeros((B, H, n, n)); Bm = np.zeros((B, H, n, nu)); rs = np.zeros((B, H, n)); ra = np.zeros((B, H, nu))
    # map records back to (b, t) through the snapshot state
    for r in rec:
        sn = np.concatenate([r["qpos"], r["qvel"]])
        hit = [(b, r["t"]) for b in range(B) if np.array_equal(states[b, r["t"]], sn)]
        assert len(hit) == 1, hit
        b, t = hit[0]
        A[b, t], Bm[b, t], rs[b, t], ra[b, t] = r["A"], r["B"], r["rs"][0], r["ra"][0]
    out = dict(env=env, weights=weights, nsteps=nsteps, H=H, q0=q0, v0=v0, actions=acts.detach().numpy(), c=cw,
               states=states, rewards=rewards, rewards_op=rewards_op, dsds=A, dsda=Bm, drds=rs, drda=ra,
               grad_actions=acts.grad.numpy(), grad_s0=s0.grad.numpy(), call_order=order,
               frame_skip=o.frame_skip)
    path = os.path.join(HERE, "ref_%s_%s_n%d.npz" % (env, weights, nsteps))
    np.savez_compressed(path, **out)
    print("wrote", os.path.basename(path), "calls", len(rec), "|grad|", float(np.abs(out["grad_actions"]).max()))


def mlp_from_arrays(w0, b0, w1, b1):
    """The closed-loop policy of the fixture: Linear -> LeakyReLU -> Linear in float32 (the shape of the
    reference's FeedForward mean network, model/layers/feed_forward.py:25-62)."""
    net = torch.nn.Sequential(torch.nn.Linear(w0.shape[1], w0.shape[0]), torch.nn.LeakyReLU(),
                              torch.nn.Linear(w1.shape[1], w1.shape[0]))
    with torch.no_grad():
        net[0].weight.copy_(torch.from_numpy(w0)); net[0].bias.copy_(torch.from_numpy(b0))
        net[2].weight.copy_(torch.from_numpy(w1)); net[2].bias.copy_(torch.from_numpy(b1))
    return net


def run_closed_loop(env, seed, K=6, hidden=16):
    """cfg.MODEL.POLICY.NETWORK = True: calculate_gradients re-queries agent.policy_net on every state-perturbed
    copy (backward.py:418-419,460-461).  K on-trajectory snapshots, Jacobians as the reference computes them."""
    rng = np.random.default_rng(seed)
    agent = FakeAgent(env, 40, "average", 1)
    agent.cfg.MODEL.POLICY.NETWORK = True
    o = agent.o
    w0 = (0.3 * rng.standard_normal((hidden, o.n))).astype(np.float32); b0 = (0.1 * rng.standard_normal(hidden)).astype(np.float32)
    w1 = (0.3 * rng.standard_normal((o.nu, hidden))).astype(np.float32); b1 = (0.1 * rng.standard_normal(o.nu)).astype(np.float32)
    agent.policy_net = mlp_from_arrays(w0, b0, w1, b1)
    q = rng.uniform(-0.05, 0.05, o.nq); v = 0.1 * rng.standard_normal(o.nv)
    if env in ("HopperEnv", "Walker2dEnv"):
        q[1] += 1.25; q[3:] = -np.abs(q[3:])
    n, nu = o.n, o.nu
    out = dict(env=env, w0=w0, b0=b0, w1=w1, b1=b1, qpos=np.zeros((K, o.nq)), qvel=np.zeros((K, o.nv)), ctrl=np.zeros((K, nu)),
               dsds=np.zeros((K, n, n)), dsda=np.zeros((K, n, nu)), drds=np.zeros((K, n)), drda=np.zeros((K, nu)),
               s_next=np.zeros((K, n)), reward=np.zeros(K), ctrl_pert=np.zeros((K, n, nu)))
    net, log = agent.policy_net, []

    def recording_policy(x):     # the actions the reference's FD loop obtained, in its own order (qvel, then qpos)
        y = net(x)
        log.append(y.double().detach().numpy().copy())
        return y
    agent.data.qpos[:] = q; agent.data.qvel[:] = v
    for k in range(K):
        for _ in range(3):      # advance a few env steps under the policy between snapshots
            a = agent.policy_net(torch.from_numpy(np.concatenate((agent.data.qpos, agent.data.qvel))).float()).double().detach().numpy()
            agent.step(np.clip(a, -1, 1))
        a = agent.policy_net(torch.from_numpy(np.concatenate((agent.data.qpos, agent.data.qvel))).float()).double().detach().numpy()
        agent.data.ctrl[:] = a
        snap = agent.get_snapshot()
        ns, r, _, _ = agent.step(agent.data.ctrl.copy())
        del log[:]
        agent.policy_net = recording_policy
        ref_backward.calculate_gradients(agent, snap, ns, r, test=True)
        agent.policy_net = net
        assert len(log) == n
        out["ctrl_pert"][k] = np.stack(log)
        out["qpos"][k], out["qvel"][k], out["ctrl"][k] = snap.qpos, snap.qvel, snap.ctrl
        out["dsds"][k], out["dsda"][k] = agent.dynamics_gradients["state"], agent.dynamics_gradients["action"]
        out["drds"][k], out["drda"][k] = agent.reward_gradients["state"][0], agent.reward_gradients["action"][0]
        out["s_next"][k], out["reward"][k] = ns, r
        agent.set_snapshot(snap)
        agent.step(np.clip(snap.ctrl, -1, 1))
    path = os.path.join(HERE, "refcl_%s.npz" % env)
    np.savez_compressed(path, **out)
    print("wrote", os.path.basename(path), "|A|", float(np.abs(out["dsds"]).max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "closed_loop":
        for env in ("HalfCheetahEnv", "SwimmerEnv", "InvertedDoublePendulumEnv"):
            run_closed_loop(env, seed=21)
        sys.exit(0)
    H = 5
    for env in ("InvertedPendulumEnv", "InvertedDoublePendulumEnv", "SwimmerEnv", "HopperEnv", "Walker2dEnv",
                "HalfCheetahEnv"):
        run(env, "average", 1, H, seed=11)
    run("HalfCheetahEnv", "prioritise", 1, H, seed=12)
    run("HalfCheetahEnv", "none", 1, H, seed=13)
    run("InvertedDoublePendulumEnv", "prioritise", 1, H, seed=14)
    run("SwimmerEnv", "average", 2, H, seed=15)
