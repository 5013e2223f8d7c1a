"""Agent object with the reference's protocol, backed by device tensors and the HIP path.

Mirrors what MjBlockWrapper(SnapshotWrapper(IndexWrapper(env))) exposes to the op
(mujoco/build.py:23-35, mujoco/utils/wrappers/{mj_block.py:8-70, etc.py:91-184}):
``forward_factory`` / ``gradient_factory``, ``get_snapshot`` / ``set_snapshot``, step / episode /
batch indices, ``data.{qpos,qvel,ctrl}``, ``model.{nq,nv,nu}``, and the result attributes
``reward``, ``is_done``, ``next_state``, ``dynamics_gradients``, ``reward_gradients``, ``running_sum``.
Unlike the reference, every array carries a leading batch dimension (``agent.batch`` episodes advance
together); with batch 1 and 1-D inputs the behaviour is the reference's.
"""
import copy
import types

import numpy as np
import torch

from . import lib, modelspec
from .engine import FdEngine


class Index(object):
    """Mutable integer shared by reference (reference: utils/index.py:1-43)."""

    def __init__(self, value=0):
        self.value = int(value)

    def __index__(self):
        return self.value

    def __int__(self):
        return self.value

    def __iadd__(self, other):
        self.value += int(other)
        return self

    def __add__(self, other):
        if isinstance(other, (Index, int)):
            return Index(self.value + int(other))
        raise NotImplementedError

    def __sub__(self, other):
        if isinstance(other, (Index, int)):
            return Index(self.value - int(other))
        raise NotImplementedError

    def __eq__(self, other):
        return self.value == (other.value if isinstance(other, Index) else other)

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return "Index({})".format(self.value)

    def __str__(self):
        return "{}".format(self.value)

    def set(self, other):
        if isinstance(other, (Index, int)):
            self.value = int(other)
        else:
            raise NotImplementedError


class DataSnapshot:
    """What SnapshotWrapper.get_snapshot keeps (etc.py:95-115) that matters for hinge/slide models."""

    def __init__(self, data, step_idx):
        self.time = data.time
        self.qpos = data.qpos.clone()
        self.qvel = data.qvel.clone()
        self.ctrl = data.ctrl.clone()
        self.step_idx = copy.deepcopy(step_idx)


class Agent:
    def __init__(self, cfg, batch=1, device=None, engine=None, shard=None):
        """shard = (first episode, episodes of the whole batch) when this agent advances one rank's block of a batch
        that is split over several GPUs; default: the agent owns the whole batch."""
        self.cfg = cfg
        self.episode_offset, self.total_batch = (int(shard[0]), int(shard[1])) if shard else (0, int(batch))
        env = cfg.MUJOCO.ENV
        self.engine = engine or FdEngine(env, frame_skip=cfg.MODEL.FRAME_SKIP, device=device)
        e = self.engine
        self.batch = int(batch)
        self.model = types.SimpleNamespace(nq=e.nq, nv=e.nv, nu=e.nu)
        z = lambda w: torch.zeros((self.batch, w), dtype=torch.float64, device=e.device)
        self.data = types.SimpleNamespace(qpos=z(e.nq), qvel=z(e.nv), ctrl=z(e.nu), time=0.0)
        spec = modelspec.MODELS[env]
        self.init_qpos = torch.tensor([j["ref"] for b in spec["bodies"] for j in b["joints"]],
                                      dtype=torch.float64, device=e.device)
        self.init_qvel = torch.zeros(e.nv, dtype=torch.float64, device=e.device)
        self.dt = e.dt
        self._step_idx, self._episode_idx, self._batch_idx = Index(0), Index(0), Index(0)
        self._batch_size = cfg.MODEL.BATCH_SIZE
        self.reward = None
        self.is_done = False
        self.next_state = None
        self.data_snapshot = None
        self.running_sum = 0
        self.dynamics_gradients = None
        self.reward_gradients = None
        self.status = None
        self.policy_net = None
        self.np_random = np.random.default_rng(cfg.MODEL.RANDOM_SEED or None)

    unwrapped = property(lambda self: self)

    # ---- IndexWrapper (etc.py:134-184) ----
    def get_step_idx(self):
        return self._step_idx

    def get_episode_idx(self):
        return self._episode_idx

    def get_batch_idx(self):
        return self._batch_idx

    def set_step_idx(self, idx):
        self._step_idx.set(idx)

    def set_episode_idx(self, idx):
        self._episode_idx.set(idx)

    def set_batch_idx(self, idx):
        self._batch_idx.set(idx)

    # ---- SnapshotWrapper (etc.py:91-131) ----
    def get_snapshot(self):
        return DataSnapshot(self.data, self.get_step_idx())

    def set_snapshot(self, snap):
        self.data.time = snap.time
        self.data.qpos[:] = snap.qpos
        self.data.qvel[:] = snap.qvel
        self.data.ctrl[:] = snap.ctrl
        self.set_step_idx(snap.step_idx)

    # ---- env ----
    def _obs(self):
        o = torch.cat([self.data.qpos, self.data.qvel], dim=-1)
        return o[0] if self.batch == 1 else o

    def reset(self, update_episode_idx=True):
        """IndexWrapper.reset (etc.py:155-166) + reset_model of the env (e.g. half_cheetah.py:42-51)."""
        self._step_idx.set(0)
        if update_episode_idx:
            if self._episode_idx == self._batch_size:
                self._batch_idx += 1
                self._episode_idx.set(1)
            else:
                self._episode_idx += 1
        env, B, e = self.cfg.MUJOCO.ENV, self.batch, self.engine
        q = self.init_qpos.repeat(B, 1)
        v = self.init_qvel.repeat(B, 1)
        if self.cfg.MODEL.POLICY.NETWORK and env in ("HalfCheetahEnv", "HopperEnv", "SwimmerEnv"):
            # Reset noise is drawn for the WHOLE batch from the (identically seeded) generator and sliced, so that a
            # rank of a sharded run starts its episodes where a single-process run starts the same episodes.
            r, T = self.np_random, self.total_batch
            if env == "HalfCheetahEnv":      # half_cheetah.py:45-48
                dq, dv = r.uniform(-.1, .1, (T, e.nq)), r.standard_normal((T, e.nv)) * .1
            elif env == "HopperEnv":         # hopper.py:58-59
                dq, dv = r.uniform(-.005, .005, (T, e.nq)), r.uniform(-.005, .005, (T, e.nv))
            else:                            # swimmer.py:44-45
                dq, dv = r.uniform(-.1, .1, (T, e.nq)), r.uniform(-.1, .1, (T, e.nv))
            sl = slice(self.episode_offset, self.episode_offset + B)
            q = q + torch.as_tensor(dq[sl], device=e.device)
            v = v + torch.as_tensor(dv[sl], device=e.device)
        if env == "InvertedPendulumEnv":     # inverted_pendulum.py:44
            q, v = torch.zeros_like(q), torch.zeros_like(v)
        self.data.qpos[:], self.data.qvel[:] = q, v
        self.data.time = 0.0
        return self._obs()

    def step(self, action):
        """IndexWrapper.step (etc.py:151-153) -> env.step: returns (obs, reward, done=False, info)."""
        self._step_idx += 1
        a = torch.as_tensor(action, dtype=torch.float64, device=self.engine.device).reshape(self.batch, -1)
        self.data.ctrl[:] = a
        qn, vn, r, st = self.engine.step(self.data.qpos, self.data.qvel, a)
        self.data.qpos[:], self.data.qvel[:] = qn, vn
        self.data.time += self.dt
        self.status = st
        return self._obs(), (r[0] if self.batch == 1 else r), False, {}

    # ---- MjBlockWrapper (mj_block.py:14-30) ----
    def forward_factory(self, mode):
        agent = self

        def mj_forward(action=None):
            """mujoco/utils/forward.py:13-38."""
            if action is None:
                action = agent.data.ctrl
            if isinstance(action, torch.Tensor):
                action = action.detach().clone()
            elif isinstance(action, np.ndarray):
                action = torch.from_numpy(action.copy())
            else:
                raise TypeError("Expecting a torch tensor or numpy ndarray")
            assert action.dtype == torch.float64, "You must use dtype numpy.float64 for actions"
            next_state, agent.reward, agent.is_done, _ = agent.step(action)
            return next_state

        return mj_forward

    def gradient_factory(self, mode):
        agent = self

        def mj_gradients(data_snapshot, next_state, reward, test=False):
            """calculate_gradients (mujoco/utils/backward.py:343-480) for the whole batch in one launch."""
            cfg = agent.cfg
            fd_mode = lib.FD_CENTRAL if str(cfg.MODEL.get("FD_MODE", "forward")) == "central" else lib.FD_FORWARD
            out = agent.engine.fd_jacobians(data_snapshot.qpos, data_snapshot.qvel, data_snapshot.ctrl,
                                            eps=float(cfg.MODEL.get("FD_EPS", 1e-6)), fd_mode=fd_mode,
                                            nsteps=int(cfg.MODEL.NSTEPS_FOR_BACKWARD),
                                            # backward.py:418-419,460-461: closed-loop re-query of the policy
                                            policy=agent.policy_net if cfg.MODEL.POLICY.NETWORK else None)
            if test:
                # backward.py:362-374: ONE env step from the snapshot must reproduce the forward pass exactly, whatever
                # NSTEPS_FOR_BACKWARD is.  With one step per FD rollout that re-step is the linearisation's own nominal
                # rollout; with more, its nominal covers nsteps env steps, so the single step is launched on its own.
                if int(cfg.MODEL.NSTEPS_FOR_BACKWARD) == 1:
                    s1, r1 = out["s_next"], out["reward"]
                else:
                    q1, v1, r1, _ = agent.engine.step(data_snapshot.qpos, data_snapshot.qvel, data_snapshot.ctrl)
                    s1 = torch.cat([q1, v1], dim=-1)
                ns = torch.as_tensor(next_state, device=s1.device).reshape(s1.shape)
                rw = torch.as_tensor(reward, device=r1.device).reshape(r1.shape)
                assert torch.equal(rw, r1), \
                    "reward is different from forward pass at timepoint {}".format(data_snapshot.time)
                assert torch.equal(ns, s1), "state is different from forward pass"
            sq = (lambda t: t[0]) if agent.batch == 1 else (lambda t: t)
            agent.dynamics_gradients = {"state": sq(out["dsds"]), "action": sq(out["dsda"])}
            agent.reward_gradients = {"state": sq(out["drds"].unsqueeze(-2)), "action": sq(out["drda"].unsqueeze(-2))}
            agent.status = out["status"]
            return

        return mj_gradients


def build_agent(cfg, batch=1, device=None, engine=None, shard=None):
    """Reference: mujoco/build.py:6-37 (viewer/video wrappers are out of scope).  `engine`: an object with FdEngine's
    interface to use instead of creating one (the multi-process CPU tests pass a stand-in); `shard`: see Agent."""
    if cfg.MUJOCO.ENV not in modelspec.MODELS:
        raise AttributeError("module 'envs' has no attribute %r" % (cfg.MUJOCO.ENV,))
    return Agent(cfg, batch=batch, device=device, engine=engine, shard=shard)
