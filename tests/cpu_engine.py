"""TEST INFRASTRUCTURE ONLY: a CPU stand-in with FdEngine's interface, built on the host build of the kernel's own
per-lane simulator (tests/_build/libhost_twin.so) and the numpy recurrence.  It exists so that the multi-process host
logic (trainer.do_training under torch.distributed: episode sharding, the gradient all-reduce, rank-0 files) can be
executed end to end with the gloo backend on a machine without a GPU.  The product path never imports it: the package
has no CPU fallback (FdEngine raises without a HIP device)."""
import ctypes as C
import os

import numpy as np
import torch

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_dp = C.POINTER(C.c_double)


def _twin():
    import __graft_entry__
    L = C.CDLL(__graft_entry__.build_host_twin())
    return L


class CpuEngine:
    def __init__(self, env, frame_skip=0):
        import mbrl_amd  # noqa: F401
        from mbrl_amd import config, modelspec
        self.L = _twin()
        self.env, self.eid = env, util.ENVS.index(env)
        spec = modelspec.MODELS[env]
        self.nq = self.nv = sum(len(b["joints"]) for b in spec["bodies"])
        self.nu = len(spec["actuators"])
        self.n = self.nq + self.nv
        self.frame_skip = int(frame_skip) or config.cfg_for_env(env).MODEL.FRAME_SKIP
        self.dt = float(spec["timestep"]) * self.frame_skip
        self.device = torch.device("cpu")

    # ---- numpy core ----
    def _step_np(self, q, v, c):
        q, v, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, c))
        N = q.shape[0]
        qn, vn, r = np.empty_like(q), np.empty_like(v), np.empty(N)
        st = np.zeros(N, dtype=np.uint32)
        self.L.twin_step(self.eid, self.frame_skip, C.c_int64(N), q.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                         c.ctypes.data_as(_dp), qn.ctypes.data_as(_dp), vn.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
                         st.ctypes.data_as(C.POINTER(C.c_uint32)))
        return qn, vn, r, st

    def _t(self, x, dtype=torch.float64):
        return torch.as_tensor(np.ascontiguousarray(x)).to(dtype)

    # ---- FdEngine interface ----
    def step(self, qpos, qvel, ctrl):
        q, v, c = (torch.as_tensor(x, dtype=torch.float64).reshape(-1, w).numpy()
                   for x, w in ((qpos, self.nq), (qvel, self.nv), (ctrl, self.nu)))
        qn, vn, r, st = self._step_np(q, v, c)
        return self._t(qn), self._t(vn), self._t(r), self._t(st.astype(np.int32), torch.int32)

    def step_into(self, qpos, qvel, ctrl, qpos_next, qvel_next, reward, status):
        qn, vn, r, st = self.step(qpos, qvel, ctrl)
        qpos_next.copy_(qn); qvel_next.copy_(vn); reward.copy_(r); status.copy_(st)

    def new_outputs(self, N):
        n, nu = self.n, self.nu
        z = lambda *s: torch.zeros(s, dtype=torch.float64)
        return dict(dsds=z(N, n, n), dsda=z(N, n, nu), drds=z(N, n), drda=z(N, nu), s_next=z(N, n), reward=z(N),
                    status=torch.zeros(N, dtype=torch.int32))

    def fd_jacobians(self, qpos, qvel, ctrl, eps=1e-6, fd_mode=0, nsteps=1, out=None, policy=None, ctrl_pert=None,
                     planned=None, plan=None):
        """calculate_gradients' schedule (backward.py:389-472): forward differences, columns ctrl, qvel, qpos."""
        assert fd_mode == 0 and nsteps == 1 and policy is None and ctrl_pert is None, "stand-in: open-loop parity mode only"
        q, v, c = (torch.as_tensor(x, dtype=torch.float64).reshape(-1, w).numpy()
                   for x, w in ((qpos, self.nq), (qvel, self.nv), (ctrl, self.nu)))
        N, nq, nv, nu, n = q.shape[0], self.nq, self.nv, self.nu, self.n
        R = 1 + nu + nv + nq
        Q, V, Cc = np.repeat(q, R, 0), np.repeat(v, R, 0), np.repeat(c, R, 0)
        for p in range(1, R):
            i = p - 1
            tgt = Cc if i < nu else (V if i < nu + nv else Q)
            tgt[p::R, i if i < nu else (i - nu if i < nu + nv else i - nu - nv)] += eps
        qn, vn, r, st = self._step_np(Q, V, Cc)
        S = np.concatenate([qn, vn], 1).reshape(N, R, n)
        r = r.reshape(N, R)
        dS = (S[:, 1:] - S[:, :1]) / eps                      # [N, NP, n]: row p = d s' / d input p
        dr = (r[:, 1:] - r[:, :1]) / eps
        o = out or self.new_outputs(N)
        o["dsda"].copy_(self._t(dS[:, :nu].transpose(0, 2, 1)))
        o["drda"].copy_(self._t(dr[:, :nu]))
        # state columns in (qpos, qvel) order; the perturbations ran in (qvel, qpos) order
        o["dsds"].copy_(self._t(np.concatenate([dS[:, nu + nv:], dS[:, nu:nu + nv]], 1).transpose(0, 2, 1)))
        o["drds"].copy_(self._t(np.concatenate([dr[:, nu + nv:], dr[:, nu:nu + nv]], 1)))
        o["s_next"].copy_(self._t(S[:, 0])); o["reward"].copy_(self._t(r[:, 0]))
        o["status"].copy_(self._t(np.bitwise_or.reduce(st.reshape(N, R), axis=1).astype(np.int32), torch.int32))
        return o

    def vjp_scan(self, jac, c, grad_weights="average", time_major=False, shard=None):
        """SURVEY.md App. C recurrence; shard = (episode_offset, total_episodes) seeds 'prioritise' as mbrl_vjp_scan_sharded."""
        assert not time_major
        c = torch.as_tensor(c, dtype=torch.float64).numpy()
        B, H = c.shape
        A, Bm = jac["dsds"].numpy().reshape(B, H, self.n, self.n), jac["dsda"].numpy().reshape(B, H, self.n, self.nu)
        rs, ra = jac["drds"].numpy().reshape(B, H, self.n), jac["drda"].numpy().reshape(B, H, self.nu)
        off, tot = shard if shard is not None else (0, B)
        da, ds0 = np.zeros((B, H, self.nu)), np.zeros((B, self.n))
        for b in range(B):
            S = (tot - 1 - (off + b)) * (0.5 * H * (H + 1))
            gv = np.zeros(self.n)
            for t in range(H - 1, -1, -1):
                W = H - t
                wa_r = wa_d = ws_r = 1.0
                if grad_weights == "average":
                    wa_r = wa_d = 1.0 / W
                elif grad_weights == "prioritise":
                    S += W
                    ws_r, wa_r, wa_d = W - 1.0, W / S, 1.0 / S
                da[b, t] = c[b, t] * wa_r * ra[b, t] + wa_d * (gv @ Bm[b, t])
                gv = c[b, t] * ws_r * rs[b, t] + gv @ A[b, t]
            ds0[b] = gv
        return self._t(da), self._t(ds0)
