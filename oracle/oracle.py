"""ctypes wrapper around the CPU oracle (oracle/_build/libmjc_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmjc_oracle.so")
MAXEFC = 2 * 9 + 4 * 24

_lib = None
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
up = C.POINTER(C.c_uint32)


def build(force=False):
    """Compile the oracle with gcc (needs only oracle/*.c,*.h)."""
    if force or not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.mjo_model_name.restype = C.c_char_p
        L.mjo_model_id.argtypes = [C.c_char_p]
        L.mjo_dims.argtypes = [C.c_int, ip, ip, ip, dp, ip]
        L.mjo_set_option.argtypes = [C.c_int, C.c_char_p, C.c_double]
        L.mjo_body_consts.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
        L.mjo_dof_invweight.argtypes = [C.c_int, dp]
        L.mjo_forward.argtypes = [C.c_int] + [dp] * 10 + [ip] + [dp] * 5 + [ip, dp, dp]
        L.mjo_step.argtypes = [C.c_int, C.c_int, C.c_int64, dp, dp, dp, dp, dp, dp, up, C.c_int]
        L.mjo_fd_jacobians.argtypes = [C.c_int, C.c_int, C.c_int64, dp, dp, dp, C.c_double, C.c_int,
                                       C.c_int, dp, dp, dp, dp, dp, dp, up, C.c_int]
        L.mjo_fd_jacobians_info.argtypes = L.mjo_fd_jacobians.argtypes[:-1] + [ip, ip, C.c_int]
        L.mjo_stats.argtypes = [dp, dp, dp]
        # mbrl_cpu.h: the oracle behind the call shapes of include/mbrl.h
        vp, i64 = C.c_void_p, C.c_int64
        L.mbrl_last_error_cpu.restype = C.c_char_p
        L.mbrl_model_create_cpu.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        L.mbrl_model_destroy_cpu.argtypes = [vp]
        L.mbrl_model_destroy_cpu.restype = None
        L.mbrl_model_dims_cpu.argtypes = [vp, ip, ip, ip, dp]
        L.mbrl_step_cpu.argtypes = [vp, i64] + [vp] * 8
        L.mbrl_fd_jacobians_cpu.argtypes = [vp, i64, vp, vp, vp, C.c_double, C.c_int, C.c_int] + [vp] * 8
        L.mbrl_vjp_scan_cpu.argtypes = [vp, i64, C.c_int, C.c_int] + [vp] * 8
        L.mbrl_vjp_scan_layout_cpu.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int] + [vp] * 8
        L.mbrl_vjp_scan_sharded_cpu.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int, i64, i64] + [vp] * 8
        _lib = L
    return _lib


CPU_TWIN_SYMBOLS = ["mbrl_abi_version_cpu", "mbrl_model_create_cpu", "mbrl_model_destroy_cpu", "mbrl_model_dims_cpu",
                    "mbrl_step_cpu", "mbrl_fd_jacobians_cpu", "mbrl_vjp_scan_cpu", "mbrl_vjp_scan_layout_cpu",
                    "mbrl_vjp_scan_sharded_cpu", "mbrl_last_error_cpu"]


def _p(a):
    return a.ctypes.data_as(dp)


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Oracle:
    def __init__(self, env):
        self.L = lib()
        self.env = env
        self.id = self.L.mjo_model_id(env.encode())
        if self.id < 0:
            raise KeyError(env)
        nq, nv, nu, fs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ts = C.c_double()
        if self.L.mjo_dims(self.id, nq, nv, nu, ts, fs):
            raise RuntimeError("oracle: model compile failed for " + env)
        self.nq, self.nv, self.nu = nq.value, nv.value, nu.value
        self.n = self.nq + self.nv
        self.timestep, self.frame_skip = ts.value, fs.value

    def set_option(self, key, value=0.0):
        rc = self.L.mjo_set_option(self.id, key.encode(), float(value))
        if rc:
            raise RuntimeError("mjo_set_option(%s) -> %d" % (key, rc))

    def body_consts(self, body):
        mass = C.c_double()
        ipos, inertia, iw = np.zeros(3), np.zeros(9), np.zeros(2)
        if self.L.mjo_body_consts(self.id, body, mass, _p(ipos), _p(inertia), _p(iw)):
            raise IndexError(body)
        return mass.value, ipos, inertia.reshape(3, 3), iw

    def dof_invweight(self):
        out = np.zeros(self.nv)
        self.L.mjo_dof_invweight(self.id, _p(out))
        return out

    def forward(self, qpos, qvel, ctrl):
        nv = self.nv
        qpos, qvel, ctrl = _c(qpos), _c(qvel), _c(ctrl)
        o = dict(qacc=np.zeros(nv), qacc_smooth=np.zeros(nv), M=np.zeros((nv, nv)), qfrc_bias=np.zeros(nv),
                 qfrc_passive=np.zeros(nv), qfrc_actuator=np.zeros(nv), qfrc_constraint=np.zeros(nv))
        J = np.zeros((MAXEFC, nv))
        pos, aref, D, force = (np.zeros(MAXEFC) for _ in range(4))
        nefc, niter = C.c_int(), C.c_int()
        en, mom = np.zeros(2), np.zeros(6)
        rc = self.L.mjo_forward(self.id, _p(qpos), _p(qvel), _p(ctrl), _p(o["qacc"]), _p(o["qacc_smooth"]),
                                _p(o["M"]), _p(o["qfrc_bias"]), _p(o["qfrc_passive"]), _p(o["qfrc_actuator"]),
                                _p(o["qfrc_constraint"]), nefc, _p(J), _p(pos), _p(aref), _p(D), _p(force),
                                niter, _p(en), _p(mom))
        k = nefc.value
        o.update(rc=rc, nefc=k, efc_J=J[:k], efc_pos=pos[:k], efc_aref=aref[:k], efc_D=D[:k],
                 efc_force=force[:k], niter=niter.value, energy=en, momentum=mom)
        return o

    def step(self, qpos, qvel, ctrl, frame_skip=0, nthreads=1):
        qpos, qvel, ctrl = _c(qpos, (-1, self.nq)), _c(qvel, (-1, self.nv)), _c(ctrl, (-1, self.nu))
        N = qpos.shape[0]
        qn, vn, r = np.empty_like(qpos), np.empty_like(qvel), np.empty(N)
        st = np.zeros(N, dtype=np.uint32)
        rc = self.L.mjo_step(self.id, frame_skip, N, _p(qpos), _p(qvel), _p(ctrl), _p(qn), _p(vn), _p(r),
                             st.ctypes.data_as(up), nthreads)
        if rc:
            raise RuntimeError("mjo_step -> %d" % rc)
        return qn, vn, r, st

    def fd_jacobians(self, qpos, qvel, ctrl, eps=1e-6, fd_mode=0, nsteps=1, frame_skip=0, nthreads=1, info=False):
        """info=True: also out["evals_constrained"] [N], the number of dynamics evaluations of each unit's rollouts that
        had an active joint-limit or contact row (0: the unit is smooth), and out["evals_crossing"] [N], the number of
        capsule/capsule contacts made by the crossing-axes rule ("parity unpinned")."""
        qpos, qvel, ctrl = _c(qpos, (-1, self.nq)), _c(qvel, (-1, self.nv)), _c(ctrl, (-1, self.nu))
        N, n, nu = qpos.shape[0], self.n, self.nu
        out = dict(dsds=np.empty((N, n, n)), dsda=np.empty((N, n, nu)), drds=np.empty((N, n)),
                   drda=np.empty((N, nu)), s_next=np.empty((N, n)), reward=np.empty(N),
                   status=np.zeros(N, dtype=np.uint32))
        args = (self.id, frame_skip, N, _p(qpos), _p(qvel), _p(ctrl), eps, fd_mode, nsteps,
                _p(out["dsds"]), _p(out["dsda"]), _p(out["drds"]), _p(out["drda"]),
                _p(out["s_next"]), _p(out["reward"]), out["status"].ctypes.data_as(up))
        if info:
            out["evals_constrained"] = np.zeros(N, dtype=np.int32)
            out["evals_crossing"] = np.zeros(N, dtype=np.int32)
            rc = self.L.mjo_fd_jacobians_info(*args, out["evals_constrained"].ctypes.data_as(ip),
                                              out["evals_crossing"].ctypes.data_as(ip), nthreads)
        else:
            rc = self.L.mjo_fd_jacobians(*args, nthreads)
        if rc:
            raise RuntimeError("mjo_fd_jacobians -> %d" % rc)
        return out

    def stats(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.mjo_stats(a, b, c)
        return dict(mean_newton_iters=a.value, frac_evals_constrained=b.value, mean_nefc=c.value)
