"""CPU: the oracle behind the call shapes of include/mbrl.h (oracle/mbrl_cpu.h, SURVEY.md 8(b) "identical *_cpu entry
points for the oracle twin") -- exports, error codes, and the reference's golden vectors through those entry points:
calculate_gradients (mujoco/utils/backward.py:343-480) bit for bit, the autograd chain of MjBlock.backward
(model/blocks/mujoco.py:43-104) to 1e-10.  The GPU side of the same call sequence is tests/test_gpu_cpu_twin.py."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz")))
WEIGHTS = {"none": 0, "average": 1, "prioritise": 2}
NULL = C.c_void_p(0)


def P(a):
    return C.c_void_p(a.ctypes.data)


def model(env, frame_skip=0):
    L = orc.lib()
    h = C.c_void_p()
    assert L.mbrl_model_create_cpu(env.encode(), frame_skip, 0, C.byref(h)) == 0, L.mbrl_last_error_cpu()
    return L, h


def test_every_declared_entry_point_is_exported():
    text = open(os.path.join(ROOT, "oracle", "mbrl_cpu.h")).read()
    declared = sorted(set(re.findall(r"\b(mbrl_\w+_cpu)\s*\(", text)))
    assert declared == sorted(orc.CPU_TWIN_SYMBOLS)
    L = orc.lib()
    for name in declared:
        assert hasattr(L, name), name
    # one twin per entry point of SURVEY.md 8(b)'s list in include/mbrl.h
    main = open(os.path.join(ROOT, "include", "mbrl.h")).read()
    for name in declared:
        assert re.search(r"\b%s\s*\(" % name[:-4], main), name
    assert L.mbrl_abi_version_cpu() == int(re.search(r"#define MBRL_ABI_VERSION (\d+)", main).group(1))


def test_same_return_codes_for_the_same_mistakes_as_the_hip_library():
    """The codes tests/test_gpu_abi_errors.py pins on the GPU side."""
    L, h = model("HalfCheetahEnv")
    q, v, c = np.zeros((4, 9)), np.zeros((4, 9)), np.zeros((4, 6))
    out = [NULL] * 7
    assert L.mbrl_fd_jacobians_cpu(h, 4, P(q), P(v), P(c), 0.0, 0, 1, *out, NULL) == -7
    assert b"eps" in L.mbrl_last_error_cpu()
    assert L.mbrl_fd_jacobians_cpu(h, 4, P(q), P(v), P(c), 1e-6, 5, 1, *out, NULL) == -7
    assert L.mbrl_fd_jacobians_cpu(h, 4, NULL, P(v), P(c), 1e-6, 0, 1, *out, NULL) == -1
    assert L.mbrl_fd_jacobians_cpu(h, -1, P(q), P(v), P(c), 1e-6, 0, 1, *out, NULL) == -6
    assert L.mbrl_fd_jacobians_cpu(None, 4, P(q), P(v), P(c), 1e-6, 0, 1, *out, NULL) == -1
    assert L.mbrl_fd_jacobians_cpu(h, 0, NULL, NULL, NULL, 1e-6, 0, 1, *out, NULL) == 0
    assert L.mbrl_step_cpu(h, 4, P(q), P(v), P(c), NULL, NULL, NULL, NULL, NULL) == -1
    hh = C.c_void_p()
    assert L.mbrl_model_create_cpu(b"LegEnv", 0, 0, C.byref(hh)) == -2 and not hh.value
    z = np.zeros(8)
    assert L.mbrl_vjp_scan_cpu(h, 1, 1, 3, P(z), P(z), P(z), P(z), P(z), P(z), NULL, NULL) == -7
    assert L.mbrl_vjp_scan_layout_cpu(h, 1, 1, 0, 2, P(z), P(z), P(z), P(z), P(z), P(z), NULL, NULL) == -7
    assert L.mbrl_vjp_scan_sharded_cpu(h, 2, 1, 0, 0, 1, 2, P(z), P(z), P(z), P(z), P(z), P(z), NULL, NULL) == -6
    assert L.mbrl_vjp_scan_cpu(h, 0, 5, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == 0
    nq, nv, nu, dt = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    assert L.mbrl_model_dims_cpu(h, nq, nv, nu, dt) == 0
    assert (nq.value, nv.value, nu.value) == (9, 9, 6) and abs(dt.value - 0.1) < 1e-15          # timestep 0.01 x frame_skip 10
    L.mbrl_model_destroy_cpu(h)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
def test_golden_vectors_through_the_twin_entry_points(path):
    g = np.load(path)
    L, h = model(str(g["env"]))
    B, H = g["c"].shape
    n, nu = g["dsds"].shape[-1], g["dsda"].shape[-1]
    nq = n // 2
    s = np.ascontiguousarray(g["states"][:, :H].reshape(B * H, n))
    q, v = np.ascontiguousarray(s[:, :nq]), np.ascontiguousarray(s[:, nq:])
    a = np.ascontiguousarray(g["actions"].reshape(B * H, nu))
    N = B * H
    A, Bm, rs, ra = np.empty((N, n, n)), np.empty((N, n, nu)), np.empty((N, n)), np.empty((N, nu))
    sn, r, st = np.empty((N, n)), np.empty(N), np.zeros(N, np.uint32)
    assert L.mbrl_fd_jacobians_cpu(h, N, P(q), P(v), P(a), 1e-6, 0, int(g["nsteps"]), P(A), P(Bm), P(rs), P(ra), P(sn), P(r),
                                   P(st), NULL) == 0
    assert not st.any()
    for got, key in ((A, "dsds"), (Bm, "dsda"), (rs, "drds"), (ra, "drda")):
        assert np.array_equal(got.reshape(g[key].shape), g[key]), key
    if int(g["nsteps"]) == 1:
        assert np.array_equal(sn.reshape(B, H, n), g["states"][:, 1:]) and np.array_equal(r.reshape(B, H), g["rewards"])
        qn, vn, r1 = np.empty_like(q), np.empty_like(v), np.empty(N)
        assert L.mbrl_step_cpu(h, N, P(q), P(v), P(a), P(qn), P(vn), P(r1), P(st), NULL) == 0
        assert np.array_equal(np.concatenate([qn, vn], 1), sn) and np.array_equal(r1, r)
    # ---- the autograd chain of one epoch ----
    gw = WEIGHTS[str(g["weights"])]
    cc = np.ascontiguousarray(g["c"])
    da, ds0 = np.empty((B, H, nu)), np.empty((B, n))
    assert L.mbrl_vjp_scan_cpu(h, B, H, gw, P(A), P(Bm), P(rs), P(ra), P(cc), P(da), P(ds0), NULL) == 0
    np.testing.assert_allclose(da, g["grad_actions"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ds0, g["grad_s0"], rtol=1e-10, atol=1e-12)
    # ---- time-major order and shards: the same numbers, bit for bit ----
    tm = lambda x: np.ascontiguousarray(x.reshape((B, H) + x.shape[1:]).swapaxes(0, 1))
    da_t = np.empty((H, B, nu)); ds0_t = np.empty((B, n))
    assert L.mbrl_vjp_scan_layout_cpu(h, B, H, gw, 1, P(tm(A)), P(tm(Bm)), P(tm(rs)), P(tm(ra)), P(np.ascontiguousarray(cc.T)),
                                      P(da_t), P(ds0_t), NULL) == 0
    assert np.array_equal(da_t.swapaxes(0, 1), da) and np.array_equal(ds0_t, ds0)
    if B >= 2:
        cut = B // 2
        parts = []
        for lo, hi in ((0, cut), (cut, B)):
            k = slice(lo * H, hi * H)
            d, d0 = np.empty((hi - lo, H, nu)), np.empty((hi - lo, n))
            assert L.mbrl_vjp_scan_sharded_cpu(h, hi - lo, H, gw, 0, lo, B, P(np.ascontiguousarray(A[k])),
                                               P(np.ascontiguousarray(Bm[k])), P(np.ascontiguousarray(rs[k])),
                                               P(np.ascontiguousarray(ra[k])), P(np.ascontiguousarray(cc[lo:hi])), P(d), P(d0),
                                               NULL) == 0
            parts.append((d, d0))
        assert np.array_equal(np.concatenate([p[0] for p in parts]), da)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), ds0)
    L.mbrl_model_destroy_cpu(h)
