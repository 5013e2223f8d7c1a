"""GPU: ONE call sequence issued twice -- through include/mbrl.h (libmbrl_hip.so, device buffers) and through its CPU
twin oracle/mbrl_cpu.h (the oracle, host buffers): create -> dims -> step -> fd_jacobians -> vjp_scan, the path
MjBlock.forward / calculate_gradients / MjBlock.backward take in the reference (model/blocks/mujoco.py:12-104,
mujoco/utils/backward.py:343-480).  Tolerances as tests/test_gpu_parity.py: next state and reward 1e-10, Jacobians 1e-6
relative to the block's scale (north_star) on smooth models and 1e-5 where contact is active, and the scan, which is the
same arithmetic on the same Jacobians, 1e-12."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NULL = C.c_void_p(0)
CASES = [("InvertedPendulumEnv", 1e-6), ("SwimmerEnv", 1e-6), ("HopperEnv", 1e-5), ("HalfCheetahEnv", 1e-5)]


def sequence(L, suffix, env, q, v, a, c, B, H, gw, put, get, stream):
    """The calls, written once.  put: numpy -> buffer of that side (+ pointer), get: buffer -> numpy."""
    f = lambda name: getattr(L, name + suffix)
    h = C.c_void_p()
    assert f("mbrl_model_create")(env.encode(), 0, 0, C.byref(h)) == 0
    nq, nv, nu, dt = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    assert f("mbrl_model_dims")(h, nq, nv, nu, dt) == 0
    n, N = nq.value + nv.value, B * H
    bufs = {k: put(x) for k, x in dict(q=q, v=v, a=a, c=c).items()}
    shapes = dict(qn=(N, nq.value), vn=(N, nv.value), r1=(N,), dsds=(N, n, n), dsda=(N, n, nu.value), drds=(N, n),
                  drda=(N, nu.value), s_next=(N, n), reward=(N,), da=(N, nu.value), ds0=(B, n))
    for k, s in shapes.items():
        bufs[k] = put(np.zeros(s))
    bufs["st"] = put(np.zeros(N, np.uint32))
    p = lambda k: bufs[k][1]
    assert f("mbrl_step")(h, N, p("q"), p("v"), p("a"), p("qn"), p("vn"), p("r1"), p("st"), stream) == 0
    assert f("mbrl_fd_jacobians")(h, N, p("q"), p("v"), p("a"), 1e-6, 0, 1, p("dsds"), p("dsda"), p("drds"), p("drda"),
                                  p("s_next"), p("reward"), p("st"), stream) == 0
    assert f("mbrl_vjp_scan")(h, B, H, gw, p("dsds"), p("dsda"), p("drds"), p("drda"), p("c"), p("da"), p("ds0"), stream) == 0
    out = {k: get(bufs[k][0]) for k in list(shapes) + ["st"]}
    out["dims"] = (nq.value, nv.value, nu.value, dt.value)
    f("mbrl_model_destroy")(h)
    return out


@pytest.mark.parametrize("env,jtol", CASES, ids=[c[0][:-3] for c in CASES])
@pytest.mark.parametrize("gw", [0, 2], ids=["none", "prioritise"])
def test_one_call_sequence_through_both_libraries(env, jtol, gw):
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd import lib, workloads
    from mbrl_amd.engine import FdEngine
    import oracle as orc
    B, H = 6, 20
    e = FdEngine(env)
    q, v, a = (t.cpu().numpy() for t in workloads.s2(e, B * H, seed=3, horizon=H))
    c = np.random.default_rng(5).standard_normal((B, H))

    def put_gpu(x):
        t = torch.as_tensor(x, device="cuda").contiguous()
        return t, C.c_void_p(t.data_ptr())

    def put_cpu(x):
        x = np.ascontiguousarray(x).copy()
        return x, C.c_void_p(x.ctypes.data)

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gpu = sequence(lib.load(), "", env, q, v, a, c, B, H, gw, put_gpu, lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1], stream)
    cpu = sequence(orc.lib(), "_cpu", env, q, v, a, c, B, H, gw, put_cpu, lambda x: x, NULL)
    assert gpu["dims"] == cpu["dims"]
    assert np.array_equal(gpu["st"], cpu["st"]) and not cpu["st"].any()
    rel = lambda x, y: float((np.abs(x - y).reshape(len(x), -1).max(1) / (1.0 + np.abs(y).reshape(len(y), -1).max(1))).max())
    for k in ("qn", "vn", "r1", "s_next", "reward"):
        assert rel(gpu[k], cpu[k]) <= 1e-10, (k, rel(gpu[k], cpu[k]))
    for k in ("dsds", "dsda", "drds", "drda"):
        assert rel(gpu[k], cpu[k]) <= jtol, (k, rel(gpu[k], cpu[k]))
    # the scan: each side ran it on its own Jacobians, so their difference is the Jacobians' carried along the horizon
    # (printed, not bounded); on the SAME Jacobians the two recurrences agree to rounding
    print("%s gw=%d: da of the two sequences differ by %.1e (relative)" % (
        env, gw, np.abs(gpu["da"] - cpu["da"]).max() / (1.0 + np.abs(cpu["da"]).max())))
    L = orc.lib()
    h = C.c_void_p()
    assert L.mbrl_model_create_cpu(env.encode(), 0, 0, C.byref(h)) == 0
    da, ds0 = np.zeros_like(gpu["da"]), np.zeros_like(gpu["ds0"])
    P = lambda x: C.c_void_p(np.ascontiguousarray(x).ctypes.data)
    J = {k: np.ascontiguousarray(gpu[k]) for k in ("dsds", "dsda", "drds", "drda")}
    cc = np.ascontiguousarray(c)
    assert L.mbrl_vjp_scan_cpu(h, B, H, gw, P(J["dsds"]), P(J["dsda"]), P(J["drds"]), P(J["drda"]), P(cc), P(da), P(ds0), NULL) == 0
    L.mbrl_model_destroy_cpu(h)
    for got, ref in ((gpu["da"], da), (gpu["ds0"], ds0)):
        assert np.abs(got - ref).max() / (1.0 + np.abs(ref).max()) <= 1e-12
