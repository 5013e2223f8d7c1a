"""GPU: the no-allocation entry point mbrl_fd_jacobians_static (include/mbrl.h; no counterpart in the reference, for
callers that cannot hipMalloc) against the existing entry point mbrl_fd_jacobians, through the C ABI.  Both run the same
rollouts, so every output must agree BIT FOR BIT (the bar for mbrl_fd_jacobians itself -- oracle and reference golden
vectors -- is in test_gpu_parity.py / test_gpu_ops.py).  Two calls in a row on the same GPU share its one static plan
array; the second batch is smaller, so stale rows of the first call sit behind the second call's rows."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

NAMES = ("dsds", "dsda", "drds", "drda", "s_next", "reward", "status")


def _call(fn, e, q, v, c, fd_mode, nsteps, o):
    import torch
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return fn(e.h, q.shape[0], p(q), p(v), p(c), 1e-6, fd_mode, nsteps, *(p(o[k]) for k in NAMES), st)


@pytest.mark.parametrize("env,fd_mode", [("HalfCheetahEnv", 0), ("HalfCheetahEnv", 1), ("HopperEnv", 0),
                                         ("InvertedPendulumEnv", 0)])
def test_static_equals_existing_entry_point_twice_in_a_row(env, fd_mode):
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd import workloads
    from mbrl_amd.engine import FdEngine
    e = FdEngine(env)
    for N in (3000, 701):                       # the second call is the smaller one
        q, v, c = workloads.s2(e, N, seed=N + fd_mode, horizon=8)
        ref, got = e.new_outputs(N), e.new_outputs(N)
        for o in (ref, got):
            for k in NAMES:
                o[k].fill_(7)
        assert _call(e.L.mbrl_fd_jacobians, e, q, v, c, fd_mode, 1, ref) == 0
        assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, fd_mode, 1, got) == 0, e.L.mbrl_last_error()
        torch.cuda.synchronize()
        for k in NAMES:
            assert torch.equal(ref[k], got[k]), (env, fd_mode, N, k)
        assert bool(torch.isfinite(got["dsds"][got["status"] == 0]).all())


def test_static_rejects_batches_beyond_its_scratch_and_allocates_nothing():
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd import lib
    from mbrl_amd.engine import FdEngine
    e = FdEngine("HalfCheetahEnv")
    N = lib.STATIC_MAX_UNITS + 1
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
    q, v, c = z(N, 9), z(N, 9), z(N, 6)
    o = e.new_outputs(N)
    assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, o) == -8
    assert b"MBRL_STATIC_MAX_UNITS" in e.L.mbrl_last_error()
    assert _call(e.L.mbrl_fd_jacobians_static, e, q[:0], v[:0], c[:0], 0, 1, o) == 0          # empty batch
    # the largest admitted batch fits the array and allocates no device memory on the way (driver-level free memory is
    # unchanged; torch's caching allocator is not involved: the outputs exist already)
    N = lib.STATIC_MAX_UNITS
    o = e.new_outputs(N)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    assert _call(e.L.mbrl_fd_jacobians_static, e, q[:N], v[:N], c[:N], 0, 1, o) == 0, e.L.mbrl_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert int((o["status"] != 0).sum()) == 0


def test_static_entry_point_replays_from_a_captured_hip_graph():
    """What the entry point is for: a caller under HIP graph capture.  The call is captured once (torch.cuda.CUDAGraph)
    and replayed three times on two different batches written into the graph's input buffers; every replay gives the
    bits of mbrl_fd_jacobians.  (The status words are cleared by a kernel: a captured hipMemsetAsync left garbage in
    them from the second replay on.)"""
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd import workloads
    from mbrl_amd.engine import FdEngine
    e = FdEngine("HalfCheetahEnv")
    N = 2048
    p = lambda t: C.c_void_p(t.data_ptr())
    qa, va, ca = workloads.s2(e, N, seed=1, horizon=8)
    qb, vb, cb = workloads.s2(e, N, seed=2, horizon=8)
    q, v, c = qa.clone(), va.clone(), ca.clone()          # the graph's input buffers
    o = e.new_outputs(N)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                         # first use outside the capture (lazy module loading)
        assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, o) == 0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, o) == 0, e.L.mbrl_last_error()
    for qq, vv, cc in ((qb, vb, cb), (qa, va, ca), (qb, vb, cb)):
        q.copy_(qq); v.copy_(vv); c.copy_(cc)
        for k in NAMES:
            o[k].fill_(7)
        g.replay()
        torch.cuda.synchronize()
        ref = e.new_outputs(N)
        assert _call(e.L.mbrl_fd_jacobians, e, qq, vv, cc, 0, 1, ref) == 0
        torch.cuda.synchronize()
        for k in NAMES:
            assert torch.equal(ref[k], o[k]), k


def test_static_calls_that_overlap_are_flagged_not_silently_wrong():
    """Every static call on a GPU shares that GPU's one plan array, so calls that overlap would race on it.  The array's
    header counts the calls in flight; a call that runs while another is in flight gets MBRL_ST_BADPLAN on every unit.
    The other call is stood in for by the test knob mbrl_debug_static_hold (no real race needed): held -> flagged;
    released -> the next call is clean again and bit-equal to mbrl_fd_jacobians."""
    import torch
    import mbrl_amd  # noqa: F401
    from mbrl_amd import lib, workloads
    from mbrl_amd.engine import FdEngine
    e = FdEngine("HalfCheetahEnv")
    N = 1500
    q, v, c = workloads.s2(e, N, seed=4, horizon=8)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref, got = e.new_outputs(N), e.new_outputs(N)
    assert _call(e.L.mbrl_fd_jacobians, e, q, v, c, 0, 1, ref) == 0
    assert e.L.mbrl_debug_static_hold(e.h, 1, st) == 0                 # "another call has started"
    assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, got) == 0
    torch.cuda.synchronize()
    assert bool(((got["status"] & lib.ST_BADPLAN) != 0).all())
    assert e.L.mbrl_debug_static_hold(e.h, 0, st) == 0                 # "... and finished"
    for k in NAMES:
        got[k].fill_(7)
    assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, got) == 0
    torch.cuda.synchronize()
    for k in NAMES:
        assert torch.equal(ref[k], got[k]), k
    # two real calls on two streams, serialised by the caller with an event: clean
    side = torch.cuda.Stream()
    other = e.new_outputs(N)
    assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, got) == 0
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        assert _call(e.L.mbrl_fd_jacobians_static, e, q, v, c, 0, 1, other) == 0
    torch.cuda.synchronize()
    for k in NAMES:
        assert torch.equal(ref[k], got[k]) and torch.equal(ref[k], other[k]), k
