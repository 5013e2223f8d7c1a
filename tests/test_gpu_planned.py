"""GPU: the two-launch linearisation (mbrl_fd_plan + mbrl_fd_jacobians_planned, include/mbrl.h) against the single
launch mbrl_fd_jacobians_closed_loop -- the same rollouts in another order, so every output must agree BIT FOR BIT --
plus the plan buffer's argument checks.  The single launch itself is checked against the oracle and the reference's
golden vectors in test_gpu_parity.py / test_gpu_ops.py."""
import ctypes as C

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _eng(env):
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    return FdEngine(env)


def _states(e, N, seed):
    """Mixed batch: reset-like states, then the same advanced a random number of steps (contacts, limits)."""
    import torch
    from mbrl_amd import workloads
    q, v, c = workloads.s2(e, N, seed=seed, horizon=8)
    return q, v, c


@pytest.mark.parametrize("fd_mode", [0, 1])
@pytest.mark.parametrize("env", util.ENVS)
def test_planned_equals_single_launch(env, fd_mode):
    import torch
    e = _eng(env)
    for N in (1, 3, 67, 1500):
        q, v, c = _states(e, N, seed=N)
        a = e.fd_jacobians(q, v, c, fd_mode=fd_mode, planned=False)
        b = e.fd_jacobians(q, v, c, fd_mode=fd_mode, planned=True)
        torch.cuda.synchronize()
        for k in a:
            assert torch.equal(a[k], b[k]), (env, fd_mode, N, k)


@pytest.mark.parametrize("env", ["HalfCheetahEnv", "HopperEnv", "InvertedDoublePendulumEnv"])
def test_planned_equals_single_launch_two_env_steps(env):
    """cfg.MODEL.NSTEPS_FOR_BACKWARD = 2 (backward.py:359): both launches run two env steps per rollout."""
    import torch
    e = _eng(env)
    q, v, c = _states(e, 500, seed=3)
    a = e.fd_jacobians(q, v, c, nsteps=2, planned=False)
    b = e.fd_jacobians(q, v, c, nsteps=2, planned=True)
    one = e.fd_jacobians(q, v, c, nsteps=1, planned=True)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), (env, k)
    assert not torch.equal(a["s_next"], one["s_next"])


def test_planned_closed_loop_and_plan_reuse():
    """One plan serves forward, central and closed-loop launches on the same inputs."""
    import torch
    e = _eng("HalfCheetahEnv")
    N = 333
    q, v, c = _states(e, N, seed=5)
    g = torch.Generator(device="cpu").manual_seed(0)
    cp = torch.rand((N, e.n, e.nu), generator=g, dtype=torch.float64).mul(2).sub(1).cuda()
    out = e.fd_plan(q, v, c)
    for kw in (dict(fd_mode=0), dict(fd_mode=1), dict(fd_mode=0, ctrl_pert=cp)):
        ref = e.fd_jacobians(q, v, c, planned=False, **kw)
        st0 = out["status"].clone()
        got = e.fd_planned(q, v, c, out, **kw)
        torch.cuda.synchronize()
        for k in ("dsds", "dsda", "drds", "drda", "s_next", "reward"):
            assert torch.equal(ref[k], got[k]), (kw.keys(), k)
        assert torch.equal(ref["status"] | st0, got["status"])      # a planned launch ORs into the plan's status words


def test_planned_status_and_optional_outputs():
    import torch
    from mbrl_amd import lib
    e = _eng("HalfCheetahEnv")
    rng = np.random.default_rng(0)
    N = 41
    q = rng.uniform(-0.1, 0.1, (N, 9)); v = 0.1 * rng.standard_normal((N, 9)); c = rng.uniform(-1, 1, (N, 6))
    q[7, 3] = np.nan
    v[12, 0] = np.inf
    full = e.fd_jacobians(q, v, c, planned=True)
    bad = np.zeros(N, bool); bad[[7, 12]] = True
    assert ((full["status"].cpu().numpy() & lib.ST_NONFINITE) != 0).tolist() == bad.tolist()
    ok = ~torch.as_tensor(bad)
    assert torch.isfinite(full["dsds"][ok]).all()
    # only drds requested, no status, no nominal outputs
    p = lambda t: C.c_void_p(t.data_ptr())
    null = C.c_void_p(0)
    qt, vt, ct = (torch.as_tensor(x, device="cuda") for x in (q, v, c))
    nbytes = e.L.mbrl_fd_plan_bytes(e.h, N)
    assert nbytes > 0
    plan = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    only = torch.full((N, 18), 7.0, dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert e.L.mbrl_fd_plan(e.h, N, p(qt), p(vt), p(ct), 1, p(plan), nbytes, null, null, null, st) == 0
    assert e.L.mbrl_fd_jacobians_planned(e.h, N, p(qt), p(vt), p(ct), null, 1e-6, 0, 1, p(plan), nbytes,
                                         null, null, p(only), null, null, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(only[ok], full["drds"][ok])


def test_plan_argument_checks():
    import torch
    e = _eng("SwimmerEnv")
    L, h = e.L, e.h
    N = 10
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
    q, v, c = z(N, 5), z(N, 5), z(N, 2)
    p = lambda t: C.c_void_p(t.data_ptr())
    null = C.c_void_p(0)
    nbytes = L.mbrl_fd_plan_bytes(h, N)
    assert nbytes > 0 and L.mbrl_fd_plan_bytes(h, 0) == 0
    assert L.mbrl_fd_plan_bytes(h, -1) == -6 and L.mbrl_fd_plan_bytes(None, N) == -1
    assert L.mbrl_fd_plan_bytes(h, 100 * N) > nbytes
    plan = torch.empty(nbytes + 512, dtype=torch.uint8, device="cuda")
    assert L.mbrl_fd_plan(h, N, p(q), p(v), p(c), 1, p(plan), nbytes - 1, null, null, null, null) == -8     # too small
    assert b"mbrl_fd_plan_bytes" in L.mbrl_last_error()
    assert L.mbrl_fd_plan(h, N, p(q), p(v), p(c), 1, C.c_void_p(plan.data_ptr() + 8), nbytes, null, null, null, null) == -7
    assert L.mbrl_fd_plan(h, N, p(q), p(v), p(c), 1, null, nbytes, null, null, null, null) == -1
    assert L.mbrl_fd_plan(h, 0, null, null, null, 1, null, 0, null, null, null, null) == 0
    assert L.mbrl_fd_plan(h, N, p(q), p(v), p(c), 1, p(plan), nbytes, null, null, null, null) == 0
    out = [null] * 5
    assert L.mbrl_fd_jacobians_planned(h, N, p(q), p(v), p(c), null, 0.0, 0, 1, p(plan), nbytes, *out, null) == -7
    assert L.mbrl_fd_jacobians_planned(h, N, p(q), p(v), p(c), null, 1e-6, 3, 1, p(plan), nbytes, *out, null) == -7
    assert L.mbrl_fd_jacobians_planned(h, N, p(q), p(v), p(c), null, 1e-6, 0, 1, p(plan), 16, *out, null) == -8
    assert L.mbrl_fd_jacobians_planned(h, N, p(q), p(v), p(c), null, 1e-6, 0, 1, null, nbytes, *out, null) == -1
    assert L.mbrl_fd_jacobians_planned(h, N, p(q), p(v), p(c), null, 1e-6, 0, 1, p(plan), nbytes, *out, null) == 0
    torch.cuda.synchronize()


def test_plan_orders_units_by_solver_key():
    """The permutation is a permutation, and its keys (segment key_out of the plan) are non-decreasing."""
    import torch
    e = _eng("HalfCheetahEnv")
    N = 5000
    q, v, c = _states(e, N, seed=9)
    e.fd_plan(q, v, c)
    torch.cuda.synchronize()
    up = lambda x: (x + 255) // 256 * 256          # segment layout of the plan buffer: csrc/mbrl_hip.hip PlanOffsets
    w, w2 = up(4 * N), up(8 * N)
    plan = e._plan
    h = 256                                         # header
    key_in = plan[h:h + 8 * N].view(torch.int64).cpu().numpy()
    key_out = plan[h + w2:h + w2 + 8 * N].view(torch.int64).cpu().numpy()
    perm = plan[h + 2 * w2 + w:h + 2 * w2 + w + 4 * N].view(torch.int32).cpu().numpy().astype(np.int64)
    head = plan[0:16].view(torch.int64).cpu().numpy()                      # PlanHeader: magic, units, nsteps, frame_skip, env
    assert head[1] == N and plan[16:28].view(torch.int32).tolist() == [1, 10, 5]
    assert sorted(perm.tolist()) == list(range(N))
    assert (np.diff(key_out) >= 0).all()
    assert (key_in[perm] == key_out).all()
    assert len(np.unique(key_in)) > 3            # the batch really mixes control-flow classes


def test_planned_launch_needs_a_plan_for_the_same_batch():
    """The engine refuses a planned launch whose plan was made for another batch size; at the C ABI a plan made for
    another N or nsteps is recognised on the device: every unit is flagged MBRL_ST_BADPLAN and nothing is written."""
    import torch
    from mbrl_amd import lib
    e = _eng("SwimmerEnv")
    q, v, c = _states(e, 40, seed=1)
    out = e.fd_plan(q, v, c)
    with pytest.raises(lib.MbrlError):
        e.fd_planned(q[:20], v[:20], c[:20], e.new_outputs(20))
    # C ABI: plan for 40 units, launch for 20 with the same (large enough) buffer -- must not fault
    p = lambda t: C.c_void_p(t.data_ptr())
    null = C.c_void_p(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for N, nsteps in ((20, 1), (40, 2)):
        o = e.new_outputs(N)
        o["dsds"].fill_(7.0)
        o["status"].zero_()
        assert e.L.mbrl_fd_jacobians_planned(e.h, N, p(q), p(v), p(c), null, 1e-6, 0, nsteps, p(e._plan), e._plan.numel(),
                                             p(o["dsds"]), p(o["dsda"]), p(o["drds"]), p(o["drda"]), p(o["status"]), st) == 0
        torch.cuda.synchronize()
        assert bool((o["status"] == lib.ST_BADPLAN).all()) and bool((o["dsds"] == 7.0).all())
    # the right arguments still work after the refused launches
    o = e.new_outputs(40)
    o["status"].zero_()
    assert e.L.mbrl_fd_jacobians_planned(e.h, 40, p(q), p(v), p(c), null, 1e-6, 0, 1, p(e._plan), e._plan.numel(),
                                         p(o["dsds"]), p(o["dsda"]), p(o["drds"]), p(o["drda"]), p(o["status"]), st) == 0
    torch.cuda.synchronize()
    assert bool((o["status"] == 0).all())
    assert torch.equal(o["dsds"], e.fd_jacobians(q, v, c, planned=False)["dsds"])


def test_two_streams_linearise_side_by_side_with_their_own_plans():
    """FdEngine keeps one launch plan per stream: two linearisations interleaved on two streams (plan A, plan B,
    planned A, planned B -- the order that would trample a shared plan) give the single-stream bits."""
    import torch
    e = _eng("HalfCheetahEnv")
    qa, va, ca = _states(e, 2500, seed=21)
    qb, vb, cb = _states(e, 1700, seed=22)
    ref_a = e.fd_jacobians(qa, va, ca, planned=True)
    ref_b = e.fd_jacobians(qb, vb, cb, planned=True)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    oa, ob = e.new_outputs(2500), e.new_outputs(1700)
    with torch.cuda.stream(s1):
        e.fd_plan(qa, va, ca, out=oa)
    with torch.cuda.stream(s2):
        e.fd_plan(qb, vb, cb, out=ob)
    with torch.cuda.stream(s1):
        e.fd_planned(qa, va, ca, oa)
    with torch.cuda.stream(s2):
        e.fd_planned(qb, vb, cb, ob)
    torch.cuda.synchronize()
    for k in ref_a:
        assert torch.equal(ref_a[k], oa[k]) and torch.equal(ref_b[k], ob[k]), k
    # an explicit, caller-owned plan works the same way
    p = e.new_plan(2500)
    oc = e.fd_jacobians(qa, va, ca, planned=True, plan=p)
    torch.cuda.synchronize()
    assert p.units == 2500 and all(torch.equal(ref_a[k], oc[k]) for k in ref_a)
    # "cached" never measures: single launch until tune_schedule has run for the size class
    assert e.known_schedule(2500) is None
    od = e.fd_jacobians(qa, va, ca, planned="cached")
    assert e.known_schedule(2500) is None and all(torch.equal(ref_a[k], od[k]) for k in ref_a)
    assert e.tune_schedule(qa, va, ca) in (True, False) and e.known_schedule(2500) is not None


def test_full_bench_size_properties():
    """BASELINE.json's full size (4096 x 40 = 163 840 HalfCheetah units), through size-independent properties: the
    two-launch schedule equals the single launch bit for bit, the nominal rollouts equal mbrl_step bit for bit (the
    reference's assert, backward.py:371,374), a second launch reproduces the bits, no unit is flagged, and the
    action columns of units with saturated controls vanish exactly."""
    import torch
    from mbrl_amd import workloads
    e = _eng("HalfCheetahEnv")
    N = 4096 * 40
    q, v, c = workloads.s2(e, N, seed=0, horizon=40)
    c = c.clone()
    c[::97] = torch.sign(c[::97]) * 1.25                      # every 97th unit: all controls beyond ctrlrange
    a = e.fd_jacobians(q, v, c, planned=False)
    b = e.fd_jacobians(q, v, c, planned=True)
    b2 = e.fd_jacobians(q, v, c, planned=True)
    qn, vn, r, st = e.step(q, v, c)
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(b[k], b2[k]), k
    assert torch.equal(torch.cat([qn, vn], dim=1), b["s_next"]) and torch.equal(r, b["reward"])
    assert int((b["status"] != 0).sum()) == 0 and int((st != 0).sum()) == 0
    assert float(b["dsda"][::97].abs().max()) == 0.0 and float(b["dsda"].abs().max()) > 0.0
    assert bool(torch.isfinite(b["dsds"]).all())
