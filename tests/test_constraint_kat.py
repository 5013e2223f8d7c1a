"""Closed-form CONSTRAINT known-answer vectors (tests/golden/gen_constraint_kat.py -> kat_constraints.npz) against the
oracle's mj_forward, the kernel's per-lane simulator built for the host, and -- marked gpu -- the HIP step kernel.

The vectors were not produced by either implementation: constraint parameters were read from the reference's MJCF
files, the smooth dynamics come from an autodiff Lagrangian, and the constrained acceleration from enumerating the
active sets of the convex problem (no Newton iteration, no line search).  They cover: every joint limit of every model
on both sides, inside and beyond the impedance ramp, with refsafe (InvertedPendulum: timeconst 0.02 < 2 dt) and a
non-zero limit margin (InvertedDoublePendulum slider); one capsule end on the floor with pyramidal friction and a
sliding velocity (Hopper, Walker2d, HalfCheetah: 0, 1, 3 or 4 rows active at the solution); two or three capsule ends
on the floor at once (both ends of Hopper's foot, two feet of Walker2d / HalfCheetah: up to 9 coupled rows); a frictionless
capsule/capsule row (Hopper foot against thigh/torso); Swimmer's inertia-box fluid forces.  The MuJoCo rules each group
encodes are listed in the generator's header and in DESIGN.md section 2; MuJoCo itself stays unavailable offline.
Tolerance: 1e-9 relative on qacc (the closed form goes through numpy solves of systems with condition ~1e6), 1e-10
on the state after one mj_step."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from oracle import Oracle

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_constraints.npz")))
G.update(np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_constraints_multi.npz")))   # 2-3 contacts at once
GROUPS = sorted({k.rsplit("_", 1)[0] for k in G if k.endswith("_qacc")})
ENV_OF = {e[:-3].lower(): e for e in util.ENVS}
dp = C.POINTER(C.c_double)


def _env(group):
    return ENV_OF[group.split("_", 1)[1]]


def _rel(a, b):
    return np.abs(a - b).max() / (1.0 + np.abs(b).max())


def test_fixture_covers_what_it_claims():
    assert {g.split("_")[0] for g in GROUPS} == {"limit", "floor", "pair", "fluid", "multi"}
    assert all(G[g + "_dof"].min() >= 2 for g in GROUPS if g.startswith("multi_"))        # 2-3 simultaneous contacts
    assert len([g for g in GROUPS if g.startswith("limit_")]) == 6
    for g in GROUPS:
        if not g.startswith("fluid"):
            assert G[g + "_nactive"].max() >= 1, g            # constraint forces really act in every group
    assert {0, 1, 3, 4} <= set(np.concatenate([G[g + "_nactive"] for g in GROUPS if g.startswith("floor_")]).tolist())


@pytest.mark.parametrize("group", GROUPS)
def test_oracle_forward_matches_closed_form(group):
    o = Oracle(_env(group))
    for q, v, c, a, na in zip(G[group + "_q"], G[group + "_v"], G[group + "_c"], G[group + "_qacc"], G[group + "_nactive"]):
        f = o.forward(q, v, c)
        assert f["rc"] == 0
        assert _rel(f["qacc"], a) < 1e-9, (group, _rel(f["qacc"], a))
        assert int((f["efc_force"] > 0).sum()) == int(na) or group.startswith(("floor_", "multi_"))   # (n, n) rows tie
    if group == "fluid_swimmer":
        for q, v, c, Qf in zip(G[group + "_q"], G[group + "_v"], G[group + "_c"], G[group + "_qfrc_fluid"]):
            f = o.forward(q, v, c)
            damp = f["qfrc_passive"] - Qf                      # what is left of qfrc_passive is the joint dampers
            assert np.abs(damp).max() <= 1e-9 * (1 + np.abs(Qf).max())     # (swimmer.xml has none)


@pytest.mark.parametrize("group", GROUPS)
def test_kernel_simulator_forward_matches_closed_form(group):
    import __graft_entry__
    L = C.CDLL(__graft_entry__.build_host_twin())
    i = util.ENVS.index(_env(group))
    for q, v, c, a in zip(G[group + "_q"], G[group + "_v"], G[group + "_c"], G[group + "_qacc"]):
        nv = len(q)
        qa, fs, fc = np.zeros(nv), np.zeros(nv), np.zeros(nv)
        it, na = C.c_int(), C.c_int()
        q, v, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, c))
        rc = L.twin_forward(i, q.ctypes.data_as(dp), v.ctypes.data_as(dp), c.ctypes.data_as(dp), qa.ctypes.data_as(dp),
                            fs.ctypes.data_as(dp), fc.ctypes.data_as(dp), C.byref(it), C.byref(na))
        assert rc == 0
        assert _rel(qa, a) < 1e-9, (group, _rel(qa, a))


@pytest.mark.parametrize("group", GROUPS)
def test_oracle_and_kernel_simulator_step_match_closed_form(group):
    """One mj_step: implicit-damping Euler / RK4 with the constraint re-solved at every stage."""
    import __graft_entry__
    L = C.CDLL(__graft_entry__.build_host_twin())
    env = _env(group)
    o = Oracle(env)
    q, v, c = (np.ascontiguousarray(G[group + k]) for k in ("_q", "_v", "_c"))
    qo, vo, _, so = o.step(q, v, c, frame_skip=1)
    N = len(q)
    qn, vn, r = np.empty_like(q), np.empty_like(v), np.empty(N)
    st = np.zeros(N, dtype=np.uint32)
    L.twin_step(util.ENVS.index(env), 1, C.c_int64(N), q.ctypes.data_as(dp), v.ctypes.data_as(dp), c.ctypes.data_as(dp),
                qn.ctypes.data_as(dp), vn.ctypes.data_as(dp), r.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert int((so != 0).sum()) == 0 and int((st != 0).sum()) == 0
    for got_q, got_v in ((qo, vo), (qn, vn)):
        assert _rel(got_v, G[group + "_v1"]) < 1e-10 and _rel(got_q, G[group + "_q1"]) < 1e-11, group


@pytest.mark.gpu
@pytest.mark.parametrize("group", GROUPS)
def test_hip_step_kernel_matches_closed_form(group):
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    e = FdEngine(_env(group), frame_skip=1)
    qn, vn, r, st = e.step(G[group + "_q"], G[group + "_v"], G[group + "_c"])
    assert int((st != 0).sum()) == 0
    assert _rel(vn.cpu().numpy(), G[group + "_v1"]) < 1e-10 and _rel(qn.cpu().numpy(), G[group + "_q1"]) < 1e-11, group
