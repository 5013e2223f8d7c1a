"""CPU: the kernel's per-lane simulator (csrc/planar_lean.hpp, built for the host by
tests/host_twin.cpp) against the 3-D oracle on identical inputs -- the same check the GPU parity
tests make through the C ABI, runnable without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "_build", "libhost_twin.so")
dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint32)


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def twin():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build_host_twin()
    L = C.CDLL(LIB)
    L.twin_count_flops.restype = C.c_longlong
    return L


def twin_step(L, env, q, v, c):
    i = util.ENVS.index(env)
    qn, vn, r = np.empty_like(q), np.empty_like(v), np.empty(len(q))
    st = np.zeros(len(q), dtype=np.uint32)
    assert L.twin_step(i, 0, C.c_int64(len(q)), _p(q), _p(v), _p(c), _p(qn), _p(vn), _p(r), st.ctypes.data_as(up)) == 0
    return qn, vn, r, st


@pytest.mark.parametrize("env", util.ENVS)
def test_twin_step_matches_oracle(env, twin):
    o = Oracle(env)
    q, v, c = util.on_trajectory(o, env, 400, seed=5)
    q, v, c = (np.ascontiguousarray(x) for x in (q, v, c))
    qo, vo, ro, so = o.step(q, v, c, nthreads=8)
    qn, vn, r, st = twin_step(twin, env, q, v, c)
    assert (st == so).all(), (np.flatnonzero(st != so)[:5], st[st != so][:5], so[st != so][:5])
    ok = so == 0
    assert ok.mean() > 0.8
    assert util.relerr(qn[ok], qo[ok]) < 1e-9
    assert util.relerr(vn[ok], vo[ok]) < 1e-8
    assert util.relerr(r[ok], ro[ok]) < 1e-8


@pytest.mark.parametrize("env", ("HopperEnv", "Walker2dEnv", "HalfCheetahEnv"))
def test_contact_slots_suffice_on_trajectory_states(env, twin):
    """The fast kernel keeps ACTIVE floor contacts in K = 8 compacted slots; units with more take the fix-up launch
    (tests/test_gpu_overflow.py), which these on-trajectory samples never need."""
    o = Oracle(env)
    q, v, c = util.on_trajectory(o, env, 2000, seed=6, max_steps=40)
    i = util.ENVS.index(env)
    nv = o.nv
    worst = 0
    for k in range(len(q)):
        qa, fs, fc = np.zeros(nv), np.zeros(nv), np.zeros(nv)
        it, na = C.c_int(), C.c_int()
        rc = twin.twin_forward(i, _p(np.ascontiguousarray(q[k])), _p(np.ascontiguousarray(v[k])),
                               _p(np.ascontiguousarray(c[k])), _p(qa), _p(fs), _p(fc), C.byref(it), C.byref(na))
        assert not (rc & 4), "contact slot overflow at unit %d" % k
        worst = max(worst, na.value)
    assert worst <= 8
