"""CPU: known-answer physics.  Closed-form cart-pole / double cart-pole accelerations derived with sympy
from the Lagrangian (tests/golden/gen_sympy_kat.py -- independent of the oracle and of the kernel) vs the
oracle's mj_forward and vs the kernel's per-lane simulator built for the host (SURVEY.md 8c pin (2))."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from oracle import Oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_cartpole.npz"))
CASES = (("InvertedPendulumEnv", "ip"), ("InvertedDoublePendulumEnv", "idp"))
dp = C.POINTER(C.c_double)


@pytest.mark.parametrize("env,pre", CASES)
def test_oracle_qacc_matches_closed_form(env, pre):
    o = Oracle(env)
    for q, v, c, a in zip(GOLD[pre + "_q"], GOLD[pre + "_v"], GOLD[pre + "_c"], GOLD[pre + "_qacc"]):
        f = o.forward(q, v, c)
        assert f["nefc"] == 0            # the vectors stay away from the joint limits
        assert np.abs(f["qacc"] - a).max() <= 1e-12 * (1 + np.abs(a).max())


@pytest.mark.parametrize("env,pre", CASES)
def test_kernel_simulator_qacc_matches_closed_form(env, pre):
    import __graft_entry__
    L = C.CDLL(__graft_entry__.build_host_twin())
    i = util.ENVS.index(env)
    for q, v, c, a in zip(GOLD[pre + "_q"], GOLD[pre + "_v"], GOLD[pre + "_c"], GOLD[pre + "_qacc"]):
        nv = len(q)
        qa, fs, fc = np.zeros(nv), np.zeros(nv), np.zeros(nv)
        it, na = C.c_int(), C.c_int()
        q, v, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, c))
        rc = L.twin_forward(i, q.ctypes.data_as(dp), v.ctypes.data_as(dp), c.ctypes.data_as(dp), qa.ctypes.data_as(dp),
                            fs.ctypes.data_as(dp), fc.ctypes.data_as(dp), C.byref(it), C.byref(na))
        assert rc == 0 and it.value == 0
        assert np.abs(qa - a).max() <= 1e-12 * (1 + np.abs(a).max())
