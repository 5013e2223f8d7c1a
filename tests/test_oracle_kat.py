"""CPU: known-answer physics (SURVEY.md 8c pin (2)), vs the oracle's mj_forward and vs the kernel's per-lane simulator
built for the host.  Two independent sources:
  * closed-form cart-pole / double cart-pole accelerations derived with sympy from the Lagrangian
    (tests/golden/gen_sympy_kat.py);
  * Hopper, Walker2d and HalfCheetah in free flight from an automatic-differentiation Lagrangian over plain forward
    kinematics (tests/golden/gen_lagrange_kat.py): mass matrix, Coriolis/centrifugal terms, gravity, joint springs,
    dampers, armature, actuator gears and the control clamp of the 6-, 9- and 9-dof trees."""
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


LAG = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_lagrange.npz"))
TREES = (("InvertedPendulumEnv", "invertedpendulum"), ("InvertedDoublePendulumEnv", "inverteddoublependulum"),
         ("HopperEnv", "hopper"), ("Walker2dEnv", "walker2d"), ("HalfCheetahEnv", "halfcheetah"))
RK4 = tuple(t for t in TREES if t[0] in ("InvertedPendulumEnv", "InvertedDoublePendulumEnv", "Walker2dEnv"))


@pytest.mark.parametrize("env,pre", TREES)
def test_oracle_matches_lagrangian_trees(env, pre):
    o = Oracle(env)
    for q, v, c, a, M in zip(LAG[pre + "_q"], LAG[pre + "_v"], LAG[pre + "_c"], LAG[pre + "_qacc"], LAG[pre + "_M"]):
        f = o.forward(q, v, c)
        assert f["nefc"] == 0            # free flight, inside the joint ranges
        assert np.abs(f["qacc"] - a).max() <= 1e-12 * (1 + np.abs(a).max())
        Mo = np.tril(f["M"]) + np.tril(f["M"], -1).T
        assert np.abs(Mo - M).max() <= 1e-13 * np.abs(M).max()


@pytest.mark.parametrize("env,pre", TREES)
def test_kernel_simulator_matches_lagrangian_trees(env, pre):
    """The kernel's simulator does not form qacc in an unconstrained Euler step (implicit joint damping folds it into
    the velocity update), so it is checked on what it does form: the smooth generalised force M qacc, and -- for the
    Euler models -- the velocities after one mj_step, v + h (M + h D)^-1 M qacc, against the Lagrangian's M and qacc."""
    import __graft_entry__
    import mbrl_amd  # noqa: F401
    from mbrl_amd import modelspec
    L = C.CDLL(__graft_entry__.build_host_twin())
    i = util.ENVS.index(env)
    m = modelspec.MODELS[env]
    damp = np.array([j["damping"] for b in m["bodies"] for j in b["joints"]])
    h = m["timestep"]
    for q, v, c, a, M in zip(LAG[pre + "_q"], LAG[pre + "_v"], LAG[pre + "_c"], LAG[pre + "_qacc"], LAG[pre + "_M"]):
        nv = len(q)
        qa, fs, fc = np.zeros(nv), np.zeros(nv), np.zeros(nv)
        it, na = C.c_int(), C.c_int()
        q, v, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (q, v, c))
        rc = L.twin_forward(i, q.ctypes.data_as(dp), v.ctypes.data_as(dp), c.ctypes.data_as(dp), qa.ctypes.data_as(dp),
                            fs.ctypes.data_as(dp), fc.ctypes.data_as(dp), C.byref(it), C.byref(na))
        assert rc == 0 and it.value == 0 and na.value == 0
        f = M @ a
        assert np.abs(fs - f).max() <= 1e-12 * (1 + np.abs(f).max())
        if m["integrator"] == "RK4":
            assert np.abs(qa - a).max() <= 1e-12 * (1 + np.abs(a).max())
            qn, vn, r = np.zeros((1, nv)), np.zeros((1, nv)), np.zeros(1)
            st = np.zeros(1, dtype=np.uint32)
            L.twin_step(i, 1, C.c_int64(1), q.ctypes.data_as(dp), v.ctypes.data_as(dp), c.ctypes.data_as(dp),
                        qn.ctypes.data_as(dp), vn.ctypes.data_as(dp), r.ctypes.data_as(dp),
                        st.ctypes.data_as(C.POINTER(C.c_uint32)))
            k = [np.array_equal(q, x) for x in LAG[pre + "_q"]].index(True)
            q1, v1 = LAG[pre + "_q1"][k], LAG[pre + "_v1"][k]        # classic RK4 over the Lagrangian's accelerations
            assert st[0] == 0
            assert np.abs(vn[0] - v1).max() <= 1e-11 * (1 + np.abs(v1).max())
            assert np.abs(qn[0] - q1).max() <= 1e-12 * (1 + np.abs(q1).max())
        else:
            qn, vn, r = np.zeros((1, nv)), np.zeros((1, nv)), np.zeros(1)
            st = np.zeros(1, dtype=np.uint32)
            L.twin_step(i, 1, C.c_int64(1), q.ctypes.data_as(dp), v.ctypes.data_as(dp), c.ctypes.data_as(dp),
                        qn.ctypes.data_as(dp), vn.ctypes.data_as(dp), r.ctypes.data_as(dp),
                        st.ctypes.data_as(C.POINTER(C.c_uint32)))
            v1 = v + h * np.linalg.solve(M + h * np.diag(damp), f)
            assert st[0] == 0
            assert np.abs(vn[0] - v1).max() <= 1e-12 * (1 + np.abs(v1).max())
            assert np.abs(qn[0] - (q + h * v1)).max() <= 1e-13 * (1 + np.abs(q).max())


@pytest.mark.parametrize("env,pre", RK4)
def test_oracle_rk4_step_matches_lagrangian(env, pre):
    o = Oracle(env)
    q, v, c = LAG[pre + "_q"], LAG[pre + "_v"], LAG[pre + "_c"]
    qn, vn, r, st = o.step(q, v, c, frame_skip=1)
    assert not st.any()
    assert np.abs(vn - LAG[pre + "_v1"]).max() <= 1e-11 * (1 + np.abs(LAG[pre + "_v1"]).max())
    assert np.abs(qn - LAG[pre + "_q1"]).max() <= 1e-12 * (1 + np.abs(LAG[pre + "_q1"]).max())


def test_oracle_swimmer_mass_matrix_and_bias_match_lagrangian():
    """Swimmer (x-y plane, hinges about +z): mass matrix and Coriolis/centrifugal bias; its fluid forces are MuJoCo's own
    model and stay outside this pin."""
    o = Oracle("SwimmerEnv")
    for q, v, M, b in zip(LAG["swimmer_q"], LAG["swimmer_v"], LAG["swimmer_M"], LAG["swimmer_bias"]):
        f = o.forward(q, v, np.zeros(2))
        Mo = np.tril(f["M"]) + np.tril(f["M"], -1).T
        assert np.abs(Mo - M).max() <= 1e-13 * np.abs(M).max()
        assert np.abs(f["qfrc_bias"] - b).max() <= 1e-12 * (1 + np.abs(b).max())


# Published values: `model.body_mass` of gym's Hopper-v2 and HalfCheetah-v2 under MuJoCo 2.0 (world body omitted), quoted
# from knowledge -- the reference repository holds no such numbers.  Hopper's are absolute (density 1000 capsules, which
# fixes MuJoCo 2.0's capsule volume rule pi r^2 (h + r)); HalfCheetah's are the same rule normalised to settotalmass 14.
GYM_BODY_MASS = {
    "HopperEnv": [3.53429174, 3.92699082, 2.71433605, 5.0893801],
    "HalfCheetahEnv": [6.36031332, 1.53524804, 1.58093995, 1.0691906, 1.42558747, 1.17885117, 0.84986945],
}


@pytest.mark.parametrize("env", sorted(GYM_BODY_MASS))
def test_body_masses_match_published_mujoco_values(env):
    import mbrl_amd  # noqa: F401
    from mbrl_amd import compile_planar
    want = np.array(GYM_BODY_MASS[env])
    o = Oracle(env)
    got_oracle = np.array([o.body_consts(b)[0] for b in range(len(want))])
    got_kernel = np.array(compile_planar.PlanarModel(env).mass, dtype=float)     # the tables the HIP kernel is built from
    assert np.abs(got_oracle - want).max() < 5e-9
    assert np.abs(got_kernel - want).max() < 5e-9


def test_oracle_counts_units_that_take_the_crossing_axes_rule():
    """The crossing-capsule-axes rule is "parity unpinned": the oracle reports which units' rollouts met
    it, so that parity figures can state how many of their units rest on it.  Only Hopper's MJCF has capsule pairs."""
    import util
    from oracle import Oracle
    o = Oracle("HopperEnv")
    q, v, c = util.on_trajectory(o, "HopperEnv", 256, seed=0, max_steps=39)
    out = o.fd_jacobians(q, v, c, nthreads=8, info=True)
    x = out["evals_crossing"]
    assert (x >= 0).all() and (x > 0).any()                     # Hopper falls and folds through itself within the horizon
    print("Hopper: %d of %d on-trajectory units took the crossing-axes rule" % (int((x > 0).sum()), len(x)))
    h = Oracle("HalfCheetahEnv")
    q, v, c = util.on_trajectory(h, "HalfCheetahEnv", 64, seed=0, max_steps=20)
    assert not h.fd_jacobians(q, v, c, nthreads=8, info=True)["evals_crossing"].any()
