"""The env reward formulas restated a THIRD time, in numpy, straight from the reference's env
files, and compared with what the oracle, the kernel's simulator and -- marked gpu -- the HIP step kernel return for the
same transition.  The transition itself (s -> s') is taken from the implementation under test; what is checked is
r(s, a, s').  File:line of every formula is given where it is restated."""
import math

import numpy as np
import pytest

import util
from oracle import Oracle

FRAME_SKIP = {"InvertedPendulumEnv": 2, "InvertedDoublePendulumEnv": 4, "SwimmerEnv": 10, "HopperEnv": 20,
              "Walker2dEnv": 25, "HalfCheetahEnv": 10}            # configs/*.yaml:7
TIMESTEP = {"InvertedPendulumEnv": 0.02, "InvertedDoublePendulumEnv": 0.01, "SwimmerEnv": 0.01, "HopperEnv": 0.005,
            "Walker2dEnv": 0.002, "HalfCheetahEnv": 0.01}          # <option timestep=...> of mujoco/assets/*.xml


def reference_reward(env, q, a, q1, v1, site=None):
    dt = TIMESTEP[env] * FRAME_SKIP[env]                            # gym MujocoEnv.dt = timestep * frame_skip
    if env == "HalfCheetahEnv":                                     # half_cheetah.py:25-31
        return -0.01 * np.square(a).sum() + (q1[0] - q[0]) / dt
    if env == "SwimmerEnv":                                         # swimmer.py:23-29
        return (q1[0] - q[0]) / dt - 0.0001 * np.square(a).sum()
    if env == "HopperEnv":                                          # hopper.py:27-41
        sig = lambda x: (1 + 200 ** (-x + 0.5)) ** (-1)              # sigmoid(x, 0, 1)
        ang_abs = abs(q1[2]) % (2 * math.pi)
        if ang_abs > math.pi:
            ang_abs = 2 * math.pi - ang_abs
        return (q1[0] - q[0]) / dt + sig(q1[1] / 1.25) + sig((math.pi - ang_abs) / math.pi) - 1e-3 * np.square(a).sum()
    if env == "Walker2dEnv":                                        # walker2d.py:23-30 (the max(..., 1) is literal)
        coeff = min(max(q1[1] / 0.8, 0), 1) * 0.5 + max((math.pi - abs(q1[2])) / math.pi, 1) * 0.5
        return (q1[0] - q[0]) / dt + coeff - 1e-3 * np.square(a).sum()
    if env == "InvertedPendulumEnv":                                # inverted_pendulum.py:26-35 (x and y both cos)
        y = 0.6 * np.cos(q1[1]); x = 0.6 * np.cos(q1[1])
        return -(0.01 * x ** 2 + (y - 1) ** 2) - 0.001 * a[0] ** 2
    if env == "InvertedDoublePendulumEnv":                          # inverted_double_pendulum.py:26-33
        x, y = site                                                 # data.site_xpos[0]: see _idp_site
        return -(0.01 * x ** 2 + (y - 2) ** 2) - (1e-3 * v1[1] ** 2 + 5e-3 * v1[2] ** 2)
    raise KeyError(env)


def _idp_site(o, q, v, c):
    """data.site_xpos after do_simulation is NOT the tip at the new qpos: mj_step does not recompute kinematics after it
    integrates, so it is the tip at the state of the LAST dynamics evaluation -- RK4's fourth stage of the last mj_step
    (MuJoCo behaviour from knowledge).  Tip (inverted_double_pendulum.xml:18-24: pole2 at z = 0.6 of
    pole, site at z = 0.6 of pole2, hinges about +y): x = cart + 0.6 sin(t1) + 0.6 sin(t1 + t2), z likewise with cos."""
    h = TIMESTEP["InvertedDoublePendulumEnv"]
    for _ in range(FRAME_SKIP["InvertedDoublePendulumEnv"] - 1):
        qn, vn, _, _ = o.step(q, v, c, frame_skip=1)
        q, v = qn[0], vn[0]
    f = lambda qq, vv: o.forward(qq, vv, c)["qacc"]
    k1v = f(q, v)
    q2, v2 = q + 0.5 * h * v, v + 0.5 * h * k1v
    k2v = f(q2, v2)
    q3, v3 = q + 0.5 * h * v2, v + 0.5 * h * k2v
    k3v = f(q3, v3)
    q4 = q + h * v3                                                  # fourth-stage position
    return q4[0] + 0.6 * math.sin(q4[1]) + 0.6 * math.sin(q4[1] + q4[2]), 0.6 * math.cos(q4[1]) + 0.6 * math.cos(q4[1] + q4[2])


def _cases(env, o, n=24):
    q, v, c = util.on_trajectory(o, env, n, seed=11, max_steps=10)
    c = c * 1.4                                                     # some controls beyond ctrlrange: the cost uses the raw action
    return q, v, c


@pytest.mark.parametrize("env", util.ENVS)
def test_oracle_and_kernel_simulator_rewards_match_the_reference_formulas(env):
    from cpu_engine import CpuEngine
    o = Oracle(env)
    q, v, c = _cases(env, o)
    for name, (q1, v1, r, st) in (("oracle", o.step(q, v, c)), ("kernel simulator", CpuEngine(env)._step_np(q, v, c))):
        assert (st == 0).all()
        for k in range(len(q)):
            site = _idp_site(o, q[k], v[k], c[k]) if env == "InvertedDoublePendulumEnv" else None
            want = reference_reward(env, q[k], c[k], q1[k], v1[k], site)
            assert abs(r[k] - want) <= 1e-9 * (1 + abs(want)), (env, name, k, r[k], want)


@pytest.mark.gpu
@pytest.mark.parametrize("env", util.ENVS)
def test_hip_step_rewards_match_the_reference_formulas(env):
    import mbrl_amd  # noqa: F401
    from mbrl_amd.engine import FdEngine
    o = Oracle(env)
    q, v, c = _cases(env, o)
    qn, vn, r, st = (t.cpu().numpy() for t in FdEngine(env).step(q, v, c))
    assert (st == 0).all()
    for k in range(len(q)):
        site = _idp_site(o, q[k], v[k], c[k]) if env == "InvertedDoublePendulumEnv" else None
        want = reference_reward(env, q[k], c[k], qn[k], vn[k], site)
        assert abs(r[k] - want) <= 1e-9 * (1 + abs(want)), (env, k, r[k], want)
