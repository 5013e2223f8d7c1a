"""CPU: physical invariants of the oracle's mj_step (SURVEY.md 8c pin (3)), through its test-only option hooks:
  * energy: with joint damping and actuation off, classic RK4 conserves kinetic + potential energy to O(h^4) per step
    (InvertedDoublePendulum, Walker2d in free flight); with damping on it decreases;
  * momentum: Swimmer with density = viscosity = 0 and no armature is an isolated planar system -- joint motors,
    dampers and joint limits are internal -- so linear momentum (x, y) and angular momentum about z are conserved under
    any control.
These hold for any correct rigid-body integrator, whatever MuJoCo's exact arithmetic is; they do not involve the kernel
(its model constants are compile-time: no hooks)."""
import numpy as np
import pytest

from oracle import Oracle


@pytest.fixture
def reset_options():
    made = []
    yield made
    for o in made:
        o.set_option("reset")


def _roll(o, q, v, c, steps, free=False):
    E, P = [], []
    for _ in range(steps):
        f = o.forward(q, v, c)
        assert not free or f["nefc"] == 0, "a joint limit or contact came into play (soft constraints dissipate)"
        E.append(f["energy"].sum()); P.append(f["momentum"].copy())
        qn, vn, _, st = o.step(q, v, c, frame_skip=1)
        assert st[0] == 0
        q, v = qn[0], vn[0]
    return np.array(E), np.array(P), q, v


@pytest.mark.parametrize("env,lift", [("InvertedDoublePendulumEnv", 0.0), ("Walker2dEnv", 5.0)])
def test_rk4_conserves_energy_without_damping_and_actuation(env, lift, reset_options):
    o = Oracle(env)
    reset_options.append(o)
    rng = np.random.default_rng(0)
    q = rng.uniform(-0.3, 0.3, o.nq)
    if lift:
        q[1] += lift                                   # far above the floor: no contact
        q[3:] = -np.abs(q[3:])                         # inside the joint ranges
    v = 0.5 * rng.standard_normal(o.nv)
    v[0] = 0.0 if not lift else v[0]                   # the cart must stay inside its +-1 range for the whole roll
    c = np.zeros(o.nu)
    o.set_option("damping_scale", 0.0)
    E, _, _, _ = _roll(o, q, v, c, 60, free=True)
    scale = np.abs(E).max() + 1.0
    assert np.abs(E - E[0]).max() < 2e-6 * scale, np.abs(E - E[0]).max() / scale   # RK4 truncation, h = 0.01 / 0.002
    if o.forward(q, v, c)["qfrc_passive"].any() or env == "Walker2dEnv":
        o.set_option("reset")
        o.set_option("damping_scale", 1.0)
        Ed, _, _, _ = _roll(o, q, v, c, 60)
        assert Ed[-1] < Ed[0] - 1e-5 * scale            # the dampers dissipate


def test_swimmer_conserves_momentum_without_fluid(reset_options):
    o = Oracle("SwimmerEnv")
    reset_options.append(o)
    for k in ("density", "viscosity", "armature_scale"):
        o.set_option(k, 0.0)
    rng = np.random.default_rng(1)
    q = rng.uniform(-0.5, 0.5, o.nq)
    v = 0.5 * rng.standard_normal(o.nv)
    c = rng.uniform(-0.2, 0.2, o.nu)                    # motors are internal torques
    # (momentum is linear in v but not in q, so RK4 conserves it to its truncation error only; the roll stays clear of
    # the joint limits, whose stiff -- equally internal -- forces would add integration error, not physics)
    _, P, _, _ = _roll(o, q, v, c, 60, free=True)
    lin, ang = P[:, :2], P[:, 5]
    assert np.abs(lin - lin[0]).max() < 1e-7 * (1 + np.abs(lin).max())
    assert np.abs(ang - ang[0]).max() < 1e-7 * (1 + np.abs(ang).max())
    # and with the fluid switched back on, momentum is NOT conserved (the test can fail)
    o.set_option("reset")
    o.set_option("armature_scale", 0.0)
    _, P2, _, _ = _roll(o, q, v, c, 50)
    assert np.abs(P2[:, :2] - P2[0, :2]).max() > 1e-3
