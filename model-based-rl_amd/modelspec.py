"""Declarative transcription of the six in-scope MJCF models.

Each entry restates, in this project's own schema, the *data* of one MJCF file of the
reference (``mujoco/assets/*.xml``) plus the ``FRAME_SKIP`` of its yaml
(``configs/*.yaml:7``).  Class defaults are resolved by hand, degrees are converted to
radians, and ``coordinate="global"`` models (hopper, walker2d) are converted to local
body frames by subtraction (no body in any of the six files carries a rotation).  No
derived quantity (mass, inertia, invweight, ...) is stored here: the oracle (C, 3-D)
and the HIP path (planar tables, ``compile_planar.py``) each derive theirs
independently from this transcription, so a mistake in either derivation shows up as a
parity failure.

Conventions
-----------
* bodies are listed parent-before-child, ``parent=-1`` is the world body;
* body ``pos`` / joint ``pos`` / geom end points ``a``,``b`` are LOCAL to the body
  frame (all body frames are axis-aligned with the world at qpos0);
* joints: ``type`` in {"slide","hinge"}; ``range`` in metres / radians; ``ref`` is
  MuJoCo's ``ref`` attribute (qpos0);
* geoms are capsules given by segment end points ``a``,``b`` and ``radius``;
* ``floor`` is the z-height plane geom of the world body (None = no collision at all).

MuJoCo 2.0 built-in defaults used where a file is silent [from knowledge]:
density 1000, friction 1, condim 3, contype/conaffinity 1, margin 0, gap 0,
solref (0.02, 1), solimp (0.9, 0.95, 0.001, 0.5, 2), limited false, ctrllimited false,
armature/damping/stiffness 0, gravity (0, 0, -9.81), integrator Euler.
"""
import math

D2R = math.pi / 180.0
SOLREF0 = (0.02, 1.0)
SOLIMP0 = (0.9, 0.95, 0.001, 0.5, 2.0)


def _solimp(*v):
    """3-number solimp -> 5 numbers (midpoint 0.5, power 2 are the 2.0 defaults)."""
    v = tuple(float(x) for x in v)
    return v + SOLIMP0[len(v):]


def _joint(name, type, axis, pos=(0, 0, 0), ref=0.0, stiffness=0.0, damping=0.0, armature=0.0,
           limited=False, range=(0.0, 0.0), margin=0.0, solref=SOLREF0, solimp=SOLIMP0):
    return dict(name=name, type=type, axis=tuple(map(float, axis)), pos=tuple(map(float, pos)),
                ref=float(ref), stiffness=float(stiffness), damping=float(damping),
                armature=float(armature), limited=bool(limited),
                range=(float(range[0]), float(range[1])), margin=float(margin),
                solref=tuple(solref), solimp=tuple(solimp))


def _geom(name, a, b, radius, density=1000.0, friction=1.0, condim=3, contype=1, conaffinity=1,
          margin=0.0, solref=SOLREF0, solimp=SOLIMP0):
    return dict(name=name, a=tuple(map(float, a)), b=tuple(map(float, b)), radius=float(radius),
                density=float(density), friction=float(friction), condim=int(condim),
                contype=int(contype), conaffinity=int(conaffinity), margin=float(margin),
                solref=tuple(solref), solimp=tuple(solimp))


def _yrot_seg(pos, angle, halflen):
    """Capsule given as pos + axisangle about +y + half-length -> end points.

    The capsule axis is the geom's local z; rotating (0,0,1) about +y by `angle`
    gives (sin, 0, cos)."""
    d = (math.sin(angle), 0.0, math.cos(angle))
    a = tuple(pos[i] - halflen * d[i] for i in range(3))
    b = tuple(pos[i] + halflen * d[i] for i in range(3))
    return a, b


def _sub(p, q):
    return tuple(p[i] - q[i] for i in range(3))


# --------------------------------------------------------------------------- #
# InvertedPendulum  (mujoco/assets/inverted_pendulum.xml, configs/inverted_pendulum.yaml)
# --------------------------------------------------------------------------- #
def _inverted_pendulum():
    # default joint: armature 0, damping 1, limited true            (xml:4)
    # default geom : contype 0, friction 1                          (xml:5)
    # motor: ctrlrange -3 3 but NOT ctrllimited -> no clamp         (xml:7)
    g = dict(contype=0, conaffinity=1, friction=1.0)
    bodies = [
        dict(name="cart", parent=-1, pos=(0, 0, 0),                                  # xml:14
             joints=[_joint("slider", "slide", (1, 0, 0), damping=1.0, limited=True,
                            range=(-1.0, 1.0))],                                     # xml:15
             # quat 0.707 0 0.707 0 (normalised) = +90deg about y: capsule axis = +x (xml:16)
             geoms=[_geom("cart", (-0.1, 0, 0), (0.1, 0, 0), 0.1, **g)]),
        dict(name="pole", parent=0, pos=(0, 0, 0),                                   # xml:17
             joints=[_joint("hinge", "hinge", (0, 1, 0), damping=1.0, limited=True,
                            range=(-90 * D2R, 90 * D2R))],                           # xml:18
             geoms=[_geom("cpole", (0, 0, 0), (0.001, 0, 0.6), 0.049, **g)]),        # xml:19
    ]
    return dict(env="InvertedPendulumEnv", src="mujoco/assets/inverted_pendulum.xml",
                plane="xz", timestep=0.02, integrator="RK4", gravity=(0.0, 0.0, -9.81),  # xml:9
                density=0.0, viscosity=0.0, frame_skip=2, settotalmass=None,
                bodies=bodies, floor=None, selfcollide=False,
                actuators=[dict(joint="slider", gear=100.0, ctrllimited=False,
                                ctrlrange=(-3.0, 3.0))],                             # xml:25
                site=None, reward="inverted_pendulum")


# --------------------------------------------------------------------------- #
# InvertedDoublePendulum  (inverted_double_pendulum.xml; FRAME_SKIP 4 from the yaml)
# --------------------------------------------------------------------------- #
def _inverted_double_pendulum():
    # default joint: damping 0.05 (xml:7); default geom contype 0 (xml:8) -> no contacts
    g = dict(contype=0, conaffinity=1, friction=1.0)
    bodies = [
        dict(name="cart", parent=-1, pos=(0, 0, 0),                                  # xml:15
             joints=[_joint("slider", "slide", (1, 0, 0), damping=0.05, limited=True,
                            range=(-1.0, 1.0), margin=0.01)],                        # xml:16
             geoms=[_geom("cart", (-0.1, 0, 0), (0.1, 0, 0), 0.1, **g)]),            # xml:17
        dict(name="pole", parent=0, pos=(0, 0, 0),                                   # xml:18
             joints=[_joint("hinge", "hinge", (0, 1, 0), damping=0.05)],             # xml:19
             geoms=[_geom("cpole", (0, 0, 0), (0, 0, 0.6), 0.045, **g)]),            # xml:20
        dict(name="pole2", parent=1, pos=(0, 0, 0.6),                                # xml:21
             joints=[_joint("hinge2", "hinge", (0, 1, 0), damping=0.05)],            # xml:22
             geoms=[_geom("cpole2", (0, 0, 0), (0, 0, 0.6), 0.045, **g)]),           # xml:23
    ]
    return dict(env="InvertedDoublePendulumEnv",
                src="mujoco/assets/inverted_double_pendulum.xml",
                plane="xz", timestep=0.01, integrator="RK4", gravity=(1e-5, 0.0, -9.81),  # xml:10
                density=0.0, viscosity=0.0, frame_skip=4, settotalmass=None,
                bodies=bodies, floor=None, selfcollide=False,
                actuators=[dict(joint="slider", gear=500.0, ctrllimited=True,
                                ctrlrange=(-1.0, 1.0))],                             # xml:30
                site=dict(body=2, pos=(0.0, 0.0, 0.6)),                              # xml:24
                reward="inverted_double_pendulum")


# --------------------------------------------------------------------------- #
# Swimmer  (swimmer.xml): motion in the x-y plane, hinges about +z, fluid forces
# --------------------------------------------------------------------------- #
def _swimmer():
    # default joint armature 0.1 (xml:6); collision="predefined" with no pairs -> no contacts (xml:3)
    g = dict(density=1000.0, contype=1, conaffinity=1, condim=1)
    bodies = [
        dict(name="torso", parent=-1, pos=(0, 0, 0),                                 # xml:19
             joints=[_joint("slider1", "slide", (1, 0, 0), armature=0.1),            # xml:22
                     _joint("slider2", "slide", (0, 1, 0), armature=0.1),            # xml:23
                     _joint("rot", "hinge", (0, 0, 1), armature=0.1)],               # xml:24
             geoms=[_geom("torso", (1.5, 0, 0), (0.5, 0, 0), 0.1, **g)]),            # xml:21
        dict(name="mid", parent=0, pos=(0.5, 0, 0),                                  # xml:25
             joints=[_joint("rot2", "hinge", (0, 0, 1), armature=0.1, limited=True,
                            range=(-100 * D2R, 100 * D2R))],                         # xml:27
             geoms=[_geom("mid", (0, 0, 0), (-1, 0, 0), 0.1, **g)]),                 # xml:26
        dict(name="back", parent=1, pos=(-1, 0, 0),                                  # xml:28
             joints=[_joint("rot3", "hinge", (0, 0, 1), armature=0.1, limited=True,
                            range=(-100 * D2R, 100 * D2R))],                         # xml:30
             geoms=[_geom("back", (0, 0, 0), (-1, 0, 0), 0.1, **g)]),                # xml:29
    ]
    return dict(env="SwimmerEnv", src="mujoco/assets/swimmer.xml",
                plane="xy", timestep=0.01, integrator="RK4", gravity=(0.0, 0.0, -9.81),
                density=4000.0, viscosity=0.1, frame_skip=10, settotalmass=None,     # xml:3
                bodies=bodies, floor=None, selfcollide=False,
                actuators=[dict(joint="rot2", gear=150.0, ctrllimited=True, ctrlrange=(-1.0, 1.0)),
                           dict(joint="rot3", gear=150.0, ctrllimited=True, ctrlrange=(-1.0, 1.0))],
                site=None, reward="swimmer")


# --------------------------------------------------------------------------- #
# Hopper / Walker2d  (coordinate="global": converted to local frames here)
# --------------------------------------------------------------------------- #
def _leg_chain(suffix, torso_g, foot_from_x, foot_to_x, foot_friction, jd, gd, foot_name_pos_x):
    """thigh/leg/foot chain shared by hopper.xml:17-27 and walker2d.xml:17-41.

    Global positions from the files: thigh body (0,0,1.05), leg body (0,0,0.35),
    foot body (x,0,0.1) [the files' literal "0.13/2" / "0.2/2" only moves the foot's
    frame origin, which is dynamically irrelevant under inertiafromgeom]."""
    thigh_g = (0.0, 0.0, 1.05)
    leg_g = (0.0, 0.0, 0.35)
    foot_g = (foot_name_pos_x, 0.0, 0.1)
    lim = dict(limited=True, **jd)
    return [
        dict(name="thigh" + suffix, parent=None, pos=_sub(thigh_g, torso_g),
             joints=[_joint("thigh%s_joint" % suffix, "hinge", (0, -1, 0),
                            pos=_sub((0, 0, 1.05), thigh_g), range=(-150 * D2R, 0.0), **lim)],
             geoms=[_geom("thigh%s_geom" % suffix, _sub((0, 0, 1.05), thigh_g),
                          _sub((0, 0, 0.6), thigh_g), 0.05, friction=0.9, **gd)]),
        dict(name="leg" + suffix, parent=None, pos=_sub(leg_g, thigh_g),
             joints=[_joint("leg%s_joint" % suffix, "hinge", (0, -1, 0),
                            pos=_sub((0, 0, 0.6), leg_g), range=(-150 * D2R, 0.0), **lim)],
             geoms=[_geom("leg%s_geom" % suffix, _sub((0, 0, 0.6), leg_g),
                          _sub((0, 0, 0.1), leg_g), 0.04, friction=0.9, **gd)]),
        dict(name="foot" + suffix, parent=None, pos=_sub(foot_g, leg_g),
             joints=[_joint("foot%s_joint" % suffix, "hinge", (0, -1, 0),
                            pos=_sub((0, 0, 0.1), foot_g), range=(-45 * D2R, 45 * D2R), **lim)],
             geoms=[_geom("foot%s_geom" % suffix, _sub((foot_from_x, 0, 0.1), foot_g),
                          _sub((foot_to_x, 0, 0.1), foot_g), 0.06, friction=foot_friction, **gd)]),
    ]


def _hopper():
    # default joint: armature 1, damping 1, limited true                      (xml:4)
    # default geom : conaffinity 1, condim 1, contype 1, margin .001,
    #                solimp .8 .8 .01, solref .05 1 (applies to the floor too) (xml:5)
    jd = dict(armature=1.0, damping=1.0)
    gd = dict(condim=1, contype=1, conaffinity=1, margin=0.001,
              solref=(0.05, 1.0), solimp=_solimp(0.8, 0.8, 0.01))
    torso_g = (0.0, 0.0, 1.25)
    bodies = [
        dict(name="torso", parent=-1, pos=torso_g,                                   # xml:12
             joints=[_joint("rootx", "slide", (1, 0, 0), pos=_sub((0, 0, 0), torso_g)),        # xml:13
                     _joint("rootz", "slide", (0, 0, 1), pos=_sub((0, 0, 0), torso_g), ref=1.25),  # xml:14
                     _joint("rooty", "hinge", (0, 1, 0), pos=_sub((0, 0, 1.25), torso_g))],    # xml:15
             geoms=[_geom("torso_geom", _sub((0, 0, 1.45), torso_g), _sub((0, 0, 1.05), torso_g),
                          0.05, friction=0.9, **gd)]),                               # xml:16
    ]
    chain = _leg_chain("", torso_g, -0.13, 0.26, 2.0, jd, gd, 0.065)                 # xml:17-27
    chain[0]["parent"], chain[1]["parent"], chain[2]["parent"] = 0, 1, 2
    bodies += chain
    floor = dict(z=0.0, friction=1.0, condim=3, contype=1, conaffinity=1, margin=0.001,
                 solref=(0.05, 1.0), solimp=_solimp(0.8, 0.8, 0.01))                 # xml:11 (+class default)
    act = [dict(joint=j, gear=200.0, ctrllimited=True, ctrlrange=(-1.0, 1.0))
           for j in ("thigh_joint", "leg_joint", "foot_joint")]                      # xml:32-34
    return dict(env="HopperEnv", src="mujoco/assets/hopper.xml",
                plane="xz", timestep=0.005, integrator="Euler", gravity=(0.0, 0.0, -9.81),  # xml:8
                density=0.0, viscosity=0.0, frame_skip=20, settotalmass=None,
                bodies=bodies, floor=floor, selfcollide=True,
                actuators=act, site=None, reward="hopper")


def _walker2d():
    # default joint: armature .01, damping .1, limited true                     (xml:4)
    # default geom : conaffinity 0, condim 3, contype 1, friction .7            (xml:5)
    jd = dict(armature=0.01, damping=0.1)
    gd = dict(condim=3, contype=1, conaffinity=0)
    torso_g = (0.0, 0.0, 1.25)
    bodies = [
        dict(name="torso", parent=-1, pos=torso_g,                                   # xml:11
             joints=[_joint("rootx", "slide", (1, 0, 0), pos=_sub((0, 0, 0), torso_g)),
                     _joint("rootz", "slide", (0, 0, 1), pos=_sub((0, 0, 0), torso_g), ref=1.25),
                     _joint("rooty", "hinge", (0, 1, 0), pos=_sub((0, 0, 1.25), torso_g))],    # xml:13-15
             geoms=[_geom("torso_geom", _sub((0, 0, 1.45), torso_g), _sub((0, 0, 1.05), torso_g),
                          0.05, friction=0.9, **gd)]),                               # xml:16
    ]
    right = _leg_chain("", torso_g, 0.0, 0.2, 0.9, jd, gd, 0.1)                      # xml:17-28
    right[0]["parent"], right[1]["parent"], right[2]["parent"] = 0, 1, 2
    left = _leg_chain("_left", torso_g, 0.0, 0.2, 1.9, jd, gd, 0.1)                  # xml:30-41
    left[0]["parent"], left[1]["parent"], left[2]["parent"] = 0, 4, 5
    bodies += right + left
    floor = dict(z=0.0, friction=0.7, condim=3, contype=1, conaffinity=1, margin=0.0,
                 solref=SOLREF0, solimp=SOLIMP0)                                     # xml:10 (+class default)
    act = [dict(joint=j, gear=100.0, ctrllimited=True, ctrlrange=(-1.0, 1.0))
           for j in ("thigh_joint", "leg_joint", "foot_joint",
                     "thigh_left_joint", "leg_left_joint", "foot_left_joint")]       # xml:46-51
    return dict(env="Walker2dEnv", src="mujoco/assets/walker2d.xml",
                plane="xz", timestep=0.002, integrator="RK4", gravity=(0.0, 0.0, -9.81),  # xml:7
                density=0.0, viscosity=0.0, frame_skip=25, settotalmass=None,
                bodies=bodies, floor=floor, selfcollide=False,
                actuators=act, site=None, reward="walker2d")


# --------------------------------------------------------------------------- #
# HalfCheetah  (half_cheetah.xml; angle="radian", settotalmass 14)
# --------------------------------------------------------------------------- #
def _half_cheetah():
    # default joint: armature .1 damping .01 limited true solimplimit 0 .8 .03
    #                solreflimit .03 1 stiffness 8                                (xml:38)
    # default geom : conaffinity 0 condim 3 contype 1 friction .4
    #                solimp 0 .8 .01 solref .03 1                                 (xml:39)
    gd = dict(condim=3, contype=1, conaffinity=0, friction=0.4,
              solref=(0.03, 1.0), solimp=_solimp(0.0, 0.8, 0.01))

    def leg_joint(name, damping, rng, stiffness):
        return _joint(name, "hinge", (0, 1, 0), stiffness=stiffness, damping=damping, armature=0.1,
                      limited=True, range=rng, solref=(0.03, 1.0), solimp=_solimp(0.0, 0.8, 0.03))

    def cap(name, pos, ang, halflen):
        a, b = _yrot_seg(pos, ang, halflen)
        return _geom(name, a, b, 0.046, **gd)

    bodies = [
        dict(name="torso", parent=-1, pos=(0, 0, 0.7),                               # xml:54
             joints=[_joint("rootx", "slide", (1, 0, 0)),                            # xml:56
                     _joint("rootz", "slide", (0, 0, 1)),                            # xml:57
                     _joint("rooty", "hinge", (0, 1, 0))],                           # xml:58
             geoms=[_geom("torso", (-0.5, 0, 0), (0.5, 0, 0), 0.046, **gd),          # xml:59
                    cap("head", (0.6, 0, 0.1), 0.87, 0.15)]),                        # xml:60
        dict(name="bthigh", parent=0, pos=(-0.5, 0, 0),                              # xml:62
             joints=[leg_joint("bthigh", 6.0, (-0.52, 1.05), 240.0)],                # xml:63
             geoms=[cap("bthigh", (0.1, 0, -0.13), -3.8, 0.145)]),                   # xml:64
        dict(name="bshin", parent=1, pos=(0.16, 0, -0.25),                           # xml:65
             joints=[leg_joint("bshin", 4.5, (-0.785, 0.785), 180.0)],               # xml:66
             geoms=[cap("bshin", (-0.14, 0, -0.07), -2.03, 0.15)]),                  # xml:67
        dict(name="bfoot", parent=2, pos=(-0.28, 0, -0.14),                          # xml:68
             joints=[leg_joint("bfoot", 3.0, (-0.4, 0.785), 120.0)],                 # xml:69
             geoms=[cap("bfoot", (0.03, 0, -0.097), -0.27, 0.094)]),                 # xml:70
        dict(name="fthigh", parent=0, pos=(0.5, 0, 0),                               # xml:74
             joints=[leg_joint("fthigh", 4.5, (-1.0, 0.7), 180.0)],                  # xml:75
             geoms=[cap("fthigh", (-0.07, 0, -0.12), 0.52, 0.133)]),                 # xml:76
        dict(name="fshin", parent=4, pos=(-0.14, 0, -0.24),                          # xml:77
             joints=[leg_joint("fshin", 3.0, (-1.2, 0.87), 120.0)],                  # xml:78
             geoms=[cap("fshin", (0.065, 0, -0.09), -0.6, 0.106)]),                  # xml:79
        dict(name="ffoot", parent=5, pos=(0.13, 0, -0.18),                           # xml:80
             joints=[leg_joint("ffoot", 1.5, (-0.5, 0.5), 60.0)],                    # xml:81
             geoms=[cap("ffoot", (0.045, 0, -0.07), -0.6, 0.07)]),                   # xml:82
    ]
    floor = dict(z=0.0, friction=0.4, condim=3, contype=1, conaffinity=1, margin=0.0,
                 solref=(0.03, 1.0), solimp=_solimp(0.0, 0.8, 0.01))                 # xml:53 (+class default)
    gears = (("bthigh", 120.0), ("bshin", 90.0), ("bfoot", 60.0),
             ("fthigh", 120.0), ("fshin", 60.0), ("ffoot", 30.0))                    # xml:89-94
    act = [dict(joint=j, gear=g, ctrllimited=True, ctrlrange=(-1.0, 1.0)) for j, g in gears]
    return dict(env="HalfCheetahEnv", src="mujoco/assets/half_cheetah.xml",
                plane="xz", timestep=0.01, integrator="Euler", gravity=(0.0, 0.0, -9.81),  # xml:43
                density=0.0, viscosity=0.0, frame_skip=10, settotalmass=14.0,        # xml:36
                bodies=bodies, floor=floor, selfcollide=False,
                actuators=act, site=None, reward="half_cheetah")


ENV_ORDER = ("InvertedPendulumEnv", "InvertedDoublePendulumEnv", "SwimmerEnv",
             "HopperEnv", "Walker2dEnv", "HalfCheetahEnv")

MODELS = {m["env"]: m for m in (_inverted_pendulum(), _inverted_double_pendulum(), _swimmer(),
                                _hopper(), _walker2d(), _half_cheetah())}


def dims(env):
    m = MODELS[env]
    nq = sum(len(b["joints"]) for b in m["bodies"])
    return nq, nq, len(m["actuators"])
