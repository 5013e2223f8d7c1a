#!/usr/bin/env python3
"""Closed-form known-answer vectors for the CONSTRAINT physics (joint limits, floor contacts with pyramidal friction,
a frictionless capsule/capsule contact) and for Swimmer's fluid forces.  Run in the build container only; the fixture
(tests/golden/kat_constraints.npz) is data.

What is independent here, and of what:
  * constraint PARAMETERS (range, margin, solref/solimp of limits and geoms, friction, condim, timestep) are read
    straight from the reference's MJCF files (/root/reference/mujoco/assets/*.xml, global <default> block resolved
    here), not from this repo's transcription modelspec.py -- and asserted equal to it;
  * smooth dynamics (M, unconstrained acceleration) come from the Lagrangian/autodiff construction of
    gen_lagrange_kat.py; point Jacobians from autodiff of plain forward kinematics;
  * the constrained acceleration is NOT iterated: the convex problem
        min_a  1/2 (a - a0)^T M (a - a0) + sum_rows 1/2 D_r min(0, J_r a - aref_r)^2
    is solved by ENUMERATING the active sets (<= 2^5) and solving each linear system -- no Newton, no line search,
    none of the oracle's or the kernel's code.
Each vector encodes these MuJoCo 2.0 rules [from knowledge; MuJoCo itself is not available offline]:
  R1 limit row: dist = q - lo | hi - q, active iff dist < margin, J = +-e_j                     (mj_instantiateLimit)
  R2 reference acceleration: aref = -B (J v) - K imp (dist - margin), K = 1/(dmax^2 tc^2 dr^2), B = 2/(dmax tc),
     tc = max(solref[0], 2 timestep) (refsafe)                                                  (mj_makeImpedance)
  R3 impedance: 5-parameter solimp sigmoid of |dist - margin| / width, 3 given => midpoint .5, power 2
  R4 regulariser R = max(1e-15, (1 - imp)/imp diagApprox), D = 1/R; diagApprox = dof_invweight0[j] (limit),
     tran = invweight0(body1) + invweight0(body2) (frictionless contact), tran (1 + mu^2) (pyramid rows)
  R5 invweight0 at qpos0: dof: (M^-1)_jj; body: trace of (J_com M^-1 J_com^T) over the 3 translations / 3
  R6 pyramidal cone, condim 3: rows n +- mu t1, n +- mu t2, every row R = 2 mu^2 R_first (impratio 1); planar model:
     t2 is the plane normal, its two rows are both `n`
  R7 geom pair mixing: friction max, condim max, margin max, solref/solimp plain average (solmix 1:1)
  R8 capsule end / plane: sphere of the capsule's radius at the segment end, contact point half-way into the overlap
  R9 Euler with implicit joint damping: v' = v + h (M + h diag(damping))^-1 (M a);  RK4: classic, stages re-solve
  R10 inertia-box fluid model (density, viscosity): see fluid_forces() below
"""
import itertools
import math
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_lagrange_kat as LK  # noqa: E402
import mbrl_amd  # noqa: F401,E402
from mbrl_amd import modelspec  # noqa: E402

ASSETS = "/root/reference/mujoco/assets"
XML = {"InvertedPendulumEnv": "inverted_pendulum.xml", "InvertedDoublePendulumEnv": "inverted_double_pendulum.xml",
       "SwimmerEnv": "swimmer.xml", "HopperEnv": "hopper.xml", "Walker2dEnv": "walker2d.xml",
       "HalfCheetahEnv": "half_cheetah.xml"}
MINVAL, MINIMP, MAXIMP = 1e-15, 1e-4, 0.9999
SOLREF0, SOLIMP0 = (0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0)


# ------------------------------------------------------------------------------------------ MJCF, read directly
def _nums(s):
    return [float(eval(t, {"__builtins__": {}})) for t in s.split()]        # hopper.xml writes "0.13/2"


def _solimp(s):
    v = _nums(s)
    return tuple(v) + SOLIMP0[len(v):]


def xml_params(env):
    root = ET.parse(os.path.join(ASSETS, XML[env])).getroot()
    comp = root.find("compiler")
    deg = comp is None or comp.get("angle", "degree") == "degree"
    dflt = root.find("default")
    dj = dict(dflt.find("joint").attrib) if dflt is not None and dflt.find("joint") is not None else {}
    dg = dict(dflt.find("geom").attrib) if dflt is not None and dflt.find("geom") is not None else {}
    opt = root.find("option")
    out = {"timestep": float(opt.get("timestep", 0.002)), "density": float(opt.get("density", 0)),
           "viscosity": float(opt.get("viscosity", 0)), "joints": {}, "geoms": [], "floor": None}

    def geom_params(g):
        a = dict(dg); a.update(g.attrib)
        return dict(friction=_nums(a.get("friction", "1"))[0], condim=int(a.get("condim", 3)),
                    margin=float(a.get("margin", 0)), solref=tuple(_nums(a.get("solref", "0.02 1"))),
                    solimp=_solimp(a.get("solimp", "0.9 0.95 0.001")), contype=int(a.get("contype", 1)),
                    conaffinity=int(a.get("conaffinity", 1)), radius=_nums(a.get("size", "0"))[0])

    wb = root.find("worldbody")
    for g in wb.findall("geom"):
        if g.get("type") == "plane":
            out["floor"] = geom_params(g)
            out["floor"]["z"] = _nums(g.get("pos", "0 0 0"))[2]
    for body in wb.iter("body"):
        for j in body.findall("joint"):
            a = dict(dj); a.update(j.attrib)
            limited = a.get("limited", "false") == "true"
            rng = _nums(a.get("range", "0 0"))
            if deg and a.get("type", "hinge") == "hinge":
                rng = [math.radians(x) for x in rng]
            out["joints"][a["name"]] = dict(limited=limited, range=tuple(rng), margin=float(a.get("margin", 0)),
                                            solref=tuple(_nums(a.get("solreflimit", "0.02 1"))),
                                            solimp=_solimp(a.get("solimplimit", "0.9 0.95 0.001")))
        for g in body.findall("geom"):
            out["geoms"].append((body.get("name"), geom_params(g)))
    return out


def check_against_transcription(env, xp, m):
    """The transcription the oracle and the kernel are compiled from must say what the XML says."""
    assert abs(xp["timestep"] - m["timestep"]) < 1e-15, env
    for b in m["bodies"]:
        for j in b["joints"]:
            x = xp["joints"][j["name"]]
            assert x["limited"] == j["limited"], (env, j["name"])
            if j["limited"]:
                assert np.allclose(x["range"], j["range"], rtol=0, atol=1e-12), (env, j["name"], x["range"], j["range"])
                assert x["margin"] == j["margin"] and np.allclose(x["solref"], j["solref"]) and \
                    np.allclose(x["solimp"], j["solimp"]), (env, j["name"], x, j)
    gx = [g for _, g in xp["geoms"]]
    gm = [g for b in m["bodies"] for g in b["geoms"]]
    assert len(gx) == len(gm), env
    for x, g in zip(gx, gm):
        for k in ("friction", "condim", "margin", "contype", "conaffinity", "radius"):
            assert x[k] == g[k], (env, g["name"], k, x[k], g[k])
        assert np.allclose(x["solref"], g["solref"]) and np.allclose(x["solimp"], g["solimp"]), (env, g["name"])
    if m["floor"]:
        f = m["floor"]
        for k in ("friction", "condim", "margin", "contype", "conaffinity"):
            assert xp["floor"][k] == f[k], (env, "floor", k, xp["floor"][k], f[k])


# ------------------------------------------------------------------------------------------ rules R2-R4
def impedance(solimp, x_signed):
    dmin, dmax = (min(MAXIMP, max(MINIMP, s)) for s in solimp[:2])
    width, mid, power = max(0.0, solimp[2]), min(MAXIMP, max(MINIMP, solimp[3])), max(1.0, solimp[4])
    if dmin == dmax or width <= MINVAL:
        return 0.5 * (dmin + dmax)
    x = abs(x_signed) / width
    if x >= 1:
        return dmax
    if x <= 0:
        return dmin
    y = x if power == 1 else (x ** power / mid ** (power - 1) if x <= mid
                              else 1 - (1 - x) ** power / (1 - mid) ** (power - 1))
    return dmin + y * (dmax - dmin)


def kb(solref, solimp, h):
    dmax = min(MAXIMP, max(MINIMP, solimp[1]))
    tc, dr = max(solref[0], 2 * h), solref[1]
    return 1.0 / max(MINVAL, dmax * dmax * tc * tc * dr * dr), 2.0 / max(MINVAL, dmax * tc)


# ------------------------------------------------------------------------------------------ kinematics (autodiff)
class Sys:
    def __init__(self, env):
        self.env, self.m, self.xp = env, modelspec.MODELS[env], xml_params(env)
        check_against_transcription(env, self.xp, self.m)
        LK.set_plane(self.m["plane"])
        self.consts = LK.body_constants(self.m)
        self.joints = [j for b in self.m["bodies"] for j in b["joints"]]
        self.nv, self.nu = len(self.joints), len(self.m["actuators"])
        self.h = self.xp["timestep"]
        self.damp = np.array([j["damping"] for j in self.joints])
        q0 = np.array([j["ref"] for j in self.joints])
        _, M0 = LK.qacc_of(self.m, q0, np.zeros(self.nv), np.zeros(self.nu))
        Mi = np.linalg.inv(M0)
        self.dof_iw = np.diag(Mi).copy()                                            # R5
        J0 = torch.autograd.functional.jacobian(lambda x: LK.poses(self.m, self.consts, x), torch.tensor(q0)).numpy()
        self.body_tran = [(J0[3 * b] @ Mi @ J0[3 * b] + J0[3 * b + 1] @ Mi @ J0[3 * b + 1]) / 3.0
                          for b in range(len(self.m["bodies"]))]                  # R5: out-of-plane row is zero
        self.gparams = [g for _, g in self.xp["geoms"]]

    def point(self, q, b, local):
        """World (in-plane) position of the body-frame point `local` of body b, and its Jacobian [2, nv]."""
        cx, cz = self.consts[b][2], self.consts[b][3]

        def f(x):
            P = LK.poses(self.m, self.consts, x)
            dx, dz = LK.rot(P[3 * b + 2], torch.tensor(local[0] - cx), torch.tensor(local[1] - cz))
            return torch.stack([P[3 * b] + dx, P[3 * b + 1] + dz])
        qt = torch.tensor(q)
        return f(qt).numpy(), torch.autograd.functional.jacobian(f, qt).numpy()

    def geom_list(self):
        out, k = [], 0
        for b, body in enumerate(self.m["bodies"]):
            for g in body["geoms"]:
                out.append((b, g, self.gparams[k])); k += 1
        return out

    # ---- constraint rows at (q, v): list of (J [nv], aref, D) ------------------------------------------------
    def rows(self, q, v):
        R, info = [], []
        for j, jt in enumerate(self.joints):
            x = self.xp["joints"][jt["name"]]
            if not x["limited"]:
                continue
            for side, dist in ((+1.0, q[j] - x["range"][0]), (-1.0, x["range"][1] - q[j])):     # R1
                if dist < x["margin"]:
                    J = np.zeros(self.nv); J[j] = side
                    imp = impedance(x["solimp"], dist - x["margin"])
                    K, B = kb(x["solref"], x["solimp"], self.h)
                    Rg = max(MINVAL, (1 - imp) / imp * self.dof_iw[j])
                    R.append((J, -B * (J @ v) - K * imp * (dist - x["margin"]), 1.0 / Rg))
                    info.append(("limit", j))
        fl = self.xp["floor"]
        U, W = LK.U, LK.W_
        if fl is not None and self.m["floor"]:
            for b, g, gp in self.geom_list():
                if not ((gp["contype"] & fl["conaffinity"]) or (fl["contype"] & gp["conaffinity"])):
                    continue
                mu, dim = max(fl["friction"], gp["friction"]), max(fl["condim"], gp["condim"])   # R7
                margin = max(fl["margin"], gp["margin"])
                solref = tuple(0.5 * (a + c) for a, c in zip(fl["solref"], gp["solref"]))
                solimp = tuple(0.5 * (a + c) for a, c in zip(fl["solimp"], gp["solimp"]))
                for end in (g["a"], g["b"]):                                                      # R8
                    e, _ = self.point(q, b, (end[U], end[W]))
                    dist = e[1] - fl["z"] - gp["radius"]
                    if dist >= margin:
                        continue
                    assert dim == 3
                    # the contact point is fixed on the body at the world position (e_x, e_z - (radius + dist/2))
                    ang = LK.poses(self.m, self.consts, torch.tensor(q))[3 * b + 2].item()
                    off = np.array([0.0, -(gp["radius"] + 0.5 * dist)])
                    c, s = math.cos(ang), LK.SENSE * math.sin(ang)
                    loc = (c * off[0] - s * off[1], s * off[0] + c * off[1])      # inverse of LK.rot
                    _, Jp = self.point(q, b, (end[U] + loc[0], end[W] + loc[1]))
                    Jn, Jt = Jp[1], Jp[0]
                    imp = impedance(solimp, dist - margin)
                    K, B = kb(solref, solimp, self.h)
                    R0 = max(MINVAL, (1 - imp) / imp * self.body_tran[b] * (1 + mu * mu))        # R4
                    Rp = 2 * mu * mu * R0                                                          # R6
                    for Jr in (Jn + mu * Jt, Jn - mu * Jt, Jn, Jn):
                        R.append((Jr, -B * (Jr @ v) - K * imp * (dist - margin), 1.0 / Rp))
                    info.append(("floor", b))
        if self.m.get("selfcollide"):
            gl = self.geom_list()
            for (b1, g1, p1), (b2, g2, p2) in itertools.combinations(gl, 2):
                par = [bd["parent"] for bd in self.m["bodies"]]
                if b1 == b2 or par[b2] == b1 or par[b1] == b2:            # parent-child pairs are filtered
                    continue
                margin = max(p1["margin"], p2["margin"])
                seg = lambda b, g: (self.point(q, b, (g["a"][U], g["a"][W]))[0], self.point(q, b, (g["b"][U], g["b"][W]))[0])
                (a1, e1), (a2, e2) = seg(b1, g1), seg(b2, g2)
                s, t, d = closest(a1, e1, a2, e2)
                dist = d - p1["radius"] - p2["radius"]
                if dist >= margin:
                    continue
                if d < 1e-9 or crosses(a1, e1, a2, e2):   # crossing axes: out-of-plane normal, the row acts on no dof
                    info.append(("crossing", b1, b2))
                    continue
                assert max(p1["condim"], p2["condim"]) == 1 and d > 1e-6
                c1, c2 = a1 + s * (e1 - a1), a2 + t * (e2 - a2)
                n = (c2 - c1) / d
                pos = c1 + n * (p1["radius"] + 0.5 * dist)

                def jac_at(b):
                    o, _ = self.point(q, b, (0.0, 0.0))
                    ang = LK.poses(self.m, self.consts, torch.tensor(q))[3 * b + 2].item()
                    w = pos - o
                    c_, s_ = math.cos(ang), LK.SENSE * math.sin(ang)
                    return self.point(q, b, (c_ * w[0] - s_ * w[1], s_ * w[0] + c_ * w[1]))
                (pw1, J1), (pw2, J2) = jac_at(b1), jac_at(b2)
                assert np.allclose(pw1, pos, atol=1e-12) and np.allclose(pw2, pos, atol=1e-12)
                Jr = n @ (J2 - J1)
                solref = tuple(0.5 * (a + c) for a, c in zip(p1["solref"], p2["solref"]))
                solimp = tuple(0.5 * (a + c) for a, c in zip(p1["solimp"], p2["solimp"]))
                imp = impedance(solimp, dist - margin)
                K, B = kb(solref, solimp, self.h)
                Rg = max(MINVAL, (1 - imp) / imp * (self.body_tran[b1] + self.body_tran[b2]))
                R.append((Jr, -B * (Jr @ v) - K * imp * (dist - margin), 1.0 / Rg))
                info.append(("pair", b1, b2))
        return R, info

    # ---- constrained acceleration by active-set enumeration ------------------------------------------------
    def qacc(self, q, v, c):
        a0, M = LK.qacc_of(self.m, q, v, c)
        if self.xp["density"] > 0 or self.xp["viscosity"] > 0:          # Swimmer: the fluid is part of qfrc_passive
            a0 = a0 + np.linalg.solve(M, fluid_forces(self, q, v))
        R, info = self.rows(q, v)
        if not R:
            return a0, M, info, 0
        merged = []                                  # identical rows (the pyramid's two `n` rows): one row, D summed
        for Jr, ar, Dr in R:
            for mrow in merged:
                if np.array_equal(mrow[0], Jr) and mrow[1] == ar:
                    mrow[2] += Dr
                    break
            else:
                merged.append([Jr, ar, Dr])
        R = merged
        J = np.array([r[0] for r in R]); aref = np.array([r[1] for r in R]); D = np.array([r[2] for r in R])
        found = []
        for mask in itertools.product((False, True), repeat=len(R)):
            on = np.array(mask)
            H = M + (J[on].T * D[on]) @ J[on]
            a = np.linalg.solve(H, M @ a0 + J[on].T @ (D[on] * aref[on]))
            res = J @ a - aref
            if all((res[on] < 0)) and all(res[~on] >= 0):
                found.append((a, int(on.sum())))
        assert len(found) >= 1, "no consistent active set"
        for a, _ in found[1:]:                   # duplicated rows (n, n) admit equivalent sets: the point is unique
            assert np.allclose(a, found[0][0], rtol=1e-10, atol=1e-10)
        return found[0][0], M, info, found[0][1]

    def step(self, q, v, c):
        """One mj_step (R9)."""
        h = self.h
        if self.m["integrator"] == "RK4":
            F = lambda qq, vv: (vv, self.qacc(qq, vv, c)[0])
            k1q, k1v = F(q, v)
            k2q, k2v = F(q + 0.5 * h * k1q, v + 0.5 * h * k1v)
            k3q, k3v = F(q + 0.5 * h * k2q, v + 0.5 * h * k2v)
            k4q, k4v = F(q + h * k3q, v + h * k3v)
            return q + h / 6 * (k1q + 2 * k2q + 2 * k3q + k4q), v + h / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
        a, M, _, _ = self.qacc(q, v, c)
        v1 = v + h * np.linalg.solve(M + h * np.diag(self.damp), M @ a)
        return q + h * v1, v1

    def free_state(self, rng, inside=0.15):
        q = np.zeros(self.nv)
        for i, j in enumerate(self.joints):
            if j["limited"]:
                lo, hi = j["range"]
                q[i] = rng.uniform(lo + inside * (hi - lo), hi - inside * (hi - lo))
            else:
                q[i] = j["ref"] + rng.uniform(-0.4, 0.4)
        if self.m["floor"]:
            q[1] += 5.0
        return q, rng.standard_normal(self.nv), rng.uniform(-1, 1, self.nu)


def crosses(a1, e1, a2, e2):
    """Do two coplanar segments intersect?  (orientation test)"""
    o = lambda p, q, r: (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
    return o(a1, e1, a2) * o(a1, e1, e2) < 0 and o(a2, e2, a1) * o(a2, e2, e1) < 0


def closest(a1, e1, a2, e2):
    """Closest points of two segments by brute force + golden refinement (deliberately not the clamped closed form the
    oracle and the kernel use)."""
    S = np.linspace(0, 1, 401)
    P1 = a1[None] + S[:, None] * (e1 - a1)[None]
    P2 = a2[None] + S[:, None] * (e2 - a2)[None]
    d = np.linalg.norm(P1[:, None, :] - P2[None, :, :], axis=2)
    i, j = np.unravel_index(d.argmin(), d.shape)
    s, t, w = S[i], S[j], 1.0 / 400
    for _ in range(60):                                   # alternate exact 1-D projections inside shrinking boxes
        p = a1 + s * (e1 - a1)
        t = float(np.clip(((p - a2) @ (e2 - a2)) / ((e2 - a2) @ (e2 - a2)), 0, 1))
        p = a2 + t * (e2 - a2)
        s = float(np.clip(((p - a1) @ (e1 - a1)) / ((e1 - a1) @ (e1 - a1)), 0, 1))
    return s, t, float(np.linalg.norm(a1 + s * (e1 - a1) - a2 - t * (e2 - a2)))


# ------------------------------------------------------------------------------------------ R10: inertia-box fluid
def fluid_forces(S, q, v):
    """MuJoCo 2.0's only fluid model [from knowledge, mj_passive]: per body an equivalent box with the body's mass and
    principal inertias (a solid box has I_x = m (b_y^2 + b_z^2) / 12, so b_x^2 = 6 (I_y + I_z - I_x) / m); with the
    6-D velocity (w, u) at the body's centre of mass in body axes, d = mean box side:
        torque_i = -pi d^3 visc w_i - dens b_i (b_j^4 + b_k^4) / 64 |w_i| w_i
        force_i  = -3 pi d visc u_i  - dens b_j b_k / 2 |u_i| u_i
    Returns the generalised force J^T (force, torque)."""
    m, xp = S.m, S.xp
    U, W = LK.U, LK.W_
    qt = torch.tensor(q)
    P = LK.poses(m, S.consts, qt).detach().numpy()
    Jp = torch.autograd.functional.jacobian(lambda x: LK.poses(m, S.consts, x), qt).numpy()
    Q = np.zeros(S.nv)
    for b, body in enumerate(m["bodies"]):
        # principal inertias of the body (capsules along the body x axis in swimmer.xml): axial and transverse
        mass, I_ax, I_tr = 0.0, 0.0, S.consts[b][1]
        for g in body["geoms"]:
            a, e, r, rho = np.array(g["a"]), np.array(g["b"]), g["radius"], g["density"]
            hh = np.linalg.norm(e - a)
            assert abs((e - a)[1]) < 1e-12 and abs((e - a)[2]) < 1e-12, "capsule along body x"
            mc, ms = rho * math.pi * r * r * hh, rho * math.pi * r ** 3
            mass += mc + ms
            I_ax += 0.5 * mc * r * r + 0.4 * ms * r * r
        assert len(body["geoms"]) == 1 and not m["settotalmass"]
        I = np.array([I_ax, I_tr, I_tr])                      # body axes x, y, z
        box = np.sqrt(np.maximum(MINVAL, [I[1] + I[2] - I[0], I[0] + I[2] - I[1], I[0] + I[1] - I[2]]) / mass * 6.0)
        d = box.mean()
        th = P[3 * b + 2]
        vel = Jp[3 * b:3 * b + 3] @ v                         # (xdot, ydot, thetadot) of the com, world axes
        c, s = math.cos(th), math.sin(th)                     # x-y plane, rotation about +z
        u = np.array([c * vel[0] + s * vel[1], -s * vel[0] + c * vel[1], 0.0])
        w = np.array([0.0, 0.0, vel[2]])
        dens, visc = xp["density"], xp["viscosity"]
        tq = -math.pi * d ** 3 * visc * w
        fr = -3 * math.pi * d * visc * u
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            tq[i] -= dens * box[i] * (box[j] ** 4 + box[k] ** 4) / 64.0 * abs(w[i]) * w[i]
            fr[i] -= 0.5 * dens * box[j] * box[k] * abs(u[i]) * u[i]
        fw = np.array([c * fr[0] - s * fr[1], s * fr[0] + c * fr[1]])
        Q += Jp[3 * b] * fw[0] + Jp[3 * b + 1] * fw[1] + Jp[3 * b + 2] * tq[2]
    return Q


# ------------------------------------------------------------------------------------------ cases
def multi_contact():
    """(ii'') several capsule ends on the floor at once (both ends of Hopper's foot, the two feet of Walker2d and
    HalfCheetah): 2-3 contacts = 6-9 distinct pyramid rows coupled through J M^-1 J^T, enumerated like the rest.
    Written to kat_constraints_multi.npz."""
    rng = np.random.default_rng(23)
    out = {}
    U, W = LK.U, LK.W_
    for env in ("HopperEnv", "Walker2dEnv", "HalfCheetahEnv"):
        S = Sys(env)
        recs = []

        def heights(q):
            hs = []
            for b, g, gp in S.geom_list():
                for end in (g["a"], g["b"]):
                    e, _ = S.point(q, b, (end[U], end[W]))
                    hs.append(e[1] - S.xp["floor"]["z"] - gp["radius"])
            return np.sort(hs)

        tries = 0
        while len(recs) < 8 and tries < 200:
            tries += 1
            q = np.array([j["ref"] for j in S.joints]) + rng.uniform(-0.06, 0.06, S.nv)
            for i, j in enumerate(S.joints):
                if j["limited"]:
                    lo, hi = j["range"]
                    q[i] = min(max(q[i], lo + 0.02), hi - 0.02)
            lo_p, hi_p = q[2] - 0.6, q[2] + 0.6           # pitch that puts the two lowest ends at the same height
            gap = lambda p: (lambda h: h[1] - h[0])(heights(np.concatenate([q[:2], [p], q[3:]])))
            ps = np.linspace(lo_p, hi_p, 61)
            gs = [gap(p) for p in ps]
            k = int(np.argmin(gs))
            a_, b_ = ps[max(k - 1, 0)], ps[min(k + 1, 60)]
            for _ in range(40):                           # golden-section style shrink on the (V-shaped) gap
                m1, m2 = a_ + 0.382 * (b_ - a_), a_ + 0.618 * (b_ - a_)
                if gap(m1) < gap(m2):
                    b_ = m2
                else:
                    a_ = m1
            q[2] = 0.5 * (a_ + b_)
            h = heights(q)
            margin = max(S.xp["floor"]["margin"], S.gparams[0]["margin"])
            width = 0.5 * (S.xp["floor"]["solimp"][2] + S.gparams[0]["solimp"][2])
            q[1] += -h[0] + margin - rng.uniform(0.2, 1.2) * width
            v = 0.5 * rng.standard_normal(S.nv)
            v[0] += rng.uniform(-1.5, 1.5)
            c = rng.uniform(-1, 1, S.nu)
            R, info = S.rows(q, v)
            ncon = sum(1 for i in info if i[0] == "floor")
            if ncon < 2 or ncon > 3 or any(i[0] not in ("floor",) for i in info):
                continue
            a, M, info, nact = S.qacc(q, v, c)
            q1, v1 = S.step(q, v, c)
            recs.append(dict(q=q, v=v, c=c, qacc=a, q1=q1, v1=v1, nactive=nact, dof=ncon))
        tag = "multi_" + env[:-3].lower()
        for k in recs[0]:
            out[tag + "_" + k] = np.array([r[k] for r in recs])
        print("%-34s %3d vectors  contacts %s  active merged rows %s" % (
            tag, len(recs), sorted({int(r["dof"]) for r in recs}), sorted({int(r["nactive"]) for r in recs})), flush=True)
    np.savez(os.path.join(HERE, "kat_constraints_multi.npz"), **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "multi":
        return multi_contact()
    rng = np.random.default_rng(11)
    out = {}

    def put(tag, recs):
        for k in recs[0]:
            out[tag + "_" + k] = np.array([r[k] for r in recs])
        print("%-34s %3d vectors  |qacc|max %.3g  active rows %s" % (
            tag, len(recs), max(np.abs(r["qacc"]).max() for r in recs), sorted({int(r["nactive"]) for r in recs})))

    # (i) every limited joint of every model, both sides, pushed past its range (IDP slider: inside its margin too)
    for env in XML:
        S = Sys(env)
        recs = []
        for j, jt in enumerate(S.joints):
            x = S.xp["joints"][jt["name"]]
            if not x["limited"]:
                continue
            width = x["solimp"][2]
            for side in (0, 1):
                for depth in (0.3, 1.7):                       # inside the impedance ramp, and beyond it
                    for _ in range(200):                       # (Hopper: re-draw poses whose capsules touch or cross)
                        q, v, c = S.free_state(rng)
                        pen = depth * width - x["margin"]
                        q[j] = x["range"][0] - pen if side == 0 else x["range"][1] + pen
                        if depth > 1:                          # driving into the limit (HalfCheetah's joint springs
                            v[j] = (-1.0 if side == 0 else 1.0) * rng.uniform(5, 25)   # alone push back out of it)
                        a, M, info, nact = S.qacc(q, v, c)
                        if info == [("limit", j)]:
                            break
                    assert info == [("limit", j)], (env, info)
                    q1, v1 = S.step(q, v, c)
                    recs.append(dict(q=q, v=v, c=c, qacc=a, q1=q1, v1=v1, nactive=nact, dof=j))
        if recs:
            put("limit_" + env[:-3].lower(), recs)

    # (ii) one capsule end on the floor, pyramidal friction, random sliding velocity
    for env in ("HopperEnv", "Walker2dEnv", "HalfCheetahEnv"):
        S = Sys(env)
        recs = []
        U, W = LK.U, LK.W_
        while len(recs) < 10:
            q, v, c = S.free_state(rng, inside=0.2)
            q[1] -= 5.0
            ends = []
            for b, g, gp in S.geom_list():
                for end in (g["a"], g["b"]):
                    e, _ = S.point(q, b, (end[U], end[W]))
                    ends.append(e[1] - S.xp["floor"]["z"] - gp["radius"])
            ends = np.sort(ends)
            if ends[1] - ends[0] < 0.03:
                continue
            margin = max(S.xp["floor"]["margin"], S.gparams[0]["margin"])
            width = 0.5 * (S.xp["floor"]["solimp"][2] + S.gparams[0]["solimp"][2])
            q[1] += -ends[0] + margin - rng.uniform(0.1, 1.6) * width
            v[0] += rng.uniform(-2, 2)                         # known tangential velocity
            v[1] = rng.uniform(-1.0, 0.3)
            a, M, info, nact = S.qacc(q, v, c)
            if [i[0] for i in info] != ["floor"]:
                continue
            q1, v1 = S.step(q, v, c)
            recs.append(dict(q=q, v=v, c=c, qacc=a, q1=q1, v1=v1, nactive=nact, dof=-1))
        put("floor_" + env[:-3].lower(), recs)

    # (ii') frictionless 1-row contact: Hopper's foot capsule against its thigh / torso capsule (condim 1)
    S = Sys("HopperEnv")
    recs, tries = [], 0
    U, W = LK.U, LK.W_

    def pair_dist(q):
        best = None
        gl = S.geom_list()
        par = [bd["parent"] for bd in S.m["bodies"]]
        for (b1, g1, p1), (b2, g2, p2) in itertools.combinations(gl, 2):
            if par[b2] == b1 or par[b1] == b2:
                continue
            seg = lambda b, g: (S.point(q, b, (g["a"][U], g["a"][W]))[0], S.point(q, b, (g["b"][U], g["b"][W]))[0])
            (a1, e1), (a2, e2) = seg(b1, g1), seg(b2, g2)
            d = closest(a1, e1, a2, e2)[2] - p1["radius"] - p2["radius"]
            best = d if best is None else min(best, d)
        return best

    while len(recs) < 8 and tries < 400:
        tries += 1
        q, v, c = S.free_state(rng, inside=0.02)
        q[4] = rng.uniform(math.radians(-149), math.radians(-120))       # knee folded
        lo, hi = S.joints[3]["range"]
        grid = np.linspace(lo + 0.02, hi - 0.02, 40)
        d = []
        for x in grid:
            q[3] = x
            d.append(pair_dist(q))
        d = np.array(d)
        k = np.where((d[:-1] > 0.002) & (d[1:] < -0.002))[0]
        if len(k) == 0:
            k = np.where((d[1:] > 0.002) & (d[:-1] < -0.002))[0]
            if len(k) == 0:
                continue
            xa, xb = grid[k[0] + 1], grid[k[0]]
        else:
            xa, xb = grid[k[0]], grid[k[0] + 1]
        target = 0.001 - rng.uniform(0.1, 1.5) * 0.01                    # margin - depth * width
        for _ in range(50):                                              # bisection: xa outside, xb inside
            xm = 0.5 * (xa + xb)
            q[3] = xm
            if pair_dist(q) > target:
                xa = xm
            else:
                xb = xm
        a, M, info, nact = S.qacc(q, v, c)
        if [i[0] for i in info] != ["pair"]:
            continue
        q1, v1 = S.step(q, v, c)
        recs.append(dict(q=q.copy(), v=v, c=c, qacc=a, q1=q1, v1=v1, nactive=nact, dof=-1))
    put("pair_hopper", recs)

    # (iii) Swimmer: inertia-box fluid forces on top of the Lagrangian's smooth dynamics, limits inactive
    S = Sys("SwimmerEnv")
    recs = []
    for _ in range(12):
        q, v, c = S.free_state(rng)
        v = v * rng.uniform(0.2, 3.0)
        Qf = fluid_forces(S, q, v)
        a, M, info, nact = S.qacc(q, v, c)
        assert not info
        q1, v1 = S.step(q, v, c)
        recs.append(dict(q=q, v=v, c=c, qacc=a, q1=q1, v1=v1, nactive=nact, dof=-1, qfrc_fluid=Qf))
    put("fluid_swimmer", recs)

    np.savez(os.path.join(HERE, "kat_constraints.npz"), **out)


if __name__ == "__main__":
    main()
