"""Planar model compiler for the HIP path.

Turns the raw transcription in ``modelspec.py`` into compile-time (``constexpr``) tables
for ``csrc/planar_sim.hpp``.  All six reference models move in one plane (x-z with
y-hinges; Swimmer: x-y with z-hinges), so the kernel works with planar spatial algebra:
motion vectors [omega, v_a, v_b], forces [tau, f_a, f_b], (a, b) = (x, z) or (x, y).

Sign convention: the planar angle phi is counter-clockwise in the (a, b) plane.  A hinge
about +z in the x-y plane has d(phi)/dq = +1; a hinge about +y in the x-z plane has
d(phi)/dq = -1 (rotating x towards -z).  ``dof_sgn`` carries that factor so that every
generalized quantity stays in the reference's own q coordinates.

Derived constants (masses, inertias, invweight0, solver stiffness/damping) are computed
here with numpy, independently of the oracle's C derivation (cross-checked in tests).
MuJoCo semantics restated here are [from knowledge] -- see DESIGN.md.
"""
import math
import os

import numpy as np

from . import modelspec

MINVAL = 1e-15
MINIMP, MAXIMP = 1e-4, 0.9999
REWARD_ID = {"inverted_pendulum": 0, "inverted_double_pendulum": 1, "swimmer": 2,
             "hopper": 3, "walker2d": 4, "half_cheetah": 5}


def _plane_axes(plane):
    # indices of (a, b, normal) in xyz, and d(phi)/d(angle about +normal)
    return ((0, 2, 1), -1.0) if plane == "xz" else ((0, 1, 2), +1.0)


def capsule_mass_props(a, b, radius, density):
    """mass, com, (I_axial, I_perp) of a capsule; MuJoCo 2.0 volume = pi r^2 (h + r)
    (reproduces the known mujoco200 gym masses; see oracle/mjc_oracle.c capsule_inertia)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    h = float(np.linalg.norm(b - a))
    r = radius
    mc = density * math.pi * r * r * h
    ms = density * math.pi * r ** 3
    i_ax = mc * r * r / 2 + ms * 2 * r * r / 5
    i_pe = mc * (3 * r * r + h * h) / 12 + ms * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8)
    return mc + ms, 0.5 * (a + b), (b - a) / h, i_ax, i_pe


def clamp_solimp(s):
    return (min(MAXIMP, max(MINIMP, s[0])), min(MAXIMP, max(MINIMP, s[1])), max(0.0, s[2]),
            min(MAXIMP, max(MINIMP, s[3])), max(1.0, s[4]))


def ref_kb(solref, solimp, timestep):
    """Stiffness K and damping B of the reference acceleration (refsafe applied)."""
    dmax = clamp_solimp(solimp)[1]
    t, d = solref
    if t > 0:
        t = max(t, 2 * timestep)
        return 1.0 / max(MINVAL, dmax * dmax * t * t * d * d), 2.0 / max(MINVAL, dmax * t)
    return -t / max(MINVAL, dmax * dmax), -d / max(MINVAL, dmax)


class PlanarModel:
    def __init__(self, env):
        m = modelspec.MODELS[env]
        self.spec = m
        self.env = env
        (ia, ib, inn), sig = _plane_axes(m["plane"])
        self.ia, self.ib, self.inn, self.sigma = ia, ib, inn, sig
        B = m["bodies"]
        self.NB = len(B)
        self.parent = [b["parent"] for b in B]
        self.bpos = [(b["pos"][ia], b["pos"][ib]) for b in B]
        assert all(abs(b["pos"][inn]) < 1e-12 for b in B)
        self.mass, self.com, self.izz, self.boxdims = [], [], [], []
        self.dof_type, self.dof_body, self.dof_sgn, self.jpos = [], [], [], []
        self.qpos0, self.stiff, self.damp, self.arma = [], [], [], []
        self.limited, self.rng, self.lim_margin, self.lim_K, self.lim_B, self.lim_solimp = [], [], [], [], [], []
        self.dofadr, self.dofnum, self.jnames = [], [], []
        self.g_body, self.g_ea, self.g_eb, self.g_rad, self.g_raw = [], [], [], [], []
        total = 0.0
        for bi, b in enumerate(B):
            self.dofadr.append(len(self.dof_type))
            self.dofnum.append(len(b["joints"]))
            seen_hinge = False
            for j in b["joints"]:
                ax = np.asarray(j["axis"], float)
                ax = ax / np.linalg.norm(ax)
                if j["type"] == "slide":
                    assert b["parent"] == -1 and not seen_hinge, "slides must precede hinges on a root body"
                    assert abs(ax[inn]) < 1e-12
                    if abs(ax[ia] - 1) < 1e-12:
                        self.dof_type.append(0)
                    elif abs(ax[ib] - 1) < 1e-12:
                        self.dof_type.append(1)
                    else:
                        raise ValueError("slide axis must be +a or +b")
                    self.dof_sgn.append(1.0)
                else:
                    seen_hinge = True
                    assert abs(abs(ax[inn]) - 1) < 1e-12, "hinge axis must be the plane normal"
                    self.dof_type.append(2)
                    self.dof_sgn.append(sig * ax[inn])
                assert abs(j["pos"][inn]) < 1e-12 or j["type"] == "slide"
                self.dof_body.append(bi)
                self.jpos.append((j["pos"][ia], j["pos"][ib]))
                self.qpos0.append(j["ref"])
                self.stiff.append(j["stiffness"])
                self.damp.append(j["damping"])
                self.arma.append(j["armature"])
                self.limited.append(bool(j["limited"]))
                self.rng.append(j["range"])
                self.lim_margin.append(j["margin"])
                K, Bd = ref_kb(j["solref"], j["solimp"], m["timestep"])
                self.lim_K.append(K)
                self.lim_B.append(Bd)
                self.lim_solimp.append(clamp_solimp(j["solimp"]))
                self.jnames.append(j["name"])
            # inertia from geoms (planar: mass, com, inertia about the plane normal)
            props = [capsule_mass_props(g["a"], g["b"], g["radius"], g["density"]) for g in b["geoms"]]
            mb = sum(p[0] for p in props)
            com = sum(p[0] * p[1] for p in props) / mb
            inn_I = 0.0
            I3 = np.zeros((3, 3))
            for (mg, cg, ug, i_ax, i_pe) in props:
                assert abs(ug[inn]) < 1e-12 and abs(cg[inn]) < 1e-12, "geoms must lie in the plane"
                d = cg - com
                Ig = i_pe * (np.eye(3) - np.outer(ug, ug)) + i_ax * np.outer(ug, ug)
                I3 += Ig + mg * (d @ d * np.eye(3) - np.outer(d, d))
            inn_I = I3[inn, inn]
            self.mass.append(mb)
            total += mb
            self.com.append((com[ia], com[ib]))
            self.izz.append(inn_I)
            self.boxdims.append(I3)
            for g in b["geoms"]:
                self.g_body.append(bi)
                self.g_ea.append((g["a"][ia], g["a"][ib]))
                self.g_eb.append((g["b"][ia], g["b"][ib]))
                self.g_rad.append(g["radius"])
                self.g_raw.append(g)
        if m["settotalmass"]:
            sc = m["settotalmass"] / total
            self.mass = [x * sc for x in self.mass]
            self.izz = [x * sc for x in self.izz]
            self.boxdims = [I * sc for I in self.boxdims]
        self.NV = len(self.dof_type)
        self.NU = len(m["actuators"])
        self.NG = len(self.g_body)
        self.act_dof = [self.jnames.index(a["joint"]) for a in m["actuators"]]
        self.gear = [a["gear"] for a in m["actuators"]]
        self.ctrllimited = [bool(a["ctrllimited"]) for a in m["actuators"]]
        self.ctrlrange = [a["ctrlrange"] for a in m["actuators"]]
        self.timestep = m["timestep"]
        self.rk4 = m["integrator"] == "RK4"
        self.frame_skip = m["frame_skip"]
        self.grav = (m["gravity"][ia], m["gravity"][ib])
        self._invweights()
        self._contacts()
        self._fluid()
        self.site = None
        if m["site"] is not None:
            self.site = (m["site"]["body"], m["site"]["pos"][ia], m["site"]["pos"][ib])

    # ---- planar kinematics / CRBA in numpy (compile-time only) ----
    def _kin(self, q):
        P, C, S = [None] * self.NB, [0.0] * self.NB, [0.0] * self.NB
        anchors = [None] * self.NV
        for b in range(self.NB):
            p = self.parent[b]
            if p < 0:
                pos, c, s = np.array(self.bpos[b], float), 1.0, 0.0
            else:
                R = np.array([[C[p], -S[p]], [S[p], C[p]]])
                pos, c, s = P[p] + R @ np.array(self.bpos[b]), C[p], S[p]
            for j in range(self.dofadr[b], self.dofadr[b] + self.dofnum[b]):
                dq = q[j] - self.qpos0[j]
                R = np.array([[c, -s], [s, c]])
                if self.dof_type[j] == 0:
                    pos = pos + np.array([dq, 0.0])
                    anchors[j] = pos.copy()
                elif self.dof_type[j] == 1:
                    pos = pos + np.array([0.0, dq])
                    anchors[j] = pos.copy()
                else:
                    A = pos + R @ np.array(self.jpos[j])
                    ang = math.atan2(s, c) + self.dof_sgn[j] * dq
                    c, s = math.cos(ang), math.sin(ang)
                    R = np.array([[c, -s], [s, c]])
                    pos = A - R @ np.array(self.jpos[j])
                    anchors[j] = A
            P[b], C[b], S[b] = pos, c, s
        return P, C, S, anchors

    def _is_anc_dof(self, j, b):
        """dof j moves body b?"""
        bb = b
        while bb >= 0:
            if bb == self.dof_body[j]:
                return True
            bb = self.parent[bb]
        return False

    def _jac_point(self, q, b, point, kin=None):
        """2 x NV translational Jacobian (a, b rows) and 1 x NV rotational (planar omega)."""
        P, C, S, anchors = kin or self._kin(q)
        Jp, Jr = np.zeros((2, self.NV)), np.zeros(self.NV)
        for j in range(self.NV):
            if not self._is_anc_dof(j, b):
                continue
            if self.dof_type[j] == 0:
                Jp[0, j] = 1
            elif self.dof_type[j] == 1:
                Jp[1, j] = 1
            else:
                r = point - anchors[j]
                Jp[:, j] = self.dof_sgn[j] * np.array([-r[1], r[0]])
                Jr[j] = self.dof_sgn[j]
        return Jp, Jr

    def mass_matrix(self, q):
        kin = self._kin(q)
        P, C, S, _ = kin
        M = np.diag(np.array(self.arma, float))
        for b in range(self.NB):
            R = np.array([[C[b], -S[b]], [S[b], C[b]]])
            cw = P[b] + R @ np.array(self.com[b])
            Jp, Jr = self._jac_point(q, b, cw, kin)
            M += self.mass[b] * Jp.T @ Jp + self.izz[b] * np.outer(Jr, Jr)
        return M

    def _invweights(self):
        q0 = np.array(self.qpos0, float)
        kin = self._kin(q0)
        P, C, S, _ = kin
        Minv = np.linalg.inv(self.mass_matrix(q0))
        self.dof_invweight0 = [float(Minv[j, j]) for j in range(self.NV)]
        self.body_invweight0 = []
        for b in range(self.NB):
            R = np.array([[C[b], -S[b]], [S[b], C[b]]])
            cw = P[b] + R @ np.array(self.com[b])
            Jp, Jr = self._jac_point(q0, b, cw, kin)
            # 3-D trace / 3: the out-of-plane translational row is identically zero;
            # rotation: only the plane-normal component is non-zero
            tran = float(np.trace(Jp @ Minv @ Jp.T)) / 3.0
            rot = float(Jr @ Minv @ Jr) / 3.0
            self.body_invweight0.append((max(MINVAL, tran), max(MINVAL, rot)))

    def _contacts(self):
        m = self.spec
        fl = m["floor"]
        self.has_floor = fl is not None
        self.floor_z = fl["z"] if fl else 0.0
        self.c_on, self.c_mu, self.c_margin, self.c_K, self.c_B, self.c_solimp, self.c_RK, self.c_dim = \
            [], [], [], [], [], [], [], []
        for gi, g in enumerate(self.g_raw):
            on = bool(fl) and bool((g["contype"] & fl["conaffinity"]) or (fl["contype"] & g["conaffinity"]))
            self.c_on.append(on)
            if not on:
                for lst, v in ((self.c_mu, 0.0), (self.c_margin, 0.0), (self.c_K, 0.0), (self.c_B, 0.0),
                               (self.c_solimp, modelspec.SOLIMP0), (self.c_RK, 0.0), (self.c_dim, 1)):
                    lst.append(v)
                continue
            mu = max(g["friction"], fl["friction"])
            dim = max(g["condim"], fl["condim"])
            margin = max(g["margin"], fl["margin"])
            solref = tuple(0.5 * (x + y) for x, y in zip(g["solref"], fl["solref"]))
            solimp = tuple(0.5 * (x + y) for x, y in zip(g["solimp"], fl["solimp"]))
            K, Bd = ref_kb(solref, solimp, m["timestep"])
            tran = self.body_invweight0[self.g_body[gi]][0]  # world body contributes 0
            if dim == 1:
                RK = tran
            else:
                assert dim == 3
                RK = 2.0 * mu * mu * (tran + mu * mu * tran)  # pyramidal: R = 2 mu^2 R0
            self.c_mu.append(mu)
            self.c_dim.append(dim)
            self.c_margin.append(margin)
            self.c_K.append(K)
            self.c_B.append(Bd)
            self.c_solimp.append(clamp_solimp(solimp))
            self.c_RK.append(RK)
        self.selfcollide = bool(m["selfcollide"])
        # capsule-capsule pairs (Hopper): same-body and parent-child (parent != world) pairs are filtered
        self.pairs = []
        if self.selfcollide:
            for g1 in range(self.NG):
                for g2 in range(g1 + 1, self.NG):
                    b1, b2 = self.g_body[g1], self.g_body[g2]
                    if b1 == b2 or self.parent[b2] == b1 or self.parent[b1] == b2:
                        continue
                    G1, G2 = self.g_raw[g1], self.g_raw[g2]
                    if not ((G1["contype"] & G2["conaffinity"]) or (G2["contype"] & G1["conaffinity"])):
                        continue
                    bb = b2
                    while bb >= 0 and bb != b1:
                        bb = self.parent[bb]
                    assert bb == b1, "pair rows assume geom1's body is an ancestor of geom2's body"
                    assert max(G1["condim"], G2["condim"]) == 1, "body-body contacts are frictionless here"
                    solref = tuple(0.5 * (x + y) for x, y in zip(G1["solref"], G2["solref"]))
                    solimp = tuple(0.5 * (x + y) for x, y in zip(G1["solimp"], G2["solimp"]))
                    K, Bd = ref_kb(solref, solimp, m["timestep"])
                    self.pairs.append(dict(g1=g1, g2=g2, margin=max(G1["margin"], G2["margin"]), K=K, B=Bd,
                                           solimp=clamp_solimp(solimp),
                                           RK=self.body_invweight0[b1][0] + self.body_invweight0[b2][0]))

    def _fluid(self):
        m = self.spec
        rho, mu = m["density"], m["viscosity"]
        self.fluid = rho > 0 or mu > 0
        self.fl = []
        ia, ib, inn = self.ia, self.ib, self.inn
        for b in range(self.NB):
            I = self.boxdims[b]
            if self.fluid:
                off = abs(I[0, 1]) + abs(I[0, 2]) + abs(I[1, 2])
                assert off < 1e-12 * np.trace(I), "fluid model needs a body-axis-aligned inertial frame"
            sc = 6.0 / self.mass[b]
            bx = [math.sqrt(max(MINVAL, I[(k + 1) % 3, (k + 1) % 3] + I[(k + 2) % 3, (k + 2) % 3] - I[k, k]) * sc)
                  for k in range(3)]
            ba, bb, bn = bx[ia], bx[ib], bx[inn]
            d = (ba + bb + bn) / 3.0
            self.fl.append(dict(kv_lin=3 * math.pi * d * mu, kv_ang=math.pi * d ** 3 * mu,
                                kq_a=0.5 * rho * bb * bn, kq_b=0.5 * rho * ba * bn,
                                kq_ang=rho * bn * (ba ** 4 + bb ** 4) / 64.0))


# --------------------------------------------------------------------------- #
# C++ emission
# --------------------------------------------------------------------------- #
def _d(x):
    return repr(float(x))


def _arr(vals, fmt=_d):
    return "{" + ", ".join(fmt(v) for v in vals) + "}"


def _arr2(rows, fmt=_d):
    return "{" + ", ".join(_arr(r, fmt) for r in rows) + "}"


def _b(x):
    return "true" if x else "false"


def emit_model(pm, cname):
    NB, NV, NU, NG = pm.NB, pm.NV, pm.NU, pm.NG
    L = []
    A = L.append
    A("struct %s {  // %s  (%s)" % (cname, pm.env, pm.spec["src"]))
    A("  static constexpr int ENV_ID = %d, NB = %d, NV = %d, NU = %d, NG = %d;" % (
        modelspec.ENV_ORDER.index(pm.env), NB, NV, NU, max(NG, 1)))
    A("  static constexpr int NGEOM = %d, REWARD = %d, FRAME_SKIP = %d;" % (
        NG, REWARD_ID[pm.spec["reward"]], pm.frame_skip))
    A("  static constexpr bool RK4 = %s, HAS_FLOOR = %s, FLUID = %s, SELFCOLLIDE = %s, HAS_SITE = %s;" % (
        _b(pm.rk4), _b(pm.has_floor), _b(pm.fluid), _b(pm.selfcollide), _b(pm.site is not None)))
    A("  static constexpr bool ANY_DAMPING = %s;" % _b(any(d > 0 for d in pm.damp)))
    A("  static constexpr double DT = %s, GRAV_A = %s, GRAV_B = %s, FLOOR_Z = %s;" % (
        _d(pm.timestep), _d(pm.grav[0]), _d(pm.grav[1]), _d(pm.floor_z)))
    A("  static constexpr int parent[NB] = %s;" % _arr(pm.parent, str))
    A("  static constexpr int dofadr[NB] = %s;" % _arr(pm.dofadr, str))
    A("  static constexpr int dofnum[NB] = %s;" % _arr(pm.dofnum, str))
    A("  static constexpr double bpos[NB][2] = %s;" % _arr2(pm.bpos))
    A("  static constexpr double com[NB][2] = %s;" % _arr2(pm.com))
    A("  static constexpr double mass[NB] = %s;" % _arr(pm.mass))
    A("  static constexpr double izz[NB] = %s;" % _arr(pm.izz))
    A("  static constexpr int dof_type[NV] = %s;  // 0 slide a, 1 slide b, 2 hinge" % _arr(pm.dof_type, str))
    A("  static constexpr int dof_body[NV] = %s;" % _arr(pm.dof_body, str))
    A("  static constexpr double dof_sgn[NV] = %s;" % _arr(pm.dof_sgn))
    A("  static constexpr double jpos[NV][2] = %s;" % _arr2(pm.jpos))
    A("  static constexpr double qpos0[NV] = %s;" % _arr(pm.qpos0))
    A("  static constexpr double stiff[NV] = %s;" % _arr(pm.stiff))
    A("  static constexpr double damp[NV] = %s;" % _arr(pm.damp))
    A("  static constexpr double arma[NV] = %s;" % _arr(pm.arma))
    A("  static constexpr bool limited[NV] = %s;" % _arr(pm.limited, _b))
    A("  static constexpr double range[NV][2] = %s;" % _arr2(pm.rng))
    A("  static constexpr double lim_margin[NV] = %s;" % _arr(pm.lim_margin))
    A("  static constexpr double lim_K[NV] = %s;" % _arr(pm.lim_K))
    A("  static constexpr double lim_B[NV] = %s;" % _arr(pm.lim_B))
    A("  static constexpr double lim_solimp[NV][5] = %s;" % _arr2(pm.lim_solimp))
    A("  static constexpr double dof_invweight0[NV] = %s;" % _arr(pm.dof_invweight0))
    A("  static constexpr int act_dof[NU] = %s;" % _arr(pm.act_dof, str))
    A("  static constexpr double gear[NU] = %s;" % _arr(pm.gear))
    A("  static constexpr bool ctrllimited[NU] = %s;" % _arr(pm.ctrllimited, _b))
    A("  static constexpr double ctrlrange[NU][2] = %s;" % _arr2(pm.ctrlrange))
    if NG == 0:
        A("  static constexpr int g_body[1] = {0};")
        A("  static constexpr bool c_on[1] = {false};")
        A("  static constexpr double g_ea[1][2] = {{0,0}}, g_eb[1][2] = {{0,0}}, g_rad[1] = {0};")
        A("  static constexpr double c_mu[1]={0}, c_margin[1]={0}, c_K[1]={0}, c_B[1]={0}, c_RK[1]={0};")
        A("  static constexpr int c_dim[1] = {1};")
        A("  static constexpr double c_solimp[1][5] = {{0,0,0,0,0}};")
    else:
        A("  static constexpr int g_body[NG] = %s;" % _arr(pm.g_body, str))
        A("  static constexpr bool c_on[NG] = %s;" % _arr(pm.c_on, _b))
        A("  static constexpr double g_ea[NG][2] = %s;" % _arr2(pm.g_ea))
        A("  static constexpr double g_eb[NG][2] = %s;" % _arr2(pm.g_eb))
        A("  static constexpr double g_rad[NG] = %s;" % _arr(pm.g_rad))
        A("  static constexpr double c_mu[NG] = %s;" % _arr(pm.c_mu))
        A("  static constexpr int c_dim[NG] = %s;" % _arr(pm.c_dim, str))
        A("  static constexpr double c_margin[NG] = %s;" % _arr(pm.c_margin))
        A("  static constexpr double c_K[NG] = %s;" % _arr(pm.c_K))
        A("  static constexpr double c_B[NG] = %s;" % _arr(pm.c_B))
        A("  static constexpr double c_RK[NG] = %s;" % _arr(pm.c_RK))
        A("  static constexpr double c_solimp[NG][5] = %s;" % _arr2(pm.c_solimp))
    prs = pm.pairs or [dict(g1=0, g2=0, margin=0.0, K=0.0, B=0.0, solimp=modelspec.SOLIMP0, RK=0.0)]
    A("  static constexpr int NPAIR = %d;" % len(pm.pairs))
    A("  static constexpr int pair_g1[%d] = %s, pair_g2[%d] = %s;" % (
        len(prs), _arr([p["g1"] for p in prs], str), len(prs), _arr([p["g2"] for p in prs], str)))
    A("  static constexpr double pair_margin[%d] = %s;" % (len(prs), _arr([p["margin"] for p in prs])))
    A("  static constexpr double pair_K[%d] = %s;" % (len(prs), _arr([p["K"] for p in prs])))
    A("  static constexpr double pair_B[%d] = %s;" % (len(prs), _arr([p["B"] for p in prs])))
    A("  static constexpr double pair_RK[%d] = %s;" % (len(prs), _arr([p["RK"] for p in prs])))
    A("  static constexpr double pair_solimp[%d][5] = %s;" % (len(prs), _arr2([p["solimp"] for p in prs])))
    A("  static constexpr double fl_kv_lin[NB] = %s;" % _arr([f["kv_lin"] for f in pm.fl]))
    A("  static constexpr double fl_kv_ang[NB] = %s;" % _arr([f["kv_ang"] for f in pm.fl]))
    A("  static constexpr double fl_kq_a[NB] = %s;" % _arr([f["kq_a"] for f in pm.fl]))
    A("  static constexpr double fl_kq_b[NB] = %s;" % _arr([f["kq_b"] for f in pm.fl]))
    A("  static constexpr double fl_kq_ang[NB] = %s;" % _arr([f["kq_ang"] for f in pm.fl]))
    s = pm.site or (0, 0.0, 0.0)
    A("  static constexpr int site_body = %d;" % s[0])
    A("  static constexpr double site_pos[2] = {%s, %s};" % (_d(s[1]), _d(s[2])))
    A("};")
    return "\n".join(L)


CNAMES = {"InvertedPendulumEnv": "MdlInvertedPendulum", "InvertedDoublePendulumEnv": "MdlInvertedDoublePendulum",
          "SwimmerEnv": "MdlSwimmer", "HopperEnv": "MdlHopper", "Walker2dEnv": "MdlWalker2d",
          "HalfCheetahEnv": "MdlHalfCheetah"}


def emit_header():
    out = ["// GENERATED by model-based-rl_amd/compile_planar.py from modelspec.py -- do not edit.",
           "#pragma once", "namespace mbrl {", ""]
    for env in modelspec.ENV_ORDER:
        out.append(emit_model(PlanarModel(env), CNAMES[env]))
        out.append("")
    out.append("}  // namespace mbrl")
    return "\n".join(out) + "\n"


def write_header(path=None):
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "models_gen.hpp")
    txt = emit_header()
    old = open(path).read() if os.path.exists(path) else None
    if old != txt:
        with open(path, "w") as fh:
            fh.write(txt)
    return path
