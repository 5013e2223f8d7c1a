#!/usr/bin/env python3
"""Known-answer vectors for the cart-pole models from a closed-form Lagrangian (sympy).

Independent of the oracle and of the kernel: the equations of motion are derived symbolically from
T - V for a cart (mass m0) on a rail and 1 or 2 rigid poles (mass, com offset, inertia about the com),
with joint damping, actuator force gear*ctrl and gravity (gx, gz).  Body constants come from this
script's own capsule formulas (MuJoCo 2.0 volume pi r^2 (h + r)).  Output:
tests/golden/kat_cartpole.npz with random states (away from the joint limits) and the exact qacc.
"""
import math
import os

import numpy as np
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))


def capsule(r, h, rho=1000.0):
    mc, ms = rho * math.pi * r * r * h, rho * math.pi * r ** 3
    i_pe = mc * (3 * r * r + h * h) / 12 + ms * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8)
    return mc + ms, i_pe


def derive(npoles):
    t = sp.symbols("t")
    x = sp.Function("x")(t)
    th = [sp.Function("th%d" % i)(t) for i in range(npoles)]
    m0, gx, gz, u = sp.symbols("m0 gx gz u")
    m = sp.symbols("m1:%d" % (npoles + 1))
    I = sp.symbols("I1:%d" % (npoles + 1))
    l = sp.symbols("l1:%d" % (npoles + 1))      # com distance along the pole
    L = sp.symbols("L1:%d" % (npoles + 1))      # pole length (next hinge)
    a0 = sp.symbols("a1:%d" % (npoles + 1))     # fixed tilt of the pole axis from +z at theta = 0
    d = sp.symbols("d0:%d" % (npoles + 1))      # damping
    q = [x] + th
    # hinge about +y: positive angle rotates +z towards +x
    T = m0 * sp.diff(x, t) ** 2 / 2
    V = -m0 * gx * x
    px, pz, ang = x, 0, 0
    for i in range(npoles):
        ang = ang + th[i]
        cx = px + l[i] * sp.sin(ang + a0[i])
        cz = pz + l[i] * sp.cos(ang + a0[i])
        T += m[i] * (sp.diff(cx, t) ** 2 + sp.diff(cz, t) ** 2) / 2 + I[i] * sp.diff(ang, t) ** 2 / 2
        V += -m[i] * (gx * cx + gz * cz)
        px = px + L[i] * sp.sin(ang + a0[i])
        pz = pz + L[i] * sp.cos(ang + a0[i])
    Lg = T - V
    qd = [sp.diff(v, t) for v in q]
    eqs = []
    for i, (qi, qdi) in enumerate(zip(q, qd)):
        Q = -d[i] * qdi + (u if i == 0 else 0)
        eqs.append(sp.diff(sp.diff(Lg, qdi), t) - sp.diff(Lg, qi) - Q)
    qdd = [sp.diff(v, t, 2) for v in q]
    # the equations are linear in qdd: eqs = Mm*qdd - rhs; solve numerically (high precision) per state
    acc_syms = sp.symbols("acc0:%d" % (npoles + 1))
    syms_q = sp.symbols("q0:%d" % (npoles + 1))
    syms_v = sp.symbols("v0:%d" % (npoles + 1))
    flat = []
    for e in eqs:
        e = sp.expand(e)
        for a, s_ in zip(qdd, acc_syms):
            e = e.subs(a, s_)
        for qdi, sv in zip(qd, syms_v):
            e = e.subs(qdi, sv)
        for qi, sq in zip(q, syms_q):
            e = e.subs(qi, sq)
        flat.append(e)
    Mm, rhs = sp.linear_eq_to_matrix(flat, list(acc_syms))
    args = list(syms_q) + list(syms_v) + [u, m0, gx, gz] + list(m) + list(I) + list(l) + list(L) + list(a0) + list(d)
    fM, fr = sp.lambdify(args, Mm, "mpmath"), sp.lambdify(args, rhs, "mpmath")

    def f(*vals):
        import mpmath
        return list(mpmath.lu_solve(fM(*vals), fr(*vals)))
    return f


def main():
    import mpmath
    mpmath.mp.dps = 40
    rng = np.random.default_rng(5)
    out = {}
    # InvertedPendulum: cart capsule r=.1 h=.2; pole from (0,0,0) to (0.001,0,0.6) r=.049; damping 1,1; gear 100
    m0, _ = capsule(0.1, 0.2)
    hp = math.sqrt(0.001 ** 2 + 0.6 ** 2)
    m1, I1 = capsule(0.049, hp)
    f = derive(1)
    N = 32
    q = np.column_stack([rng.uniform(-0.9, 0.9, N), rng.uniform(-1.4, 1.4, N)])
    v = rng.normal(0, 2.0, (N, 2))
    c = rng.uniform(-3, 3, (N, 1))
    acc = np.zeros((N, 2))
    for i in range(N):
        r = f(q[i, 0], q[i, 1], v[i, 0], v[i, 1], 100.0 * c[i, 0], m0, 0.0, -9.81, m1, I1, hp / 2, hp,
              math.atan2(0.001, 0.6), 1.0, 1.0)
        acc[i] = [float(z) for z in r]
    out.update(ip_q=q, ip_v=v, ip_c=c, ip_qacc=acc)
    # InvertedDoublePendulum: poles r=.045 h=.6; damping .05; gear 500 (ctrl clamped to [-1,1]); gravity (1e-5,0,-9.81)
    m1, I1 = capsule(0.045, 0.6)
    f = derive(2)
    q = np.column_stack([rng.uniform(-0.9, 0.9, N), rng.uniform(-3, 3, N), rng.uniform(-3, 3, N)])
    v = rng.normal(0, 2.0, (N, 3))
    c = rng.uniform(-1.5, 1.5, (N, 1))
    acc = np.zeros((N, 3))
    for i in range(N):
        u = 500.0 * min(1.0, max(-1.0, c[i, 0]))
        r = f(q[i, 0], q[i, 1], q[i, 2], v[i, 0], v[i, 1], v[i, 2], u, m0, 1e-5, -9.81, m1, m1, I1, I1,
              0.3, 0.3, 0.6, 0.6, 0.0, 0.0, 0.05, 0.05, 0.05)
        acc[i] = [float(z) for z in r]
    out.update(idp_q=q, idp_v=v, idp_c=c, idp_qacc=acc)
    np.savez_compressed(os.path.join(HERE, "kat_cartpole.npz"), **out)
    print("wrote kat_cartpole.npz")


if __name__ == "__main__":
    main()
