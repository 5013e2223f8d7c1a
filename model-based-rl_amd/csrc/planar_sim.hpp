// planar_sim.hpp -- per-lane planar rigid-body simulator used by the HIP kernels.
//
// One instance = one rollout (one lane).  Everything is templated on a compile-time model
// table (models_gen.hpp) so that, after full unrolling, every array index is a constant
// and the working set lives in VGPRs.  The scalar type T is `double` on the device; tests
// instantiate it on the host with a counting type to obtain the algorithmic flop count
// that bench.py prices against the fp64 vector roofline.
//
// What it computes (the arithmetic the reference reaches through mujoco-py; SURVEY.md 8a
// rows a8/a9):  MuJoCo 2.0 `mj_step` for planar hinge/slide trees --
//   kinematics -> composite inertias -> recursive Newton-Euler bias -> passive (spring,
//   damper, inertia-box fluid) -> actuation -> joint-limit + plane/capsule pyramidal
//   contact rows -> primal Newton solve to the exact active-set fixed point
//   (opt.tolerance = 0, mujoco/build.py:35) -> Euler (implicit joint damping) or RK4;
// followed by the env reward (mujoco/envs/*.py `step`).
//
// MI355X-specific formulation (not how MuJoCo or the oracle do it):
//   * planar spatial algebra about the root body origin O: motion [w, va, vb], 3x3
//     symmetric inertias (6 numbers);
//   * the constraint Jacobian is never formed.  A contact row is a 3-vector `w` acting on
//     the spatial velocity of its body, so  J x  is one tree pass,  J^T f  is one backward
//     pass, and the Newton Hessian  M + J^T D J  is the composite-rigid-body algorithm run
//     on inertias augmented by  sum D w w^T  -- no row storage, no dense J^T D J product;
//   * the mass matrix / Hessian is factorised with the tree-sparse L^T D L scheme, with
//     the sparsity resolved at compile time.
#pragma once
#include <cmath>
#include <cstdint>

#include "models_gen.hpp"

#if defined(__HIPCC__)
#define MBRL_HD __host__ __device__ __forceinline__
#else
#define MBRL_HD inline
#endif
#define MBRL_UNROLL _Pragma("unroll")

namespace mbrl {

constexpr double kMinVal = 1e-15;
constexpr unsigned ST_NONFINITE = 1u, ST_NOCONVERGE = 2u, ST_OVERFLOW = 4u, ST_DEGENERATE = 8u;
constexpr int kMaxNewton = 100;
constexpr int kMaxLineSearch = 40;

// ---- scalar helpers (overloaded by the host-side counting type) ----
MBRL_HD double t_abs(double x) { return fabs(x); }
MBRL_HD double t_max(double x, double y) { return fmax(x, y); }
MBRL_HD double t_min(double x, double y) { return fmin(x, y); }
MBRL_HD double t_sqrt(double x) { return sqrt(x); }
MBRL_HD double t_pow(double x, double y) { return pow(x, y); }
MBRL_HD double t_fmod(double x, double y) { return fmod(x, y); }
MBRL_HD double t_cos(double x) { return cos(x); }
MBRL_HD void t_sincos(double x, double *s, double *c) {
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(x, s, c);
#else
    *s = sin(x); *c = cos(x);
#endif
}
MBRL_HD bool t_finite(double x) { return x - x == 0.0; }
MBRL_HD double t_val(double x) { return x; }

template <class T>
struct Sp {  // planar spatial vector: angular part w, linear part (a, b)
    T w, a, b;
};

template <class M>
struct Topo {
    // dof j moves the body of dof i (j <= i in index order)?
    static constexpr bool body_anc(int anc, int b) {
        while (b >= 0) {
            if (b == anc) return true;
            b = M::parent[b];
        }
        return false;
    }
    static constexpr bool dof_anc(int j, int i) { return j <= i && body_anc(M::dof_body[j], M::dof_body[i]); }
    static constexpr int ncand() { return 2 * M::NGEOM; }
};

template <class M, class T>
struct Sim {
    static constexpr int NB = M::NB, NV = M::NV, NU = M::NU, NG = M::NG, NC = 2 * M::NG;
    using TP = Topo<M>;

    // ---- state ----
    T q[NV], v[NV], ctrl[NU];
    // ---- position-dependent ----
    T Oa, Ob;                            // absolute position of the root body origin
    T pa[NB], pb[NB], c[NB], s[NB];      // body origin relative to O, orientation
    T ra[NV], rb[NV];                    // hinge lever r = O - anchor
    T ha[NB], hb[NB], IO[NB];            // first moment m*(com - O) and inertia about O
    // ---- forces / accelerations ----
    T fsm[NV], qs[NV], qacc[NV], fcon[NV];
    // ---- constraint candidates ----
    uint64_t lim_on, con_on;             // active candidates (bit j / bit c)
    T lsg[NV], lD[NV], laref[NV];        // limit rows: J = lsg (+-1) on dof j
    T cpa[NC], cpb[NC], cD[NC], can[NC], cat[NC];  // contact point rel. O, D, aref (normal / tangent parts)
    static constexpr int NPR = M::NPAIR > 0 ? M::NPAIR : 1;
    uint64_t pair_on;                    // active capsule-capsule pairs
    T pw0[NPR], pw1[NPR], pw2[NPR], pD[NPR], paref[NPR];  // row = w . (V_b2 - V_b1)
    T site_a, site_b;
    unsigned status;
    int newton_iters;

    // ------------------------------------------------------------------ motion subspace
    MBRL_HD Sp<T> Scol(int j) const {  // S_j (compile-time j after unrolling)
        Sp<T> r;
        if (M::dof_type[j] == 0) { r.w = T(0.0); r.a = T(1.0); r.b = T(0.0); }
        else if (M::dof_type[j] == 1) { r.w = T(0.0); r.a = T(0.0); r.b = T(1.0); }
        else { r.w = T(M::dof_sgn[j]); r.a = -M::dof_sgn[j] * rb[j]; r.b = M::dof_sgn[j] * ra[j]; }
        return r;
    }

    // spatial velocity of every body for generalized velocity x
    MBRL_HD void vel_pass(const T *x, Sp<T> *V) const {
        MBRL_UNROLL
        for (int b = 0; b < NB; b++) {
            Sp<T> u;
            if (M::parent[b] >= 0) u = V[M::parent[b]];
            else { u.w = T(0.0); u.a = T(0.0); u.b = T(0.0); }
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                if (M::dof_type[j] == 0) u.a = u.a + x[j];
                else if (M::dof_type[j] == 1) u.b = u.b + x[j];
                else {
                    const T xs = M::dof_sgn[j] * x[j];
                    u.w = u.w + xs;
                    u.a = u.a - rb[j] * xs;
                    u.b = u.b + ra[j] * xs;
                }
            }
            V[b] = u;
        }
    }

    // generalized forces from per-body spatial forces (F is destroyed: subtree sums)
    MBRL_HD void back_pass(Sp<T> *F, T *g) const {
        MBRL_UNROLL
        for (int b = NB - 1; b >= 0; b--) {
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                if (M::dof_type[j] == 0) g[j] = F[b].a;
                else if (M::dof_type[j] == 1) g[j] = F[b].b;
                else g[j] = M::dof_sgn[j] * (F[b].w - rb[j] * F[b].a + ra[j] * F[b].b);
            }
            if (M::parent[b] >= 0) {
                const int p = M::parent[b];
                F[p].w = F[p].w + F[b].w; F[p].a = F[p].a + F[b].a; F[p].b = F[p].b + F[b].b;
            }
        }
    }

    // rigid-body momentum  I_b * V  (about O)
    MBRL_HD Sp<T> inert_mul(int b, const Sp<T> &V) const {
        Sp<T> r;
        r.w = IO[b] * V.w + (ha[b] * V.b - hb[b] * V.a);
        r.a = M::mass[b] * V.a - V.w * hb[b];
        r.b = M::mass[b] * V.b + V.w * ha[b];
        return r;
    }

    // ------------------------------------------------------------------ kinematics
    MBRL_HD void kinematics() {
        T absa = T(0.0), absb = T(0.0);
        MBRL_UNROLL
        for (int b = 0; b < NB; b++) {
            T px, py, cc, ss;
            if (M::parent[b] < 0) {
                px = T(M::bpos[b][0]); py = T(M::bpos[b][1]); cc = T(1.0); ss = T(0.0);
            } else {
                const int p = M::parent[b];
                px = pa[p] + (c[p] * M::bpos[b][0] - s[p] * M::bpos[b][1]);
                py = pb[p] + (s[p] * M::bpos[b][0] + c[p] * M::bpos[b][1]);
                cc = c[p]; ss = s[p];
            }
            T anca[3], ancb[3];  // anchors of this body's hinges (same frame as px,py)
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                const T dq = q[j] - M::qpos0[j];
                if (M::dof_type[j] == 0) px = px + dq;
                else if (M::dof_type[j] == 1) py = py + dq;
                else {
                    const T ax = px + (cc * M::jpos[j][0] - ss * M::jpos[j][1]);
                    const T ay = py + (ss * M::jpos[j][0] + cc * M::jpos[j][1]);
                    T sj, cj;
                    t_sincos(M::dof_sgn[j] * dq, &sj, &cj);
                    const T cn = cc * cj - ss * sj, sn = ss * cj + cc * sj;
                    cc = cn; ss = sn;
                    px = ax - (cc * M::jpos[j][0] - ss * M::jpos[j][1]);
                    py = ay - (ss * M::jpos[j][0] + cc * M::jpos[j][1]);
                    anca[k] = ax; ancb[k] = ay;
                }
            }
            if (M::parent[b] < 0) {  // root: its final origin is O; store relative quantities
                absa = px; absb = py;
                MBRL_UNROLL
                for (int k = 0; k < M::dofnum[b]; k++) {
                    const int j = M::dofadr[b] + k;
                    if (M::dof_type[j] == 2) { ra[j] = px - anca[k]; rb[j] = py - ancb[k]; }
                }
                pa[b] = T(0.0); pb[b] = T(0.0);
            } else {
                MBRL_UNROLL
                for (int k = 0; k < M::dofnum[b]; k++) {
                    const int j = M::dofadr[b] + k;
                    if (M::dof_type[j] == 2) { ra[j] = -anca[k]; rb[j] = -ancb[k]; }
                }
                pa[b] = px; pb[b] = py;
            }
            c[b] = cc; s[b] = ss;
            const T cwa = pa[b] + (cc * M::com[b][0] - ss * M::com[b][1]);
            const T cwb = pb[b] + (ss * M::com[b][0] + cc * M::com[b][1]);
            ha[b] = M::mass[b] * cwa; hb[b] = M::mass[b] * cwb;
            IO[b] = M::izz[b] + M::mass[b] * (cwa * cwa + cwb * cwb);
        }
        Oa = absa; Ob = absb;
        if (M::HAS_SITE) {
            const int b = M::site_body;
            site_a = Oa + pa[b] + (c[b] * M::site_pos[0] - s[b] * M::site_pos[1]);
            site_b = Ob + pb[b] + (s[b] * M::site_pos[0] + c[b] * M::site_pos[1]);
        }
    }

    // ------------------------------------------------------------------ impedance / regulariser
    MBRL_HD static T impedance(const double (&si)[5], T pm /* pos - margin */) {
        const double dmin = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
        if (dmin == dmax || width <= kMinVal) return T(0.5 * (dmin + dmax));
        const T x = t_abs(pm) / width;
        T y;
        if (power == 2.0 && mid == 0.5) {
            const T u = T(1.0) - x;
            y = (t_val(x) <= 0.5) ? T(2.0) * x * x : T(1.0) - T(2.0) * u * u;
        } else if (power == 1.0) {
            y = x;
        } else {
            y = (t_val(x) <= mid) ? t_pow(x, T(power)) / pow(mid, power - 1.0)
                                  : T(1.0) - t_pow(T(1.0) - x, T(power)) / pow(1.0 - mid, power - 1.0);
        }
        T imp = T(dmin) + y * (dmax - dmin);
        if (t_val(x) >= 1.0) imp = T(dmax);
        if (t_val(x) <= 0.0) imp = T(dmin);
        return imp;
    }

    // constraint candidates: joint limits and capsule-end/plane contacts; needs kinematics(), v
    MBRL_HD void make_constraints() {
        lim_on = 0; con_on = 0; pair_on = 0;
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            if (!M::limited[j]) continue;
            const T dlo = q[j] - M::range[j][0], dhi = M::range[j][1] - q[j];
            const bool lo = t_val(dlo) < M::lim_margin[j], hi = t_val(dhi) < M::lim_margin[j];
            if (lo || hi) {
                const T dist = lo ? dlo : dhi;
                const T sg = lo ? T(1.0) : T(-1.0);
                const T imp = impedance(M::lim_solimp[j], dist - M::lim_margin[j]);
                const T R = t_max(T(kMinVal), (T(1.0) - imp) * M::dof_invweight0[j] / imp);
                lsg[j] = sg; lD[j] = T(1.0) / R;
                laref[j] = -M::lim_B[j] * (sg * v[j]) - M::lim_K[j] * imp * (dist - M::lim_margin[j]);
                lim_on |= (uint64_t(1) << j);
            }
        }
        if (!M::HAS_FLOOR && M::NPAIR == 0) return;
        Sp<T> V[NB];
        bool haveV = false;
        MBRL_UNROLL
        for (int k = 0; k < M::NPAIR; k++) {
            // capsule-capsule: closest points of the two axis segments, then sphere-sphere
            const int g1 = M::pair_g1[k], g2 = M::pair_g2[k], b1 = M::g_body[g1], b2 = M::g_body[g2];
            const T p1a = pa[b1] + (c[b1] * M::g_ea[g1][0] - s[b1] * M::g_ea[g1][1]);
            const T p1b = pb[b1] + (s[b1] * M::g_ea[g1][0] + c[b1] * M::g_ea[g1][1]);
            const T q1a = pa[b1] + (c[b1] * M::g_eb[g1][0] - s[b1] * M::g_eb[g1][1]);
            const T q1b = pb[b1] + (s[b1] * M::g_eb[g1][0] + c[b1] * M::g_eb[g1][1]);
            const T p2a = pa[b2] + (c[b2] * M::g_ea[g2][0] - s[b2] * M::g_ea[g2][1]);
            const T p2b = pb[b2] + (s[b2] * M::g_ea[g2][0] + c[b2] * M::g_ea[g2][1]);
            const T q2a = pa[b2] + (c[b2] * M::g_eb[g2][0] - s[b2] * M::g_eb[g2][1]);
            const T q2b = pb[b2] + (s[b2] * M::g_eb[g2][0] + c[b2] * M::g_eb[g2][1]);
            const T d1a = q1a - p1a, d1b = q1b - p1b, d2a = q2a - p2a, d2b = q2b - p2b;
            const T rra = p1a - p2a, rrb = p1b - p2b;
            const T aa = d1a * d1a + d1b * d1b, ee = d2a * d2a + d2b * d2b, ff = d2a * rra + d2b * rrb;
            const T cc = d1a * rra + d1b * rrb, bb = d1a * d2a + d1b * d2b;
            const T den = aa * ee - bb * bb;
            T ss = (t_val(den) < 1e-12 * t_val(aa) * t_val(ee)) ? T(0.0) : (bb * ff - cc * ee) / den;
            ss = t_min(T(1.0), t_max(T(0.0), ss));
            T tt = (bb * ss + ff) / ee;
            if (t_val(tt) < 0.0) { tt = T(0.0); ss = t_min(T(1.0), t_max(T(0.0), -cc / aa)); }
            else if (t_val(tt) > 1.0) { tt = T(1.0); ss = t_min(T(1.0), t_max(T(0.0), (bb - cc) / aa)); }
            const T c1a = p1a + ss * d1a, c1b = p1b + ss * d1b, c2a = p2a + tt * d2a, c2b = p2b + tt * d2b;
            const T dfa = c2a - c1a, dfb = c2b - c1b;
            const T len = t_sqrt(dfa * dfa + dfb * dfb);
            const T dist = len - M::g_rad[g1] - M::g_rad[g2];
            if (t_val(dist) < M::pair_margin[k]) {
                if (!haveV) { vel_pass(v, V); haveV = true; }
                if (t_val(len) < 1e-10) status |= ST_DEGENERATE;  // crossing axes: normal is round-off noise
                T na = T(1.0), nb = T(0.0);
                if (!(t_val(len) < kMinVal)) { na = dfa / len; nb = dfb / len; }
                const T Pa = c1a + na * (M::g_rad[g1] + T(0.5) * dist);
                const T Pb = c1b + nb * (M::g_rad[g1] + T(0.5) * dist);
                const T pm = dist - M::pair_margin[k];
                const T imp = impedance(M::pair_solimp[k], pm);
                const T R = t_max(T(kMinVal), (T(1.0) - imp) * M::pair_RK[k] / imp);
                pw0[k] = nb * Pa - na * Pb; pw1[k] = na; pw2[k] = nb; pD[k] = T(1.0) / R;
                const T vel = pw0[k] * (V[b2].w - V[b1].w) + pw1[k] * (V[b2].a - V[b1].a) + pw2[k] * (V[b2].b - V[b1].b);
                paref[k] = -M::pair_B[k] * vel - M::pair_K[k] * imp * pm;
                pair_on |= (uint64_t(1) << k);
            }
        }
        if (!M::HAS_FLOOR) return;
        MBRL_UNROLL
        for (int g = 0; g < M::NGEOM; g++) {
            if (!M::c_on[g]) continue;
            const int b = M::g_body[g];
            MBRL_UNROLL
            for (int e = 0; e < 2; e++) {
                const int ci = 2 * g + e;
                const double la = e ? M::g_eb[g][0] : M::g_ea[g][0], lb = e ? M::g_eb[g][1] : M::g_ea[g][1];
                const T ea = pa[b] + (c[b] * la - s[b] * lb);   // sphere centre rel. O
                const T eb = pb[b] + (s[b] * la + c[b] * lb);
                const T dist = ((Ob + eb) - M::FLOOR_Z) - M::g_rad[g];
                if (t_val(dist) < M::c_margin[g]) {
                    if (!haveV) { vel_pass(v, V); haveV = true; }
                    const T Pa = ea, Pb = eb - (M::g_rad[g] + T(0.5) * dist);
                    const T pm = dist - M::c_margin[g];
                    const T imp = impedance(M::c_solimp[g], pm);
                    const T R = t_max(T(kMinVal), (T(1.0) - imp) * M::c_RK[g] / imp);
                    const T vn = V[b].b + V[b].w * Pa, vt = V[b].a - V[b].w * Pb;
                    cpa[ci] = Pa; cpb[ci] = Pb; cD[ci] = T(1.0) / R;
                    can[ci] = -M::c_B[g] * vn - M::c_K[g] * imp * pm;
                    cat[ci] = -M::c_B[g] * vt;
                    con_on |= (uint64_t(1) << ci);
                }
            }
        }
    }

    // ------------------------------------------------------------------ smooth dynamics
    // qfrc_smooth = passive - bias + actuation
    MBRL_HD void smooth_forces() {
        Sp<T> V[NB], A[NB], F[NB];
        MBRL_UNROLL
        for (int b = 0; b < NB; b++) {
            Sp<T> u, a;
            if (M::parent[b] >= 0) { u = V[M::parent[b]]; a = A[M::parent[b]]; }
            else {
                u.w = T(0.0); u.a = T(0.0); u.b = T(0.0);
                a.w = T(0.0); a.a = T(-M::GRAV_A); a.b = T(-M::GRAV_B);
            }
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                const Sp<T> S = Scol(j);
                // Sdot = crm(u) S = (0, -u.w*S.b + u.b*S.w, u.w*S.a - u.a*S.w)
                if (M::dof_type[j] == 2) {
                    a.a = a.a + (u.b * S.w - u.w * S.b) * v[j];
                    a.b = a.b + (u.w * S.a - u.a * S.w) * v[j];
                    u.w = u.w + S.w * v[j]; u.a = u.a + S.a * v[j]; u.b = u.b + S.b * v[j];
                } else if (M::dof_type[j] == 0) {
                    a.b = a.b + u.w * v[j];
                    u.a = u.a + v[j];
                } else {
                    a.a = a.a - u.w * v[j];
                    u.b = u.b + v[j];
                }
            }
            V[b] = u; A[b] = a;
            const Sp<T> Ia = inert_mul(b, a), Iv = inert_mul(b, u);
            // crf(u) Iv = (u.a*Iv.b - u.b*Iv.a, -u.w*Iv.b, u.w*Iv.a)
            F[b].w = Ia.w + (u.a * Iv.b - u.b * Iv.a);
            F[b].a = Ia.a - u.w * Iv.b;
            F[b].b = Ia.b + u.w * Iv.a;
            if (M::FLUID) {
                // inertia-box fluid model: velocity at the body com in body axes
                const T rca = ha[b] / M::mass[b], rcb = hb[b] / M::mass[b];   // com - O
                const T vca = u.a - u.w * rcb, vcb = u.b + u.w * rca;
                const T la = c[b] * vca + s[b] * vcb, lb = -s[b] * vca + c[b] * vcb;
                const T fla = -M::fl_kv_lin[b] * la - M::fl_kq_a[b] * t_abs(la) * la;
                const T flb = -M::fl_kv_lin[b] * lb - M::fl_kq_b[b] * t_abs(lb) * lb;
                const T tq = -M::fl_kv_ang[b] * u.w - M::fl_kq_ang[b] * t_abs(u.w) * u.w;
                const T fa = c[b] * fla - s[b] * flb, fb = s[b] * fla + c[b] * flb;
                F[b].w = F[b].w - (tq + (rca * fb - rcb * fa));
                F[b].a = F[b].a - fa;
                F[b].b = F[b].b - fb;
            }
        }
        T bias[NV];
        back_pass(F, bias);
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) fsm[j] = (-M::stiff[j] * q[j] - M::damp[j] * v[j]) - bias[j];
        MBRL_UNROLL
        for (int u = 0; u < NU; u++) {
            T cu = ctrl[u];
            if (M::ctrllimited[u]) cu = t_min(T(M::ctrlrange[u][1]), t_max(T(M::ctrlrange[u][0]), cu));
            fsm[M::act_dof[u]] = fsm[M::act_dof[u]] + M::gear[u] * cu;
        }
    }

    // ------------------------------------------------------------------ CRBA on 3x3 symmetric inertias
    struct Sym3 { T ww, wa, wb, aa, ab, bb; };

    MBRL_HD void body_inertias(Sym3 *Ic) const {
        MBRL_UNROLL
        for (int b = 0; b < NB; b++) {
            Ic[b].ww = IO[b]; Ic[b].wa = -hb[b]; Ic[b].wb = ha[b];
            Ic[b].aa = T(M::mass[b]); Ic[b].ab = T(0.0); Ic[b].bb = T(M::mass[b]);
        }
    }

    // H (lower triangle, tree-sparse) = sum_b A_b^T Ic_b A_b + diag(add)
    MBRL_HD void crba(Sym3 *Ic, T (*H)[NV], const T *diag_add) const {
        MBRL_UNROLL
        for (int b = NB - 1; b >= 1; b--) {
            const int p = M::parent[b];
            Ic[p].ww = Ic[p].ww + Ic[b].ww; Ic[p].wa = Ic[p].wa + Ic[b].wa; Ic[p].wb = Ic[p].wb + Ic[b].wb;
            Ic[p].aa = Ic[p].aa + Ic[b].aa; Ic[p].ab = Ic[p].ab + Ic[b].ab; Ic[p].bb = Ic[p].bb + Ic[b].bb;
        }
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) {
            const Sym3 &I = Ic[M::dof_body[i]];
            const Sp<T> S = Scol(i);
            Sp<T> F;
            if (M::dof_type[i] == 0) { F.w = I.wa; F.a = I.aa; F.b = I.ab; }
            else if (M::dof_type[i] == 1) { F.w = I.wb; F.a = I.ab; F.b = I.bb; }
            else {
                F.w = I.ww * S.w + I.wa * S.a + I.wb * S.b;
                F.a = I.wa * S.w + I.aa * S.a + I.ab * S.b;
                F.b = I.wb * S.w + I.ab * S.a + I.bb * S.b;
            }
            MBRL_UNROLL
            for (int j = 0; j <= i; j++) {
                if (!TP::dof_anc(j, i)) continue;
                T h;
                if (M::dof_type[j] == 0) h = F.a;
                else if (M::dof_type[j] == 1) h = F.b;
                else { const Sp<T> Sj = Scol(j); h = Sj.w * F.w + Sj.a * F.a + Sj.b * F.b; }
                if (j == i) h = h + diag_add[i];
                H[i][j] = h;
            }
        }
    }

    // in-place tree-sparse L^T D L factorisation; returns false if a pivot is not positive
    MBRL_HD bool factor(T (*H)[NV]) const {
        bool ok = true;
        MBRL_UNROLL
        for (int k = NV - 1; k >= 0; k--) {
            if (!(t_val(H[k][k]) > 0.0)) ok = false;
            const T inv = T(1.0) / H[k][k];
            MBRL_UNROLL
            for (int i = k - 1; i >= 0; i--) {
                if (!TP::dof_anc(i, k)) continue;
                const T a = H[k][i] * inv;
                MBRL_UNROLL
                for (int j = i; j >= 0; j--) {
                    if (!TP::dof_anc(j, i)) continue;
                    H[i][j] = H[i][j] - a * H[k][j];
                }
                H[k][i] = a;
            }
            H[k][k] = inv;  // store 1/D
        }
        return ok;
    }

    MBRL_HD void solve(const T (*H)[NV], T *x) const {
        MBRL_UNROLL
        for (int i = NV - 1; i >= 0; i--) {
            MBRL_UNROLL
            for (int j = i - 1; j >= 0; j--)
                if (TP::dof_anc(j, i)) x[j] = x[j] - H[i][j] * x[i];
        }
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) x[i] = x[i] * H[i][i];
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) {
            MBRL_UNROLL
            for (int j = i - 1; j >= 0; j--)
                if (TP::dof_anc(j, i)) x[i] = x[i] - H[i][j] * x[j];
        }
    }

    // ------------------------------------------------------------------ constraint rows
    // Visit every row of every active candidate.  f(bit, D, mult, jar_of(x)...) is given the
    // row residual pieces through the functor; rows of a contact: n+mu t, n-mu t, n (x2).
    // Body spatial velocities Va (for x = qacc) are supplied by the caller.
    template <class Fn>
    MBRL_HD void for_rows(const Sp<T> *Va, const T *x, Fn &&fn) const {
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            if (!M::limited[j]) continue;
            if (lim_on & (uint64_t(1) << j)) fn.limit(j, lsg[j] * x[j]);
        }
        MBRL_UNROLL
        for (int k = 0; k < M::NPAIR; k++) {
            if (pair_on & (uint64_t(1) << k)) {
                const int b1 = M::g_body[M::pair_g1[k]], b2 = M::g_body[M::pair_g2[k]];
                fn.pair(k, b1, b2, pw0[k] * (Va[b2].w - Va[b1].w) + pw1[k] * (Va[b2].a - Va[b1].a) +
                                       pw2[k] * (Va[b2].b - Va[b1].b));
            }
        }
        if (!M::HAS_FLOOR) return;
        MBRL_UNROLL
        for (int g = 0; g < M::NGEOM; g++) {
            if (!M::c_on[g]) continue;
            const int b = M::g_body[g];
            MBRL_UNROLL
            for (int e = 0; e < 2; e++) {
                const int ci = 2 * g + e;
                if (con_on & (uint64_t(1) << ci)) {
                    const T vn = Va[b].b + Va[b].w * cpa[ci];
                    const T vt = Va[b].a - Va[b].w * cpb[ci];
                    fn.contact(g, ci, b, vn, vt);
                }
            }
        }
    }

    struct GradPass {  // residuals at qacc -> active mask, body forces, Hessian augmentation
        Sim *S; Sp<T> *Fc; Sym3 *Ic; T *flim; T *hdiag; uint64_t mask;
        MBRL_HD void limit(int j, T jx) {
            const T jar = jx - S->laref[j];
            if (t_val(jar) < 0.0) {
                mask |= uint64_t(1) << j;
                flim[j] = S->lsg[j] * (-S->lD[j] * jar);
                hdiag[j] = hdiag[j] + S->lD[j];
            }
        }
        MBRL_HD void row(int ci, int b, int which, T jar, T wn, T wt, T mult) {
            if (t_val(jar) < 0.0) {
                mask |= uint64_t(1) << (NV + 3 * ci + which);
                const T D = S->cD[ci] * mult;
                const T f = -D * jar;
                // w = wn*(Pa, 0, 1) + wt*(-Pb, 1, 0)
                const T w0 = wn * S->cpa[ci] - wt * S->cpb[ci], w1 = wt, w2 = wn;
                Fc[b].w = Fc[b].w + f * w0; Fc[b].a = Fc[b].a + f * w1; Fc[b].b = Fc[b].b + f * w2;
                if (Ic) {
                    Ic[b].ww = Ic[b].ww + D * w0 * w0; Ic[b].wa = Ic[b].wa + D * w0 * w1;
                    Ic[b].wb = Ic[b].wb + D * w0 * w2; Ic[b].aa = Ic[b].aa + D * w1 * w1;
                    Ic[b].ab = Ic[b].ab + D * w1 * w2; Ic[b].bb = Ic[b].bb + D * w2 * w2;
                }
            }
        }
        MBRL_HD void pair(int k, int b1, int b2, T vrel) {
            const T jar = vrel - S->paref[k];
            if (t_val(jar) < 0.0) {
                mask |= uint64_t(1) << (NV + 3 * NC + k);
                const T f = -S->pD[k] * jar;
                Fc[b2].w = Fc[b2].w + f * S->pw0[k]; Fc[b2].a = Fc[b2].a + f * S->pw1[k]; Fc[b2].b = Fc[b2].b + f * S->pw2[k];
                Fc[b1].w = Fc[b1].w - f * S->pw0[k]; Fc[b1].a = Fc[b1].a - f * S->pw1[k]; Fc[b1].b = Fc[b1].b - f * S->pw2[k];
            }
        }
        MBRL_HD void contact(int g, int ci, int b, T vn, T vt) {
            const T rn = vn - S->can[ci], rt = vt - S->cat[ci];
            if (M::c_dim[g] == 1) { row(ci, b, 2, rn, T(1.0), T(0.0), T(1.0)); return; }
            const double mu = M::c_mu[g];
            row(ci, b, 0, rn + mu * rt, T(1.0), T(mu), T(1.0));
            row(ci, b, 1, rn - mu * rt, T(1.0), T(-mu), T(1.0));
            row(ci, b, 2, rn, T(1.0), T(0.0), T(2.0));
        }
    };

    struct LinePass {  // derivative of the cost along the search direction at step alpha
        const Sim *S; const Sp<T> *Vs; const T *sx; T alpha; T g, h; uint64_t mask;
        MBRL_HD void acc(int bit, T jar, T jv, T D) {
            const T r = jar + alpha * jv;
            if (t_val(r) < 0.0) { mask |= uint64_t(1) << bit; g = g + D * r * jv; h = h + D * jv * jv; }
        }
        MBRL_HD void limit(int j, T jx) { acc(j, jx - S->laref[j], S->lsg[j] * sx[j], S->lD[j]); }
        MBRL_HD void pair(int k, int b1, int b2, T vrel) {
            const T sv = S->pw0[k] * (Vs[b2].w - Vs[b1].w) + S->pw1[k] * (Vs[b2].a - Vs[b1].a) +
                         S->pw2[k] * (Vs[b2].b - Vs[b1].b);
            acc(NV + 3 * NC + k, vrel - S->paref[k], sv, S->pD[k]);
        }
        MBRL_HD void contact(int gi, int ci, int b, T vn, T vt) {
            const T rn = vn - S->can[ci], rt = vt - S->cat[ci];
            const T sn = Vs[b].b + Vs[b].w * S->cpa[ci], st = Vs[b].a - Vs[b].w * S->cpb[ci];
            if (M::c_dim[gi] == 1) { acc(NV + 3 * ci + 2, rn, sn, S->cD[ci]); return; }
            const double mu = M::c_mu[gi];
            acc(NV + 3 * ci + 0, rn + mu * rt, sn + mu * st, S->cD[ci]);
            acc(NV + 3 * ci + 1, rn - mu * rt, sn - mu * st, S->cD[ci]);
            acc(NV + 3 * ci + 2, rn, sn, T(2.0) * S->cD[ci]);
        }
    };

    // primal Newton with exact line search, run until the active set reproduces itself
    MBRL_HD void solve_constraints() {
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) { qacc[j] = qs[j]; fcon[j] = T(0.0); }
        newton_iters = 0;
        if (lim_on == 0 && con_on == 0 && pair_on == 0) return;
        uint64_t prev = ~uint64_t(0);
        bool converged = false;
        T arm[NV];
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) arm[j] = T(M::arma[j]);
        for (int it = 0; it <= kMaxNewton; it++) {
            Sp<T> Va[NB], Fc[NB];
            Sym3 Ic[NB];
            T flim[NV], hdiag[NV];
            vel_pass(qacc, Va);
            body_inertias(Ic);
            MBRL_UNROLL
            for (int b = 0; b < NB; b++) { Fc[b].w = T(0.0); Fc[b].a = T(0.0); Fc[b].b = T(0.0); }
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { flim[j] = T(0.0); hdiag[j] = arm[j]; }
            GradPass gp{this, Fc, Ic, flim, hdiag, 0};
            for_rows(Va, qacc, gp);
            // constraint force at the current point
            T jtf[NV];
            back_pass(Fc, jtf);
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) jtf[j] = jtf[j] + flim[j];
            if (gp.mask == prev || it == kMaxNewton) {
                converged = (gp.mask == prev);
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) fcon[j] = jtf[j];
                break;
            }
            prev = gp.mask;
            newton_iters = it + 1;
            // gradient  M qacc - qfrc_smooth - J^T f
            Sp<T> Fm[NB];
            MBRL_UNROLL
            for (int b = 0; b < NB; b++) Fm[b] = inert_mul(b, Va[b]);
            T Ma[NV], sx[NV];
            back_pass(Fm, Ma);
            T H[NV][NV];
            crba(Ic, H, hdiag);
            MBRL_UNROLL
            for (int k = 0; k < M::NPAIR; k++) {
                // a pair row couples two bodies: explicit D r r^T on the dofs between them (b1 is an ancestor of b2)
                if (gp.mask & (uint64_t(1) << (NV + 3 * NC + k))) {
                    const int b1 = M::g_body[M::pair_g1[k]], b2 = M::g_body[M::pair_g2[k]];
                    T rr[NV];
                    MBRL_UNROLL
                    for (int j = 0; j < NV; j++) {
                        if (!(TP::body_anc(M::dof_body[j], b2) && !TP::body_anc(M::dof_body[j], b1))) continue;
                        const Sp<T> Sj = Scol(j);
                        rr[j] = pw0[k] * Sj.w + pw1[k] * Sj.a + pw2[k] * Sj.b;
                    }
                    MBRL_UNROLL
                    for (int i = 0; i < NV; i++) {
                        if (!(TP::body_anc(M::dof_body[i], b2) && !TP::body_anc(M::dof_body[i], b1))) continue;
                        MBRL_UNROLL
                        for (int j = 0; j <= i; j++) {
                            if (!(TP::body_anc(M::dof_body[j], b2) && !TP::body_anc(M::dof_body[j], b1))) continue;
                            H[i][j] = H[i][j] + pD[k] * rr[i] * rr[j];
                        }
                    }
                }
            }
            const bool ok = factor(H);
            if (!ok) { status |= ST_NONFINITE; break; }
            T gnorm = T(0.0), a1 = T(0.0);
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) {
                Ma[j] = (Ma[j] + arm[j] * qacc[j]) - fsm[j];     // M qacc - qfrc_smooth
                sx[j] = -(Ma[j] - jtf[j]);
            }
            solve(H, sx);
            T snorm = T(0.0), qnorm = T(0.0);
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { snorm = snorm + sx[j] * sx[j]; qnorm = qnorm + qacc[j] * qacc[j]; a1 = a1 + sx[j] * Ma[j]; }
            (void)gnorm;
            if (!t_finite(snorm)) { status |= ST_NONFINITE; break; }
            if (t_val(t_sqrt(snorm)) <= 1e-15 * (1.0 + t_val(t_sqrt(qnorm)))) {
                converged = true;
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) fcon[j] = jtf[j];
                break;
            }
            // exact line search: root of the piecewise-linear derivative (safeguarded Newton)
            Sp<T> Vs[NB];
            vel_pass(sx, Vs);
            T a2 = T(0.0);
            MBRL_UNROLL
            for (int b = 0; b < NB; b++) {
                const Sp<T> Is = inert_mul(b, Vs[b]);
                a2 = a2 + (Vs[b].w * Is.w + Vs[b].a * Is.a + Vs[b].b * Is.b);
            }
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) a2 = a2 + arm[j] * sx[j] * sx[j];
            T alpha = T(1.0), lo = T(0.0), hi = T(-1.0);  // hi < 0: no upper bracket yet
            uint64_t lmask = gp.mask;                      // active set the current alpha was derived from
            for (int ls = 0; ls < kMaxLineSearch; ls++) {
                LinePass lp{this, Vs, sx, alpha, a1 + alpha * a2, a2, 0};
                for_rows(Va, qacc, lp);
                if (lp.mask == lmask) break;               // alpha is the exact root of its own piece
                lmask = lp.mask;
                if (t_val(lp.g) < 0.0) lo = alpha; else hi = alpha;
                T an = alpha - lp.g / lp.h;
                const bool inside = t_val(an) > t_val(lo) && (t_val(hi) < 0.0 || t_val(an) < t_val(hi));
                if (!inside) an = (t_val(hi) < 0.0) ? T(2.0) * alpha : T(0.5) * (lo + hi);
                alpha = an;
            }
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) qacc[j] = qacc[j] + alpha * sx[j];
        }
        if (!converged) status |= ST_NOCONVERGE;
    }

    // ------------------------------------------------------------------ mj_forward
    MBRL_HD void forward() {
        kinematics();
        make_constraints();
        smooth_forces();
        {
            Sym3 Ic[NB];
            T H[NV][NV], arm[NV];
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { arm[j] = T(M::arma[j]); qs[j] = fsm[j]; }
            body_inertias(Ic);
            crba(Ic, H, arm);
            if (!factor(H)) status |= ST_NONFINITE;
            solve(H, qs);
        }
        solve_constraints();
    }

    // ------------------------------------------------------------------ integrators
    MBRL_HD void euler() {
        T x[NV];
        if (M::ANY_DAMPING) {
            Sym3 Ic[NB];
            T H[NV][NV], dg[NV];
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { dg[j] = T(M::arma[j]) + T(M::DT) * M::damp[j]; x[j] = fsm[j] + fcon[j]; }
            body_inertias(Ic);
            crba(Ic, H, dg);
            if (!factor(H)) status |= ST_NONFINITE;
            solve(H, x);
        } else {
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) x[j] = qacc[j];
        }
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) { v[j] = v[j] + M::DT * x[j]; q[j] = q[j] + M::DT * v[j]; }
    }

    MBRL_HD void step() {
        forward();
        if (!M::RK4) { euler(); return; }
        // classic RK4; stage i uses only stage i-1 (tableau 1/2, 1/2, 1), weights 1/6 1/3 1/3 1/6
        T q0[NV], v0[NV], aq[NV], av[NV];
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            q0[j] = q[j]; v0[j] = v[j];
            aq[j] = T(1.0 / 6) * v[j]; av[j] = T(1.0 / 6) * qacc[j];
        }
        for (int i = 1; i < 4; i++) {
            const double ai = (i == 3) ? 1.0 : 0.5, bi = (i == 3) ? 1.0 / 6 : 1.0 / 3;
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) {
                const T dv = ai * v[j], da = ai * qacc[j];
                q[j] = q0[j] + M::DT * dv;
                v[j] = v0[j] + M::DT * da;
            }
            forward();
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { aq[j] = aq[j] + bi * v[j]; av[j] = av[j] + bi * qacc[j]; }
        }
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) { q[j] = q0[j] + M::DT * aq[j]; v[j] = v0[j] + M::DT * av[j]; }
    }

    // ------------------------------------------------------------------ env.step
    // `act` is the action handed to step() (== ctrl); returns the reward.
    MBRL_HD T env_step(int frame_skip) {
        const T xb = q[0];
        const double dt = M::DT * frame_skip;
        for (int k = 0; k < frame_skip; k++) step();
        T asq = T(0.0);
        MBRL_UNROLL
        for (int u = 0; u < NU; u++) asq = asq + ctrl[u] * ctrl[u];
        const double PI = 3.14159265358979323846;
        T r = T(0.0);
        if (M::REWARD == 0) {         // inverted_pendulum.py:26-35
            const T y = T(0.6) * t_cos(q[1]), x = y;
            r = -(T(0.01) * (x * x) + (y - T(1.0)) * (y - T(1.0))) - T(0.001) * (ctrl[0] * ctrl[0]);
        } else if (M::REWARD == 1) {  // inverted_double_pendulum.py:28-33 (stale site_xpos)
            const T x = site_a, y = site_b;
            const T pen = T(0.01) * (x * x) + (y - T(2.0)) * (y - T(2.0));
            const T vp = T(1e-3) * (v[1] * v[1]) + T(5e-3) * (v[2] * v[2]);
            r = -pen - vp;
        } else if (M::REWARD == 2) {  // swimmer.py:23-29
            r = (q[0] - xb) / dt + (T(-0.0001) * asq);
        } else if (M::REWARD == 3) {  // hopper.py:30-41
            r = (q[0] - xb) / dt;
            T aa = t_fmod(t_abs(q[2]), T(2 * PI));
            if (t_val(aa) > PI) aa = T(2 * PI) - aa;
            const T c1 = T(1.0) / (T(1.0) + t_pow(T(200.0), -(q[1] / 1.25) + T(0.5)));
            const T c2 = T(1.0) / (T(1.0) + t_pow(T(200.0), -((T(PI) - aa) / PI) + T(0.5)));
            r = r + (c1 * 1.0 + c2 * 1.0);
            r = r - T(1e-3) * asq;
        } else if (M::REWARD == 4) {  // walker2d.py:23-30
            r = (q[0] - xb) / dt;
            const T coeff = t_min(t_max(q[1] / 0.8, T(0.0)), T(1.0)) * 0.5 +
                            t_max((T(PI) - t_abs(q[2])) / PI, T(1.0)) * 0.5;
            r = r + coeff * 1.0;
            r = r - T(1e-3) * asq;
        } else {                      // half_cheetah.py:25-31
            r = T(-0.01) * asq + (q[0] - xb) / dt;
        }
        return r;
    }

    MBRL_HD void check_finite() {
        bool ok = true;
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) ok = ok && t_finite(q[j]) && t_finite(v[j]);
        if (!ok) status |= ST_NONFINITE;
    }
};

}  // namespace mbrl
