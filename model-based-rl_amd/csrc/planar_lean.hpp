// planar_lean.hpp -- per-lane planar rigid-body simulator of the HIP kernels (register-lean form).
//
// One instance = one rollout (one lane).  Templated on a compile-time model table
// (models_gen.hpp): after full unrolling every register-array index is a constant.  What it
// computes is the arithmetic the reference reaches through mujoco-py:
// MuJoCo 2.0 `mj_step` for planar hinge/slide trees --
//   kinematics -> recursive Newton-Euler bias -> passive (spring, damper, inertia-box fluid) ->
//   actuation -> joint-limit + plane/capsule pyramidal contact rows (+ capsule/capsule rows) ->
//   primal Newton solve to the exact active-set fixed point (opt.tolerance = 0,
//   mujoco/build.py:35) -> Euler (implicit joint damping) or RK4 -> env reward (mujoco/envs/*.py).
//
// MI355X-specific organisation (why this file exists next to the algorithm description above):
//   * the path is fp64-VALU bound and one lane owns one rollout, so the limiter is the per-lane
//     live state: 512 registers (256 VGPR + 256 AGPR at one wave per SIMD) = 256 doubles.
//     Everything that is touched rarely, or that needs a lane-dependent index, lives in a per-lane
//     LDS column instead (lane-interleaved, element i of lane t at word i*BLOCK + t: conflict
//     free for any per-lane index):  q, v, ctrl;  active joint-limit rows (D, aref);  and the
//     ACTIVE contact rows compacted into K slots of 5 doubles (point, D, aref_n, aref_t) -- a
//     capsule end that is not within its margin costs no storage at all;
//   * one fused top-down sweep per dynamics evaluation does kinematics, body velocities, the
//     Newton-Euler forward recursion, contact detection and limit detection body by body, so
//     poses and velocities die as soon as the subtree below them is done;
//   * the constraint Jacobian is never formed.  A contact row is a 3-vector `w` acting on the
//     spatial velocity of its body:  J x  is one tree pass,  J^T f  one backward pass, and the
//     Newton Hessian  M + J^T D J  is the composite-rigid-body algorithm run on inertias
//     augmented by  sum D w w^T ; the tree-sparse L^T D L factorisation has its sparsity
//     resolved at compile time.
//
// The scalar type T is `double` on the device; tests instantiate it on the host with `double`
// (comparison with the 3-D oracle) and with a counting type (algorithmic flop count for bench.py).
#pragma once
#include <cmath>
#include <cstdint>

#include "models_gen.hpp"
#include "wave_or.hpp"

#if defined(__HIPCC__)
#define MBRL_HD __host__ __device__ __forceinline__
#else
#define MBRL_HD inline
#endif
#define MBRL_UNROLL _Pragma("unroll")
#define MBRL_NOUNROLL _Pragma("nounroll")
// constraint rows are rare per candidate: keep the inactive path the fall-through one (a taken branch refills the
// instruction buffer, which nothing hides at one wave per SIMD)
#ifdef MBRL_NO_EXPECT
#define MBRL_RARE(c) (c)
#elif defined(MBRL_UNIFORM_RARE) && defined(__HIP_DEVICE_COMPILE__)
// A/B: wave-uniform outer test (scalar branch, exec untouched on the fall-through path), divergent test only inside
#define MBRL_RARE(c) (__builtin_expect(__builtin_amdgcn_ballot_w64(!!(c)) != 0, 0) && (c))
#else
#define MBRL_RARE(c) __builtin_expect(!!(c), 0)
#endif
// Compiler-only memory barrier at the top of the solver loops: without it the loop-invariant LDS loads
// (limit / contact rows) are hoisted out of the loops and held in registers, which defeats the point
// of keeping those rows in LDS.
#if defined(__HIP_DEVICE_COMPILE__)
#define MBRL_LOOP_FENCE asm volatile("" ::: "memory")
#define MBRL_OPAQUE(x) asm volatile("" : "+v"(x))
#define MBRL_OPAQUE_S(x) asm volatile("" : "+s"(x))     // the same for a wave-uniform value (stays in a scalar register)
#define MBRL_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)   // true in every active lane if any active lane has c
// Placed right after a row prefetch (Pop::next, which ends in a fence the loads cannot cross): everything computed from x
// is ordered AFTER the loads, so they are issued at the top of the block and their latency runs under the block's own
// arithmetic.  (Left to itself the scheduler sinks the loads to the bottom of the block, next to the copies into the
// loop-carried row registers, and the wait for them is fully exposed -- once per active row and pass.)
#ifdef MBRL_NO_ANCHOR
#define MBRL_ANCHOR(x)
#else
#define MBRL_ANCHOR(x) asm volatile("" : "+v"(x))
#endif
#else
#define MBRL_LOOP_FENCE
#define MBRL_OPAQUE(x)
#define MBRL_OPAQUE_S(x)
#define MBRL_WAVE_ANY(c) (c)
#define MBRL_ANCHOR(x)
#endif

// profiling builds only (tools/phase_timing.py): the kernel source defines MBRL_PHASE(k) to charge the shader clock
// since the previous mark to phase k
// -DMBRL_ASM_MARKS (ISA inspection only): named comment lines in the assembly at the phase boundaries
#if defined(MBRL_ASM_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define MBRL_MARK(name) asm volatile("; MBRL_MARK " name)
#else
#define MBRL_MARK(name)
#endif
#ifndef MBRL_PHASE
#define MBRL_PHASE(k)
#define MBRL_PHASE_MEMBER
#define MBRL_PIN(p, n)
#else   // make the n doubles at p materialise here, so that their arithmetic is charged to the phase that owns it
#define MBRL_PIN(p, n) do { _Pragma("unroll") for (int i_ = 0; i_ < (n); i_++) asm volatile("" : "+v"((p)[i_])); } while (0)
#endif

namespace mbrl {

constexpr double kMinVal = 1e-15;
constexpr unsigned ST_NONFINITE = 1u, ST_NOCONVERGE = 2u, ST_OVERFLOW = 4u, ST_DEGENERATE = 8u;
constexpr int kMaxNewton = 100;
constexpr int kMaxLineSearch = 40;
// The first Newton iterations take the full step without a line search (measured on on-trajectory HalfCheetah
// states, per wave and mj_step: 1.9 passes + 3.0 line evaluations with the search from the start, 2.7 + 0.2 with
// two full steps first).  The minimum is unique and every exit tests the active set, so the path does not matter.
// How many: 4 for the three models with floor contacts -- against 2, the perturbed rollouts of Hopper and Walker2d run
// 2-3 % faster and HalfCheetah's the same, and the nominal / step launches (64 unrelated units per wave, where every
// line search any lane needs is paid by all) 5-10 % faster: 0.6 ms of a 12.4 ms training epoch (1 is slower everywhere,
// 6 and 8 no better; profiles/r02_experiments.md).  2 for the models that only have joint limits.
// -DMBRL_UNIT_ITERS=n overrides it for every model (A/B builds).
#ifdef MBRL_UNIT_ITERS
template <class M> constexpr int kUnitIters = MBRL_UNIT_ITERS;
#else
template <class M> constexpr int kUnitIters = M::HAS_FLOOR ? 4 : 2;
#endif

// ---- scalar helpers (overloaded by the host-side counting type) ----
MBRL_HD double t_abs(double x) { return fabs(x); }
MBRL_HD double t_max(double x, double y) { return fmax(x, y); }
MBRL_HD double t_min(double x, double y) { return fmin(x, y); }
MBRL_HD double t_sqrt(double x) { return sqrt(x); }
MBRL_HD double t_pow(double x, double y) { return pow(x, y); }
MBRL_HD double t_fmod(double x, double y) { return fmod(x, y); }
MBRL_HD double t_cos(double x) { return cos(x); }
MBRL_HD bool t_finite(double x) {
    // Exponent test on the high word, made opaque as an INTEGER: under no-NaN / no-infinity compilation the
    // optimiser recognises the plain bit test as a floating-point class test and folds it to true.
    uint32_t hi = uint32_t(__builtin_bit_cast(uint64_t, x) >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(hi));
#endif
    return (hi & 0x7ff00000u) != 0x7ff00000u;
}
MBRL_HD double t_val(double x) { return x; }
MBRL_HD double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MBRL_HD double t_rint(double x) { return __builtin_rint(x); }
MBRL_HD double t_rcp(double x) {   // reciprocal to ~1 ulp (pivot inverses, regularisers)
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// sin and cos of a joint angle: two-term Cody-Waite reduction by pi/2 and the fdlibm kernel
// polynomials, branch-free (|x| < ~1e5 keeps the reduction exact to an ulp; joint angles are O(1)).
template <class T>
MBRL_HD void t_sincos(T x, T *s, T *c) {
    const T k = t_rint(x * T(6.36619772367581382433e-01));
    T r = t_fma(-k, T(1.5707963267948966), x);
    r = t_fma(-k, T(6.123233995736766e-17), r);
    const T z = r * r;
    T ps = t_fma(z, T(1.58969099521155010221e-10), T(-2.50507602534068634195e-08));
    ps = t_fma(z, ps, T(2.75573137070700676789e-06));
    ps = t_fma(z, ps, T(-1.98412698298579493134e-04));
    ps = t_fma(z, ps, T(8.33333333332248946124e-03));
    ps = t_fma(z, ps, T(-1.66666666666666324348e-01));
    const T sr = t_fma(r * z, ps, r);
    T pc = t_fma(z, T(-1.13596475577881948265e-11), T(2.08757232129817482790e-09));
    pc = t_fma(z, pc, T(-2.75573143513906633035e-07));
    pc = t_fma(z, pc, T(2.48015872894767294178e-05));
    pc = t_fma(z, pc, T(-1.38888888888741095749e-03));
    pc = t_fma(z, pc, T(4.16666666666666019037e-02));
    const T cr = t_fma(z * z, pc, t_fma(z, T(-0.5), T(1.0)));
    const int q = int(t_val(k)) & 3;
    const T ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// sin and cos of N angles at once, coefficient by coefficient: the twelve polynomial constants are then
// materialised once per batch instead of once per angle (they are scalar-register literals on the device).
#define MBRL_SINCOS_TABLE                                                                                        \
    {1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04, \
     8.33333333332248946124e-03, -1.66666666666666324348e-01, -1.13596475577881948265e-11, 2.08757232129817482790e-09, \
     -2.75573143513906633035e-07, 2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MBRL_LITERAL_SINCOS)
// On the device the twelve coefficients come from constant memory: two scalar loads per sweep put them in scalar
// registers, where every Horner step reads them as an operand; as literals each use cost two vector moves (the fmac
// form wants the addend in its destination).  Not `const`, so the optimiser cannot fold them back into literals.
// (The same for the whole model-constant table was measured and rejected: the loads are hoisted to the top of the
// sweep and the scalar registers spill into vector lanes.)
__constant__ double mbrl_sincos_tab[12] = MBRL_SINCOS_TABLE;
#define MBRL_SC(i) mbrl_sincos_tab[i]
#else
constexpr double mbrl_sincos_tab_host[12] = MBRL_SINCOS_TABLE;
#define MBRL_SC(i) mbrl_sincos_tab_host[i]
#endif
template <class T, int N>
MBRL_HD void t_sincos_batch(const T *x, T *s, T *c) {
    static_assert(N >= 1, "");
    const double SC[6] = {MBRL_SC(0), MBRL_SC(1), MBRL_SC(2), MBRL_SC(3), MBRL_SC(4), MBRL_SC(5)};
    const double CC[6] = {MBRL_SC(6), MBRL_SC(7), MBRL_SC(8), MBRL_SC(9), MBRL_SC(10), MBRL_SC(11)};
    T k[N], r[N], z[N], ps[N], pc[N];
    MBRL_UNROLL
    for (int i = 0; i < N; i++) k[i] = t_rint(x[i] * T(6.36619772367581382433e-01));
    MBRL_UNROLL
    for (int i = 0; i < N; i++) r[i] = t_fma(-k[i], T(1.5707963267948966), x[i]);
    MBRL_UNROLL
    for (int i = 0; i < N; i++) { r[i] = t_fma(-k[i], T(6.123233995736766e-17), r[i]); z[i] = r[i] * r[i]; }
    MBRL_UNROLL
    for (int i = 0; i < N; i++) ps[i] = t_fma(z[i], T(SC[0]), T(SC[1]));
    MBRL_UNROLL
    for (int m = 2; m < 6; m++) {
        MBRL_UNROLL
        for (int i = 0; i < N; i++) ps[i] = t_fma(z[i], ps[i], T(SC[m]));
    }
    MBRL_UNROLL
    for (int i = 0; i < N; i++) pc[i] = t_fma(z[i], T(CC[0]), T(CC[1]));
    MBRL_UNROLL
    for (int m = 2; m < 6; m++) {
        MBRL_UNROLL
        for (int i = 0; i < N; i++) pc[i] = t_fma(z[i], pc[i], T(CC[m]));
    }
    MBRL_UNROLL
    for (int i = 0; i < N; i++) {
        const T sr = t_fma(r[i] * z[i], ps[i], r[i]);
        const T cr = t_fma(z[i] * z[i], pc[i], t_fma(z[i], T(-0.5), T(1.0)));
        const int q = int(t_val(k[i])) & 3;
        const T ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
        s[i] = (q & 2) ? -ss : ss;
        c[i] = ((q + 1) & 2) ? -cc : cc;
    }
}

// (cc*x - ss*y, ss*x + cc*y) for a compile-time constant point (x, y); zero components cost nothing
// (x * 0.0 is not folded by the compiler: it is not an identity for NaN / infinity)
template <class T>
MBRL_HD void t_rot(T cc, T ss, double x, double y, T *oa, T *ob) {
    if (x == 0.0 && y == 0.0) { *oa = T(0.0); *ob = T(0.0); }
    else if (y == 0.0) { *oa = cc * x; *ob = ss * x; }
    else if (x == 0.0) { *oa = -(ss * y); *ob = cc * y; }
    else { *oa = cc * x - ss * y; *ob = ss * x + cc * y; }
}

template <class T>
struct Sp {  // planar spatial vector: angular part w, linear part (a, b)
    T w, a, b;
};

// per-lane memory column: element i of this lane at p[i * STRIDE]
template <class T, int STRIDE>
struct LaneMem {
    T *p;
    MBRL_HD T ld(int i) const { return p[i * STRIDE]; }
    MBRL_HD void st(int i, T x) const { p[i * STRIDE] = x; }
};

// Topology predicates of a model, tabulated at compile time.  (They are looked up with unrolled loop
// variables; a constexpr FUNCTION called that way is ordinary run-time code to the optimiser -- the
// ancestor walk became scalar loads and loops inside the factorisation -- a constant TABLE folds.)
template <class M>
struct Topo {
    static constexpr int NB = M::NB, NV = M::NV, NG = M::NGEOM, NPR = M::NPAIR > 0 ? M::NPAIR : 1;
    static constexpr bool body_anc(int anc, int b) {
        while (b >= 0) {
            if (b == anc) return true;
            b = M::parent[b];
        }
        return false;
    }
    struct Tables {
        bool dof_anc[NV][NV];     // [j][i]: dof j moves the body of dof i (j <= i in index order)
        int lim_rank[NV];         // index of dof j among the limited dofs
        int hinge_rank[NV];       // index of dof j among the hinge dofs
        int hinge_dof[NV];        // its inverse
        int nhinge;
        int cand_rank[NG][2];     // index of capsule end (g, e) among the floor-contact candidates
        bool any_child[NB];
        bool first_folded[NB];    // b is the highest-index child of its parent (first one a leaf-to-root pass folds)
        bool between[NPR][NV];    // pair k: dof j moves body 2 but not body 1 of the pair
        bool moves[NV][NB];       // dof j moves body b
        int nlim, ncand;
    };
    static constexpr Tables make() {
        Tables t{};
        for (int j = 0; j < NV; j++)
            for (int i = 0; i < NV; i++) t.dof_anc[j][i] = j <= i && body_anc(M::dof_body[j], M::dof_body[i]);
        int n = 0;
        for (int j = 0; j < NV; j++) { t.lim_rank[j] = n; n += M::limited[j] ? 1 : 0; }
        t.nlim = n;
        n = 0;
        for (int j = 0; j < NV; j++) {
            t.hinge_rank[j] = n; t.hinge_dof[j] = 0;
            if (M::dof_type[j] == 2) { t.hinge_dof[n] = j; n++; }
        }
        t.nhinge = n;
        n = 0;
        for (int g = 0; g < NG; g++) {
            t.cand_rank[g][0] = n; t.cand_rank[g][1] = n + 1;
            n += (M::HAS_FLOOR && M::c_on[g]) ? 2 : 0;
        }
        t.ncand = n;
        for (int b = 0; b < NB; b++) {
            t.any_child[b] = false; t.first_folded[b] = true;
            for (int o = 0; o < NB; o++) {
                if (M::parent[o] == b) t.any_child[b] = true;
                if (o > b && M::parent[o] == M::parent[b]) t.first_folded[b] = false;
            }
        }
        for (int j = 0; j < NV; j++)
            for (int b = 0; b < NB; b++) t.moves[j][b] = body_anc(M::dof_body[j], b);
        for (int k = 0; k < M::NPAIR; k++) {
            const int b1 = M::g_body[M::pair_g1[k]], b2 = M::g_body[M::pair_g2[k]];
            for (int j = 0; j < NV; j++) t.between[k][j] = body_anc(M::dof_body[j], b2) && !body_anc(M::dof_body[j], b1);
        }
        return t;
    }
    static constexpr Tables T = make();
};

// layout of the per-lane memory column; KCAP = most floor contacts held at once (the fast kernels keep 8 slots, the
// fix-up kernel of mbrl_hip.hip one per candidate)
template <class M, int KCAP = 8>
struct Layout {
    using TP = Topo<M>;
    static constexpr int NV = M::NV, NU = M::NU;
    static constexpr int NLIM = TP::T.nlim, NCAND = TP::T.ncand;
    static constexpr int K = NCAND < KCAP ? NCAND : KCAP;      // contact slots (active rows beyond K: ST_OVERFLOW)
    // contact slots first: their (lane-dependent) addresses are then  base + slot * stride + small immediate
    static constexpr int OS = 0, OP = 5 * K, OL = OP + 5 * M::NPAIR, OQ = OL + 2 * NLIM, OV = OQ + NV, OC = OV + NV;
    static constexpr int NMEM = OC + NU;
};

template <bool WIDE> struct RowMask { using type = uint32_t; };
template <> struct RowMask<true> { using type = uint64_t; };

template <class M, class T, class Mem, int KCAP = 8>
struct Lean {
    static constexpr int NB = M::NB, NV = M::NV, NU = M::NU;
    using TP = Topo<M>;
    using LY = Layout<M, KCAP>;
    static constexpr int K = LY::K;
    using CM = typename RowMask<(3 * K > 32)>::type;      // contact-row mask: 3 bits per slot
    static_assert(3 * K <= 64 && NV + M::NPAIR <= 32 && LY::NCAND <= 32, "row masks are 32-bit (contacts: up to 64)");

    Mem mem;
    // ---- position-dependent, alive through a dynamics evaluation ----
    T ra[NV], rb[NV];                    // hinge lever r = O - anchor (O = origin of the root body)
    T ha[NB], hb[NB], IO[NB];            // first moment m*(com - O) and inertia about O
    T fsm[NV], qacc[NV], fcon[NV];
    T site_a, site_b;
    // Joint-space inertia  M + armature  (lower triangle, tree-sparse) of the CURRENT dynamics evaluation: built once by
    // the composite-rigid-body pass and then shared by everything that needs it -- the unconstrained solve, every Newton
    // Hessian (HM + explicit J^T D J of the few active rows), (M + arm) s of the line search, and Euler's implicit-damping
    // factorisation (HM + h diag(damping)).  Before, each of those ran its own leaf-to-root pass over the body inertias.
    T HM[NV][NV];
#ifdef MBRL_NO_HM
    static constexpr bool kHM = false;      // A/B: round-2 formulation (inertias augmented inside a tree pass per Newton iteration)
#else
    static constexpr bool kHM = true;
#endif
    // Active rows sit in two LDS stacks in detection order (ascending dof / ascending capsule end):
    // limit rows (D, aref) and contact slots (point, D, aref_n, aref_t).  The solver passes walk the
    // constraints in DESCENDING order, so they pop the stacks top-down and always have the rows of the
    // next active constraint in flight while they work on the current one (LDS latency is otherwise
    // fully exposed at one wave per SIMD).
    // Rows that are active in ANY lane of this wave (wave-uniform; -DMBRL_NO_WAVE_ROWS turns it off): cand_on | lim_on << NCAND |
    // pair_on << (NCAND + NV), OR-ed over the wave once per dynamics evaluation.  The solver passes test it with scalar
    // branches before they test their own lane's bit: a row nobody has then costs a scalar compare and branch instead of
    // a vector compare feeding s_and_saveexec (11 against ~50 clocks for a lone wave, tools/micro/branch_cost2.hip).
    uint32_t wave_rows;
#if defined(MBRL_NO_WAVE_ROWS)
    static constexpr bool kWaveRows = false;    // A/B: every row test is the lane's own (round 3 before this change)
#else
    // Models with floor contacts only: with the one or two limit rows of the pendulums and the swimmer there is nothing
    // to skip, and the LDS round trip costs them 2-3 % (profiles/r03_sweep.md before / after).
    static constexpr bool kWaveRows = M::HAS_FLOOR;
#endif
    static_assert(Layout<M, KCAP>::NCAND + M::NV + M::NPAIR <= 32, "the wave-wide row mask is one 32-bit word");
    // (each test reads its own opaque copy: left to itself the compiler evaluates all of them ahead of the solver loops
    // and keeps thirty 64-bit booleans alive in scalar registers, which then spill into VGPR lanes)
    static constexpr uint32_t low_bits(int n) { return n >= 32 ? ~0u : (n <= 0 ? 0u : (1u << n) - 1u); }
    MBRL_HD bool w_test(uint32_t bits) const {
        if (!kWaveRows) return true;
        uint32_t w = wave_rows;
        MBRL_OPAQUE_S(w);
        return (w & bits) != 0;
    }
    MBRL_HD bool w_cand(uint32_t bits) const { return w_test(bits); }
    MBRL_HD bool w_lim(uint32_t bits) const { return w_test(bits << Layout<M, KCAP>::NCAND); }
    MBRL_HD bool w_pair(uint32_t bits) const { return w_test(bits << (Layout<M, KCAP>::NCAND + M::NV)); }
    uint32_t lim_on, lim_hi;             // bit j: limit row of dof j active / it is the upper limit
    uint32_t cand_on;                    // bit c: floor candidate (capsule end) c is in contact and owns a slot
    int nslot, nlimrow;                  // heights of the two stacks
    bool was_constrained;                // the last dynamics evaluation had active rows (its qacc came from the solver)
    uint32_t pair_on;
    unsigned status;
    int newton_iters;
    // Control-flow signature of the rollout (the launch planner sorts the units by it, mbrl_fd_plan; units with equal
    // signatures take the same branches).  Accumulated per mj_step by key_note, assembled by key_value:
    //   rollouts of up to 10 mj_steps:  [floor contacts that were EVER active][3 bits of Hessian passes per step, oldest
    //     step most significant][limits / pairs that were ever active];
    //   longer ones (Hopper 20, Walker2d 25 steps):  [floor contacts ever active][passes of the whole rollout / 4 : 6 bits]
    //     [2 bits of passes per step, first KEY_REC steps][limits / pairs ever active].
    // While it accumulates, `key` holds the per-step pass counts and key2 the row union (bits 0..23; long rollouts keep
    // the pass total in bits 24..31).  Round 2 chose pass counts first from a dozen orders (tools/key_probe.py,
    // profiles/r02_experiments.md; -DMBRL_KEY_ROUND2 restores it); with the wave-wide row mask, units that share their
    // contacts are worth more together: ten layouts re-measured in profiles/r03_experiments.md.
    uint64_t key;
    uint32_t key2;
    int key_left;                        // long rollouts: steps still to record; -1: short rollout
    uint32_t ls_iters;
    MBRL_PHASE_MEMBER

    // ------------------------------------------------------------------ motion subspace
    MBRL_HD Sp<T> Scol(int j) const {
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

    // spatial velocity of ONE body for generalized velocity x (same summation order as vel_pass).  The solver
    // passes need body velocities only inside the rare active-contact blocks, so they take them on demand
    // instead of keeping 3 NB doubles alive through the Newton iteration.
    MBRL_HD Sp<T> body_vel(int b, const T *x) const {
        Sp<T> u;
        u.w = T(0.0); u.a = T(0.0); u.b = T(0.0);
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            if (!TP::T.moves[j][b]) continue;
            if (M::dof_type[j] == 0) u.a = u.a + x[j];
            else if (M::dof_type[j] == 1) u.b = u.b + x[j];
            else {
                const T xs = M::dof_sgn[j] * x[j];
                u.w = u.w + xs;
                u.a = u.a - rb[j] * xs;
                u.b = u.b + ra[j] * xs;
            }
        }
        return u;
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

    // ------------------------------------------------------------------ impedance / regulariser
    // returns D = 1 / R,  R = max(minval, (1 - imp) * RK / imp), and the impedance itself
    MBRL_HD static T impedance(double dmin, double dmax, double width, double mid, double power, T pm) {
        if (dmin == dmax || width <= kMinVal) return T(0.5 * (dmin + dmax));
        const T x = t_abs(pm) * (1.0 / width);
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

    // ------------------------------------------------------------------ fused top-down sweep
    // kinematics + body velocities + Newton-Euler forward recursion + constraint detection.
    // Leaves: ra, rb, ha, hb, IO, site, constraint rows in `mem`, and the per-body net forces F.
    MBRL_HD void sweep(Sp<T> *F) {
        lim_on = 0; lim_hi = 0; cand_on = 0; pair_on = 0;
        nslot = 0; nlimrow = 0;
        // all of q, v in one batch of LDS reads, ahead of the row stores (which may alias them for the compiler)
        T qall[NV], vall[NV];
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) { qall[j] = mem.ld(LY::OQ + j); vall[j] = mem.ld(LY::OV + j); }
        constexpr int NH = TP::T.nhinge > 0 ? TP::T.nhinge : 1;
        T hs[NH], hc[NH];          // sin / cos of every hinge angle, evaluated as one batch
        {
            T ang[NH];
            MBRL_UNROLL
            for (int i = 0; i < TP::T.nhinge; i++) {
                const int j = TP::T.hinge_dof[i];
                ang[i] = M::dof_sgn[j] * (qall[j] - M::qpos0[j]);
            }
            if (TP::T.nhinge > 0) t_sincos_batch<T, NH>(ang, hs, hc);
        }
        T pa[NB], pb[NB], c[NB], s[NB];
        Sp<T> V[NB], A[NB];
        T Ob = T(0.0), Oa = T(0.0);
        MBRL_UNROLL
        for (int b = 0; b < NB; b++) {
            // ---- pose ----
            T px, py, cc, ss;
            if (M::parent[b] < 0) {
                px = T(M::bpos[b][0]); py = T(M::bpos[b][1]); cc = T(1.0); ss = T(0.0);
            } else {
                const int p = M::parent[b];
                T da, db;
                t_rot(c[p], s[p], M::bpos[b][0], M::bpos[b][1], &da, &db);
                px = pa[p] + da; py = pb[p] + db;
                cc = c[p]; ss = s[p];
            }
            T qj[3], vj[3], anca[3], ancb[3];
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                qj[k] = qall[j]; vj[k] = vall[j];
                const T dq = qj[k] - M::qpos0[j];
                if (M::dof_type[j] == 0) px = px + dq;
                else if (M::dof_type[j] == 1) py = py + dq;
                else {
                    T ja, jb;
                    t_rot(cc, ss, M::jpos[j][0], M::jpos[j][1], &ja, &jb);
                    const T ax = px + ja, ay = py + jb;
                    const T sj = hs[TP::T.hinge_rank[j]], cj = hc[TP::T.hinge_rank[j]];
                    const T cn = cc * cj - ss * sj, sn = ss * cj + cc * sj;
                    cc = cn; ss = sn;
                    t_rot(cc, ss, M::jpos[j][0], M::jpos[j][1], &ja, &jb);
                    px = ax - ja; py = ay - jb;
                    anca[k] = ax; ancb[k] = ay;
                }
            }
            if (M::parent[b] < 0) {  // root: its final origin is O; everything else is relative to it
                Oa = px; Ob = py;
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
            T cwa, cwb;
            t_rot(cc, ss, M::com[b][0], M::com[b][1], &cwa, &cwb);
            cwa = pa[b] + cwa; cwb = pb[b] + cwb;
            ha[b] = M::mass[b] * cwa; hb[b] = M::mass[b] * cwb;
            IO[b] = M::izz[b] + M::mass[b] * (cwa * cwa + cwb * cwb);
            if (M::HAS_SITE && b == M::site_body) {
                site_a = Oa + pa[b] + (cc * M::site_pos[0] - ss * M::site_pos[1]);
                site_b = Ob + pb[b] + (ss * M::site_pos[0] + cc * M::site_pos[1]);
            }
            // ---- velocity and bias acceleration (Newton-Euler forward recursion) ----
            Sp<T> u, a;
            if (M::parent[b] >= 0) { u = V[M::parent[b]]; a = A[M::parent[b]]; }
            else {
                u.w = T(0.0); u.a = T(0.0); u.b = T(0.0);
                a.w = T(0.0); a.a = T(-M::GRAV_A); a.b = T(-M::GRAV_B);
            }
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                if (M::dof_type[j] == 2) {
                    const Sp<T> S = Scol(j);
                    // Sdot = crm(u) S = (0, -u.w*S.b + u.b*S.w, u.w*S.a - u.a*S.w)
                    a.a = a.a + (u.b * S.w - u.w * S.b) * vj[k];
                    a.b = a.b + (u.w * S.a - u.a * S.w) * vj[k];
                    u.w = u.w + S.w * vj[k]; u.a = u.a + S.a * vj[k]; u.b = u.b + S.b * vj[k];
                } else if (M::dof_type[j] == 0) {
                    a.b = a.b + u.w * vj[k];
                    u.a = u.a + vj[k];
                } else {
                    a.a = a.a - u.w * vj[k];
                    u.b = u.b + vj[k];
                }
            }
            V[b] = u; A[b] = a;
            {
                const Sp<T> Ia = inert_mul(b, a), Iv = inert_mul(b, u);
                // crf(u) Iv = (u.a*Iv.b - u.b*Iv.a, -u.w*Iv.b, u.w*Iv.a)
                F[b].w = Ia.w + (u.a * Iv.b - u.b * Iv.a);
                F[b].a = Ia.a - u.w * Iv.b;
                F[b].b = Ia.b + u.w * Iv.a;
            }
            if (M::FLUID) {
                // inertia-box fluid model: velocity at the body com in body axes
                const T rca = cwa, rcb = cwb;                                   // com - O
                const T vca = u.a - u.w * rcb, vcb = u.b + u.w * rca;
                const T la = cc * vca + ss * vcb, lb = -ss * vca + cc * vcb;
                const T fla = -M::fl_kv_lin[b] * la - M::fl_kq_a[b] * t_abs(la) * la;
                const T flb = -M::fl_kv_lin[b] * lb - M::fl_kq_b[b] * t_abs(lb) * lb;
                const T tq = -M::fl_kv_ang[b] * u.w - M::fl_kq_ang[b] * t_abs(u.w) * u.w;
                const T fa = cc * fla - ss * flb, fb = ss * fla + cc * flb;
                F[b].w = F[b].w - (tq + (rca * fb - rcb * fa));
                F[b].a = F[b].a - fa;
                F[b].b = F[b].b - fb;
            }
            // ---- joint limits of this body's dofs ----
            MBRL_UNROLL
            for (int k = 0; k < M::dofnum[b]; k++) {
                const int j = M::dofadr[b] + k;
                if (!M::limited[j]) continue;
                const T dlo = qj[k] - M::range[j][0], dhi = M::range[j][1] - qj[k];
                const bool lo = t_val(dlo) < M::lim_margin[j], hi = t_val(dhi) < M::lim_margin[j];
                if (MBRL_RARE(lo || hi)) {
                    const T dist = lo ? dlo : dhi;
                    const T sg = lo ? T(1.0) : T(-1.0);
                    const T imp = impedance(M::lim_solimp[j][0], M::lim_solimp[j][1], M::lim_solimp[j][2],
                                            M::lim_solimp[j][3], M::lim_solimp[j][4], dist - M::lim_margin[j]);
                    const T R = t_max(T(kMinVal), (T(1.0) - imp) * M::dof_invweight0[j] / imp);
                    const int lo2 = LY::OL + 2 * nlimrow;
                    mem.st(lo2, t_rcp(R));
                    mem.st(lo2 + 1, -M::lim_B[j] * (sg * vj[k]) - M::lim_K[j] * imp * (dist - M::lim_margin[j]));
                    nlimrow++;
                    lim_on |= (1u << j);
                    if (!lo) lim_hi |= (1u << j);
                }
            }
            // ---- capsule-end / floor contacts of this body's geoms ----
            if (M::HAS_FLOOR) {
#ifndef MBRL_NO_FAST_FLOOR_TEST
                // Broad phase, two fused multiply-adds per capsule end and ONE test per capsule: the height of the end's
                // sphere centre above the floor against radius + margin.  A capsule whose ends are both clear (the usual
                // case) costs nothing else; the exact distance -- MuJoCo's expression, with its own rounding -- decides
                // inside.  The broad phase is widened by 1e-9 so that it can never reject what the exact test accepts.
                const T zb = (Ob + pb[b]) - M::FLOOR_Z;
#endif
                MBRL_UNROLL
                for (int g = 0; g < M::NGEOM; g++) {
                    if (M::g_body[g] != b || !M::c_on[g]) continue;
#ifndef MBRL_NO_FAST_FLOOR_TEST
                    const T h0 = t_fma(cc, T(M::g_ea[g][1]), t_fma(ss, T(M::g_ea[g][0]), zb));
                    const T h1 = t_fma(cc, T(M::g_eb[g][1]), t_fma(ss, T(M::g_eb[g][0]), zb));
                    if (!MBRL_RARE(t_val(t_min(h0, h1)) < M::g_rad[g] + M::c_margin[g] + 1e-9)) continue;
#endif
                    MBRL_UNROLL
                    for (int e = 0; e < 2; e++) {
                        const double la = e ? M::g_eb[g][0] : M::g_ea[g][0], lb = e ? M::g_eb[g][1] : M::g_ea[g][1];
                        T ea, eb;                                   // sphere centre rel. O
                        t_rot(cc, ss, la, lb, &ea, &eb);
                        ea = pa[b] + ea; eb = pb[b] + eb;
                        const T dist = ((Ob + eb) - M::FLOOR_Z) - M::g_rad[g];
                        if (MBRL_RARE(t_val(dist) < M::c_margin[g])) {
                            if (nslot >= K) { status |= ST_OVERFLOW; continue; }
                            const T Pa = ea, Pb = eb - (M::g_rad[g] + T(0.5) * dist);
                            const T pm = dist - M::c_margin[g];
                            const T imp = impedance(M::c_solimp[g][0], M::c_solimp[g][1], M::c_solimp[g][2],
                                                    M::c_solimp[g][3], M::c_solimp[g][4], pm);
                            const T R = t_max(T(kMinVal), (T(1.0) - imp) * M::c_RK[g] / imp);
                            const T vn = u.b + u.w * Pa, vt = u.a - u.w * Pb;
                            const int o = LY::OS + 5 * nslot;
                            mem.st(o, Pa); mem.st(o + 1, Pb); mem.st(o + 2, t_rcp(R));
                            mem.st(o + 3, -M::c_B[g] * vn - M::c_K[g] * imp * pm);
                            mem.st(o + 4, -M::c_B[g] * vt);
                            nslot++;
                            cand_on |= 1u << TP::T.cand_rank[g][e];
                        }
                    }
                }
            }
        }
        // ---- capsule-capsule pairs (closest points of the two axis segments, then sphere-sphere) ----
        MBRL_UNROLL
        for (int k = 0; k < M::NPAIR; k++) {
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
            // Crossing axes: the closest points coincide and their difference is round-off noise.  (In a plane this is
            // no rare event: two segments cross over a whole range of poses, e.g. a Hopper that has folded through
            // itself.)  Rule, MuJoCo's sphere-sphere one [from knowledge, mjraw_SphereSphere]: the normal is the cross
            // product of the two capsule axes, or (1,0,0) when those are parallel.  For a planar model that cross product
            // points OUT of the plane, so the row acts on no dof and exerts no force: the pair gets no row at all.
            // The rule applies below 1e-10 (MuJoCo: 1e-15, and a normalised noise vector in between -- not reproducible
            // by any second implementation), identically in the oracle (sphereSphere in oracle/mjc_oracle.c).
            bool inert = false;
            if (t_val(len) < 1e-10) {
                const T cr = d1a * d2b - d1b * d2a;
                inert = t_val(cr) * t_val(cr) >= (kMinVal * kMinVal) * (t_val(aa) * t_val(ee));
            }
            if (t_val(dist) < M::pair_margin[k] && !inert) {
                T na = T(1.0), nb = T(0.0);
                if (!(t_val(len) < 1e-10)) { na = dfa / len; nb = dfb / len; }
                const T Pa = c1a + na * (M::g_rad[g1] + T(0.5) * dist);
                const T Pb = c1b + nb * (M::g_rad[g1] + T(0.5) * dist);
                const T pm = dist - M::pair_margin[k];
                const T imp = impedance(M::pair_solimp[k][0], M::pair_solimp[k][1], M::pair_solimp[k][2],
                                        M::pair_solimp[k][3], M::pair_solimp[k][4], pm);
                const T R = t_max(T(kMinVal), (T(1.0) - imp) * M::pair_RK[k] / imp);
                const T w0 = nb * Pa - na * Pb;
                const T vel = w0 * (V[b2].w - V[b1].w) + na * (V[b2].a - V[b1].a) + nb * (V[b2].b - V[b1].b);
                const int o = LY::OP + 5 * k;
                mem.st(o, w0); mem.st(o + 1, na); mem.st(o + 2, nb); mem.st(o + 3, t_rcp(R));
                mem.st(o + 4, -M::pair_B[k] * vel - M::pair_K[k] * imp * pm);
                pair_on |= (1u << k);
            }
        }
    }

    // qfrc_smooth = passive - bias + actuation
    MBRL_HD void smooth_forces(Sp<T> *F) {
        T bias[NV];
        back_pass(F, bias);
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            T f = -bias[j];
            if (M::stiff[j] != 0.0) f = f - M::stiff[j] * mem.ld(LY::OQ + j);
            if (M::damp[j] != 0.0) f = f - M::damp[j] * mem.ld(LY::OV + j);
            fsm[j] = f;
        }
        MBRL_UNROLL
        for (int u = 0; u < NU; u++) {
            T cu = mem.ld(LY::OC + u);
            if (M::ctrllimited[u]) cu = t_min(T(M::ctrlrange[u][1]), t_max(T(M::ctrlrange[u][0]), cu));
            fsm[M::act_dof[u]] = fsm[M::act_dof[u]] + M::gear[u] * cu;
        }
    }

    // ------------------------------------------------------------------ CRBA on 3x3 symmetric inertias
    struct Sym3 { T ww, wa, wb, aa, ab, bb; };

    // H (lower triangle, tree-sparse) = sum_b A_b^T I_b A_b + diag(add).  Bodies are folded leaf-to-root
    // and a body's rows are emitted as soon as its subtree is complete, so at most one composite per
    // tree level is alive.
    MBRL_HD void crba(T (*H)[NV], const T *diag_add) const {
        Sym3 Ic[NB];
        MBRL_UNROLL
        for (int b = NB - 1; b >= 0; b--) {
            Sym3 I;
            I.ww = IO[b]; I.wa = -hb[b]; I.wb = ha[b]; I.aa = T(M::mass[b]); I.ab = T(0.0); I.bb = T(M::mass[b]);
            if (TP::T.any_child[b]) {   // children were folded into Ic[b] already
                I.ww = I.ww + Ic[b].ww; I.wa = I.wa + Ic[b].wa; I.wb = I.wb + Ic[b].wb;
                I.aa = I.aa + Ic[b].aa; I.ab = I.ab + Ic[b].ab; I.bb = I.bb + Ic[b].bb;
            }
            MBRL_UNROLL
            for (int k = M::dofnum[b] - 1; k >= 0; k--) {
                const int i = M::dofadr[b] + k;
                const Sp<T> S = Scol(i);
                Sp<T> Fi;
                if (M::dof_type[i] == 0) { Fi.w = I.wa; Fi.a = I.aa; Fi.b = I.ab; }
                else if (M::dof_type[i] == 1) { Fi.w = I.wb; Fi.a = I.ab; Fi.b = I.bb; }
                else {
                    Fi.w = I.ww * S.w + I.wa * S.a + I.wb * S.b;
                    Fi.a = I.wa * S.w + I.aa * S.a + I.ab * S.b;
                    Fi.b = I.wb * S.w + I.ab * S.a + I.bb * S.b;
                }
                MBRL_UNROLL
                for (int j = 0; j <= i; j++) {
                    if (!TP::T.dof_anc[j][i]) continue;
                    T h;
                    if (M::dof_type[j] == 0) h = Fi.a;
                    else if (M::dof_type[j] == 1) h = Fi.b;
                    else { const Sp<T> Sj = Scol(j); h = Sj.w * Fi.w + Sj.a * Fi.a + Sj.b * Fi.b; }
                    if (j == i) h = h + diag_add[i];
                    H[i][j] = h;
                }
            }
            if (M::parent[b] >= 0) {
                const int p = M::parent[b];
                if (TP::T.first_folded[b]) Ic[p] = I;
                else {
                    Ic[p].ww = Ic[p].ww + I.ww; Ic[p].wa = Ic[p].wa + I.wa; Ic[p].wb = Ic[p].wb + I.wb;
                    Ic[p].aa = Ic[p].aa + I.aa; Ic[p].ab = Ic[p].ab + I.ab; Ic[p].bb = Ic[p].bb + I.bb;
                }
            }
        }
    }

    // in-place tree-sparse L^T D L factorisation; returns false if a pivot is not positive
    MBRL_HD bool factor(T (*H)[NV]) const {
        bool ok = true;
        MBRL_UNROLL
        for (int k = NV - 1; k >= 0; k--) {
            if (!(t_val(H[k][k]) > 0.0)) ok = false;
            const T inv = t_rcp(H[k][k]);
            MBRL_UNROLL
            for (int i = k - 1; i >= 0; i--) {
                if (!TP::T.dof_anc[i][k]) continue;
                const T a = H[k][i] * inv;
                MBRL_UNROLL
                for (int j = i; j >= 0; j--) {
                    if (!TP::T.dof_anc[j][i]) continue;
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
                if (TP::T.dof_anc[j][i]) x[j] = x[j] - H[i][j] * x[i];
        }
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) x[i] = x[i] * H[i][i];
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) {
            MBRL_UNROLL
            for (int j = i - 1; j >= 0; j--)
                if (TP::T.dof_anc[j][i]) x[i] = x[i] - H[i][j] * x[j];
        }
    }

    // ------------------------------------------------------------------ constraint rows
    // Row bookkeeping: `lm` bit j = limit row of dof j, bit NV+k = pair row k; `cm` bit 3*slot+r =
    // row r of the contact in that slot (pyramid: n + mu t, n - mu t, n (weight 2); condim 1: r = 2).
    // 1 + slot of floor candidate c (0: not in contact)
    MBRL_HD int slot1_of(int c) const {
        return ((cand_on >> c) & 1u) ? __builtin_popcount(cand_on & ((1u << c) - 1u)) + 1 : 0;
    }

    // top-down reader of a row stack (W doubles per row at element offset BASE): `cur` holds the top row
    template <int BASE, int W>
    struct Pop {
        T cur[W];
        int off;      // element offset of the row in `cur` (clamped at 0: popping an empty stack re-reads row 0)
        MBRL_HD void start(const Mem &mem, int height) {
            off = W * (height - 1);
            if (off < 0) off = 0;
            fetch(mem);
        }
        MBRL_HD void next(const Mem &mem) {
            off = off >= W ? off - W : 0;
            fetch(mem);
        }
        MBRL_HD void fetch(const Mem &mem) {
            MBRL_UNROLL
            for (int k = 0; k < W; k++) cur[k] = mem.ld(BASE + off + k);
            MBRL_LOOP_FENCE;   // pins the reads here: they must not sink into the block that consumes them
        }
    };

    // One leaf-to-root pass per Newton iteration: constraint residuals at x = qacc (body velocities Va),
    // the active-row masks, J^T f, and the Newton Hessian  M + arm + J^T D J  (lower triangle) -- contact
    // rows augment their body's inertia (D w w^T) and force (f w) before the body is folded into its
    // parent, so no per-body arrays of augmentations or forces exist.
    // WITH_H = false: forces and masks only (the closing evaluation of a converged solve).
    template <bool WITH_H>
    MBRL_HD void newton_pass(const T *x, T (*H)[NV], T *jtf, uint32_t &lm, CM &cm) const {
        lm = 0; cm = 0;
        Sp<T> Fp[NB];          // pair-row forces per body (models with capsule/capsule pairs only)
        if (M::NPAIR > 0) {
            MBRL_UNROLL
            for (int b = 0; b < NB; b++) { Fp[b].w = T(0.0); Fp[b].a = T(0.0); Fp[b].b = T(0.0); }
            MBRL_UNROLL
            for (int k = 0; k < M::NPAIR; k++) {
                if (pair_on & (1u << k)) {
                    const int b1 = M::g_body[M::pair_g1[k]], b2 = M::g_body[M::pair_g2[k]];
                    const int o = LY::OP + 5 * k;
                    const T w0 = mem.ld(o), w1 = mem.ld(o + 1), w2 = mem.ld(o + 2), D = mem.ld(o + 3), aref = mem.ld(o + 4);
                    const Sp<T> V1 = body_vel(b1, x), V2 = body_vel(b2, x);
                    const T jar = (w0 * (V2.w - V1.w) + w1 * (V2.a - V1.a) + w2 * (V2.b - V1.b)) - aref;
                    if (t_val(jar) < 0.0) {
                        lm |= 1u << (NV + k);
                        const T f = -D * jar;
                        Fp[b2].w = Fp[b2].w + f * w0; Fp[b2].a = Fp[b2].a + f * w1; Fp[b2].b = Fp[b2].b + f * w2;
                        Fp[b1].w = Fp[b1].w - f * w0; Fp[b1].a = Fp[b1].a - f * w1; Fp[b1].b = Fp[b1].b - f * w2;
                    }
                }
            }
        }
        Sym3 Ic[NB];
        Sp<T> Fs[NB];
        Pop<LY::OS, 5> cs;
        Pop<LY::OL, 2> ls;
        if (M::HAS_FLOOR) cs.start(mem, nslot);
        if (LY::NLIM > 0) ls.start(mem, nlimrow);
        int csh = 3 * (nslot - 1);          // bit position of the top slot's rows in `cm`
        MBRL_UNROLL
        for (int b = NB - 1; b >= 0; b--) {
            Sym3 I;
            I.ww = IO[b]; I.wa = -hb[b]; I.wb = ha[b]; I.aa = T(M::mass[b]); I.ab = T(0.0); I.bb = T(M::mass[b]);
            Sp<T> Fc;
            if (M::NPAIR > 0) Fc = Fp[b];
            else { Fc.w = T(0.0); Fc.a = T(0.0); Fc.b = T(0.0); }
            // ---- floor contacts of this body's capsules (descending candidate order = stack order) ----
            if (M::HAS_FLOOR) {
                MBRL_UNROLL
                for (int g = M::NGEOM - 1; g >= 0; g--) {
                    if (M::g_body[g] != b || !M::c_on[g]) continue;
                    if (!MBRL_RARE(cand_on & (3u << TP::T.cand_rank[g][0]))) continue;   // neither end of this capsule
                    MBRL_UNROLL
                    for (int e = 1; e >= 0; e--) {
                        if (cand_on & (1u << TP::T.cand_rank[g][e])) {
                            const T Pa = cs.cur[0], Pb = cs.cur[1], D = cs.cur[2], arn = cs.cur[3], art = cs.cur[4];
                            const int sh = csh;
                            csh -= 3;
                            cs.next(mem);
                            const Sp<T> Vb = body_vel(b, x);
                            const T rn = (Vb.b + Vb.w * Pa) - arn;
                            const T rt = (Vb.a - Vb.w * Pb) - art;
                            // Row weights w = wn*(Pa,0,1) + wt*(-Pb,1,0); force f w, Hessian D w w^T.  Branch-free:
                            // an inactive row (residual >= 0) contributes min(r, 0) = 0 and D_row = 0.
                            T fn, ft, dnn, dnt, dtt;
                            uint32_t rows;
                            if (M::c_dim[g] == 1) {
                                const bool on = t_val(rn) < 0.0;
                                rows = on ? 4u : 0u;
                                dnn = on ? D : T(0.0);
                                fn = -D * t_min(rn, T(0.0)); ft = T(0.0); dnt = T(0.0); dtt = T(0.0);
                            } else {
                                const double mu = M::c_mu[g];
                                const T r0 = rn + mu * rt, r1 = rn - mu * rt;
                                const bool on0 = t_val(r0) < 0.0, on1 = t_val(r1) < 0.0, on2 = t_val(rn) < 0.0;
                                rows = (on0 ? 1u : 0u) | (on1 ? 2u : 0u) | (on2 ? 4u : 0u);
                                const T d0 = on0 ? D : T(0.0), d1 = on1 ? D : T(0.0), d2 = on2 ? T(2.0) * D : T(0.0);
                                const T m0 = t_min(r0, T(0.0)), m1 = t_min(r1, T(0.0)), m2 = t_min(rn, T(0.0));
                                fn = -D * ((m0 + m1) + T(2.0) * m2);
                                ft = (-mu * D) * (m0 - m1);
                                dnn = (d0 + d1) + d2;
                                dnt = mu * (d0 - d1);
                                dtt = (mu * mu) * (d0 + d1);
                            }
                            cm |= CM(rows) << sh;
                            // n-direction (Pa, 0, 1), t-direction (-Pb, 1, 0)
                            Fc.w = Fc.w + (fn * Pa - ft * Pb); Fc.a = Fc.a + ft; Fc.b = Fc.b + fn;
                            if (WITH_H) {
                                I.ww = I.ww + ((dnn * Pa) * Pa - T(2.0) * (dnt * Pa) * Pb + (dtt * Pb) * Pb);
                                I.wa = I.wa + (dnt * Pa - dtt * Pb);
                                I.wb = I.wb + (dnn * Pa - dnt * Pb);
                                I.aa = I.aa + dtt; I.ab = I.ab + dnt; I.bb = I.bb + dnn;
                            }
                        }
                    }
                }
            }
            if (TP::T.any_child[b]) {   // children were folded into Ic[b], Fs[b] already
                if (WITH_H) {
                    I.ww = I.ww + Ic[b].ww; I.wa = I.wa + Ic[b].wa; I.wb = I.wb + Ic[b].wb;
                    I.aa = I.aa + Ic[b].aa; I.ab = I.ab + Ic[b].ab; I.bb = I.bb + Ic[b].bb;
                }
                Fc.w = Fc.w + Fs[b].w; Fc.a = Fc.a + Fs[b].a; Fc.b = Fc.b + Fs[b].b;
            }
            MBRL_UNROLL
            for (int k = M::dofnum[b] - 1; k >= 0; k--) {
                const int i = M::dofadr[b] + k;
                const Sp<T> S = Scol(i);
                T diag = T(M::arma[i]), fl = T(0.0);
                if (M::limited[i]) {
                    if (MBRL_RARE(lim_on & (1u << i))) {
                        const T D = ls.cur[0], aref = ls.cur[1];
                        ls.next(mem);
                        const bool up = (lim_hi >> i) & 1u;
                        const T jar = (up ? -x[i] : x[i]) - aref;
                        if (t_val(jar) < 0.0) {
                            lm |= 1u << i;
                            const T f = -D * jar;
                            fl = up ? -f : f;
                            diag = diag + D;
                        }
                    }
                }
                Sp<T> Fi;
                if (M::dof_type[i] == 0) { Fi.w = I.wa; Fi.a = I.aa; Fi.b = I.ab; jtf[i] = Fc.a + fl; }
                else if (M::dof_type[i] == 1) { Fi.w = I.wb; Fi.a = I.ab; Fi.b = I.bb; jtf[i] = Fc.b + fl; }
                else {
                    if (WITH_H) {
                        Fi.w = I.ww * S.w + I.wa * S.a + I.wb * S.b;
                        Fi.a = I.wa * S.w + I.aa * S.a + I.ab * S.b;
                        Fi.b = I.wb * S.w + I.ab * S.a + I.bb * S.b;
                    }
                    jtf[i] = (S.w * Fc.w + S.a * Fc.a + S.b * Fc.b) + fl;
                }
                MBRL_UNROLL
                for (int j = 0; j <= i; j++) {
                    if (!WITH_H || !TP::T.dof_anc[j][i]) continue;
                    T h;
                    if (M::dof_type[j] == 0) h = Fi.a;
                    else if (M::dof_type[j] == 1) h = Fi.b;
                    else { const Sp<T> Sj = Scol(j); h = Sj.w * Fi.w + Sj.a * Fi.a + Sj.b * Fi.b; }
                    if (j == i) h = h + diag;
                    H[i][j] = h;
                }
            }
            if (M::parent[b] >= 0) {
                const int p = M::parent[b];
                if (TP::T.first_folded[b]) { Ic[p] = I; Fs[p] = Fc; }
                else {
                    if (WITH_H) {
                        Ic[p].ww = Ic[p].ww + I.ww; Ic[p].wa = Ic[p].wa + I.wa; Ic[p].wb = Ic[p].wb + I.wb;
                        Ic[p].aa = Ic[p].aa + I.aa; Ic[p].ab = Ic[p].ab + I.ab; Ic[p].bb = Ic[p].bb + I.bb;
                    }
                    Fs[p].w = Fs[p].w + Fc.w; Fs[p].a = Fs[p].a + Fc.a; Fs[p].b = Fs[p].b + Fc.b;
                }
            }
        }
        if (WITH_H) pair_hessian(lm, H);
    }

    // derivative g and curvature h of the constraint cost along the search direction at step alpha
    MBRL_HD void line_pass(const T *x, const T *sx, T alpha, T &g, T &h, uint32_t &lm, CM &cm) const {
        lm = 0; cm = 0;
        Pop<LY::OS, 5> cs;
        Pop<LY::OL, 2> ls;
        if (M::HAS_FLOOR) cs.start(mem, nslot);
        if (LY::NLIM > 0) ls.start(mem, nlimrow);
        if (kWaveRows ? w_lim(low_bits(NV)) : MBRL_RARE(lim_on != 0)) {     // one test for all limit rows: active joint limits are rare
            MBRL_UNROLL
            for (int j = NV - 1; j >= 0; j--) {
                if (!M::limited[j]) continue;
                if (!w_lim(1u << j)) continue;
                if (lim_on & (1u << j)) {
                    const T D = ls.cur[0], aref = ls.cur[1];
                    ls.next(mem);
                    const bool up = (lim_hi >> j) & 1u;
                    const T jar = (up ? -x[j] : x[j]) - aref, jv = up ? -sx[j] : sx[j];
                    const T r = jar + alpha * jv;
                    if (t_val(r) < 0.0) { lm |= 1u << j; g = g + D * r * jv; h = h + D * jv * jv; }
                }
            }
        }
        MBRL_UNROLL
        for (int k = 0; k < M::NPAIR; k++) {
            if (!w_pair(1u << k)) continue;
            if (pair_on & (1u << k)) {
                const int b1 = M::g_body[M::pair_g1[k]], b2 = M::g_body[M::pair_g2[k]];
                const int o = LY::OP + 5 * k;
                const T w0 = mem.ld(o), w1 = mem.ld(o + 1), w2 = mem.ld(o + 2), D = mem.ld(o + 3), aref = mem.ld(o + 4);
                const Sp<T> V1 = body_vel(b1, x), V2 = body_vel(b2, x), S1 = body_vel(b1, sx), S2 = body_vel(b2, sx);
                const T jar = (w0 * (V2.w - V1.w) + w1 * (V2.a - V1.a) + w2 * (V2.b - V1.b)) - aref;
                const T jv = w0 * (S2.w - S1.w) + w1 * (S2.a - S1.a) + w2 * (S2.b - S1.b);
                const T r = jar + alpha * jv;
                if (t_val(r) < 0.0) { lm |= 1u << (NV + k); g = g + D * r * jv; h = h + D * jv * jv; }
            }
        }
        if (!M::HAS_FLOOR) return;
        if (kWaveRows ? !w_cand(low_bits(LY::NCAND)) : cand_on == 0) return;
        int csh = 3 * (nslot - 1);
        MBRL_UNROLL
        for (int gi = M::NGEOM - 1; gi >= 0; gi--) {
            if (!M::c_on[gi]) continue;
            if (kWaveRows ? !w_cand(3u << TP::T.cand_rank[gi][0])
                          : !MBRL_RARE(cand_on & (3u << TP::T.cand_rank[gi][0]))) continue;   // neither end of this capsule
            const int b = M::g_body[gi];
            MBRL_UNROLL
            for (int e = 1; e >= 0; e--) {
                if (!w_cand(1u << TP::T.cand_rank[gi][e])) continue;
                if (cand_on & (1u << TP::T.cand_rank[gi][e])) {
                    const T Pa = cs.cur[0], Pb = cs.cur[1], D = cs.cur[2], arn = cs.cur[3], art = cs.cur[4];
                    const int sh = csh;
                    csh -= 3;
                    cs.next(mem);
                    const Sp<T> Vb = body_vel(b, x), Sb = body_vel(b, sx);
                    const T rn = (Vb.b + Vb.w * Pa) - arn;
                    const T rt = (Vb.a - Vb.w * Pb) - art;
                    const T sn = Sb.b + Sb.w * Pa, st = Sb.a - Sb.w * Pb;
                    if (M::c_dim[gi] == 1) {
                        const T r = rn + alpha * sn;
                        const bool on = t_val(r) < 0.0;
                        cm |= CM(on ? 4u : 0u) << sh;
                        const T d = on ? D : T(0.0);
                        g = g + (d * r) * sn; h = h + (d * sn) * sn;
                    } else {
                        const double mu = M::c_mu[gi];
                        const T j0 = sn + mu * st, j1 = sn - mu * st;
                        const T r0 = (rn + mu * rt) + alpha * j0, r1 = (rn - mu * rt) + alpha * j1, r2 = rn + alpha * sn;
                        const bool on0 = t_val(r0) < 0.0, on1 = t_val(r1) < 0.0, on2 = t_val(r2) < 0.0;
                        cm |= CM((on0 ? 1u : 0u) | (on1 ? 2u : 0u) | (on2 ? 4u : 0u)) << sh;
                        const T d0 = on0 ? D : T(0.0), d1 = on1 ? D : T(0.0), d2 = on2 ? T(2.0) * D : T(0.0);
                        g = g + (((d0 * r0) * j0 + (d1 * r1) * j1) + (d2 * r2) * sn);
                        h = h + (((d0 * j0) * j0 + (d1 * j1) * j1) + (d2 * sn) * sn);
                    }
                }
            }
        }
    }

    // explicit D r r^T of an active pair row on the dofs between its two bodies (b1 is an ancestor of b2)
    MBRL_HD void pair_hessian(uint32_t lm, T (*H)[NV]) const {
        MBRL_UNROLL
        for (int k = 0; k < M::NPAIR; k++) {
            if (lm & (1u << (NV + k))) {
                const int b1 = M::g_body[M::pair_g1[k]], b2 = M::g_body[M::pair_g2[k]];
                const int o = LY::OP + 5 * k;
                const T w0 = mem.ld(o), w1 = mem.ld(o + 1), w2 = mem.ld(o + 2), D = mem.ld(o + 3);
                T rr[NV];
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) {
                    if (!TP::T.between[k][j]) continue;
                    const Sp<T> Sj = Scol(j);
                    rr[j] = w0 * Sj.w + w1 * Sj.a + w2 * Sj.b;
                }
                MBRL_UNROLL
                for (int i = 0; i < NV; i++) {
                    if (!TP::T.between[k][i]) continue;
                    MBRL_UNROLL
                    for (int j = 0; j <= i; j++) {
                        if (!TP::T.between[k][j]) continue;
                        H[i][j] = H[i][j] + D * rr[i] * rr[j];
                    }
                }
            }
        }
    }

    // ------------------------------------------------------------------ explicit rows on the shared inertia HM
    // Row of a point constraint at P (relative to O) on body b: along the floor normal  Jn[j] = w_n . S_j  with
    // w_n = (Pa, 0, 1), along the tangent  Jt[j] = w_t . S_j  with w_t = (-Pb, 1, 0).  For a planar tree these are just
    // lever arms: a hinge j gives  sgn_j (Pa + ra_j)  and  -sgn_j (Pb + rb_j),  a slide along a gives (0, 1), along b (1, 0).
    // Only the dofs that move b are filled in; the slides' constants are folded where they are used.
    MBRL_HD void point_rows(int b, T Pa, T Pb, T *Jn, T *Jt) const {
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            if (!TP::T.moves[j][b] || M::dof_type[j] != 2) continue;
            Jn[j] = M::dof_sgn[j] * (Pa + ra[j]);
            Jt[j] = -M::dof_sgn[j] * (Pb + rb[j]);
        }
    }
    // (Jn . x, Jt . x) over the dofs that move b
    MBRL_HD void point_vel(int b, const T *Jn, const T *Jt, const T *x, T &vn, T &vt) const {
        vn = T(0.0); vt = T(0.0);
        bool fn = true, ft = true;          // first term: assign instead of adding to zero
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            if (!TP::T.moves[j][b]) continue;
            if (M::dof_type[j] == 0) { vt = ft ? x[j] : vt + x[j]; ft = false; }
            else if (M::dof_type[j] == 1) { vn = fn ? x[j] : vn + x[j]; fn = false; }
            else {
                vn = fn ? Jn[j] * x[j] : vn + Jn[j] * x[j]; fn = false;
                vt = ft ? Jt[j] * x[j] : vt + Jt[j] * x[j]; ft = false;
            }
        }
    }

    // One Newton iteration's constraint pass on the shared inertia: residuals at x, the active-row masks, jtf = J^T f and
    // H = HM + J^T D J (lower triangle).  No tree walk: every active row adds its force and its rank-one (contacts: rank-
    // two) block explicitly, on the dofs that move its body.  Stacks are popped top-down: contacts in descending candidate
    // order, then limits in descending dof order.
    MBRL_HD void newton_rows(const T *x, T (*H)[NV], T *jtf, uint32_t &lm, CM &cm) const {
        lm = 0; cm = 0;
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) {
            jtf[i] = T(0.0);
            MBRL_UNROLL
            for (int j = 0; j <= i; j++)
                if (TP::T.dof_anc[j][i]) H[i][j] = HM[i][j];
        }
        // ---- capsule/capsule pairs (b1 is an ancestor of b2: the row acts on the dofs between them) ----
        MBRL_UNROLL
        for (int k = 0; k < M::NPAIR; k++) {
            if (!w_pair(1u << k)) continue;
            if (pair_on & (1u << k)) {
                const int o = LY::OP + 5 * k;
                const T w0 = mem.ld(o), w1 = mem.ld(o + 1), w2 = mem.ld(o + 2), D = mem.ld(o + 3), aref = mem.ld(o + 4);
                T rr[NV];
                T jar = -aref;
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) {
                    if (!TP::T.between[k][j]) continue;
                    const Sp<T> Sj = Scol(j);
                    rr[j] = w0 * Sj.w + w1 * Sj.a + w2 * Sj.b;
                    jar = jar + rr[j] * x[j];
                }
                if (t_val(jar) < 0.0) {
                    lm |= 1u << (NV + k);
                    const T f = -D * jar;
                    MBRL_UNROLL
                    for (int i = 0; i < NV; i++) {
                        if (!TP::T.between[k][i]) continue;
                        jtf[i] = jtf[i] + f * rr[i];
                        const T di = D * rr[i];
                        MBRL_UNROLL
                        for (int j = 0; j <= i; j++) {
                            if (!TP::T.between[k][j]) continue;
                            H[i][j] = H[i][j] + di * rr[j];
                        }
                    }
                }
            }
        }
        // ---- floor contacts ----
        if (M::HAS_FLOOR) {
            if (kWaveRows ? w_cand(low_bits(LY::NCAND)) : MBRL_RARE(cand_on != 0)) {
                Pop<LY::OS, 5> cs;
                cs.start(mem, nslot);
                int csh = 3 * (nslot - 1);          // bit position of the top slot's rows in `cm`
                MBRL_UNROLL
                for (int g = M::NGEOM - 1; g >= 0; g--) {
                    if (!M::c_on[g]) continue;
                    if (kWaveRows ? !w_cand(3u << TP::T.cand_rank[g][0])
                                  : !MBRL_RARE(cand_on & (3u << TP::T.cand_rank[g][0]))) continue;   // neither end of this capsule
                    const int b = M::g_body[g];
                    MBRL_UNROLL
                    for (int e = 1; e >= 0; e--) {
                        if (!w_cand(1u << TP::T.cand_rank[g][e])) continue;
                        if (cand_on & (1u << TP::T.cand_rank[g][e])) {
                            T Pa = cs.cur[0], Pb = cs.cur[1];
                            const T D = cs.cur[2], arn = cs.cur[3], art = cs.cur[4];
                            const int sh = csh;
                            csh -= 3;
                            cs.next(mem);
                            MBRL_ANCHOR(Pa); MBRL_ANCHOR(Pb);
                            T Jn[NV], Jt[NV], vn, vt;
                            point_rows(b, Pa, Pb, Jn, Jt);
                            point_vel(b, Jn, Jt, x, vn, vt);
                            const T rn = vn - arn, rt = vt - art;
                            // Pyramid rows n + mu t, n - mu t, n (weight 2) in the (n, t) basis.  Branch-free: an inactive
                            // row (residual >= 0) contributes min(r, 0) = 0 and D_row = 0.
                            T fn, ft, dnn, dnt, dtt;
                            uint32_t rows;
                            if (M::c_dim[g] == 1) {
                                const bool on = t_val(rn) < 0.0;
                                rows = on ? 4u : 0u;
                                dnn = on ? D : T(0.0);
                                fn = -D * t_min(rn, T(0.0)); ft = T(0.0); dnt = T(0.0); dtt = T(0.0);
                            } else {
                                const double mu = M::c_mu[g];
                                const T r0 = rn + mu * rt, r1 = rn - mu * rt;
                                const bool on0 = t_val(r0) < 0.0, on1 = t_val(r1) < 0.0, on2 = t_val(rn) < 0.0;
                                rows = (on0 ? 1u : 0u) | (on1 ? 2u : 0u) | (on2 ? 4u : 0u);
                                const T d0 = on0 ? D : T(0.0), d1 = on1 ? D : T(0.0), d2 = on2 ? T(2.0) * D : T(0.0);
                                const T m0 = t_min(r0, T(0.0)), m1 = t_min(r1, T(0.0)), m2 = t_min(rn, T(0.0));
                                fn = -D * ((m0 + m1) + T(2.0) * m2);
                                ft = (-mu * D) * (m0 - m1);
                                dnn = (d0 + d1) + d2;
                                dnt = mu * (d0 - d1);
                                dtt = (mu * mu) * (d0 + d1);
                            }
                            cm |= CM(rows) << sh;
                            const bool tang = M::c_dim[g] != 1;
                            // force and Hessian block on the dofs that move b:  An = dnn Jn + dnt Jt,  At = dnt Jn + dtt Jt,
                            // H[i][j] += An_i Jn_j + At_i Jt_j  (slides: Jn, Jt are 0 / 1)
                            MBRL_UNROLL
                            for (int i = 0; i < NV; i++) {
                                if (!TP::T.moves[i][b]) continue;
                                T An, At;
                                if (M::dof_type[i] == 0) { An = dnt; At = dtt; if (tang) jtf[i] = jtf[i] + ft; }
                                else if (M::dof_type[i] == 1) { An = dnn; At = dnt; jtf[i] = jtf[i] + fn; }
                                else {
                                    if (tang) { An = dnn * Jn[i] + dnt * Jt[i]; At = dnt * Jn[i] + dtt * Jt[i]; jtf[i] = (jtf[i] + fn * Jn[i]) + ft * Jt[i]; }
                                    else { An = dnn * Jn[i]; At = T(0.0); jtf[i] = jtf[i] + fn * Jn[i]; }
                                }
                                MBRL_UNROLL
                                for (int j = 0; j <= i; j++) {
                                    if (!TP::T.moves[j][b]) continue;
                                    if (M::dof_type[j] == 0) { if (tang) H[i][j] = H[i][j] + At; }
                                    else if (M::dof_type[j] == 1) H[i][j] = H[i][j] + An;
                                    else if (tang) H[i][j] = (H[i][j] + An * Jn[j]) + At * Jt[j];
                                    else H[i][j] = H[i][j] + An * Jn[j];
                                }
                            }
                        }
                    }
                }
            }
        }
        // ---- joint limits ----
        if (LY::NLIM > 0) {
            if (kWaveRows ? w_lim(low_bits(NV)) : MBRL_RARE(lim_on != 0)) {
                Pop<LY::OL, 2> ls;
                ls.start(mem, nlimrow);
                MBRL_UNROLL
                for (int i = NV - 1; i >= 0; i--) {
                    if (!M::limited[i]) continue;
                    if (!w_lim(1u << i)) continue;
                    if (lim_on & (1u << i)) {
                        T D = ls.cur[0], aref = ls.cur[1];
                        ls.next(mem);
                        MBRL_ANCHOR(aref);
                        const bool up = (lim_hi >> i) & 1u;
                        const T jar = (up ? -x[i] : x[i]) - aref;
                        if (t_val(jar) < 0.0) {
                            lm |= 1u << i;
                            const T f = -D * jar;
                            jtf[i] = jtf[i] + (up ? -f : f);
                            H[i][i] = H[i][i] + D;
                        }
                    }
                }
            }
        }
    }

    // y = HM x = (M + arm) x  from the stored lower triangle (tree-sparse, symmetric)
    MBRL_HD void mass_mul_hm(const T *x, T *y) const {
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) y[i] = HM[i][i] * x[i];
        MBRL_UNROLL
        for (int i = 0; i < NV; i++) {
            MBRL_UNROLL
            for (int j = 0; j < i; j++) {
                if (!TP::T.dof_anc[j][i]) continue;
                y[i] = y[i] + HM[i][j] * x[j];
                y[j] = y[j] + HM[i][j] * x[i];
            }
        }
    }

    // y = (M + arm) x
    MBRL_HD void mass_mul(const T *x, T *y) const {
        Sp<T> Vs[NB], Fm[NB];
        vel_pass(x, Vs);
        MBRL_UNROLL
        for (int b = 0; b < NB; b++) Fm[b] = inert_mul(b, Vs[b]);
        back_pass(Fm, y);
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) y[j] = y[j] + M::arma[j] * x[j];
    }

    // primal Newton with exact line search, run until the active set reproduces itself.
    // In: qacc = unconstrained acceleration (M + arm) qacc = qfrc_smooth.  Out: qacc, fcon.
    // The smooth part of the gradient, gs = (M + arm) qacc - qfrc_smooth, is carried along the
    // iteration (0 at the start, += alpha (M + arm) s per step) instead of being re-evaluated.
    MBRL_HD void solve_constraints() {
        newton_iters = 0;
        uint32_t plm = ~0u;
        CM pcm = ~CM(0);
        bool converged = false;
        T gs[NV];
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) gs[j] = T(0.0);
        auto body = [&](const int it) -> bool {       // one Newton iteration; true: leave the loop
            MBRL_LOOP_FENCE;
            MBRL_OPAQUE(cand_on); MBRL_OPAQUE(lim_on);   // keep the per-row bit tests inside the loop
#ifdef MBRL_OPAQUE_MORE
            MBRL_OPAQUE(lim_hi); MBRL_OPAQUE(nslot); MBRL_OPAQUE(nlimrow);   // ... and the stack offsets / limit sides
#endif
            T H[NV][NV], jtf[NV], sx[NV];
            uint32_t lm;
            CM cm;
            MBRL_PHASE(8);
            newton_iters = it + 1;
            MBRL_MARK("pass_begin");
            if constexpr (kHM) newton_rows(qacc, H, jtf, lm, cm);
            else newton_pass<true>(qacc, H, jtf, lm, cm);
            MBRL_MARK("pass_end");
            MBRL_PIN(jtf, NV);
            MBRL_UNROLL
            for (int i = 0; i < NV; i++) {
                MBRL_UNROLL
                for (int j = 0; j <= i; j++)
                    if (TP::T.dof_anc[j][i]) MBRL_PIN(&H[i][j], 1);
            }
            MBRL_PHASE(2);
            if ((lm == plm && cm == pcm) || it == kMaxNewton) {
                converged = (lm == plm && cm == pcm);
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) fcon[j] = jtf[j];
                return true;
            }
            plm = lm; pcm = cm;
            MBRL_MARK("factor_begin");
            const bool ok = factor(H);
            if (!ok) { status |= ST_NONFINITE; return true; }
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) sx[j] = jtf[j] - gs[j];     // - gradient
            solve(H, sx);
            T snorm = T(0.0), qnorm = T(0.0), a1 = T(0.0);
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { snorm = snorm + sx[j] * sx[j]; qnorm = qnorm + qacc[j] * qacc[j]; a1 = a1 + sx[j] * gs[j]; }
            MBRL_PIN(sx, NV);
            MBRL_MARK("solve_end");
            MBRL_PHASE(3);
            if (!t_finite(snorm)) { status |= ST_NONFINITE; return true; }
            if (t_val(t_sqrt(snorm)) <= 1e-15 * (1.0 + t_val(t_sqrt(qnorm)))) {
                converged = true;
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) fcon[j] = jtf[j];
                return true;
            }
            // exact line search: root of the piecewise-linear derivative (safeguarded Newton)
            T Ms[NV];
            if constexpr (kHM) mass_mul_hm(sx, Ms);
            else mass_mul(sx, Ms);
            T a2 = T(0.0);
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) a2 = a2 + sx[j] * Ms[j];
            MBRL_PIN(Ms, NV);
            MBRL_MARK("ms_end");
            MBRL_PHASE(4);
            if (it < kUnitIters<M>) {
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) { qacc[j] = qacc[j] + sx[j]; gs[j] = gs[j] + Ms[j]; }
                return false;
            }
            T alpha = T(1.0), lo = T(0.0), hi = T(-1.0);  // hi < 0: no upper bracket yet
            uint32_t llm = lm;                             // active set the current alpha was derived from
            CM lcm = cm;
            bool exact = false;
            MBRL_NOUNROLL
            for (int ls = 0; ls < kMaxLineSearch; ls++) {
                MBRL_LOOP_FENCE;
                MBRL_OPAQUE(cand_on); MBRL_OPAQUE(lim_on);
                T g = a1 + alpha * a2, h = a2;
                uint32_t nlm;
                CM ncm;
                ls_iters++;
                line_pass(qacc, sx, alpha, g, h, nlm, ncm);
                if (nlm == llm && ncm == lcm) { exact = true; break; }   // alpha is the exact root of its own piece
                llm = nlm; lcm = ncm;
                if (t_val(g) < 0.0) lo = alpha; else hi = alpha;
                T an = alpha - g / h;
                const bool inside = t_val(an) > t_val(lo) && (t_val(hi) < 0.0 || t_val(an) < t_val(hi));
                if (!inside) an = (t_val(hi) < 0.0) ? T(2.0) * alpha : T(0.5) * (lo + hi);
                alpha = an;
            }
            MBRL_PHASE(5);
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { qacc[j] = qacc[j] + alpha * sx[j]; gs[j] = gs[j] + alpha * Ms[j]; }
#ifndef MBRL_NO_STATIONARY_EXIT
            // The search ended on the very active set the Hessian was built from: the new point is the Newton point
            // of a set that reproduces itself, i.e. the solution, and stationarity there reads  J^T f = gs.
            if (exact && llm == plm && lcm == pcm) {
                converged = true;
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) fcon[j] = gs[j];
                return true;
            }
#endif
            return false;
        };
        // How many iterations run as straight-line code before the loop (the usual solve is pass, step, closing pass).
        // Same arithmetic either way; measured per model (profiles/r02_experiments.md, 65 536 units): Hopper gains 6 % with
        // two peeled iterations, HalfCheetah and Walker2d lose 2-4 % (larger kernels: the copies cost more than the
        // loop's register shuffle), so only Hopper peels.
#ifdef MBRL_PEEL
        constexpr int kPeel = MBRL_PEEL;                     // A/B builds
#else
        constexpr int kPeel = (M::ENV_ID == 3) ? 2 : 0;
#endif
        bool done = false;
        MBRL_UNROLL
        for (int it = 0; it < kPeel; it++)
            if (!done) done = body(it);
        if (kPeel == 0 || MBRL_RARE(!done)) {
            MBRL_NOUNROLL
            for (int it = kPeel; it <= kMaxNewton; it++)
                if (body(it)) break;
        }
        if (!converged) status |= ST_NOCONVERGE;
    }

    // ------------------------------------------------------------------ mj_forward
    MBRL_HD void forward() {
        MBRL_PHASE(8);
        MBRL_MARK("sweep_begin");
        {
            Sp<T> F[NB];
            sweep(F);
            smooth_forces(F);
        }
        MBRL_PIN(fsm, NV); MBRL_PIN(ra, NV); MBRL_PIN(rb, NV); MBRL_PIN(ha, NB); MBRL_PIN(hb, NB); MBRL_PIN(IO, NB);
        MBRL_MARK("sweep_end");
        MBRL_PHASE(0);
        const bool constrained = (lim_on | cand_on | pair_on) != 0;
        was_constrained = constrained;
        // the wave-wide OR of the row masks goes through the LDS while HM is built and factorised
        const uint32_t my_rows = cand_on | (lim_on << LY::NCAND) | (pair_on << (LY::NCAND + NV));
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kWaveRows) wave_or_begin(my_rows);
#endif
        // Start of the solver: the unconstrained acceleration.  (Starting from the previous mj_step's solution was
        // measured: it does not lower the number of Newton passes on on-trajectory HalfCheetah states, 2.67 vs 2.57
        // per constrained step.)
        if constexpr (kHM) {
            T arm[NV];
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) arm[j] = T(M::arma[j]);
            crba(HM, arm);                                 // once per dynamics evaluation; see the member's comment
            if (M::RK4 || !M::ANY_DAMPING || constrained) {   // Euler with implicit damping needs qacc only for the solver
                T H[NV][NV];
                MBRL_UNROLL
                for (int i = 0; i < NV; i++) {
                    qacc[i] = fsm[i];
                    MBRL_UNROLL
                    for (int j = 0; j <= i; j++)
                        if (TP::T.dof_anc[j][i]) H[i][j] = HM[i][j];
                }
                if (!factor(H)) status |= ST_NONFINITE;
                solve(H, qacc);
            }
        } else if (M::RK4 || !M::ANY_DAMPING || constrained) {   // Euler with implicit damping needs qacc only for the solver
            T H[NV][NV], arm[NV];
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { arm[j] = T(M::arma[j]); qacc[j] = fsm[j]; }
            crba(H, arm);
            if (!factor(H)) status |= ST_NONFINITE;
            solve(H, qacc);
        }
        MBRL_PIN(qacc, NV);
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kWaveRows) wave_rows = wave_or_end();
#else
        wave_rows = my_rows;
#endif
        MBRL_MARK("unconstrained_end");
        MBRL_PHASE(1);
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) fcon[j] = T(0.0);
        newton_iters = 0;
        ls_iters = 0;
        if (constrained) solve_constraints();
        MBRL_PIN(qacc, NV); MBRL_PIN(fcon, NV);
        MBRL_MARK("solver_end");
        MBRL_PHASE(6);
    }

    // ------------------------------------------------------------------ integrators
    MBRL_HD void euler() {
        MBRL_MARK("euler_begin");
        T x[NV];
        if (M::ANY_DAMPING) {
            T H[NV][NV], dg[NV];
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { dg[j] = T(M::arma[j]) + T(M::DT) * M::damp[j]; x[j] = fsm[j] + fcon[j]; }
            if constexpr (kHM) {        // M + arm is at hand: add h D on the diagonal
#ifndef MBRL_NO_STATIONARY_FORCE
                // Total force of a constrained evaluation: at the solver's fixed point  (M + arm) qacc = qfrc_smooth + J^T f,
                // so it is HM qacc -- neither qfrc_smooth nor the constraint force has to outlive the solver.
                if (was_constrained) mass_mul_hm(qacc, x);
                else {
                    MBRL_UNROLL
                    for (int j = 0; j < NV; j++) x[j] = fsm[j];
                }
#endif
                MBRL_UNROLL
                for (int i = 0; i < NV; i++) {
                    MBRL_UNROLL
                    for (int j = 0; j <= i; j++)
                        if (TP::T.dof_anc[j][i]) H[i][j] = (j == i && M::damp[i] != 0.0) ? HM[i][i] + T(M::DT * M::damp[i]) : HM[i][j];
                }
            } else crba(H, dg);
            if (!factor(H)) status |= ST_NONFINITE;
            solve(H, x);
        } else {
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) x[j] = qacc[j];
        }
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            const T vn = mem.ld(LY::OV + j) + M::DT * x[j];
            mem.st(LY::OV + j, vn);
            mem.st(LY::OQ + j, mem.ld(LY::OQ + j) + M::DT * vn);
        }
        MBRL_MARK("euler_end");
    }

    static constexpr int key_first_lim() { int f = 0; while (f < NV - 1 && !M::limited[f]) f++; return f; }
    static constexpr int key_last_lim() { int l = NV - 1; while (l > 0 && !M::limited[l]) l--; return l; }
    static constexpr int KEY_LIM0 = key_first_lim(), KEY_NLIM = key_last_lim() - key_first_lim() + 1;
    static constexpr int KEY_U = LY::NCAND + KEY_NLIM + M::NPAIR;      // bits of the row union
    static constexpr int KEY_REC = (62 - 6 - KEY_U) / 2;               // steps whose passes fit behind it
    static_assert(KEY_U <= 24 && KEY_REC >= 8, "row union must fit key2's low 24 bits");
    // `steps` = mj_steps of the whole rollout (frame_skip x env steps)
    MBRL_HD void key_init(int steps) { key = 0; key2 = 0; key_left = steps <= 10 ? -1 : KEY_REC; }
    // rows active in the dynamics evaluation just made, packed for the union
    MBRL_HD uint32_t key_rows() const {
        return cand_on | ((lim_on >> KEY_LIM0) << LY::NCAND) | (pair_on << (LY::NCAND + KEY_NLIM));
    }
    // one mj_step: its Hessian passes (RK4: most of a stage), line-search evaluations and active rows
    MBRL_HD void key_note(uint32_t passes, uint32_t line_evals, uint32_t rows) {
        const uint32_t p7 = passes < 7u ? passes : 7u;
        if (key_left < 0) {
            key = (key << 3) | uint64_t(p7);
#if defined(MBRL_KEY_ROUND2)
            key2 = (key2 << 3) | (line_evals < 7u ? line_evals : 7u);
#else
            key2 |= rows;                                   // the union of the rows that were ever active
#endif
            return;
        }
        if (key_left > 0) {
            key = (key << 2) | uint64_t(p7 < 2u ? 0u : (p7 > 4u ? 3u : p7 - 1u));   // 0-1, 2, 3, 4+ passes
            key_left--;
        }
        uint32_t tot = (key2 >> 24) + p7;
        tot = tot < 255u ? tot : 255u;
        key2 = ((key2 | rows) & 0xffffffu) | (tot << 24);
    }
    MBRL_HD uint64_t key_value(int steps) const {
        // Layouts since the wave-wide row mask (profiles/r03_experiments.md, ten layouts measured): units that share
        // their floor contacts share a wave -- the solver passes then skip the other contacts' rows with scalar branches --
        // and inside such a group the pass counts order them as before.
#if defined(MBRL_KEY_ROUND2)
        if (steps <= 10) return ((key & 0x3fffffffull) << 32) | uint64_t(key2 & 0x3fffffffu);
#else
        if (steps <= 10) {   // [floor-contact union][3 bits of passes per step][limit / pair union]
            const uint32_t u = key2 & 0xffffffu, cu = u & low_bits(LY::NCAND), ru = u >> LY::NCAND;
            return (uint64_t(cu) << 38) | ((key & 0x3fffffffull) << 8) | uint64_t(ru & 0xffu);
        }
#endif
        const uint32_t tq = (key2 >> 26) < 63u ? (key2 >> 26) : 63u;           // pass total / 4
#if defined(MBRL_KEY_ROUND2)
        return (uint64_t(tq) << (KEY_U + 2 * KEY_REC)) | (uint64_t(key2 & ((1u << KEY_U) - 1u)) << (2 * KEY_REC)) | key;
#else
        {   // [floor-contact union][pass total / 4][2 bits of passes per recorded step][limit / pair union]
            const uint32_t u = key2 & ((1u << KEY_U) - 1u), cu = u & low_bits(LY::NCAND), ru = u >> LY::NCAND;
            constexpr int RB = KEY_U - LY::NCAND;
            return (uint64_t(cu) << (6 + 2 * KEY_REC + RB)) | (uint64_t(tq) << (2 * KEY_REC + RB)) | (key << RB) | uint64_t(ru);
        }
#endif
    }

    MBRL_HD void step() {
        if (!M::RK4) { forward(); key_note(uint32_t(newton_iters), ls_iters, key_rows()); euler(); MBRL_PHASE(7); return; }
        // classic RK4; stage i uses only stage i-1 (tableau 1/2, 1/2, 1), weights 1/6 1/3 1/3 1/6.  ONE call site of
        // forward() for the four stages: inlined twice (first stage + loop) the Walker2d kernel was 97 KB of code,
        // beyond the 64 KB instruction cache.
        T q0[NV], v0[NV], aq[NV], av[NV];
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) {
            q0[j] = mem.ld(LY::OQ + j); v0[j] = mem.ld(LY::OV + j);
            aq[j] = T(0.0); av[j] = T(0.0);
        }
        uint32_t kpass = 0, kline = 0, krows = 0;   // this step's signature: most passes of a stage, line evaluations and rows of all
        MBRL_NOUNROLL
        for (int i = 0; i < 4; i++) {
            if (i > 0) {
                const double ai = (i == 3) ? 1.0 : 0.5;
                MBRL_UNROLL
                for (int j = 0; j < NV; j++) {
                    const T dv = ai * mem.ld(LY::OV + j), da = ai * qacc[j];
                    mem.st(LY::OQ + j, q0[j] + M::DT * dv);
                    mem.st(LY::OV + j, v0[j] + M::DT * da);
                }
            }
            forward();
            kpass = uint32_t(newton_iters) > kpass ? uint32_t(newton_iters) : kpass;
            kline += ls_iters;
            krows |= key_rows();
            const double bi = (i == 0 || i == 3) ? 1.0 / 6 : 1.0 / 3;
            MBRL_UNROLL
            for (int j = 0; j < NV; j++) { aq[j] = aq[j] + bi * mem.ld(LY::OV + j); av[j] = av[j] + bi * qacc[j]; }
        }
        key_note(kpass, kline, krows);
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) { mem.st(LY::OQ + j, q0[j] + M::DT * aq[j]); mem.st(LY::OV + j, v0[j] + M::DT * av[j]); }
    }

    // ------------------------------------------------------------------ env.step
    // state and ctrl are in `mem`; returns the reward.
    MBRL_HD T env_step(int frame_skip) {
        const T xb = mem.ld(LY::OQ);
        const double dt = M::DT * frame_skip;
        MBRL_NOUNROLL
        for (int k = 0; k < frame_skip; k++) step();
        T asq = T(0.0);
        MBRL_UNROLL
        for (int u = 0; u < NU; u++) { const T cu = mem.ld(LY::OC + u); asq = asq + cu * cu; }
        const double PI = 3.14159265358979323846;
        const T q0 = mem.ld(LY::OQ);
        T r = T(0.0);
        if (M::REWARD == 0) {         // inverted_pendulum.py:26-35
            const T y = T(0.6) * t_cos(mem.ld(LY::OQ + 1)), x = y;
            const T cu = mem.ld(LY::OC);
            r = -(T(0.01) * (x * x) + (y - T(1.0)) * (y - T(1.0))) - T(0.001) * (cu * cu);
        } else if (M::REWARD == 1) {  // inverted_double_pendulum.py:28-33 (stale site_xpos)
            const T x = site_a, y = site_b;
            const T pen = T(0.01) * (x * x) + (y - T(2.0)) * (y - T(2.0));
            const T v1 = mem.ld(LY::OV + 1), v2 = mem.ld(LY::OV + (NV > 2 ? 2 : 0));
            const T vp = T(1e-3) * (v1 * v1) + T(5e-3) * (v2 * v2);
            r = -pen - vp;
        } else if (M::REWARD == 2) {  // swimmer.py:23-29
            r = (q0 - xb) / dt + (T(-0.0001) * asq);
        } else if (M::REWARD == 3) {  // hopper.py:30-41
            const T q1 = mem.ld(LY::OQ + 1), q2 = mem.ld(LY::OQ + (NV > 2 ? 2 : 0));
            r = (q0 - xb) / dt;
            T aa = t_fmod(t_abs(q2), T(2 * PI));
            if (t_val(aa) > PI) aa = T(2 * PI) - aa;
            const T c1 = T(1.0) / (T(1.0) + t_pow(T(200.0), -(q1 / 1.25) + T(0.5)));
            const T c2 = T(1.0) / (T(1.0) + t_pow(T(200.0), -((T(PI) - aa) / PI) + T(0.5)));
            r = r + (c1 * 1.0 + c2 * 1.0);
            r = r - T(1e-3) * asq;
        } else if (M::REWARD == 4) {  // walker2d.py:23-30
            const T q1 = mem.ld(LY::OQ + 1), q2 = mem.ld(LY::OQ + (NV > 2 ? 2 : 0));
            r = (q0 - xb) / dt;
            const T coeff = t_min(t_max(q1 / 0.8, T(0.0)), T(1.0)) * 0.5 +
                            t_max((T(PI) - t_abs(q2)) / PI, T(1.0)) * 0.5;
            r = r + coeff * 1.0;
            r = r - T(1e-3) * asq;
        } else {                      // half_cheetah.py:25-31
            r = T(-0.01) * asq + (q0 - xb) / dt;
        }
        return r;
    }

    MBRL_HD void check_finite() {
        bool ok = true;
        MBRL_UNROLL
        for (int j = 0; j < NV; j++) ok = ok && t_finite(mem.ld(LY::OQ + j)) && t_finite(mem.ld(LY::OV + j));
        if (!ok) status |= ST_NONFINITE;
    }
};

}  // namespace mbrl
