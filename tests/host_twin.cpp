// Host build of the kernel's per-lane simulator (planar_lean.hpp) -- TEST TOOL ONLY.
// Lets CPU tests compare the planar formulation with the 3-D oracle without a GPU and
// counts its floating-point operations (FMA = 2, add/mul = 1, div/sqrt/transcendental = 1).
#include <cstdint>
#include <cstring>
#include "../model-based-rl_amd/csrc/planar_lean.hpp"

namespace mbrl {
struct Cnt {
    double x;
    static long long n;
    Cnt() : x(0) {}
    Cnt(double v) : x(v) {}
};
long long Cnt::n = 0;
inline Cnt operator+(Cnt a, Cnt b) { Cnt::n++; return Cnt(a.x + b.x); }
inline Cnt operator-(Cnt a, Cnt b) { Cnt::n++; return Cnt(a.x - b.x); }
inline Cnt operator*(Cnt a, Cnt b) { Cnt::n++; return Cnt(a.x * b.x); }
inline Cnt operator/(Cnt a, Cnt b) { Cnt::n++; return Cnt(a.x / b.x); }
inline Cnt operator-(Cnt a) { return Cnt(-a.x); }
inline Cnt operator+(double a, Cnt b) { return Cnt(a) + b; }
inline Cnt operator-(double a, Cnt b) { return Cnt(a) - b; }
inline Cnt operator*(double a, Cnt b) { return Cnt(a) * b; }
inline Cnt operator/(double a, Cnt b) { return Cnt(a) / b; }
inline Cnt operator+(Cnt a, double b) { return a + Cnt(b); }
inline Cnt operator-(Cnt a, double b) { return a - Cnt(b); }
inline Cnt operator*(Cnt a, double b) { return a * Cnt(b); }
inline Cnt operator/(Cnt a, double b) { return a / Cnt(b); }
inline Cnt t_abs(Cnt a) { return Cnt(fabs(a.x)); }
inline Cnt t_max(Cnt a, Cnt b) { return Cnt(fmax(a.x, b.x)); }
inline Cnt t_min(Cnt a, Cnt b) { return Cnt(fmin(a.x, b.x)); }
inline Cnt t_sqrt(Cnt a) { Cnt::n++; return Cnt(sqrt(a.x)); }
inline Cnt t_pow(Cnt a, Cnt b) { Cnt::n++; return Cnt(pow(a.x, b.x)); }
inline Cnt t_fmod(Cnt a, Cnt b) { Cnt::n++; return Cnt(fmod(a.x, b.x)); }
inline Cnt t_cos(Cnt a) { Cnt::n++; return Cnt(cos(a.x)); }
inline Cnt t_fma(Cnt a, Cnt b, Cnt c) { Cnt::n += 2; return Cnt(__builtin_fma(a.x, b.x, c.x)); }
inline Cnt t_rint(Cnt a) { return Cnt(__builtin_rint(a.x)); }
inline Cnt t_rcp(Cnt a) { Cnt::n++; return Cnt(1.0 / a.x); }
inline bool t_finite(Cnt a) { return a.x - a.x == 0.0; }
inline double t_val(Cnt a) { return a.x; }
}  // namespace mbrl

using namespace mbrl;

template <class M, class T>
struct Host {
    T col[Layout<M>::NMEM > 0 ? Layout<M>::NMEM : 1];
    Lean<M, T, LaneMem<T, 1>> S;
    void load(const double *q, const double *v, const double *c) {
        using LY = Layout<M>;
        S.mem.p = col;
        for (int j = 0; j < M::NV; j++) { col[LY::OQ + j] = T(q[j]); col[LY::OV + j] = T(v[j]); }
        for (int u = 0; u < M::NU; u++) col[LY::OC + u] = T(c[u]);
        S.status = 0;
    }
};

template <class M>
static int fwd(const double *q, const double *v, const double *c, double *qacc, double *fsm, double *fcon,
               int *iters, int *nactive) {
    Host<M, double> h;
    h.load(q, v, c);
    h.S.forward();
    for (int j = 0; j < M::NV; j++) { qacc[j] = h.S.qacc[j]; fsm[j] = h.S.fsm[j]; fcon[j] = h.S.fcon[j]; }
    *iters = h.S.newton_iters;
    int na = 0;
    for (int c = 0; c < 16; c++) na += h.S.slot1_of(c) ? 1 : 0;
    *nactive = na;
    return (int)h.S.status;
}

template <class M, class T>
static void stepN(int frame_skip, int64_t N, const double *q, const double *v, const double *c, double *qn,
                  double *vn, double *r, uint32_t *st) {
    using LY = Layout<M>;
    for (int64_t i = 0; i < N; i++) {
        Host<M, T> h;
        h.load(q + i * M::NV, v + i * M::NV, c + i * M::NU);
        T rew = h.S.env_step(frame_skip > 0 ? frame_skip : M::FRAME_SKIP);
        h.S.check_finite();
        for (int j = 0; j < M::NV; j++) { qn[i * M::NV + j] = t_val(h.col[LY::OQ + j]); vn[i * M::NV + j] = t_val(h.col[LY::OV + j]); }
        r[i] = t_val(rew);
        st[i] = h.S.status;
    }
}

#define DISPATCH(id, CALL)                                             \
    switch (id) {                                                      \
    case 0: { using M = MdlInvertedPendulum; CALL; } break;            \
    case 1: { using M = MdlInvertedDoublePendulum; CALL; } break;      \
    case 2: { using M = MdlSwimmer; CALL; } break;                     \
    case 3: { using M = MdlHopper; CALL; } break;                      \
    case 4: { using M = MdlWalker2d; CALL; } break;                    \
    case 5: { using M = MdlHalfCheetah; CALL; } break;                 \
    default: return -1;                                                \
    }

extern "C" {
// one dynamics evaluation (mj_forward); nactive = floor contacts that own a slot
int twin_forward(int id, const double *q, const double *v, const double *c, double *qacc, double *fsm,
                 double *fcon, int *iters, int *nactive) {
    int rc = 0;
    DISPATCH(id, rc = fwd<M>(q, v, c, qacc, fsm, fcon, iters, nactive));
    return rc;
}
int twin_step(int id, int frame_skip, int64_t N, const double *q, const double *v, const double *c, double *qn,
              double *vn, double *r, uint32_t *st) {
    DISPATCH(id, (stepN<M, double>(frame_skip, N, q, v, c, qn, vn, r, st)));
    return 0;
}
// flops of N env steps (returns total counted operations)
long long twin_count_flops(int id, int frame_skip, int64_t N, const double *q, const double *v, const double *c,
                           double *qn, double *vn, double *r, uint32_t *st) {
    Cnt::n = 0;
    DISPATCH(id, (stepN<M, Cnt>(frame_skip, N, q, v, c, qn, vn, r, st)));
    return Cnt::n;
}
}
