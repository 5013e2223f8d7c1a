// mbrl_hip.hip -- HIP kernels (gfx950) + C ABI (include/mbrl.h) for the FD linearisation path.
//
// Work decomposition (MI355X, wave64):
//   * one LANE per rollout (= one perturbed copy of one unit); a unit's 1+nu+nv+nq rollouts sit
//     in consecutive lanes, so the lanes of a wave are perturbations of at most a few nominal
//     states and follow the same contact / Newton control flow;
//   * a 256-thread workgroup owns UPB = 256 / R whole units; rollout results meet in LDS and the
//     workgroup writes that block of Jacobians with fully coalesced stores (the outputs of
//     consecutive units are contiguous in every output array);
//   * the path is fp64-VALU bound (~1 MFLOP per HalfCheetah unit vs 4 KB of HBM traffic); the per-lane
//     simulator (planar_lean.hpp) keeps its hot working set in registers and its cold / lane-indexed
//     state (q, v, ctrl, active limit and contact rows) in a lane-interleaved LDS column, which the
//     result staging buffer re-uses once the rollouts are done.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/mbrl.h"

#ifdef MBRL_PHASE_TIMING   // profiling build only (tools/phase_timing.py): shader-clock totals per solver phase
// Light marks: the previous stamp lives in a scalar register pair, the per-wave totals in LDS (fire-and-forget
// ds_add); phase 10 is a back-to-back pair of marks, i.e. the cost of one mark, for calibration.
__device__ unsigned long long g_phase_acc[16];
__shared__ unsigned long long s_phase_acc[4][16];
__device__ __forceinline__ void mbrl_phase_mark(int k, unsigned long long &last) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    const unsigned long long dt = now - last;
    last = now;
    if (k != 15 && (threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
        atomicAdd(&s_phase_acc[threadIdx.x >> 6][k], dt);
}
#define MBRL_PHASE(k) mbrl_phase_mark(k, mbrl_phase_last)
#define MBRL_PHASE_MEMBER unsigned long long mbrl_phase_last;
#endif

#include "planar_lean.hpp"

namespace mbrl {

#ifndef MBRL_BLOCK
#define MBRL_BLOCK 128
#endif
constexpr int kBlock = MBRL_BLOCK;

constexpr int MODE_FORWARD = 0, MODE_CENTRAL = 1, MODE_STEP = 2;

#ifdef MBRL_WAVE_TIMING   // profiling build only (tools/wave_timing.py): per-wave shader-clock stamps
__device__ unsigned long long *g_wave_timing = nullptr;
#define MBRL_STAMP(k) do { if (g_wave_timing && (threadIdx.x & 63) == 0) \
    g_wave_timing[(size_t(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6)) * 4 + (k)] = clock64(); } while (0)
#else
#define MBRL_STAMP(k)
#endif

// ONE kernel per model for env.step, forward-difference and central-difference linearisation: the
// mode is a run-time argument, so the nominal rollout of an FD launch and a plain step launch execute
// the same machine code and agree bit for bit (the reference asserts exactly that, backward.py:371,374).
//   mode FORWARD: R = 1 + NP rollouts per unit (NP = nu + nv + nq), CENTRAL: R = 1 + 2 NP, STEP: R = 1.
// Column order of calculate_gradients: ctrl, then qvel, then qpos (backward.py:389,409,433).
// STEP_NAME only gives plain-step launches their own symbol in profiler output (the body does not use it,
// so both instantiations are compiled from identical IR; tests/test_gpu_parity.py checks the bits).
template <class M, bool STEP_NAME>
__global__ __launch_bounds__(kBlock) void rollout_kernel(int64_t N, int frame_skip, int nsteps, double eps, int mode,
                                                          const double *__restrict__ qpos,
                                                          const double *__restrict__ qvel,
                                                          const double *__restrict__ ctrl,
                                                          const double *__restrict__ ctrl_pert, double *__restrict__ dsds,
                                                          double *__restrict__ dsda, double *__restrict__ drds,
                                                          double *__restrict__ drda, double *__restrict__ s_next,
                                                          double *__restrict__ qpos_next, double *__restrict__ qvel_next,
                                                          double *__restrict__ reward, uint32_t *__restrict__ status) {
    using LY = Layout<M>;
    constexpr int NV = M::NV, NU = M::NU, NS = 2 * NV, ROW = NS + 1, NP = NU + 2 * NV;
    static_assert(LY::NMEM >= ROW, "the staging buffer re-uses the lane columns");
    static_assert(1 + 2 * NP <= kBlock, "a unit's rollouts must fit one workgroup");
    __shared__ double lds[LY::NMEM * kBlock];   // lane columns during the rollouts, then res[UPB*R][ROW]
    __shared__ unsigned stat[kBlock];
    double *res = lds;

    const bool central = mode == MODE_CENTRAL;
    const int R = mode == MODE_STEP ? 1 : (central ? 1 + 2 * NP : 1 + NP);   // rollouts per unit
    const int UPB = kBlock / R;                                              // units per workgroup
    const int t = threadIdx.x;
    const int u = t / R, p = t - u * R;
    const int64_t unit0 = int64_t(blockIdx.x) * UPB;
    const int64_t unit = unit0 + u;
    stat[t] = 0;
    __syncthreads();
    MBRL_STAMP(0);

    const bool live = u < UPB && unit < N;
    double o[ROW];
    unsigned st = 0;
    if (live) {
        Lean<M, double, LaneMem<double, kBlock>> S;
        S.status = 0;
        S.mem.p = lds + t;
        const double *qp = qpos + unit * NV, *vp = qvel + unit * NV, *cp = ctrl + unit * NU;
        int idx = -1;
        double d = eps;
        if (p > 0) {
            idx = p - 1;
            if (central) { d = (idx & 1) ? -eps : eps; idx >>= 1; }
        }
        // closed-loop linearisation (backward.py:418-419,460-461): a state-perturbed rollout runs under the
        // policy's action AT the perturbed state, row (unit, column[, sign]) of ctrl_pert; columns in qvel, qpos order
        if (ctrl_pert && idx >= NU) {
            const int64_t col = unit * (2 * NV) + (idx - NU);
            cp = ctrl_pert + (central ? 2 * col + (d < 0.0 ? 1 : 0) : col) * NU;
        }
#pragma unroll
        for (int k = 0; k < NU; k++) S.mem.st(LY::OC + k, idx == k ? cp[k] + d : cp[k]);
#pragma unroll
        for (int j = 0; j < NV; j++) S.mem.st(LY::OV + j, idx == NU + j ? vp[j] + d : vp[j]);
#pragma unroll
        for (int j = 0; j < NV; j++) S.mem.st(LY::OQ + j, idx == NU + NV + j ? qp[j] + d : qp[j]);
        double r = 0.0;
#ifdef MBRL_PHASE_TIMING
        {
            unsigned long long &mbrl_phase_last = S.mbrl_phase_last;
            for (int k = 0; k < 16; k++) s_phase_acc[threadIdx.x >> 6][k] = 0;
            MBRL_PHASE(15);
            MBRL_PHASE(10);
        }
#endif
        for (int k = 0; k < nsteps; k++) r = S.env_step(frame_skip);
#ifdef MBRL_PHASE_TIMING
        {
            unsigned long long &mbrl_phase_last = S.mbrl_phase_last;
            MBRL_PHASE(9);
            MBRL_PHASE(10);
            if ((threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                for (int k = 0; k < 16; k++) atomicAdd(&g_phase_acc[k], s_phase_acc[threadIdx.x >> 6][k]);
        }
#endif
        S.check_finite();
#pragma unroll
        for (int j = 0; j < NV; j++) { o[j] = S.mem.ld(LY::OQ + j); o[NV + j] = S.mem.ld(LY::OV + j); }
        o[NS] = r;
        st = S.status;
    }
    MBRL_STAMP(1);
    __syncthreads();                      // every lane is done with its column: re-use LDS as res
    MBRL_STAMP(2);
    if (live) {
        double *ro = res + (u * R + p) * ROW;
#pragma unroll
        for (int j = 0; j < ROW; j++) ro[j] = o[j];
        if (st) atomicOr(&stat[u], st);
    }
    __syncthreads();

    int64_t nun = N - unit0;
    if (nun > UPB) nun = UPB;
    const int nu_blk = int(nun);
    const double den = central ? 2.0 * eps : eps;
    // perturbation slot of input column `in` (0..NP-1 in ctrl,qvel,qpos order)
    auto plus = [central](int in) { return central ? 1 + 2 * in : 1 + in; };
    auto minus = [central](int in) { return central ? 2 + 2 * in : 0; };

    if (dsds) {  // [unit][o][col], col < NV: qpos input, else qvel input
        double *base = dsds + unit0 * NS * NS;
        for (int e = t; e < nu_blk * NS * NS; e += kBlock) {
            const int uu = e / (NS * NS), rem = e - uu * NS * NS, o = rem / NS, col = rem - o * NS;
            const int in = col < NV ? NU + NV + col : NU + (col - NV);
            const double *ru = res + uu * R * ROW;
            base[e] = (ru[plus(in) * ROW + o] - ru[minus(in) * ROW + o]) / den;
        }
    }
    if (dsda) {
        double *base = dsda + unit0 * NS * NU;
        for (int e = t; e < nu_blk * NS * NU; e += kBlock) {
            const int uu = e / (NS * NU), rem = e - uu * NS * NU, o = rem / NU, k = rem - o * NU;
            const double *ru = res + uu * R * ROW;
            base[e] = (ru[plus(k) * ROW + o] - ru[minus(k) * ROW + o]) / den;
        }
    }
    if (drds) {
        double *base = drds + unit0 * NS;
        for (int e = t; e < nu_blk * NS; e += kBlock) {
            const int uu = e / NS, col = e - uu * NS;
            const int in = col < NV ? NU + NV + col : NU + (col - NV);
            const double *ru = res + uu * R * ROW;
            base[e] = (ru[plus(in) * ROW + NS] - ru[minus(in) * ROW + NS]) / den;
        }
    }
    if (drda) {
        double *base = drda + unit0 * NU;
        for (int e = t; e < nu_blk * NU; e += kBlock) {
            const int uu = e / NU, k = e - uu * NU;
            const double *ru = res + uu * R * ROW;
            base[e] = (ru[plus(k) * ROW + NS] - ru[minus(k) * ROW + NS]) / den;
        }
    }
    if (s_next) {
        double *base = s_next + unit0 * NS;
        for (int e = t; e < nu_blk * NS; e += kBlock) {
            const int uu = e / NS, o = e - uu * NS;
            base[e] = res[uu * R * ROW + o];
        }
    }
    if (qpos_next) {   // split form of s_next (mbrl_step)
        double *bq = qpos_next + unit0 * NV, *bv = qvel_next + unit0 * NV;
        for (int e = t; e < nu_blk * NV; e += kBlock) {
            const int uu = e / NV, o = e - uu * NV;
            bq[e] = res[uu * R * ROW + o];
            bv[e] = res[uu * R * ROW + NV + o];
        }
    }
    if (t < nu_blk) {
        if (reward) reward[unit0 + t] = res[t * R * ROW + NS];
        if (status) status[unit0 + t] = stat[t];
    }
    MBRL_STAMP(3);
}

// Backward recurrence of the op chain (mujoco.py:43-104): one thread per (episode, output lane).
// n <= 18, nu <= 6: each episode is walked by one 32-lane group; lane l owns g[l].
__global__ __launch_bounds__(kBlock) void vjp_scan_kernel(int64_t B, int H, int n, int nu, int gw,
                                                           const double *__restrict__ dsds,
                                                           const double *__restrict__ dsda,
                                                           const double *__restrict__ drds,
                                                           const double *__restrict__ drda,
                                                           const double *__restrict__ c, double *__restrict__ da,
                                                           double *__restrict__ ds0) {
    __shared__ double gs[kBlock / 32][32];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int64_t b = int64_t(blockIdx.x) * (kBlock / 32) + grp;
    const bool live = b < B;
    double g = 0.0;  // g_{t+1}[l]
    // 'prioritise': autograd runs episodes last-to-first, the running sum carries over (mujoco.py:65)
    double S = live ? double(B - 1 - b) * (0.5 * double(H) * double(H + 1)) : 0.0;
    for (int t = H - 1; t >= 0; t--) {
        gs[grp][l] = g;
        __syncthreads();
        if (live) {
            const int64_t k = b * H + t;
            const double ct = c[k];
            const double W = double(H - t);
            double wa_r = 1.0, wa_d = 1.0, ws_r = 1.0;  // action weights (reward / dynamics op), state weight (reward op)
            if (gw == 1) { wa_r = 1.0 / W; wa_d = 1.0 / W; }
            else if (gw == 2) { S += W; ws_r = W - 1.0; wa_r = W / S; wa_d = 1.0 / S; }
            if (l < nu) {
                double acc = 0.0;
                for (int o = 0; o < n; o++) acc += gs[grp][o] * (wa_d * dsda[(k * n + o) * nu + l]);
                da[k * nu + l] = ct * (wa_r * drda[k * nu + l]) + acc;
            }
            if (l < n) {
                double acc = 0.0;
                for (int o = 0; o < n; o++) acc += gs[grp][o] * dsds[(k * n + o) * n + l];
                g = ct * (ws_r * drds[k * n + l]) + acc;
            }
        }
        __syncthreads();
    }
    if (live && ds0 && l < n) ds0[b * n + l] = g;
}

}  // namespace mbrl

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace mbrl;

struct mbrl_model {
    int env_id, frame_skip, device, nq, nv, nu;
    double timestep;
};

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *a = "", long long b = 0) {
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}

#define MBRL_DISPATCH(id, CALL)                                        \
    switch (id) {                                                      \
    case 0: { using M = MdlInvertedPendulum; CALL; } break;            \
    case 1: { using M = MdlInvertedDoublePendulum; CALL; } break;      \
    case 2: { using M = MdlSwimmer; CALL; } break;                     \
    case 3: { using M = MdlHopper; CALL; } break;                      \
    case 4: { using M = MdlWalker2d; CALL; } break;                    \
    case 5: { using M = MdlHalfCheetah; CALL; } break;                 \
    default: break;                                                    \
    }

static const char *kEnvNames[] = {"InvertedPendulumEnv", "InvertedDoublePendulumEnv", "SwimmerEnv",
                                  "HopperEnv", "Walker2dEnv", "HalfCheetahEnv"};

template <class M>
static void fill_dims(mbrl_model *m) {
    m->nq = M::NV; m->nv = M::NV; m->nu = M::NU; m->timestep = M::DT;
    if (m->frame_skip <= 0) m->frame_skip = M::FRAME_SKIP;
}

static int check_launch(const mbrl_model *m, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-10, "%s: launch failed (hip error %lld)", what, (long long)e);
    (void)m;
    return 0;
}

template <class M>
static void launch_rollouts(const mbrl_model *m, int64_t N, int mode, const double *qpos, const double *qvel,
                            const double *ctrl, const double *ctrl_pert, double eps, int nsteps, double *dsds, double *dsda, double *drds,
                            double *drda, double *s_next, double *qpos_next, double *qvel_next, double *reward,
                            uint32_t *status, hipStream_t st) {
    constexpr int NP = M::NU + 2 * M::NV;
    const int R = mode == MODE_STEP ? 1 : (mode == MODE_CENTRAL ? 1 + 2 * NP : 1 + NP);
    const int upb = kBlock / R;
    const int64_t blocks = (N + upb - 1) / upb;
    if (mode == MODE_STEP)
        hipLaunchKernelGGL((rollout_kernel<M, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, N, m->frame_skip, nsteps,
                           eps, mode, qpos, qvel, ctrl, ctrl_pert, dsds, dsda, drds, drda, s_next, qpos_next, qvel_next, reward,
                           status);
    else
        hipLaunchKernelGGL((rollout_kernel<M, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, N, m->frame_skip, nsteps,
                           eps, mode, qpos, qvel, ctrl, ctrl_pert, dsds, dsda, drds, drda, s_next, qpos_next, qvel_next, reward,
                           status);
}

#ifdef MBRL_PHASE_TIMING
extern "C" int mbrl_debug_phase_read(unsigned long long *host16, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_phase_acc), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_acc), z, sizeof z); }
    return rc;
}
#endif

#ifdef MBRL_WAVE_TIMING
extern "C" int mbrl_debug_set_wave_timing(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_timing), &buf, sizeof buf);
}
#endif

extern "C" {

int mbrl_abi_version(void) { return MBRL_ABI_VERSION; }

const char *mbrl_last_error(void) { return g_err; }

int mbrl_model_create(const char *env_name, int frame_skip, int device, mbrl_model **out) {
    if (!env_name || !out) return fail(-1, "mbrl_model_create: null argument");
    int id = -1;
    for (int i = 0; i < 6; i++)
        if (!strcmp(env_name, kEnvNames[i])) id = i;
    if (id < 0) return fail(-2, "mbrl_model_create: unknown env '%s'", env_name);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(-3, "mbrl_model_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(-4, "mbrl_model_create: bad device ordinal %s%lld", "", device);
    mbrl_model *m = new (std::nothrow) mbrl_model();
    if (!m) return fail(-5, "mbrl_model_create: out of host memory");
    m->env_id = id; m->frame_skip = frame_skip; m->device = device;
    MBRL_DISPATCH(id, fill_dims<M>(m));
    *out = m;
    return 0;
}

void mbrl_model_destroy(mbrl_model *m) { delete m; }

int mbrl_model_dims(const mbrl_model *m, int *nq, int *nv, int *nu, double *dt) {
    if (!m) return fail(-1, "mbrl_model_dims: null model");
    if (nq) *nq = m->nq;
    if (nv) *nv = m->nv;
    if (nu) *nu = m->nu;
    if (dt) *dt = m->timestep * m->frame_skip;
    return 0;
}

int mbrl_step(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
              double *qpos_next, double *qvel_next, double *reward, uint32_t *status, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_step: null model");
    if (N < 0) return fail(-6, "mbrl_step: negative N");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl || !qpos_next || !qvel_next) return fail(-1, "mbrl_step: null buffer");
    if (N > (int64_t(1) << 31)) return fail(-6, "mbrl_step: N too large for one launch");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_step: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    MBRL_DISPATCH(m->env_id, (launch_rollouts<M>(m, N, MODE_STEP, qpos, qvel, ctrl, nullptr, 0.0, 1, nullptr, nullptr, nullptr,
                                                 nullptr, nullptr, qpos_next, qvel_next, reward, status, st)));
    return check_launch(m, "mbrl_step");
}

int mbrl_fd_jacobians(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                      double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                      double *s_next, double *reward, uint32_t *status, void *hip_stream) {
    return mbrl_fd_jacobians_closed_loop(m, N, qpos, qvel, ctrl, nullptr, eps, fd_mode, nsteps, dsds, dsda, drds, drda,
                                         s_next, reward, status, hip_stream);
}

int mbrl_fd_jacobians_closed_loop(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel,
                                  const double *ctrl, const double *ctrl_pert, double eps, int fd_mode, int nsteps,
                                  double *dsds, double *dsda, double *drds, double *drda, double *s_next,
                                  double *reward, uint32_t *status, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_fd_jacobians: null model");
    if (N < 0) return fail(-6, "mbrl_fd_jacobians: negative N");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl) return fail(-1, "mbrl_fd_jacobians: null input buffer");
    if (!(eps > 0)) return fail(-7, "mbrl_fd_jacobians: eps must be > 0");
    if (fd_mode != MBRL_FD_FORWARD && fd_mode != MBRL_FD_CENTRAL) return fail(-7, "mbrl_fd_jacobians: bad fd_mode");
    if (nsteps <= 0) nsteps = 1;
    if (N > (int64_t(1) << 31)) return fail(-6, "mbrl_fd_jacobians: N too large for one launch");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_fd_jacobians: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    MBRL_DISPATCH(m->env_id, (launch_rollouts<M>(m, N, fd_mode == MBRL_FD_CENTRAL ? MODE_CENTRAL : MODE_FORWARD, qpos, qvel,
                                                 ctrl, ctrl_pert, eps, nsteps, dsds, dsda, drds, drda, s_next, nullptr, nullptr,
                                                 reward, status, st)));
    return check_launch(m, "mbrl_fd_jacobians");
}

int mbrl_vjp_scan(const mbrl_model *m, int64_t B, int H, int grad_weights, const double *dsds, const double *dsda,
                  const double *drds, const double *drda, const double *c, double *da, double *ds0, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_vjp_scan: null model");
    if (B < 0 || H < 0) return fail(-6, "mbrl_vjp_scan: negative size");
    if (B == 0 || H == 0) return 0;
    if (!dsds || !dsda || !drds || !drda || !c || !da) return fail(-1, "mbrl_vjp_scan: null buffer");
    if (grad_weights < 0 || grad_weights > 2) return fail(-7, "mbrl_vjp_scan: bad grad_weights");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_vjp_scan: hipSetDevice failed");
    const int per = kBlock / 32;
    const unsigned blocks = (unsigned)((B + per - 1) / per);
    hipLaunchKernelGGL(vjp_scan_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)hip_stream, B, H, m->nq + m->nv,
                       m->nu, grad_weights, dsds, dsda, drds, drda, c, da, ds0);
    return check_launch(m, "mbrl_vjp_scan");
}

}  // extern "C"
