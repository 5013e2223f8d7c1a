// mbrl_hip.hip -- HIP kernels (gfx950) + C ABI (include/mbrl.h) for the FD linearisation path.
//
// Work decomposition (MI355X, wave64):
//   * one LANE per rollout (= one perturbed copy of one unit); the path is fp64-VALU bound (~1 MFLOP per HalfCheetah
//     unit vs 4 KB of HBM traffic), the per-lane simulator (planar_lean.hpp) keeps its hot working set in registers
//     (one wave per SIMD) and its lane-indexed state (q, v, ctrl, active limit and contact rows) in a
//     lane-interleaved LDS column;
//   * a workgroup is ONE wave, so waves never wait for each other;
//   * single launch (mbrl_fd_jacobians): a unit's 1+nu+nv+nq rollouts sit in consecutive lanes, two consecutive waves
//     share one unit to keep the lanes full (WaveMap), every perturbed lane differences against the nominal lane of
//     its own wave with a lane shuffle and writes its Jacobian column straight from registers;
//   * two launches (mbrl_fd_plan + mbrl_fd_jacobians_planned): the nominal rollouts run first, one lane each, and
//     leave their results and a control-flow key per unit in the caller's plan buffer; the units are radix-sorted by
//     key; the second launch runs only the perturbed rollouts, in that order, every lane busy, waves of like units.
//   All of these run the same rollout_kernel body, so their results agree bit for bit.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>

#include "../../include/mbrl.h"

#ifdef MBRL_PHASE_TIMING   // profiling build only (tools/phase_timing.py): shader-clock totals per solver phase
// Light marks: the previous stamp lives in a scalar register pair, the per-wave totals in LDS (fire-and-forget
// ds_add); phase 10 is a back-to-back pair of marks, i.e. the cost of one mark, for calibration.
__device__ unsigned long long g_phase_acc[32];       // [0..15] shader clocks per phase, [16..31] times the phase was charged
__shared__ unsigned long long s_phase_acc[4][32];
__device__ __forceinline__ void mbrl_phase_mark(int k, unsigned long long &last) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    const unsigned long long dt = now - last;
    last = now;
    if (k != 15 && (threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
        atomicAdd(&s_phase_acc[threadIdx.x >> 6][k], dt);
        atomicAdd(&s_phase_acc[threadIdx.x >> 6][16 + k], 1ull);
    }
}
#define MBRL_PHASE(k) mbrl_phase_mark(k, mbrl_phase_last)
#define MBRL_PHASE_MEMBER unsigned long long mbrl_phase_last;
#endif

#include <rocprim/device/device_radix_sort.hpp>

#include "planar_lean.hpp"

namespace mbrl {

#ifndef MBRL_BLOCK
#define MBRL_BLOCK 64
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
//
// Workgroup = ONE wave.  Waves that share a workgroup wait for each other at the end (their LDS is released
// together) and the rollouts of different units take different numbers of solver iterations, so a workgroup of
// 2 (4) waves idled 8 % (12 %) of its time; single-wave workgroups finish independently, four resident per CU.
// To keep the lanes full although R does not divide 64, two consecutive waves (a "pair") share one unit: wave 0
// runs its nominal rollout and the first perturbations, wave 1 runs the nominal rollout AGAIN (1 lane) and the
// remaining perturbations; each wave differences against its own copy of the nominal (the two copies execute the
// same instructions on the same inputs), so the waves never exchange data.  HalfCheetah, forward mode: 5 units
// per pair, 126 of 128 lanes busy.
struct WaveMap {
    int R, A, L, nsp, A1, upp;   // rollouts/unit; full units in wave 0; lanes of the shared unit in wave 0 (0: none);
                                 // its perturbations in wave 1; full units in wave 1; units per pair
    __host__ __device__ WaveMap(int R_, bool central) : R(R_) {
        A = 64 / R;
        L = 64 - A * R;
        if (central && L > 0 && ((L - 1) & 1)) L -= 1;   // a +eps / -eps couple stays in one wave
        if (L < (central ? 3 : 2) || R == 1) L = 0;
        if (L) { nsp = R - L; A1 = (64 - 1 - nsp) / R; upp = A + 1 + A1; }
        else { nsp = 0; A1 = A; upp = 2 * A; }
    }
    // lane t of wave `half` -> unit u of the pair, rollout p of that unit, lane of the nominal it differences against
    __host__ __device__ bool lane(int half, int t, int &u, int &p, int &nom, bool &primary) const {
        primary = true;
        if (half == 0) {
            if (t < A * R) { u = t / R; p = t - u * R; nom = u * R; return true; }
            if (t < A * R + L) { u = A; p = t - A * R; nom = A * R; return true; }
            return false;
        }
        if (L) {
            if (t <= nsp) { u = A; p = t ? L + t - 1 : 0; nom = 0; primary = false; return true; }
            const int k = (t - 1 - nsp) / R;
            u = A + 1 + k; p = (t - 1 - nsp) - k * R; nom = 1 + nsp + k * R;
            return k < A1;
        }
        const int k = t / R;
        u = A + k; p = t - k * R; nom = k * R;
        return k < A1;
    }
};

// ---- launch plan (mbrl_fd_plan): the nominal rollouts, their control-flow keys and the sorted unit order ----
// Segments of the caller's plan buffer, each 256-byte aligned:
//   header | key_in u64[N] | key_out u64[N] | idx_in u32[N] | perm u32[N] | nom_s f64[N * ns] | nom_r f64[N] | sort scratch
// The header names the batch the plan was made for (PlanHeader); a planned launch with other arguments flags every
// unit MBRL_ST_BADPLAN and runs nothing.
// The kernel gets the buffer as ONE pointer and derives the segments where it uses them (kernel arguments stay live in
// scalar registers for the whole rollout, and those are short).
// Waves are dispatched in grid order: walking the sorted units from the LARGEST key down starts the longest rollouts
// first and leaves the short ones to fill the tail of the launch.
#ifdef MBRL_PLAN_ASCENDING
#define MBRL_PLAN_SLOT(slot) (slot)
#else
#define MBRL_PLAN_SLOT(slot) (N - 1 - (slot))
#endif
constexpr int kKeyBits = 62;   // Lean::key_value: floor-contact union, solver passes (per mj_step; long rollouts: total first), limit / pair union
constexpr int PLAN_NONE = 0, PLAN_WRITE = 1, PLAN_READ = 2, PLAN_RECORD = 3;   // RECORD: keys only (mbrl_step_record)
constexpr unsigned ST_BADPLAN = MBRL_ST_BADPLAN;
struct PlanHeader {
    unsigned long long magic;   // kPlanMagic
    long long units;            // N
    int nsteps, frame_skip, env_id, pad;
};
constexpr unsigned long long kPlanMagic = 0x4d42524c504c4e31ull;   // "MBRLPLN1"
struct PlanOffsets {
    size_t key_in, idx_in, key_out, perm, nom_s, nom_r, end;
    __host__ __device__ static size_t up(size_t x) { return (x + 255) & ~size_t(255); }
    __host__ __device__ PlanOffsets(int64_t N, int ns) {
        const size_t w = up(sizeof(uint32_t) * size_t(N)), w2 = up(sizeof(uint64_t) * size_t(N));
        key_in = 256; key_out = key_in + w2; idx_in = key_out + w2; perm = idx_in + w; nom_s = perm + w;
        nom_r = nom_s + up(sizeof(double) * size_t(N) * size_t(ns));
        end = nom_r + up(sizeof(double) * size_t(N));
    }
};
struct PlanLayout {
    size_t sort_tmp, sort_bytes, total;
};

// One lane's rollout: nsteps env steps from (qpos, qvel, ctrl)[unit] with input `idx` perturbed by d (idx < 0: nominal).
// -> o = (s', r), the status bits and the control-flow key.  KCAP = contact slots, LSTRIDE = lanes per LDS column
// (`col` = this lane's column).  Force-inlined: the kernels stay single functions.
template <class M, int KCAP, int LSTRIDE>
__device__ __forceinline__ void lane_rollout(double *col, const int64_t unit, const int idx, const double d, const bool central,
                                             const bool minus, int frame_skip, int nsteps, const double *__restrict__ qpos,
                                             const double *__restrict__ qvel, const double *__restrict__ ctrl,
                                             const double *__restrict__ ctrl_pert, double *o, unsigned &st, uint64_t &key) {
    using LY = Layout<M, KCAP>;
    constexpr int NV = M::NV, NU = M::NU, NS = 2 * NV;
    Lean<M, double, LaneMem<double, LSTRIDE>, KCAP> S;
    S.status = 0;
    S.key_init(frame_skip * nsteps);
    S.mem.p = col;
    const double *qp = qpos + unit * NV, *vp = qvel + unit * NV, *cp = ctrl + unit * NU;
    // closed-loop linearisation (backward.py:418-419,460-461): a state-perturbed rollout runs under the
    // policy's action AT the perturbed state, row (unit, column[, sign]) of ctrl_pert; columns in qvel, qpos order
    if (ctrl_pert && idx >= NU) {
        const int64_t c = unit * (2 * NV) + (idx - NU);
        cp = ctrl_pert + (central ? 2 * c + (minus ? 1 : 0) : c) * NU;
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
        for (int k = 0; k < 32; k++) s_phase_acc[threadIdx.x >> 6][k] = 0;
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
            for (int k = 0; k < 32; k++) atomicAdd(&g_phase_acc[k], s_phase_acc[threadIdx.x >> 6][k]);
    }
#endif
    S.check_finite();
#pragma unroll
    for (int j = 0; j < NV; j++) { o[j] = S.mem.ld(LY::OQ + j); o[NV + j] = S.mem.ld(LY::OV + j); }
    o[NS] = r;
    st = S.status;
    key = S.key_value(frame_skip * nsteps);
}

// What a wave does with its lanes' results: every perturbed lane differences its own (s', r) against the reference lane
// of its unit in THIS wave (forward: the nominal, lane `nom`, or the plan's nominal; central: its -eps partner, the next
// lane) and writes its Jacobian column; the nominal lane writes the nominal outputs.
template <class M>
__device__ __forceinline__ void wave_outputs(const double *o, const int t, const int64_t unit, const int p, const int idx,
                                             const bool minus, const int nom, const bool primary, const bool live,
                                             const bool planned, const bool central, int64_t N, double eps,
                                             double *__restrict__ dsds, double *__restrict__ dsda, double *__restrict__ drds,
                                             double *__restrict__ drda, double *__restrict__ s_next,
                                             double *__restrict__ qpos_next, double *__restrict__ qvel_next,
                                             double *__restrict__ reward, const char *__restrict__ plan) {
    constexpr int NV = M::NV, NU = M::NU, NS = 2 * NV, ROW = NS + 1;
    // ---- Jacobian columns: every perturbed lane differences its own result against the reference lane of its
    // unit in THIS wave (forward: the nominal; central: its -eps partner, the next lane) and writes its column.
    // Lanes of one unit hold consecutive columns, so the stores of a wave fall in runs of nu / nv / nq doubles.
    const double den = central ? 2.0 * eps : eps;
    const int ref = central ? (t + 1) & 63 : nom;
    const bool writes = live && p > 0 && !minus;
    const int col = idx < NU ? idx : (idx < NU + NV ? NV + (idx - NU) : idx - NU - NV);   // column in its output array
    double *js = nullptr, *jr = nullptr;     // this lane's column of the state / reward Jacobian
    int stride = 0;
    if (writes) {
        if (idx < NU) { if (dsda) js = dsda + unit * (NS * NU) + col; if (drda) jr = drda + unit * NU + col; stride = NU; }
        else { if (dsds) js = dsds + unit * (NS * NS) + col; if (drds) jr = drds + unit * NS + col; stride = NS; }
    }
    const double *nom_s = nullptr, *nom_r = nullptr;
    if (planned && !central) {
        const PlanOffsets po(N, NS);
        nom_s = reinterpret_cast<const double *>(plan + po.nom_s) + unit * NS;
        nom_r = reinterpret_cast<const double *>(plan + po.nom_r) + unit;
    }
#pragma unroll
    for (int j = 0; j < ROW; j++) {
        double base = __shfl(o[j], ref, 64);
        if (nom_s) base = j < NS ? nom_s[j] : *nom_r;
        const double dv = (o[j] - base) / den;
        if (j < NS) { if (js) js[j * stride] = dv; }
        else if (jr) *jr = dv;
    }
    // ---- the nominal rollout (the wave that owns the unit's first lanes writes it) ----
    if (live && p == 0 && primary) {
        if (s_next) {
#pragma unroll
            for (int j = 0; j < NS; j++) s_next[unit * NS + j] = o[j];
        }
        if (qpos_next) {
#pragma unroll
            for (int j = 0; j < NV; j++) { qpos_next[unit * NV + j] = o[j]; qvel_next[unit * NV + j] = o[NV + j]; }
        }
        if (reward) reward[unit] = o[NS];
    }
}

// STEP_NAME only gives plain-step launches their own symbol in profiler output (the body does not use it,
// so both instantiations are compiled from identical IR; tests/test_gpu_parity.py checks the bits).
#ifdef MBRL_WAVES_PER_EU
#define MBRL_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(MBRL_WAVES_PER_EU)))
#else
#define MBRL_KERNEL_ATTR
#endif
constexpr int MODE_MASK = 3, MODE_NO_REDO = 8, MODE_FORCE_WIDE = 16;   // mode argument: MODE_* | flags (debug: mbrl_debug_set_flags)
// Units with more floor contacts than the 8 slots of the fast rollouts are re-run with one slot per floor candidate:
// either by the wave that found them, right away, or by a fix-up launch afterwards (overflow_kernel).  Measured
// (profiles/r02_experiments.md):
//   * Walker2d: in the wave, everywhere.  Walkers that lie flat turn up in most batches, and a planned launch starts
//     them first, so the second run hides under the rest of the launch (a fix-up launch costs one serial wave-time of a
//     flat walker, 2.65 ms, with the chip idle).
//   * HalfCheetah: in the wave for step / plan launches (STEP_NAME kernels: no extra launch after each of a rollout's
//     40 step launches, 1 ms per training epoch), a fix-up launch after the FD launches (never needed on trajectories;
//     the wide code inside the FD kernel cost the pipelined bench 1.7 %, the fix-up launch costs it 0.2 %).
//   * FD launches WITHOUT a status array (the ABI allows status == NULL) have nothing a fix-up launch could scan: they run
//     a third instantiation of the kernel, REDO_FD, that re-runs in the wave like Walker2d's.  Only callers that pass no
//     status pay its 1.7 %.
template <class M, bool STEP_NAME, bool REDO_FD = false> constexpr bool kRedoInWave = M::ENV_ID == 4 || STEP_NAME || REDO_FD;
template <class M> constexpr bool kNeedsRedo = (Layout<M>::NCAND > Layout<M>::K);
template <class M, bool STEP_NAME, bool REDO_FD = false>
__global__ __launch_bounds__(kBlock) MBRL_KERNEL_ATTR void rollout_kernel(int64_t N, int frame_skip, int nsteps, double eps, int mode_arg,
                                                          const double *__restrict__ qpos,
                                                          const double *__restrict__ qvel,
                                                          const double *__restrict__ ctrl,
                                                          const double *__restrict__ ctrl_pert, double *__restrict__ dsds,
                                                          double *__restrict__ dsda, double *__restrict__ drds,
                                                          double *__restrict__ drda, double *__restrict__ s_next,
                                                          double *__restrict__ qpos_next, double *__restrict__ qvel_next,
                                                          double *__restrict__ reward, uint32_t *__restrict__ status,
                                                          char *__restrict__ plan, int plan_mode) {
    using LY = Layout<M>;
    constexpr int NV = M::NV, NU = M::NU, NS = 2 * NV, ROW = NS + 1, NP = NU + 2 * NV;
    static_assert(kBlock == 64, "one wave per workgroup");
    static_assert(1 + 2 * NP <= kBlock, "a unit's rollouts must fit one wave");
    __shared__ double lds[LY::NMEM * kBlock];   // lane columns

    const int mode = mode_arg & MODE_MASK;
    const bool central = mode == MODE_CENTRAL;
    const WaveMap wm(mode == MODE_STEP ? 1 : (central ? 1 + 2 * NP : 1 + NP), central);
    const int t = threadIdx.x;
    int u, p, nom;
    bool primary, live;
    int64_t unit;
    const bool planned = plan_mode == PLAN_READ;
    if (planned) {
        const PlanHeader *hd = reinterpret_cast<const PlanHeader *>(plan);
        if (hd->magic != kPlanMagic || hd->units != N || hd->nsteps != nsteps || hd->frame_skip != frame_skip ||
            hd->env_id != M::ENV_ID) {
            const int64_t g = int64_t(blockIdx.x) * kBlock + t;     // the first N lanes flag one unit each
            if (status && g < N) atomicOr(&status[g], ST_BADPLAN);
            return;
        }
    }
    if (plan_mode == PLAN_WRITE && blockIdx.x == 0 && t == 0) {
        PlanHeader *hd = reinterpret_cast<PlanHeader *>(plan);
        hd->magic = kPlanMagic; hd->units = N; hd->nsteps = nsteps; hd->frame_skip = frame_skip; hd->env_id = M::ENV_ID;
        hd->pad = 0;
    }
    if (planned) {
        // planned launch (mbrl_fd_jacobians_planned): perturbed rollouts only, units in the planner's order, every
        // lane busy; the nominal results come from the plan (nom_s, nom_r).  2 NP is even, so a central couple
        // (+eps on the even lane, -eps on the next) never straddles two waves.
        const int64_t g = int64_t(blockIdx.x) * kBlock + t;
        const int64_t slot = central ? g / (2 * NP) : g / NP;
        p = 1 + int(g - slot * (central ? 2 * NP : NP));
        live = slot < N;
        unit = live ? int64_t(reinterpret_cast<const uint32_t *>(plan + PlanOffsets(N, NS).perm)[MBRL_PLAN_SLOT(slot)]) : 0;
        // a plan made for another batch holds arbitrary indices here: never follow one out of the arrays
        if (unit >= N) { live = false; unit = 0; }
        nom = 0; primary = true;
    } else {
        const bool mapped = wm.lane(int(blockIdx.x & 1u), t, u, p, nom, primary);
        unit = int64_t(blockIdx.x >> 1) * wm.upp + u;
        live = mapped && unit < N;
    }
    MBRL_STAMP(0);
    double o[ROW];
    unsigned st = 0;
    uint64_t key = 0;
    int idx = -1;            // perturbed input (ctrl, qvel, qpos order), -1: nominal
    bool minus = false;      // the -eps member of a central couple
    double d = eps;
    if (live && p > 0) {
        idx = p - 1;
        if (central) { minus = idx & 1; d = minus ? -eps : eps; idx >>= 1; }
    }
    if (live) {
        lane_rollout<M, 8, kBlock>(lds + t, unit, idx, d, central, minus, frame_skip, nsteps, qpos, qvel, ctrl, ctrl_pert, o, st, key);
    } else {
#pragma unroll
        for (int j = 0; j < ROW; j++) o[j] = 0.0;
    }
    // ---- more than eight floor contacts at once (a walker lying flat): the rollout above dropped rows and flagged
    // itself MBRL_ST_OVERFLOW.  Re-run, in this very wave, every lane of a unit that has such a lane here, with one
    // contact slot per floor candidate (64-bit row masks).  The wider column is strided for 32 lanes so that it fits
    // the LDS this workgroup already owns: lanes 0-31 first, then lanes 32-63.  A column may be differenced against a
    // nominal of the OTHER kind (a planned launch's nominal comes from the plan, which stays 8-slot when the nominal
    // itself did not overflow; a unit split over two waves can mix wide and narrow lanes): that is sound because the wide
    // and the fast instantiation compute the same bits whenever at most eight contacts are active -- same operations in
    // the same order, only the LDS addressing and the mask width differ -- which tests/test_gpu_overflow.py pins with
    // MODE_FORCE_WIDE on units that do not overflow.  Costs the wave a second (slower) set of rollouts -- planned
    // launches start such units first (they carry the largest keys), so it hides under the rest of the launch; callers
    // never see the flag.
    if constexpr (kNeedsRedo<M> && kRedoInWave<M, STEP_NAME, REDO_FD>) {
        // (MODE_FORCE_WIDE, tests only: every unit takes the wide path, to pin that it computes the fast path's bits)
        unsigned long long need = __builtin_amdgcn_ballot_w64(live && ((st & ST_OVERFLOW) || (mode_arg & MODE_FORCE_WIDE)));
        if (MBRL_RARE(need != 0 && !(mode_arg & MODE_NO_REDO))) {
            constexpr int KALL = 32, LANES = 32;                  // Layout clamps KALL to the number of floor candidates
            static_assert(Layout<M, KALL>::NMEM * LANES <= LY::NMEM * kBlock, "the wide columns must fit this workgroup's LDS");
            bool redo = false;
            while (need) {                                        // wave-uniform loop over the flagged lanes
                const int l = __builtin_ctzll(need);
                need &= need - 1;
                const int64_t ul = __shfl(unit, l, 64);
                redo = redo || (live && unit == ul);
            }
            for (int c = 0; c < 2; c++) {
                const bool mine = redo && (t >> 5) == c;
                if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
                if (mine)
                    lane_rollout<M, KALL, LANES>(lds + (t & 31), unit, idx, d, central, minus, frame_skip, nsteps, qpos, qvel, ctrl,
                                                 ctrl_pert, o, st, key);
            }
        }
    }
    if (live && (plan_mode == PLAN_WRITE || plan_mode == PLAN_RECORD)) {
        // (the key segment starts at a fixed offset, so a step launch that records into a larger plan gets the
        // buffer pointer advanced by its unit offset and needs to know nothing else about that plan)
        const PlanOffsets po(N, NS);
        reinterpret_cast<uint64_t *>(plan + po.key_in)[unit] = key;
        if (plan_mode == PLAN_WRITE) reinterpret_cast<uint32_t *>(plan + po.idx_in)[unit] = uint32_t(unit);
    }
    MBRL_STAMP(1);
    MBRL_STAMP(2);
    wave_outputs<M>(o, t, unit, p, idx, minus, nom, primary, live, planned, central, N, eps, dsds, dsda, drds, drda, s_next,
                    qpos_next, qvel_next, reward, plan);
    // status: OR over the unit's rollouts; a unit can live in two waves, so it is accumulated in HBM (zeroed by the host)
    if (status && live && st) atomicOr(&status[unit], st);
    MBRL_STAMP(3);
}

// Fix-up launch for FD launches of models that do not re-run in the wave (kRedoInWave): each wave scans a slice of the status words
// and, for every unit flagged MBRL_ST_OVERFLOW, runs ALL of that unit's rollouts itself with one contact slot per floor
// candidate and overwrites the unit's outputs, plan entries and status word.  Launched after every rollout launch of
// such a model; with nothing flagged it reads N status words and exits.
// The wider column must not make the workgroup harder to place than a fast-kernel wave: a launch on a second stream
// competes with those for the slots they free (measured with 59 KB per workgroup: the dispatcher held CUs half empty
// until two waves had drained, and the pipelined bench lost 2 %).  So the column is strided for 32 lanes, not 64 --
// under the fast kernel's 38 912 B -- and a unit's rollouts run in chunks of at most 31 lanes: the nominal in lane 0
// every time, perturbed rollouts (whole +-eps couples in central mode) in the lanes after it.
template <class M>
__global__ __launch_bounds__(kBlock) void overflow_kernel(int64_t N, int frame_skip, int nsteps, double eps, int mode,
                                                           const double *__restrict__ qpos,
                                                           const double *__restrict__ qvel,
                                                           const double *__restrict__ ctrl,
                                                           const double *__restrict__ ctrl_pert, double *__restrict__ dsds,
                                                           double *__restrict__ dsda, double *__restrict__ drds,
                                                           double *__restrict__ drda, double *__restrict__ s_next,
                                                           double *__restrict__ qpos_next, double *__restrict__ qvel_next,
                                                           double *__restrict__ reward, uint32_t *__restrict__ status,
                                                           char *__restrict__ plan, int plan_mode) {
    constexpr int KALL = 32;                       // Layout clamps it to the number of floor candidates
    constexpr int LANES = 32;
    using LY = Layout<M, KALL>;
    constexpr int NV = M::NV, NS = 2 * NV, ROW = NS + 1, NP = M::NU + 2 * NV;
    static_assert(LY::NMEM * LANES <= Layout<M>::NMEM * kBlock, "the fix-up workgroup must fit wherever a fast-kernel wave fits");
    __shared__ double lds[LY::NMEM * LANES];
    const bool force = (mode & MODE_FORCE_WIDE) != 0;                   // tests only: redo every unit
    mode &= MODE_MASK;
    const bool central = mode == MODE_CENTRAL;
    const int P = mode == MODE_STEP ? 0 : (central ? 2 * NP : NP);      // perturbed rollouts per unit
    const int cap = central ? LANES - 2 : LANES - 1;                    // ... per chunk (couples stay whole)
    const int t = threadIdx.x;
    for (int64_t base = int64_t(blockIdx.x) * kBlock; base < N; base += int64_t(gridDim.x) * kBlock) {
        const unsigned s = base + t < N ? status[base + t] : 0u;
        unsigned long long pend = __builtin_amdgcn_ballot_w64((s & ST_OVERFLOW) != 0u || (force && base + t < N));
        while (pend) {                             // wave-uniform: every lane sees the same ballot
            const int l = __builtin_ctzll(pend);
            pend &= pend - 1;
            const int64_t unit = base + l;
            unsigned all = 0;
            int off = 0;
            do {
                const int cnt = P - off < cap ? P - off : cap;
                const bool live = t <= cnt;
                const int p = t == 0 ? 0 : off + t;
                double o[ROW];
                unsigned st = 0;
                uint64_t key = 0;
                int idx = -1;
                bool minus = false;
                double d = eps;
                if (live && p > 0) {
                    idx = p - 1;
                    if (central) { minus = idx & 1; d = minus ? -eps : eps; idx >>= 1; }
                }
                if (live) {
                    lane_rollout<M, KALL, LANES>(lds + t, unit, idx, d, central, minus, frame_skip, nsteps, qpos, qvel, ctrl,
                                                 ctrl_pert, o, st, key);
                } else {
#pragma unroll
                    for (int j = 0; j < ROW; j++) o[j] = 0.0;
                }
                if (live && p == 0 && (plan_mode == PLAN_WRITE || plan_mode == PLAN_RECORD))
                    reinterpret_cast<uint64_t *>(plan + PlanOffsets(N, NS).key_in)[unit] = key;
                wave_outputs<M>(o, t, unit, p, idx, minus, 0, true, live, false, central, N, eps, dsds, dsda, drds, drda, s_next,
                                qpos_next, qvel_next, reward, plan);
#pragma unroll
                for (int b = 0; b < 5; b++)
                    if (__builtin_amdgcn_ballot_w64(live && ((st >> b) & 1u))) all |= 1u << b;
                off += cnt;
            } while (off < P);
            if (t == 0) status[unit] = all;        // REPLACES the word the fast launch left
        }
    }
}

// Stream probe (mbrl_stream_probe_spin): single-wave workgroups that spin for `ticks` of the 100 MHz real-time counter.
// Launched with a large dynamic LDS size, only a couple fit a CU, so a grid of a few thousand keeps the queue's
// dispatcher busy for many rounds -- like the rollout kernels do -- and a launch queued BEHIND it on the same hardware
// queue starts only when the last round has been placed.  Every wave leaves after `ticks` or 2^20 polls, whichever is first.
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks) {
    extern __shared__ unsigned char spin_pad[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int guard = 0; guard < (1 << 20) && __builtin_amdgcn_s_memrealtime() - t0 < ticks; guard++) __builtin_amdgcn_s_sleep(16);
    if (ticks == ~0ull) spin_pad[threadIdx.x] = 0;     // (keeps the allocation; never true for a real probe)
}

// mbrl_rollout_record: T consecutive env steps of B episodes in ONE launch, time-major buffers
//   Q [T+1, B, nq], V [T+1, B, nv]: step k reads slice k and writes slice k + 1;  C [T, B, nu];  R, status [T, B].
// Lane = episode, as in a step launch; a wave keeps its 64 episodes for all T steps, so the launch lasts as long as the
// slowest wave's SUM over the steps instead of the sum of every step's slowest wave (and T - 1 launch gaps go away).
// Every step is the very lane_rollout call a step launch makes, inputs read back from the slices this lane wrote one
// step earlier, so the states, rewards, status words and recorded plan entries are those of T mbrl_step_record
// launches bit for bit.  Units with more than eight floor contacts are re-run in the wave, as in a step launch.
template <class M>
__global__ __launch_bounds__(kBlock) void steps_kernel(int64_t B, int T, int frame_skip, int mode_flags, double *Q, double *V,
                                                        const double *__restrict__ C, double *__restrict__ R,
                                                        uint32_t *__restrict__ status, char *__restrict__ plan,
                                                        int64_t plan_units, int64_t unit_offset) {
    using LY = Layout<M>;
    constexpr int NV = M::NV, NU = M::NU, NS = 2 * NV, ROW = NS + 1;
    __shared__ double lds[LY::NMEM * kBlock];
    const int t = threadIdx.x;
    const int64_t b = int64_t(blockIdx.x) * kBlock + t;
    const bool live = b < B;
    const int64_t ep = live ? b : 0;
    for (int k = 0; k < T; k++) {
        const double *qk = Q + int64_t(k) * B * NV, *vk = V + int64_t(k) * B * NV, *ck = C + int64_t(k) * B * NU;
        double o[ROW];
        unsigned st = 0;
        uint64_t key = 0;
        if (live) lane_rollout<M, 8, kBlock>(lds + t, ep, -1, 0.0, false, false, frame_skip, 1, qk, vk, ck, nullptr, o, st, key);
        if constexpr (kNeedsRedo<M>) {
            unsigned long long need = __builtin_amdgcn_ballot_w64(live && ((st & ST_OVERFLOW) || (mode_flags & MODE_FORCE_WIDE)));
            if (MBRL_RARE(need != 0 && !(mode_flags & MODE_NO_REDO))) {
                constexpr int KALL = 32, LANES = 32;
                static_assert(Layout<M, KALL>::NMEM * LANES <= LY::NMEM * kBlock, "the wide columns must fit this workgroup's LDS");
                const bool redo = live && ((need >> t) & 1ull);
                for (int c = 0; c < 2; c++) {
                    const bool mine = redo && (t >> 5) == c;
                    if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
                    if (mine) lane_rollout<M, KALL, LANES>(lds + (t & 31), ep, -1, 0.0, false, false, frame_skip, 1, qk, vk, ck, nullptr, o, st, key);
                }
            }
        }
        if (live) {
            double *qn = Q + int64_t(k + 1) * B * NV + ep * NV, *vn = V + int64_t(k + 1) * B * NV + ep * NV;
#pragma unroll
            for (int j = 0; j < NV; j++) { qn[j] = o[j]; vn[j] = o[NV + j]; }
            R[int64_t(k) * B + ep] = o[NS];
            if (status) status[int64_t(k) * B + ep] = st;
            if (plan) {
                const int64_t u = unit_offset + int64_t(k) * B + ep;
                const PlanOffsets po(plan_units, NS);
                reinterpret_cast<uint64_t *>(plan + po.key_in)[u] = key;
                double *ns = reinterpret_cast<double *>(plan + po.nom_s) + u * NS;
#pragma unroll
                for (int j = 0; j < NS; j++) ns[j] = o[j];
            }
        }
        // the next step reads what this one stored (same lane, same addresses): keep the order explicit
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// status words start from zero (the rollout waves OR into them)
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *__restrict__ w, int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) w[i] = 0u;
}

// ---- mbrl_fd_jacobians_static: overlapping calls on one GPU share that GPU's one plan array and would race on it.
// Two control words in the (otherwise unused) tail of the plan header detect that on the device: ctl[0] counts the calls
// in flight, ctl[1] is raised by a call that starts while another is in flight and stays up until the last one leaves.
// Every call ends with static_flag_kernel -- if the word is up, its results cannot be trusted and every unit's status
// gets MBRL_ST_BADPLAN -- and static_release_kernel.  Conservative: a call that starts in the short window between
// another call's last rollout and its release is flagged although nothing was overwritten.
constexpr size_t kStaticCtlOffset = 64;      // bytes into the plan header (PlanHeader uses 32 of its 256)
__global__ void static_acquire_kernel(unsigned *ctl) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && atomicAdd(&ctl[0], 1u) != 0u) atomicExch(&ctl[1], 1u);
}
__global__ __launch_bounds__(256) void static_flag_kernel(unsigned *ctl, uint32_t *__restrict__ status, int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicOr(&status[i], ST_BADPLAN);
}
__global__ void static_release_kernel(unsigned *ctl) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && atomicSub(&ctl[0], 1u) == 1u) atomicExch(&ctl[1], 0u);
}

// mbrl_fd_plan_finish: header and identity order of a plan whose keys and nominal results were recorded by step launches.
__global__ __launch_bounds__(256) void plan_finish_kernel(int64_t N, int nsteps, int frame_skip, int env_id, int ns,
                                                           char *__restrict__ plan) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i == 0) {
        PlanHeader *hd = reinterpret_cast<PlanHeader *>(plan);
        hd->magic = kPlanMagic; hd->units = N; hd->nsteps = nsteps; hd->frame_skip = frame_skip; hd->env_id = env_id;
        hd->pad = 0;
    }
    if (i < N) reinterpret_cast<uint32_t *>(plan + PlanOffsets(N, ns).idx_in)[i] = uint32_t(i);
}

// Backward recurrence of the op chain (mujoco.py:43-104): one thread per (episode, output lane).
// n <= 18, nu <= 6: each episode is walked by one 32-lane group; lane l owns g[l].
// Unit (b, t) sits at index b * sb + t * st of every array: episode-major (sb = H, st = 1) or time-major (sb = 1, st = B).
// Episode b of this launch is episode b_off + b of a batch of b_tot (a rank's shard of a sharded batch; 0, B otherwise).
__global__ __launch_bounds__(kBlock) void vjp_scan_kernel(int64_t B, int H, int n, int nu, int gw, int64_t sb, int64_t st,
                                                           int64_t b_off, int64_t b_tot,
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
    // 'prioritise': autograd runs episodes last-to-first, the running sum carries over (mujoco.py:65) -- over the
    // episodes of the WHOLE batch, so a shard seeds it with what the later episodes of every rank have added
    double S = live ? double(b_tot - 1 - (b_off + b)) * (0.5 * double(H) * double(H + 1)) : 0.0;
    for (int t = H - 1; t >= 0; t--) {
        gs[grp][l] = g;
        __syncthreads();
        if (live) {
            const int64_t k = b * sb + t * st;
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
// Test knobs (mbrl_debug_set_flags, below; not part of include/mbrl.h): MODE_NO_REDO shows the flags the 8-slot rollouts
// raise, MODE_FORCE_WIDE sends every unit through the wide re-run.  Zero in production; read once per launch.
static std::atomic<int> g_debug_flags{0};
static int fail(int code, const char *fmt, const char *a = "", long long b = 0) {
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}

// -DMBRL_ONLY_ENV=k (A/B and ISA-inspection builds): only model k is instantiated, the others report "unknown env"
#ifdef MBRL_ONLY_ENV
#define MBRL_CASE(k, MODEL, CALL) case k: if constexpr (k == MBRL_ONLY_ENV) { using M = MODEL; CALL; } break;
#else
#define MBRL_CASE(k, MODEL, CALL) case k: { using M = MODEL; CALL; } break;
#endif
#define MBRL_DISPATCH(id, CALL)                                        \
    switch (id) {                                                      \
    MBRL_CASE(0, MdlInvertedPendulum, CALL)                            \
    MBRL_CASE(1, MdlInvertedDoublePendulum, CALL)                      \
    MBRL_CASE(2, MdlSwimmer, CALL)                                     \
    MBRL_CASE(3, MdlHopper, CALL)                                      \
    MBRL_CASE(4, MdlWalker2d, CALL)                                    \
    MBRL_CASE(5, MdlHalfCheetah, CALL)                                 \
    default: break;                                                    \
    }

// ---- scratch of the no-allocation entry point (mbrl_fd_jacobians_static) ----
// The launch plan of the two-launch schedule (PlanOffsets + sort scratch) normally lives in a caller-supplied workspace.
// Callers that cannot allocate get it here: a file-scope device array, one copy per GPU, held until the process ends.
// `scratch` is only a host-side stand-in for the symbol; host code never passes it anywhere.  Its device address is
// looked up once per device (scratch_init, with that device current) and kept in g_scratch_ptr, and that pointer is
// what every kernel argument / memset / copy uses.
constexpr size_t kStaticScratchBytes = size_t(MBRL_STATIC_SCRATCH_BYTES);
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mu;

// Once per device, at set-up (mbrl_model_create): the address of THIS device's copy of `scratch`.
static int scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices) return -1;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return 0;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    (void)hipSetDevice(prev);
    if (e != hipSuccess || !ptr) return -1;
    g_scratch_ptr[device] = ptr;
    return 0;
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
static int launch_rollouts(const mbrl_model *m, int64_t N, int mode, const double *qpos, const double *qvel,
                           const double *ctrl, const double *ctrl_pert, double eps, int nsteps, double *dsds, double *dsda,
                           double *drds, double *drda, double *s_next, double *qpos_next, double *qvel_next, double *reward,
                           uint32_t *status, hipStream_t st, char *plan = nullptr, int plan_mode = PLAN_NONE) {
    constexpr int NP = M::NU + 2 * M::NV;
    const bool central = mode == MODE_CENTRAL;
    const WaveMap wm(mode == MODE_STEP ? 1 : (central ? 1 + 2 * NP : 1 + NP), central);
    int64_t blocks = 2 * ((N + wm.upp - 1) / wm.upp);
    const bool planned = plan_mode == PLAN_READ;
    if (planned) blocks = (N * (central ? 2 * NP : NP) + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return -2;
    // per-unit status words are OR-ed into HBM by the waves (a unit can span two): start from zero, in stream order
    // (a planned launch ORs into the words its plan launch wrote)
    // (a kernel, not hipMemsetAsync: replayed from a captured HIP graph, the memset node left garbage in the words from
    // the second replay on -- ROCm 7.2; tests/test_zz_no_alloc.py replays mbrl_fd_jacobians_static three times)
    if (status && !planned)
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, status, N);
    const int mode_arg = mode | g_debug_flags.load(std::memory_order_relaxed);
    if (mode == MODE_STEP)
        hipLaunchKernelGGL((rollout_kernel<M, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, N, m->frame_skip, nsteps,
                           eps, mode_arg, qpos, qvel, ctrl, ctrl_pert, dsds, dsda, drds, drda, s_next, qpos_next, qvel_next, reward,
                           status, plan, plan_mode);
    else if (kNeedsRedo<M> && !kRedoInWave<M, false> && !status) {
        // no status array, so no fix-up launch: the instantiation that re-runs units with more than eight floor
        // contacts in the wave (same results; include/mbrl.h "Any output pointer may be NULL")
        if constexpr (kNeedsRedo<M> && !kRedoInWave<M, false>)
            hipLaunchKernelGGL((rollout_kernel<M, false, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, N, m->frame_skip,
                               nsteps, eps, mode_arg, qpos, qvel, ctrl, ctrl_pert, dsds, dsda, drds, drda, s_next, qpos_next,
                               qvel_next, reward, status, plan, plan_mode);
    } else
        hipLaunchKernelGGL((rollout_kernel<M, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, N, m->frame_skip, nsteps,
                           eps, mode_arg, qpos, qvel, ctrl, ctrl_pert, dsds, dsda, drds, drda, s_next, qpos_next, qvel_next, reward,
                           status, plan, plan_mode);
    if constexpr (kNeedsRedo<M> && !kRedoInWave<M, false>) {
      if (mode != MODE_STEP)   // step / plan launches (rollout_kernel<M, true>) re-run in the wave
        // fix-up launch (overflow_kernel).  A planned launch's flagged units are redone whole, nominal included, so it
        // needs no plan; plan and record launches re-record their units' keys.
        if (status && !(mode_arg & MODE_NO_REDO)) {
            int64_t fb = (N + kBlock - 1) / kBlock;
            if (fb > 1024) fb = 1024;
            hipLaunchKernelGGL((overflow_kernel<M>), dim3((unsigned)fb), dim3(kBlock), 0, st, N, m->frame_skip, nsteps, eps,
                               mode | (mode_arg & MODE_FORCE_WIDE), qpos, qvel, ctrl, ctrl_pert, dsds, dsda, drds, drda, s_next,
                               qpos_next, qvel_next, reward, status, planned ? nullptr : plan, planned ? PLAN_NONE : plan_mode);
        }
    }
    return 0;
}

// Size of rocprim's sort scratch for n units: a host-side query (no launch) whose answer depends on n only, so it is
// asked once per n and remembered -- mbrl_step_record / _planned / _finish sit in the latency-bound rollout loop.
static bool sort_scratch_bytes(int64_t n, size_t &bytes) {
    struct Entry { int64_t n; size_t bytes; };
    static thread_local Entry cache[8] = {};
    static thread_local int next = 0;
    for (const Entry &e : cache)
        if (e.n == n && e.bytes) { bytes = e.bytes; return true; }
    bytes = 0;
    uint32_t *nil = nullptr;
    uint64_t *nil64 = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, nil64, nil64, nil, nil, size_t(n), 0, kKeyBits, hipStream_t(nullptr)) != hipSuccess)
        return false;
    if (bytes == 0) bytes = 1;    // (never returned by rocprim for n > 0; keeps "0 = empty slot" unambiguous)
    cache[next] = Entry{n, bytes};
    next = (next + 1) % 8;
    return true;
}

template <class M>
static void launch_steps(const mbrl_model *m, int64_t blocks, hipStream_t st, int64_t B, int T, int flags, double *qpos, double *qvel,
                         const double *ctrl, double *reward, uint32_t *status, char *plan, int64_t plan_units, int64_t unit_offset) {
    hipLaunchKernelGGL((steps_kernel<M>), dim3((unsigned)blocks), dim3(kBlock), 0, st, B, T, m->frame_skip, flags, qpos, qvel, ctrl,
                       reward, status, plan, plan_units, unit_offset);
}

static bool plan_layout(int64_t N, int ns, hipStream_t, PlanLayout &L) {
    L.sort_tmp = PlanOffsets(N, ns).end;
    if (!sort_scratch_bytes(N, L.sort_bytes)) return false;
    L.total = (L.sort_tmp + L.sort_bytes + 255) & ~size_t(255);
    return true;
}

#ifdef MBRL_PHASE_TIMING
extern "C" int mbrl_debug_phase_read(unsigned long long *host32, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_phase_acc), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_acc), z, sizeof z); }
    return rc;
}
#endif

// Utility for the host wrapper (engine.py picks a side stream that is SEEN to overlap the caller's stream; HIP has no query
// for which hardware queue a stream uses): `blocks` single-wave workgroups with lds_bytes of LDS each spin for `usec`
// microseconds on hip_stream.  Not part of include/mbrl.h.
extern "C" int mbrl_stream_probe_spin(const mbrl_model *m, int blocks, int usec, int lds_bytes, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_stream_probe_spin: null model");
    if (blocks < 1 || blocks > (1 << 16) || usec < 1 || usec > 20000 || lds_bytes < 0 || lds_bytes > 64 * 1024)
        return fail(-7, "mbrl_stream_probe_spin: argument out of range");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_stream_probe_spin: hipSetDevice failed");
    hipLaunchKernelGGL(spin_kernel, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)hip_stream,
                       (unsigned long long)usec * 100ull);
    return check_launch(m, "mbrl_stream_probe_spin");
}

// Test knob, deliberately outside include/mbrl.h: bit 0 = no re-run of units with more than eight floor contacts (their
// MBRL_ST_OVERFLOW flag stays visible), bit 1 = every unit takes the wide re-run.  Returns the previous setting.
extern "C" int mbrl_debug_set_flags(int flags) {
    const int v = ((flags & 1) ? MODE_NO_REDO : 0) | ((flags & 2) ? MODE_FORCE_WIDE : 0);
    const int old = g_debug_flags.exchange(v);
    return ((old & MODE_NO_REDO) ? 1 : 0) | ((old & MODE_FORCE_WIDE) ? 2 : 0);
}

#ifdef MBRL_WAVE_TIMING
extern "C" int mbrl_debug_set_wave_timing(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_timing), &buf, sizeof buf);
}
#endif

extern "C" {

int mbrl_abi_version(void) { return MBRL_ABI_VERSION; }

int mbrl_lane_map(int nq, int nv, int nu, int fd_mode, int *units_per_pair, int *unit, int *rollout, int *ref_lane,
                  int *primary) {
    if (!unit || !rollout || !ref_lane || !primary) return fail(-1, "mbrl_lane_map: null argument");
    if (fd_mode < -1 || fd_mode > MBRL_FD_CENTRAL) return fail(-7, "mbrl_lane_map: bad fd_mode");
    const int NP = nu + nq + nv;
    if (nq < 1 || nv < 1 || nu < 0 || 1 + 2 * NP > 64) return fail(-6, "mbrl_lane_map: a unit's rollouts must fit one wave");
    const bool central = fd_mode == MBRL_FD_CENTRAL;
    const WaveMap wm(fd_mode < 0 ? 1 : (central ? 1 + 2 * NP : 1 + NP), central);
    if (units_per_pair) *units_per_pair = wm.upp;
    for (int half = 0; half < 2; half++)
        for (int t = 0; t < 64; t++) {
            int u = -1, p = -1, nom = -1;
            bool prim = false;
            const bool ok = wm.lane(half, t, u, p, nom, prim);
            const int i = half * 64 + t;
            unit[i] = ok ? u : -1; rollout[i] = ok ? p : -1; ref_lane[i] = ok ? nom : -1; primary[i] = ok && prim;
        }
    return 0;
}

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
    if (device < kMaxDevices && scratch_init(device))
        return fail(-10, "mbrl_model_create: hipGetSymbolAddress(scratch) failed on device %s%lld", "", device);
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
    int rc = 0;
    MBRL_DISPATCH(m->env_id, (rc = launch_rollouts<M>(m, N, MODE_STEP, qpos, qvel, ctrl, nullptr, 0.0, 1, nullptr, nullptr,
                                                      nullptr, nullptr, nullptr, qpos_next, qvel_next, reward, status, st)));
    if (rc) return fail(-10, "mbrl_step: clearing the status words failed");
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
    int rc = 0;
    MBRL_DISPATCH(m->env_id, (rc = launch_rollouts<M>(m, N, fd_mode == MBRL_FD_CENTRAL ? MODE_CENTRAL : MODE_FORWARD, qpos,
                                                      qvel, ctrl, ctrl_pert, eps, nsteps, dsds, dsda, drds, drda, s_next,
                                                      nullptr, nullptr, reward, status, st)));
    if (rc) return fail(-10, "mbrl_fd_jacobians: clearing the status words failed");
    return check_launch(m, "mbrl_fd_jacobians");
}

int64_t mbrl_fd_plan_bytes(const mbrl_model *m, int64_t N) {
    if (!m) return fail(-1, "mbrl_fd_plan_bytes: null model");
    if (N < 0 || N > (int64_t(1) << 31)) return fail(-6, "mbrl_fd_plan_bytes: bad N");
    if (N == 0) return 0;
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_fd_plan_bytes: hipSetDevice failed");
    PlanLayout L;
    if (!plan_layout(N, m->nq + m->nv, nullptr, L)) return fail(-10, "mbrl_fd_plan_bytes: sort size query failed");
    return int64_t(L.total);
}

int mbrl_fd_plan(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl, int nsteps,
                 void *plan, int64_t plan_bytes, double *s_next, double *reward, uint32_t *status, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_fd_plan: null model");
    if (N < 0) return fail(-6, "mbrl_fd_plan: negative N");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl || !plan) return fail(-1, "mbrl_fd_plan: null buffer");
    if (nsteps <= 0) nsteps = 1;
    if (N > (int64_t(1) << 31)) return fail(-6, "mbrl_fd_plan: N too large for one launch");
    if (reinterpret_cast<uintptr_t>(plan) & 255) return fail(-7, "mbrl_fd_plan: the plan buffer must be 256-byte aligned");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_fd_plan: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    const int ns = m->nq + m->nv;
    PlanLayout L;
    if (!plan_layout(N, ns, st, L)) return fail(-10, "mbrl_fd_plan: sort size query failed");
    if (plan_bytes < int64_t(L.total))
        return fail(-8, "mbrl_fd_plan: plan buffer too small%s (need %lld bytes, see mbrl_fd_plan_bytes)", "", (long long)L.total);
    char *w = static_cast<char *>(plan);
    const PlanOffsets po(N, ns);
    uint64_t *key_in = reinterpret_cast<uint64_t *>(w + po.key_in), *key_out = reinterpret_cast<uint64_t *>(w + po.key_out);
    uint32_t *idx_in = reinterpret_cast<uint32_t *>(w + po.idx_in), *perm = reinterpret_cast<uint32_t *>(w + po.perm);
    double *nom_s = reinterpret_cast<double *>(w + po.nom_s), *nom_r = reinterpret_cast<double *>(w + po.nom_r);
    int rc = 0;
    MBRL_DISPATCH(m->env_id, (rc = launch_rollouts<M>(m, N, MODE_STEP, qpos, qvel, ctrl, nullptr, 0.0, nsteps, nullptr, nullptr,
                                                      nullptr, nullptr, nom_s, nullptr, nullptr, nom_r, status, st, w,
                                                      PLAN_WRITE)));
    if (rc) return fail(-10, "mbrl_fd_plan: clearing the status words failed");
    if ((rc = check_launch(m, "mbrl_fd_plan"))) return rc;
    size_t tmp = L.sort_bytes;
    if (rocprim::radix_sort_pairs(w + L.sort_tmp, tmp, key_in, key_out, idx_in, perm, size_t(N), 0, kKeyBits, st) != hipSuccess)
        return fail(-10, "mbrl_fd_plan: sorting the units failed");
    if (s_next && hipMemcpyAsync(s_next, nom_s, sizeof(double) * size_t(N) * ns, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(-10, "mbrl_fd_plan: copying next_state failed");
    if (reward && hipMemcpyAsync(reward, nom_r, sizeof(double) * size_t(N), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(-10, "mbrl_fd_plan: copying reward failed");
    return 0;
}

int mbrl_step_record(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                     double *qpos_next, double *qvel_next, double *reward, uint32_t *status, void *plan, int64_t plan_bytes,
                     int64_t plan_units, int64_t unit_offset, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_step_record: null model");
    if (N < 0 || plan_units < 0 || unit_offset < 0) return fail(-6, "mbrl_step_record: negative size");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl || !qpos_next || !qvel_next || !plan) return fail(-1, "mbrl_step_record: null buffer");
    if (unit_offset + N > plan_units) return fail(-6, "mbrl_step_record: units [offset, offset + N) must lie inside the plan");
    if (plan_units > (int64_t(1) << 31)) return fail(-6, "mbrl_step_record: plan too large");
    if (reinterpret_cast<uintptr_t>(plan) & 255) return fail(-7, "mbrl_step_record: the plan buffer must be 256-byte aligned");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_step_record: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    const int ns = m->nq + m->nv;
    PlanLayout L;
    if (!plan_layout(plan_units, ns, st, L)) return fail(-10, "mbrl_step_record: sort size query failed");
    if (plan_bytes < int64_t(L.total))
        return fail(-8, "mbrl_step_record: plan buffer too small%s (need %lld bytes, see mbrl_fd_plan_bytes)", "", (long long)L.total);
    char *w = static_cast<char *>(plan);
    const PlanOffsets po(plan_units, ns);
    double *nom_s = reinterpret_cast<double *>(w + po.nom_s) + unit_offset * ns;
    int rc = 0;
    // the launch is mbrl_step's; it also leaves s' in the plan's nominal segment and its key in the key segment
    MBRL_DISPATCH(m->env_id, (rc = launch_rollouts<M>(m, N, MODE_STEP, qpos, qvel, ctrl, nullptr, 0.0, 1, nullptr, nullptr,
                                                      nullptr, nullptr, nom_s, qpos_next, qvel_next, reward, status, st,
                                                      w + sizeof(uint64_t) * size_t(unit_offset), PLAN_RECORD)));
    if (rc) return fail(-10, "mbrl_step_record: clearing the status words failed");
    return check_launch(m, "mbrl_step_record");
}

int mbrl_rollout_record(const mbrl_model *m, int64_t B, int T, double *qpos, double *qvel, const double *ctrl, double *reward,
                        uint32_t *status, void *plan, int64_t plan_bytes, int64_t plan_units, int64_t unit_offset,
                        void *hip_stream) {
    if (!m) return fail(-1, "mbrl_rollout_record: null model");
    if (B < 0 || T < 0 || plan_units < 0 || unit_offset < 0) return fail(-6, "mbrl_rollout_record: negative size");
    if (B == 0 || T == 0) return 0;
    if (!qpos || !qvel || !ctrl || !reward) return fail(-1, "mbrl_rollout_record: null buffer");
    if (B > (int64_t(1) << 31) || int64_t(T) * B > (int64_t(1) << 40)) return fail(-6, "mbrl_rollout_record: batch too large");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_rollout_record: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    char *w = static_cast<char *>(plan);
    if (plan) {
        if (unit_offset + int64_t(T) * B > plan_units)
            return fail(-6, "mbrl_rollout_record: units [offset, offset + T * B) must lie inside the plan");
        if (plan_units > (int64_t(1) << 31)) return fail(-6, "mbrl_rollout_record: plan too large");
        if (reinterpret_cast<uintptr_t>(plan) & 255) return fail(-7, "mbrl_rollout_record: the plan buffer must be 256-byte aligned");
        PlanLayout L;
        if (!plan_layout(plan_units, m->nq + m->nv, st, L)) return fail(-10, "mbrl_rollout_record: sort size query failed");
        if (plan_bytes < int64_t(L.total))
            return fail(-8, "mbrl_rollout_record: plan buffer too small%s (need %lld bytes, see mbrl_fd_plan_bytes)", "", (long long)L.total);
    }
    const int64_t blocks = (B + kBlock - 1) / kBlock;
    const int flags = g_debug_flags.load(std::memory_order_relaxed);
    MBRL_DISPATCH(m->env_id, (launch_steps<M>(m, blocks, st, B, T, flags, qpos, qvel, ctrl, reward, status, w, plan_units, unit_offset)));
    return check_launch(m, "mbrl_rollout_record");
}

int mbrl_fd_plan_finish(const mbrl_model *m, int64_t plan_units, const double *reward, void *plan, int64_t plan_bytes,
                        void *hip_stream) {
    if (!m) return fail(-1, "mbrl_fd_plan_finish: null model");
    if (plan_units < 0) return fail(-6, "mbrl_fd_plan_finish: negative size");
    if (plan_units == 0) return 0;
    if (!plan || !reward) return fail(-1, "mbrl_fd_plan_finish: null buffer");
    if (plan_units > (int64_t(1) << 31)) return fail(-6, "mbrl_fd_plan_finish: plan too large");
    if (reinterpret_cast<uintptr_t>(plan) & 255) return fail(-7, "mbrl_fd_plan_finish: the plan buffer must be 256-byte aligned");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_fd_plan_finish: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    const int ns = m->nq + m->nv;
    PlanLayout L;
    if (!plan_layout(plan_units, ns, st, L)) return fail(-10, "mbrl_fd_plan_finish: sort size query failed");
    if (plan_bytes < int64_t(L.total)) return fail(-8, "mbrl_fd_plan_finish: plan buffer too small");
    char *w = static_cast<char *>(plan);
    const PlanOffsets po(plan_units, ns);
    hipLaunchKernelGGL(plan_finish_kernel, dim3((unsigned)((plan_units + 255) / 256)), dim3(256), 0, st, plan_units, 1,
                       m->frame_skip, m->env_id, ns, w);
    int rc = check_launch(m, "mbrl_fd_plan_finish");
    if (rc) return rc;
    if (hipMemcpyAsync(w + po.nom_r, reward, sizeof(double) * size_t(plan_units), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(-10, "mbrl_fd_plan_finish: copying the rewards failed");
    size_t tmp = L.sort_bytes;
    if (rocprim::radix_sort_pairs(w + L.sort_tmp, tmp, reinterpret_cast<uint64_t *>(w + po.key_in),
                                  reinterpret_cast<uint64_t *>(w + po.key_out), reinterpret_cast<uint32_t *>(w + po.idx_in),
                                  reinterpret_cast<uint32_t *>(w + po.perm), size_t(plan_units), 0, kKeyBits, st) != hipSuccess)
        return fail(-10, "mbrl_fd_plan_finish: sorting the units failed");
    return 0;
}

int mbrl_fd_jacobians_planned(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                              const double *ctrl_pert, double eps, int fd_mode, int nsteps, const void *plan,
                              int64_t plan_bytes, double *dsds, double *dsda, double *drds, double *drda, uint32_t *status,
                              void *hip_stream) {
    if (!m) return fail(-1, "mbrl_fd_jacobians_planned: null model");
    if (N < 0) return fail(-6, "mbrl_fd_jacobians_planned: negative N");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl || !plan) return fail(-1, "mbrl_fd_jacobians_planned: null input buffer");
    if (!(eps > 0)) return fail(-7, "mbrl_fd_jacobians_planned: eps must be > 0");
    if (fd_mode != MBRL_FD_FORWARD && fd_mode != MBRL_FD_CENTRAL) return fail(-7, "mbrl_fd_jacobians_planned: bad fd_mode");
    if (nsteps <= 0) nsteps = 1;
    if (N > (int64_t(1) << 31)) return fail(-6, "mbrl_fd_jacobians_planned: N too large for one launch");
    if (reinterpret_cast<uintptr_t>(plan) & 255) return fail(-7, "mbrl_fd_jacobians_planned: the plan buffer must be 256-byte aligned");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_fd_jacobians_planned: hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    PlanLayout L;
    if (!plan_layout(N, m->nq + m->nv, st, L)) return fail(-10, "mbrl_fd_jacobians_planned: sort size query failed");
    if (plan_bytes < int64_t(L.total)) return fail(-8, "mbrl_fd_jacobians_planned: plan buffer too small");
    int rc = 0;
    MBRL_DISPATCH(m->env_id,
                  (rc = launch_rollouts<M>(m, N, fd_mode == MBRL_FD_CENTRAL ? MODE_CENTRAL : MODE_FORWARD, qpos, qvel, ctrl,
                                           ctrl_pert, eps, nsteps, dsds, dsda, drds, drda, nullptr, nullptr, nullptr, nullptr,
                                           status, st, static_cast<char *>(const_cast<void *>(plan)), PLAN_READ)));
    if (rc) return fail(-6, "mbrl_fd_jacobians_planned: N too large for one launch");
    return check_launch(m, "mbrl_fd_jacobians_planned");
}

int mbrl_fd_jacobians_static(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                             double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                             double *s_next, double *reward, uint32_t *status, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_fd_jacobians_static: null model");
    if (N < 0) return fail(-6, "mbrl_fd_jacobians_static: negative N");
    if (N == 0) return 0;
    if (N > int64_t(MBRL_STATIC_MAX_UNITS))
        return fail(-8, "mbrl_fd_jacobians_static: N exceeds MBRL_STATIC_MAX_UNITS%s (%lld units asked; use mbrl_fd_plan + "
                        "mbrl_fd_jacobians_planned with a caller-owned plan buffer)", "", (long long)N);
    void *ptr = (m->device >= 0 && m->device < kMaxDevices) ? g_scratch_ptr[m->device] : nullptr;
    if (!ptr) return fail(-10, "mbrl_fd_jacobians_static: no static scratch on this device");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_fd_jacobians_static: hipSetDevice failed");
    PlanLayout L;
    if (!plan_layout(N, m->nq + m->nv, (hipStream_t)hip_stream, L))
        return fail(-10, "mbrl_fd_jacobians_static: sort size query failed");
    if (L.total > kStaticScratchBytes)
        return fail(-8, "mbrl_fd_jacobians_static: the plan of this batch does not fit the static scratch%s (%lld bytes)", "",
                    (long long)L.total);
    // every launch takes the device address looked up at set-up, never the symbol's name
    hipStream_t st = (hipStream_t)hip_stream;
    unsigned *ctl = reinterpret_cast<unsigned *>(static_cast<char *>(ptr) + kStaticCtlOffset);
    hipLaunchKernelGGL(static_acquire_kernel, dim3(1), dim3(64), 0, st, ctl);
    int rc = mbrl_fd_plan(m, N, qpos, qvel, ctrl, nsteps, ptr, int64_t(kStaticScratchBytes), s_next, reward, status, hip_stream);
    if (!rc)
        rc = mbrl_fd_jacobians_planned(m, N, qpos, qvel, ctrl, nullptr, eps, fd_mode, nsteps, ptr, int64_t(kStaticScratchBytes),
                                       dsds, dsda, drds, drda, status, hip_stream);
    // a call that overlapped another one on this GPU raced on the array: say so in the status words, then leave
    if (!rc && status)
        hipLaunchKernelGGL(static_flag_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, ctl, status, N);
    hipLaunchKernelGGL(static_release_kernel, dim3(1), dim3(64), 0, st, ctl);
    if (rc) return rc;
    return check_launch(m, "mbrl_fd_jacobians_static");
}

// Test knob, outside include/mbrl.h: stands in for ANOTHER mbrl_fd_jacobians_static call in flight on the model's GPU
// (enter = 1: as if it had started; 0: as if it had finished), so that the overlap detection can be tested without a race.
int mbrl_debug_static_hold(const mbrl_model *m, int enter, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_debug_static_hold: null model");
    void *ptr = (m->device >= 0 && m->device < kMaxDevices) ? g_scratch_ptr[m->device] : nullptr;
    if (!ptr) return fail(-10, "mbrl_debug_static_hold: no static scratch on this device");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_debug_static_hold: hipSetDevice failed");
    unsigned *ctl = reinterpret_cast<unsigned *>(static_cast<char *>(ptr) + kStaticCtlOffset);
    if (enter) hipLaunchKernelGGL(static_acquire_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, ctl);
    else hipLaunchKernelGGL(static_release_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, ctl);
    return check_launch(m, "mbrl_debug_static_hold");
}

int mbrl_vjp_scan(const mbrl_model *m, int64_t B, int H, int grad_weights, const double *dsds, const double *dsda,
                  const double *drds, const double *drda, const double *c, double *da, double *ds0, void *hip_stream) {
    return mbrl_vjp_scan_layout(m, B, H, grad_weights, MBRL_EPISODE_MAJOR, dsds, dsda, drds, drda, c, da, ds0, hip_stream);
}

int mbrl_vjp_scan_layout(const mbrl_model *m, int64_t B, int H, int grad_weights, int layout, const double *dsds,
                         const double *dsda, const double *drds, const double *drda, const double *c, double *da,
                         double *ds0, void *hip_stream) {
    return mbrl_vjp_scan_sharded(m, B, H, grad_weights, layout, 0, B, dsds, dsda, drds, drda, c, da, ds0, hip_stream);
}

int mbrl_vjp_scan_sharded(const mbrl_model *m, int64_t B, int H, int grad_weights, int layout, int64_t episode_offset,
                          int64_t total_episodes, const double *dsds, const double *dsda, const double *drds,
                          const double *drda, const double *c, double *da, double *ds0, void *hip_stream) {
    if (!m) return fail(-1, "mbrl_vjp_scan: null model");
    if (episode_offset < 0 || total_episodes < episode_offset + B)
        return fail(-6, "mbrl_vjp_scan_sharded: the shard [offset, offset + B) must lie inside [0, total_episodes)");
    if (layout != MBRL_EPISODE_MAJOR && layout != MBRL_TIME_MAJOR) return fail(-7, "mbrl_vjp_scan: bad layout");
    if (B < 0 || H < 0) return fail(-6, "mbrl_vjp_scan: negative size");
    if (B == 0 || H == 0) return 0;
    if (!dsds || !dsda || !drds || !drda || !c || !da) return fail(-1, "mbrl_vjp_scan: null buffer");
    if (grad_weights < 0 || grad_weights > 2) return fail(-7, "mbrl_vjp_scan: bad grad_weights");
    if (hipSetDevice(m->device) != hipSuccess) return fail(-4, "mbrl_vjp_scan: hipSetDevice failed");
    const int per = kBlock / 32;
    const unsigned blocks = (unsigned)((B + per - 1) / per);
    const int64_t sb = layout == MBRL_TIME_MAJOR ? 1 : H, st = layout == MBRL_TIME_MAJOR ? B : 1;
    hipLaunchKernelGGL(vjp_scan_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)hip_stream, B, H, m->nq + m->nv,
                       m->nu, grad_weights, sb, st, episode_offset, total_episodes, dsds, dsda, drds, drda, c, da, ds0);
    return check_launch(m, "mbrl_vjp_scan");
}

}  // extern "C"
