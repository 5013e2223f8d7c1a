/*
 * mbrl.h -- C ABI of the MI355X-native FD-linearisation path (libmbrl_hip.so).
 *
 * The reference (MahanFathi/Model-Based-RL) has no FFI of its own: its hot path is the
 * Python protocol  MjBlock.forward/backward  (model/blocks/mujoco.py:12-104) over
 * mj_forward / calculate_gradients (mujoco/utils/forward.py:13-38,
 * mujoco/utils/backward.py:343-480), which reach MuJoCo through mujoco-py.  These entry
 * points are what a binding for that path would bind; each cites what it replaces.
 *
 * Conventions: plain C, no C++/torch types.  Every pointer named q*, ctrl, ds*, dr*,
 * s_next, reward, status is DEVICE memory (HBM) owned by the caller, row-major,
 * fp64 (status: uint32).  Calls are asynchronous on `hip_stream` (a hipStream_t, may be
 * NULL for the default stream).  The library allocates nothing per call and keeps no
 * mutable global state: a model handle is immutable after creation, so calls are
 * re-entrant.  Return value 0 = ok, negative = error (text via mbrl_last_error()).
 */
#ifndef MBRL_H
#define MBRL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBRL_ABI_VERSION 9

/* per-unit status bits (replace mujoco-py's MujocoException / the reference's asserts,
 * backward.py:371,374) */
#define MBRL_ST_NONFINITE  1u  /* NaN/Inf state or non-positive pivot */
#define MBRL_ST_NOCONVERGE 2u  /* constraint solver hit its iteration cap */
#define MBRL_ST_OVERFLOW   4u  /* internal: more than 8 floor contacts at once.  Callers never see it: every entry point
                                * re-runs such units with full contact capacity, with or without a status array
                                *  */
#define MBRL_ST_DEGENERATE 8u  /* reserved; not raised since ABI 6: crossing capsule axes have a defined normal */
#define MBRL_ST_BADPLAN    16u /* mbrl_fd_jacobians_planned: the plan was made for another batch (N, nsteps, model);
                                * mbrl_fd_jacobians_static: the call overlapped another one on the same GPU */

#define MBRL_FD_FORWARD 0      /* reference parity: one-sided, (f(x+eps)-f(x))/eps, backward.py:402-406 */
#define MBRL_FD_CENTRAL 1      /* (f(x+eps)-f(x-eps))/(2 eps) */

typedef struct mbrl_model mbrl_model;

int mbrl_abi_version(void);

/* Replaces build_agent(cfg) -> env ctor (mujoco/build.py:6-37, e.g.
 * mujoco/envs/half_cheetah.py:15-22): env_name is cfg.MUJOCO.ENV ("HalfCheetahEnv",
 * "HopperEnv", "Walker2dEnv", "SwimmerEnv", "InvertedPendulumEnv",
 * "InvertedDoublePendulumEnv"), frame_skip is cfg.MODEL.FRAME_SKIP (<=0: the yaml's
 * value for that env), device is the HIP device ordinal the handle launches on. */
int mbrl_model_create(const char *env_name, int frame_skip, int device, mbrl_model **out);
void mbrl_model_destroy(mbrl_model *m);

/* agent.model.nq / nv / nu and env.dt (= timestep * frame_skip). */
int mbrl_model_dims(const mbrl_model *m, int *nq, int *nv, int *nu, double *dt);

/* Replaces mj_forward() -> agent.step(action) (mujoco/utils/forward.py:13-38; env.step
 * e.g. mujoco/envs/half_cheetah.py:24-33): N independent units, one env step each.
 * qpos [N,nq], qvel [N,nv], ctrl [N,nu] -> qpos_next, qvel_next, reward [N], status [N]. */
int mbrl_step(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
              double *qpos_next, double *qvel_next, double *reward, uint32_t *status, void *hip_stream);

/* Replaces calculate_gradients(agent, snapshot, next_state, reward, test=True)
 * (mujoco/utils/backward.py:343-480) for N snapshots in ONE launch.  eps is
 * backward.py:13 (1e-6), nsteps is cfg.MODEL.NSTEPS_FOR_BACKWARD (backward.py:359).
 * With n = nq+nv:
 *   dsds [N,n,n]   rows = outputs (qpos',qvel'), cols = inputs (qpos,qvel)   (backward.py:475 "state")
 *   dsda [N,n,nu]                                                            (backward.py:475 "action")
 *   drds [N,n], drda [N,nu]                                                  (backward.py:478)
 *   s_next [N,n], reward [N]: the nominal rollout (the `test=True` re-step, backward.py:362-374)
 * Any output pointer may be NULL, status included: the results do not depend on which outputs are asked for (without
 * a status array a HalfCheetah launch re-runs units with more than eight floor contacts inside the rollout kernel
 * instead of in a fix-up launch; same bits, ~2 % slower). */
int mbrl_fd_jacobians(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel,
                      const double *ctrl, double eps, int fd_mode, int nsteps, double *dsds, double *dsda,
                      double *drds, double *drda, double *s_next, double *reward, uint32_t *status,
                      void *hip_stream);

/* The same with cfg.MODEL.POLICY.NETWORK = True (mujoco/utils/backward.py:418-419, 460-461): a rollout whose
 * qvel / qpos entry is perturbed runs under the policy's action AT the perturbed state, so the state
 * columns become total derivatives through the policy.  The caller evaluates the policy (a GEMM-shaped
 * batch, outside this library) and passes the actions:
 *   ctrl_pert [N, nv+nq, nu]      forward mode; row j < nv: qvel[j] + eps, row nv+j: qpos[j] + eps
 *   ctrl_pert [N, nv+nq, 2, nu]   central mode; [..,0,:] at +eps, [..,1,:] at -eps
 * (backward.py's own order: velocities first, then positions).  The nominal rollout and the ctrl-perturbed
 * rollouts use `ctrl`.  ctrl_pert == NULL is mbrl_fd_jacobians. */
int mbrl_fd_jacobians_closed_loop(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel,
                                  const double *ctrl, const double *ctrl_pert, double eps, int fd_mode, int nsteps,
                                  double *dsds, double *dsda, double *drds, double *drda, double *s_next,
                                  double *reward, uint32_t *status, void *hip_stream);

/* The same linearisation in two launches, as the reference splits it: the forward op runs the nominal step
 * (model/blocks/mujoco.py:20-31 -> mujoco/utils/forward.py:13-38) and calculate_gradients receives its next_state and
 * reward as arguments (mujoco/utils/backward.py:343) before it runs the perturbed copies (backward.py:376-470).
 *   mbrl_fd_plan               runs the N nominal rollouts (one lane each), keeps their results in `plan`, and sorts the
 *                              units by the control-flow signature of their rollout (contact solver passes per
 *                              mj_step), so that the lanes of a wave of the second launch take the same branches;
 *                              s_next [N,n], reward [N] (either may be NULL) and status [N] are written here.
 *   mbrl_fd_jacobians_planned  runs the perturbed rollouts only, in the planned order with every lane busy, differences
 *                              them against the nominal results held in `plan`, writes dsds / dsda / drds / drda (as
 *                              mbrl_fd_jacobians_closed_loop; ctrl_pert may be NULL) and ORs the perturbed rollouts'
 *                              status bits into status [N] (pass the array mbrl_fd_plan wrote, or NULL).
 * Both take the same N, qpos, qvel, ctrl, nsteps and the same stream (or streams ordered by the caller).  `plan` is
 * caller-owned device memory, 256-byte aligned, of at least mbrl_fd_plan_bytes(m, N) bytes (< 0: error); the library
 * allocates nothing.  A plan can be re-used for several mbrl_fd_jacobians_planned calls on the same inputs (forward and
 * central, open and closed loop).  Results are those of mbrl_fd_jacobians_closed_loop bit for bit.  A plan made for
 * another N, nsteps or model is recognised on the device: every unit's status gets MBRL_ST_BADPLAN, no rollout runs and
 * the Jacobian outputs are left untouched (with status == NULL the mismatch goes unreported). */
int64_t mbrl_fd_plan_bytes(const mbrl_model *m, int64_t N);
int mbrl_fd_plan(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl, int nsteps,
                 void *plan, int64_t plan_bytes, double *s_next, double *reward, uint32_t *status, void *hip_stream);
int mbrl_fd_jacobians_planned(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                              const double *ctrl_pert, double eps, int fd_mode, int nsteps, const void *plan,
                              int64_t plan_bytes, double *dsds, double *dsda, double *drds, double *drda,
                              uint32_t *status, void *hip_stream);

/* The plan launch folded into step launches that run anyway.  In the reference the forward op's env step IS the nominal
 * rollout calculate_gradients later receives (model/blocks/mujoco.py:20-31 -> mujoco/utils/backward.py:343); a batched
 * rollout runs those same steps as mbrl_step launches, so it can record the plan while it goes:
 *   mbrl_step_record     mbrl_step (same launch, same bits) that also leaves, for unit unit_offset + i of a plan made
 *                        for plan_units units, the control-flow key of its rollout and its next state in `plan`
 *                        (the N units of one time step of B episodes are units [t*B, t*B + B) of a time-major chunk);
 *   mbrl_fd_plan_finish  once every unit of the plan has been recorded: takes the rewards [plan_units] the step launches
 *                        wrote, writes the plan's header and sorts the units by key.
 * The finished plan is what mbrl_fd_plan would have left for the same (state, action) units with nsteps = 1, and
 * mbrl_fd_jacobians_planned runs on it unchanged -- no separate launch of nominal rollouts.  Same plan buffer rules as
 * mbrl_fd_plan (mbrl_fd_plan_bytes(m, plan_units), 256-byte aligned, caller-owned). */
int mbrl_step_record(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                     double *qpos_next, double *qvel_next, double *reward, uint32_t *status, void *plan, int64_t plan_bytes,
                     int64_t plan_units, int64_t unit_offset, void *hip_stream);
int mbrl_fd_plan_finish(const mbrl_model *m, int64_t plan_units, const double *reward, void *plan, int64_t plan_bytes,
                        void *hip_stream);

/* T consecutive mbrl_step_record calls in ONE launch, for rollouts whose actions are known ahead -- the reference's
 * default policy is an open-loop action sequence (cfg.MODEL.POLICY.NETWORK = False, model/blocks/policy/strategies.py:
 * 402-408), so the loop `for t: state, reward = model(state)` of model/engine/trainer.py:63-65 needs no state feedback.
 * Time-major buffers: qpos [T+1, B, nq] and qvel [T+1, B, nv] (slice 0 = the start states, step k reads slice k and
 * writes slice k + 1), ctrl [T, B, nu], reward [T, B], status [T, B] (may be NULL).  With plan != NULL step k records
 * its B units as units [unit_offset + k B, unit_offset + (k + 1) B) of a plan for plan_units units, exactly as
 * mbrl_step_record does (plan == NULL: mbrl_step).  A wave keeps its 64 episodes for all T steps, so the launch lasts
 * as long as the slowest wave's SUM over the steps rather than the sum of every step's slowest wave.  Results are those
 * of the T single-step launches bit for bit. */
int mbrl_rollout_record(const mbrl_model *m, int64_t B, int T, double *qpos, double *qvel, const double *ctrl, double *reward,
                        uint32_t *status, void *plan, int64_t plan_bytes, int64_t plan_units, int64_t unit_offset,
                        void *hip_stream);

/* NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc) and so cannot bring the plan workspace mbrl_fd_plan wants: the same
 * two-launch linearisation (mbrl_fd_plan then mbrl_fd_jacobians_planned, open loop), with the plan held in a static
 * device array inside the library -- MBRL_STATIC_SCRATCH_BYTES per GPU, held until the process ends; its device
 * address is looked up once per device in mbrl_model_create.  Inputs and outputs are the caller's, exactly as in
 * mbrl_fd_jacobians, and so are the results, bit for bit.  The call itself allocates nothing (no hipMalloc / hipFree,
 * no allocating library routine).  N > MBRL_STATIC_MAX_UNITS (or a plan beyond the array) is rejected with -8.
 * Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two threads) would RACE
 * on it -- serialise them, or use the caller-owned plan buffers of mbrl_fd_plan.  An overlap is detected on the device
 * (an in-flight counter in the array's header, bumped by the call's first launch and dropped by its last): every unit
 * of a call that ran while another was in flight gets MBRL_ST_BADPLAN in its status word instead of silently wrong
 * Jacobians (conservatively: also a call that started just before the other's last launch retired; with status == NULL
 * the overlap goes unreported).  Nothing else in the array needs clearing between calls (each call rewrites the header,
 * keys, order and nominal results of its own N units before reading them; the counter returns to zero). */
#define MBRL_STATIC_MAX_UNITS 32768
#define MBRL_STATIC_SCRATCH_BYTES (8u << 20)
int mbrl_fd_jacobians_static(const mbrl_model *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                             double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                             double *s_next, double *reward, uint32_t *status, void *hip_stream);

/* Replaces the chain of MjBlock.backward calls that autograd issues for one epoch
 * (model/blocks/mujoco.py:43-104; recurrence in SURVEY.md App. C): for each of B
 * episodes, t = H-1..0:
 *     da[b,t] = w_t * ( c[b,t]*drda[b,t] + g * dsda[b,t] ),  g <- c[b,t]*drds[b,t] + g * dsds[b,t]
 * Jacobians are indexed [b*H + t].  grad_weights: 0 = none, 1 = 'average' (w_t = 1/(H-t)),
 * 2 = 'prioritise' (running-sum weights, mujoco.py:46-47,57-58,64-65,72-74; the running sum
 * restarts at 0 for every call = trainer.py:76).  c [B,H] = dL/dreward; outputs da [B,H,nu]
 * and ds0 [B,n] (gradient w.r.t. the initial state; may be NULL). */
int mbrl_vjp_scan(const mbrl_model *m, int64_t B, int H, int grad_weights, const double *dsds,
                  const double *dsda, const double *drds, const double *drda, const double *c, double *da,
                  double *ds0, void *hip_stream);

/* The same with the unit index spelled out: MBRL_EPISODE_MAJOR, unit (b, t) at b*H + t (mbrl_vjp_scan), or
 * MBRL_TIME_MAJOR, at t*B + b -- the order in which a batched rollout produces its (s_t, a_t), so that the
 * linearisation of the first time steps can run (on another stream) while the later ones are still being simulated.
 * c, da and all four Jacobian arrays use the same order; ds0 is [B, n] either way. */
#define MBRL_EPISODE_MAJOR 0
#define MBRL_TIME_MAJOR 1
int mbrl_vjp_scan_layout(const mbrl_model *m, int64_t B, int H, int grad_weights, int layout, const double *dsds,
                         const double *dsda, const double *drds, const double *drda, const double *c, double *da,
                         double *ds0, void *hip_stream);

/* The same for ONE SHARD of a batch that is split over several GPUs (one process each; the reference is single-process,
 * model/engine/trainer.py:57-77): the B episodes of this call are episodes [episode_offset, episode_offset + B) of
 * total_episodes.  Only 'prioritise' depends on it -- its running sum carries over from episode to episode in the order
 * autograd visits them (last to first, model/blocks/mujoco.py:65), so a shard starts from what the later episodes of
 * the WHOLE batch have added; with (0, B) this is mbrl_vjp_scan_layout.  No communication: every rank computes its seed. */
int mbrl_vjp_scan_sharded(const mbrl_model *m, int64_t B, int H, int grad_weights, int layout, int64_t episode_offset,
                          int64_t total_episodes, const double *dsds, const double *dsda, const double *drds,
                          const double *drda, const double *c, double *da, double *ds0, void *hip_stream);

/* Introspection (host only, no GPU call; no counterpart in the reference): how the rollouts of a launch are laid
 * out over the 2 x 64 lanes of a wave pair for a model with the given dimensions.  fd_mode: MBRL_FD_FORWARD, MBRL_FD_CENTRAL, or -1 for
 * mbrl_step.  For lane i of the pair (i < 64: first wave): unit[i] = unit index within the pair (-1: idle lane),
 * rollout[i] = 0 for the nominal rollout, k >= 1 for the k-th perturbed one (forward: input k-1 in ctrl, qvel, qpos
 * order; central: input (k-1)/2, +eps for odd k, -eps for even k), ref_lane[i] = lane (within the same wave) of the
 * nominal rollout it is differenced against, primary[i] = 1 if this wave writes the unit's nominal outputs.
 * All four arrays have 128 entries. */
int mbrl_lane_map(int nq, int nv, int nu, int fd_mode, int *units_per_pair, int *unit, int *rollout, int *ref_lane,
                  int *primary);

const char *mbrl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
