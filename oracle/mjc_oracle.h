/*
 * mjc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT THE PRODUCT).
 *
 * Plain-C, fp64, general 3-D restatement of the arithmetic the reference's hot path
 * reaches through mujoco-py:  MuJoCo 2.0 `mj_step` (mujoco200 is NOT under
 * /root/reference: third-party, pinned by configs/Model-Based-RL.yml:196
 * `mujoco-py==2.0.2.5`, README.md:45) restricted to the feature set of the six MJCF
 * models in mujoco/assets/, plus the reference's own Python procedure around it:
 *   - env.step / reward formulas          mujoco/envs/<env>.py
 *   - calculate_gradients (FD schedule)   mujoco/utils/backward.py:343-480
 *
 * PARITY STATUS: the rigid-body physics is restated from MuJoCo's published
 * computation pipeline (no MuJoCo binary, header or source exists offline), so
 * numeric parity with mujoco-py is UNPINNED.  What is pinned (tests/):
 *   - closed-form Lagrangian known-answer vectors (sympy) for the cart-pole models;
 *   - physical invariants (energy, momentum), Taylor and determinism checks that the
 *     reference's own tests/test_gradients.py specifies;
 *   - the FD schedule / layout / op weighting against golden vectors produced by the
 *     reference's own Python (backward.py + blocks/mujoco.py) driven over this step.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef MJC_ORACLE_H
#define MJC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJO_MAXB 9      /* bodies incl. world */
#define MJO_MAXV 9      /* dofs (= nq for hinge/slide models) */
#define MJO_MAXG 10     /* capsule geoms */
#define MJO_MAXU 6
#define MJO_MAXCON 24   /* contacts */
#define MJO_MAXEFC (2 * MJO_MAXV + 4 * MJO_MAXCON)

/* status bits (per unit) */
#define MJO_ST_NONFINITE   1u   /* NaN/Inf in state or acceleration */
#define MJO_ST_NOCONVERGE  2u   /* constraint solver hit its iteration cap */
#define MJO_ST_OVERFLOW    4u   /* contact capacity exceeded */
#define MJO_ST_DEGENERATE  8u   /* reserved (crossing capsule axes now have a defined normal, see sphereSphere) */

/* ---- raw transcription (filled by models_raw_gen.h) ---- */
typedef struct {
    const char *name; int type;           /* 0 slide, 1 hinge */
    double axis[3], pos[3], ref, stiffness, damping, armature;
    int limited; double range[2], margin, solref[2], solimp[5];
} RawJoint;
typedef struct {
    double a[3], b[3], radius, density, friction;
    int condim, contype, conaffinity;
    double margin, solref[2], solimp[5];
} RawGeom;
typedef struct { int parent; double pos[3]; int njnt; RawJoint jnt[3]; int ngeom; RawGeom geom[2]; } RawBody;
typedef struct { int dof; double gear; int ctrllimited; double ctrlrange[2]; } RawAct;
typedef struct {
    const char *env; int plane; double timestep; int integrator; /* 0 Euler, 1 RK4 */
    double gravity[3], density, viscosity; int frame_skip; double settotalmass;
    int nbody; RawBody body[8];
    int has_floor; RawGeom floor;
    int selfcollide; int nu; RawAct act[MJO_MAXU];
    int has_site, site_body; double site_pos[3]; int reward;
} RawModel;

/* ---- public API (all return 0 on success, <0 on error) ---- */
int mjo_num_models(void);
const char *mjo_model_name(int id);
int mjo_model_id(const char *env);
int mjo_dims(int id, int *nq, int *nv, int *nu, double *timestep, int *frame_skip);
/* test hooks: override an option of model `id` and recompile ("density", "viscosity",
 * "gravity_x|y|z", "damping_scale", "stiffness_scale", "armature_scale", "reset") */
int mjo_set_option(int id, const char *key, double value);
/* compiled per-body constants, for cross-checking the planar compile */
int mjo_body_consts(int id, int body, double *mass, double *ipos3, double *inertia9,
                    double *invweight2);
int mjo_dof_invweight(int id, double *out_nv);

/* one forward-dynamics evaluation (mj_forward).  Any output may be NULL. */
int mjo_forward(int id, const double *qpos, const double *qvel, const double *ctrl,
                double *qacc, double *qacc_smooth, double *M /*nv*nv*/, double *qfrc_bias,
                double *qfrc_passive, double *qfrc_actuator, double *qfrc_constraint,
                int *nefc, double *efc_J /*MAXEFC*nv*/, double *efc_pos, double *efc_aref,
                double *efc_D, double *efc_force, int *niter, double *energy2 /*pot,kin*/,
                double *momentum6);

/* env.step(): frame_skip x mj_step + reward, batched over N units (row-major [N,*]). */
int mjo_step(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel,
             const double *ctrl, double *qpos_next, double *qvel_next, double *reward,
             uint32_t *status, int nthreads);

/* calculate_gradients() of backward.py:343-480, batched.
 * fd_mode 0 = forward differences (reference parity), 1 = central.
 * Layouts: dsds [N,n,n] rows=outputs(qpos',qvel') cols=inputs(qpos,qvel);
 * dsda [N,n,nu]; drds [N,n]; drda [N,nu]; s_next [N,n]; reward [N]. */
int mjo_fd_jacobians(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel,
                     const double *ctrl, double eps, int fd_mode, int nsteps, double *dsds,
                     double *dsda, double *drds, double *drda, double *s_next, double *reward,
                     uint32_t *status, int nthreads);

/* instrumentation: mean Newton iterations / active constraint statistics of the last call
 * (single-threaded calls only) */
/* mjo_fd_jacobians that also reports, per unit, how many dynamics evaluations of its rollouts had an active constraint
 * row (joint limit or contact); evals_constrained [N]; and how many capsule/capsule contacts were made by the
 * crossing-axes rule ("parity unpinned"); evals_crossing [N], may be NULL. */
int mjo_fd_jacobians_info(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                          double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                          double *s_next, double *reward, uint32_t *status, int32_t *evals_constrained,
                          int32_t *evals_crossing, int nthreads);
int mjo_stats(double *mean_newton_iters, double *frac_evals_constrained, double *mean_nefc);

#ifdef __cplusplus
}
#endif
#endif
