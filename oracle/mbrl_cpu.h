/*
 * mbrl_cpu.h -- the CPU oracle behind the SAME call shapes as include/mbrl.h ("identical *_cpu entry
 * points for the oracle twin"), so that a parity test issues one call sequence twice.
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker.  It is part of libmjc_oracle.so, never of libmbrl_hip.so.
 *
 * Differences from include/mbrl.h, all forced by running on the host:
 *   - every buffer is HOST memory; the last argument (`hip_stream` there) is ignored and the call is synchronous;
 *   - `device` of mbrl_model_create_cpu is ignored;
 *   - MBRL_CPU_THREADS in the environment (default 1) sets the OpenMP threads of step / fd_jacobians.
 * Same argument meaning, same layouts, same return codes for the same mistakes (-1 null, -2 unknown env, -6 size,
 * -7 value), same status bits.  Each entry cites the reference code it restates through mjc_oracle.c.
 */
#ifndef MBRL_CPU_H
#define MBRL_CPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbrl_model_cpu mbrl_model_cpu;

int mbrl_abi_version_cpu(void);
/* build_agent(cfg) -> env ctor, mujoco/build.py:6-37 */
int mbrl_model_create_cpu(const char *env_name, int frame_skip, int device, mbrl_model_cpu **out);
void mbrl_model_destroy_cpu(mbrl_model_cpu *m);
int mbrl_model_dims_cpu(const mbrl_model_cpu *m, int *nq, int *nv, int *nu, double *dt);
/* mj_forward() -> agent.step(action), mujoco/utils/forward.py:13-38 */
int mbrl_step_cpu(const mbrl_model_cpu *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                  double *qpos_next, double *qvel_next, double *reward, uint32_t *status, void *ignored);
/* calculate_gradients(..., test=True), mujoco/utils/backward.py:343-480 */
int mbrl_fd_jacobians_cpu(const mbrl_model_cpu *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                          double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                          double *s_next, double *reward, uint32_t *status, void *ignored);
/* the chain of MjBlock.backward calls of one epoch, model/blocks/mujoco.py:43-104 */
int mbrl_vjp_scan_cpu(const mbrl_model_cpu *m, int64_t B, int H, int grad_weights, const double *dsds, const double *dsda,
                      const double *drds, const double *drda, const double *c, double *da, double *ds0, void *ignored);
int mbrl_vjp_scan_layout_cpu(const mbrl_model_cpu *m, int64_t B, int H, int grad_weights, int layout, const double *dsds,
                             const double *dsda, const double *drds, const double *drda, const double *c, double *da,
                             double *ds0, void *ignored);
int mbrl_vjp_scan_sharded_cpu(const mbrl_model_cpu *m, int64_t B, int H, int grad_weights, int layout,
                              int64_t episode_offset, int64_t total_episodes, const double *dsds, const double *dsda,
                              const double *drds, const double *drda, const double *c, double *da, double *ds0,
                              void *ignored);
const char *mbrl_last_error_cpu(void);

#ifdef __cplusplus
}
#endif
#endif
