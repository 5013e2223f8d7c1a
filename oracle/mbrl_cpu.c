/*
 * mbrl_cpu.c -- the oracle behind the call shapes of include/mbrl.h (see mbrl_cpu.h).  TEST INFRASTRUCTURE ONLY.
 * Nothing here computes dynamics: step / fd_jacobians forward to mjc_oracle.c (mjo_step, mjo_fd_jacobians); the
 * backward recurrence is restated here from model/blocks/mujoco.py:43-104.
 */
#include "mbrl_cpu.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mjc_oracle.h"

struct mbrl_model_cpu { int id, nq, nv, nu, frame_skip; double timestep; };

static __thread char g_err[256];

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

static int threads(void) {
    const char *s = getenv("MBRL_CPU_THREADS");
    int n = s ? atoi(s) : 1;
    return n > 0 ? n : 1;
}

const char *mbrl_last_error_cpu(void) { return g_err; }
int mbrl_abi_version_cpu(void) { return 9; /* the include/mbrl.h version these call shapes follow */ }

int mbrl_model_create_cpu(const char *env_name, int frame_skip, int device, mbrl_model_cpu **out) {
    (void)device;
    if (!env_name || !out) return fail(-1, "mbrl_model_create_cpu: null argument");
    *out = NULL;
    int id = mjo_model_id(env_name);
    if (id < 0) return fail(-2, "mbrl_model_create_cpu: unknown env '%s'", env_name);
    mbrl_model_cpu *m = (mbrl_model_cpu *)calloc(1, sizeof *m);
    if (!m) return fail(-5, "mbrl_model_create_cpu: out of host memory");
    int fs = 0;
    if (mjo_dims(id, &m->nq, &m->nv, &m->nu, &m->timestep, &fs)) {
        free(m);
        return fail(-10, "mbrl_model_create_cpu: model compile failed");
    }
    m->id = id;
    m->frame_skip = frame_skip > 0 ? frame_skip : fs;   /* <= 0: the yaml's value, as mbrl_model_create */
    *out = m;
    return 0;
}

void mbrl_model_destroy_cpu(mbrl_model_cpu *m) { free(m); }

int mbrl_model_dims_cpu(const mbrl_model_cpu *m, int *nq, int *nv, int *nu, double *dt) {
    if (!m) return fail(-1, "mbrl_model_dims_cpu: null model");
    if (nq) *nq = m->nq;
    if (nv) *nv = m->nv;
    if (nu) *nu = m->nu;
    if (dt) *dt = m->timestep * m->frame_skip;
    return 0;
}

int mbrl_step_cpu(const mbrl_model_cpu *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                  double *qpos_next, double *qvel_next, double *reward, uint32_t *status, void *ignored) {
    (void)ignored;
    if (!m) return fail(-1, "mbrl_step_cpu: null model");
    if (N < 0) return fail(-6, "mbrl_step_cpu: negative N");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl || !qpos_next || !qvel_next) return fail(-1, "mbrl_step_cpu: null buffer");
    if (mjo_step(m->id, m->frame_skip, N, qpos, qvel, ctrl, qpos_next, qvel_next, reward, status, threads()))
        return fail(-10, "mbrl_step_cpu: oracle failed");
    return 0;
}

int mbrl_fd_jacobians_cpu(const mbrl_model_cpu *m, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                          double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                          double *s_next, double *reward, uint32_t *status, void *ignored) {
    (void)ignored;
    if (!m) return fail(-1, "mbrl_fd_jacobians_cpu: null model");
    if (N < 0) return fail(-6, "mbrl_fd_jacobians_cpu: negative N");
    if (N == 0) return 0;
    if (!qpos || !qvel || !ctrl) return fail(-1, "mbrl_fd_jacobians_cpu: null input buffer");
    if (!(eps > 0)) return fail(-7, "mbrl_fd_jacobians_cpu: eps must be > 0");
    if (fd_mode != 0 && fd_mode != 1) return fail(-7, "mbrl_fd_jacobians_cpu: bad fd_mode");
    if (mjo_fd_jacobians(m->id, m->frame_skip, N, qpos, qvel, ctrl, eps, fd_mode, nsteps, dsds, dsda, drds, drda, s_next,
                         reward, status, threads()))
        return fail(-10, "mbrl_fd_jacobians_cpu: oracle failed");
    return 0;
}

/* model/blocks/mujoco.py:43-104 as autograd drives it for one epoch (model/engine/trainer.py:57-77): episodes are
 * visited last to first, inside an episode t = H-1..0.  With g the gradient w.r.t. the state after step t:
 *   dynamics op (mujoco.py:52-66):  d action = w_d * g . ds'/da,   d state = g . ds'/ds
 *   reward   op (mujoco.py:68-104): d action = w_r * c * dr/da,    d state = w_s * c * dr/ds
 * 'average' (mujoco.py:54-55, 70-71): w_d = w_r = 1/W, W = H - t.  'prioritise' (mujoco.py:46-47, 57-58, 64-65, 72-74):
 * a running sum S over every op of the epoch, S += W before step t's ops use it; w_s = W - 1, w_r = W / S, w_d = 1 / S. */
int mbrl_vjp_scan_sharded_cpu(const mbrl_model_cpu *m, int64_t B, int H, int grad_weights, int layout,
                              int64_t episode_offset, int64_t total_episodes, const double *dsds, const double *dsda,
                              const double *drds, const double *drda, const double *c, double *da, double *ds0,
                              void *ignored) {
    (void)ignored;
    if (!m) return fail(-1, "mbrl_vjp_scan_cpu: null model");
    if (episode_offset < 0 || total_episodes < episode_offset + B)
        return fail(-6, "mbrl_vjp_scan_sharded_cpu: the shard [offset, offset + B) must lie inside [0, total_episodes)");
    if (layout != 0 && layout != 1) return fail(-7, "mbrl_vjp_scan_cpu: bad layout");
    if (B < 0 || H < 0) return fail(-6, "mbrl_vjp_scan_cpu: negative size");
    if (B == 0 || H == 0) return 0;
    if (!dsds || !dsda || !drds || !drda || !c || !da) return fail(-1, "mbrl_vjp_scan_cpu: null buffer");
    if (grad_weights < 0 || grad_weights > 2) return fail(-7, "mbrl_vjp_scan_cpu: bad grad_weights");
    const int n = m->nq + m->nv, nu = m->nu;
    /* what the episodes after this shard have already added to the running sum: H + (H-1) + ... + 1 each */
    double S = (double)(total_episodes - episode_offset - B) * (0.5 * (double)H * (double)(H + 1));
    double g[2 * MJO_MAXV], gn[2 * MJO_MAXV];
    for (int64_t b = B - 1; b >= 0; b--) {
        memset(g, 0, sizeof g);
        for (int t = H - 1; t >= 0; t--) {
            const int64_t k = layout == 1 ? (int64_t)t * B + b : b * H + t;
            const double W = (double)(H - t), ct = c[k];
            double w_r = 1.0, w_d = 1.0, w_s = 1.0;
            if (grad_weights == 1) w_r = w_d = 1.0 / W;
            else if (grad_weights == 2) { S += W; w_s = W - 1.0; w_r = W / S; w_d = 1.0 / S; }
            for (int j = 0; j < nu; j++) {
                double acc = 0.0;
                for (int o = 0; o < n; o++) acc += g[o] * (w_d * dsda[(k * n + o) * nu + j]);
                da[k * nu + j] = ct * (w_r * drda[k * nu + j]) + acc;
            }
            for (int j = 0; j < n; j++) {
                double acc = 0.0;
                for (int o = 0; o < n; o++) acc += g[o] * dsds[(k * n + o) * n + j];
                gn[j] = ct * (w_s * drds[k * n + j]) + acc;
            }
            memcpy(g, gn, n * sizeof(double));
        }
        if (ds0) memcpy(ds0 + b * n, g, n * sizeof(double));
    }
    return 0;
}

int mbrl_vjp_scan_layout_cpu(const mbrl_model_cpu *m, int64_t B, int H, int grad_weights, int layout, const double *dsds,
                             const double *dsda, const double *drds, const double *drda, const double *c, double *da,
                             double *ds0, void *ignored) {
    return mbrl_vjp_scan_sharded_cpu(m, B, H, grad_weights, layout, 0, B, dsds, dsda, drds, drda, c, da, ds0, ignored);
}

int mbrl_vjp_scan_cpu(const mbrl_model_cpu *m, int64_t B, int H, int grad_weights, const double *dsds, const double *dsda,
                      const double *drds, const double *drda, const double *c, double *da, double *ds0, void *ignored) {
    return mbrl_vjp_scan_sharded_cpu(m, B, H, grad_weights, 0, 0, B, dsds, dsda, drds, drda, c, da, ds0, ignored);
}
