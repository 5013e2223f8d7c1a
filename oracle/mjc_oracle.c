/*
 * mjc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT THE PRODUCT).  See mjc_oracle.h.
 *
 * General 3-D restatement of MuJoCo 2.0's mj_step for hinge/slide trees with capsule
 * geoms, a ground plane, joint limits, pyramidal contacts, inertia-box fluid forces and
 * the Euler (implicit joint damping) / RK4 integrators, followed by the reference's own
 * env.step reward formulas and its finite-difference procedure.
 *
 * Source of each algorithm (MuJoCo itself is absent from /root/reference; everything
 * marked [MJ] is restated from MuJoCo 2.0's published computation pipeline):
 *   kinematics, comPos, crb, rne, comVel, passive, actuation   [MJ] engine_core_smooth / engine_passive
 *   collision plane-capsule, capsule-capsule                    [MJ] engine_collision_primitive
 *   makeConstraint, impedance, reference acceleration           [MJ] engine_core_constraint
 *   primal Newton solver with exact line search                 [MJ] engine_solver (restated: run to the
 *         exact active-set fixed point; the reference sets opt.tolerance=0, mujoco/build.py:35)
 *   Euler / RK4                                                 [MJ] engine_forward
 *   env.step rewards        mujoco/envs/{half_cheetah.py:24-33, hopper.py:27-46, walker2d.py:22-36,
 *                           swimmer.py:22-32, inverted_pendulum.py:24-38, inverted_double_pendulum.py:25-35}
 *   FD schedule             mujoco/utils/backward.py:343-480
 */
#include "mjc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "models_raw_gen.h"

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define PI 3.14159265358979323846

/* ------------------------------------------------------------------------------------ */
/* compiled model                                                                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const RawModel *raw;
    int nb, nv, nu, ng;
    int parent[MJO_MAXB], rootid[MJO_MAXB];
    double bpos[MJO_MAXB][3], ipos[MJO_MAXB][3], inertia[MJO_MAXB][9], mass[MJO_MAXB];
    int jadr[MJO_MAXB], jnum[MJO_MAXB];
    int jtype[MJO_MAXV], jbody[MJO_MAXV], dof_parent[MJO_MAXV];
    double jpos[MJO_MAXV][3], jaxis[MJO_MAXV][3], qpos0[MJO_MAXV];
    double stiff[MJO_MAXV], damp[MJO_MAXV], arma[MJO_MAXV];
    int limited[MJO_MAXV];
    double range[MJO_MAXV][2], jmargin[MJO_MAXV], jsolref[MJO_MAXV][2], jsolimp[MJO_MAXV][5];
    int gbody[MJO_MAXG];
    RawGeom geom[MJO_MAXG];
    int has_floor;
    RawGeom floor;
    int selfcollide;
    int act_dof[MJO_MAXU], ctrllimited[MJO_MAXU];
    double gear[MJO_MAXU], ctrlrange[MJO_MAXU][2];
    double timestep, gravity[3], density, viscosity;
    int integrator, frame_skip, any_damping;
    double dof_invweight0[MJO_MAXV], body_invweight0[MJO_MAXB][2];
    double box[MJO_MAXB][3];
    int has_site, site_body;
    double site_pos[3];
    int reward;
    /* test overrides */
    double ov_density, ov_viscosity, ov_gravity[3], damping_scale, stiffness_scale, armature_scale;
    int ov_set;
    int compiled;
} Model;

typedef struct {
    double pos[3], frame[9]; /* frame rows: normal, t1, t2 */
    double dist, includemargin, mu;
    int dim, body1, body2;
    double solref[2], solimp[5];
} Contact;

typedef struct {
    double qpos[MJO_MAXV], qvel[MJO_MAXV], ctrl[MJO_MAXU];
    double xpos[MJO_MAXB][3], xmat[MJO_MAXB][9], xipos[MJO_MAXB][3];
    double xanchor[MJO_MAXV][3], xaxis[MJO_MAXV][3];
    double com[MJO_MAXB][3]; /* subtree com of each body's tree root (indexed by body) */
    double cinert[MJO_MAXB][13]; /* I_O (9), m*c (3), m */
    double cdof[MJO_MAXV][6], cdof_dot[MJO_MAXV][6], cvel[MJO_MAXB][6];
    double M[MJO_MAXV][MJO_MAXV];
    double qfrc_bias[MJO_MAXV], qfrc_passive[MJO_MAXV], qfrc_actuator[MJO_MAXV];
    double qfrc_smooth[MJO_MAXV], qacc_smooth[MJO_MAXV], qacc[MJO_MAXV], qfrc_constraint[MJO_MAXV];
    int ncon;
    Contact con[MJO_MAXCON];
    int nefc;
    double J[MJO_MAXEFC][MJO_MAXV], efc_pos[MJO_MAXEFC], efc_margin[MJO_MAXEFC];
    double efc_D[MJO_MAXEFC], efc_aref[MJO_MAXEFC], efc_force[MJO_MAXEFC];
    double site_xpos[3];
    int niter;
    unsigned status;
    /* stats */
    long n_eval, n_eval_con, sum_iter, sum_nefc, n_cross;   /* n_cross: contacts made by the crossing-axes rule */
} Data;

static Model g_models[MJO_NUM_RAW_MODELS];

/* ------------------------------------------------------------------------------------ */
/* small vector helpers                                                                 */
/* ------------------------------------------------------------------------------------ */
static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double *r, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulMatVec3(double *r, const double *m, const double *v) {
    double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulMatTVec3(double *r, const double *m, const double *v) {
    double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
    double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
    double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static void mulMat3(double *r, const double *a, const double *b) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    memcpy(r, t, sizeof t);
}
static void mulMat3T(double *r, const double *a, const double *b) { /* a * b^T */
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
    memcpy(r, t, sizeof t);
}
/* Rodrigues rotation matrix about unit axis by angle */
static void axisAngle2Mat(double *R, const double *ax, double ang) {
    double c = cos(ang), s = sin(ang), t = 1.0 - c;
    double x = ax[0], y = ax[1], z = ax[2];
    R[0] = t * x * x + c;     R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
    R[3] = t * x * y + s * z; R[4] = t * y * y + c;     R[5] = t * y * z - s * x;
    R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}

/* spatial (6-D, [ang; lin]) helpers, reference point = subtree com */
static void mulInertVec(double *r, const double *ci, const double *v) {
    /* ci: I_O (9), mc (3), m.  r_ang = I w + mc x v ; r_lin = m v - mc x w */
    double Iw[3], t[3];
    mulMatVec3(Iw, ci, v);
    cross3(t, ci + 9, v + 3);
    r[0] = Iw[0] + t[0]; r[1] = Iw[1] + t[1]; r[2] = Iw[2] + t[2];
    cross3(t, ci + 9, v);
    r[3] = ci[12] * v[3] - t[0]; r[4] = ci[12] * v[4] - t[1]; r[5] = ci[12] * v[5] - t[2];
}
static void crossMotion(double *r, const double *vel, const double *v) {
    double a[3], b[3], c[3];
    cross3(a, vel, v);
    cross3(b, vel, v + 3);
    cross3(c, vel + 3, v);
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void crossForce(double *r, const double *vel, const double *f) {
    double a[3], b[3], c[3];
    cross3(a, vel, f);
    cross3(b, vel + 3, f + 3);
    cross3(c, vel, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}

/* dense Cholesky (lower) of n x n SPD matrix stored [MJO_MAXV][MJO_MAXV]; returns 0 ok */
static int chol_factor(double L[MJO_MAXV][MJO_MAXV], int n) {
    for (int j = 0; j < n; j++) {
        double s = L[j][j];
        for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        if (!(s > 0)) return -1;
        double d = sqrt(s);
        L[j][j] = d;
        for (int i = j + 1; i < n; i++) {
            double t = L[i][j];
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t / d;
        }
    }
    return 0;
}
static void chol_solve(double L[MJO_MAXV][MJO_MAXV], int n, double *x) {
    for (int i = 0; i < n; i++) {
        double t = x[i];
        for (int k = 0; k < i; k++) t -= L[i][k] * x[k];
        x[i] = t / L[i][i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double t = x[i];
        for (int k = i + 1; k < n; k++) t -= L[k][i] * x[k];
        x[i] = t / L[i][i];
    }
}

/* ------------------------------------------------------------------------------------ */
/* position stage                                                                       */
/* ------------------------------------------------------------------------------------ */
static void kinematics(const Model *m, Data *d) {
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memset(d->xpos[0], 0, sizeof d->xpos[0]);
    memcpy(d->xmat[0], I3, sizeof I3);
    memset(d->xipos[0], 0, sizeof d->xipos[0]);
    for (int b = 1; b < m->nb; b++) {
        int p = m->parent[b];
        double t[3];
        mulMatVec3(t, d->xmat[p], m->bpos[b]);
        for (int k = 0; k < 3; k++) d->xpos[b][k] = d->xpos[p][k] + t[k];
        memcpy(d->xmat[b], d->xmat[p], sizeof I3);
        for (int j = m->jadr[b]; j < m->jadr[b] + m->jnum[b]; j++) {
            mulMatVec3(t, d->xmat[b], m->jpos[j]);
            for (int k = 0; k < 3; k++) d->xanchor[j][k] = d->xpos[b][k] + t[k];
            mulMatVec3(d->xaxis[j], d->xmat[b], m->jaxis[j]);
            double dq = d->qpos[j] - m->qpos0[j];
            if (m->jtype[j] == 0) {
                for (int k = 0; k < 3; k++) d->xpos[b][k] += d->xaxis[j][k] * dq;
            } else {
                double R[9];
                axisAngle2Mat(R, m->jaxis[j], dq);
                mulMat3(d->xmat[b], d->xmat[b], R);
                mulMatVec3(t, d->xmat[b], m->jpos[j]);
                for (int k = 0; k < 3; k++) d->xpos[b][k] = d->xanchor[j][k] - t[k];
            }
        }
        mulMatVec3(t, d->xmat[b], m->ipos[b]);
        for (int k = 0; k < 3; k++) d->xipos[b][k] = d->xpos[b][k] + t[k];
    }
    if (m->has_site) {
        double t[3];
        mulMatVec3(t, d->xmat[m->site_body], m->site_pos);
        for (int k = 0; k < 3; k++) d->site_xpos[k] = d->xpos[m->site_body][k] + t[k];
    }
}

static void comPos(const Model *m, Data *d) {
    double msum[MJO_MAXB], csum[MJO_MAXB][3];
    memset(msum, 0, sizeof msum);
    memset(csum, 0, sizeof csum);
    for (int b = 1; b < m->nb; b++) {
        int r = m->rootid[b];
        msum[r] += m->mass[b];
        for (int k = 0; k < 3; k++) csum[r][k] += m->mass[b] * d->xipos[b][k];
    }
    for (int b = 1; b < m->nb; b++) {
        int r = m->rootid[b];
        for (int k = 0; k < 3; k++) d->com[b][k] = csum[r][k] / (msum[r] > MINVAL ? msum[r] : MINVAL);
    }
    for (int b = 1; b < m->nb; b++) {
        double RI[9], Iw[9], c[3];
        mulMat3(RI, d->xmat[b], m->inertia[b]);
        mulMat3T(Iw, RI, d->xmat[b]);
        for (int k = 0; k < 3; k++) c[k] = d->xipos[b][k] - d->com[b][k];
        double mm = m->mass[b], cc = dot3(c, c);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                d->cinert[b][3 * i + j] = Iw[3 * i + j] + mm * ((i == j ? cc : 0.0) - c[i] * c[j]);
        for (int k = 0; k < 3; k++) d->cinert[b][9 + k] = mm * c[k];
        d->cinert[b][12] = mm;
    }
    for (int j = 0; j < m->nv; j++) {
        int b = m->jbody[j];
        if (m->jtype[j] == 1) {
            double off[3];
            for (int k = 0; k < 3; k++) {
                d->cdof[j][k] = d->xaxis[j][k];
                off[k] = d->com[b][k] - d->xanchor[j][k];
            }
            cross3(d->cdof[j] + 3, d->xaxis[j], off);
        } else {
            for (int k = 0; k < 3; k++) {
                d->cdof[j][k] = 0.0;
                d->cdof[j][3 + k] = d->xaxis[j][k];
            }
        }
    }
}

static void crb(const Model *m, Data *d) {
    double c[MJO_MAXB][13];
    memcpy(c, d->cinert, sizeof c);
    for (int b = m->nb - 1; b >= 1; b--) {
        int p = m->parent[b];
        if (p > 0)
            for (int k = 0; k < 13; k++) c[p][k] += c[b][k];
    }
    memset(d->M, 0, sizeof d->M);
    for (int i = 0; i < m->nv; i++) {
        double buf[6];
        mulInertVec(buf, c[m->jbody[i]], d->cdof[i]);
        for (int j = i; j >= 0; j = m->dof_parent[j]) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += d->cdof[j][k] * buf[k];
            d->M[i][j] = s;
            d->M[j][i] = s;
        }
        d->M[i][i] += m->arma[i] * m->armature_scale;
    }
}

/* translational / rotational Jacobian of a world point attached to `body` */
static void jac(const Model *m, const Data *d, int body, const double *point, double jacp[3][MJO_MAXV],
                double jacr[3][MJO_MAXV]) {
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < m->nv; j++) {
            jacp[k][j] = 0;
            if (jacr) jacr[k][j] = 0;
        }
    for (int b = body; b > 0; b = m->parent[b]) {
        for (int j = m->jadr[b]; j < m->jadr[b] + m->jnum[b]; j++) {
            if (m->jtype[j] == 1) {
                double off[3], t[3];
                for (int k = 0; k < 3; k++) off[k] = point[k] - d->xanchor[j][k];
                cross3(t, d->xaxis[j], off);
                for (int k = 0; k < 3; k++) {
                    jacp[k][j] = t[k];
                    if (jacr) jacr[k][j] = d->xaxis[j][k];
                }
            } else {
                for (int k = 0; k < 3; k++) jacp[k][j] = d->xaxis[j][k];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* collision                                                                            */
/* ------------------------------------------------------------------------------------ */
static void makeFrame(double *frame) {
    /* frame[0..2] = unit normal; build two tangents [MJ mju_makeFrame] */
    double tmp[3] = {0, 0, 0};
    if (frame[1] < 0.5 && frame[1] > -0.5) tmp[1] = 1; else tmp[2] = 1;
    double dd = dot3(tmp, frame);
    double t1[3] = {tmp[0] - dd * frame[0], tmp[1] - dd * frame[1], tmp[2] - dd * frame[2]};
    double n = sqrt(dot3(t1, t1));
    for (int k = 0; k < 3; k++) frame[3 + k] = t1[k] / n;
    cross3(frame + 6, frame, frame + 3);
}

static void mixParams(Contact *c, const RawGeom *g1, const RawGeom *g2) {
    /* equal priority, solmix 1:1 -> plain average; friction/condim/margin = max [MJ] */
    c->mu = g1->friction > g2->friction ? g1->friction : g2->friction;
    c->dim = g1->condim > g2->condim ? g1->condim : g2->condim;
    for (int k = 0; k < 2; k++) c->solref[k] = 0.5 * (g1->solref[k] + g2->solref[k]);
    for (int k = 0; k < 5; k++) c->solimp[k] = 0.5 * (g1->solimp[k] + g2->solimp[k]);
    c->includemargin = g1->margin > g2->margin ? g1->margin : g2->margin; /* gap = 0 */
}

static void addContact(Data *d, const Contact *c) {
    if (d->ncon >= MJO_MAXCON) { d->status |= MJO_ST_OVERFLOW; return; }
    d->con[d->ncon++] = *c;
}

static void geomEnds(const Model *m, const Data *d, int g, double *pa, double *pb) {
    int b = m->gbody[g];
    double t[3];
    mulMatVec3(t, d->xmat[b], m->geom[g].a);
    for (int k = 0; k < 3; k++) pa[k] = d->xpos[b][k] + t[k];
    mulMatVec3(t, d->xmat[b], m->geom[g].b);
    for (int k = 0; k < 3; k++) pb[k] = d->xpos[b][k] + t[k];
}

/* closest points between segments p1 + s*d1, p2 + t*d2, s,t in [0,1] */
static void segseg(const double *p1, const double *d1, const double *p2, const double *d2, double *s_out,
                   double *t_out, int *parallel) {
    double r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), c = dot3(d1, r), b = dot3(d1, d2);
    double den = a * e - b * b, s, t;
    *parallel = den < 1e-12 * a * e;
    s = *parallel ? 0.0 : (b * f - c * e) / den;
    if (s < 0) s = 0; else if (s > 1) s = 1;
    t = (b * s + f) / e;
    if (t < 0) { t = 0; s = -c / a; if (s < 0) s = 0; else if (s > 1) s = 1; }
    else if (t > 1) { t = 1; s = (b - c) / a; if (s < 0) s = 0; else if (s > 1) s = 1; }
    *s_out = s; *t_out = t;
}

/* ax1, ax2: the two capsule axes (any length).  Coinciding centres (crossing axes): normal = cross product of the axes,
 * or (1,0,0) when they are parallel [MJ mjraw_SphereSphere, from knowledge].  The rule applies below 1e-10 (MuJoCo: 1e-15
 * and a normalised round-off vector in between, which no second implementation can reproduce); the kernel does the same
 * (planar_lean.hpp: for a planar model the cross product points out of the plane, so the row acts on no dof). */
static void sphereSphere(Data *d, const Contact *proto, const double *c1, double r1, const double *c2, double r2,
                         int b1, int b2, const double *ax1, const double *ax2) {
    double dif[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
    double len = sqrt(dot3(dif, dif));
    double dist = len - r1 - r2;
    if (dist >= proto->includemargin) return;
    Contact c = *proto;
    if (len < 1e-10) {
        d->n_cross++;
        double cr[3];
        cross3(cr, ax1, ax2);
        double n2 = dot3(cr, cr), l2 = dot3(ax1, ax1) * dot3(ax2, ax2);
        if (n2 >= MINVAL * MINVAL * l2) { double n = sqrt(n2); for (int k = 0; k < 3; k++) c.frame[k] = cr[k] / n; }
        else { c.frame[0] = 1; c.frame[1] = 0; c.frame[2] = 0; }
    }
    else for (int k = 0; k < 3; k++) c.frame[k] = dif[k] / len;
    for (int k = 0; k < 3; k++) c.pos[k] = c1[k] + c.frame[k] * (r1 + 0.5 * dist);
    makeFrame(c.frame);
    c.dist = dist; c.body1 = b1; c.body2 = b2;
    addContact(d, &c);
}

static void collision(const Model *m, Data *d) {
    d->ncon = 0;
    if (m->has_floor) {
        for (int g = 0; g < m->ng; g++) {
            const RawGeom *G = &m->geom[g];
            if (!((G->contype & m->floor.conaffinity) || (m->floor.contype & G->conaffinity))) continue;
            double e[2][3];
            geomEnds(m, d, g, e[0], e[1]);
            Contact proto;
            memset(&proto, 0, sizeof proto);
            mixParams(&proto, &m->floor, G);
            for (int s = 0; s < 2; s++) {
                double dist = (e[s][2] - m->floor.a[2]) - G->radius;
                if (dist >= proto.includemargin) continue;
                Contact c = proto;
                c.frame[0] = 0; c.frame[1] = 0; c.frame[2] = 1;
                makeFrame(c.frame);
                c.pos[0] = e[s][0]; c.pos[1] = e[s][1];
                c.pos[2] = e[s][2] - (G->radius + 0.5 * dist);
                c.dist = dist; c.body1 = 0; c.body2 = m->gbody[g];
                addContact(d, &c);
            }
        }
    }
    if (m->selfcollide) {
        for (int g1 = 0; g1 < m->ng; g1++)
            for (int g2 = g1 + 1; g2 < m->ng; g2++) {
                int b1 = m->gbody[g1], b2 = m->gbody[g2];
                if (b1 == b2) continue;
                /* parent-child filter (parent != world) [MJ filterparent] */
                if ((m->parent[b1] == b2 && b2 != 0) || (m->parent[b2] == b1 && b1 != 0)) continue;
                const RawGeom *G1 = &m->geom[g1], *G2 = &m->geom[g2];
                if (!((G1->contype & G2->conaffinity) || (G2->contype & G1->conaffinity))) continue;
                double a1[3], e1[3], a2[3], e2[3], d1[3], d2[3];
                geomEnds(m, d, g1, a1, e1);
                geomEnds(m, d, g2, a2, e2);
                for (int k = 0; k < 3; k++) { d1[k] = e1[k] - a1[k]; d2[k] = e2[k] - a2[k]; }
                Contact proto;
                memset(&proto, 0, sizeof proto);
                mixParams(&proto, G1, G2);
                double s, t; int par;
                segseg(a1, d1, a2, d2, &s, &t, &par);
                double c1[3], c2[3];
                for (int k = 0; k < 3; k++) { c1[k] = a1[k] + s * d1[k]; c2[k] = a2[k] + t * d2[k]; }
                sphereSphere(d, &proto, c1, G1->radius, c2, G2->radius, b1, b2, d1, d2);
            }
    }
}

/* ------------------------------------------------------------------------------------ */
/* constraints                                                                          */
/* ------------------------------------------------------------------------------------ */
static double impedance(const double *solimp_in, double pos, double margin) {
    double dmin = fmin(MAXIMP, fmax(MINIMP, solimp_in[0]));
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp_in[1]));
    double width = fmax(0.0, solimp_in[2]);
    double mid = fmin(MAXIMP, fmax(MINIMP, solimp_in[3]));
    double power = fmax(1.0, solimp_in[4]);
    if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
    double x = fabs((pos - margin) / width), y;
    if (x >= 1.0) return dmax;
    if (x <= 0.0) return dmin;
    if (power == 1.0) y = x;
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1.0);
    else y = 1.0 - pow(1.0 - x, power) / pow(1.0 - mid, power - 1.0);
    return dmin + y * (dmax - dmin);
}

/* fills K, B (stiffness/damping of the reference acceleration) and imp */
static void kbi(const Model *m, const double *solref_in, const double *solimp, double pos, double margin,
                double *K, double *B, double *imp) {
    double sr0 = solref_in[0], sr1 = solref_in[1];
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
    if (sr0 > 0) {
        sr0 = fmax(sr0, 2.0 * m->timestep); /* refsafe */
        *K = 1.0 / fmax(MINVAL, dmax * dmax * sr0 * sr0 * sr1 * sr1);
        *B = 2.0 / fmax(MINVAL, dmax * sr0);
    } else {
        *K = -sr0 / fmax(MINVAL, dmax * dmax);
        *B = -sr1 / fmax(MINVAL, dmax);
    }
    *imp = impedance(solimp, pos, margin);
}

static void makeConstraint(const Model *m, Data *d) {
    int n = 0, nv = m->nv;
    /* joint limits: lower side then upper side */
    for (int j = 0; j < nv; j++) {
        if (!m->limited[j]) continue;
        for (int side = -1; side <= 1; side += 2) {
            double lim = side < 0 ? m->range[j][0] : m->range[j][1];
            double dist = side * (lim - d->qpos[j]);
            if (dist >= m->jmargin[j]) continue;
            memset(d->J[n], 0, sizeof d->J[n]);
            d->J[n][j] = -side;
            double K, B, imp;
            kbi(m, m->jsolref[j], m->jsolimp[j], dist, m->jmargin[j], &K, &B, &imp);
            double R = fmax(MINVAL, (1 - imp) * m->dof_invweight0[j] / imp);
            d->efc_pos[n] = dist; d->efc_margin[n] = m->jmargin[j];
            d->efc_D[n] = 1.0 / R;
            d->efc_aref[n] = -B * (-side * d->qvel[j]) - K * imp * (dist - m->jmargin[j]);
            n++;
        }
    }
    /* contacts */
    for (int ci = 0; ci < d->ncon; ci++) {
        Contact *c = &d->con[ci];
        double jp1[3][MJO_MAXV], jp2[3][MJO_MAXV], Jc[3][MJO_MAXV];
        jac(m, d, c->body1, c->pos, jp1, NULL);
        jac(m, d, c->body2, c->pos, jp2, NULL);
        for (int r = 0; r < 3; r++)
            for (int j = 0; j < nv; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += c->frame[3 * r + k] * (jp2[k][j] - jp1[k][j]);
                Jc[r][j] = s;
            }
        double K, B, imp;
        kbi(m, c->solref, c->solimp, c->dist, c->includemargin, &K, &B, &imp);
        double tran = m->body_invweight0[c->body1][0] + m->body_invweight0[c->body2][0];
        int nrow = c->dim == 1 ? 1 : 2 * (c->dim - 1);
        double R;
        if (c->dim == 1) {
            R = fmax(MINVAL, (1 - imp) * tran / imp);
        } else {
            double dA = tran + c->mu * c->mu * tran;
            double R0 = fmax(MINVAL, (1 - imp) * dA / imp);
            R = 2.0 * c->mu * c->mu * R0; /* pyramidal rescale, impratio = 1 [MJ] */
        }
        for (int r = 0; r < nrow; r++) {
            if (n >= MJO_MAXEFC) { d->status |= MJO_ST_OVERFLOW; break; }
            if (c->dim == 1) memcpy(d->J[n], Jc[0], sizeof d->J[n]);
            else {
                int t = 1 + r / 2;
                double sg = (r % 2 == 0) ? c->mu : -c->mu;
                for (int j = 0; j < nv; j++) d->J[n][j] = Jc[0][j] + sg * Jc[t][j];
            }
            double vel = 0;
            for (int j = 0; j < nv; j++) vel += d->J[n][j] * d->qvel[j];
            d->efc_pos[n] = c->dist; d->efc_margin[n] = c->includemargin;
            d->efc_D[n] = 1.0 / R;
            d->efc_aref[n] = -B * vel - K * imp * (c->dist - c->includemargin);
            n++;
        }
    }
    d->nefc = n;
}

/* ------------------------------------------------------------------------------------ */
/* velocity / force stages                                                              */
/* ------------------------------------------------------------------------------------ */
static void comVel(const Model *m, Data *d) {
    memset(d->cvel[0], 0, sizeof d->cvel[0]);
    for (int b = 1; b < m->nb; b++) {
        double v[6];
        memcpy(v, d->cvel[m->parent[b]], sizeof v);
        for (int j = m->jadr[b]; j < m->jadr[b] + m->jnum[b]; j++) {
            crossMotion(d->cdof_dot[j], v, d->cdof[j]);
            for (int k = 0; k < 6; k++) v[k] += d->cdof[j][k] * d->qvel[j];
        }
        memcpy(d->cvel[b], v, sizeof v);
    }
}

static void rne_bias(const Model *m, Data *d) {
    double cacc[MJO_MAXB][6], cfrc[MJO_MAXB][6];
    const double *g = m->ov_set ? m->ov_gravity : m->gravity;
    for (int k = 0; k < 3; k++) { cacc[0][k] = 0; cacc[0][3 + k] = -g[k]; }
    memset(cfrc[0], 0, sizeof cfrc[0]);
    for (int b = 1; b < m->nb; b++) {
        memcpy(cacc[b], cacc[m->parent[b]], sizeof cacc[b]);
        for (int j = m->jadr[b]; j < m->jadr[b] + m->jnum[b]; j++)
            for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof_dot[j][k] * d->qvel[j];
        double Ia[6], Iv[6], t[6];
        mulInertVec(Ia, d->cinert[b], cacc[b]);
        mulInertVec(Iv, d->cinert[b], d->cvel[b]);
        crossForce(t, d->cvel[b], Iv);
        for (int k = 0; k < 6; k++) cfrc[b][k] = Ia[k] + t[k];
    }
    for (int b = m->nb - 1; b >= 1; b--) {
        int p = m->parent[b];
        if (p > 0)
            for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k];
    }
    for (int j = 0; j < m->nv; j++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += d->cdof[j][k] * cfrc[m->jbody[j]][k];
        d->qfrc_bias[j] = s;
    }
}

static void passive(const Model *m, Data *d) {
    double dens = m->ov_set ? m->ov_density : m->density, visc = m->ov_set ? m->ov_viscosity : m->viscosity;
    for (int j = 0; j < m->nv; j++)
        d->qfrc_passive[j] = -m->stiff[j] * m->stiffness_scale * d->qpos[j] /* springref = 0 */
                             - m->damp[j] * m->damping_scale * d->qvel[j];
    if (dens > 0 || visc > 0) {
        for (int b = 1; b < m->nb; b++) {
            if (m->mass[b] < MINVAL) continue;
            const double *box = m->box[b];
            /* 6-D velocity at the body com, in the body inertial frame (= body axes here) */
            double off[3], vlin[3], t[3], lw[3], lv[3], lf[6] = {0, 0, 0, 0, 0, 0};
            for (int k = 0; k < 3; k++) off[k] = d->xipos[b][k] - d->com[b][k];
            cross3(t, d->cvel[b], off);
            for (int k = 0; k < 3; k++) vlin[k] = d->cvel[b][3 + k] + t[k];
            mulMatTVec3(lw, d->xmat[b], d->cvel[b]);
            mulMatTVec3(lv, d->xmat[b], vlin);
            if (visc > 0) {
                double diam = (box[0] + box[1] + box[2]) / 3.0;
                for (int k = 0; k < 3; k++) {
                    lf[k] = -PI * diam * diam * diam * visc * lw[k];
                    lf[3 + k] = -3.0 * PI * diam * visc * lv[k];
                }
            }
            if (dens > 0) {
                lf[3] -= 0.5 * dens * box[1] * box[2] * fabs(lv[0]) * lv[0];
                lf[4] -= 0.5 * dens * box[0] * box[2] * fabs(lv[1]) * lv[1];
                lf[5] -= 0.5 * dens * box[0] * box[1] * fabs(lv[2]) * lv[2];
                double b0 = box[0], b1 = box[1], b2 = box[2];
                lf[0] -= dens * b0 * (b1 * b1 * b1 * b1 + b2 * b2 * b2 * b2) * fabs(lw[0]) * lw[0] / 64.0;
                lf[1] -= dens * b1 * (b0 * b0 * b0 * b0 + b2 * b2 * b2 * b2) * fabs(lw[1]) * lw[1] / 64.0;
                lf[2] -= dens * b2 * (b0 * b0 * b0 * b0 + b1 * b1 * b1 * b1) * fabs(lw[2]) * lw[2] / 64.0;
            }
            double tq[3], fr[3], jp[3][MJO_MAXV], jr[3][MJO_MAXV];
            mulMatVec3(tq, d->xmat[b], lf);
            mulMatVec3(fr, d->xmat[b], lf + 3);
            jac(m, d, b, d->xipos[b], jp, jr);
            for (int j = 0; j < m->nv; j++)
                for (int k = 0; k < 3; k++) d->qfrc_passive[j] += jp[k][j] * fr[k] + jr[k][j] * tq[k];
        }
    }
}

static void actuation(const Model *m, Data *d) {
    memset(d->qfrc_actuator, 0, sizeof d->qfrc_actuator);
    for (int u = 0; u < m->nu; u++) {
        double c = d->ctrl[u];
        if (m->ctrllimited[u]) c = fmin(m->ctrlrange[u][1], fmax(m->ctrlrange[u][0], c));
        d->qfrc_actuator[m->act_dof[u]] += m->gear[u] * c;
    }
}

/* ------------------------------------------------------------------------------------ */
/* primal Newton solver, exact line search, run to the active-set fixed point           */
/* ------------------------------------------------------------------------------------ */
typedef struct { double a; int i; } Brk;
static int brk_cmp(const void *x, const void *y) {
    double a = ((const Brk *)x)->a, b = ((const Brk *)y)->a;
    return a < b ? -1 : (a > b ? 1 : 0);
}

static void solve_constraints(const Model *m, Data *d) {
    int nv = m->nv, ne = d->nefc;
    memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
    d->niter = 0;
    memset(d->qfrc_constraint, 0, sizeof d->qfrc_constraint);
    if (ne == 0) return;
    double jar[MJO_MAXEFC], jv[MJO_MAXEFC];
    int act[MJO_MAXEFC], converged = 0;
    for (int it = 0; it < 100; it++) {
        double Ma[MJO_MAXV], grad[MJO_MAXV];
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int j = 0; j < nv; j++) s += d->M[i][j] * d->qacc[j];
            Ma[i] = s;
        }
        for (int r = 0; r < ne; r++) {
            double s = -d->efc_aref[r];
            for (int j = 0; j < nv; j++) s += d->J[r][j] * d->qacc[j];
            jar[r] = s;
            act[r] = s < 0;
        }
        double H[MJO_MAXV][MJO_MAXV];
        memcpy(H, d->M, sizeof H);
        for (int i = 0; i < nv; i++) grad[i] = Ma[i] - d->qfrc_smooth[i];
        for (int r = 0; r < ne; r++) {
            if (!act[r]) continue;
            double f = -d->efc_D[r] * jar[r];
            for (int i = 0; i < nv; i++) {
                grad[i] -= d->J[r][i] * f;
                double di = d->efc_D[r] * d->J[r][i];
                if (di != 0)
                    for (int j = 0; j <= i; j++) H[i][j] += di * d->J[r][j];
            }
        }
        if (chol_factor(H, nv)) { d->status |= MJO_ST_NONFINITE; break; }
        double s[MJO_MAXV];
        for (int i = 0; i < nv; i++) s[i] = -grad[i];
        chol_solve(H, nv, s);
        double snorm = 0, qnorm = 0;
        for (int i = 0; i < nv; i++) { snorm += s[i] * s[i]; qnorm += d->qacc[i] * d->qacc[i]; }
        d->niter = it + 1;
        if (!(snorm == snorm)) { d->status |= MJO_ST_NONFINITE; break; }
        if (sqrt(snorm) <= 1e-15 * (1.0 + sqrt(qnorm))) { converged = 1; break; }
        /* exact line search on the piecewise-quadratic cost along s */
        double a1 = 0, a2 = 0;
        for (int i = 0; i < nv; i++) {
            double Mv = 0;
            for (int j = 0; j < nv; j++) Mv += d->M[i][j] * s[j];
            a1 += s[i] * (Ma[i] - d->qfrc_smooth[i]);
            a2 += s[i] * Mv;
        }
        Brk brk[MJO_MAXEFC];
        int nb = 0, cur[MJO_MAXEFC];
        double c0 = a1, sl = a2;
        for (int r = 0; r < ne; r++) {
            double t = 0;
            for (int j = 0; j < nv; j++) t += d->J[r][j] * s[j];
            jv[r] = t;
            cur[r] = jar[r] < 0 || (jar[r] == 0 && t < 0);
            if (cur[r]) { c0 += d->efc_D[r] * jar[r] * t; sl += d->efc_D[r] * t * t; }
            if (t != 0) {
                double a = -jar[r] / t;
                if (a > 0) { brk[nb].a = a; brk[nb].i = r; nb++; }
            }
        }
        qsort(brk, nb, sizeof(Brk), brk_cmp);
        double alpha = -c0 / sl;
        for (int k = 0; k < nb; k++) {
            if (alpha <= brk[k].a) break;
            int r = brk[k].i;
            double sg = cur[r] ? -1.0 : 1.0;
            c0 += sg * d->efc_D[r] * jar[r] * jv[r];
            sl += sg * d->efc_D[r] * jv[r] * jv[r];
            cur[r] = !cur[r];
            alpha = -c0 / sl;
            if (alpha < brk[k].a) alpha = brk[k].a; /* derivative jumps nowhere (C1), guard roundoff */
        }
        for (int i = 0; i < nv; i++) d->qacc[i] += alpha * s[i];
        /* fixed point: active set unchanged by the step => exact minimiser of the active quadratic */
        int same = 1;
        for (int r = 0; r < ne; r++) {
            double t = -d->efc_aref[r];
            for (int j = 0; j < nv; j++) t += d->J[r][j] * d->qacc[j];
            if ((t < 0) != act[r]) { same = 0; break; }
        }
        if (same) { converged = 1; break; }
    }
    if (!converged) d->status |= MJO_ST_NOCONVERGE;
    for (int r = 0; r < ne; r++) {
        double t = -d->efc_aref[r];
        for (int j = 0; j < nv; j++) t += d->J[r][j] * d->qacc[j];
        d->efc_force[r] = t < 0 ? -d->efc_D[r] * t : 0.0;
        for (int j = 0; j < nv; j++) d->qfrc_constraint[j] += d->J[r][j] * d->efc_force[r];
    }
}

/* ------------------------------------------------------------------------------------ */
/* mj_forward / integrators / env step                                                  */
/* ------------------------------------------------------------------------------------ */
static void forward(const Model *m, Data *d) {
    int nv = m->nv;
    kinematics(m, d);
    comPos(m, d);
    crb(m, d);
    collision(m, d);
    makeConstraint(m, d);
    comVel(m, d);
    passive(m, d);
    rne_bias(m, d);
    actuation(m, d);
    double L[MJO_MAXV][MJO_MAXV];
    memcpy(L, d->M, sizeof L);
    for (int i = 0; i < nv; i++) {
        d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
        d->qacc_smooth[i] = d->qfrc_smooth[i];
    }
    if (chol_factor(L, nv)) d->status |= MJO_ST_NONFINITE;
    chol_solve(L, nv, d->qacc_smooth);
    solve_constraints(m, d);
    d->n_eval++;
    d->sum_iter += d->niter;
    d->sum_nefc += d->nefc;
    if (d->nefc) d->n_eval_con++;
}

static void euler(const Model *m, Data *d) {
    int nv = m->nv;
    double h = m->timestep, qacc[MJO_MAXV];
    if (m->any_damping && m->damping_scale != 0) {
        double L[MJO_MAXV][MJO_MAXV];
        memcpy(L, d->M, sizeof L);
        for (int i = 0; i < nv; i++) {
            L[i][i] += h * m->damp[i] * m->damping_scale;
            qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
        }
        if (chol_factor(L, nv)) d->status |= MJO_ST_NONFINITE;
        chol_solve(L, nv, qacc);
    } else memcpy(qacc, d->qacc, sizeof qacc);
    for (int i = 0; i < nv; i++) {
        d->qvel[i] += h * qacc[i];
        d->qpos[i] += h * d->qvel[i];
    }
}

static void rk4(const Model *m, Data *d) {
    static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1.0}};
    static const double Bw[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    int nv = m->nv;
    double h = m->timestep, q0[MJO_MAXV], v0[MJO_MAXV], Fv[4][MJO_MAXV], Fa[4][MJO_MAXV];
    memcpy(q0, d->qpos, sizeof q0);
    memcpy(v0, d->qvel, sizeof v0);
    memcpy(Fv[0], d->qvel, sizeof v0);
    memcpy(Fa[0], d->qacc, sizeof v0);
    for (int i = 1; i < 4; i++) {
        for (int k = 0; k < nv; k++) {
            double dv = 0, da = 0;
            for (int j = 0; j < i; j++) { dv += A[i - 1][j] * Fv[j][k]; da += A[i - 1][j] * Fa[j][k]; }
            d->qpos[k] = q0[k] + h * dv;
            d->qvel[k] = v0[k] + h * da;
        }
        forward(m, d);
        memcpy(Fv[i], d->qvel, sizeof v0);
        memcpy(Fa[i], d->qacc, sizeof v0);
    }
    for (int k = 0; k < nv; k++) {
        double dv = 0, da = 0;
        for (int j = 0; j < 4; j++) { dv += Bw[j] * Fv[j][k]; da += Bw[j] * Fa[j][k]; }
        d->qpos[k] = q0[k] + h * dv;
        d->qvel[k] = v0[k] + h * da;
    }
}

static void mj_step(const Model *m, Data *d) {
    forward(m, d);
    if (m->integrator) rk4(m, d); else euler(m, d);
}

static double hop_sigmoid(double x) { return 1.0 / (1.0 + pow(200.0, -x + 0.5)); }

/* env.step: returns reward; d->qpos/qvel advanced.  `a` = the action handed to step(). */
static double env_step(const Model *m, Data *d, const double *a, int frame_skip) {
    int nu = m->nu;
    double dt = m->timestep * frame_skip, xb = d->qpos[0], asq = 0, r = 0;
    memcpy(d->ctrl, a, nu * sizeof(double));
    for (int k = 0; k < frame_skip; k++) mj_step(m, d);
    for (int u = 0; u < nu; u++) asq += a[u] * a[u];
    switch (m->reward) {
    case 0: { /* inverted_pendulum.py:26-35 (x and y both use cos, literally) */
        double y = 0.6 * cos(d->qpos[1]), x = 0.6 * cos(d->qpos[1]);
        double pen = 0.01 * (x * x) + (y - 1) * (y - 1);
        r = -pen - 0.001 * (a[0] * a[0]);
        break;
    }
    case 1: { /* inverted_double_pendulum.py:28-33, site_xpos is stale (last forward call) */
        double x = d->site_xpos[0], y = d->site_xpos[2];
        double pen = 0.01 * (x * x) + (y - 2) * (y - 2);
        double v1 = d->qvel[1], v2 = d->qvel[2];
        double vp = 1e-3 * (v1 * v1) + 5e-3 * (v2 * v2);
        r = -pen - vp;
        break;
    }
    case 2: /* swimmer.py:23-29 */
        r = (d->qpos[0] - xb) / dt + (-0.0001 * asq);
        break;
    case 3: { /* hopper.py:30-41 */
        double h = d->qpos[1], ang = d->qpos[2];
        r = (d->qpos[0] - xb) / dt;
        double aa = fmod(fabs(ang), 2 * PI);
        if (aa > PI) aa = 2 * PI - aa;
        double c1 = hop_sigmoid(h / 1.25), c2 = hop_sigmoid((PI - aa) / PI);
        r += c1 * 1.0 + c2 * 1.0;
        r -= 1e-3 * asq;
        break;
    }
    case 4: { /* walker2d.py:23-30 (max(...,1) is literal) */
        double h = d->qpos[1], ang = d->qpos[2];
        r = (d->qpos[0] - xb) / dt;
        double coeff = fmin(fmax(h / 0.8, 0.0), 1.0) * 0.5 + fmax((PI - fabs(ang)) / PI, 1.0) * 0.5;
        r += coeff * 1.0;
        r -= 1e-3 * asq;
        break;
    }
    case 5: /* half_cheetah.py:25-31 */
        r = -0.01 * asq + (d->qpos[0] - xb) / dt;
        break;
    }
    return r;
}

/* ------------------------------------------------------------------------------------ */
/* model compile                                                                        */
/* ------------------------------------------------------------------------------------ */
static void capsule_inertia(const RawGeom *g, double *mass, double *com, double *I /*9, about com, body axes*/) {
    double d[3] = {g->b[0] - g->a[0], g->b[1] - g->a[1], g->b[2] - g->a[2]};
    double h = sqrt(dot3(d, d)), r = g->radius;
    double u[3] = {d[0] / h, d[1] / h, d[2] / h};
    /* MuJoCo 2.0 capsule volume is pi r^2 (h + r) -- NOT pi r^2 (h + 4r/3): this reproduces the
     * publicly known mujoco200 masses of the gym -v2 models (Hopper torso 3.53429174, thigh 3.92699082,
     * leg 2.71433605, foot 5.0893801; HalfCheetah torso 6.36031332 ... with settotalmass 14)
     * [from knowledge; later MuJoCo releases fixed the cap volume to 4/3 pi r^3].  The two caps
     * therefore carry mass rho*pi*r^3 in total; shape inertias are the standard solid formulas. */
    double mc = g->density * PI * r * r * h, ms = g->density * PI * r * r * r;
    double Iax = mc * r * r / 2.0 + ms * 2.0 * r * r / 5.0;
    double Ipe = mc * (3 * r * r + h * h) / 12.0 + ms * (2.0 * r * r / 5.0 + h * h / 4.0 + 3.0 * h * r / 8.0);
    *mass = mc + ms;
    for (int k = 0; k < 3; k++) com[k] = 0.5 * (g->a[k] + g->b[k]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) I[3 * i + j] = Ipe * ((i == j) - u[i] * u[j]) + Iax * u[i] * u[j];
}

static void set_qpos0_state(const Model *m, Data *d) {
    memset(d, 0, sizeof *d);
    for (int j = 0; j < m->nv; j++) d->qpos[j] = m->qpos0[j];
}

static int compile_model(int id) {
    Model *m = &g_models[id];
    const RawModel *raw = &MJO_RAW_MODELS[id];
    double ovd = m->ov_density, ovv = m->ov_viscosity, ovg[3] = {m->ov_gravity[0], m->ov_gravity[1], m->ov_gravity[2]};
    double ds = m->damping_scale, ss = m->stiffness_scale, as = m->armature_scale;
    int ovs = m->ov_set, was = m->compiled;
    memset(m, 0, sizeof *m);
    m->raw = raw;
    if (was) {
        m->ov_density = ovd; m->ov_viscosity = ovv; memcpy(m->ov_gravity, ovg, sizeof ovg);
        m->damping_scale = ds; m->stiffness_scale = ss; m->armature_scale = as; m->ov_set = ovs;
    } else {
        m->damping_scale = m->stiffness_scale = m->armature_scale = 1.0;
        m->ov_density = raw->density; m->ov_viscosity = raw->viscosity;
        memcpy(m->ov_gravity, raw->gravity, sizeof ovg);
    }
    m->nb = raw->nbody + 1;
    m->parent[0] = 0;
    double total = 0;
    for (int rb = 0; rb < raw->nbody; rb++) {
        int b = rb + 1;
        const RawBody *B = &raw->body[rb];
        m->parent[b] = B->parent + 1;
        m->rootid[b] = m->parent[b] == 0 ? b : m->rootid[m->parent[b]];
        memcpy(m->bpos[b], B->pos, sizeof B->pos);
        m->jadr[b] = m->nv;
        m->jnum[b] = B->njnt;
        for (int k = 0; k < B->njnt; k++) {
            const RawJoint *Jn = &B->jnt[k];
            int j = m->nv++;
            m->jtype[j] = Jn->type; m->jbody[j] = b;
            double n = sqrt(dot3(Jn->axis, Jn->axis));
            for (int c = 0; c < 3; c++) { m->jaxis[j][c] = Jn->axis[c] / n; m->jpos[j][c] = Jn->pos[c]; }
            m->qpos0[j] = Jn->ref; m->stiff[j] = Jn->stiffness; m->damp[j] = Jn->damping; m->arma[j] = Jn->armature;
            m->limited[j] = Jn->limited; m->range[j][0] = Jn->range[0]; m->range[j][1] = Jn->range[1];
            m->jmargin[j] = Jn->margin;
            memcpy(m->jsolref[j], Jn->solref, sizeof Jn->solref);
            memcpy(m->jsolimp[j], Jn->solimp, sizeof Jn->solimp);
            if (Jn->damping > 0) m->any_damping = 1;
            /* dof parent: previous dof of this body, else last dof of nearest ancestor with dofs */
            if (k > 0) m->dof_parent[j] = j - 1;
            else {
                int p = m->parent[b], dp = -1;
                while (p > 0) { if (m->jnum[p] > 0) { dp = m->jadr[p] + m->jnum[p] - 1; break; } p = m->parent[p]; }
                m->dof_parent[j] = dp;
            }
        }
        /* inertia from geoms */
        double mass = 0, com[3] = {0, 0, 0};
        double gm[2], gc[2][3], gI[2][9];
        for (int k = 0; k < B->ngeom; k++) {
            capsule_inertia(&B->geom[k], &gm[k], gc[k], gI[k]);
            mass += gm[k];
            for (int c = 0; c < 3; c++) com[c] += gm[k] * gc[k][c];
            int g = m->ng++;
            m->gbody[g] = b; m->geom[g] = B->geom[k];
        }
        for (int c = 0; c < 3; c++) com[c] /= mass;
        double I[9] = {0};
        for (int k = 0; k < B->ngeom; k++) {
            double dd[3] = {gc[k][0] - com[0], gc[k][1] - com[1], gc[k][2] - com[2]}, d2 = dot3(dd, dd);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) I[3 * i + j] += gI[k][3 * i + j] + gm[k] * ((i == j ? d2 : 0.0) - dd[i] * dd[j]);
        }
        m->mass[b] = mass; total += mass;
        memcpy(m->ipos[b], com, sizeof com);
        memcpy(m->inertia[b], I, sizeof I);
    }
    if (raw->settotalmass > 0) {
        double sc = raw->settotalmass / total;
        for (int b = 1; b < m->nb; b++) {
            m->mass[b] *= sc;
            for (int k = 0; k < 9; k++) m->inertia[b][k] *= sc;
        }
    }
    m->nu = raw->nu;
    for (int u = 0; u < raw->nu; u++) {
        m->act_dof[u] = raw->act[u].dof; m->gear[u] = raw->act[u].gear;
        m->ctrllimited[u] = raw->act[u].ctrllimited;
        m->ctrlrange[u][0] = raw->act[u].ctrlrange[0]; m->ctrlrange[u][1] = raw->act[u].ctrlrange[1];
    }
    m->has_floor = raw->has_floor; m->floor = raw->floor; m->selfcollide = raw->selfcollide;
    m->timestep = raw->timestep; memcpy(m->gravity, raw->gravity, sizeof raw->gravity);
    m->density = raw->density; m->viscosity = raw->viscosity;
    m->integrator = raw->integrator; m->frame_skip = raw->frame_skip;
    m->has_site = raw->has_site; m->site_body = raw->site_body + 1;
    memcpy(m->site_pos, raw->site_pos, sizeof raw->site_pos);
    m->reward = raw->reward;
    /* fluid boxes: principal frame = body axes (requires diagonal body-frame inertia) */
    for (int b = 1; b < m->nb; b++) {
        const double *I = m->inertia[b];
        double sc = 6.0 / m->mass[b];
        m->box[b][0] = sqrt(fmax(MINVAL, I[4] + I[8] - I[0]) * sc);
        m->box[b][1] = sqrt(fmax(MINVAL, I[0] + I[8] - I[4]) * sc);
        m->box[b][2] = sqrt(fmax(MINVAL, I[0] + I[4] - I[8]) * sc);
        if ((raw->density > 0 || raw->viscosity > 0) &&
            (fabs(I[1]) + fabs(I[2]) + fabs(I[5]) > 1e-12 * (I[0] + I[4] + I[8])))
            return -2; /* fluid forces need a body-axis-aligned inertial frame in this restatement */
    }
    /* invweight0 at qpos0 [MJ mj_setConst] */
    Data *d = (Data *)malloc(sizeof(Data));
    set_qpos0_state(m, d);
    kinematics(m, d); comPos(m, d); crb(m, d);
    int nv = m->nv;
    double L[MJO_MAXV][MJO_MAXV], Minv[MJO_MAXV][MJO_MAXV];
    memcpy(L, d->M, sizeof L);
    if (chol_factor(L, nv)) { free(d); return -3; }
    for (int j = 0; j < nv; j++) {
        double e[MJO_MAXV] = {0};
        e[j] = 1;
        chol_solve(L, nv, e);
        for (int i = 0; i < nv; i++) Minv[i][j] = e[i];
    }
    for (int j = 0; j < nv; j++) m->dof_invweight0[j] = Minv[j][j];
    for (int b = 1; b < m->nb; b++) {
        double jp[3][MJO_MAXV], jr[3][MJO_MAXV], tp = 0, tr = 0;
        jac(m, d, b, d->xipos[b], jp, jr);
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < nv; i++)
                for (int j = 0; j < nv; j++) { tp += jp[k][i] * Minv[i][j] * jp[k][j]; tr += jr[k][i] * Minv[i][j] * jr[k][j]; }
        m->body_invweight0[b][0] = fmax(MINVAL, tp / 3.0);
        m->body_invweight0[b][1] = fmax(MINVAL, tr / 3.0);
    }
    free(d);
    m->compiled = 1;
    return 0;
}

static Model *get_model(int id) {
    if (id < 0 || id >= MJO_NUM_RAW_MODELS) return NULL;
    if (!g_models[id].compiled) {
#pragma omp critical(mjo_compile)
        { if (!g_models[id].compiled && compile_model(id)) g_models[id].compiled = 0; }
    }
    return g_models[id].compiled ? &g_models[id] : NULL;
}

/* ------------------------------------------------------------------------------------ */
/* public API                                                                           */
/* ------------------------------------------------------------------------------------ */
int mjo_num_models(void) { return MJO_NUM_RAW_MODELS; }
const char *mjo_model_name(int id) { return (id >= 0 && id < MJO_NUM_RAW_MODELS) ? MJO_RAW_MODELS[id].env : NULL; }
int mjo_model_id(const char *env) {
    for (int i = 0; i < MJO_NUM_RAW_MODELS; i++)
        if (!strcmp(env, MJO_RAW_MODELS[i].env)) return i;
    return -1;
}
int mjo_dims(int id, int *nq, int *nv, int *nu, double *timestep, int *frame_skip) {
    Model *m = get_model(id);
    if (!m) return -1;
    if (nq) *nq = m->nv;
    if (nv) *nv = m->nv;
    if (nu) *nu = m->nu;
    if (timestep) *timestep = m->timestep;
    if (frame_skip) *frame_skip = m->frame_skip;
    return 0;
}
int mjo_set_option(int id, const char *key, double value) {
    Model *m = get_model(id);
    if (!m) return -1;
    if (!strcmp(key, "reset")) { m->compiled = 0; memset(m, 0, sizeof *m); return get_model(id) ? 0 : -1; }
    m->ov_set = 1;
    if (!strcmp(key, "density")) m->ov_density = value;
    else if (!strcmp(key, "viscosity")) m->ov_viscosity = value;
    else if (!strcmp(key, "gravity_x")) m->ov_gravity[0] = value;
    else if (!strcmp(key, "gravity_y")) m->ov_gravity[1] = value;
    else if (!strcmp(key, "gravity_z")) m->ov_gravity[2] = value;
    else if (!strcmp(key, "damping_scale")) m->damping_scale = value;
    else if (!strcmp(key, "stiffness_scale")) m->stiffness_scale = value;
    else if (!strcmp(key, "armature_scale")) m->armature_scale = value;
    else return -2;
    return 0;
}
int mjo_body_consts(int id, int body, double *mass, double *ipos3, double *inertia9, double *invweight2) {
    Model *m = get_model(id);
    if (!m || body < 0 || body >= m->nb - 1) return -1;
    int b = body + 1;
    if (mass) *mass = m->mass[b];
    if (ipos3) memcpy(ipos3, m->ipos[b], 3 * sizeof(double));
    if (inertia9) memcpy(inertia9, m->inertia[b], 9 * sizeof(double));
    if (invweight2) memcpy(invweight2, m->body_invweight0[b], 2 * sizeof(double));
    return 0;
}
int mjo_dof_invweight(int id, double *out) {
    Model *m = get_model(id);
    if (!m) return -1;
    memcpy(out, m->dof_invweight0, m->nv * sizeof(double));
    return 0;
}

static void load_state(const Model *m, Data *d, const double *qpos, const double *qvel, const double *ctrl) {
    memcpy(d->qpos, qpos, m->nv * sizeof(double));
    memcpy(d->qvel, qvel, m->nv * sizeof(double));
    if (ctrl) memcpy(d->ctrl, ctrl, m->nu * sizeof(double));
}

int mjo_forward(int id, const double *qpos, const double *qvel, const double *ctrl, double *qacc,
                double *qacc_smooth, double *M, double *qfrc_bias, double *qfrc_passive, double *qfrc_actuator,
                double *qfrc_constraint, int *nefc, double *efc_J, double *efc_pos, double *efc_aref, double *efc_D,
                double *efc_force, int *niter, double *energy2, double *momentum6) {
    Model *m = get_model(id);
    if (!m) return -1;
    Data *d = (Data *)calloc(1, sizeof(Data));
    load_state(m, d, qpos, qvel, ctrl);
    forward(m, d);
    int nv = m->nv;
    if (qacc) memcpy(qacc, d->qacc, nv * sizeof(double));
    if (qacc_smooth) memcpy(qacc_smooth, d->qacc_smooth, nv * sizeof(double));
    if (M) for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) M[i * nv + j] = d->M[i][j];
    if (qfrc_bias) memcpy(qfrc_bias, d->qfrc_bias, nv * sizeof(double));
    if (qfrc_passive) memcpy(qfrc_passive, d->qfrc_passive, nv * sizeof(double));
    if (qfrc_actuator) memcpy(qfrc_actuator, d->qfrc_actuator, nv * sizeof(double));
    if (qfrc_constraint) memcpy(qfrc_constraint, d->qfrc_constraint, nv * sizeof(double));
    if (nefc) *nefc = d->nefc;
    for (int r = 0; r < d->nefc; r++) {
        if (efc_J) memcpy(efc_J + r * nv, d->J[r], nv * sizeof(double));
        if (efc_pos) efc_pos[r] = d->efc_pos[r];
        if (efc_aref) efc_aref[r] = d->efc_aref[r];
        if (efc_D) efc_D[r] = d->efc_D[r];
        if (efc_force) efc_force[r] = d->efc_force[r];
    }
    if (niter) *niter = d->niter;
    if (energy2) {
        const double *g = m->ov_set ? m->ov_gravity : m->gravity;
        double pot = 0, kin = 0;
        for (int b = 1; b < m->nb; b++) pot -= m->mass[b] * dot3(g, d->xipos[b]);
        for (int j = 0; j < nv; j++) pot += 0.5 * m->stiff[j] * m->stiffness_scale * d->qpos[j] * d->qpos[j];
        for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) kin += 0.5 * d->qvel[i] * d->M[i][j] * d->qvel[j];
        energy2[0] = pot; energy2[1] = kin;
    }
    if (momentum6) {
        /* total linear momentum and angular momentum about the world origin (armature excluded) */
        double P[3] = {0, 0, 0}, Lm[3] = {0, 0, 0};
        for (int b = 1; b < m->nb; b++) {
            double h[6], O[3], t[3];
            mulInertVec(h, d->cinert[b], d->cvel[b]);
            memcpy(O, d->com[b], sizeof O);
            cross3(t, O, h + 3);
            for (int k = 0; k < 3; k++) { P[k] += h[3 + k]; Lm[k] += h[k] + t[k]; }
        }
        memcpy(momentum6, P, sizeof P); memcpy(momentum6 + 3, Lm, sizeof Lm);
    }
    int st = (int)d->status;
    free(d);
    return st ? -(100 + st) : 0;
}

static unsigned check_finite(const Model *m, const Data *d) {
    for (int j = 0; j < m->nv; j++)
        if (!isfinite(d->qpos[j]) || !isfinite(d->qvel[j])) return MJO_ST_NONFINITE;
    return 0;
}

static long g_n_eval, g_n_eval_con, g_sum_iter, g_sum_nefc;

int mjo_step(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
             double *qpos_next, double *qvel_next, double *reward, uint32_t *status, int nthreads) {
    Model *m = get_model(id);
    if (!m) return -1;
    if (frame_skip <= 0) frame_skip = m->frame_skip;
    int nv = m->nv, nu = m->nu;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 16) if (nthreads > 1)
    for (int64_t i = 0; i < N; i++) {
        Data d;
        memset(&d, 0, sizeof d);
        load_state(m, &d, qpos + i * nv, qvel + i * nv, ctrl + i * nu);
        double r = env_step(m, &d, ctrl + i * nu, frame_skip);
        memcpy(qpos_next + i * nv, d.qpos, nv * sizeof(double));
        memcpy(qvel_next + i * nv, d.qvel, nv * sizeof(double));
        if (reward) reward[i] = r;
        if (status) status[i] = d.status | check_finite(m, &d);
        if (nthreads <= 1) { g_n_eval += d.n_eval; g_n_eval_con += d.n_eval_con; g_sum_iter += d.sum_iter; g_sum_nefc += d.sum_nefc; }
    }
    return 0;
}

/* one rollout of calculate_gradients: restore snapshot, perturb, step nsteps times */
static double rollout(const Model *m, Data *d, const double *qpos, const double *qvel, const double *ctrl,
                      int pert_kind /*0 none,1 ctrl,2 qvel,3 qpos*/, int idx, double delta, int nsteps,
                      int frame_skip, double *s_out) {
    int nv = m->nv;
    double a[MJO_MAXU], r = 0;
    unsigned st = d->status;
    long ne = d->n_eval, nc = d->n_eval_con, si = d->sum_iter, sn = d->sum_nefc, nx = d->n_cross;
    memset(d, 0, sizeof *d);
    d->status = st; d->n_eval = ne; d->n_eval_con = nc; d->sum_iter = si; d->sum_nefc = sn; d->n_cross = nx;
    load_state(m, d, qpos, qvel, ctrl);
    memcpy(a, ctrl, m->nu * sizeof(double));
    if (pert_kind == 1) a[idx] += delta;
    else if (pert_kind == 2) d->qvel[idx] += delta;
    else if (pert_kind == 3) d->qpos[idx] += delta;
    for (int k = 0; k < nsteps; k++) r = env_step(m, d, a, frame_skip);
    memcpy(s_out, d->qpos, nv * sizeof(double));
    memcpy(s_out + nv, d->qvel, nv * sizeof(double));
    d->status |= check_finite(m, d);
    return r;
}

static int fd_impl(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                   double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                   double *s_next, double *reward, uint32_t *status, int32_t *evals_constrained, int32_t *evals_crossing,
                   int nthreads) {
    Model *m = get_model(id);
    if (!m) return -1;
    if (frame_skip <= 0) frame_skip = m->frame_skip;
    if (nsteps <= 0) nsteps = 1;
    int nv = m->nv, nu = m->nu, n = 2 * nv;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 4) if (nthreads > 1)
    for (int64_t i = 0; i < N; i++) {
        Data d;
        memset(&d, 0, sizeof d);
        const double *q = qpos + i * nv, *v = qvel + i * nv, *c = ctrl + i * nu;
        double s0[2 * MJO_MAXV], sp[2 * MJO_MAXV], sm[2 * MJO_MAXV];
        double r0 = rollout(m, &d, q, v, c, 0, 0, 0.0, nsteps, frame_skip, s0);
        if (s_next) memcpy(s_next + i * n, s0, n * sizeof(double));
        if (reward) reward[i] = r0;
        /* column order of the reference: ctrl, qvel, qpos (backward.py:389,409,433) */
        for (int kind = 1; kind <= 3; kind++) {
            int cnt = kind == 1 ? nu : nv;
            for (int k = 0; k < cnt; k++) {
                double rp = rollout(m, &d, q, v, c, kind, k, eps, nsteps, frame_skip, sp), rm = r0, den = eps;
                const double *base = s0;
                if (fd_mode == 1) { rm = rollout(m, &d, q, v, c, kind, k, -eps, nsteps, frame_skip, sm); base = sm; den = 2 * eps; }
                for (int o = 0; o < n; o++) {
                    double val = (sp[o] - base[o]) / den;
                    if (kind == 1) { if (dsda) dsda[(i * n + o) * nu + k] = val; }
                    else if (dsds) dsds[(i * n + o) * n + (kind == 3 ? k : nv + k)] = val;
                }
                double rv = (rp - rm) / den;
                if (kind == 1) { if (drda) drda[i * nu + k] = rv; }
                else if (drds) drds[i * n + (kind == 3 ? k : nv + k)] = rv;
            }
        }
        if (status) status[i] = d.status;
        if (evals_constrained) evals_constrained[i] = (int32_t)d.n_eval_con;
        if (evals_crossing) evals_crossing[i] = (int32_t)d.n_cross;
        if (nthreads <= 1) { g_n_eval += d.n_eval; g_n_eval_con += d.n_eval_con; g_sum_iter += d.sum_iter; g_sum_nefc += d.sum_nefc; }
    }
    return 0;
}

int mjo_fd_jacobians(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                     double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                     double *s_next, double *reward, uint32_t *status, int nthreads) {
    return fd_impl(id, frame_skip, N, qpos, qvel, ctrl, eps, fd_mode, nsteps, dsds, dsda, drds, drda, s_next, reward, status,
                   NULL, NULL, nthreads);
}

/* The same, also reporting per unit how many dynamics evaluations of its rollouts (nominal and perturbed) had at least
 * one active constraint row -- joint limit or contact.  The full-size parity tests use it to tell the contact-rich units,
 * for which the FD tolerance is stated separately, from the smooth ones.  evals_crossing [N] (may be NULL): how many
 * capsule/capsule contacts of the unit's rollouts were made by the crossing-axes rule of sphereSphere ("parity unpinned"). */
int mjo_fd_jacobians_info(int id, int frame_skip, int64_t N, const double *qpos, const double *qvel, const double *ctrl,
                          double eps, int fd_mode, int nsteps, double *dsds, double *dsda, double *drds, double *drda,
                          double *s_next, double *reward, uint32_t *status, int32_t *evals_constrained,
                          int32_t *evals_crossing, int nthreads) {
    return fd_impl(id, frame_skip, N, qpos, qvel, ctrl, eps, fd_mode, nsteps, dsds, dsda, drds, drda, s_next, reward, status,
                   evals_constrained, evals_crossing, nthreads);
}

int mjo_stats(double *mean_newton_iters, double *frac_evals_constrained, double *mean_nefc) {
    if (mean_newton_iters) *mean_newton_iters = g_n_eval_con ? (double)g_sum_iter / g_n_eval_con : 0;
    if (frac_evals_constrained) *frac_evals_constrained = g_n_eval ? (double)g_n_eval_con / g_n_eval : 0;
    if (mean_nefc) *mean_nefc = g_n_eval_con ? (double)g_sum_nefc / g_n_eval_con : 0;
    g_n_eval = g_n_eval_con = g_sum_iter = g_sum_nefc = 0;
    return 0;
}
