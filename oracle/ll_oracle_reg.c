/*
 * ll_oracle_reg.c -- CPU ORACLE (test infrastructure only, see ll_oracle.h) for the scan-to-map
 * registration inner loop: a plain-C restatement of hku-mars/loam_livox
 *   source/point_cloud_registration.hpp (PCR:163-583, 607-685)
 *   source/ceres_icp.hpp                (ICP:81-380, residual functors)
 * Pinned against the reference's own text compiled here (oracle/_ref, tests/test_ref_pin.py): functor residuals and
 * Jacobians to 1e-12, the driver's pose / costs / thresholds to 1e-9; FLANN, Ceres' LM and Eigen stay restated
 * (see ll_oracle.h).
 *
 * Third-party behaviour restated from the libraries' published algorithms:
 *
 *  Eigen3 (unpinned, 3.3 semantics):
 *    q * v            : uv = 2 (q.vec x v);  v + w uv + q.vec x uv          (QuaternionBase::_transformVector)
 *    slerp            : QuaternionBase::slerp with the 1-eps linear fallback
 *    angularDistance  : 2 atan2(|vec(q1 q2*)|, |w(q1 q2*)|)
 *    AngleAxis(q)     : n=|vec|, (w<0 -> n=-n), angle = 2 atan2(n, |w|), axis = vec/n (or (1,0,0) if n==0)
 *
 *  Ceres Solver (< 2.2, unpinned), default Solver::Options except max_num_iterations and
 *  linear_solver_type=DENSE_SCHUR (PCR:43,465-467,501-502):
 *    - AutoDiffCostFunction<F,3,4,3>: forward-mode duals (Jet<double,7>) through the functor.
 *    - EigenQuaternionParameterization: Plus(x,d) = [sin|d| d/|d|, cos|d|] (x) x, storage (x,y,z,w);
 *      ComputeJacobian 4x3 at d=0.
 *    - Loss via Corrector: rho''<=0 for Huber  =>  r *= sqrt(rho'), J *= sqrt(rho'); cost = 1/2 rho(s).
 *    - Box bounds on t (PCR:143-151): ParameterBlock::Plus clamps; the trust-region loop runs a
 *      projected ARMIJO line search on the step (TrustRegionMinimizer::DoLineSearch).
 *    - TrustRegionMinimizer + LevenbergMarquardtStrategy: radius0 1e4, max 1e16, min 1e-32,
 *      min_relative_decrease 1e-3, min/max_lm_diagonal 1e-6/1e32, jacobi_scaling (1/(1+sqrt(colnorm^2)) at
 *      iteration 0), function/gradient/parameter tolerance 1e-6/1e-10/1e-8, monotonic steps,
 *      radius update r/max(1/3, 1-(2q-1)^3) on success, r/nu (nu*=2) on failure, r/2 on invalid step,
 *      max 5 consecutive invalid steps.  The 6x6 system (J'J + D'D) y = J'r is solved by dense Cholesky
 *      (DENSE_SCHUR on a 2-block problem is algebraically the same system).
 *    - When the ARMIJO search has to contract more than once, Ceres fits a quintic through three samples:
 *      quintic_min_step below (round 2 re-fitted the two-sample cubic).
 *    - summary.final_cost = min over iteration costs, initial_cost = iteration-0 cost (SetSummaryFinalCost).
 *
 * Deviations from the reference that the oracle DEFINES (reference behaviour is undefined/UB):
 *    - degenerate plane triples (a==b or a==c -> NaN normal, ICP:328-334) are skipped;
 *    - non-finite surface query points are skipped like corner ones (PCR:242-245 checks corners only);
 *    - an empty residual set yields inlier threshold = m_inliner_dis (PCR:160 would dereference end());
 *    - m_interpolatation_* on the first deblur iteration are defined as theta=0, omega_hat=0 (PCR:58,66-68);
 *    - random sub-sampling (PCR:232-238,339-345,438-458) is NOT restated: parity runs require
 *      maximum_residual_blocks >= feature count.
 */
#include "ll_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small vector / quaternion helpers */

typedef struct {
    double x, y, z, w;
} quat; /* Eigen coefficient order */

static inline void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline double norm3(const double a[3]) { return sqrt(dot3(a, a)); }

/* Eigen QuaternionBase::_transformVector */
static void quat_rot(const quat *q, const double v[3], double o[3])
{
    double qv[3] = {q->x, q->y, q->z}, uv[3], t[3];
    cross3(qv, v, uv);
    uv[0] += uv[0];
    uv[1] += uv[1];
    uv[2] += uv[2];
    cross3(qv, uv, t);
    o[0] = v[0] + q->w * uv[0] + t[0];
    o[1] = v[1] + q->w * uv[1] + t[1];
    o[2] = v[2] + q->w * uv[2] + t[2];
}

/* Eigen quaternion product a*b */
static quat quat_mul(const quat *a, const quat *b)
{
    quat r;
    r.w = a->w * b->w - a->x * b->x - a->y * b->y - a->z * b->z;
    r.x = a->w * b->x + a->x * b->w + a->y * b->z - a->z * b->y;
    r.y = a->w * b->y + a->y * b->w + a->z * b->x - a->x * b->z;
    r.z = a->w * b->z + a->z * b->w + a->x * b->y - a->y * b->x;
    return r;
}

static double quat_angular_distance(const quat *a, const quat *b)
{
    quat bc = {-b->x, -b->y, -b->z, b->w};
    quat d = quat_mul(a, &bc);
    double v[3] = {d.x, d.y, d.z};
    return 2.0 * atan2(norm3(v), fabs(d.w));
}

static quat pose_q(const double p[7])
{
    quat q = {p[0], p[1], p[2], p[3]};
    return q;
}

/* ------------------------------------------------------------------ forward-mode duals (ceres::Jet<double,7>) */

#define NJ 7
typedef struct {
    double a;
    double v[NJ];
} jet;

static inline jet jc(double c)
{
    jet r;
    r.a = c;
    memset(r.v, 0, sizeof(r.v));
    return r;
}
static inline jet jvar(double c, int k)
{
    jet r = jc(c);
    r.v[k] = 1.0;
    return r;
}
static inline jet jadd(jet x, jet y)
{
    jet r;
    r.a = x.a + y.a;
    for (int i = 0; i < NJ; i++) r.v[i] = x.v[i] + y.v[i];
    return r;
}
static inline jet jsub(jet x, jet y)
{
    jet r;
    r.a = x.a - y.a;
    for (int i = 0; i < NJ; i++) r.v[i] = x.v[i] - y.v[i];
    return r;
}
static inline jet jmul(jet x, jet y)
{
    jet r;
    r.a = x.a * y.a;
    for (int i = 0; i < NJ; i++) r.v[i] = x.a * y.v[i] + x.v[i] * y.a;
    return r;
}
static inline jet jdiv(jet x, jet y)
{
    jet r;
    double inv = 1.0 / y.a;
    r.a = x.a * inv;
    for (int i = 0; i < NJ; i++) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv;
    return r;
}
static inline jet jscale(jet x, double s)
{
    jet r;
    r.a = x.a * s;
    for (int i = 0; i < NJ; i++) r.v[i] = x.v[i] * s;
    return r;
}
static inline jet jneg(jet x) { return jscale(x, -1.0); }
static inline jet jsin(jet x)
{
    jet r;
    double c = cos(x.a);
    r.a = sin(x.a);
    for (int i = 0; i < NJ; i++) r.v[i] = c * x.v[i];
    return r;
}
static inline jet jacos(jet x)
{
    jet r;
    double t = -1.0 / sqrt(1.0 - x.a * x.a);
    r.a = acos(x.a);
    for (int i = 0; i < NJ; i++) r.v[i] = t * x.v[i];
    return r;
}

typedef struct {
    jet x, y, z, w;
} jquat;

static void jcross(const jet a[3], const jet b[3], jet o[3])
{
    o[0] = jsub(jmul(a[1], b[2]), jmul(a[2], b[1]));
    o[1] = jsub(jmul(a[2], b[0]), jmul(a[0], b[2]));
    o[2] = jsub(jmul(a[0], b[1]), jmul(a[1], b[0]));
}

static void jquat_rot(const jquat *q, const jet v[3], jet o[3])
{
    jet qv[3] = {q->x, q->y, q->z}, uv[3], t[3];
    jcross(qv, v, uv);
    for (int i = 0; i < 3; i++) uv[i] = jadd(uv[i], uv[i]);
    jcross(qv, uv, t);
    for (int i = 0; i < 3; i++) o[i] = jadd(jadd(v[i], jmul(q->w, uv[i])), t[i]);
}

/* Eigen::Quaternion<T>::Identity().slerp(t, other), ICP:116,197 */
static jquat jquat_slerp_from_identity(double t, const jquat *o)
{
    const double one = 1.0 - DBL_EPSILON;
    jet d = o->w; /* identity.dot(other) = other.w */
    jet absD = d.a < 0 ? jneg(d) : d;
    jet scale0, scale1;
    if (absD.a >= one) {
        scale0 = jc(1.0 - t);
        scale1 = jc(t);
    } else {
        jet theta = jacos(absD);
        jet sinTheta = jsin(theta);
        scale0 = jdiv(jsin(jscale(theta, 1.0 - t)), sinTheta);
        scale1 = jdiv(jsin(jscale(theta, t)), sinTheta);
    }
    if (d.a < 0) scale1 = jneg(scale1);
    jquat r;
    r.x = jmul(scale1, o->x);
    r.y = jmul(scale1, o->y);
    r.z = jmul(scale1, o->z);
    r.w = jadd(scale0, jmul(scale1, o->w)); /* scale0 * 1 + scale1 * w */
    return r;
}

/* ------------------------------------------------------------------ residual blocks */

void orc_block_line(orc_block *b, const double f[3], const double pa[3], const double pb[3], double s)
{
    b->kind = 0;
    b->s = s;
    for (int i = 0; i < 3; i++) {
        b->f[i] = f[i];
        b->a[i] = pa[i];
        b->v[i] = pb[i] - pa[i]; /* ICP:255 */
    }
    double n = norm3(b->v); /* ICP:256 */
    for (int i = 0; i < 3; i++) b->v[i] = b->v[i] / n;
}

void orc_block_plane(orc_block *b, const double f[3], const double pa[3], const double pb[3], const double pc[3], double s)
{
    double ab[3], ac[3];
    b->kind = 1;
    b->s = s;
    for (int i = 0; i < 3; i++) {
        b->f[i] = f[i];
        b->a[i] = pa[i];
        ab[i] = pb[i] - pa[i]; /* ICP:328 */
        ac[i] = pc[i] - pa[i]; /* ICP:331 */
    }
    double nab = norm3(ab), nac = norm3(ac);
    for (int i = 0; i < 3; i++) {
        ab[i] = ab[i] / nab; /* ICP:329 */
        ac[i] = ac[i] / nac; /* ICP:332 */
    }
    cross3(ab, ac, b->v); /* ICP:334, NOT re-normalised */
}

/* functor operator() with T = Jet<double,7> (ICP:262-288, 338-366; _mb: ICP:106-134, 187-218).
 * x = {qx,qy,qz,qw,tx,ty,tz}; derivative slot k = ambient parameter k. */
static void block_residual_jet(const orc_block *b, const double pose_last[7], const double x[7], int deblur, jet r[3])
{
    jquat q_incre = {jvar(x[0], 0), jvar(x[1], 1), jvar(x[2], 2), jvar(x[3], 3)}; /* {_q[3],_q[0],_q[1],_q[2]} -> (w,x,y,z) */
    jet t_incre[3] = {jvar(x[4], 4), jvar(x[5], 5), jvar(x[6], 6)};
    jquat q_last = {jc(pose_last[0]), jc(pose_last[1]), jc(pose_last[2]), jc(pose_last[3])};
    jet pt[3] = {jc(b->f[0]), jc(b->f[1]), jc(b->f[2])};
    jet inner[3], pw[3];
    if (deblur) {
        jquat qi = jquat_slerp_from_identity(b->s, &q_incre); /* ICP:116 */
        jet ti[3] = {jscale(t_incre[0], b->s), jscale(t_incre[1], b->s), jscale(t_incre[2], b->s)}; /* ICP:117 */
        jquat_rot(&qi, pt, inner);
        for (int i = 0; i < 3; i++) inner[i] = jadd(inner[i], ti[i]);
    } else {
        jquat_rot(&q_incre, pt, inner);
        for (int i = 0; i < 3; i++) inner[i] = jadd(inner[i], t_incre[i]);
    }
    jquat_rot(&q_last, inner, pw); /* ICP:275 */
    for (int i = 0; i < 3; i++) pw[i] = jadd(pw[i], jc(pose_last[4 + i]));
    jet vac[3];
    for (int i = 0; i < 3; i++) vac[i] = jsub(pw[i], jc(b->a[i]));
    /* vector_project_on_unit_vector: vec_a.dot(vec_b) * vec_b, EM:19-22 */
    jet d = jadd(jadd(jmul(vac[0], jc(b->v[0])), jmul(vac[1], jc(b->v[1]))), jmul(vac[2], jc(b->v[2])));
    for (int i = 0; i < 3; i++) {
        jet proj = jscale(d, b->v[i]);
        if (b->kind == 0)
            r[i] = jsub(vac[i], proj); /* ICP:281 */
        else
            r[i] = proj; /* ICP:356 (m_weigh = 1) */
    }
}

void orc_block_residual(const orc_block *b, const double pose_last[7], const double x[7], int deblur, double r[3])
{
    jet jr[3];
    block_residual_jet(b, pose_last, x, deblur, jr);
    for (int i = 0; i < 3; i++) r[i] = jr[i].a;
}

/* ceres::HuberLoss::Evaluate */
static void huber(double a, double s, double rho[3])
{
    double b = a * a;
    if (s > b) {
        double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = fmax(DBL_MIN, a / r);
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s;
        rho[1] = 1.0;
        rho[2] = 0.0;
    }
}

/* EigenQuaternionParameterization::ComputeJacobian (4x3, row-major), x = (qx,qy,qz,qw) */
static void quat_plus_jacobian(const double x[4], double J[12])
{
    J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
    J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
    J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
    J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
}

/* ProgramEvaluator::Plus : quaternion Plus, t += d then clamp to bounds (ParameterBlock::Plus) */
static void state_plus(const double x[7], const double d[6], double bound, double out[7])
{
    double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd == 0.0) {
        for (int i = 0; i < 4; i++) out[i] = x[i];
    } else {
        double sn = sin(nd) / nd;
        quat dq = {sn * d[0], sn * d[1], sn * d[2], cos(nd)};
        quat q = {x[0], x[1], x[2], x[3]};
        quat r = quat_mul(&dq, &q);
        out[0] = r.x;
        out[1] = r.y;
        out[2] = r.z;
        out[3] = r.w;
    }
    for (int i = 0; i < 3; i++) {
        double v = x[4 + i] + d[3 + i];
        if (bound >= 0) {
            v = fmax(v, -bound);
            v = fmin(v, bound);
        }
        out[4 + i] = v;
    }
}

/* evaluate active blocks: cost, gradient (6), H = J'J (6x6); optional per-block corrected residuals */
static void eval_blocks(const orc_block *blocks, const unsigned char *active, int nb, const double pose_last[7],
                        const double x[7], int deblur, double huber_a, double *cost, double g[6], double H[36],
                        double *residuals_out /* 3*nb or NULL, loss-corrected */)
{
    double P[12];
    quat_plus_jacobian(x, P);
    double c = 0.0;
    if (g) memset(g, 0, sizeof(double) * 6);
    if (H) memset(H, 0, sizeof(double) * 36);
    for (int k = 0; k < nb; k++) {
        if (active && !active[k]) continue;
        jet r[3];
        block_residual_jet(&blocks[k], pose_last, x, deblur, r);
        double s = r[0].a * r[0].a + r[1].a * r[1].a + r[2].a * r[2].a;
        double rho[3];
        huber(huber_a, s, rho);
        c += 0.5 * rho[0];
        double sq = sqrt(rho[1]); /* Corrector with rho''<=0 (or s==0): plain sqrt(rho') scaling */
        if (residuals_out)
            for (int i = 0; i < 3; i++) residuals_out[3 * k + i] = sq * r[i].a;
        if (!g && !H) continue;
        double Jl[3][6];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                double acc = 0.0;
                for (int m = 0; m < 4; m++) acc += r[i].v[m] * P[m * 3 + j];
                Jl[i][j] = sq * acc;
            }
            for (int j = 0; j < 3; j++) Jl[i][3 + j] = sq * r[i].v[4 + j];
        }
        for (int i = 0; i < 3; i++) {
            double ri = sq * r[i].a;
            for (int a = 0; a < 6; a++) {
                if (g) g[a] += Jl[i][a] * ri;
                if (H)
                    for (int bcol = 0; bcol < 6; bcol++) H[a * 6 + bcol] += Jl[i][a] * Jl[i][bcol];
            }
        }
    }
    *cost = c;
}

void orc_blocks_eval(const orc_block *blocks, int nb, const double pose_last[7], const double x[7],
                     int deblur, double huber_a, double *cost, double g[6], double H[36])
{
    eval_blocks(blocks, NULL, nb, pose_last, x, deblur, huber_a, cost, g, H, NULL);
}

/* dense Cholesky solve of a 6x6 SPD system; returns 0 on failure */
static int chol_solve6(const double A[36], const double b[6], double x[6])
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0.0)) return 0;
                L[i * 6 + i] = sqrt(s);
            } else {
                L[i * 6 + j] = s / L[j * 6 + j];
            }
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    for (int i = 0; i < 6; i++)
        if (!isfinite(x[i])) return 0;
    return 1;
}

typedef struct {
    double initial_cost, final_cost;
    int iterations;
    int n_blocks;
} lm_summary;

static double gradient_max_norm(const double x[7], const double g[6], double bound)
{
    double ng[6], xp[7], m = 0.0;
    for (int i = 0; i < 6; i++) ng[i] = -g[i];
    state_plus(x, ng, bound, xp);
    for (int i = 0; i < 7; i++) m = fmax(m, fabs(x[i] - xp[i]));
    return m;
}

/* Ceres 1.14 LineSearch::InterpolatingPolynomialMinimizingStepSize with interpolation_type = CUBIC from the SECOND contraction
 * of a line search on: three samples with value and gradient -- the start (0, f0, g0), the current trial (x1, f1, g1) and the
 * previous one (x2, f2, g2) -- give six constraints, i.e. the interpolating QUINTIC (polynomial.cc FindInterpolatingPolynomial:
 * rows [x^5 .. 1] for a value, [5 x^4 .. 0] for a gradient, solved with a fully pivoted LU; this restatement builds the same
 * interpolant in Newton form by divided differences -- the two agree to rounding, checked against a dense solve in
 * tests/test_hostcheck.py), minimised over [lo, hi]
 * (MinimizePolynomial: the better end point, then every real root of the derivative inside the interval).  Ceres takes the roots
 * from the eigenvalues of the companion matrix and so sees every real root of the quartic derivative.  So does this restatement
 * (round 6): the roots are isolated exactly by the derivative chain -- p' is monotone between consecutive roots of the cubic p'',
 * which is monotone between the roots of the quadratic p''' (closed form) -- so every interval between consecutive break points holds
 * at most one root, found by its sign change and bisected.  (A real root without a sign change is no minimum; real parts of complex
 * roots, which Ceres also evaluates, can never beat the stationary points and end points.)  Rounds 2 - 5 bracketed sign changes on a
 * fixed 32-cell grid and could miss two roots inside one cell.  The first contraction keeps the two-sample cubic below. */
static int g_ls_max_contractions = 0; /* test instrumentation: deepest line search seen */
static long long g_ls_quintic_fits = 0; /* ... and how many three-sample fits ran */
int orc_dbg_ls_max_contractions(int reset)
{
    int v = g_ls_max_contractions;
    if (reset) g_ls_max_contractions = 0;
    return v;
}
long long orc_dbg_ls_quintic_fits(int reset)
{
    long long v = g_ls_quintic_fits;
    if (reset) g_ls_quintic_fits = 0;
    return v;
}
static double qm_poly4(const double k[5], double x) { return fma(fma(fma(fma(k[4], x, k[3]), x, k[2]), x, k[1]), x, k[0]); }
/* the root of the polynomial k in (a, b], where it is monotone (sign change of the end values, or vb == 0); 0 when there is none.
 * Refined by the Illinois form of regula falsi: the secant through the bracket's ends, the retained end's value halved whenever the same
 * end is replaced twice in a row, a bisection step whenever rounding puts the secant point on an end; ends when the iterate stops
 * moving, hits a zero, or the bracket cannot shrink (ll_reg_core.h quintic_interval_root: the same operations) */
static int qm_interval_root(const double k[5], double a, double b, double va, double vb, double *root)
{
    if (vb == 0.0) {
        *root = b;
        return 1;
    }
    if (!((va < 0.0 && vb > 0.0) || (va > 0.0 && vb < 0.0))) return 0;
    double l = a, r = b, wl = va, wr = vb;
    const int neg_left = va < 0.0;
    double x = b;
    int side = 0;
    for (int it = 0; it < 64; it++) {
        double c = (wl * r - wr * l) / (wl - wr);
        if (!(c > l && c < r)) c = 0.5 * (l + r);
        if (c == l || c == r || c == x) {
            x = c;
            break;
        }
        x = c;
        const double vc = qm_poly4(k, c);
        if (vc == 0.0) break;
        if ((vc < 0.0) == neg_left) {
            l = c;
            wl = vc;
            if (side == -1) wr *= 0.5;
            side = -1;
        } else {
            r = c;
            wr = vc;
            if (side == 1) wl *= 0.5;
            side = 1;
        }
    }
    *root = x;
    return 1;
}
/* roots of k inside (lo, hi] given its break points bp[0 .. nb) (ascending, strictly inside): at most nb + 1, ascending */
static int qm_roots_between(const double k[5], double lo, double hi, const double *bp, int nb, double *roots)
{
    int n = 0;
    double a = lo, va = qm_poly4(k, lo);
    for (int i = 0; i <= nb; i++) {
        const double b = (i == nb) ? hi : bp[i];
        const double vb = qm_poly4(k, b);
        double r;
        if (qm_interval_root(k, a, b, va, vb, &r)) roots[n++] = r;
        a = b;
        va = vb;
    }
    return n;
}
static double quintic_min_step(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo, double hi)
{
    /* Newton form on the nodes z = {0, 0, x1, x1, x2} (the sixth, x2 again, closes the table): divided differences with the
     * derivative in place of the quotient at a repeated node */
    const double h1 = x1, h2 = x2, h21 = x2 - x1;
    if (!(h1 != 0.0) || !(h2 != 0.0) || !(h21 != 0.0)) return fmin(fmax(0.5 * x1, lo), hi); /* coincident samples: bisect like an invalid sample */
    const double e01 = g0, e12 = (f1 - f0) / h1, e23 = g1, e34 = (f2 - f1) / h21, e45 = g2;
    const double a0 = (e12 - e01) / h1, a1 = (e23 - e12) / h1, a2 = (e34 - e23) / h21, a3 = (e45 - e34) / h21;
    const double b0 = (a1 - a0) / h1, b1 = (a2 - a1) / h2, b2 = (a3 - a2) / h21;
    const double c0 = (b1 - b0) / h2, c1 = (b2 - b1) / h2;
    const double d0 = (c1 - c0) / h2;
    /* p(x) = f0 + x (e01 + x (a0 + (x - x1) (b0 + (x - x1) (c0 + (x - x2) d0)))); value and derivative by one nested sweep of
     * explicitly fused multiply-adds (IEEE: the same bits in the oracle, the stand-in and on the device) */
#define LL_Q_EVAL(X, PV, DV)                         \
    do {                                             \
        const double x_ = (X);                       \
        const double u1_ = x_ - x1, u2_ = x_ - x2;   \
        double b_ = d0, db_ = 0.0;                   \
        db_ = fma(u2_, db_, b_);                     \
        b_ = fma(u2_, b_, c0);                       \
        db_ = fma(u1_, db_, b_);                     \
        b_ = fma(u1_, b_, b0);                       \
        db_ = fma(u1_, db_, b_);                     \
        b_ = fma(u1_, b_, a0);                       \
        db_ = fma(x_, db_, b_);                      \
        b_ = fma(x_, b_, e01);                       \
        db_ = fma(x_, db_, b_);                      \
        b_ = fma(x_, b_, f0);                        \
        (PV) = b_;                                   \
        (DV) = db_;                                  \
    } while (0)
    /* monomial coefficients: three synthetic multiplications with (x - node) give p = u3 x^5 + u2 x^4 + u1 x^3 + u0 x^2 + e01 x + f0 */
    const double s1 = d0, s0 = fma(-x2, d0, c0);
    const double t2 = s1, t1 = fma(-x1, s1, s0), t0 = fma(-x1, s0, b0);
    const double u3 = t2, u2 = fma(-x1, t2, t1), u1 = fma(-x1, t1, t0), u0 = fma(-x1, t0, a0);
    const double dq[5] = {e01, 2.0 * u0, 3.0 * u1, 4.0 * u2, 5.0 * u3};    /* p'   */
    const double d2[5] = {2.0 * u0, 6.0 * u1, 12.0 * u2, 20.0 * u3, 0.0};  /* p''  */
    const double A = 60.0 * u3, B = 24.0 * u2, C = 6.0 * u1;               /* p''' */
    /* roots of p''' strictly inside (lo, hi), ascending */
    double r3[2];
    int n3 = 0;
    {
        double q0 = 0.0, q1 = 0.0;
        int n = 0;
        if (A == 0.0) {
            if (B != 0.0) q0 = -C / B, n = 1;
        } else {
            const double disc = fma(B, B, -4.0 * A * C);
            if (disc >= 0.0) {
                const double sq = sqrt(disc);
                const double qq = -0.5 * (B + (B < 0.0 ? -sq : sq));
                q0 = qq / A;
                n = 1;
                if (qq != 0.0) {
                    q1 = C / qq;
                    n = 2;
                    if (q1 < q0) {
                        const double t = q0;
                        q0 = q1;
                        q1 = t;
                    }
                }
            }
        }
        if (n >= 1 && q0 > lo && q0 < hi) r3[n3++] = q0;
        if (n >= 2 && q1 > lo && q1 < hi && !(n3 == 1 && q1 == r3[0])) r3[n3++] = q1;
    }
    double r2[3], r1[4];
    int n2 = qm_roots_between(d2, lo, hi, r3, n3, r2);
    if (n2 > 0 && !(r2[n2 - 1] < hi)) n2--;
    const int n1 = qm_roots_between(dq, lo, hi, r2, n2, r1);
    double best_x = lo, best_v, vh, da, dh;
    LL_Q_EVAL(lo, best_v, da);
    LL_Q_EVAL(hi, vh, dh);
    (void)da;
    (void)dh;
    if (!(best_v < vh)) { /* MinimizePolynomial: x_min wins only when strictly smaller */
        best_v = vh;
        best_x = hi;
    }
    for (int i = 0; i < n1; i++) {
        double v, dv;
        LL_Q_EVAL(r1[i], v, dv);
        (void)dv;
        if (v < best_v) {
            best_v = v;
            best_x = r1[i];
        }
    }
#undef LL_Q_EVAL
    return best_x;
}
/* test tap (tests/test_hostcheck.py: the adversarial fits) */
double orc_dbg_quintic_min_step(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo, double hi)
{
    return quintic_min_step(f0, g0, x1, f1, g1, x2, f2, g2, lo, hi);
}

/* minimiser of the cubic Hermite interpolant through (0,f0,g0) and (x1,f1,g1) on [lo,hi] */
static double cubic_min_step(double f0, double g0, double x1, double f1, double g1, double lo, double hi)
{
    /* p(x) = a x^3 + b x^2 + g0 x + f0 with p(x1)=f1, p'(x1)=g1 */
    double x12 = x1 * x1, x13 = x12 * x1;
    /* [x13 x12; 3x12 2x1] [a b]' = [f1 - f0 - g0 x1; g1 - g0] */
    double r0 = f1 - f0 - g0 * x1, r1 = g1 - g0;
    double det = x13 * 2.0 * x1 - x12 * 3.0 * x12; /* = -x1^4 */
    double a = (r0 * 2.0 * x1 - x12 * r1) / det;
    double b = (x13 * r1 - 3.0 * x12 * r0) / det;
    double best_x = lo, best_v;
#define POLY(x) (((a * (x) + b) * (x) + g0) * (x) + f0)
    best_v = POLY(lo);
    double vh = POLY(hi);
    if (vh < best_v) {
        best_v = vh;
        best_x = hi;
    }
    /* p'(x) = 3a x^2 + 2b x + g0 */
    double A = 3.0 * a, B = 2.0 * b, C = g0;
    double roots[2];
    int nr = 0;
    if (fabs(A) < 1e-300) {
        if (fabs(B) > 1e-300) roots[nr++] = -C / B;
    } else {
        double disc = B * B - 4.0 * A * C;
        if (disc >= 0) {
            double sq = sqrt(disc);
            roots[nr++] = (-B + sq) / (2.0 * A);
            roots[nr++] = (-B - sq) / (2.0 * A);
        }
    }
    for (int i = 0; i < nr; i++) {
        if (roots[i] > lo && roots[i] < hi) {
            double v = POLY(roots[i]);
            if (v < best_v) {
                best_v = v;
                best_x = roots[i];
            }
        }
    }
#undef POLY
    return best_x;
}

/* ceres::Solve on the active blocks, starting from x (in/out). */
static void lm_solve(const orc_block *blocks, const unsigned char *active, int nb, const double pose_last[7],
                     double x[7], int deblur, double huber_a, double bound, int max_iterations, lm_summary *sum)
{
    double cost, g[6], H[36];
    double cand[7], cand_cost, cand_g[6], cand_H[36];
    double scale[6], diag[6];
    double zero6[6] = {0, 0, 0, 0, 0, 0};

    int n_active = 0;
    for (int k = 0; k < nb; k++) n_active += (!active || active[k]) ? 1 : 0;
    sum->n_blocks = n_active;

    /* IterationZero: project onto the bounds */
    state_plus(x, zero6, bound, cand);
    memcpy(x, cand, sizeof(cand));
    double x_norm = 0;
    for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
    x_norm = sqrt(x_norm);

    eval_blocks(blocks, active, nb, pose_last, x, deblur, huber_a, &cost, g, H, NULL);
    for (int j = 0; j < 6; j++) scale[j] = 1.0 / (1.0 + sqrt(H[j * 6 + j]));
    double gmax = gradient_max_norm(x, g, bound);

    sum->initial_cost = cost;
    sum->final_cost = cost;
    sum->iterations = 0;

    double radius = 1e4, decrease_factor = 2.0;
    int reuse_diagonal = 0, invalid_steps = 0, iteration = 0;

    if (n_active == 0) return; /* Ceres: nothing to optimise */

    for (;;) {
        if (iteration >= max_iterations) break;
        if (gmax <= 1e-10) break;
        if (radius < 1e-32) break;
        iteration++;
        sum->iterations = iteration;

        /* LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system */
        double Hs[36], gs[6], A[36], y[6], step[6], delta[6];
        for (int a = 0; a < 6; a++) {
            gs[a] = g[a] * scale[a];
            for (int b = 0; b < 6; b++) Hs[a * 6 + b] = H[a * 6 + b] * scale[a] * scale[b];
        }
        if (!reuse_diagonal)
            for (int j = 0; j < 6; j++) diag[j] = fmin(fmax(Hs[j * 6 + j], 1e-6), 1e32);
        memcpy(A, Hs, sizeof(A));
        for (int j = 0; j < 6; j++) A[j * 6 + j] += diag[j] / radius;
        int ok = chol_solve6(A, gs, y);
        reuse_diagonal = 1;
        double model_cost_change = 0.0;
        if (ok) {
            for (int j = 0; j < 6; j++) step[j] = -y[j];
            double sg = 0.0, sHs = 0.0;
            for (int a = 0; a < 6; a++) {
                sg += step[a] * gs[a];
                double t = 0.0;
                for (int b = 0; b < 6; b++) t += Hs[a * 6 + b] * step[b];
                sHs += step[a] * t;
            }
            model_cost_change = -sg - 0.5 * sHs;
        }
        if (!ok || !(model_cost_change > 0.0)) {
            /* HandleInvalidStep */
            if (++invalid_steps >= 5) break;
            radius *= 0.5;
            reuse_diagonal = 1;
            continue;
        }
        invalid_steps = 0;
        for (int j = 0; j < 6; j++) delta[j] = step[j] * scale[j];

        /* projected ARMIJO line search (problem is bounds-constrained) */
        double gd = 0.0;
        for (int j = 0; j < 6; j++) gd += g[j] * delta[j];
        state_plus(x, delta, bound, cand);
        eval_blocks(blocks, active, nb, pose_last, cand, deblur, huber_a, &cand_cost, cand_g, cand_H, NULL);
        if (bound >= 0) {
            double step_size = 1.0, cur_cost = cand_cost;
            double cur_g[6];
            memcpy(cur_g, cand_g, sizeof(cur_g));
            double dmax = 0.0;
            for (int j = 0; j < 6; j++) dmax = fmax(dmax, fabs(delta[j]));
            int ls_iter = 0, success = 1;
            int prev_valid = 0; /* `previous` of ArmijoLineSearch::DoSearch: the trial before the current one */
            double prev_x = 0.0, prev_f = 0.0, prev_g = 0.0;
            while (!isfinite(cur_cost) || cur_cost > cost + 1e-4 * gd * step_size) {
                if (++ls_iter >= 20) {
                    success = 0;
                    break;
                }
                if (ls_iter > g_ls_max_contractions) g_ls_max_contractions = ls_iter;
                double new_step, cg = 0.0;
                const int cur_valid = isfinite(cur_cost) ? 1 : 0;
                if (!cur_valid) {
                    new_step = fmin(fmax(step_size * 0.5, 1e-3 * step_size), 0.6 * step_size);
                } else {
                    for (int j = 0; j < 6; j++) cg += cur_g[j] * delta[j];
                    if (prev_valid) g_ls_quintic_fits++;
                    if (prev_valid)
                        new_step = quintic_min_step(cost, gd, step_size, cur_cost, cg, prev_x, prev_f, prev_g, 1e-3 * step_size, 0.6 * step_size);
                    else
                        new_step = cubic_min_step(cost, gd, step_size, cur_cost, cg, 1e-3 * step_size, 0.6 * step_size);
                }
                if (new_step * dmax < 1e-9) {
                    success = 0;
                    break;
                }
                prev_valid = cur_valid;
                prev_x = step_size;
                prev_f = cur_cost;
                prev_g = cg;
                step_size = new_step;
                double sd[6], sx[7], sH[36];
                for (int j = 0; j < 6; j++) sd[j] = delta[j] * step_size;
                state_plus(x, sd, bound, sx);
                eval_blocks(blocks, active, nb, pose_last, sx, deblur, huber_a, &cur_cost, cur_g, sH, NULL);
            }
            if (success && step_size != 1.0) {
                for (int j = 0; j < 6; j++) delta[j] *= step_size;
                state_plus(x, delta, bound, cand);
                eval_blocks(blocks, active, nb, pose_last, cand, deblur, huber_a, &cand_cost, cand_g, cand_H, NULL);
            }
        }
        if (!isfinite(cand_cost)) cand_cost = DBL_MAX;

        /* ParameterToleranceReached */
        double step_norm = 0.0;
        for (int i = 0; i < 7; i++) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
        step_norm = sqrt(step_norm);
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) break;
        /* FunctionToleranceReached */
        double cost_change = cost - cand_cost;
        if (fabs(cost_change) <= 1e-6 * cost) break;

        double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > 1e-3) {
            memcpy(x, cand, sizeof(cand));
            x_norm = 0;
            for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
            x_norm = sqrt(x_norm);
            cost = cand_cost;
            memcpy(g, cand_g, sizeof(g));
            memcpy(H, cand_H, sizeof(H));
            gmax = gradient_max_norm(x, g, bound);
            if (cost < sum->final_cost) sum->final_cost = cost;
            double t = 2.0 * relative_decrease - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            radius = fmin(1e16, radius);
            decrease_factor = 2.0;
            reuse_diagonal = 0;
        } else {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            reuse_diagonal = 1;
        }
    }
}

/* ------------------------------------------------------------------ PCA feature checks (PCR:259-292, 357-389) */

/* eigenvalues (ascending) of a symmetric 3x3 by the trigonometric closed form (Smith 1961), refined by one
 * Newton step per root on the characteristic polynomial; independent of the Jacobi iteration used on the device */
static void sym3_eig(const double A[9], double ev[3])
{
    const double p1 = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double q = (A[0] + A[4] + A[8]) / 3.0;
    if (p1 == 0.0) {
        ev[0] = A[0]; ev[1] = A[4]; ev[2] = A[8];
    } else {
        const double p2 = (A[0] - q) * (A[0] - q) + (A[4] - q) * (A[4] - q) + (A[8] - q) * (A[8] - q) + 2.0 * p1;
        const double p = sqrt(p2 / 6.0);
        double B[9];
        for (int i = 0; i < 9; i++) B[i] = (A[i] - ((i % 4 == 0) ? q : 0.0)) / p;
        const double detB = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) + B[2] * (B[3] * B[7] - B[4] * B[6]);
        double r = detB / 2.0;
        r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        const double phi = acos(r) / 3.0;
        ev[2] = q + 2.0 * p * cos(phi);
        ev[0] = q + 2.0 * p * cos(phi + 2.0943951023931954923 /* 2 pi / 3 */);
        ev[1] = 3.0 * q - ev[0] - ev[2];
    }
    /* Newton polish on det(A - x I) = 0 (characteristic cubic), keeps ~1e-16 relative accuracy near repeated roots */
    const double c2 = -(A[0] + A[4] + A[8]);
    const double c1 = A[0] * A[4] + A[0] * A[8] + A[4] * A[8] - A[1] * A[1] - A[2] * A[2] - A[5] * A[5];
    const double c0 = -(A[0] * (A[4] * A[8] - A[5] * A[5]) - A[1] * (A[1] * A[8] - A[5] * A[2]) + A[2] * (A[1] * A[5] - A[4] * A[2]));
    for (int k = 0; k < 3; k++)
        for (int itn = 0; itn < 2; itn++) {
            const double x = ev[k];
            const double f = ((x + c2) * x + c1) * x + c0, df = (3.0 * x + 2.0 * c2) * x + c1;
            if (df != 0.0 && isfinite(f / df)) ev[k] = x - f / df;
        }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2 - i; j++)
            if (ev[j] > ev[j + 1]) { double t = ev[j]; ev[j] = ev[j + 1]; ev[j + 1] = t; }
}

static int pca_check(int is_plane, const float *map, int stride, const int32_t idx[5])
{
    double center[3] = {0, 0, 0}, pts[5][3];
    for (int j = 0; j < 5; j++)
        for (int c = 0; c < 3; c++) {
            pts[j][c] = (double)map[(size_t)idx[j] * stride + c]; /* PCR:263-265 */
            center[c] = center[c] + pts[j][c];
        }
    for (int c = 0; c < 3; c++) center[c] = center[c] / 5.0; /* PCR:270 */
    double cov[9] = {0};
    for (int j = 0; j < 5; j++) { /* PCR:274-278 */
        double z[3] = {pts[j][0] - center[0], pts[j][1] - center[1], pts[j][2] - center[2]};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) cov[r * 3 + c] += z[r] * z[c];
    }
    double ev[3];
    sym3_eig(cov, ev);
    if (is_plane) return (ev[2] > 3 * ev[0]) && (ev[2] < 10 * ev[1]); /* PCR:380-381 */
    return ev[2] > 3 * ev[1];                                          /* PCR:284 */
}

int orc_pca_check(int is_plane, const float pts[15], double ev_out[3])
{
    const int32_t idx[5] = {0, 1, 2, 3, 4};
    double center[3] = {0, 0, 0}, cov[9] = {0};
    for (int j = 0; j < 5; j++)
        for (int c = 0; c < 3; c++) center[c] = center[c] + (double)pts[j * 3 + c];
    for (int c = 0; c < 3; c++) center[c] = center[c] / 5.0;
    for (int j = 0; j < 5; j++) {
        double z[3] = {pts[j * 3] - center[0], pts[j * 3 + 1] - center[1], pts[j * 3 + 2] - center[2]};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) cov[r * 3 + c] += z[r] * z[c];
    }
    sym3_eig(cov, ev_out);
    return pca_check(is_plane, pts, 3, idx);
}

/* ------------------------------------------------------------------ sub-sampling (PCR:232-238, 339-345, 438-458) */

/* uniform [0,1) from (seed, stream, ICP iteration, item index); stream 0 corner features, 1 surface features, 2 blocks */
static float subsample_uniform(uint32_t seed, uint32_t stream, uint32_t iter, uint32_t index)
{
    uint32_t h = seed ^ (stream * 0x9E3779B9u) ^ (iter * 0x85EBCA6Bu) ^ (index * 0xC2B2AE35u);
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ PCR helpers */

/* refine_blur PCR:128-141 (float arithmetic) */
static float refine_blur(int deblur, float in_blur, float min_blur, float max_blur)
{
    float res = 1.0f;
    if (deblur) {
        res = (in_blur - min_blur) / (max_blur - min_blur);
        if (!isfinite(res) || res > 1.0)
            return 1.0f;
        else
            return res;
    }
    return res;
}

void orc_point_to_map(const double pose[7], const float p[3], float out[3])
{
    quat q = pose_q(pose);
    double v[3] = {p[0], p[1], p[2]}, o[3];
    quat_rot(&q, v, o);
    out[0] = (float)(o[0] + pose[4]); /* PCR:629,656-658 */
    out[1] = (float)(o[1] + pose[5]);
    out[2] = (float)(o[2] + pose[6]);
}

void orc_cloud_transform(const double pose[7], const float *in_xyzi, float *out_xyzi, int n)
{
    for (int i = 0; i < n; i++) {
        float o[3];
        float inten = in_xyzi[4 * i + 3];
        orc_point_to_map(pose, &in_xyzi[4 * i], o);
        out_xyzi[4 * i] = o[0];
        out_xyzi[4 * i + 1] = o[1];
        out_xyzi[4 * i + 2] = o[2];
        out_xyzi[4 * i + 3] = inten; /* PCR:659 */
    }
}

typedef struct {
    double theta;
    double hat[9], hat_sq[9];
} interp_state;

/* compute_interpolatation_rodrigue PCR:607-620 */
static void compute_interp(const quat *q_in, interp_state *st)
{
    double v[3] = {q_in->x, q_in->y, q_in->z};
    double n = norm3(v), w = q_in->w, axis[3];
    if (w < 0) n = -n; /* Eigen AngleAxis(QuaternionBase): if(q.w()<0) n = -n */
    if (n != 0.0) {
        st->theta = 2.0 * atan2(n, fabs(w));
        for (int i = 0; i < 3; i++) axis[i] = v[i] / n;
    } else {
        st->theta = 0;
        axis[0] = 1;
        axis[1] = 0;
        axis[2] = 0;
    }
    double an = norm3(axis); /* PCR:611 */
    for (int i = 0; i < 3; i++) axis[i] /= an;
    memset(st->hat, 0, sizeof(st->hat));
    st->hat[0 * 3 + 1] = -axis[2];
    st->hat[1 * 3 + 0] = axis[2];
    st->hat[0 * 3 + 2] = axis[1];
    st->hat[2 * 3 + 0] = -axis[1];
    st->hat[1 * 3 + 2] = -axis[0];
    st->hat[2 * 3 + 1] = axis[0];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += st->hat[i * 3 + k] * st->hat[k * 3 + j];
            st->hat_sq[i * 3 + j] = s; /* PCR:512 */
        }
}

/* pointAssociateToMap PCR:622-661 */
static void point_associate(int deblur, const double pose_curr[7], const double pose_last[7], const double t_incre[3],
                            const interp_state *st, const float p[3], double s, float out[3])
{
    if (deblur == 0 || s == 1.0) {
        orc_point_to_map(pose_curr, p, out);
        return;
    }
    double pc[3] = {p[0], p[1], p[2]};
    double T[3] = {t_incre[0] * (s * 1.0), t_incre[1] * (s * 1.0), t_incre[2] * (s * 1.0)}; /* PCR:641 */
    double th = st->theta * s;
    double sn = sin(th), cs1 = 1 - cos(th);
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sn * st->hat[i] + cs1 * st->hat_sq[i]; /* PCR:645 */
    double inner[3], o[3];
    for (int i = 0; i < 3; i++) inner[i] = R[i * 3] * pc[0] + R[i * 3 + 1] * pc[1] + R[i * 3 + 2] * pc[2] + T[i];
    quat ql = pose_q(pose_last);
    quat_rot(&ql, inner, o); /* PCR:646 */
    out[0] = (float)(o[0] + pose_last[4]);
    out[1] = (float)(o[1] + pose_last[5]);
    out[2] = (float)(o[2] + pose_last[6]);
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* compute_inlier_residual_threshold PCR:153-161: std::set<double> (dedups) -> element at (int)(ratio*size) */
static double inlier_threshold(const double *l1, int n, double ratio, double fallback)
{
    if (n == 0) return fallback;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(tmp, l1, sizeof(double) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(double), cmp_double);
    int u = 0;
    for (int i = 0; i < n; i++)
        if (u == 0 || tmp[i] != tmp[u - 1]) tmp[u++] = tmp[i];
    double r = tmp[(int)(ratio * u)];
    free(tmp);
    return r;
}

int orc_reg_solve(const orc_kdtree *tree_corner, const float *map_corner, int64_t n_map_corner,
                  const orc_kdtree *tree_surf, const float *map_surf, int64_t n_map_surf, int map_stride,
                  const float *scan_corner, int n_corner, const float *scan_surf, int n_surf,
                  const orc_reg_params *prm, const double pose_last[7], double pose_curr[7],
                  double pose_incre[7], orc_reg_report *rep)
{
    memset(rep, 0, sizeof(*rep));
    rep->accepted = 1;
    /* PCR:199 gate */
    if (!(n_map_corner > 0 && n_map_surf > 50 && prm->current_frame_index > prm->mapping_init_accumulate_frames)) {
        rep->gated = 1;
        return 1;
    }
    const int deblur = prm->if_motion_deblur;
    const int kl = prm->line_search_num, kp = prm->plane_search_num;
    int cap = n_corner + n_surf;
    orc_block *blocks = (orc_block *)malloc(sizeof(orc_block) * (size_t)(cap > 0 ? cap : 1));
    unsigned char *active = (unsigned char *)malloc((size_t)(cap > 0 ? cap : 1));
    double *resid = (double *)malloc(sizeof(double) * 3 * (size_t)(cap > 0 ? cap : 1));
    double *l1 = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    int32_t *blk_query = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1)); /* position of the block's feature: corner i, or n_corner + surface i */
    int32_t nn_idx[16];
    float nn_d2[16];
    quat q_last_opt = {0, 0, 0, 1};
    double t_last_opt[3] = {0, 0, 0};
    interp_state st;
    memset(&st, 0, sizeof(st));
    lm_summary sum;
    memset(&sum, 0, sizeof(sum));
    double inlier_thr = 0.0;
    double angular_diff = 0, t_diff = 0;
    int corner_avail = 0, surf_avail = 0;
    int it;
    const quat ql = pose_q(pose_last);

    for (it = 0; it < prm->icp_max_iterations; it++) { /* PCR:211 */
        int nb = 0;
        corner_avail = 0;
        surf_avail = 0;
        const uint32_t seed = (uint32_t)prm->subsample_seed;
        const int max_blk = prm->maximum_allow_residual_block;
        for (int i = 0; i < n_corner; i++) { /* PCR:230-333 */
            if (seed && n_corner > 2 * max_blk && subsample_uniform(seed, 0u, (uint32_t)it, (uint32_t)i) * (float)n_corner > (float)(2 * max_blk))
                continue; /* PCR:232-238 */
            const float *po = &scan_corner[4 * i];
            if (!isfinite(po[0]) || !isfinite(po[1]) || !isfinite(po[2])) continue;
            float s = refine_blur(deblur, po[3], prm->minimum_pt_time_stamp, prm->maximum_pt_time_stamp);
            float sel[3];
            point_associate(deblur, pose_curr, pose_last, &pose_incre[4], &st, po, (double)s, sel);
            if (orc_kdtree_knn(tree_corner, sel, kl, nn_idx, nn_d2) != kl) continue; /* PCR:249 */
            if ((double)nn_d2[kl - 1] < prm->maximum_dis_line_for_match) {           /* PCR:254 */
                if (prm->if_line_feature_check && !pca_check(0, map_corner, map_stride, nn_idx)) continue; /* PCR:259-292,328-331 */
                if (prm->icp_line) {
                    const float *pa = &map_corner[(size_t)nn_idx[0] * map_stride];
                    const float *pb = &map_corner[(size_t)nn_idx[1] * map_stride];
                    double a[3] = {pa[0], pa[1], pa[2]}, b[3] = {pb[0], pb[1], pb[2]};
                    double d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
                    if (norm3(d) < 0.0001) continue; /* PCR:302 */
                    double f[3] = {po[0], po[1], po[2]};
                    blk_query[nb] = i;
                    orc_block_line(&blocks[nb++], f, a, b, deblur ? (double)s * 1.0 : 1.0);
                    corner_avail++;
                }
            }
        }
        for (int i = 0; i < n_surf; i++) { /* PCR:336-432 */
            if (seed && n_surf > 2 * max_blk && subsample_uniform(seed, 1u, (uint32_t)it, (uint32_t)i) * (float)n_surf > (float)(2 * max_blk))
                continue; /* PCR:339-345 */
            const float *po = &scan_surf[4 * i];
            if (!isfinite(po[0]) || !isfinite(po[1]) || !isfinite(po[2])) continue; /* defined deviation */
            float s = refine_blur(deblur, po[3], prm->minimum_pt_time_stamp, prm->maximum_pt_time_stamp);
            float sel[3];
            point_associate(deblur, pose_curr, pose_last, &pose_incre[4], &st, po, (double)s, sel);
            if (orc_kdtree_knn(tree_surf, sel, kp, nn_idx, nn_d2) != kp) continue; /* PCR:351 */
            if ((double)nn_d2[kp - 1] < prm->maximum_dis_plane_for_match) {        /* PCR:353 */
                if (prm->if_plane_feature_check && !pca_check(1, map_surf, map_stride, nn_idx)) continue; /* PCR:357-389,427-430 (surface cloud: bug fixed) */
                if (prm->icp_plane) {
                    const float *pa = &map_surf[(size_t)nn_idx[0] * map_stride];
                    const float *pb = &map_surf[(size_t)nn_idx[kp / 2] * map_stride];
                    const float *pc = &map_surf[(size_t)nn_idx[kp - 1] * map_stride];
                    double a[3] = {pa[0], pa[1], pa[2]}, b[3] = {pb[0], pb[1], pb[2]}, c[3] = {pc[0], pc[1], pc[2]};
                    double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                    if (norm3(ab) == 0.0 || norm3(ac) == 0.0) continue; /* defined deviation: NaN normal */
                    double f[3] = {po[0], po[1], po[2]};
                    blk_query[nb] = n_corner + i;
                    orc_block_plane(&blocks[nb++], f, a, b, c, deblur ? (double)s * 1.0 : 1.0);
                }
                surf_avail++; /* PCR:425 */
            }
        }
        for (int k = 0; k < nb; k++) active[k] = 1;
        if (seed && nb > max_blk) { /* PCR:438-458: "Number of residual blocks too Large, drop them" */
            const float threshold_to_reserve = (float)max_blk / (float)nb;
            for (int k = 0; k < nb; k++)
                if (subsample_uniform(seed, 2u, (uint32_t)it, (uint32_t)blk_query[k]) > threshold_to_reserve) active[k] = 0;
        }

        /* prerun solve, PCR:463-474 */
        lm_summary pre;
        lm_solve(blocks, active, nb, pose_last, pose_incre, deblur, prm->huber_a, (double)prm->para_max_speed,
                 prm->ceres_prerun_times, &pre);
        rep->lm_iterations_total += pre.iterations;

        /* problem.Evaluate (loss applied) -> L1 per block -> threshold -> prune, PCR:476-499 */
        {
            double c;
            eval_blocks(blocks, active, nb, pose_last, pose_incre, deblur, prm->huber_a, &c, NULL, NULL, resid);
            /* only the blocks still in the problem are evaluated (residual_block_ids after the drop of PCR:438-458) */
            int na = 0;
            double *l1a = (double *)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
            for (int k = 0; k < nb; k++) {
                l1[k] = fabs(resid[3 * k]) + fabs(resid[3 * k + 1]) + fabs(resid[3 * k + 2]);
                if (active[k]) l1a[na++] = l1[k];
            }
            double thr = inlier_threshold(l1a, na, prm->inlier_ratio, prm->inliner_dis);
            free(l1a);
            inlier_thr = fmax(prm->inliner_dis, thr);
            for (int k = 0; k < nb; k++)
                if (active[k] && l1[k] > inlier_thr) active[k] = 0;
        }

        /* final solve, PCR:501-508 */
        lm_solve(blocks, active, nb, pose_last, pose_incre, deblur, prm->huber_a, (double)prm->para_max_speed,
                 prm->ceres_max_iterations, &sum);
        rep->lm_iterations_total += sum.iterations;

        quat q_incre = pose_q(pose_incre);
        if (deblur) compute_interp(&q_incre, &st); /* PCR:509-513 */
        {
            double tw[3];
            quat_rot(&ql, &pose_incre[4], tw); /* PCR:514 */
            pose_curr[4] = tw[0] + pose_last[4];
            pose_curr[5] = tw[1] + pose_last[5];
            pose_curr[6] = tw[2] + pose_last[6];
            quat qc = quat_mul(&ql, &q_incre); /* PCR:515 */
            pose_curr[0] = qc.x;
            pose_curr[1] = qc.y;
            pose_curr[2] = qc.z;
            pose_curr[3] = qc.w;
            angular_diff = (double)((float)quat_angular_distance(&qc, &ql) * 57.3); /* PCR:517: (float) cast then * 57.3 */
            double dt[3] = {pose_curr[4] - pose_last[4], pose_curr[5] - pose_last[5], pose_curr[6] - pose_last[6]};
            t_diff = norm3(dt);
        }
        double dto[3] = {t_last_opt[0] - pose_incre[4], t_last_opt[1] - pose_incre[5], t_last_opt[2] - pose_incre[6]};
        if (!prm->force_all_iterations && quat_angular_distance(&q_last_opt, &q_incre) < 57.3 * prm->minimum_icp_R_diff &&
            norm3(dto) < prm->minimum_icp_T_diff) { /* PCR:521-526 */
            it++; /* count this iteration as executed */
            break;
        } else {
            q_last_opt = q_incre;
            t_last_opt[0] = pose_incre[4];
            t_last_opt[1] = pose_incre[5];
            t_last_opt[2] = pose_incre[6];
        }
    }

    rep->icp_iterations = it;
    rep->n_blocks_last = sum.n_blocks;
    rep->corner_avail = corner_avail;
    rep->surf_avail = surf_avail;
    rep->final_cost = sum.final_cost;
    rep->initial_cost = sum.initial_cost;
    rep->angular_diff_deg = angular_diff;
    rep->t_diff = t_diff;
    rep->inlier_threshold = inlier_thr * sum.final_cost / sum.initial_cost; /* PCR:559 */

    free(blocks);
    free(active);
    free(resid);
    free(l1);
    free(blk_query);

    /* PCR:561-573; minimize_cost is a float (PCR:192,519) */
    float minimize_cost = (float)sum.final_cost;
    if (angular_diff > prm->para_max_angular_rate || minimize_cost > prm->max_final_cost) {
        memcpy(pose_curr, pose_last, sizeof(double) * 7);
        rep->accepted = 0;
        return 0;
    }
    return 1;
}
