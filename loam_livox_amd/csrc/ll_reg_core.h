// ll_reg_core.h -- per-thread arithmetic of the scan-to-map registrar shared by the HIP kernels
// (ll_reg_kernels.hip) and tests/hostcheck: pose algebra, residual-block construction, per-block
// cost / gradient / Gauss-Newton accumulation, and the Levenberg-Marquardt step controller that replaces the
// ceres::Solve calls of hku-mars/loam_livox source/point_cloud_registration.hpp:474,508.
//
// Formulation (SURVEY App. C.2 "replacement contract").  With x = (q_inc, t_inc) the increment optimised by
// the reference and T_last the pose before the scan, the reference residuals (source/ceres_icp.hpp:262-288,
// 338-366) are  r = A (R_last (R_inc f + t_inc) + t_last - a),  A = I - u u^T (line) or n n^T (plane, n not
// normalised).  We pre-rotate the block constants into the frame of T_last:  a' = R_last^T (a - t_last),
// u' = R_last^T u, so  r' = R_last^T r = A' (R_inc f + t_inc - a').  |r'| = |r|, and J'^T J' = J^T J.
// The tangent space is Ceres' EigenQuaternionParameterization: q+ = [sin|d| d/|d|, cos|d|] (x) q, i.e.
// R_inc+ = Exp(2 d) R_inc, hence  d(R_inc f)/dd = -2 [R_inc f]x.
#pragma once
#include <float.h>

#include "ll_fe_core.h"

namespace ll {

// ---------------------------------------------------------------------------------------------- pose algebra

LL_HD void cross3(const double a[3], const double b[3], double o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
LL_HD double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Eigen QuaternionBase::_transformVector: v + w*uv + qv x uv, uv = 2 (qv x v). q = (x,y,z,w).
LL_HD void quat_rot(const double q[4], const double v[3], double o[3])
{
    double uv[3], t[3];
    cross3(q, v, uv);
    uv[0] += uv[0];
    uv[1] += uv[1];
    uv[2] += uv[2];
    cross3(q, uv, t);
    o[0] = v[0] + q[3] * uv[0] + t[0];
    o[1] = v[1] + q[3] * uv[1] + t[1];
    o[2] = v[2] + q[3] * uv[2] + t[2];
}
// rotate by the conjugate (R^T v)
LL_HD void quat_rot_inv(const double q[4], const double v[3], double o[3])
{
    const double qc[4] = {-q[0], -q[1], -q[2], q[3]};
    quat_rot(qc, v, o);
}
// Eigen quaternion product a*b, (x,y,z,w) storage
LL_HD void quat_mul(const double a[4], const double b[4], double r[4])
{
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    r[0] = x;
    r[1] = y;
    r[2] = z;
    r[3] = w;
}
// Eigen angularDistance: 2 atan2(|vec(a b*)|, |w(a b*)|)
LL_HD double quat_angular_distance(const double a[4], const double b[4])
{
    const double bc[4] = {-b[0], -b[1], -b[2], b[3]};
    double d[4];
    quat_mul(a, bc, d);
    return 2.0 * atan2(sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), fabs(d[3]));
}
// rotation matrix (row-major) that applies the same map as quat_rot for a unit quaternion
LL_HD void quat_to_mat(const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - z * w);
    R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);
    R[7] = 2 * (y * z + x * w);
    R[8] = 1 - 2 * (x * x + y * y);
}

// pointAssociateToMap, no-deblur branch (point_cloud_registration.hpp:629,656-658): double math, float store
LL_HD void point_to_map(const double pose[7], float px, float py, float pz, float out[3])
{
    const double v[3] = {(double)px, (double)py, (double)pz};
    double o[3];
    quat_rot(pose, v, o);
    out[0] = (float)(o[0] + pose[4]);
    out[1] = (float)(o[1] + pose[5]);
    out[2] = (float)(o[2] + pose[6]);
}

// refine_blur (point_cloud_registration.hpp:128-141), float arithmetic
LL_HD float refine_blur(int deblur, float in_blur, float min_blur, float max_blur)
{
    if (!deblur) return 1.0f;
    const float res = (in_blur - min_blur) / (max_blur - min_blur);
    if (!ll_isfinite(res) || res > 1.0f) return 1.0f;
    return res;
}

// ---------------------------------------------------------------------------------------------- a13: sub-sampling
// point_cloud_registration.hpp:232-238, 339-345 (features) and :438-458 (residual blocks) drop work at random when a scan
// has more than 2 x / 1 x maximum_allow_residual_block of it.  The reference draws from mt19937(random_device), which
// nobody can reproduce; here the uniform numbers come from a counter-based hash of (seed, stream, ICP iteration, item
// index), so the same keep / drop rules give the same result on every run and on the oracle:
//   feature i of a kind with n > 2 M features is skipped when        u * n > 2 M            (:234, :341)
//   block i of a problem with n_blocks > M blocks is removed when    u > (float)M / n_blocks (:442-449)
// streams: 0 = corner features, 1 = surface features, 2 = residual blocks (index = position of the block's query in the
// corner-then-surface order).
LL_HD float subsample_uniform(unsigned int seed, unsigned int stream, unsigned int iter, unsigned int index)
{
    unsigned int h = seed ^ (stream * 0x9E3779B9u) ^ (iter * 0x85EBCA6Bu) ^ (index * 0xC2B2AE35u);
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}
LL_HD bool subsample_skip_feature(unsigned int seed, int kind, int iter, int i, int n, int max_blocks)
{
    if (!seed || n <= 2 * max_blocks) return false;
    return subsample_uniform(seed, (unsigned int)kind, (unsigned int)iter, (unsigned int)i) * (float)n > (float)(2 * max_blocks);
}
LL_HD bool subsample_drop_block(unsigned int seed, int iter, int j, int n_blocks, int max_blocks)
{
    if (!seed || n_blocks <= max_blocks) return false;
    const float threshold_to_reserve = (float)max_blocks / (float)n_blocks;
    return subsample_uniform(seed, 2u, (unsigned int)iter, (unsigned int)j) > threshold_to_reserve;
}

// ---------------------------------------------------------------------------------------------- residual blocks

enum : int { BLK_NONE = 0, BLK_LINE = 1, BLK_PLANE = 2, BLK_ACTIVE = 4 };  // (8: counted as available, PCR:325,425)

// line block from the two nearest map points (point_cloud_registration.hpp:300-303, ceres_icp.hpp:255-256).
// a_out / v_out are expressed in the frame of pose_last.  Returns false when |a-b| < 1e-4 (:302).
LL_HD bool block_line(const double pose_last[7], const double pa[3], const double pb[3], double a_out[3], double v_out[3])
{
    double d[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    if (sqrt(dot3(d, d)) < 0.0001) return false;
    double u[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double n = sqrt(dot3(u, u));
    u[0] = u[0] / n;
    u[1] = u[1] / n;
    u[2] = u[2] / n;
    const double rel[3] = {pa[0] - pose_last[4], pa[1] - pose_last[5], pa[2] - pose_last[6]};
    quat_rot_inv(pose_last, rel, a_out);
    quat_rot_inv(pose_last, u, v_out);
    return true;
}

// plane block from neighbours 0, k/2, k-1 (point_cloud_registration.hpp:416-418, ceres_icp.hpp:328-334); a_out[0] = n'.a';
// n = (ab/|ab|) x (ac/|ac|) is NOT re-normalised.  Degenerate triples (a==b or a==c) are skipped (the
// reference would produce NaN residuals).
LL_HD bool block_plane(const double pose_last[7], const double pa[3], const double pb[3], const double pc[3],
                       double a_out[3], double v_out[3])
{
    double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    double ac[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
    const double nab = sqrt(dot3(ab, ab)), nac = sqrt(dot3(ac, ac));
    if (nab == 0.0 || nac == 0.0) return false;
    for (int i = 0; i < 3; i++) {
        ab[i] = ab[i] / nab;
        ac[i] = ac[i] / nac;
    }
    double n[3];
    cross3(ab, ac, n);
    const double rel[3] = {pa[0] - pose_last[4], pa[1] - pose_last[5], pa[2] - pose_last[6]};
    double a_loc[3];
    quat_rot_inv(pose_last, rel, a_loc);
    quat_rot_inv(pose_last, n, v_out);
    // a plane block only ever needs n'.a' (r = ((p - a').n') n'), so it is stored as one scalar: 8 B instead of 24 B
    // per block on every cost evaluation
    a_out[0] = dot3(v_out, a_loc);
    a_out[1] = 0.0;
    a_out[2] = 0.0;
    return true;
}

// the early-out of block_plane alone (same arithmetic): the plane-table solver path decides a block's flag at build time
// and computes the plane constants once per distinct neighbour triple inside the solver
LL_HD bool plane_degenerate(const double pa[3], const double pb[3], const double pc[3])
{
    const double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double ac[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
    const double nab = sqrt(dot3(ab, ab)), nac = sqrt(dot3(ac, ac));
    return nab == 0.0 || nac == 0.0;
}

// ---------------------------------------------------------------------------------------------- K7: PCA checks
// Optional neighbourhood checks of point_cloud_registration.hpp:259-292 (line) and :357-389 (plane), switched by
// IF_LINE_FEATURE_CHECK / IF_PLANE_FEATURE_CHECK (:46,48; 0 in every shipped configuration).  Covariance of the
// five neighbours (not divided by n) in double, eigenvalues ascending (Eigen::SelfAdjointEigenSolver order):
//   line  ok  <=>  l2 > 3 l1                       (:284)
//   plane ok  <=>  l2 > 3 l0  &&  l2 < 10 l1       (:380-381)
// The reference's plane branch reads laser_cloud_corner_from_map with surface indices (:361-363), a bug (SURVEY
// App. B-5); this implementation uses the surface cloud.
LL_HD void sym3_eigenvalues(const double A_in[9], double ev[3])
{
    // cyclic Jacobi rotations on the symmetric 3x3 (converges to ~1 ulp in <= 6 sweeps)
    double a[9];
    for (int i = 0; i < 9; i++) a[i] = A_in[i];
    for (int sweep = 0; sweep < 12; sweep++) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const double apq = a[p * 3 + q];
            if (apq == 0.0) continue;
            const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            // A <- J^T A J for the rotation in the (p,q) plane
            for (int k = 0; k < 3; k++) {
                const double akp = a[k * 3 + p], akq = a[k * 3 + q];
                a[k * 3 + p] = c * akp - sn * akq;
                a[k * 3 + q] = sn * akp + c * akq;
            }
            for (int k = 0; k < 3; k++) {
                const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                a[p * 3 + k] = c * apk - sn * aqk;
                a[q * 3 + k] = sn * apk + c * aqk;
            }
        }
    }
    double e0 = a[0], e1 = a[4], e2 = a[8], tmp;
    if (e0 > e1) { tmp = e0; e0 = e1; e1 = tmp; }
    if (e1 > e2) { tmp = e1; e1 = e2; e2 = tmp; }
    if (e0 > e1) { tmp = e0; e0 = e1; e1 = tmp; }
    ev[0] = e0;
    ev[1] = e1;
    ev[2] = e2;
}

// pts: the five neighbours (float coordinates promoted to double as at :263-265)
LL_HD bool pca_check(int is_plane, const double pts[5][3])
{
    double center[3] = {0, 0, 0};
    for (int j = 0; j < 5; j++)
        for (int c = 0; c < 3; c++) center[c] = center[c] + pts[j][c];
    for (int c = 0; c < 3; c++) center[c] = center[c] / 5.0;  // center / ((float) line_search_num), :270
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 5; j++) {
        const double z[3] = {pts[j][0] - center[0], pts[j][1] - center[1], pts[j][2] - center[2]};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) cov[r * 3 + c] = cov[r * 3 + c] + z[r] * z[c];
    }
    double ev[3];
    sym3_eigenvalues(cov, ev);
    if (is_plane) return (ev[2] > 3 * ev[0]) && (ev[2] < 10 * ev[1]);
    return ev[2] > 3 * ev[1];
}

// ceres::HuberLoss(a): rho(s), rho'(s)
LL_HD void huber(double a, double s, double *rho0, double *rho1)
{
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        *rho0 = 2.0 * a * r - b;
        *rho1 = fmax(DBL_MIN, a / r);
    } else {
        *rho0 = s;
        *rho1 = 1.0;
    }
}

// Accumulator layout: acc[0..20] = upper triangle of H = sum rho' J^T J (row-major, a <= b),
// acc[21..26] = g = sum rho' J^T r, acc[27] = cost = 1/2 sum rho(|r|^2).
#define LL_NACC 28
LL_HD int hidx(int a, int b) { return a * 6 - (a * (a - 1)) / 2 + (b - a); }  // a <= b

// residual (frame of pose_last) of one block at increment (R = R_inc row-major, t = t_inc):
// returns s = |r|^2; fills y = R f, r.
LL_HD double block_residual(int kind, const double R[9], const double t[3], const double f[3], const double a[3],
                            const double v[3], double y[3], double r[3], double *dd_out)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    y[0] = R[0] * f[0] + R[1] * f[1] + R[2] * f[2];
    y[1] = R[3] * f[0] + R[4] * f[1] + R[5] * f[2];
    y[2] = R[6] * f[0] + R[7] * f[1] + R[8] * f[2];
    double dd;
    if (kind == BLK_LINE) {
        const double d[3] = {y[0] + t[0] - a[0], y[1] + t[1] - a[1], y[2] + t[2] - a[2]};
        dd = dot3(d, v);
        r[0] = d[0] - dd * v[0];
        r[1] = d[1] - dd * v[1];
        r[2] = d[2] - dd * v[2];
    } else {
        const double p[3] = {y[0] + t[0], y[1] + t[1], y[2] + t[2]};
        dd = dot3(p, v) - a[0];  // a[0] = n'.a'
        r[0] = dd * v[0];
        r[1] = dd * v[1];
        r[2] = dd * v[2];
    }
    *dd_out = dd;
    return dot3(r, r);
}

// cost only
LL_HD double block_cost(int kind, const double R[9], const double t[3], const double f[3], const double a[3],
                        const double v[3], double huber_a)
{
    double y[3], r[3], dd, rho0, rho1;
    const double s = block_residual(kind, R, t, f, a, v, y, r, &dd);
    huber(huber_a, s, &rho0, &rho1);
    return 0.5 * rho0;
}

// cost + gradient + Gauss-Newton matrix of one block, accumulated into acc[28]
LL_HD void block_accumulate(int kind, const double R[9], const double t[3], const double f[3], const double a[3],
                            const double v[3], double huber_a, double acc[LL_NACC])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    double y[3], r[3], dd, rho0, w;
    const double s = block_residual(kind, R, t, f, a, v, y, r, &dd);
    huber(huber_a, s, &rho0, &w);
    acc[27] += 0.5 * rho0;
    // c(z) = B^T z = [2 y x z ; z]  with B = [-2[y]x | I]
    double cv[6];
    {
        double yxv[3];
        cross3(y, v, yxv);
        cv[0] = 2.0 * yxv[0];
        cv[1] = 2.0 * yxv[1];
        cv[2] = 2.0 * yxv[2];
        cv[3] = v[0];
        cv[4] = v[1];
        cv[5] = v[2];
    }
    if (kind == BLK_PLANE) {
        // J = n (B^T n)^T : J^T J = (n.n) cn cn^T, J^T r = dd (n.n) cn
        const double nn2 = dot3(v, v);
        const double wn = w * nn2;
        const double gs = wn * dd;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc[21 + i] += gs * cv[i];
            const double wi = wn * cv[i];
#pragma unroll
            for (int j = i; j < 6; j++) acc[hidx(i, j)] += wi * cv[j];
        }
    } else {
        // J = A B, A = I - u u^T.  J^T r = B^T (A r);  J^T J = B^T A^T A B = B^T B - kappa cu cu^T, kappa = 2 - u.u
        const double ur = dot3(v, r);
        const double ar[3] = {r[0] - ur * v[0], r[1] - ur * v[1], r[2] - ur * v[2]};
        double yxr[3];
        cross3(y, ar, yxr);
        acc[21] += w * 2.0 * yxr[0];
        acc[22] += w * 2.0 * yxr[1];
        acc[23] += w * 2.0 * yxr[2];
        acc[24] += w * ar[0];
        acc[25] += w * ar[1];
        acc[26] += w * ar[2];
        const double kappa = 2.0 - dot3(v, v);
        const double yy = dot3(y, y);
        // B^T B: [4(|y|^2 I - y y^T), 2[y]x ; . , I]
        double BtB[21];
        BtB[hidx(0, 0)] = 4.0 * (yy - y[0] * y[0]);
        BtB[hidx(0, 1)] = -4.0 * y[0] * y[1];
        BtB[hidx(0, 2)] = -4.0 * y[0] * y[2];
        BtB[hidx(1, 1)] = 4.0 * (yy - y[1] * y[1]);
        BtB[hidx(1, 2)] = -4.0 * y[1] * y[2];
        BtB[hidx(2, 2)] = 4.0 * (yy - y[2] * y[2]);
        BtB[hidx(0, 3)] = 0.0;
        BtB[hidx(0, 4)] = -2.0 * y[2];
        BtB[hidx(0, 5)] = 2.0 * y[1];
        BtB[hidx(1, 3)] = 2.0 * y[2];
        BtB[hidx(1, 4)] = 0.0;
        BtB[hidx(1, 5)] = -2.0 * y[0];
        BtB[hidx(2, 3)] = -2.0 * y[1];
        BtB[hidx(2, 4)] = 2.0 * y[0];
        BtB[hidx(2, 5)] = 0.0;
        BtB[hidx(3, 3)] = 1.0;
        BtB[hidx(3, 4)] = 0.0;
        BtB[hidx(3, 5)] = 0.0;
        BtB[hidx(4, 4)] = 1.0;
        BtB[hidx(4, 5)] = 0.0;
        BtB[hidx(5, 5)] = 1.0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double ki = kappa * cv[i];
#pragma unroll
            for (int j = i; j < 6; j++) acc[hidx(i, j)] += w * (BtB[hidx(i, j)] - ki * cv[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------- scaled plane blocks (round 6)
// A plane block's residual is r = dd n' with dd = n'.p - c and n' NOT normalised (ceres_icp.hpp:328-366), its Jacobian J = n' (B^T n')^T.
// With m = |n'| n' and beta = |n'| c the block is the SCALAR residual e = m.p - beta (= |n'| dd, so e^2 = |r|^2) with Jacobian row
// (B^T m)^T:  J^T J = (B^T m)(B^T m)^T,  J^T r = e B^T m,  and r = e m / |m|.  The plane-table solver (solve_fast3) stores {m, beta} per
// distinct neighbour triple, which takes |n'|^2, the vector r and its square out of every block evaluation; the factor 2 of
// B^T m = [2 y x m ; m] is left out of the accumulation and put back once per evaluation (plane_unfold2: exact, powers of two).
// Same numbers as block_accumulate / block_l1 up to rounding (tests/test_hostcheck.py: 1e-13 relative on random blocks).
LL_HD void plane_scale(const double v[3], double a0, double m[3], double *beta)
{
    const double nn = sqrt(dot3(v, v));
    m[0] = v[0] * nn;
    m[1] = v[1] * nn;
    m[2] = v[2] * nn;
    *beta = a0 * nn;
}
// acc += the block's cost / gradient / Gauss-Newton terms, rotation rows and columns WITHOUT their factor 2 (plane_unfold2)
LL_HD void plane_accumulate_scaled(const double R[9], const double t[3], const double f[3], const double m[3], double beta, double huber_a,
                                   double acc[LL_NACC])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    const double p0 = R[0] * f[0] + R[1] * f[1] + R[2] * f[2], p1 = R[3] * f[0] + R[4] * f[1] + R[5] * f[2], p2 = R[6] * f[0] + R[7] * f[1] + R[8] * f[2];
    const double e = m[0] * (p0 + t[0]) + m[1] * (p1 + t[1]) + m[2] * (p2 + t[2]) - beta;
    double rho0, w;
    huber(huber_a, e * e, &rho0, &w);
    acc[27] += 0.5 * rho0;
    const double cv[6] = {p1 * m[2] - p2 * m[1], p2 * m[0] - p0 * m[2], p0 * m[1] - p1 * m[0], m[0], m[1], m[2]};  // [y x m ; m]
    const double we = w * e;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        acc[21 + i] += we * cv[i];
        const double wi = w * cv[i];
#pragma unroll
        for (int j = i; j < 6; j++) acc[hidx(i, j)] += wi * cv[j];
    }
}
// the factors 2 of the rotation part, once per thread and evaluation (before any line block is added): H_rr x 4, H_rt x 2, g_r x 2
LL_HD void plane_unfold2(double acc[LL_NACC])
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        acc[21 + i] *= 2.0;
#pragma unroll
        for (int j = i; j < 6; j++) acc[hidx(i, j)] *= (j < 3 ? 4.0 : 2.0);
    }
}
// loss-corrected L1 norm of the world-frame residual (block_l1) of a scaled plane block
LL_HD double plane_l1_scaled(const double R[9], const double t[3], const double f[3], const double m[3], double beta, double huber_a,
                             const double q_last[4])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    const double p0 = R[0] * f[0] + R[1] * f[1] + R[2] * f[2], p1 = R[3] * f[0] + R[4] * f[1] + R[5] * f[2], p2 = R[6] * f[0] + R[7] * f[1] + R[8] * f[2];
    const double e = m[0] * (p0 + t[0]) + m[1] * (p1 + t[1]) + m[2] * (p2 + t[2]) - beta;
    double rho0, w;
    huber(huber_a, e * e, &rho0, &w);
    const double mm = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    const double k = mm > 0.0 ? e / sqrt(mm) : 0.0;  // r = e m / |m|
    const double r[3] = {k * m[0], k * m[1], k * m[2]};
    double rw[3];
    quat_rot(q_last, r, rw);
    const double sq = sqrt(w);
    return fabs(sq * rw[0]) + fabs(sq * rw[1]) + fabs(sq * rw[2]);
}

// ---------------------------------------------------------------------------------------------- motion-deblur blocks
// ceres_icp_point2line_mb / point2plane_mb (ceres_icp.hpp:81-233):  p = q_last (slerp(I, q_inc, s) f + s t_inc) + t_last.
// With w = Log(R_inc) (rotation vector, |w| = W, axis n, K = [n]x) the interpolated rotation is R_s = Exp(s w) and,
// for the Ceres perturbation R_inc+ = Exp(2 d) R_inc,
//     d(R_s f)/dd = -[R_s f]x M,   M = 2 s J_l(s w) J_l(w)^-1 = m0 (I + beta K + gamma K^2)
// (left Jacobians of SO(3); both are polynomials in K, K^3 = -K).  d p/d t_inc = s I.
// cross3 / dot3 with the multiply-adds fused (the motion-deblur block functions below: round 6; the plain forms above keep the
// un-contracted arithmetic the block constants were pinned with)
LL_HD void cross3c(const double a[3], const double b[3], double o[3])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
LL_HD double dot3c(const double a[3], const double b[3])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
struct MbRot {
    double n[3];       // rotation axis of the increment
    double W;          // rotation angle of the increment
    double c_half, d;  // J_l(w)^-1 = I + c_half K + d K^2 : c_half = -W/2, d = 1 - (W/2) cot(W/2)
};

LL_HD void mb_prepare(const double q_in[4], MbRot &m)
{
    double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
    if (q[3] < 0.0) {  // slerp takes the shorter arc (Eigen: scale1 = -scale1 when the dot product is negative)
        q[0] = -q[0];
        q[1] = -q[1];
        q[2] = -q[2];
        q[3] = -q[3];
    }
    const double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv > 0.0) {
        m.n[0] = q[0] / nv;
        m.n[1] = q[1] / nv;
        m.n[2] = q[2] / nv;
        m.W = 2.0 * atan2(nv, q[3]);
    } else {
        m.n[0] = 1.0;
        m.n[1] = 0.0;
        m.n[2] = 0.0;
        m.W = 0.0;
    }
    m.c_half = -0.5 * m.W;
    if (m.W > 1e-4) {
        const double h = 0.5 * m.W;
        m.d = 1.0 - h * cos(h) / sin(h);
    } else {
        m.d = m.W * m.W / 12.0;
    }
}

// sin and cos of a small angle by their Taylor polynomials (|x| < 0.25: the first neglected terms, x^19/19! and
// x^18/18!, are below 1e-27 -- far under one ulp).  The interpolated rotation angle s W of a scan-to-scan increment is a
// fraction of a degree, and the library sincos with its full range reduction was a third of the instructions of a
// motion-deblur block evaluation.
LL_HD void sincos_small(double x, double *sn, double *cs)
{
    const double z = x * x;
    double p = -1.0 / 1307674368000.0;            // -1/15!
    p = p * z + 1.0 / 6227020800.0;               //  1/13!
    p = p * z - 1.0 / 39916800.0;                 // -1/11!
    p = p * z + 1.0 / 362880.0;                   //  1/9!
    p = p * z - 1.0 / 5040.0;                     // -1/7!
    p = p * z + 1.0 / 120.0;                      //  1/5!
    p = p * z - 1.0 / 6.0;                        // -1/3!
    *sn = x + x * (z * p);
    double q = 1.0 / 20922789888000.0;            //  1/16!
    q = q * z - 1.0 / 87178291200.0;              // -1/14!
    q = q * z + 1.0 / 479001600.0;                //  1/12!
    q = q * z - 1.0 / 3628800.0;                  // -1/10!
    q = q * z + 1.0 / 40320.0;                    //  1/8!
    q = q * z - 1.0 / 720.0;                      // -1/6!
    q = q * z + 1.0 / 24.0;                       //  1/4!
    q = q * z - 0.5;                              // -1/2!
    *cs = 1.0 + z * q;
}

// y = R_s f and the coefficients (m0, beta, gamma) of M for blur ratio s
// For a small angle x (|x| < 0.25) everything mb_block needs of it comes from two even polynomials, no division:
//   B(z) = (1 - sin x / x) / z = 1/3! - z/5! + z^2/7! - ...      C(z) = (1 - cos x) / z = 1/2! - z/4! + z^2/6! - ...      (z = x^2)
//   sin x = x - x z B,   1 - cos x = z C,   a = (1 - cos x) / x = x C,   b = 1 - sin x / x = z B
// (first neglected terms z^7/17! and z^8/18!: below 1e-24 at |x| = 0.25.)  Round 6: the two divisions by s W of the general form below
// cost as much as both Taylor polynomials of sincos_small together, per residual block and cost evaluation.
LL_HD void mb_small_angle(double x, double *sn, double *omc, double *a, double *b)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    const double z = x * x;
    double pb = -1.0 / 1307674368000.0;  // -1/15!
    pb = pb * z + 1.0 / 6227020800.0;    //  1/13!
    pb = pb * z - 1.0 / 39916800.0;      // -1/11!
    pb = pb * z + 1.0 / 362880.0;        //  1/9!
    pb = pb * z - 1.0 / 5040.0;          // -1/7!
    pb = pb * z + 1.0 / 120.0;           //  1/5!
    pb = pb * z - 1.0 / 6.0;             // -1/3!   (pb = -B)
    double pc = 1.0 / 20922789888000.0;  //  1/16!
    pc = pc * z - 1.0 / 87178291200.0;   // -1/14!
    pc = pc * z + 1.0 / 479001600.0;     //  1/12!
    pc = pc * z - 1.0 / 3628800.0;       // -1/10!
    pc = pc * z + 1.0 / 40320.0;         //  1/8!
    pc = pc * z - 1.0 / 720.0;           // -1/6!
    pc = pc * z + 1.0 / 24.0;            //  1/4!
    pc = pc * z - 0.5;                   // -1/2!   (pc = -C)
    const double zb = z * pb;            // -(1 - sin x / x)
    *sn = x + x * zb;
    *omc = -(z * pc);
    *a = -(x * pc);
    *b = -zb;
}

LL_HD void mb_block(const MbRot &m, double s, const double f[3], double y[3], double coef[3])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    const double sW = s * m.W;
    double sn, omc, a, b;  // sin, 1 - cos, and J_l(s w) = I + a K + b K^2
    if (fabs(sW) < 0.25) {
        mb_small_angle(sW, &sn, &omc, &a, &b);
    } else {
        sn = sin(sW);
        omc = 1.0 - cos(sW);
        a = omc / sW;
        b = 1.0 - sn / sW;
    }
    double nf[3], nnf[3];
    cross3c(m.n, f, nf);
    cross3c(m.n, nf, nnf);
    y[0] = f[0] + sn * nf[0] + omc * nnf[0];
    y[1] = f[1] + sn * nf[1] + omc * nnf[1];
    y[2] = f[2] + sn * nf[2] + omc * nnf[2];
    const double c = m.c_half, d = m.d;
    coef[0] = 2.0 * s;
    coef[1] = a + c - a * d - b * c;
    coef[2] = b + d + a * c - b * d;
}

// M^T z = m0 (z - beta K z + gamma K^2 z)
LL_HD void mb_Mt(const MbRot &m, const double coef[3], const double z[3], double o[3])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    double kz[3], kkz[3];
    cross3c(m.n, z, kz);
    cross3c(m.n, kz, kkz);
    o[0] = coef[0] * (z[0] - coef[1] * kz[0] + coef[2] * kkz[0]);
    o[1] = coef[0] * (z[1] - coef[1] * kz[1] + coef[2] * kkz[1]);
    o[2] = coef[0] * (z[2] - coef[1] * kz[2] + coef[2] * kkz[2]);
}

// residual of a deblur block (frame of pose_last): returns |r|^2
LL_HD double block_residual_mb(int kind, const MbRot &m, const double t[3], double s, const double f[3], const double a[3],
                               const double v[3], double y[3], double coef[3], double r[3], double *dd_out)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    mb_block(m, s, f, y, coef);
    double dd;
    if (kind == BLK_LINE) {
        const double d[3] = {y[0] + s * t[0] - a[0], y[1] + s * t[1] - a[1], y[2] + s * t[2] - a[2]};
        dd = dot3c(d, v);
        r[0] = d[0] - dd * v[0];
        r[1] = d[1] - dd * v[1];
        r[2] = d[2] - dd * v[2];
    } else {
        const double p[3] = {y[0] + s * t[0], y[1] + s * t[1], y[2] + s * t[2]};
        dd = dot3c(p, v) - a[0];  // a[0] = n'.a'
        r[0] = dd * v[0];
        r[1] = dd * v[1];
        r[2] = dd * v[2];
    }
    *dd_out = dd;
    return dot3c(r, r);
}

LL_HD void block_accumulate_mb(int kind, const MbRot &m, const double t[3], double s, const double f[3], const double a[3],
                               const double v[3], double huber_a, double acc[LL_NACC])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    double y[3], coef[3], r[3], dd, rho0, w;
    const double ss = block_residual_mb(kind, m, t, s, f, a, v, y, coef, r, &dd);
    huber(huber_a, ss, &rho0, &w);
    acc[27] += 0.5 * rho0;
    if (kind == BLK_PLANE) {
        // J = n (B^T n)^T with B^T n = [M^T (y x n) ; s n]
        double yxn[3], top[3];
        cross3c(y, v, yxn);
        mb_Mt(m, coef, yxn, top);
        const double cv[6] = {top[0], top[1], top[2], s * v[0], s * v[1], s * v[2]};
        const double nn2 = dot3c(v, v);
        const double wn = w * nn2, gs = wn * dd;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc[21 + i] += gs * cv[i];
            const double wi = wn * cv[i];
#pragma unroll
            for (int j = i; j < 6; j++) acc[hidx(i, j)] += wi * cv[j];
        }
    } else {
        // explicit J = A B (3x6): column j of B is (M e_j) x y for the rotation part, s e_j for the translation part
        double J[3][6];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double ej[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
            double kz[3], kkz[3], me[3], bj[3];
            cross3c(m.n, ej, kz);
            cross3c(m.n, kz, kkz);
            for (int i = 0; i < 3; i++) me[i] = coef[0] * (ej[i] + coef[1] * kz[i] + coef[2] * kkz[i]);  // M e_j
            cross3c(me, y, bj);                                                                        // -[y]x M e_j
            const double ub = dot3c(v, bj);
            for (int i = 0; i < 3; i++) J[i][j] = bj[i] - ub * v[i];
            const double us = s * v[j];
            for (int i = 0; i < 3; i++) J[i][3 + j] = (i == j ? s : 0.0) - us * v[i];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            acc[21 + i] += w * (J[0][i] * r[0] + J[1][i] * r[1] + J[2][i] * r[2]);
#pragma unroll
            for (int j = i; j < 6; j++) acc[hidx(i, j)] += w * (J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j]);
        }
    }
}

LL_HD double block_l1_mb(int kind, const MbRot &m, const double t[3], double s, const double f[3], const double a[3],
                         const double v[3], double huber_a, const double q_last[4])
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    double y[3], coef[3], r[3], dd, rho0, w, rw[3];
    const double ss = block_residual_mb(kind, m, t, s, f, a, v, y, coef, r, &dd);
    huber(huber_a, ss, &rho0, &w);
    quat_rot(q_last, r, rw);
    const double sq = sqrt(w);
    return fabs(sq * rw[0]) + fabs(sq * rw[1]) + fabs(sq * rw[2]);
}

// loss-corrected L1 norm of the world-frame residual (problem.Evaluate + point_cloud_registration.hpp:158,489)
LL_HD double block_l1(int kind, const double R[9], const double t[3], const double f[3], const double a[3],
                      const double v[3], double huber_a, const double q_last[4])
{
    double y[3], r[3], dd, rho0, w, rw[3];
    const double s = block_residual(kind, R, t, f, a, v, y, r, &dd);
    huber(huber_a, s, &rho0, &w);
    quat_rot(q_last, r, rw);
    const double sq = sqrt(w);
    return fabs(sq * rw[0]) + fabs(sq * rw[1]) + fabs(sq * rw[2]);
}

// ---------------------------------------------------------------------------------------------- LM controller

// ProgramEvaluator::Plus: EigenQuaternionParameterization::Plus on q, t += d then clamp to +-bound
LL_HD void state_plus(const double x[7], const double d[6], double bound, double out[7])
{
    const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd == 0.0) {
        out[0] = x[0];
        out[1] = x[1];
        out[2] = x[2];
        out[3] = x[3];
    } else {
        double sv, cv;
        if (nd < 0.25) {  // LM steps are a fraction of a degree: Taylor polynomials, exact to the last ulp or two
            sincos_small(nd, &sv, &cv);
        } else {
            sv = sin(nd);
            cv = cos(nd);
        }
        const double sn = sv / nd;
        const double dq[4] = {sn * d[0], sn * d[1], sn * d[2], cv};
        quat_mul(dq, x, out);
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

// Every loop is fully unrolled so that the factor lives in registers on the GPU (with rolled loops the 6x6 arrays are
// indexed dynamically and end up in scratch memory: ~250 dependent memory round trips per LM iteration on the one lane
// that runs the controller).  A failed pivot clears `ok` instead of returning early; the arithmetic of a successful
// factorisation is unchanged.
LL_HD int chol_solve6(const double A[36], const double b[6], double x[6])
{
    // One reciprocal per pivot instead of a division per element: an fp64 division is a ~30-instruction dependent chain
    // on the one lane that runs the controller, and the factorisation had 21 of them (+ 6 square roots).
    double L[36], inv[6];
    int ok = 1;
    LL_UNROLL
    for (int i = 0; i < 6; i++) {
        LL_UNROLL
        for (int j = 0; j < 6; j++) {
            if (j > i) continue;
            double s = A[i * 6 + j];
            LL_UNROLL
            for (int k = 0; k < 6; k++)
                if (k < j) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0.0)) ok = 0;
                L[i * 6 + i] = sqrt(s);
                inv[i] = 1.0 / L[i * 6 + i];
            } else {
                L[i * 6 + j] = s * inv[j];
            }
        }
    }
    double y[6];
    LL_UNROLL
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        LL_UNROLL
        for (int k = 0; k < 6; k++)
            if (k < i) s -= L[i * 6 + k] * y[k];
        y[i] = s * inv[i];
    }
    LL_UNROLL
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        LL_UNROLL
        for (int k = 0; k < 6; k++)
            if (k > i) s -= L[k * 6 + i] * x[k];
        x[i] = s * inv[i];
    }
    LL_UNROLL
    for (int i = 0; i < 6; i++)
        if (!((x[i] - x[i]) == 0.0)) ok = 0;
    return ok;
}

// Trust-region Levenberg-Marquardt step controller with Ceres' default options (see oracle/ll_oracle_reg.c
// header for the list).  The caller evaluates (cost, g, H) at the points the controller asks for.
struct LmCtl {
    // configuration
    int max_iter;
    double bound;
    // current iterate
    double x[7], x_norm, cost, g[6], H[21], gmax;
    double scale[6], diag[6];
    double radius, decrease_factor;
    int reuse_diagonal, invalid_steps, iteration;
    // summary
    double initial_cost, final_cost;
    // pending step
    double delta[6], model_cost_change, gd, dmax;
    double cand[7];
    double first_eval[LL_NACC];  // evaluation at the full step (kept for a failed line search)
    double first_cand[7];
    double ls_step;
    double ls_prev_x, ls_prev_f, ls_prev_g;  // the trial before the current one (`previous` of ArmijoLineSearch::DoSearch)
    int ls_prev_valid;
    int ls_iter, ls_active;
    int done;
    int last_accept;  // set by lm_update: 0 = the step was not accepted, 1 = accepted and x is the point just evaluated,
                      // 2 = accepted but x is the first point of a failed line search (not the one just evaluated)
};

LL_HD double lm_gradient_max_norm(const double x[7], const double g[6], double bound)
{
    double ng[6], xp[7], m = 0.0;
    for (int i = 0; i < 6; i++) ng[i] = -g[i];
    state_plus(x, ng, bound, xp);
    for (int i = 0; i < 7; i++) m = fmax(m, fabs(x[i] - xp[i]));
    return m;
}

LL_HD double lm_cubic_min_step(double f0, double g0, double x1, double f1, double g1, double lo, double hi)
{
    const double x12 = x1 * x1, x13 = x12 * x1;
    const double r0 = f1 - f0 - g0 * x1, r1 = g1 - g0;
    const double det = x13 * 2.0 * x1 - x12 * 3.0 * x12;
    const double a = (r0 * 2.0 * x1 - x12 * r1) / det;
    const double b = (x13 * r1 - 3.0 * x12 * r0) / det;
    double best_x = lo;
    double best_v = ((a * lo + b) * lo + g0) * lo + f0;
    const double vh = ((a * hi + b) * hi + g0) * hi + f0;
    if (vh < best_v) {
        best_v = vh;
        best_x = hi;
    }
    const double A = 3.0 * a, B = 2.0 * b, C = g0;
    double roots[2];
    int nr = 0;
    if (fabs(A) < 1e-300) {
        if (fabs(B) > 1e-300) roots[nr++] = -C / B;
    } else {
        const double disc = B * B - 4.0 * A * C;
        if (disc >= 0) {
            const double sq = sqrt(disc);
            roots[nr++] = (-B + sq) / (2.0 * A);
            roots[nr++] = (-B - sq) / (2.0 * A);
        }
    }
    for (int i = 0; i < nr; i++) {
        if (roots[i] > lo && roots[i] < hi) {
            const double v = ((a * roots[i] + b) * roots[i] + g0) * roots[i] + f0;
            if (v < best_v) {
                best_v = v;
                best_x = roots[i];
            }
        }
    }
    return best_x;
}

// Second and later contractions of a projected line search: Ceres (line_search.cc, CUBIC interpolation with a valid `previous`
// sample) minimises the QUINTIC through the start (0, f0, g0), the current trial (x1, f1, g1) and the previous one (x2, f2, g2)
// over [lo, hi] -- the polynomial of FindInterpolatingPolynomial (polynomial.cc solves a 6 x 6 Vandermonde system; here the same
// interpolant in Newton form, by divided differences: no pivot search, so nothing is indexed dynamically and the whole fit stays in
// registers on the controller lane), then the better end point or a real root of the derivative inside the interval
// (MinimizePolynomial).  Ceres takes the roots of the quartic derivative as the eigenvalues of its companion matrix (Eigen, third
// party) and so sees every real root.  So does this restatement (round 6; rounds 2 - 5 bracketed sign changes on a 32-cell grid and
// could miss a pair of roots inside one cell): the roots are ISOLATED EXACTLY by the derivative chain -- the quartic p' is monotone
// between consecutive roots of the cubic p'', which is monotone between the roots of the quadratic p''' (closed form) -- so every
// interval between consecutive break points holds at most one root, found by its sign change and bisected.  A root without a sign
// change (even multiplicity) is an inflection of p, never its minimum; the real parts of complex roots, which Ceres also tries, are
// not stationary and cannot beat the true minimiser.  Operation for operation the same in oracle/ll_oracle_reg.c quintic_min_step and
// in the Ceres stand-in of oracle/_ref (ll_stub_ceres_solver.h quintic_min); tests/test_hostcheck.py holds all three to a dense
// companion-matrix root finder on adversarial fits (two stationary points 1e-4 of the interval apart).
// the interpolant in Newton form: p(x) = f0 + x (e01 + x (a0 + (x - x1) (b0 + (x - x1) (c0 + (x - x2) d0))))
struct Quintic {
    double x1, x2, f0, e01, a0, b0, c0, d0;
};
// false: coincident samples (the caller bisects like an invalid sample)
LL_HD bool quintic_fit(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, Quintic &q)
{
    /* Newton form on the nodes z = {0, 0, x1, x1, x2} (the sixth, x2 again, closes the table): divided differences with the
     * derivative in place of the quotient at a repeated node */
    const double h1 = x1, h2 = x2, h21 = x2 - x1;
    if (!(h1 != 0.0) || !(h2 != 0.0) || !(h21 != 0.0)) return false;
    const double e01 = g0, e12 = (f1 - f0) / h1, e23 = g1, e34 = (f2 - f1) / h21, e45 = g2;
    const double a0 = (e12 - e01) / h1, a1 = (e23 - e12) / h1, a2 = (e34 - e23) / h21, a3 = (e45 - e34) / h21;
    const double b0 = (a1 - a0) / h1, b1 = (a2 - a1) / h2, b2 = (a3 - a2) / h21;
    const double c0 = (b1 - b0) / h2, c1 = (b2 - b1) / h2;
    q.x1 = x1;
    q.x2 = x2;
    q.f0 = f0;
    q.e01 = e01;
    q.a0 = a0;
    q.b0 = b0;
    q.c0 = c0;
    q.d0 = (c1 - c0) / h2;
    return true;
}
/* value and derivative by one nested sweep of explicitly fused multiply-adds (IEEE: the same bits in the oracle, the stand-in and
 * on the device; half the dependent chain of a multiply and an add per step, and the sweep runs ~100 times per fit) */
LL_HD void quintic_eval(const Quintic &q, double x, double &pv, double &dv)
{
    const double u1 = x - q.x1, u2 = x - q.x2;
    double b = q.d0, db = 0.0;
    db = fma(u2, db, b);
    b = fma(u2, b, q.c0);
    db = fma(u1, db, b);
    b = fma(u1, b, q.b0);
    db = fma(u1, db, b);
    b = fma(u1, b, q.a0);
    db = fma(x, db, b);
    b = fma(x, b, q.e01);
    db = fma(x, db, b);
    b = fma(x, b, q.f0);
    pv = b;
    dv = db;
}
// monomial coefficients of the derivative chain: p' = dq[0] + dq[1] x + .. + dq[4] x^4, p'' = d2[0] + .. + d2[3] x^3 (d2[4] = 0),
// p''' = A x^2 + B x + C.  From the Newton form by three synthetic multiplications with (x - node), every step one fused multiply-add.
struct QuinticChain {
    double dq[5], d2[5], A, B, C;
};
LL_HD void quintic_chain(const Quintic &q, QuinticChain &c)
{
    const double s1 = q.d0, s0 = fma(-q.x2, q.d0, q.c0);                              // c0 + (x - x2) d0
    const double t2 = s1, t1 = fma(-q.x1, s1, s0), t0 = fma(-q.x1, s0, q.b0);         // b0 + (x - x1) (..)
    const double u3 = t2, u2 = fma(-q.x1, t2, t1), u1 = fma(-q.x1, t1, t0), u0 = fma(-q.x1, t0, q.a0);  // a0 + (x - x1) (..)
    // p(x) = u3 x^5 + u2 x^4 + u1 x^3 + u0 x^2 + e01 x + f0
    c.dq[4] = 5.0 * u3, c.dq[3] = 4.0 * u2, c.dq[2] = 3.0 * u1, c.dq[1] = 2.0 * u0, c.dq[0] = q.e01;
    c.d2[4] = 0.0, c.d2[3] = 20.0 * u3, c.d2[2] = 12.0 * u2, c.d2[1] = 6.0 * u1, c.d2[0] = 2.0 * u0;
    c.A = 60.0 * u3, c.B = 24.0 * u2, c.C = 6.0 * u1;
}
LL_HD double quintic_poly4(const double k[5], double x) { return fma(fma(fma(fma(k[4], x, k[3]), x, k[2]), x, k[1]), x, k[0]); }
// real roots of A x^2 + B x + C strictly inside (lo, hi), ascending; returns how many (0 .. 2)
LL_HD int quintic_quadratic_roots(double A, double B, double C, double lo, double hi, double r[2])
{
    double c0 = 0.0, c1 = 0.0;
    int n = 0;
    if (A == 0.0) {
        if (B != 0.0) c0 = -C / B, n = 1;
    } else {
        const double disc = fma(B, B, -4.0 * A * C);
        if (disc >= 0.0) {
            const double sq = sqrt(disc);
            const double qq = -0.5 * (B + (B < 0.0 ? -sq : sq));  // the numerically stable pair: qq / A and C / qq
            c0 = qq / A;
            n = 1;
            if (qq != 0.0) {
                c1 = C / qq;
                n = 2;
                if (c1 < c0) {
                    const double t = c0;
                    c0 = c1;
                    c1 = t;
                }
            }
        }
    }
    int m = 0;
    if (n >= 1 && c0 > lo && c0 < hi) r[m++] = c0;
    if (n >= 2 && c1 > lo && c1 < hi && !(m == 1 && c1 == r[0])) r[m++] = c1;
    return m;
}
#define LL_QUINTIC_ROOT_STEPS 64 /* bound on the steps of one root refinement (it ends on its own after ~10) */
// the root of the polynomial k in (a, b], where it is monotone: found by the sign change of its end values va, vb (or vb == 0) and refined
// by the ILLINOIS form of regula falsi -- the secant through the bracket's ends, the retained end's value halved whenever the same end is
// replaced twice in a row, a bisection step whenever rounding puts the secant point on an end -- which keeps the root bracketed like
// bisection and converges superlinearly: ~10 polynomial evaluations instead of 60 (the fit sits on the critical path of every scan
// whose step runs into the bounds: with 60-step bisections the small-scan solver lost a third of its speed).  Ends when the iterate
// stops moving, hits a zero, or the bracket cannot shrink.  Returns false when there is no root.
LL_HD bool quintic_interval_root(const double k[5], double a, double b, double va, double vb, double *root)
{
    if (vb == 0.0) {
        *root = b;
        return true;
    }
    if (!((va < 0.0 && vb > 0.0) || (va > 0.0 && vb < 0.0))) return false;
    double l = a, r = b, wl = va, wr = vb;  // bracket and the (possibly halved) values the secant uses; sign(wl) = sign of k at l, likewise r
    const bool neg_left = va < 0.0;
    double x = b;
    int side = 0;
    for (int it = 0; it < LL_QUINTIC_ROOT_STEPS; it++) {
        double c = (wl * r - wr * l) / (wl - wr);
        if (!(c > l && c < r)) c = 0.5 * (l + r);
        if (c == l || c == r || c == x) {  // adjacent doubles, or no progress: done
            x = c;
            break;
        }
        x = c;
        const double vc = quintic_poly4(k, c);
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
    return true;
}
// roots of k inside (lo, hi] given its break points bp[0 .. nb) (ascending, strictly inside (lo, hi)): at most nb + 1, ascending
LL_HD int quintic_roots_between(const double k[5], double lo, double hi, const double *bp, int nb, double *roots)
{
    int n = 0;
    double a = lo, va = quintic_poly4(k, lo);
    for (int i = 0; i <= nb; i++) {
        const double b = (i == nb) ? hi : bp[i];
        const double vb = quintic_poly4(k, b);
        double r;
        if (quintic_interval_root(k, a, b, va, vb, &r)) roots[n++] = r;
        a = b;
        va = vb;
    }
    return n;
}
// MinimizePolynomial's choice among the stationary points: the better end point unless a root's value is strictly smaller, roots in
// ascending order
LL_HD double quintic_pick(const Quintic &q, double lo, double hi, const double *roots, int n)
{
    double best_x = lo, best_v, vh, dl, dh;
    quintic_eval(q, lo, best_v, dl);
    quintic_eval(q, hi, vh, dh);
    (void)dl;
    (void)dh;
    if (!(best_v < vh)) { /* MinimizePolynomial: x_min wins only when strictly smaller */
        best_v = vh;
        best_x = hi;
    }
    for (int i = 0; i < n; i++) {
        double v, dv;
        quintic_eval(q, roots[i], v, dv);
        (void)dv;
        if (v < best_v) {
            best_v = v;
            best_x = roots[i];
        }
    }
    return best_x;
}

LL_HD_NOINLINE double lm_quintic_min_step(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo,
                                          double hi)
{
    Quintic q;
    if (!quintic_fit(f0, g0, x1, f1, g1, x2, f2, g2, q)) return fmin(fmax(0.5 * x1, lo), hi); /* coincident samples: bisect like an invalid sample */
    QuinticChain c;
    quintic_chain(q, c);
    double r3[2], r2[3], r1[4];
    const int n3 = quintic_quadratic_roots(c.A, c.B, c.C, lo, hi, r3);   // p''' = 0: where p'' turns
    int n2 = quintic_roots_between(c.d2, lo, hi, r3, n3, r2);            // p''  = 0: where p' turns
    if (n2 > 0 && !(r2[n2 - 1] < hi)) n2--;                              // (a break point lies strictly inside; hi closes the last interval anyway)
    const int n1 = quintic_roots_between(c.dq, lo, hi, r2, n2, r1);      // p'   = 0: the stationary points
    return quintic_pick(q, lo, hi, r1, n1);
}

// The controller runs on one lane while the workgroup waits.  As out-of-line device functions (round 1) each call paid
// the AMDGPU calling convention -- lm_propose alone saved / restored ~120 VGPRs through scratch memory, 250 scratch
// instructions on the one lane everybody waits for -- so they are inlined into the solver loop by default
// (-DLL_LM_NOINLINE restores the calls for an A/B measurement).
#if defined(LL_LM_NOINLINE)
#define LL_LM_FN LL_HD_NOINLINE
#else
#define LL_LM_FN LL_HD
#endif

// Start a solve at x0: projects onto the bounds; the caller must evaluate at c.x and call lm_init.
LL_LM_FN void lm_begin(LmCtl &c, const double x0[7], int max_iter, double bound)
{
    const double zero6[6] = {0, 0, 0, 0, 0, 0};
    c.max_iter = max_iter;
    c.bound = bound;
    state_plus(x0, zero6, bound, c.x);
    double n = 0;
    for (int i = 0; i < 7; i++) n += c.x[i] * c.x[i];
    c.x_norm = sqrt(n);
    c.done = 0;
}

// Decide the next trial step from the current iterate.  Returns 1 if c.cand must be evaluated, 0 if finished.
LL_LM_FN int lm_propose(LmCtl &c)
{
    for (;;) {
        if (c.iteration >= c.max_iter || c.gmax <= 1e-10 || c.radius < 1e-32) {
            c.done = 1;
            return 0;
        }
        c.iteration++;
        double Hs[36], gs[6], A[36], y[6], step[6];
        LL_UNROLL
        for (int a = 0; a < 6; a++) {
            gs[a] = c.g[a] * c.scale[a];
            LL_UNROLL
            for (int b = 0; b < 6; b++) {
                const double h = (a <= b) ? c.H[hidx(a, b)] : c.H[hidx(b, a)];
                Hs[a * 6 + b] = h * c.scale[a] * c.scale[b];
            }
        }
        if (!c.reuse_diagonal) {
            LL_UNROLL
            for (int j = 0; j < 6; j++) c.diag[j] = fmin(fmax(Hs[j * 6 + j], 1e-6), 1e32);
        }
        LL_UNROLL
        for (int i = 0; i < 36; i++) A[i] = Hs[i];
        LL_UNROLL
        for (int j = 0; j < 6; j++) A[j * 6 + j] += c.diag[j] / c.radius;
        const int ok = chol_solve6(A, gs, y);
        c.reuse_diagonal = 1;
        double mcc = 0.0;
        if (ok) {
            double sg = 0.0, sHs = 0.0;
            LL_UNROLL
            for (int a = 0; a < 6; a++) step[a] = -y[a];
            LL_UNROLL
            for (int a = 0; a < 6; a++) {
                sg += step[a] * gs[a];
                double t = 0.0;
                LL_UNROLL
                for (int b = 0; b < 6; b++) t += Hs[a * 6 + b] * step[b];
                sHs += step[a] * t;
            }
            mcc = -sg - 0.5 * sHs;
        }
        if (!ok || !(mcc > 0.0)) {
            if (++c.invalid_steps >= 5) {
                c.done = 1;
                return 0;
            }
            c.radius *= 0.5;
            c.reuse_diagonal = 1;
            continue;
        }
        c.invalid_steps = 0;
        c.model_cost_change = mcc;
        c.gd = 0.0;
        c.dmax = 0.0;
        LL_UNROLL
        for (int j = 0; j < 6; j++) {
            c.delta[j] = step[j] * c.scale[j];
            c.gd += c.g[j] * c.delta[j];
            c.dmax = fmax(c.dmax, fabs(c.delta[j]));
        }
        state_plus(c.x, c.delta, c.bound, c.cand);
        c.ls_step = 1.0;
        c.ls_iter = 0;
        c.ls_active = 0;
        c.ls_prev_valid = 0;
        return 1;
    }
}

// First evaluation (at c.x after lm_begin).  e = acc[28].  Returns like lm_propose.
LL_LM_FN int lm_init(LmCtl &c, const double e[LL_NACC], int n_active)
{
    c.cost = e[27];
    for (int i = 0; i < 6; i++) c.g[i] = e[21 + i];
    for (int i = 0; i < 21; i++) c.H[i] = e[i];
    for (int j = 0; j < 6; j++) c.scale[j] = 1.0 / (1.0 + sqrt(c.H[hidx(j, j)]));
    c.gmax = lm_gradient_max_norm(c.x, c.g, c.bound);
    c.initial_cost = c.cost;
    c.final_cost = c.cost;
    c.iteration = 0;
    c.radius = 1e4;
    c.decrease_factor = 2.0;
    c.reuse_diagonal = 0;
    c.invalid_steps = 0;
    if (n_active == 0) {
        c.done = 1;
        return 0;
    }
    return lm_propose(c);
}

// ---- lm_update, in the pieces the device needs: the three-sample fit of a line search (lm_quintic_min_step) is the one part of
// the controller with parallel work in it, and the solver runs it on the controller's whole wavefront (ll_reg_kernels.hip
// lm_quintic_min_step_wave) between lm_update_pre and lm_update_post.  lm_update below is the same code in one call.

// the part of lm_update behind the line search: e is the evaluation that stands (at c.cand)
LL_LM_FN int lm_update_finish(LmCtl &c, const double *e, int from_first_eval)
{
    double cand_cost = e[27];
    if (!((cand_cost - cand_cost) == 0.0)) cand_cost = DBL_MAX;

    double step_norm = 0.0;
    for (int i = 0; i < 7; i++) step_norm += (c.x[i] - c.cand[i]) * (c.x[i] - c.cand[i]);
    step_norm = sqrt(step_norm);
    if (step_norm <= 1e-8 * (c.x_norm + 1e-8)) {  // ParameterToleranceReached
        c.done = 1;
        return 0;
    }
    const double cost_change = c.cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * c.cost) {  // FunctionToleranceReached
        c.done = 1;
        return 0;
    }
    const double relative_decrease = cost_change / c.model_cost_change;
    if (relative_decrease > 1e-3) {
        c.last_accept = from_first_eval ? 2 : 1;
        double n = 0;
        for (int i = 0; i < 7; i++) {
            c.x[i] = c.cand[i];
            n += c.x[i] * c.x[i];
        }
        c.x_norm = sqrt(n);
        c.cost = cand_cost;
        for (int i = 0; i < 6; i++) c.g[i] = e[21 + i];
        for (int i = 0; i < 21; i++) c.H[i] = e[i];
        c.gmax = lm_gradient_max_norm(c.x, c.g, c.bound);
        if (c.cost < c.final_cost) c.final_cost = c.cost;
        const double t = 2.0 * relative_decrease - 1.0;
        c.radius = c.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        c.radius = fmin(1e16, c.radius);
        c.decrease_factor = 2.0;
        c.reuse_diagonal = 0;
    } else {
        c.radius = c.radius / c.decrease_factor;
        c.decrease_factor *= 2.0;
        c.reuse_diagonal = 1;
    }
    return lm_propose(c);
}

// lm_update_pre / _post return one of these; LM_FINISH_* -> lm_update_finish is still to be called (once, by the caller: the
// controller is inlined into the solver loop, and two copies of lm_propose's factorisation cost the loop registers and code)
#define LM_DONE 0         /* (only from lm_update_finish) */
#define LM_EVAL 1         /* evaluate at c.cand and call lm_update again */
#define LM_FIT 2          /* evaluate the three-sample fit and call lm_update_post */
#define LM_FINISH_E 3     /* lm_update_finish( c, e_in, 0 ): the evaluation just made stands */
#define LM_FINISH_FIRST 4 /* lm_update_finish( c, c.first_eval, c.ls_iter > 1 ): a failed search keeps the full step */

// a contraction of the line search has its new step (or has failed): take it, or fall back to the full step (Ceres keeps it)
LL_LM_FN int lm_update_contract(LmCtl &c, int failed, double new_step, int finite_cost, double cur_cost, double cg)
{
    if (!failed) {
        c.ls_prev_valid = finite_cost ? 1 : 0;
        c.ls_prev_x = c.ls_step;
        c.ls_prev_f = cur_cost;
        c.ls_prev_g = cg;
        c.ls_step = new_step;
        double sd[6];
        for (int j = 0; j < 6; j++) sd[j] = c.delta[j] * c.ls_step;
        state_plus(c.x, sd, c.bound, c.cand);
        return LM_EVAL;
    }
    for (int i = 0; i < 7; i++) c.cand[i] = c.first_cand[i];
    return LM_FINISH_FIRST;
}

// Evaluation e at c.cand is available: the line search's part of lm_update.  LM_FIT: the caller has to evaluate
// lm_quintic_min_step( c.cost, c.gd, c.ls_step, *cur_cost, *cg, c.ls_prev_x, c.ls_prev_f, c.ls_prev_g, 1e-3 * c.ls_step, 0.6 * c.ls_step )
// and hand the step to lm_update_post.
// e_in: the caller's copy of the evaluation (registers on the device: one batch of loads from the LDS sums, which may alias c as
// far as the compiler can tell); lm_update_close gets the same array.
LL_LM_FN int lm_update_pre(LmCtl &c, const double e_in[LL_NACC], double *cur_cost_out, double *cg_out)
{
    c.last_accept = 0;
    if (c.bound >= 0) {
        // projected ARMIJO line search (TrustRegionMinimizer::DoLineSearch)
        if (!c.ls_active) {
            for (int i = 0; i < LL_NACC; i++) c.first_eval[i] = e_in[i];
            for (int i = 0; i < 7; i++) c.first_cand[i] = c.cand[i];
            c.ls_active = 1;
        }
        const double cur_cost = e_in[27];
        const bool finite_cost = (cur_cost - cur_cost) == 0.0;
        if (!finite_cost || cur_cost > c.cost + 1e-4 * c.gd * c.ls_step) {
            int failed = 0;
            double new_step = 0.0, cg = 0.0;
            if (++c.ls_iter >= 20) {
                failed = 1;
            } else {
                if (!finite_cost) {
                    new_step = fmin(fmax(c.ls_step * 0.5, 1e-3 * c.ls_step), 0.6 * c.ls_step);
                } else {
                    for (int j = 0; j < 6; j++) cg += e_in[21 + j] * c.delta[j];
                    if (c.ls_prev_valid) {  // three samples: the quintic (second and later contractions)
                        *cur_cost_out = cur_cost;
                        *cg_out = cg;
                        return LM_FIT;
                    }
                    new_step = lm_cubic_min_step(c.cost, c.gd, c.ls_step, cur_cost, cg, 1e-3 * c.ls_step, 0.6 * c.ls_step);
                }
                if (new_step * c.dmax < 1e-9) failed = 1;
            }
            return lm_update_contract(c, failed, new_step, finite_cost ? 1 : 0, cur_cost, cg);
        } else if (c.ls_step != 1.0) {
            for (int j = 0; j < 6; j++) c.delta[j] *= c.ls_step;
        }
    }
    return LM_FINISH_E;
}

LL_LM_FN int lm_update_post(LmCtl &c, double cur_cost, double cg, double new_step)
{
    return lm_update_contract(c, (new_step * c.dmax < 1e-9) ? 1 : 0, new_step, 1, cur_cost, cg);
}

// what is left after lm_update_pre / _post answered `code` (e: the array lm_update_pre had; overwritten when the full step comes back)
LL_LM_FN int lm_update_close(LmCtl &c, double e[LL_NACC], int code)
{
    if (code == LM_EVAL) return 1;
    if (code == LM_FINISH_FIRST)
        for (int i = 0; i < LL_NACC; i++) e[i] = c.first_eval[i];
    // (from_first_eval: the point just evaluated is the first one only when no retry ran)
    return lm_update_finish(c, e, (code == LM_FINISH_FIRST && c.ls_iter > 1) ? 1 : 0);
}

// Evaluation e at c.cand is available.  Returns 1 if another evaluation (at the new c.cand) is needed.
LL_LM_FN int lm_update(LmCtl &c, const double e_p[LL_NACC])
{
    double e_in[LL_NACC];
    for (int i = 0; i < LL_NACC; i++) e_in[i] = e_p[i];
    double cur_cost = 0.0, cg = 0.0;
    int code = lm_update_pre(c, e_in, &cur_cost, &cg);
    if (code == LM_FIT)
        code = lm_update_post(c, cur_cost, cg,
                              lm_quintic_min_step(c.cost, c.gd, c.ls_step, cur_cost, cg, c.ls_prev_x, c.ls_prev_f, c.ls_prev_g, 1e-3 * c.ls_step,
                                                  0.6 * c.ls_step));
    return lm_update_close(c, e_in, code);
}

}  // namespace ll
