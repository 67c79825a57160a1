// ll_cellmap_core.h -- per-point / per-cell arithmetic of the cell map that feeds the match buffer in the reference's
// "cube" matching mode (m_matching_mode == 1): Points_cloud_map<float> of hku-mars/loam_livox,
// source/cell_map_keyframe.hpp:477-790, as used by Laser_mapping::update_buff_for_matching,
// source/laser_mapping.hpp:471-513, and Laser_mapping::if_pt_in_fov, :310-324.  Shared by the HIP kernels
// (ll_cellmap_kernels.hip) and the test-only host build (tests/hostcheck).  Same float / double operations in the
// same order as the reference (the library is compiled with -ffp-contract=off).
#pragma once
#include <math.h>
#include <stdint.h>

#include "ll_fe_core.h"  // LL_HD, ll_isfinite

namespace ll {

#define LL_CELL_KEY_NONE 0xffffffffffffffffull
#define LL_CELL_K_LIMIT 1048576  // |cell index| < 2^20 per axis: 21 bits each in the packed key

struct CellGeom {
    float box;   // edge of a cell: set_resolution(r) stores m_resolution = r * 0.5 and find_cell_center uses it as the box size
    float half;  // half_of_box_size
};

// Points_cloud_map::set_resolution (CMK:675-680) followed by the constants of find_cell_center (CMK:559-560)
LL_HD CellGeom cell_geom(float resolution)
{
    CellGeom g;
    const float m_resolution = (float)((double)resolution * 0.5);
    g.box = (float)((double)m_resolution * 1.0);
    g.half = (float)((double)m_resolution * 0.5);
    return g;
}

// find_cell_center (CMK:566-568): round((p - half) / box) per axis, in float, std::round = half away from zero.
// The integer triple stands for the float centre the reference keys its hash map with.  false: non-finite or too far.
LL_HD bool cell_index(float x, float y, float z, const CellGeom &g, int k[3])
{
    const float p[3] = {x, y, z};
    for (int c = 0; c < 3; c++) {
        const float r = roundf((p[c] - g.half) / g.box);
        if (!(fabsf(r) < (float)LL_CELL_K_LIMIT)) return false;  // also rejects NaN
        k[c] = (int)r;
    }
    return true;
}

// packed key: ascending key order = lexicographic (kx, ky, kz); this order is the library's definition of the order in
// which find_cells_in_radius returns cells (the reference's order is that of a PCL octree traversal, not reproducible)
LL_HD unsigned long long cell_pack(const int k[3])
{
    return ((unsigned long long)(unsigned)(k[0] + LL_CELL_K_LIMIT) << 42) | ((unsigned long long)(unsigned)(k[1] + LL_CELL_K_LIMIT) << 21) |
           (unsigned long long)(unsigned)(k[2] + LL_CELL_K_LIMIT);
}
LL_HD void cell_unpack(unsigned long long key, int k[3])
{
    k[0] = (int)((key >> 42) & 0x1fffffu) - LL_CELL_K_LIMIT;
    k[1] = (int)((key >> 21) & 0x1fffffu) - LL_CELL_K_LIMIT;
    k[2] = (int)(key & 0x1fffffu) - LL_CELL_K_LIMIT;
}

// the centre find_cell_center returns for the cell (CMK:566-568)
LL_HD void cell_centre(const int k[3], const CellGeom &g, float c[3])
{
    for (int d = 0; d < 3; d++) c[d] = (float)k[d] * g.box + g.half;
}

// pcl::octree::OctreePointCloudSearch::radiusSearch over the cell centres (CMK:761-777): a centre is returned when its
// float squared distance to the float search point does not exceed radius^2 (double)
LL_HD bool cell_in_radius(const float c[3], const float sp[3], float radius)
{
    const float dx = c[0] - sp[0], dy = c[1] - sp[1], dz = c[2] - sp[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    return !((double)d2 > (double)radius * (double)radius);
}

// Laser_mapping::if_pt_in_fov (LM:310-324) on a cell centre; q = (x, y, z, w), t = translation of the current pose
// maximum_in_fov_angle >= 360: no field-of-view test at all (find_cells_in_radius on its own, CMK:761-788 as service_pub_surround_pts calls it, LM:1172)
LL_HD bool cell_in_fov(const float c[3], const double q[4], const double t[3], double maximum_in_fov_angle)
{
    if (maximum_in_fov_angle >= 360.0) return true;
    const double v[3] = {(double)c[0] - t[0], (double)c[1] - t[1], (double)c[2] - t[2]};
    // Eigen: q.inverse() = conjugate / squaredNorm, then the quaternion-vector product v + w * 2(u x v) + u x 2(u x v)
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double u[3] = {-q[0] / n2, -q[1] / n2, -q[2] / n2}, w = q[3] / n2;
    double uv[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    for (int d = 0; d < 3; d++) uv[d] += uv[d];
    const double r[3] = {v[0] + w * uv[0] + (u[1] * uv[2] - u[2] * uv[1]), v[1] + w * uv[1] + (u[2] * uv[0] - u[0] * uv[2]),
                         v[2] + w * uv[2] + (u[0] * uv[1] - u[1] * uv[0])};
    if (r[0] < 0) return false;
    const double nrm = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    // Eigen_math::vector_angle(pt_affine, (1,0,0), 1) (eigen_math.hpp:25-46): a zero vector is "parallel"
    const float angle = nrm == 0 ? 0.0f : (float)acos(fabs(r[0]) / (nrm * 1.0));
    return (double)angle * 57.3 < maximum_in_fov_angle;
}

// pcl::VoxelGrid applied to the points of ONE cell (LM:488-495): PCL's leaf index is floor(x * inv_leaf) - min_b per
// axis, linearised z-major over the cell's own bounding box, i.e. the leaves of a cell are ordered by the global
// lattice coordinate (iz, iy, ix).  The coordinate relative to a lower bound of the cell keeps the same order in 10
// bits per axis.  cell_leaf_span() must stay below 1024.
LL_HD int cell_leaf_local(float x, int k, const CellGeom &g, float inv_leaf)
{
    const int base = (int)floorf(((float)k * g.box) * inv_leaf) - 1;
    return (int)floorf(x * inv_leaf) - base;
}
LL_HD float cell_leaf_span(const CellGeom &g, float inv_leaf) { return g.box * inv_leaf + 4.0f; }

// ------------------------------------------------------------------------------------------------------------
// Cell statistics: Points_cloud_cell::get_mean / get_covmat / covmat_eig_decompose / determine_feature
// (cell_map_keyframe.hpp:225-237, 280-315, 239-249, 436-473) with COMP_TYPE = float (:41) and the non-incremental
// update (IF_ENABLE_INCREMENTAL_UPDATE_MEAN_COV 0, :30), evaluated on the points the cell holds now
// (determine_feature( if_recompute = 1 ), :439-442).
enum : int { CELL_FEATURE_SPHERE = 0, CELL_FEATURE_LINE = 1, CELL_FEATURE_PLANE = 2 };  // Feature_type, CMK:46-51

struct CellStats {
    int type;        // m_feature_type
    float vec[3];    // m_feature_vector (zero for a sphere)
    float mean[3];   // m_mean
    float cov[6];    // m_cov_mat, upper triangle xx xy xz yy yz zz
    float eval[3];   // m_eigen_val, ascending
};

// Eigen decomposition of a symmetric 3x3 (upper triangle m[6] = xx xy xz yy yz zz) by cyclic Jacobi rotations in
// double: eigenvalues ascending, eigenvectors in the columns of V (row-major), each with its first non-zero component
// positive.  The reference calls Eigen::SelfAdjointEigenSolver<Matrix3f> (CMK:245-248), whose iteration and vector
// signs are not reproduced; this is the same decomposition of the same matrix to better than float precision.
LL_HD void sym3_eigen(const double m[6], double val[3], double V[9])
{
    double a[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
    double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 16; sweep++) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
        if (off <= 1e-34 * diag || off == 0.0) break;
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const double apq = a[p * 3 + q];
            if (apq == 0.0) continue;
            const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
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
            for (int k = 0; k < 3; k++) {
                const double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
                v[k * 3 + p] = c * vkp - sn * vkq;
                v[k * 3 + q] = sn * vkp + c * vkq;
            }
        }
    }
    int o[3] = {0, 1, 2};
    const double d[3] = {a[0], a[4], a[8]};
    int tmp;
    if (d[o[0]] > d[o[1]]) { tmp = o[0]; o[0] = o[1]; o[1] = tmp; }
    if (d[o[1]] > d[o[2]]) { tmp = o[1]; o[1] = o[2]; o[2] = tmp; }
    if (d[o[0]] > d[o[1]]) { tmp = o[0]; o[0] = o[1]; o[1] = tmp; }
    for (int j = 0; j < 3; j++) {
        val[j] = d[o[j]];
        double col[3] = {v[0 * 3 + o[j]], v[1 * 3 + o[j]], v[2 * 3 + o[j]]};
        const double lead = col[0] != 0.0 ? col[0] : (col[1] != 0.0 ? col[1] : col[2]);
        const double sgn = lead < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < 3; k++) V[k * 3 + j] = sgn * col[k];
    }
}

// xyz: the cell's points in insertion order (stride floats apart), n of them; ctr: the cell centre; box: m_resolution of
// the cell (= the map's halved resolution, CMK:699)
LL_HD void cell_stats(const float *xyz, int stride, int n, const float ctr[3], float box, CellStats &s)
{
    s.type = CELL_FEATURE_SPHERE;
    for (int d = 0; d < 3; d++) s.vec[d] = s.mean[d] = s.eval[d] = 0.0f;
    for (int d = 0; d < 6; d++) s.cov[d] = 0.0f;
    if (n <= 0) return;
    float sum[3] = {0.f, 0.f, 0.f};  // m_xyz_sum, CMK:201-206
    for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) sum[d] = sum[d] + xyz[(size_t)i * stride + d];
    for (int d = 0; d < 3; d++) s.mean[d] = sum[d] / (float)n;  // CMK:233
    if (n < 5) return;                                          // CMK:446-451
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                // IF_COV_INIT_IDENTITY 0, CMK:294-297
    for (int i = 0; i < n; i++) {                               // CMK:304-307
        const float *p = xyz + (size_t)i * stride;
        c[0] = c[0] + p[0] * p[0];
        c[1] = c[1] + p[0] * p[1];
        c[2] = c[2] + p[0] * p[2];
        c[3] = c[3] + p[1] * p[1];
        c[4] = c[4] + p[1] * p[2];
        c[5] = c[5] + p[2] * p[2];
    }
    const float fn = (float)n, fn1 = (float)(n - 1);
    const float mm[6] = {s.mean[0] * s.mean[0], s.mean[0] * s.mean[1], s.mean[0] * s.mean[2],
                         s.mean[1] * s.mean[1], s.mean[1] * s.mean[2], s.mean[2] * s.mean[2]};
    for (int d = 0; d < 6; d++) s.cov[d] = (c[d] - fn * mm[d]) / fn1;  // CMK:309-310
    const double m[6] = {s.cov[0], s.cov[1], s.cov[2], s.cov[3], s.cov[4], s.cov[5]};
    double val[3], V[9];
    sym3_eigen(m, val, V);
    for (int d = 0; d < 3; d++) s.eval[d] = (float)val[d];
    const float dx = ctr[0] - s.mean[0], dy = ctr[1] - s.mean[1], dz = ctr[2] - s.mean[2];
    if ((double)sqrtf(dx * dx + dy * dy + dz * dz) > (double)box * 0.75) return;  // CMK:455-460
    const double third = 1.0 / 3.0;  // m_feature_determine_threshold_{plane,line}, CMK:76-77
    if ((double)s.eval[1] * third > (double)s.eval[0]) {  // CMK:462-467
        s.type = CELL_FEATURE_PLANE;
        for (int d = 0; d < 3; d++) s.vec[d] = (float)V[d * 3 + 0];
        return;
    }
    if ((double)s.eval[2] * third > (double)s.eval[1]) {  // CMK:468-472
        s.type = CELL_FEATURE_LINE;
        for (int d = 0; d < 3; d++) s.vec[d] = (float)V[d * 3 + 2];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Key-frame descriptors: Maps_keyframe::feature_direction / generate_feature_img / apply_guassian_blur
// (cell_map_keyframe.hpp:1071-1089, 1385-1427, 1360-1372) and max_similiarity_of_two_image (:1155-1224).
#define LL_KF_RES 60  // PHI_RESOLUTION = THETA_RESOLUTION, CMK:35-36
#define LL_KF_BLUR 4  // apply_guassian_blur( img, 4, 4 ): 9 x 9 kernel, sigma 4, CMK:1425-1426

// feature_direction (CMK:1071-1089) of v = R^T * feature_vector (float): bin indices in [0, 60)
LL_HD void feature_direction(const float v_in[3], int *phi_idx, int *theta_idx)
{
    double v[3] = {(double)v_in[0], (double)v_in[1], (double)v_in[2]};
    if (v_in[0] < 0)
        for (int d = 0; d < 3; d++) v[d] = (double)(float)(v_in[d] * -1.0f);  // vec_3d *= ( -1.0 ) on a float vector
    const double kPi = 3.14159265358979323846;
    const double phi_step = kPi / LL_KF_RES, theta_step = kPi / LL_KF_RES;
    const double phi = atan2(v[1], v[0]) + kPi / 2;
    const double theta = asin(v[2]) + kPi / 2;
    const int a = (int)floor(phi / phi_step);
    // |z| can exceed 1 by a rounding of the rotated unit vector: asin is NaN there (the reference then converts a NaN to
    // int); the vector is given the bin of the pole it overshot
    const int b = (theta == theta) ? (int)floor(theta / theta_step) : (v[2] > 0 ? LL_KF_RES - 1 : 0);
    *phi_idx = a < 0 ? 0 : (a >= LL_KF_RES ? LL_KF_RES - 1 : a);              // make_index_in_matrix_range, CMK:1056-1068
    *theta_idx = b < 0 ? 0 : (b >= LL_KF_RES ? LL_KF_RES - 1 : b);
}

// cv::getGaussianKernel( 9, 4, CV_32F ): exp in double, stored float, normalised by the double sum of the floats
LL_HD void kf_gauss_kernel(float k[2 * LL_KF_BLUR + 1])
{
    const int n = 2 * LL_KF_BLUR + 1;
    const double sigma = (double)LL_KF_BLUR, scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        k[i] = (float)exp(scale2x * x * x);
        sum += (double)k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) k[i] = (float)((double)k[i] * sum);
}

}  // namespace ll
