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
LL_HD bool cell_in_fov(const float c[3], const double q[4], const double t[3], double maximum_in_fov_angle)
{
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

}  // namespace ll
