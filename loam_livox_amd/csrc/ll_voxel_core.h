// ll_voxel_core.h -- the arithmetic of pcl::VoxelGrid<pcl::PointXYZI>::applyFilter (PCL 1.9,
// filters/impl/voxel_grid.hpp) that decides which voxel a point falls in; shared by the HIP kernels and the host
// side of the C ABI.  Used by hku-mars/loam_livox at source/laser_feature_extractor.hpp:192-193,372-381 and
// source/laser_mapping.hpp:742-743,1367-1373,1434-1437,533-537.  All float, no contraction: the same operations in
// the same order as PCL, so the voxel of every point -- an integer -- is reproduced exactly.
#pragma once
#include <stdint.h>

#include "ll_fe_core.h"  // LL_HD, ll_isfinite

namespace ll {

enum : int { VOX_OK = 0, VOX_PASSTHROUGH = 1, VOX_EMPTY = 2 };

struct VoxelParams {
    int status;      // VOX_*
    int min_b[3];    // floor(min * inverse_leaf)
    int mul[3];      // divb_mul = (1, div_b.x, div_b.x * div_b.y)
};

// steps 2-3 of applyFilter from the bounding box of the finite points (n_valid of them)
LL_HD void voxel_params(const float mn[3], const float mx[3], int n_valid, const float inv[3], VoxelParams &p)
{
    p.status = VOX_OK;
    for (int c = 0; c < 3; c++) p.min_b[c] = p.mul[c] = 0;
    if (n_valid <= 0) {
        p.status = VOX_EMPTY;
        return;
    }
    long long d[3];
    for (int c = 0; c < 3; c++) {
        const float e = (mx[c] - mn[c]) * inv[c];
        if (!(e < 9.0e18f)) {  // the int64 cast of PCL would be undefined: treated as "leaf too small"
            p.status = VOX_PASSTHROUGH;
            return;
        }
        d[c] = (long long)e + 1;
    }
    const long long lim = 2147483647LL;
    if (d[0] > lim || d[1] > lim || d[0] * d[1] > lim || d[2] > lim || d[0] * d[1] * d[2] > lim) {
        p.status = VOX_PASSTHROUGH;  // "Leaf size is too small for the input dataset": output = input
        return;
    }
    int div_b[3];
    for (int c = 0; c < 3; c++) {
        p.min_b[c] = (int)floorf(mn[c] * inv[c]);
        const int max_b = (int)floorf(mx[c] * inv[c]);
        div_b[c] = max_b - p.min_b[c] + 1;
    }
    p.mul[0] = 1;
    p.mul[1] = div_b[0];
    p.mul[2] = div_b[0] * div_b[1];
}

// step 4: centroid leaf index of one finite point
LL_HD unsigned int voxel_index(float x, float y, float z, const float inv[3], const VoxelParams &p)
{
    const int i0 = (int)(floorf(x * inv[0]) - (float)p.min_b[0]);
    const int i1 = (int)(floorf(y * inv[1]) - (float)p.min_b[1]);
    const int i2 = (int)(floorf(z * inv[2]) - (float)p.min_b[2]);
    return (unsigned int)(i0 * p.mul[0] + i1 * p.mul[1] + i2 * p.mul[2]);
}

}  // namespace ll
