/*
 * ll_oracle_voxel.c -- CPU ORACLE (test infrastructure only, see ll_oracle.h) for the voxel-grid
 * down-sampling that sits on both sides of the registration hot path in hku-mars/loam_livox:
 *   source/laser_feature_extractor.hpp:192-193,372-381   (after extraction, leaf plane_res/2 and line_res)
 *   source/laser_mapping.hpp:742-743                     (setLeafSize line_res / plane_res)
 *   source/laser_mapping.hpp:1367-1373                   (m_if_input_downsample_mode, before registration)
 *   source/laser_mapping.hpp:1434-1437                   (new features, after registration)
 *   source/laser_mapping.hpp:533-537                     (match buffer refresh)
 * The algorithm itself lives in PCL (pcl::VoxelGrid<pcl::PointXYZI>), which is absent here and not pinned by
 * the reference (README.md:51 mentions PCL 1.7 and 1.9).  PARITY UNPINNED.  This file restates the published
 * PCL 1.9 behaviour of VoxelGrid<PointT>::applyFilter (filters/include/pcl/filters/impl/voxel_grid.hpp) with
 * default settings (downsample_all_data = true, min_points_per_voxel = 0, no filter field, no leaf layout):
 *
 *   1. getMinMax3D over the points with finite x, y, z (float min / max per axis).
 *   2. inverse_leaf = 1.0f / leaf (float).  d{x,y,z} = (int64)((max - min) * inverse_leaf) + 1; if dx*dy*dz
 *      exceeds INT32_MAX PCL warns "Leaf size is too small for the input dataset" and copies the input to the
 *      output unchanged.  -> status 1.
 *   3. min_b = (int)floor(min * inverse_leaf), max_b likewise, div_b = max_b - min_b + 1,
 *      divb_mul = (1, div_b.x, div_b.x * div_b.y).
 *   4. per finite point: ijk = (int)(floor(p * inverse_leaf) - (float)min_b)  (float arithmetic),
 *      idx = ijk . divb_mul.
 *   5. sort by idx; one output point per distinct idx, in ascending idx order.
 *   6. output = CentroidPoint<PointXYZI>: float sums of x, y, z and of intensity over the voxel's points,
 *      each divided by (float)count.
 *
 * Deviations the oracle DEFINES:
 *   - PCL sorts with std::sort, which leaves the order of the points inside a voxel (and therefore the rounding
 *     of the float sums) implementation-defined.  Here the points of a voxel are added in ascending input order
 *     (a stable sort), which is what makes the result reproducible.
 *   - PCL skips non-finite points only when cloud.is_dense is false; here they are always skipped.
 *   - An input with no finite point gives an empty output (status 2); PCL would evaluate FLT_MAX - (-FLT_MAX).
 *   - An axis extent whose cell count does not fit an int64 cast counts as "leaf too small" (status 1).
 */
#include "ll_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t idx;
    int32_t pt;
} vox_key;

static int vox_cmp(const void *a, const void *b)
{
    const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0); /* stable: ascending input order inside a voxel */
}

int orc_voxel_grid(const float *xyzi, int32_t n, const float leaf[3], float *out_xyzi, int32_t *n_out)
{
    *n_out = 0;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int32_t n_valid = 0;
    for (int32_t i = 0; i < n; i++) { /* getMinMax3D, common.hpp */
        const float *p = xyzi + 4 * (size_t)i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int c = 0; c < 3; c++) {
            mn[c] = p[c] < mn[c] ? p[c] : mn[c];
            mx[c] = p[c] > mx[c] ? p[c] : mx[c];
        }
        n_valid++;
    }
    if (n_valid == 0) return 2;
    float inv[3];
    for (int c = 0; c < 3; c++) inv[c] = 1.0f / leaf[c];
    int64_t d[3];
    for (int c = 0; c < 3; c++) {
        const float e = (mx[c] - mn[c]) * inv[c];
        if (!(e < 9.0e18f)) goto passthrough;
        d[c] = (int64_t)e + 1;
    }
    /* dx*dy*dz > INT32_MAX, evaluated without overflowing the product */
    if (d[0] > INT32_MAX || d[1] > INT32_MAX || d[0] * d[1] > INT32_MAX || d[2] > INT32_MAX || d[0] * d[1] * d[2] > INT32_MAX)
        goto passthrough;
    {
        int32_t min_b[3], max_b[3], div_b[3], mul[3];
        for (int c = 0; c < 3; c++) {
            min_b[c] = (int32_t)floorf(mn[c] * inv[c]);
            max_b[c] = (int32_t)floorf(mx[c] * inv[c]);
            div_b[c] = max_b[c] - min_b[c] + 1;
        }
        mul[0] = 1;
        mul[1] = div_b[0];
        mul[2] = div_b[0] * div_b[1];
        vox_key *keys = (vox_key *)malloc(sizeof(vox_key) * (size_t)n_valid);
        int32_t m = 0;
        for (int32_t i = 0; i < n; i++) {
            const float *p = xyzi + 4 * (size_t)i;
            if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
            const int32_t i0 = (int32_t)(floorf(p[0] * inv[0]) - (float)min_b[0]);
            const int32_t i1 = (int32_t)(floorf(p[1] * inv[1]) - (float)min_b[1]);
            const int32_t i2 = (int32_t)(floorf(p[2] * inv[2]) - (float)min_b[2]);
            keys[m].idx = (uint32_t)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]);
            keys[m].pt = i;
            m++;
        }
        qsort(keys, (size_t)m, sizeof(vox_key), vox_cmp);
        int32_t k = 0, nv = 0;
        while (k < m) {
            int32_t e = k;
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            while (e < m && keys[e].idx == keys[k].idx) {
                const float *p = xyzi + 4 * (size_t)keys[e].pt;
                for (int c = 0; c < 4; c++) s[c] = s[c] + p[c]; /* AccumulatorXYZ / AccumulatorIntensity */
                e++;
            }
            const float cnt = (float)(e - k);
            for (int c = 0; c < 4; c++) out_xyzi[4 * (size_t)nv + c] = s[c] / cnt;
            nv++;
            k = e;
        }
        free(keys);
        *n_out = nv;
        return 0;
    }
passthrough:
    memcpy(out_xyzi, xyzi, sizeof(float) * 4 * (size_t)n);
    *n_out = n;
    return 1;
}
