// ll_cellmap.h -- device buffers of the cell map (ll_cellmap_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "ll_cellmap_core.h"

namespace ll {

// Points of all cells in one array, ordered by (cell key, insertion order); a table of the occupied cells beside it.
struct CellMapDev {
    int cap;                 // points the map can hold; the arrays below are 2 * cap long (append / replace scratch)
    CellGeom geom;
    float resolution;
    int revisit_threshold;   // m_minimum_revisit_threshold (CMK:506)
    int frame;               // m_current_frame_idx (CMK:505)
    int n_pts, n_cells;      // host mirrors of the device state
    float4 *pts, *pts2;                    // {x, y, z, 0}: the reference's cells hold xyz only (CMK:82, pcl_tools.hpp:94-101)
    unsigned long long *pkey, *pkey2;      // cell key of every stored point
    unsigned int *val, *val2;
    unsigned long long *ckey, *ckey2;      // occupied cells, ascending
    int *cstart, *cstart2;                 // [n_cells + 1] first point of each cell
    int *clast, *clast2;                   // m_last_update_frame_idx (CMK:75)
    unsigned int *flag, *rank;             // head flags / prefix sums over points or cells
    unsigned int *csel, *csel_rank;        // per cell: selected by the last query / its rank among the selected
    unsigned long long *skey, *skey2;      // (selected cell rank, leaf) sort keys of the per-cell VoxelGrid
    unsigned int *head_pos;
    float4 *filt;                          // per-cell filtered clouds, concatenated in cell order
    unsigned long long *filt_key;          // cell key of every filtered point
    int *counts;                           // device scalars: [0] valid points, [1] cells, [2] selected cells, [3] voxels
    int n_filt, n_sel;                     // result of the last query
    void *tmp;
    size_t tmp_bytes;
};

int cellmap_alloc(CellMapDev &m, int cap, float resolution, int revisit_threshold, const char **err);
void cellmap_free(CellMapDev &m);
// capacity -> new_cap (no-op when not larger), content kept
int cellmap_grow(CellMapDev &m, int new_cap, hipStream_t s, const char **err);
// append_cloud (CMK:619-672): n points at d_src (device)
int cellmap_append(CellMapDev &m, const float4 *d_src, int n, hipStream_t s, const char **err);
// after cellmap_append of n_appended points onto n_before stored ones: points received per cell of the new table -> m.csel
int cellmap_touch_counts(CellMapDev &m, int n_before, int n_appended, hipStream_t s, const char **err);
// find_cells_in_radius + if_pt_in_fov + per-cell VoxelGrid (+ set_pointcloud), LM:475-513 for one feature kind;
// d_pose: device copy of {qx, qy, qz, qw, tx, ty, tz}.  Result in m.filt[0 .. m.n_filt)
int cellmap_query_filter(CellMapDev &m, const double *d_pose, float radius, float max_fov_deg, float leaf, int replace, hipStream_t s,
                         const char **err);

// result block of cellmap_keyframe_images (device memory)
struct KfOut {
    float img[4][LL_KF_RES * LL_KF_RES];  // m_feature_img_line, _plane, _line_roi, _plane_roi ([phi][theta], blurred)
    float ratio[4];                       // ratio_of_nonzero_in_img of the same before the blur
    float R[2][9];                        // m_eigen_R, m_eigen_R_roi (row-major)
    int n_vec[4];                         // feature vectors that entered each image
    float centre[3];                      // key-frame centre (get_center)
    float roi_range;                      // m_roi_range
    int n_distinct;
};

// Maps_keyframe::analyze over all cells of the map (CMK:1429-1493): cell features, ROI range, the four direction images
int cellmap_keyframe_images(CellMapDev &m, CellStats *d_stats, float roi_ratio, KfOut *d_out, hipStream_t s, const char **err);
// max_similiarity_of_two_image (CMK:1155-1224): d_a, d_b 60 x 60 device images
int keyframe_similarity(const float *d_a, const float *d_b, float *d_result, hipStream_t s, const char **err);

// determine_feature (CMK:436-473) for every cell, in cell-table order; d_out: device array of n_cells entries
int cellmap_stats(CellMapDev &m, CellStats *d_out, hipStream_t s, const char **err);

}  // namespace ll
