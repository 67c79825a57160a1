// ll_cellmap_kernels.hip -- the cell map behind the "cube" matching mode (m_matching_mode == 1) on the device:
// Points_cloud_map<float>::append_cloud / find_cells_in_radius (source/cell_map_keyframe.hpp:619-672, 761-788) and the
// cell part of Laser_mapping::update_buff_for_matching (source/laser_mapping.hpp:471-513).
//
// The reference keeps a hash map from cell centre to a heap-allocated cell with its own point vector, and an octree
// over the centres.  Here the whole map is one point array ordered by (cell key, insertion order) plus a sorted table
// of the occupied cells:
//   append     new points get their cell key, cells that were not updated for `revisit_threshold` frames and are hit
//              again lose their old points (the reference swaps in a fresh cell, CMK:735-756), then one stable radix
//              sort of (key, position) restores the order and the cell table is rebuilt from the key runs;
//   query      one thread per cell evaluates radius + field of view on the cell centre; the selected cells are ranked in
//              key order; every point of a selected cell gets the key (cell rank, leaf z, leaf y, leaf x); one stable
//              sort and one thread per leaf reproduce pcl::VoxelGrid on each cell separately (float sums in insertion
//              order); with down_sample_replace the leaves replace the points of their cells (one more re-sort).
// Byte / integer work bound by the radix sorts; per-frame, not per-ICP-iteration.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <utility>

#include "ll_cellmap.h"

namespace ll {

typedef unsigned long long u64;
typedef unsigned int u32;

#define CMCHK(x)                              \
    do {                                      \
        hipError_t e_ = (x);                  \
        if (e_ != hipSuccess) {               \
            *err = hipGetErrorString(e_);     \
            return -1;                        \
        }                                     \
    } while (0)

static inline int blocks(int n) { return (n + 255) / 256 > 0 ? (n + 255) / 256 : 1; }

// position of `k` in the ascending table ckey[0 .. n), or -1
__device__ __forceinline__ int cell_find(const u64 *ckey, int n, u64 k)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ckey[mid] < k)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < n && ckey[lo] == k) ? lo : -1;
}

// ---------------------------------------------------------------------------------------------------------- append
// new points behind the stored ones: key, {x, y, z, 0}; a hit on a stale cell marks it for reset
__global__ __launch_bounds__(256) void cm_new_points_kernel(const float4 *src, int n, int n_old, CellGeom g, const u64 *ckey, const int *clast,
                                                            int n_cells, int frame, int thr, float4 *pts, u64 *pkey, u32 *creset)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = src[i];
    int k[3];
    u64 key = LL_CELL_KEY_NONE;
    if (ll_isfinite(p.x) && ll_isfinite(p.y) && ll_isfinite(p.z) && cell_index(p.x, p.y, p.z, g, k)) {
        key = cell_pack(k);
        const int c = cell_find(ckey, n_cells, key);
        if (c >= 0 && !(frame - clast[c] < thr)) creset[c] = 1u;  // CMK:737 fails -> a fresh cell replaces the old one
    }
    pts[n_old + i] = make_float4(p.x, p.y, p.z, 0.0f);
    pkey[n_old + i] = key;
}

// stored points of the cells marked for reset are dropped
__global__ __launch_bounds__(256) void cm_drop_reset_kernel(int n_old, const u64 *ckey, int n_cells, const u32 *creset, u64 *pkey)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_old) return;
    const int c = cell_find(ckey, n_cells, pkey[i]);
    if (c >= 0 && creset[c]) pkey[i] = LL_CELL_KEY_NONE;
}

__global__ __launch_bounds__(256) void cm_iota_kernel(u32 *v, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (u32)i;
}

// after the sort: gather the points, flag the first point of every cell, find the number of valid points
__global__ __launch_bounds__(256) void cm_gather_kernel(const float4 *pts, const u64 *key_sorted, const u32 *val_sorted, int total, float4 *pts_out,
                                                        u32 *flag, int *counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const u64 k = key_sorted[i];
    pts_out[i] = pts[val_sorted[i]];
    const bool valid = k != LL_CELL_KEY_NONE;
    flag[i] = (valid && (i == 0 || key_sorted[i - 1] != k)) ? 1u : 0u;
    if (valid && (i + 1 == total || key_sorted[i + 1] == LL_CELL_KEY_NONE)) counts[0] = i + 1;
}

// rank = exclusive prefix of flag: the cell table from the key runs
__global__ __launch_bounds__(256) void cm_cells_kernel(const u64 *key_sorted, const u32 *flag, const u32 *rank, const int *counts, u64 *ckey,
                                                       int *cstart, int *counts_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n_valid = counts[0];
    if (i >= n_valid) return;
    if (flag[i]) {
        ckey[rank[i]] = key_sorted[i];
        cstart[rank[i]] = i;
    }
    if (i == n_valid - 1) {
        const int n_cells = (int)(rank[i] + flag[i]);
        counts_out[1] = n_cells;
        cstart[n_cells] = n_valid;
    }
}

// m_last_update_frame_idx of the rebuilt table: carried over from the old table, `frame` for cells that are new
__global__ __launch_bounds__(256) void cm_carry_kernel(const u64 *ckey_new, int n_new, const u64 *ckey_old, const int *clast_old, int n_old,
                                                       int frame, int *clast_new)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_new) return;
    const int j = cell_find(ckey_old, n_old, ckey_new[c]);
    clast_new[c] = j >= 0 ? clast_old[j] : frame;
}

// ... and `frame` for every cell the appended points fell in (CMK:739, 700-702, 751-752)
__global__ __launch_bounds__(256) void cm_touch_kernel(const u64 *pkey_new, int n, const u64 *ckey, int n_cells, int frame, int *clast)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 k = pkey_new[i];
    if (k == LL_CELL_KEY_NONE) return;
    const int c = cell_find(ckey, n_cells, k);
    if (c >= 0) clast[c] = frame;  // every writer stores the same value
}

// how many of the points appended last fell in each cell (append_cloud's appeared_cell_count, CMK:638-651)
__global__ __launch_bounds__(256) void cm_touch_count_kernel(const u64 *pkey_new, int n, const u64 *ckey, int n_cells, u32 *cnt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 k = pkey_new[i];
    if (k == LL_CELL_KEY_NONE) return;
    const int c = cell_find(ckey, n_cells, k);
    if (c >= 0) atomicAdd(&cnt[c], 1u);
}

// ----------------------------------------------------------------------------------------------------------- query
__global__ __launch_bounds__(256) void cm_select_kernel(const u64 *ckey, int n_cells, CellGeom g, const double *pose, float radius,
                                                        double max_fov_deg, u32 *csel)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    int k[3];
    cell_unpack(ckey[c], k);
    float ctr[3];
    cell_centre(k, g, ctr);
    const double q[4] = {pose[0], pose[1], pose[2], pose[3]}, t[3] = {pose[4], pose[5], pose[6]};
    const float sp[3] = {(float)t[0], (float)t[1], (float)t[2]};  // eigen_to_pcl_pt<pcl::PointXYZ>( m_t_w_curr )
    csel[c] = (cell_in_radius(ctr, sp, radius) && cell_in_fov(ctr, q, t, max_fov_deg)) ? 1u : 0u;
}

__global__ void cm_count_sel_kernel(const u32 *csel, const u32 *csel_rank, int n_cells, int *counts)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) counts[2] = n_cells > 0 ? (int)(csel_rank[n_cells - 1] + csel[n_cells - 1]) : 0;
}

// sort key of the per-cell VoxelGrid: (rank of the cell among the selected) << 30 | leaf z << 20 | leaf y << 10 | leaf x
__global__ __launch_bounds__(256) void cm_leaf_key_kernel(const float4 *pts, const u64 *pkey, int n_pts, const u64 *ckey, int n_cells,
                                                          const u32 *csel, const u32 *csel_rank, CellGeom g, float inv_leaf, u64 *skey, u32 *val)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pts) return;
    u64 key = LL_CELL_KEY_NONE;
    const u64 ck = pkey[i];
    const int c = cell_find(ckey, n_cells, ck);
    if (c >= 0 && csel[c]) {
        int k[3];
        cell_unpack(ck, k);
        const float4 p = pts[i];
        int l[3] = {cell_leaf_local(p.x, k[0], g, inv_leaf), cell_leaf_local(p.y, k[1], g, inv_leaf), cell_leaf_local(p.z, k[2], g, inv_leaf)};
        for (int d = 0; d < 3; d++) l[d] = l[d] < 0 ? 0 : (l[d] > 1023 ? 1023 : l[d]);
        key = ((u64)csel_rank[c] << 30) | ((u64)l[2] << 20) | ((u64)l[1] << 10) | (u64)l[0];
    }
    skey[i] = key;
    val[i] = (u32)i;
}

__global__ __launch_bounds__(256) void cm_leaf_head_kernel(const u64 *skey_sorted, int n, u32 *flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 k = skey_sorted[i];
    flag[i] = (k != LL_CELL_KEY_NONE && (i == 0 || skey_sorted[i - 1] != k)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void cm_leaf_pos_kernel(const u32 *flag, const u32 *rank, int n, u32 *head_pos, int *counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) head_pos[rank[i]] = (u32)i;
    if (i == n - 1) counts[3] = (int)(rank[i] + flag[i]);
}

// one thread per leaf: the centroid of pcl::VoxelGrid, float sums in insertion order (ll_voxel_kernels.hip does the
// same for whole clouds).  The intensity of a cell point is 0, so is the centroid's.
__global__ __launch_bounds__(256) void cm_centroid_kernel(const float4 *pts, const u64 *pkey, const u64 *skey_sorted, const u32 *val_sorted,
                                                          const u32 *head_pos, const int *counts, int n_pts, float4 *filt, u64 *filt_key)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= counts[3]) return;
    const int i = (int)head_pos[t];
    const u64 k = skey_sorted[i];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    int cnt = 0;
    bool more = true;
    for (int j = i; more && j < n_pts; j += 4) {
        bool ok[4];
        u32 v[4];
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ok[u] = (j + u < n_pts) && skey_sorted[j + u] == k;
#pragma unroll
        for (int u = 1; u < 4; u++) ok[u] = ok[u] && ok[u - 1];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = ok[u] ? val_sorted[j + u] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = pts[v[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (ok[u]) {
                sx = sx + p[u].x;
                sy = sy + p[u].y;
                sz = sz + p[u].z;
                si = si + p[u].w;
                cnt++;
            }
        }
        more = ok[3];
    }
    const float c = (float)cnt;
    filt[t] = make_float4(sx / c, sy / c, sz / c, si / c);
    filt_key[t] = pkey[val_sorted[i]];
}

// down_sample_replace (LM:492-495): the points of the selected cells go, their leaves come in behind the stored points
__global__ __launch_bounds__(256) void cm_replace_kernel(int n_pts, const u64 *ckey, int n_cells, const u32 *csel, const float4 *filt,
                                                         const u64 *filt_key, int n_filt, float4 *pts, u64 *pkey)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_pts) {
        const int c = cell_find(ckey, n_cells, pkey[i]);
        if (c >= 0 && csel[c]) pkey[i] = LL_CELL_KEY_NONE;
    } else if (i < n_pts + n_filt) {
        pts[i] = filt[i - n_pts];
        pkey[i] = filt_key[i - n_pts];
    }
}

// ------------------------------------------------------------------------------------------------------- statistics
// determine_feature for every cell (CMK:436-473): one thread per cell walks the cell's points (contiguous in the store)
// twice -- sum, then the six second moments -- in insertion order, as the reference's float accumulators do.
__global__ __launch_bounds__(128) void cm_stats_kernel(const float4 *pts, const u64 *ckey, const int *cstart, int n_cells, CellGeom g,
                                                       CellStats *out)
{
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= n_cells) return;
    int k[3];
    cell_unpack(ckey[c], k);
    float ctr[3];
    cell_centre(k, g, ctr);
    const int first = cstart[c];
    CellStats s;
    cell_stats((const float *)(pts + first), 4, cstart[c + 1] - first, ctr, g.box, s);
    out[c] = s;
}

int cellmap_stats(CellMapDev &m, CellStats *d_out, hipStream_t s, const char **err)
{
    if (m.n_cells > 0) hipLaunchKernelGGL(cm_stats_kernel, dim3((m.n_cells + 127) / 128), dim3(128), 0, s, m.pts, m.ckey, m.cstart, m.n_cells, m.geom, d_out);
    CMCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- key-frame images
// get_center (CMK:1291-1301): float sum of the cell centres in cell order, times (float)(1 / n)
__global__ void cm_kf_centre_kernel(const u64 *ckey, int n_cells, CellGeom g, KfOut *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < n_cells; c++) {
        int k[3];
        cell_unpack(ckey[c], k);
        float ctr[3];
        cell_centre(k, g, ctr);
        for (int d = 0; d < 3; d++) s[d] = s[d] + ctr[d];
    }
    const float inv = (float)(1.0 / (double)(float)n_cells);
    for (int d = 0; d < 3; d++) out->centre[d] = s[d] * inv;
}

// |centre of the cell - key-frame centre| in float (CMK:1309-1315, 1462)
__global__ __launch_bounds__(256) void cm_kf_dist_kernel(const u64 *ckey, int n_cells, CellGeom g, const KfOut *out, float *dist)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    int k[3];
    cell_unpack(ckey[c], k);
    float ctr[3];
    cell_centre(k, g, ctr);
    const float dx = ctr[0] - out->centre[0], dy = ctr[1] - out->centre[1], dz = ctr[2] - out->centre[2];
    dist[c] = sqrtf(dx * dx + dy * dy + dz * dz);
}

__global__ __launch_bounds__(256) void cm_kf_distinct_kernel(const float *sorted, int n, u32 *flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
}

// get_ratio_range_of_cell (CMK:1303-1319): element ceil((size - 1) * ratio) of the std::set<float> of distances
__global__ __launch_bounds__(256) void cm_kf_pick_kernel(const float *sorted, const u32 *flag, const u32 *rank, int n, float ratio, KfOut *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int n_distinct = (int)(rank[n - 1] + flag[n - 1]);
    const int target = (int)ceilf((float)(n_distinct - 1) * ratio);
    if (flag[i] && (int)rank[i] == target) out->roi_range = sorted[i];
    if (i == 0) out->n_distinct = n_distinct;
}

#define KF_THREADS 1024
#define KF_BINS (LL_KF_RES * LL_KF_RES)

// generate_feature_img (CMK:1385-1427) for the whole key frame (block 0) and for the cells within roi_range of its centre
// (block 1).  One workgroup each: the work is a few passes over the cell statistics and a 60 x 60 image.
__global__ __launch_bounds__(KF_THREADS) void cm_kf_image_kernel(const CellStats *st, const float *dist, int n_cells, int use_roi, KfOut *out)
{
    __shared__ double s_red[KF_THREADS / 64][6];
    __shared__ float s_R[9];
    __shared__ int s_hist[2][KF_BINS];
    __shared__ float s_img[KF_BINS], s_tmp[KF_BINS];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int roi = blockIdx.x;
    if (roi && !use_roi) return;
    const float range = out->roi_range;
    for (int e = tid; e < KF_BINS; e += KF_THREADS) s_hist[0][e] = s_hist[1][e] = 0;
    if (tid < 4) s_cnt[tid] = 0;
    // eigen_decompose_of_featurevector (CMK:1554-1567): I + sum of (v v^T) (float products) in double, over the plane cells
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int c = tid; c < n_cells; c += KF_THREADS) {
        if (st[c].type != CELL_FEATURE_PLANE || (roi && !(dist[c] < range))) continue;
        const float *v = st[c].vec;
        acc[0] += (double)(v[0] * v[0]);
        acc[1] += (double)(v[0] * v[1]);
        acc[2] += (double)(v[0] * v[2]);
        acc[3] += (double)(v[1] * v[1]);
        acc[4] += (double)(v[1] * v[2]);
        acc[5] += (double)(v[2] * v[2]);
    }
    for (int e = 0; e < 6; e++) {  // fixed-order reduction: lanes, then waves
        double x = acc[e];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if (lane == 0) s_red[wave][e] = x;
    }
    __syncthreads();
    if (tid == 0) {
        double m[6];
        for (int e = 0; e < 6; e++) {
            double x = 0.0;
            for (int w = 0; w < KF_THREADS / 64; w++) x += s_red[w][e];
            m[e] = x;
        }
        m[0] += 1.0;  // mat_cov.setIdentity()
        m[3] += 1.0;
        m[5] += 1.0;
        double val[3], V[9];
        sym3_eigen(m, val, V);
        // rowwise().reverse(): largest eigenvalue first; the third axis is the cross product of the first two (CMK:1394-1396)
        float R[9];
        for (int k = 0; k < 3; k++) {
            R[k * 3 + 0] = (float)V[k * 3 + 2];
            R[k * 3 + 1] = (float)V[k * 3 + 1];
        }
        R[0 * 3 + 2] = R[1 * 3 + 0] * R[2 * 3 + 1] - R[2 * 3 + 0] * R[1 * 3 + 1];
        R[1 * 3 + 2] = R[2 * 3 + 0] * R[0 * 3 + 1] - R[0 * 3 + 0] * R[2 * 3 + 1];
        R[2 * 3 + 2] = R[0 * 3 + 0] * R[1 * 3 + 1] - R[1 * 3 + 0] * R[0 * 3 + 1];
        for (int e = 0; e < 9; e++) s_R[e] = out->R[roi][e] = R[e];
    }
    __syncthreads();
    // direction histograms (CMK:1409-1421): image 0 = line cells, 1 = plane cells
    for (int c = tid; c < n_cells; c += KF_THREADS) {
        const int type = st[c].type;
        if (type == CELL_FEATURE_SPHERE || (roi && !(dist[c] < range))) continue;
        const float *v = st[c].vec;
        float a[3];
        for (int j = 0; j < 3; j++) a[j] = (s_R[0 * 3 + j] * v[0] + s_R[1 * 3 + j] * v[1]) + s_R[2 * 3 + j] * v[2];  // R^T v
        int pi, ti;
        feature_direction(a, &pi, &ti);
        const int which = type == CELL_FEATURE_PLANE ? 1 : 0;
        atomicAdd(&s_hist[which][pi * LL_KF_RES + ti], 1);  // integer counts: order-independent
        atomicAdd(&s_cnt[which], 1);
    }
    __syncthreads();
    float gk[2 * LL_KF_BLUR + 1];
    kf_gauss_kernel(gk);
    for (int which = 0; which < 2; which++) {
        // ratio_of_nonzero_in_img (CMK:1142-1152), then the 9 x 9 Gaussian on the wrap-padded image (CMK:1360-1372)
        int nz = 0;
        for (int e = tid; e < KF_BINS; e += KF_THREADS) {
            s_img[e] = (float)s_hist[which][e];
            nz += s_hist[which][e] >= 1 ? 1 : 0;
        }
        for (int off = 32; off > 0; off >>= 1) nz += __shfl_down(nz, off);
        if (lane == 0 && nz) atomicAdd(&s_cnt[2 + which], nz);
        __syncthreads();
        for (int e = tid; e < KF_BINS; e += KF_THREADS) {  // along a row
            const int r = e / LL_KF_RES, col = e % LL_KF_RES;
            float x = 0.f;
            for (int k = 0; k < 2 * LL_KF_BLUR + 1; k++) x = x + gk[k] * s_img[r * LL_KF_RES + (col + k - LL_KF_BLUR + LL_KF_RES) % LL_KF_RES];
            s_tmp[e] = x;
        }
        __syncthreads();
        float *dst = out->img[2 * roi + which];
        for (int e = tid; e < KF_BINS; e += KF_THREADS) {  // along a column
            const int r = e / LL_KF_RES, col = e % LL_KF_RES;
            float x = 0.f;
            for (int k = 0; k < 2 * LL_KF_BLUR + 1; k++) x = x + gk[k] * s_tmp[((r + k - LL_KF_BLUR + LL_KF_RES) % LL_KF_RES) * LL_KF_RES + col];
            dst[e] = x;
        }
        __syncthreads();
    }
    if (tid < 2) {
        out->n_vec[2 * roi + tid] = s_cnt[tid];
        out->ratio[2 * roi + tid] = (float)s_cnt[2 + tid] / (float)KF_BINS;
    }
}

// max_similiarity_of_two_image (CMK:1155-1224): cv::matchTemplate( wrap-padded b, a, CV_TM_CCORR_NORMED ) and its maximum,
// i.e. the largest normalised correlation of a with b over the circular shifts -30 .. +30 of both axes.
__global__ __launch_bounds__(KF_THREADS) void cm_kf_similarity_kernel(const float *a, const float *b, float *result)
{
    __shared__ float s_a[KF_BINS], s_b[KF_BINS];
    __shared__ double s_w[KF_THREADS / 64][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double na = 0.0, nb = 0.0;
    for (int e = tid; e < KF_BINS; e += KF_THREADS) {
        s_a[e] = a[e];
        s_b[e] = b[e];
        na += (double)a[e] * (double)a[e];
        nb += (double)b[e] * (double)b[e];
    }
    __syncthreads();
    const int half = LL_KF_RES / 2, span = LL_KF_RES + 1;  // result is 61 x 61
    double best = -1.0e300;
    for (int sft = tid; sft < span * span; sft += KF_THREADS) {
        const int Y = sft / span, X = sft % span;
        double num = 0.0;
        for (int i = 0; i < LL_KF_RES; i++) {
            const int bi = (Y + i - half + LL_KF_RES) % LL_KF_RES;
            for (int j = 0; j < LL_KF_RES; j++) num += (double)s_a[i * LL_KF_RES + j] * (double)s_b[bi * LL_KF_RES + (X + j - half + LL_KF_RES) % LL_KF_RES];
        }
        best = num > best ? num : best;
    }
    for (int off = 32; off > 0; off >>= 1) {
        na += __shfl_down(na, off);
        nb += __shfl_down(nb, off);
        const double o = __shfl_down(best, off);
        best = o > best ? o : best;
    }
    if (lane == 0) {
        s_w[wave][0] = na;
        s_w[wave][1] = nb;
        s_w[wave][2] = best;
    }
    __syncthreads();
    if (tid == 0) {
        double A = 0.0, B = 0.0, M = -1.0e300;
        for (int w = 0; w < KF_THREADS / 64; w++) {
            A += s_w[w][0];
            B += s_w[w][1];
            M = s_w[w][2] > M ? s_w[w][2] : M;
        }
        const double t = sqrt(A * B);
        *result = t > 0.0 ? (float)(M / t) : 0.0f;
    }
}

int cellmap_keyframe_images(CellMapDev &m, CellStats *d_stats, float roi_ratio, KfOut *d_out, hipStream_t s, const char **err)
{
    CMCHK(hipMemsetAsync(d_out, 0, sizeof(KfOut), s));
    if (m.n_cells == 0) return 0;
    if (cellmap_stats(m, d_stats, s, err)) return -1;
    const int nc = m.n_cells;
    float *dist = (float *)m.val, *sorted = (float *)m.val2;  // scratch of the sorts, free between operations
    const int use_roi = roi_ratio > 0.f ? 1 : 0;
    if (use_roi) {
        hipLaunchKernelGGL(cm_kf_centre_kernel, dim3(1), dim3(64), 0, s, m.ckey, nc, m.geom, d_out);
        hipLaunchKernelGGL(cm_kf_dist_kernel, dim3(blocks(nc)), dim3(256), 0, s, m.ckey, nc, m.geom, d_out, dist);
        size_t tb = m.tmp_bytes;
        CMCHK(hipcub::DeviceRadixSort::SortKeys(m.tmp, tb, dist, sorted, nc, 0, 32, s));
        hipLaunchKernelGGL(cm_kf_distinct_kernel, dim3(blocks(nc)), dim3(256), 0, s, sorted, nc, m.flag);
        tb = m.tmp_bytes;
        CMCHK(hipcub::DeviceScan::ExclusiveSum(m.tmp, tb, m.flag, m.rank, nc, s));
        hipLaunchKernelGGL(cm_kf_pick_kernel, dim3(blocks(nc)), dim3(256), 0, s, sorted, m.flag, m.rank, nc, roi_ratio, d_out);
    }
    hipLaunchKernelGGL(cm_kf_image_kernel, dim3(2), dim3(KF_THREADS), 0, s, d_stats, dist, nc, use_roi, d_out);
    CMCHK(hipGetLastError());
    return 0;
}

int keyframe_similarity(const float *d_a, const float *d_b, float *d_result, hipStream_t s, const char **err)
{
    hipLaunchKernelGGL(cm_kf_similarity_kernel, dim3(1), dim3(KF_THREADS), 0, s, d_a, d_b, d_result);
    CMCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------ host
int cellmap_alloc(CellMapDev &m, int cap, float resolution, int revisit_threshold, const char **err)
{
    memset(&m, 0, sizeof(m));
    if (cap < 1 || (long long)cap * 2 >= 0x7fffffffLL) {
        *err = "cell map capacity out of range";
        return -1;
    }
    if (!(resolution > 0.f)) {
        *err = "cell resolution must be positive";
        return -1;
    }
    m.cap = cap;
    m.resolution = resolution;
    m.geom = cell_geom(resolution);
    m.revisit_threshold = revisit_threshold;
    const size_t n2 = (size_t)cap * 2;
    CMCHK(hipMalloc(&m.pts, n2 * sizeof(float4)));
    CMCHK(hipMalloc(&m.pts2, n2 * sizeof(float4)));
    CMCHK(hipMalloc(&m.pkey, n2 * sizeof(u64)));
    CMCHK(hipMalloc(&m.pkey2, n2 * sizeof(u64)));
    CMCHK(hipMalloc(&m.val, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.val2, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.ckey, (n2 + 1) * sizeof(u64)));
    CMCHK(hipMalloc(&m.ckey2, (n2 + 1) * sizeof(u64)));
    CMCHK(hipMalloc(&m.cstart, (n2 + 1) * sizeof(int)));
    CMCHK(hipMalloc(&m.cstart2, (n2 + 1) * sizeof(int)));
    CMCHK(hipMalloc(&m.clast, (n2 + 1) * sizeof(int)));
    CMCHK(hipMalloc(&m.clast2, (n2 + 1) * sizeof(int)));
    CMCHK(hipMalloc(&m.flag, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.rank, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.csel, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.csel_rank, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.skey, n2 * sizeof(u64)));
    CMCHK(hipMalloc(&m.skey2, n2 * sizeof(u64)));
    CMCHK(hipMalloc(&m.head_pos, n2 * sizeof(u32)));
    CMCHK(hipMalloc(&m.filt, (size_t)cap * sizeof(float4)));
    CMCHK(hipMalloc(&m.filt_key, (size_t)cap * sizeof(u64)));
    CMCHK(hipMalloc(&m.counts, 8 * sizeof(int)));
    CMCHK(hipMemset(m.counts, 0, 8 * sizeof(int)));
    CMCHK(hipStreamSynchronize(nullptr));  // (a null-stream memset is not ordered with the non-blocking streams the map is used on)
    size_t t1 = 0, t2 = 0;
    CMCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, m.pkey, m.pkey2, m.val, m.val2, (int)n2, 0, 64));
    CMCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, m.flag, m.rank, (int)n2));
    m.tmp_bytes = t1 > t2 ? t1 : t2;
    CMCHK(hipMalloc(&m.tmp, m.tmp_bytes));
    return 0;
}

// more room, same content: the stored points, their keys, the cell table and the frame counter move to arrays of the new capacity
int cellmap_grow(CellMapDev &m, int new_cap, hipStream_t s, const char **err)
{
    if (new_cap <= m.cap) return 0;
    CMCHK(hipStreamSynchronize(s));
    CellMapDev n;
    if (cellmap_alloc(n, new_cap, m.resolution, m.revisit_threshold, err)) {
        cellmap_free(n);
        return -1;
    }
    hipError_t e = hipSuccess;  // (a failing copy must not leak the ~25 arrays of the new map)
    auto copy = [&](void *dst, const void *src, size_t bytes) {
        if (e == hipSuccess && bytes > 0) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
    };
    copy(n.pts, m.pts, (size_t)m.n_pts * sizeof(float4));
    copy(n.pkey, m.pkey, (size_t)m.n_pts * sizeof(u64));
    if (m.n_cells > 0) {
        copy(n.ckey, m.ckey, (size_t)m.n_cells * sizeof(u64));
        copy(n.cstart, m.cstart, (size_t)(m.n_cells + 1) * sizeof(int));
        copy(n.clast, m.clast, (size_t)m.n_cells * sizeof(int));
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        cellmap_free(n);
        *err = hipGetErrorString(e);
        return -1;
    }
    n.frame = m.frame;
    n.n_pts = m.n_pts;
    n.n_cells = m.n_cells;
    cellmap_free(m);
    m = n;
    return 0;
}

void cellmap_free(CellMapDev &m)
{
    void *ptrs[] = {m.pts,  m.pts2, m.pkey,  m.pkey2,     m.val,  m.val2,  m.ckey,     m.ckey2, m.cstart,   m.cstart2, m.clast, m.clast2,
                    m.flag, m.rank, m.csel,  m.csel_rank, m.skey, m.skey2, m.head_pos, m.filt,  m.filt_key, m.counts,  m.tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    memset(&m, 0, sizeof(m));
}

// pts / pkey [0 .. total) (dropped entries carry the NONE key) -> ordered store + rebuilt cell table.  n_appended: the
// last n_appended entries are new points whose cells are stamped with the current frame.
static int cellmap_resort(CellMapDev &m, int total, int n_appended, hipStream_t s, const char **err)
{
    const int n_cells_old = m.n_cells;
    if (total > 0) {
        hipLaunchKernelGGL(cm_iota_kernel, dim3(blocks(total)), dim3(256), 0, s, m.val, total);
        size_t tb = m.tmp_bytes;
        CMCHK(hipcub::DeviceRadixSort::SortPairs(m.tmp, tb, m.pkey, m.pkey2, m.val, m.val2, total, 0, 64, s));
        CMCHK(hipMemsetAsync(m.counts, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(cm_gather_kernel, dim3(blocks(total)), dim3(256), 0, s, m.pts, m.pkey2, m.val2, total, m.pts2, m.flag, m.counts);
        tb = m.tmp_bytes;
        CMCHK(hipcub::DeviceScan::ExclusiveSum(m.tmp, tb, m.flag, m.rank, total, s));
        hipLaunchKernelGGL(cm_cells_kernel, dim3(blocks(total)), dim3(256), 0, s, m.pkey2, m.flag, m.rank, m.counts, m.ckey2, m.cstart2, m.counts);
    } else {
        CMCHK(hipMemsetAsync(m.counts, 0, 2 * sizeof(int), s));
    }
    int h[2] = {0, 0};
    CMCHK(hipMemcpyAsync(h, m.counts, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    CMCHK(hipStreamSynchronize(s));
    const int n_valid = h[0], n_cells = h[1];
    if (n_cells > 0) {
        hipLaunchKernelGGL(cm_carry_kernel, dim3(blocks(n_cells)), dim3(256), 0, s, m.ckey2, n_cells, m.ckey, m.clast, n_cells_old, m.frame, m.clast2);
        if (n_appended > 0)  // the appended points sit at [total - n_appended, total) of the unsorted arrays
            hipLaunchKernelGGL(cm_touch_kernel, dim3(blocks(n_appended)), dim3(256), 0, s, m.pkey + (total - n_appended), n_appended, m.ckey2,
                               n_cells, m.frame, m.clast2);
    }
    CMCHK(hipGetLastError());
    CMCHK(hipStreamSynchronize(s));
    std::swap(m.pts, m.pts2);
    std::swap(m.pkey, m.pkey2);
    std::swap(m.ckey, m.ckey2);
    std::swap(m.cstart, m.cstart2);
    std::swap(m.clast, m.clast2);
    m.n_pts = n_valid;
    m.n_cells = n_cells;
    return 0;
}

// Per cell of the table (after the append that brought n_appended points): how many of those points it received -> m.csel
// (the query's selection flags: free between queries).  The appended points' keys still sit behind the old ones in the unsorted
// key array, which the re-sort left in pkey2.
int cellmap_touch_counts(CellMapDev &m, int n_before, int n_appended, hipStream_t s, const char **err)
{
    if (m.n_cells <= 0) return 0;
    CMCHK(hipMemsetAsync(m.csel, 0, (size_t)m.n_cells * sizeof(u32), s));
    if (n_appended > 0)
        hipLaunchKernelGGL(cm_touch_count_kernel, dim3(blocks(n_appended)), dim3(256), 0, s, m.pkey2 + n_before, n_appended, m.ckey, m.n_cells, m.csel);
    CMCHK(hipGetLastError());
    return 0;
}

int cellmap_append(CellMapDev &m, const float4 *d_src, int n, hipStream_t s, const char **err)
{
    if (n < 0 || (long long)m.n_pts + n > m.cap) {
        *err = "cell map is full (max_points)";
        return -1;
    }
    const bool was_empty = m.n_cells == 0;
    if (n > 0) {
        if (m.n_cells > 0) CMCHK(hipMemsetAsync(m.csel, 0, (size_t)m.n_cells * sizeof(u32), s));
        hipLaunchKernelGGL(cm_new_points_kernel, dim3(blocks(n)), dim3(256), 0, s, d_src, n, m.n_pts, m.geom, m.ckey, m.clast, m.n_cells, m.frame,
                           m.revisit_threshold, m.pts, m.pkey, m.csel);
        if (m.n_cells > 0 && m.n_pts > 0)
            hipLaunchKernelGGL(cm_drop_reset_kernel, dim3(blocks(m.n_pts)), dim3(256), 0, s, m.n_pts, m.ckey, m.n_cells, m.csel, m.pkey);
        if (cellmap_resort(m, m.n_pts + n, n, s, err)) return -1;
    }
    // m_current_frame_idx++ (CMK:667) -- and once more when the map was empty at the call: append_cloud then goes through
    // set_point_cloud, which increments it too (CMK:615; pinned against the reference's own class, tests/test_ref_cells.py)
    m.frame += was_empty ? 2 : 1;
    return 0;
}

int cellmap_query_filter(CellMapDev &m, const double *d_pose, float radius, float max_fov_deg, float leaf, int replace, hipStream_t s,
                         const char **err)
{
    m.n_filt = m.n_sel = 0;
    if (!(leaf > 0.f)) {
        *err = "leaf size must be positive";
        return -1;
    }
    const float inv_leaf = 1.0f / leaf;
    if (!(cell_leaf_span(m.geom, inv_leaf) < 1024.0f)) {
        *err = "leaf size too small for the cell size (more than 1020 leaves across one cell)";
        return -1;
    }
    if (m.n_cells == 0 || m.n_pts == 0) return 0;
    const int nc = m.n_cells, np = m.n_pts;
    hipLaunchKernelGGL(cm_select_kernel, dim3(blocks(nc)), dim3(256), 0, s, m.ckey, nc, m.geom, d_pose, radius, (double)max_fov_deg, m.csel);
    size_t tb = m.tmp_bytes;
    CMCHK(hipcub::DeviceScan::ExclusiveSum(m.tmp, tb, m.csel, m.csel_rank, nc, s));
    hipLaunchKernelGGL(cm_count_sel_kernel, dim3(1), dim3(64), 0, s, m.csel, m.csel_rank, nc, m.counts);
    hipLaunchKernelGGL(cm_leaf_key_kernel, dim3(blocks(np)), dim3(256), 0, s, m.pts, m.pkey, np, m.ckey, nc, m.csel, m.csel_rank, m.geom, inv_leaf,
                       m.skey, m.val);
    tb = m.tmp_bytes;
    CMCHK(hipcub::DeviceRadixSort::SortPairs(m.tmp, tb, m.skey, m.skey2, m.val, m.val2, np, 0, 64, s));
    hipLaunchKernelGGL(cm_leaf_head_kernel, dim3(blocks(np)), dim3(256), 0, s, m.skey2, np, m.flag);
    tb = m.tmp_bytes;
    CMCHK(hipcub::DeviceScan::ExclusiveSum(m.tmp, tb, m.flag, m.rank, np, s));
    hipLaunchKernelGGL(cm_leaf_pos_kernel, dim3(blocks(np)), dim3(256), 0, s, m.flag, m.rank, np, m.head_pos, m.counts);
    hipLaunchKernelGGL(cm_centroid_kernel, dim3(blocks(np)), dim3(256), 0, s, m.pts, m.pkey, m.skey2, m.val2, m.head_pos, m.counts, np, m.filt,
                       m.filt_key);
    int h[4] = {0, 0, 0, 0};
    CMCHK(hipMemcpyAsync(h, m.counts, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    CMCHK(hipStreamSynchronize(s));
    CMCHK(hipGetLastError());
    m.n_sel = h[2];
    m.n_filt = h[3];
    if (replace && m.n_filt > 0) {
        const int total = np + m.n_filt;  // n_filt <= np <= cap: fits the 2 * cap arrays
        hipLaunchKernelGGL(cm_replace_kernel, dim3(blocks(total)), dim3(256), 0, s, np, m.ckey, nc, m.csel, m.filt, m.filt_key, m.n_filt, m.pts,
                           m.pkey);
        if (cellmap_resort(m, total, 0, s, err)) return -1;
    }
    return 0;
}

}  // namespace ll
