// ll_device.h -- device-side data layout shared by the kernel files and the C-ABI host code (ll_api.hip).
// All buffers are HBM-resident SoA planes; [scan][point] with a fixed per-scan stride so one launch covers a
// whole batch of scans.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ll_fe_core.h"
#include "ll_knn_core.h"

namespace ll {

#define LL_MAX_PIECES 8
#define LL_GRP 8             // workgroups per scan of the grouped solver
#define LL_GRP_MIN_BLOCKS 6000  // ... whose largest scan has at least this many features (below, one CU's LDS cache holds most of a scan)
#define LL_GRP_MAX_SCANS 16  // batches up to this size use it (LL_GRP * LL_GRP_MAX_SCANS workgroups stay below the CU count)
// One wavefront per query (ll_knn_coop.h) where a launch is bound by its longest single-lane search chain:
#define LL_KNN_COOP_MAX_QUERIES 8192  // ll_map_knn5 batches up to this size; corner searches of a late ICP iteration's work lists
#define LL_KNN_COOP_MAX_SCANS 16      // all corner queries of ICP iterations 0 / 1 for batches up to this size ...
#define LL_KNN_COOP_MAX_SURF 2048     // ... and the surface queries of scans with up to this many (voxel-filtered clouds)
// Tile search of the surface queries (ll_knn_tile.h, ll_knn_kernels.hip): queries sorted by map cell once per registration, one
// wavefront per 64 of them against the LDS-staged points of their cells' common neighbourhood
#define LL_KNN_TILE_MIN_SURF 1024   // batches whose largest scan has at least this many surface queries (below: the wavefront-per-query search)
#define LL_KNN_TILE_SEG 24576       // queries one sorting workgroup orders (1024 threads x 24); a multiple of the tile kernel's workgroup
#define LL_KNN_TILE_MAX_SURF (4 * LL_KNN_TILE_SEG)  // ... and at most this many: larger scans are sorted in segments (Mid-100: three heads)

// Small scans (voxel-filtered feature clouds: a few hundred residual blocks) have a solver of their own (ll_reg_small_kernels.hip)
#define LL_SMALL_MAX_BLOCKS 2048     // batches whose largest scan has at most this many corner + surface queries (more than 1024: eight wavefronts per scan)
#define LL_SMALL_W1_MIN_SCANS 512    // ... one or two wavefronts per scan from this batch size on (many scans per CU), four wavefronts below
#define LL_SMALL_ORDER_MIN_SCANS 512  // batches of this many scans or more start their longest scans first (reg_solve_order_kernel) ...
#define LL_SMALL_ORDER_MAX_SCANS 8192 // ... up to this many (one ordering workgroup)

struct FeScanInfo {
    int n_split;         // entries in split_idx (incl. the closing n-1)
    int clutter_size;    // return value of projection_scan_3d_2d (LFE:606; 0 when fewer than 6 entries)
    int n_petal_clouds;  // laserCloudScans.size() after split_laser_scan (LFE:718)
    int pad;
    float piece_start[LL_MAX_PIECES], piece_end[LL_MAX_PIECES];  // LFX:321-322
};

struct FeDev {
    // inputs
    const float4 *xyzi;   // [B][stride] raw points (x,y,z,intensity)
    const int *npts;      // [B]
    const double *time0;  // [B] m_current_time of each scan
    int stride;           // points per scan slot
    // per-point planes [B][stride]
    int *type, *label;
    float *depth2, *polar2, *curv, *view, *tstamp, *polar_angle;
    float2 *img;
    signed char *flags;   // bit0: defines own polar/img values, bit1: reached the split logic
    int *cand;            // split candidates scratch
    // per-scan split bookkeeping [B][split_cap]
    int split_cap;
    int *split_idx, *petal_first, *petal_last;
    float *run_angle;
    FeScanInfo *info;     // [B]
    // selection [B][stride]
    int *corner_idx, *surf_idx, *full_idx;
    float4 *corner_feat, *surf_feat;
    int *n_corner, *n_surf, *n_full;  // [B]
    // view-angle ambiguity list
    int *n_ambig;
    int2 *ambig_list;
    int ambig_cap;
};

void launch_fe_point(const FeDev &fb, const FeConst &fc, int n_scans, int max_n, hipStream_t s);
void launch_fe_split(const FeDev &fb, int pieces, int n_scans, hipStream_t s);
void launch_fe_select(const FeDev &fb, int n_scans, int piece, float min_blur, float max_blur, hipStream_t s);

// ------------------------------------------------------------------------------------------------ map grid

struct MapKind {
    f4 *pts = nullptr;          // sorted by cell, w = original index bits (nullptr for an fp16-point map)
    int *cell_start = nullptr;  // [ncell + 1]
    unsigned long long *pts16 = nullptr;  // fp16-point records (ll_knn_core.h, Grid::pts16), same order as pts
    int *perm = nullptr;        // original index of every fp16 record
    int64_t n = 0;              // points given
    int64_t n_valid = 0;        // finite points stored in the grid
    Grid grid{};                // device pointers + geometry
    size_t ncell = 0;
    // Buffers are kept between builds and only grow: the mapping loop rebuilds its match buffer every frame
    // (laser_mapping.hpp:539-546), and a hipMalloc / hipFree pair per temporary costs more than the build itself.
    size_t cap_pts = 0, cap_cells = 0;       // capacity of pts / cell_start (elements)
    size_t cap_n = 0, cap_tmp = 0;           // capacity of the per-point scratch (elements) and of b_tmp (bytes)
    unsigned int *b_keys = nullptr, *b_keys2 = nullptr;
    int *b_vals = nullptr, *b_vals2 = nullptr, *b_counts = nullptr;   // b_counts shares cap_cells
    void *b_tmp = nullptr;
    float *b_mm = nullptr;
};

// builds the grid for `n` device-resident raw points (stride floats apart); fills mk. Returns 0 or a HIP error.
int map_build(MapKind &mk, const float *d_raw, int stride, int64_t n, float cell, hipStream_t s, const char **err);
void map_free(MapKind &mk);
// replaces the fp32 records of a built map by fp16-point records (BASELINE config C5); returns 0 or -1 with *err set
int map_to_f16(MapKind &mk, hipStream_t s, const char **err);
// dequantised coordinates of an fp16-point map in ORIGINAL index order (d_out: n_valid... n points x 3 floats, NaN where dropped)
int map_f16_dequant(const MapKind &mk, float *d_out_xyz, hipStream_t s, const char **err);
void launch_knn5(const Grid &g, const float *d_q, int nq, float max_d2, int *d_idx, float *d_d2, hipStream_t s);

// ------------------------------------------------------------------------------------------------ registrar

// Per-scan solver state, resident in HBM between the kernels of one registration.
struct RegState {
    double pose_last[7], pose_curr[7], inc[7];
    double prev_q[4], prev_t[3];          // q_last_optimize / t_last_optimize (PCR:204-205,529-530)
    double interp_theta, hat[9], hat_sq[9];  // m_interpolatation_* (PCR:58,66-68)
    double inlier_thr, final_cost, initial_cost, angular_diff, t_diff;
    int icp_iters, n_blocks_last, corner_avail, surf_avail, lm_total;
    int done, accepted, gated, result;
    int last_work;  // small solver: cost evaluations of this scan's last solver launch (the next launch starts the longest scans first)
    int aborted;  // the grouped solver gave up on a barrier (bounded spin): the scan is rejected and ll_reg_collect reports it
    long long dbg_cycles[16];  // LL_SOLVE_TIMING builds (shader clocks): eval, LM controller, L1, dedupe, select, total, census (+ triple inserts), prune,
                               // epilogue (plane-table path: table build), [9] = exchanges / L1 shortcuts taken; plane-table path: [10] census load waits,
                               // [11] inserts, [12] block sums, [13] id compaction, [14] plane constants, [15] id pass + LDS fill
};

struct RegConst {
    int if_motion_deblur, icp_max_iterations, ceres_max_iterations, ceres_prerun_times;
    int icp_line, icp_plane, force_all_iterations, debug_knn;
    int debug_knn_iter;  // the ICP iteration whose neighbour lists the debug taps record (ll_reg_set_debug_knn_iteration; default 0)
    int force_general;   // test switch: run the HBM-resident solver path even for small scans
    int knn_reuse;       // exact neighbour reuse across ICP iterations (ll_knn_core.h)
    int knn_reuse_from;  // first ICP iteration that tries it (iteration 1 usually moves the queries too far)
    int check_line_pca, check_plane_pca;  // K7 (PCR:46,48)
    int solve_group;     // workgroups per scan of the compact solver (1, or LL_GRP for small batches: ll_reg_kernels.hip, group_*)
    int xch_epoch;       // ... number of this solver launch within its registration, from 1 (tags of the exchange granules, group_reduce)
    int test_group_abort; // test switch: the grouped solver behaves as if its first barrier had timed out
    int knn_coop;        // corner searches by whole wavefronts where a launch has few of them (ll_knn_coop.h); 0 = A/B switch off
    int knn_tile_last_sort;  // the tile search re-sorts a scan's queries by map cell in ICP iterations 0 .. this one (1: after the first pose update too)
    int no_solve_order;  // A/B switch: the small solver's workgroups in scan order, not longest first (ll_reg_set_debug bit 18)
    int no_small_solver; // A/B switch: small scans take the 512-thread solver too (ll_reg_set_debug bit 15)
    int small_waves;     // test switch: wavefronts per scan of the small solver whatever the batch size (0 = by batch size; bits 16 / 17: 1 / 4)
    int no_line_cache;   // A/B switch: the solver reads line blocks from HBM in every evaluation (no LDS copy)
    int knn_tile;        // surface searches by the tile kernel (ll_knn_tile.h): 0 = off (A/B), 1 = wherever the per-lane search of ALL surface
                         // queries would run (ICP iterations before knn_reuse_from, or every iteration without reuse), 2 = every ICP iteration
                         // (the reuse machinery is then off: a tile search of everything costs less than classifying + searching the lists)
    unsigned int subsample_seed;  // a13 (0 = off)
    int max_blocks;               // maximum_allow_residual_block
    float max_d2_line, max_d2_plane;      // compared against fp32 squared distances (PCR:254,353)
    double max_d2_line_d, max_d2_plane_d;
    double huber_a, inliner_dis, inlier_ratio, minimum_icp_R_diff, minimum_icp_T_diff;
    double bound;                          // m_para_max_speed (PCR:143-151)
    float para_max_angular_rate, max_final_cost, min_ts, max_ts;
};

struct RegDev {
    RegState *state;              // [B]
    const float4 *corner_feat;    // [B][feat_stride_c]
    const float4 *surf_feat;      // [B][feat_stride_s]
    const int *n_corner, *n_surf; // [B]
    int feat_stride_c, feat_stride_s;
    // residual blocks, slot layout per scan: [0, cap_c) corner queries, [cap_c, cap_c + cap_s) surface queries
    int cap_c, cap_s, cap;        // cap = cap_c + cap_s
    float4 *blk_f;                // [B][cap]  f.xyz (sensor frame), w = motion-blur ratio s
    double *blk_av;               // [B][6 * cap] per scan {a0, v0}[cap], {v1, v2}[cap], {a1, a2}[cap] (16-byte pairs; ll_reg_kernels.hip av_load), frame of pose_last
    // round-3 compact path (solve_fast3): the plane constants {n', c} are stored once per DISTINCT neighbour triple of a scan
    // (a scan's ~17 k plane blocks share 2.4 - 4.6 k triples), built by the solver itself at the start of every launch
    unsigned short *blk_id;       // [B][cap_s] plane id of every surface block (relative to its solver workgroup's table region)
    int4 *pl_tab;                 // [B][tab_cap][2] plane table: {n'.x, n'.y}, {n'.z, c = n'.a'} (fp64), frame of pose_last
    int tab_cap;                  // entries per scan: min(cap_s, 61440) rounded up to 4096 (LL_GRP regions of whole 512-thread rounds)
    float4 *qw;                   // [B][cap]  queries transformed into the map frame (K6t -> K6a)
    float4 *ref_q;                // [B][cap]  query position where the neighbour list was established, w = m_strong
    int4 *ref_p;                  // [B][cap]  its neighbours 0..3 (positions in the cell-sorted array)
    float2 *ref_s;                // [B][cap]  x = bits(neighbour 4, -1 when fewer than 5 inside the radius), y = m_set
    int *solve_order;             // [B] small solver: the scans in the order their workgroups start (reg_solve_order_kernel)
    int *grp_ctl;                 // [1 + 2 B] grouped solver: [0] ticket counter, [1 + 2 b] arrival counter of scan b's group barrier, [2 + 2 b] its abort word (zeroed per launch)
    double *grp_part;             // [B][2][LL_GRP][28] grouped solver: the workgroups' partial sums of the evaluation that also publishes the L1 values
    unsigned long long *grp_xch;  // [B][2][LL_GRP][56] ... of every other evaluation, as self-validating 8-byte granules {32-bit half, tag}; zeroed per registration
    unsigned char *blk_flag0;     // [B][cap]  block flag as built; the solver prunes a copy (LDS, or blk_flag in the general path)
    int *work_search;             // [B][cap]  slots that need a full search this iteration
    int *work_build;              // [B][cap]  slots that were re-sorted (block must be rebuilt)
    int *work_off;                // [3][2 B + 1] exclusive prefix sums of work_cnt per list (reg_list_offsets_kernel): searches, re-sorts, and
                                  // the searches of the corner segments alone (their surface segments counted as empty)
    int *work_cnt;                // [B][2 kinds][2] lengths of this ICP iteration's work lists (search, re-sorted) per scan and kind:
                                  // work_search / work_build hold (scan * cap + slot) entries, dense inside the scan-and-kind's own
                                  // segment [scan * cap + kind * cap_c, ...); one atomic per re-query workgroup and list reserves a range
    int n_chunks;                 // chunks of 256 queries (RQ_THREADS) per (scan, kind): the re-query kernel's grid
    int4 *nn;                     // [B][cap]  neighbour positions (cell-sorted order) + found flag (K6a -> K6b)
    unsigned short *qperm;        // [B][cap_s] surface queries of a scan ordered by the map cell they fall into at ICP iteration 0 (reg_qsort_kernel)
    unsigned short *qperm_c;      // [B][LL_QSORT_CORNER_MAX] a scan's corner queries in the order of the corner map's cells (scans with more corner
                                  // queries than that keep the feature order)
    float4 *qsorted;              // [B][cap_s] the surface FEATURES (sensor frame) in that order, written with qperm when the tile kernel transforms
                                  // the queries itself (no motion deblur): its loads are then coalesced and independent of each other
    unsigned char *blk_flag;      // [B][cap]  BLK_* bits
    double *blk_l1;               // [B][cap]  scratch for the inlier threshold
    unsigned long long *hash;     // [B][hash_cap] dedup table for the std::set semantics of PCR:155-160
    int hash_cap;
    // debug taps (first ICP iteration)
    int *dbg_idx;                 // [B][cap][5]
    float *dbg_d2;                // [B][cap][5]
};

void launch_reg_knn_build(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int n_scans, int iter,
                          int max_nc, int max_ns, hipStream_t s);
#define LL_QSORT_CORNER_MAX 2048
void launch_reg_qsort(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int n_scans, int max_nc, int max_ns, bool fused, hipStream_t s);
void launch_debug_quintic(const double *args, int n, double *out_seq, double *out_wave, hipStream_t s);
void launch_reg_knn_tile(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int n_scans, int iter, int max_nc, int max_ns,
                         bool fused, hipStream_t s);
void launch_reg_solve(const RegDev &rd, const RegConst &rc, const Grid &gs, int n_scans, int max_nc, int max_ns, int iter, hipStream_t s);
bool reg_solve_small_eligible(const RegConst &rc, int max_nc, int max_ns);
bool reg_solve_fast_eligible(const RegConst &rc, int max_nc, int max_ns);
void launch_reg_solve_small(const RegDev &rd, const RegConst &rc, const Grid &gs, int n_scans, int max_nc, int max_ns, int iter, hipStream_t s);
int reg_solve_small_waves(const RegConst &rc, int n_scans, int max_nc, int max_ns);
void launch_reg_finalize(const RegDev &rd, const RegConst &rc, int n_scans, hipStream_t s);
void launch_cloud_transform(const float4 *in, float4 *out, int n, const double *d_pose, hipStream_t s);
#define LL_HIST_CONCAT_MAX 1023  // frames one history_concat_kernel launch gathers (longer histories: the per-frame copies)
void launch_history_concat(const float4 *frames, const int2 *d_table, int n_seg, int total, float4 *out, hipStream_t s);
void launch_reg_merge_heads(const float4 *fe_corner, const float4 *fe_surf, const int *fe_nc, const int *fe_ns, int fe_stride, int heads,
                            float4 *dst_corner, float4 *dst_surf, int *dst_nc, int *dst_ns, int dst_stride, int n_scans, hipStream_t s);

}  // namespace ll
