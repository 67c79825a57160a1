/*
 * loam_livox_hip.h -- C ABI of libloamlivox_hip.so: MI355X (gfx950) scan-to-map registration core for
 * Loam-Livox.  This is the drop-in boundary for the hot path of hku-mars/loam_livox; every entry point
 * cites the reference interface it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C types only; the caller owns every host buffer, the library owns all device memory;
 *   - pose[7] = {qx,qy,qz,qw,tx,ty,tz}: the storage order of m_para_buffer_RT / m_para_buffer_incremental
 *     (source/point_cloud_registration.hpp:51-56, source/laser_mapping.hpp:206-213);
 *   - clouds are AoS float xyzi (4 floats per point) like pcl::PointXYZI payloads; feature clouds carry the
 *     per-point time stamp in the intensity slot (source/livox_feature_extractor.hpp:246,255);
 *   - return value: >= 0 success (meaning per function), < 0 library error (text via ll_last_error()).
 *     No exceptions, no abort.  There is NO CPU fallback: without a HIP device every *_create fails.
 *   - handles are not thread-safe individually (the reference serialises the extractor with a mutex,
 *     laser_feature_extractor.hpp:244); different handles may be used from different threads, and one
 *     ll_map may be shared read-only by several ll_reg (laser_mapping.hpp:1737-1742 runs several
 *     process_new_scan concurrently against one map snapshot).
 */
#ifndef LOAM_LIVOX_HIP_H
#define LOAM_LIVOX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ll_fe ll_fe;   /* replaces class Livox_laser                (livox_feature_extractor.hpp:77)  */
typedef struct ll_map ll_map; /* replaces the map clouds + KdTreeFLANN pair (point_cloud_registration.hpp:72-73,
                                 laser_mapping.hpp:539-546)                                               */
typedef struct ll_reg ll_reg; /* replaces class Point_cloud_registration    (point_cloud_registration.hpp:38) */

/* Point-type bit masks (livox_feature_extractor.hpp:82-92) and feature labels (:94-103). */
enum { LL_PT_NORMAL = 0, LL_PT_000 = 1, LL_PT_TOO_NEAR = 2, LL_PT_REFL_LOW = 4, LL_PT_REFL_HIGH = 8,
       LL_PT_CIRCLE_EDGE = 16, LL_PT_NAN = 32, LL_PT_SMALL_VIEW_ANGLE = 64 };
enum { LL_LABEL_INVALID = -1, LL_LABEL_UNLABELED = 0, LL_LABEL_CORNER = 1, LL_LABEL_SURFACE = 2,
       LL_LABEL_NEAR_NAN = 4, LL_LABEL_NEAR_ZERO = 8, LL_LABEL_HIGH_INTENSITY = 16 };

/* ------------------------------------------------------------------------------------------------ extractor */

/* Public tunables of Livox_laser (livox_feature_extractor.hpp:143-167) as the node sets them
 * (laser_feature_extractor.hpp:146-154,854,859), plus capacities. */
typedef struct {
    float thr_corner_curvature;  /* :153, ROS feature_extraction/corner_curvature   */
    float thr_surface_curvature; /* :154, ROS feature_extraction/surface_curvature  */
    float minimum_view_angle;    /* :155, ROS feature_extraction/minimum_view_angle */
    float livox_min_allow_dis;   /* :166, ROS feature_extraction/livox_min_dis      */
    float livox_min_sigma;       /* :167, ROS feature_extraction/livox_min_sigma    */
    float max_fov;               /* :143 (17 deg) */
    float time_internal_pts;     /* :145 (1e-5 s) */
    int32_t device;              /* HIP device ordinal */
    int32_t max_points;          /* capacity: points per scan */
    int32_t max_scans;           /* capacity: scans resident at once (1 for the per-message node path) */
    int32_t piecewise_number;    /* ROS common/piecewise_number (laser_feature_extractor.hpp:142); 1 when deblur */
} ll_fe_params;

void ll_fe_default_params(ll_fe_params *p); /* node defaults, max_points 24000, max_scans 1 */
int ll_fe_create(const ll_fe_params *p, ll_fe **out);
void ll_fe_destroy(ll_fe *h);

/* Livox_laser::extract_laser_features(cloud, time_stamp) (livox_feature_extractor.hpp:722-766) for ONE scan
 * given on the host: projection_scan_3d_2d (:458), eval_point (:343), add_mask_of_point (:322),
 * compute_features (:361), petal split + split_laser_scan (:657).  Keeps the sequential time base of the
 * class (:150-152,724-736).  Synchronous.  *n_petal_clouds = laserCloudScans.size() of the reference
 * (the caller drops the scan when it is <= 5, laser_feature_extractor.hpp:287).  Returns 0. */
int ll_fe_extract(ll_fe *h, const float *xyzi, int32_t n, double time_stamp, int32_t *n_petal_clouds);

/* m_pts_info_vec fields of scan slot `scan` (livox_feature_extractor.hpp:118-133). Any pointer may be NULL. */
int ll_fe_labels(ll_fe *h, int32_t scan, int32_t *pt_type, int32_t *pt_label, float *depth_sq2,
                 float *polar_dis_sq2, float *curvature, float *view_angle, float *time_stamp, float *polar_angle);

/* split bookkeeping of scan slot `scan`: split_idx (livox_feature_extractor.hpp:465-565; capacity
 * max_points/50+8), the return value of projection_scan_3d_2d (:606), and for every surviving petal cloud the
 * index of its first/last point, plus the piece-wise windows of laser_feature_extractor.hpp:305-323
 * (piecewise_number entries; boundary points go through find_pt_info's first-occurrence lookup, :321-322). */
int ll_fe_splits(ll_fe *h, int32_t scan, int32_t *split_idx, int32_t *n_split, int32_t *clutter_size,
                 int32_t *n_petal_clouds, int32_t *first_idx, int32_t *last_idx, float *piece_start, float *piece_end);

/* Livox_laser::get_features(corners, surface, full, minimum_blur, maximum_blur)
 * (livox_feature_extractor.hpp:219-272) on scan slot 0: ascending index lists (the integer parity artefact)
 * and, optionally, the packed clouds (xyz + time stamp).  Output buffers sized max_points; NULL to skip. */
int ll_fe_select(ll_fe *h, float minimum_blur, float maximum_blur, int32_t *corner_idx, int32_t *n_corner,
                 int32_t *surf_idx, int32_t *n_surf, int32_t *full_idx, int32_t *n_full, float *corner_xyzi,
                 float *surf_xyzi);

/* Batched, device-resident form (offline map building / throughput): scans are independent, so the time base
 * is given explicitly per scan (current_time = m_current_time of livox_feature_extractor.hpp:727-731). */
int ll_fe_upload(ll_fe *h, int32_t first_scan, int32_t n_scans, const float *xyzi, int32_t n_points,
                 const double *current_time);
/* The same without the final wait: the copies are queued on the handle's stream and the call returns; xyzi and
 * current_time must stay valid (and should be page-locked host memory for a true asynchronous copy) until ll_fe_sync or
 * a later synchronous call on this handle.  Lets the next batch cross PCIe while the previous one is being processed
 * on another handle (bench.py "streamed" figure). */
int ll_fe_upload_async(ll_fe *h, int32_t first_scan, int32_t n_scans, const float *xyzi, int32_t n_points,
                       const double *current_time);
int ll_fe_extract_batch(ll_fe *h, int32_t n_scans); /* asynchronous on the handle's stream */
/* piece >= 0: use the device-computed piece-wise window `piece`; piece < 0: explicit [minimum,maximum]_blur */
int ll_fe_select_batch(ll_fe *h, int32_t n_scans, int32_t piece, float minimum_blur, float maximum_blur);
/* The selection ll_fe_select_batch left in slot `scan`: what ll_fe_select returns for slot 0 (get_features' outputs,
 * livox_feature_extractor.hpp:219-272), for any slot of a batch.  Output buffers sized max_points; NULL to skip.  Synchronises. */
int ll_fe_selection(ll_fe *h, int32_t scan, int32_t *corner_idx, int32_t *n_corner, int32_t *surf_idx, int32_t *n_surf,
                    int32_t *full_idx, int32_t *n_full, float *corner_xyzi, float *surf_xyzi);
int ll_fe_counts(ll_fe *h, int32_t n_scans, int32_t *n_corner, int32_t *n_surf, int32_t *n_full,
                 int32_t *n_ambiguous); /* synchronises */
int ll_fe_sync(ll_fe *h);
/* Labels depend on `view_angle > minimum_view_angle` where view_angle comes from the C library's acosf
 * (tools_eigen_math.hpp:38).  Device and host acosf can differ in the last ulp, so the kernels flag the (very
 * rare) points whose angle falls inside a few-ulp band around the threshold; this call re-derives those labels
 * with the host libm (what the reference executable itself would have computed on this machine) and patches
 * them on the device.  Synchronises; returns the number of flagged points.  ll_fe_extract calls it itself;
 * batch users call it between ll_fe_extract_batch and ll_fe_select_batch. */
int ll_fe_resolve(ll_fe *h);

/* ------------------------------------------------------------------------------------------------ map */

enum { LL_MAP_CORNER = 0, LL_MAP_SURF = 1 };

int ll_map_create(int32_t device, ll_map **out);
void ll_map_destroy(ll_map *m);
/* Replaces pcl::KdTreeFLANN::setInputCloud for the match buffers (laser_mapping.hpp:544-545,
 * point_cloud_registration.hpp:596-597): uploads the cloud and builds the device search grid.
 * xyz: m points, stride_floats apart (3 = xyz, 4 = xyzi).  cell_size <= 0 selects the default
 * (1.45 m corner: the sparse edge map is searched out to the sqrt(2) m line radius, one cell ring covers it;
 * 0.6 m surface: about 1.5x the 0.4 m surface-map leaf).  The results do not depend on it, the speed does: a search reads
 * every point of the query's own cells before it can prune, so a cloud that was voxel-filtered at leaf l wants cells of
 * about 3-4 l (ll_history_refresh picks that itself); a 1.45 m cell on an edge map with 0.1 m spacing costs 10x.
 * Point indices reported by the library refer to this input order. */
int ll_map_upload(ll_map *m, int32_t kind, const float *xyz, int32_t stride_floats, int64_t n, float cell_size);
int64_t ll_map_size(const ll_map *m, int32_t kind);

/* Number of search structures published for `kind` so far (by ll_map_upload, ll_map_to_f16, ll_history_refresh*).  A host-side
 * cache of "what did I upload last" (include/loam_livox_adapter.hpp keys the map clouds of find_out_incremental_transfrom,
 * point_cloud_registration.hpp:163-168, by their contents) stays valid only while this number is the one it saw after its own
 * upload; -1 for a bad argument. */
int64_t ll_map_generation(const ll_map *m, int32_t kind);
/* cells of the published grid of `kind` (the cell table is 4 bytes per cell): with ll_map_size() the map's footprint in HBM */
int64_t ll_map_cells(const ll_map *m, int32_t kind);
/* ll_map_upload that also reports the generation number ITS publication got (taken under the map's lock): a cache keyed by
 * "generation after my upload" must use this one -- reading ll_map_generation after ll_map_upload returns would pick up a
 * publication another thread (ll_history_refresh*) made in between and later skip a required re-upload. */
int ll_map_upload_gen(ll_map *m, int32_t kind, const float *xyz, int32_t stride_floats, int64_t n, float cell_size, int64_t *generation);
/* BASELINE config C5 ("fp16 points / fp32 accumulate k-NN"): replaces the 16-byte fp32 records of an uploaded map kind by
 * 8-byte records -- the point's position inside its grid cell in binary16 (<= 2^-11 cell sizes off) + the cell index
 * bits -- and ll_map_knn5 then returns the exact 5-NN of that dequantised cloud, distances accumulated in fp32.  The
 * registrar refuses such a map (its parity contract is fp32 points, like pcl::KdTreeFLANN<PointXYZI>).
 * ll_map_dequantized: the cloud the fp16 records stand for, in upload order (NaN rows for dropped points). */
int ll_map_to_f16(ll_map *m, int32_t kind);
int ll_map_dequantized(ll_map *m, int32_t kind, float *xyz, int64_t capacity_points);

/* pcl::KdTreeFLANN::nearestKSearch(pt, 5, idx, sq_dis) (point_cloud_registration.hpp:249,351) for a batch of
 * host query points: exact 5-NN among map points with squared distance < max_sq_dis, ascending (d2, idx);
 * missing entries are idx -1 / d2 +inf.  Returns 0. */
int ll_map_knn5(ll_map *m, int32_t kind, const float *queries_xyz, int32_t n_queries, float max_sq_dis,
                int32_t *idx5, float *sq_dis5);
/* The same search with the queries and both result arrays resident on the device (hipMalloc'd / a framework tensor's pointer): nothing
 * crosses PCIe.  *kernel_ms (may be NULL): duration of the search kernel from HIP events on the map's stream.  Synchronous. */
int ll_map_knn5_device(ll_map *m, int32_t kind, const float *dev_queries_xyz, int64_t n_queries, float max_sq_dis, int32_t *dev_idx5,
                       float *dev_sq_dis5, float *kernel_ms);

/* ------------------------------------------------------------------------------------------------ registrar */

/* Public configuration fields of Point_cloud_registration set by Laser_mapping::init_pointcloud_registration
 * (laser_mapping.hpp:1266-1297) and class defaults (point_cloud_registration.hpp:45-103). */
typedef struct {
    int32_t if_motion_deblur;               /* :60  */
    int32_t icp_max_iterations;             /* :89  m_para_icp_max_iterations   */
    int32_t ceres_max_iterations;           /* :90  m_para_cere_max_iterations  */
    int32_t ceres_prerun_times;             /* :91  (2) */
    int32_t icp_line;                       /* :50  ICP_LINE  */
    int32_t icp_plane;                      /* :49  ICP_PLANE */
    int32_t current_frame_index;            /* :83  */
    int32_t mapping_init_accumulate_frames; /* :84  */
    int32_t maximum_allow_residual_block;   /* :103; random sub-sampling (:232-238,339-345,438-458) is only
                                               engaged when feature counts exceed it; must be >= feature count
                                               for deterministic parity runs */
    int32_t force_all_iterations;           /* harness switch: ignore the :521-526 convergence break */
    double maximum_dis_line_for_match;      /* :65 (2.0)  squared distance */
    double maximum_dis_plane_for_match;     /* :64 (50.0) squared distance */
    double huber_a;                         /* :220 ceres::HuberLoss(0.1) */
    double inliner_dis;                     /* :97  */
    double inlier_ratio;                    /* :98  */
    double minimum_icp_R_diff;              /* :94  */
    double minimum_icp_T_diff;              /* :95  */
    float para_max_angular_rate;            /* :86  (ROS optimization/max_allow_incre_R) */
    float para_max_speed;                   /* :87  (ROS optimization/max_allow_incre_T): box bound on t_incre */
    float max_final_cost;                   /* :88  (ROS optimization/max_allow_final_cost) */
    float minimum_pt_time_stamp;            /* :92  */
    float maximum_pt_time_stamp;            /* :93  */
    int32_t if_line_feature_check;          /* :46  IF_LINE_FEATURE_CHECK  (0): PCA test of the 5 line neighbours, :259-292  */
    int32_t if_plane_feature_check;         /* :48  IF_PLANE_FEATURE_CHECK (0): PCA test of the 5 plane neighbours, :357-389;
                                               uses the SURFACE cloud (the reference indexes the corner cloud, a bug) */
    int32_t subsample_seed;                 /* a13: 0 = refuse scans with more features than maximum_allow_residual_block (strict
                                               parity mode); otherwise the reference's random sub-sampling (:232-238,339-345,
                                               438-458) with a reproducible counter-based stream seeded by this value          */
} ll_reg_params;

void ll_reg_default_params(ll_reg_params *p);

/* What callers read back from the object after the call (point_cloud_registration.hpp:62-63,100-102,555-559). */
typedef struct {
    double final_cost, initial_cost;  /* summary of the last ceres::Solve */
    double inlier_threshold;          /* m_inlier_threshold (:100,:559) */
    double angular_diff_deg, t_diff;  /* m_angular_diff, m_t_diff (:517-518) */
    int32_t icp_iterations;
    int32_t n_blocks_last;            /* summary.num_residual_blocks */
    int32_t corner_avail, surf_avail; /* :325,:425 */
    int32_t lm_iterations_total;
    int32_t accepted;                 /* return value of find_out_incremental_transfrom */
    int32_t gated;                    /* :199 gate skipped the optimisation */
    int32_t aborted;                  /* the small-batch solver gave up on this scan (a barrier between its workgroups timed out):
                                         the scan is rejected (accepted 0) and its pose restored */
} ll_reg_report;

int ll_reg_create(int32_t device, int32_t max_scans, int32_t max_features_per_scan, ll_reg **out);
void ll_reg_destroy(ll_reg *r);

/* int Point_cloud_registration::find_out_incremental_transfrom(map_corner, map_surf, kd_corner, kd_surf,
 * scan_corner, scan_surf) (point_cloud_registration.hpp:163-583; the 4-argument overload :585-605 is the same
 * call after ll_map_upload).  pose_last = m_q_w_last/m_t_w_last, pose_curr in = initial guess
 * (m_q_w_curr/m_t_w_curr) / out = result, pose_incre in/out = m_para_buffer_incremental (identity for a fresh
 * object, laser_mapping.hpp:1348).  Returns 1 (accepted, or skipped by the :199 gate), 0 (rejected: pose_curr
 * restored to pose_last, :561-573), < 0 library error. */
int ll_reg_solve(ll_reg *r, const ll_map *map, const float *scan_corner_xyzi, int32_t n_corner,
                 const float *scan_surf_xyzi, int32_t n_surf, const ll_reg_params *prm, const double pose_last[7],
                 double pose_curr[7], double pose_incre[7], ll_reg_report *rep);

/* Batched form: n_scans independent registrations against the same map snapshot (the reference's
 * maximum_parallel_thread model, laser_mapping.hpp:1737-1742).  Features are taken from the extractor's
 * device-resident selection (ll_fe_select_batch) without a host round trip.  poses_* are [n_scans][7],
 * reports [n_scans], results[n_scans] = per-scan return value.  Returns 0. */
int ll_reg_solve_batch_fe(ll_reg *r, const ll_map *map, ll_fe *fe, int32_t n_scans, const ll_reg_params *prm,
                          const double *poses_last, double *poses_curr, double *poses_incre, ll_reg_report *reports,
                          int32_t *results);
/* The two halves of ll_reg_solve_batch for callers that register the same host-provided feature clouds more than once
 * (or want the upload outside a timed region): upload the per-scan corner / surface clouds into the registrar's own
 * HBM buffers, then enqueue registrations of the first n_scans of them (collect with ll_reg_collect). */
int ll_reg_upload_features(ll_reg *r, int32_t n_scans, const float *corner_xyzi, const int32_t *n_corner, int32_t stride_corner,
                           const float *surf_xyzi, const int32_t *n_surf, int32_t stride_surf);
int ll_reg_enqueue_uploaded(ll_reg *r, const ll_map *map, int32_t n_scans, const ll_reg_params *prm, const double *poses_last,
                            const double *poses_curr, const double *poses_incre);

/* Same with host feature clouds: corner_xyzi [n_scans][stride_c][4] with counts n_corner[n_scans], etc. */
int ll_reg_solve_batch(ll_reg *r, const ll_map *map, int32_t n_scans, const float *corner_xyzi,
                       const int32_t *n_corner, int32_t stride_corner, const float *surf_xyzi, const int32_t *n_surf,
                       int32_t stride_surf, const ll_reg_params *prm, const double *poses_last, double *poses_curr,
                       double *poses_incre, ll_reg_report *reports, int32_t *results);

/* Asynchronous halves of ll_reg_solve_batch_fe for benchmarking with inputs resident in HBM:
 * _enqueue uploads only the poses (n_scans*7 doubles) and launches; _collect synchronises and downloads. */
/* Mid-100 / multi-lidar form (laser_feature_extractor.hpp:348-358: the clouds of all lidars are concatenated before they are
 * published): registrar scan b is the concatenation, in head order, of the selected features of extractor slots
 * b * heads ... b * heads + heads - 1 (corner clouds together, surface clouds together), merged device to device.  Every
 * head keeps its own time base (its slot's stamp); n_scans * heads <= the extractor's max_scans, the merged counts must fit
 * the registrar's max_features.  Followed by ll_reg_collect like the other _enqueue forms. */
int ll_reg_enqueue_fe_merged(ll_reg *r, const ll_map *map, ll_fe *fe, int32_t n_scans, int32_t heads, const ll_reg_params *prm,
                             const double *poses_last, const double *poses_curr, const double *poses_incre);
int ll_reg_enqueue_fe(ll_reg *r, const ll_map *map, ll_fe *fe, int32_t n_scans, const ll_reg_params *prm,
                      const double *poses_last, const double *poses_curr, const double *poses_incre);
/* Returns 0; a negative value on an error of the call; or the number (> 0) of scans whose registration was aborted on the device
 * (ll_reg_report.aborted) -- not an error: all outputs are filled in, the aborted scans are rejected with their pose restored, the
 * results of the other scans of the batch are valid, and ll_last_error() carries the reason. */
/* RETURN VALUE of ll_reg_collect and of the calls that end in it (ll_reg_solve, ll_reg_solve_batch, ll_reg_solve_fe ...; the adapter's
 * find_out_incremental_transfrom): < 0 an error; 0 every scan solved; > 0 the NUMBER OF SCANS whose solve was abandoned (a bounded
 * wait of the small-batch grouped solver ran out, or a launch that could not hold the scan): those scans come back rejected -- result 0,
 * report.aborted 1, pose restored -- the others are valid, and ll_last_error() says why.  Callers that treat any non-zero status as
 * failure should test `< 0`. */
int ll_reg_collect(ll_reg *r, int32_t n_scans, double *poses_curr, double *poses_incre, ll_reg_report *reports,
                   int32_t *results);

/* Test tap: the three-sample line-search fit (ll_reg_core.h lm_quintic_min_step: Ceres' InterpolatingPolynomialMinimizingStepSize with
 * three samples, restated) in its sequential form and in the wavefront form the solver kernel runs, on n argument sets
 * {f0, g0, x1, f1, g1, x2, f2, g2, lo, hi}: the two must agree to the bit. */
int ll_debug_quintic(int32_t device, const double *args10, int32_t n, double *out_sequential, double *out_wavefront);

/* Debug/parity taps of the first ICP iteration of scan slot `scan` after a solve: 5-NN indices/sq-distances
 * per corner / surface query (row = query, -1/inf when not found within the match radius). NULL to skip. */
int ll_reg_debug_knn(ll_reg *r, int32_t scan, int32_t *corner_idx5, float *corner_d25, int32_t *surf_idx5,
                     float *surf_d25);
/* enable: bit 0 = record the k-NN taps; bit 1 = force the general (HBM-resident) solver path that scans with
 * more than 24576 residual blocks use, for testing it on small inputs; bit 2 = disable the exact neighbour reuse
 * across ICP iterations (every iteration runs the full 5-NN search); bit 3 = try the reuse from ICP iteration 1
 * already (default: from iteration 2, the first pose update usually moves the queries too far); bit 4 = run the
 * round-1 solver fast path (49-byte fp64 plane blocks re-read on every evaluation) instead of the packed 48-byte
 * records with the LDS record cache -- an A/B switch, results agree to rounding; bit 5 = one workgroup per scan whatever
 * the batch size (by default batches of up to 16 scans give every scan a group of 8 workgroups whose cost evaluations
 * each cover an eighth of the blocks, resident in LDS; the sums are then grouped differently, so results agree with the
 * one-workgroup form to rounding, not bit for bit). */
int ll_reg_set_debug(ll_reg *r, int32_t enable);
/* Test tap: the ICP iteration (0-based, point_cloud_registration.hpp:211 `iterCount`) whose 5-NN lists ll_reg_debug_knn returns;
 * default 0.  A scan that has converged before that iteration keeps the lists of an earlier registration (or zeros). */
int ll_reg_set_debug_knn_iteration(ll_reg *r, int32_t icp_iteration);
/* Further A/B switches of ll_reg_set_debug (measurement and tests only): bit 8 = searches one per lane everywhere (no
 * wavefront-per-query search where the work is small).  Environment switches read by the library, same purpose:
 * LL_LIST_NO_LOCAL_OFFSETS (small batches launch the work-list offsets kernel like large ones), LL_VOXEL_GENERAL_PATH (read
 * when a voxel filter is created: every cloud through the multi-kernel pipeline instead of one workgroup per small cloud).
 * None of them changes a result bit.
 * Small scans (at most 1024 corner + surface queries in the largest scan of a batch: voxel-filtered clouds, laser_mapping.hpp:1367-1373)
 * take a solver of their own, one wavefront per scan in batches of 512 scans or more and four below (ll_reg_small_kernels.hip):
 * bit 15 = off (such scans on the 512-thread solver: A/B; results agree to rounding, counts exactly); bit 16 / bit 17 = one / four
 * wavefronts per scan whatever the batch size, both = two (tests); bit 18 = its workgroups in scan order instead of longest first (A/B). */

/* ------------------------------------------------------------------------------------------------------------
 * VoxelGrid  (SURVEY 8(f) row 1).  pcl::VoxelGrid<pcl::PointXYZI> as hku-mars/loam_livox uses it:
 *   m_voxel_filter_for_surface / _corner     laser_feature_extractor.hpp:192-193, 372-381
 *   m_down_sample_filter_corner / _surface   laser_mapping.hpp:742-743, 1367-1373, 1434-1437, 533-537
 * i.e. setLeafSize(l, l, l); setInputCloud(c); filter(out).  Semantics: PCL 1.9 VoxelGrid::applyFilter with its
 * defaults (centroid of x, y, z and intensity per leaf, float sums, output in ascending leaf index), made
 * deterministic: the points of a leaf are summed in input order (PCL's std::sort leaves that order open), and
 * non-finite points are always skipped.  One call filters n_clouds independent clouds.
 * status[b]: 0 = filtered; 1 = "leaf size is too small for the input dataset" -> the output is a copy of the input,
 * like PCL; 2 = no finite point -> empty output. */
typedef struct ll_voxel ll_voxel;
int ll_voxel_create(int32_t device, int32_t max_clouds, int32_t max_points_per_cloud, ll_voxel **out);
void ll_voxel_destroy(ll_voxel *v);
/* xyzi / out_xyzi: [n_clouds][stride_points][4] floats on the host; n_points / n_out / status: [n_clouds]. */
int ll_voxel_filter(ll_voxel *v, int32_t n_clouds, const float *xyzi, const int32_t *n_points, int32_t stride_points,
                    const float leaf[3], float *out_xyzi, int32_t *n_out, int32_t *status);

/* m_if_input_downsample_mode (laser_mapping.hpp:1367-1373): the corner / surface clouds selected by the extractor are
 * voxel-filtered on the device (leaf line_res / plane_res, laser_mapping.hpp:742-743) and registered, without leaving
 * HBM.  vox_corner / vox_surf need max_clouds >= n_scans and max_points_per_cloud >= the extractor's max_points.
 * Collect with ll_reg_collect. */
int ll_reg_enqueue_fe_downsampled(ll_reg *r, const ll_map *map, ll_fe *fe, ll_voxel *vox_corner, ll_voxel *vox_surf,
                                  float line_res, float plane_res, int32_t n_scans, const ll_reg_params *prm,
                                  const double *poses_last, const double *poses_curr, const double *poses_incre);
/* the filtered feature counts of the last ll_reg_enqueue_fe_downsampled (after ll_reg_collect) */
int ll_voxel_counts(ll_voxel *v, int32_t n_clouds, int32_t *n_out, int32_t *status);

/* ------------------------------------------------------------------------------------------------------------
 * Match buffer, history mode  (m_matching_mode == 0).  Device-resident stand-in for
 *   m_laser_cloud_corner_history / m_laser_cloud_surface_history      laser_mapping.hpp:1443-1478
 *   update_buff_for_matching(), history branch                        laser_mapping.hpp:517-546
 * ll_history_add = "Add new frame" (laser_mapping.hpp:1417-1478): the scan's corner / surface features (sensor frame)
 * are moved to the map frame with `pose` (pointAssociateToMap without undistortion, g_if_undistore = 0), voxel-filtered
 * (leaf line_res / plane_res) and pushed onto the two FIFO histories when
 *     history.size() < maximum_history_size  ||  t_diff > history_add_t_step  ||  r_diff > history_add_angle_step * 57.3
 * with t_diff / r_diff measured from the pose of the last frame that was pushed; the oldest frame is dropped when the
 * history is longer than maximum_history_size.  *added = 1 when the frame was pushed.
 * ll_history_refresh = update_buff_for_matching: concatenation of the history frames (oldest first) -> VoxelGrid
 * (line_res / plane_res) -> the two search structures of `map` (device grids instead of two k-d tree builds).
 * Everything stays in HBM; only the counts come back. */
typedef struct ll_history ll_history;
int ll_history_create(int32_t device, int32_t maximum_history_size, int32_t max_points_per_frame, float line_res, float plane_res,
                      ll_history **out);
void ll_history_destroy(ll_history *h);
int ll_history_add(ll_history *h, const float *corner_xyzi, int32_t n_corner, const float *surf_xyzi, int32_t n_surf,
                   const double pose[7], double history_add_t_step, double history_add_angle_step, int32_t *added);
/* same, taking the clouds selected by the extractor for scan slot `scan` (device to device) */
int ll_history_add_fe(ll_history *h, ll_fe *fe, int32_t scan, const double pose[7], double history_add_t_step,
                      double history_add_angle_step, int32_t *added);
/* same, taking cloud `cloud` of the last ll_reg_enqueue_fe_downsampled (the down-sampled stacks the reference pushes,
 * laser_mapping.hpp:1367-1373,1421-1431) */
int ll_history_add_voxel(ll_history *h, ll_voxel *vox_corner, ll_voxel *vox_surf, int32_t cloud, const double pose[7],
                         double history_add_t_step, double history_add_angle_step, int32_t *added);
/* The add-frame rule of laser_mapping.hpp:1439-1451 compares, and on a push records, the node's m_q_w_curr / m_t_w_curr,
 * which at that point is still the pose BEFORE the registration just done (the new pose is copied back at :1496-1500),
 * while the clouds are moved with the registered pose.  Hand that pre-registration pose over here before an
 * ll_history_add* call (it applies to the next add only); without it the registered pose gates as well, which gives the
 * same pushes as long as both steps are 0.0 (the reference's fixed values). */
int ll_history_set_gate_pose(ll_history *h, const double pose[7]);
int ll_history_refresh(ll_history *h, ll_map *map, int64_t *n_map_corner, int64_t *n_map_surf);
int32_t ll_history_size(const ll_history *h);
/* the match buffer clouds of the last refresh (host copy, for inspection / tests): returns the number of points written */
int64_t ll_history_map_cloud(ll_history *h, int32_t kind, float *xyzi, int64_t capacity_points);
/* The same clouds where they lie: a BORROWED device pointer (float4 x,y,z,intensity per point, on the history's device)
 * and the point count.  Valid until the next ll_history_refresh* on this handle; the handle's stream has been drained
 * when the call returns, so any stream of the caller may read it.  This is what the multi-GPU sub-map gather sends over
 * RCCL without a host hop. */
int ll_history_map_cloud_device(ll_history *h, int32_t kind, const float **dev_xyzi, int64_t *n_points);

/* ------------------------------------------------------------------------------------------------------------
 * Match buffer, cell ("cube") mode  (m_matching_mode == 1, the default in code,
 * laser_mapping.hpp:689).  Device-resident stand-in for
 *   Points_cloud_map<float> m_pt_cell_map_corners / m_pt_cell_map_planes   laser_mapping.hpp:274-275, 617-624
 *   Points_cloud_map::append_cloud / find_cell / find_cell_center          cell_map_keyframe.hpp:619-672, 716-759, 556-571
 *   Points_cloud_map::find_cells_in_radius                                 cell_map_keyframe.hpp:761-788
 *   Laser_mapping::if_pt_in_fov                                            laser_mapping.hpp:310-324
 *   update_buff_for_matching(), cell branch                                laser_mapping.hpp:471-513
 * A cell map bins xyz points (intensity is not kept, cell_map_keyframe.hpp:82) into cubes of edge resolution / 2
 * (set_resolution halves it, cell_map_keyframe.hpp:675-680); a cell that is hit again after not being updated for
 * minimum_revisit_threshold appended clouds starts over empty (cell_map_keyframe.hpp:735-756).
 * ll_cellmap_query_filter: the cells whose centre lies within `radius` of the pose's translation and within
 * maximum_in_fov_angle degrees of its x axis, each passed through pcl::VoxelGrid(leaf) on its own, concatenated;
 * down_sample_replace != 0 stores the filtered points back into the cells (laser_mapping.hpp:492-495).
 * maximum_in_fov_angle >= 360 switches the field-of-view test off: find_cells_in_radius on its own, as service_pub_surround_pts
 * uses it for /laser_cloud_surround (laser_mapping.hpp:1172-1187).
 * Order of the result: cells ascending by (ix, iy, iz) -- the reference's order is that of a PCL octree traversal and
 * is not reproducible; within a cell, PCL's leaf order.  Non-finite points and points beyond +-2^20 cells are dropped. */
typedef struct ll_cellmap ll_cellmap;
int ll_cellmap_create(int32_t device, int64_t max_points, float resolution, int32_t minimum_revisit_threshold, ll_cellmap **out);
void ll_cellmap_destroy(ll_cellmap *c);
/* raises the capacity to max_points (no-op when not larger); stored points, cells, revisit stamps and the frame counter are kept.  The
 * reference's Points_cloud_map grows on the heap without bound (cell_map_keyframe.hpp:619-672). */
int ll_cellmap_reserve(ll_cellmap *c, int64_t max_points);
int ll_cellmap_append(ll_cellmap *c, const float *xyzi, int32_t n);
/* append_cloud( pts, &cell_vec ) (cell_map_keyframe.hpp:619-672, the form the mapping node calls when loop closure is on,
 * laser_mapping.hpp:1527): the append, plus the cells that received at least min_points of this cloud's points (3 in the
 * reference, :646; on an empty map every cell that received a point, set_point_cloud :596-607), as cell indices [n][3] in
 * ascending cell order.  cell_ijk == NULL only counts. */
/* (ll_cellmap_append_touched never fails after the cloud has been stored: *n_touched is always the full number of touched cells; when it
 * exceeds capacity_cells the list was cut to capacity_cells entries -- call again with nothing to append is NOT a retry, the cloud is in.) */
int ll_cellmap_append_touched(ll_cellmap *c, const float *xyzi, int32_t n, int32_t min_points, int32_t *cell_ijk, int64_t capacity_cells,
                              int64_t *n_touched);
int ll_cellmap_query_filter(ll_cellmap *c, const double pose[7], float radius, float maximum_in_fov_angle, float leaf,
                            int32_t down_sample_replace, int64_t *n_cells_selected, int64_t *n_out);
/* the concatenated cloud of the last query (host copy); xyzi == NULL returns its size */
int64_t ll_cellmap_result(ll_cellmap *c, float *xyzi, int64_t capacity_points);
int ll_cellmap_stats(const ll_cellmap *c, int64_t *n_cells, int64_t *n_points, int32_t *frame_idx);
/* the whole map for inspection / tests: points in (cell, insertion) order, cell indices [n_cells][3], first point of
 * each cell [n_cells + 1], m_last_update_frame_idx [n_cells]; any output may be NULL */
int ll_cellmap_dump(ll_cellmap *c, float *xyzi, int64_t capacity_points, int32_t *cell_ijk, int32_t *cell_start,
                    int32_t *cell_last_update, int64_t capacity_cells);
/* The map where it lies: device pointers to the stored points ({x, y, z, 0} float4, ordered by (cell key, insertion order)) and to the
 * 64-bit cell key of every point.  Valid until the next call that changes this map.  Input of the multi-GPU gather of cell maps
 * (BASELINE config C4; laser_mapping.hpp:274-275, 1492-1493). */
int ll_cellmap_device_view(ll_cellmap *c, const float **dev_xyz0, const uint64_t **dev_point_keys, int64_t *n_points, int64_t *n_cells);

/* Points_cloud_cell::determine_feature( if_recompute = 1 ) for every cell (cell_map_keyframe.hpp:436-473 with get_mean
 * :225-237, get_covmat :280-315, covmat_eig_decompose :239-249; SURVEY 8(f) row 4, first half): float sums and second
 * moments over the cell's points in insertion order (COMP_TYPE = float, :41), cov = (sum p p^T - n mean mean^T) / (n - 1),
 * eigen decomposition, then  < 5 points or |centre - mean| > 0.75 cell edge -> sphere;  l1 / 3 > l0 -> plane, vector =
 * eigenvector of l0;  l2 / 3 > l1 -> line, vector = eigenvector of l2;  otherwise sphere (vector reported as zero).
 * feature_type: 0 sphere, 1 line, 2 plane (Feature_type, :46-51).  Outputs in the cell order of ll_cellmap_dump:
 * feature_vector / mean / eigen_val [n_cells][3] (eigenvalues ascending), cov [n_cells][6] (xx xy xz yy yz zz); any may
 * be NULL.  The eigen decomposition is a double-precision Jacobi iteration of the float covariance; eigenvectors are
 * reported with their first non-zero component positive (Eigen's signs are arbitrary). */
int ll_cellmap_features(ll_cellmap *c, int32_t *feature_type, float *feature_vector, float *mean, float *cov, float *eigen_val,
                        int64_t capacity_cells);

/* Key-frame descriptors over the cells of a cell map: Maps_keyframe::analyze ->
 * extract_feature_mapping_new -> generate_feature_img (cell_map_keyframe.hpp:1486-1493, 1429-1484, 1385-1427) with the map
 * standing for the key frame's cell set.  Every line / plane cell contributes its feature vector, rotated into the
 * principal axes of the plane normals (eigen_decompose_of_featurevector, :1554-1567; largest eigenvalue first, third axis =
 * first x second), to a 60 x 60 (phi, theta) direction histogram (feature_direction, :1071-1089), which is then blurred
 * with a 9 x 9, sigma 4 Gaussian on the wrap-padded image (apply_guassian_blur, :1360-1372).
 *   images[4][60*60]   m_feature_img_line, m_feature_img_plane, then the same two over the cells closer to the key-frame
 *                      centre than m_roi_range (row = phi index, column = theta index)
 *   ratio_nonzero[4]   ratio_of_nonzero_in_img of the four histograms before the blur (:1142-1152)
 *   eigen_R[2][9]      the two rotations, row-major          n_vectors[4]   feature vectors per image
 *   centre_and_range   get_center() (:1291-1301) and m_roi_range = element ceil((k - 1) * roi_ratio) of the k distinct
 *                      centre distances (get_ratio_range_of_cell, :1303-1319; roi_ratio 0.9 in the reference, :1438);
 *                      roi_ratio = 0 skips the two ROI images
 * The reference iterates a std::set of cell pointers (address order) and takes Eigen's eigenvector signs; here cells come
 * in cell-index order and eigenvectors have their first non-zero component positive.  OpenCV's float summation order in
 * cv::GaussianBlur is not reproduced (kernel coefficients are cv::getGaussianKernel's). */
int ll_cellmap_keyframe_images(ll_cellmap *c, float roi_ratio, float *images, float *ratio_nonzero, float *eigen_R, int32_t *n_vectors,
                               float *centre_and_range);
/* Maps_keyframe::max_similiarity_of_two_image (cell_map_keyframe.hpp:1155-1224, minimum_zero_ratio = 0): the maximum of
 * cv::matchTemplate( wrap-padded img_b, img_a, CV_TM_CCORR_NORMED ), i.e. of the normalised correlation over circular
 * shifts of -30 .. +30 bins along both axes.  img_a, img_b: 60 x 60 host images. */
int ll_keyframe_similarity(int32_t device, const float *img_a, const float *img_b, float *similarity);

/* The two cell maps of the mapping node, fed by every frame ll_history_add* receives (laser_mapping.hpp:1492-1493: the
 * voxel-filtered map-frame features, whether or not the frame enters the history), and the cell branch of
 * update_buff_for_matching: query + per-cell VoxelGrid (leaf line_res / plane_res) -> VoxelGrid of the concatenation ->
 * the two search structures of `map`.  ll_history_cell_map returns a borrowed handle (stats / dump). */
int ll_history_enable_cell_map(ll_history *h, int64_t max_points, float cell_resolution, int32_t threshold_cell_revisit);
ll_cellmap *ll_history_cell_map(ll_history *h, int32_t kind);
/* enable != 0: frames handed to ll_history_add* reach the two cell maps through a service thread of the handle, in order, beside the
 * caller -- in matching mode 0 nothing reads them between frames (laser_mapping.hpp:1492-1493 only appends; the reference runs its map
 * services on threads as well, :568-594), and an append re-sorts the stored map.  Every entry point that reads the cell maps
 * (ll_history_cell_map, ll_history_refresh_cells) waits for the frames handed over so far, as does ll_history_sync_cell_maps; a failure
 * of the thread is reported by the next of those calls.  The history-owned cell maps double their capacity when a frame would not fit. */
int ll_history_set_cell_map_async(ll_history *h, int32_t enable);
int ll_history_sync_cell_maps(ll_history *h);
int ll_history_refresh_cells(ll_history *h, ll_map *map, const double pose[7], float maximum_search_range_corner,
                             float maximum_search_range_surface, float maximum_in_fov_angle, int32_t down_sample_replace,
                             int64_t *n_map_corner, int64_t *n_map_surf);

/* unsigned int Point_cloud_registration::pointcloudAssociateToMap(pc_in, pc_out, if_undistore = 0)
 * (point_cloud_registration.hpp:673-685, no-deblur branch :629): p_w = q*p + t in double, stored float. */
int ll_cloud_transform(ll_reg *r, const float *in_xyzi, float *out_xyzi, int32_t n, const double pose[7]);
/* Device-resident form over an extractor's batch (the sub-map of a batched offline run): for every scan
 * b < n_scans with accept[b] != 0, the selected features of `kind` (0 corner, 1 surface; the last ll_fe_select* of slot b)
 * are moved to the map frame with poses[b] (pointAssociateToMap, no deblur) and appended, in scan order, to the
 * caller's DEVICE buffer dev_out_xyzi (float4 per point, capacity_points points) starting at point *n_points, which is
 * advanced.  Nothing crosses PCIe but the per-scan counts.  Fails, leaving *n_points untouched, if the capacity would
 * be exceeded. */
int ll_cloud_transform_fe_device(ll_reg *r, ll_fe *fe, int32_t n_scans, int32_t kind, const int32_t *accept, const double *poses7,
                                 float *dev_out_xyzi, int64_t capacity_points, int64_t *n_points);

/* HIP-event timing of the kernels launched by the last enqueue/solve on this handle (milliseconds, summed
 * over launches) and launch counts: [0] knn+block-build, [1] solver, [2] finalize. Enabled by
 * ll_reg_set_profiling(r, 1). */
int ll_reg_set_profiling(ll_reg *r, int32_t enable);
int ll_reg_kernel_times(ll_reg *r, float ms[3], int32_t launches[3]);

/* Builds compiled with -DLL_SOLVE_TIMING accumulate shader-clock counts per solver phase of scan slot `scan`:
 * [0] cost evaluations, [1] LM controller, [2] L1 pass, [3] de-duplication, [4] rank select, [5] total, [6] flag census,
 * [7] prune, [8] epilogue (plane-table path: table build); [9] counts how often the L1 pass took the values its last prerun
 * evaluation had left behind; plane-table path: [10] census load waits, [11] triple inserts, [12] block sums, [13] id compaction,
 * [14] plane constants, [15] id pass + LDS fill.  Zeros in normal builds. */
int ll_reg_debug_cycles(ll_reg *r, int32_t scan, long long out[16]);

/* Lengths of the neighbour-reuse work lists left by the last ICP iteration of the last solve, summed over the first
 * n_scans slots: out[0] / out[1] = corner queries that needed a full search / whose five neighbours were re-sorted,
 * out[2] / out[3] = the same for surface queries. */
int ll_reg_debug_worklists(ll_reg *r, int32_t n_scans, int64_t out[4]);

/* The HIP stream the handle launches on (hipStream_t), for callers that want their own events on it. */
void *ll_reg_stream(ll_reg *r);
void *ll_fe_stream(ll_fe *h);

/* Process-wide runtime hint.  The ROCm runtime multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4);
 * every handle launches on its own stream, so a host that keeps several registrations in flight (the reference runs up to
 * maximum_parallel_thread process_new_scan tasks at once, laser_mapping.hpp:1737-1742) gets two of them on one queue, where their
 * kernels serialise (measured on MI355X: 44.0 k -> 46.5 k scans/s with three 256-scan batches in flight, profiles/README.md).
 * ll_runtime_hint_hw_queues(n) sets GPU_MAX_HW_QUEUES=n for this process when the caller's environment does not set it.
 * It only takes effect when called before the process's FIRST HIP call (any library's); it never overrides the environment.
 * Returns 1 if it set the variable, 0 if the environment already had it (nothing changed), < 0 on a bad argument.
 * The library never calls it on its own: loam_livox_adapter.hpp does (once, before its first *_create) unless
 * LOAM_LIVOX_HIP_NO_RUNTIME_HINTS is defined; the Python package leaves it to the application (bench.py sets the variable itself). */
int ll_runtime_hint_hw_queues(int32_t n);

const char *ll_last_error(void);
const char *ll_version(void);

#ifdef __cplusplus
}
#endif
#endif
