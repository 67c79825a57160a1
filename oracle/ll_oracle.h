/*
 * ll_oracle.h -- CPU ORACLE for the Loam-Livox scan-to-map hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (loam_livox_amd/,
 * include/loam_livox_hip.h) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm (hku-mars/loam_livox):
 *   feature extraction : source/livox_feature_extractor.hpp           (LFE)
 *   registration       : source/point_cloud_registration.hpp          (PCR)
 *   residual functors  : source/ceres_icp.hpp                         (ICP)
 *   helpers            : include/tools/tools_eigen_math.hpp           (EM)
 * Each function cites the reference file:line it follows.
 *
 * HOW IT IS PINNED.  The reference ships no tests, golden vectors or fixtures, and PCL, Ceres, Eigen3 and ROS are
 * absent here and not vendored.  The reference's OWN hot-path headers are nevertheless compiled verbatim in this
 * container (oracle/_ref/libll_ref.so, `make -C oracle ref`) against minimal stand-in third-party headers
 * (oracle/ref_stubs/), and this restatement is checked against that library (tests/test_ref_pin.py) and against its
 * committed outputs (tests/golden/ref_scene*.npz, tests/test_ref_golden.py):
 *   - feature extraction (LFE, rows a1-a6): every Pt_infos field, get_features clouds / index sets, petal clouds:
 *     BIT-EXACT.  The only third-party arithmetic on that path is Eigen's 3-vector dot()/norm() order.
 *   - residual functors (ICP, rows a10/a11, incl. _mb): residuals and AutoDiff Jacobians to 1e-12.
 *   - registration driver (PCR:163-583, rows a7, a9 line check, a12, a14, a15): the reference's own control flow,
 *     pose / costs / thresholds / block counts to 1e-9.
 * STILL UNPINNED (restated from the libraries' published behaviour, in ref_stubs/ and here alike):
 *   - PCL KdTreeFLANN::nearestKSearch (FLANN KDTreeSingleIndex, L2_Simple<float>):
 *     exact k-NN, squared distance accumulated in fp32 in x,y,z order, sorted ascending.
 *     Exact-distance ties are broken by the lower point index (FLANN's tie order depends
 *     on its private tree layout and is not reproducible).
 *   - Ceres Solver (< 2.2) trust-region Levenberg-Marquardt with default options
 *     (see ll_oracle_reg.c header; ref_stubs/ll_stub_ceres_solver.h is a second, dense formulation of it).
 *   - Eigen3 fixed-size reductions: dot()/squaredNorm() of a float 3-vector evaluate as
 *     e0 + (e1 + e2) (redux_novec_unroller), used by EM::vector_angle.
 *   - pcl::VoxelGrid (PCL 1.9 semantics, ll_oracle_voxel.c).
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off; x86-64 baseline => no FMA,
 * matching the reference's "-std=c++14 -O3" CMake flags, CMakeLists.txt:5-6).
 */
#ifndef LL_ORACLE_H
#define LL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- feature extraction (LFE) ---------------- */

/* Point-type bit masks, LFE:82-92 */
enum {
    ORC_PT_NORMAL = 0,
    ORC_PT_000 = 1,
    ORC_PT_TOO_NEAR = 2,
    ORC_PT_REFL_LOW = 4,
    ORC_PT_REFL_HIGH = 8,
    ORC_PT_CIRCLE_EDGE = 16,
    ORC_PT_NAN = 32,
    ORC_PT_SMALL_VIEW_ANGLE = 64
};
/* Feature labels, LFE:94-103 */
enum {
    ORC_LABEL_INVALID = -1,
    ORC_LABEL_UNLABELED = 0,
    ORC_LABEL_CORNER = 1,
    ORC_LABEL_SURFACE = 2,
    ORC_LABEL_NEAR_NAN = 4,
    ORC_LABEL_NEAR_ZERO = 8,
    ORC_LABEL_HIGH_INTENSITY = 16
};

typedef struct {
    float thr_corner_curvature;  /* LFE:153 (0.05) */
    float thr_surface_curvature; /* LFE:154 (0.01) */
    float minimum_view_angle;    /* LFE:155 (10)   */
    float livox_min_allow_dis;   /* LFE:166 (1.0; node sets 0.1, LFX:854) */
    float livox_min_sigma;       /* LFE:167 (7e-3; node sets 7e-4, LFX:859) */
    float max_fov;               /* LFE:143 (17)   */
    float time_internal_pts;     /* LFE:145 (1e-5) */
} orc_fe_params;

/* Sequential time-base state of Livox_laser, LFE:150-152,722-736.
 * Reference leaves m_last_maximum_time_stamp uninitialised; the oracle defines it as 0. */
typedef struct {
    double first_receive_time;      /* init -1 */
    double current_time;            /* init 0  */
    double last_maximum_time_stamp; /* init 0  */
} orc_fe_timebase;

void orc_fe_timebase_init(orc_fe_timebase *tb);
/* LFE:724-736: returns m_current_time to be used for this scan and updates first_receive_time. */
double orc_fe_timebase_next(orc_fe_timebase *tb, double time_stamp);

/* m_max_edge_polar_pos = pow(tan(max_fov/57.3)*1, 2), LFE:185 (double math, float store) */
float orc_fe_max_edge_polar_pos(float max_fov);

/* projection_scan_3d_2d (LFE:458-607) + eval_point (LFE:343-358) + add_mask_of_point (LFE:322-341)
 * + compute_features (LFE:361-455).
 * xyzi: n x 4 floats (x,y,z,intensity).  All output arrays have n entries (img2d: 2n; split_idx: n+1).
 * Returns the petal count ("clutter_size", LFE:606), 0 if fewer than 6 split entries (LFE:572).
 * On return *last_time_stamp is the float time stamp of the last point (-> m_last_maximum_time_stamp). */
int orc_fe_extract(const orc_fe_params *p, const float *xyzi, int n, double current_time,
                   int32_t *pt_type, int32_t *pt_label, float *time_stamp, float *polar_angle,
                   int32_t *polar_direction, float *polar_dis_sq2, float *depth_sq2,
                   float *curvature, float *view_angle, float *sigma, float *img2d,
                   int32_t *split_idx, int32_t *n_split, float *last_time_stamp);

/* get_features (LFE:219-272): index lists (ascending) of corner / surface / full selections. */
void orc_fe_get_features(int n, const int32_t *pt_type, const int32_t *pt_label, const float *depth_sq2,
                         float minimum_blur, float maximum_blur,
                         int32_t *corner_idx, int32_t *n_corner,
                         int32_t *surf_idx, int32_t *n_surf,
                         int32_t *full_idx, int32_t *n_full);

/* split_laser_scan (LFE:657-719): returns the number of surviving petal clouds S; for each, the index of its
 * first and last surviving point (the points LFX:317-322 then looks up through find_pt_info).
 * clutter_size is the value returned by orc_fe_extract. first_idx/last_idx have capacity clutter_size. */
int orc_fe_split_scan(int n, int clutter_size, const float *xyzi, const int32_t *pt_type,
                      const float *polar_angle, int32_t *first_idx, int32_t *last_idx);

/* piece-wise windows (LFX:305-323): start/end blur for each of `pieces` windows (boundary points resolved
 * through the first-occurrence semantics of find_pt_info). */
void orc_fe_piecewise(int n, const float *xyzi, int n_petal_clouds, const int32_t *first_idx, const int32_t *last_idx,
                      int pieces, float *piece_start, float *piece_end);

/* ---------------- k-NN (PCL KdTreeFLANN restatement) ---------------- */

typedef struct orc_kdtree orc_kdtree;
/* xyz: m points, `stride` floats apart (3 or 4). The tree keeps a pointer to xyz (not copied). */
orc_kdtree *orc_kdtree_build(const float *xyz, int stride, int64_t m);
void orc_kdtree_free(orc_kdtree *t);
/* exact k-NN, ascending (d2, idx). Returns number found (= min(k, m)). PCR:249,351 */
int orc_kdtree_knn(const orc_kdtree *t, const float q[3], int k, int32_t *idx, float *d2);
/* brute force version with identical semantics (used to validate the tree). */
int orc_bruteforce_knn(const float *xyz, int stride, int64_t m, const float q[3], int k, int32_t *idx, float *d2);

/* ---------------- registration (PCR + ICP) ---------------- */

typedef struct {
    int if_motion_deblur;           /* PCR:60   */
    int icp_max_iterations;         /* PCR:89   */
    int ceres_max_iterations;       /* PCR:90   */
    int ceres_prerun_times;         /* PCR:91 (2) */
    int line_search_num;            /* PCR:45 (5) */
    int plane_search_num;           /* PCR:47 (5) */
    int icp_line;                   /* PCR:50   */
    int icp_plane;                  /* PCR:49   */
    int current_frame_index;        /* PCR:83   */
    int mapping_init_accumulate_frames; /* PCR:84 */
    int force_all_iterations;       /* harness switch: disable the PCR:521-526 break (BASELINE config C2) */
    double maximum_dis_line_for_match;  /* PCR:65 (2.0)  */
    double maximum_dis_plane_for_match; /* PCR:64 (50.0) */
    double huber_a;                 /* PCR:220 (0.1) */
    double inliner_dis;             /* PCR:97 (0.02) */
    double inlier_ratio;            /* PCR:98 (0.8)  */
    double minimum_icp_R_diff;      /* PCR:94 */
    double minimum_icp_T_diff;      /* PCR:95 */
    float para_max_angular_rate;    /* PCR:86 */
    float para_max_speed;           /* PCR:87 */
    float max_final_cost;           /* PCR:88 */
    float minimum_pt_time_stamp;    /* PCR:92 */
    float maximum_pt_time_stamp;    /* PCR:93 */
    int if_line_feature_check;      /* PCR:46 (0) */
    int if_plane_feature_check;     /* PCR:48 (0); uses the surface cloud (reference bug PCR:361-363 fixed) */
    int maximum_allow_residual_block; /* PCR:103 */
    int subsample_seed;             /* 0: the sub-sampling branches (PCR:232-238,339-345,438-458) are dead; else they run on the
                                       counter-based uniform stream defined in ll_oracle_reg.c (the reference's mt19937 seeded from
                                       random_device cannot be reproduced) */
} orc_reg_params;

typedef struct {
    double final_cost, initial_cost;     /* summary of the LAST ceres::Solve, PCR:508,559 */
    double inlier_threshold;             /* PCR:485,559 */
    double angular_diff_deg, t_diff;     /* PCR:517-518 */
    int icp_iterations;                  /* number of ICP iterations executed */
    int n_blocks_last;                   /* residual blocks in the last final solve */
    int corner_avail, surf_avail;        /* PCR:325,425 (last ICP iteration) */
    int lm_iterations_total;             /* sum of LM iterations over all solves */
    int accepted;                        /* 1 ok / skipped-by-gate, 0 rejected */
    int gated;                           /* 1 if the PCR:199 gate skipped optimisation */
} orc_reg_report;

/* find_out_incremental_transfrom (PCR:163-583). pose arrays are {qx,qy,qz,qw,tx,ty,tz} (PCR:51-56).
 * map_* are xyz with `map_stride` floats per point; scan_* are xyzi (4 floats, intensity = time stamp).
 * pose_last: m_q_w_last/m_t_w_last; pose_curr in: initial guess, out: result; pose_incre: in/out
 * (identity for a fresh Point_cloud_registration, LM:1348).
 * Returns 1 (accepted / gated) or 0 (rejected, pose_curr := pose_last). */
/* test instrumentation: most contractions any projected line search has taken since the last reset (>= 2: the three-sample
 * interpolation ran) */
int orc_dbg_ls_max_contractions(int reset);
long long orc_dbg_ls_quintic_fits(int reset);
double orc_dbg_quintic_min_step(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo, double hi);
int orc_reg_solve(const orc_kdtree *tree_corner, const float *map_corner, int64_t n_map_corner,
                  const orc_kdtree *tree_surf, const float *map_surf, int64_t n_map_surf, int map_stride,
                  const float *scan_corner, int n_corner, const float *scan_surf, int n_surf,
                  const orc_reg_params *prm, const double pose_last[7], double pose_curr[7],
                  double pose_incre[7], orc_reg_report *rep);

/* Pieces exposed for unit tests / cross-checks -------------------------------------- */

/* One residual block in the form the cost functors hold it (ICP:238-380): kind 0 = line, 1 = plane.
 * f = sensor-frame point, a = first neighbour, v = unit line direction (line) or un-normalised n (plane). */
typedef struct {
    int kind;
    double f[3];
    double a[3];
    double v[3];
    double s; /* motion blur ratio (1 when deblur off) */
} orc_block;

/* Build the block constants from neighbours exactly like the functor constructors
 * (ICP:255-256 line; ICP:328-334 plane). */
void orc_block_line(orc_block *b, const double f[3], const double pa[3], const double pb[3], double s);
void orc_block_plane(orc_block *b, const double f[3], const double pa[3], const double pb[3], const double pc[3], double s);

/* Raw (un-robustified) residual of one block at increment x = {qx,qy,qz,qw,tx,ty,tz}. ICP:262-288,338-366 */
void orc_block_residual(const orc_block *b, const double pose_last[7], const double x[7], int deblur, double r[3]);

/* cost = 1/2 sum rho(|r|^2), g (6), H (6x6 row-major) in the Ceres local parameterisation
 * (EigenQuaternionParameterization tangent then t), loss-corrected like ceres::Corrector. */
void orc_blocks_eval(const orc_block *blocks, int nb, const double pose_last[7], const double x[7],
                     int deblur, double huber_a, double *cost, double g[6], double H[36]);

/* pointAssociateToMap (PCR:622-661), no-deblur branch: p_w = q*p + t in double, stored float. */
/* pcl::VoxelGrid<PointXYZI>::filter, PCL 1.9 semantics with a stable in-voxel order (ll_oracle_voxel.c).
 * out_xyzi must hold n points.  Returns 0 = filtered, 1 = leaf too small (output is a copy of the input), 2 = no finite point. */
int orc_voxel_grid(const float *xyzi, int32_t n, const float leaf[3], float *out_xyzi, int32_t *n_out);

/* PCA feature checks (PCR:259-292 line, :357-389 plane): pts = 5 float points; ev_out = eigenvalues ascending; returns pass(1)/fail(0) */
int orc_pca_check(int is_plane, const float pts[15], double ev_out[3]);

void orc_point_to_map(const double pose[7], const float p[3], float out[3]);

/* cloud transform, pointcloudAssociateToMap PCR:673-685 (no-deblur branch). xyzi in/out, n points. */
void orc_cloud_transform(const double pose[7], const float *in_xyzi, float *out_xyzi, int n);

#ifdef __cplusplus
}
#endif
#endif
