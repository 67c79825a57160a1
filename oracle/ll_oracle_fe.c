/*
 * ll_oracle_fe.c -- CPU ORACLE (test infrastructure only, see ll_oracle.h) for the Livox
 * feature extractor: a plain-C restatement of hku-mars/loam_livox
 * source/livox_feature_extractor.hpp (LFE).  Pinned BIT-EXACT against the reference's own header compiled here
 * (oracle/_ref, tests/test_ref_pin.py, tests/golden/ref_scene*.npz).
 *
 * All arithmetic is fp32 unless the reference promotes to double (noted inline).
 * Compile with -ffp-contract=off: the reference is built without FMA contraction
 * (x86-64 baseline, CMakeLists.txt:5-6).
 */
#include "ll_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_fe_timebase_init(orc_fe_timebase *tb)
{
    tb->first_receive_time = -1.0; /* LFE:150 */
    tb->current_time = 0.0;
    tb->last_maximum_time_stamp = 0.0; /* LFE:152 uninitialised in the reference; defined as 0 */
}

/* LFE:724-736 */
double orc_fe_timebase_next(orc_fe_timebase *tb, double time_stamp)
{
    if (time_stamp <= 0.0000001 || (time_stamp < tb->last_maximum_time_stamp)) {
        tb->current_time = tb->last_maximum_time_stamp; /* LFE:727 */
    } else {
        tb->current_time = time_stamp - tb->first_receive_time; /* LFE:731 */
    }
    if (tb->first_receive_time <= 0) {
        tb->first_receive_time = time_stamp; /* LFE:735 */
    }
    return tb->current_time;
}

/* LFE:185: std::pow( tan( max_fov / 57.3 ) * 1, 2 ), double math stored to a float member */
float orc_fe_max_edge_polar_pos(float max_fov)
{
    return (float)pow(tan((double)max_fov / 57.3) * 1, 2);
}

/* add_mask_of_point LFE:322-341: OR `mask` into point idx and, when neighbor_count>0,
 * into idx+i for i in [-neighbor_count, neighbor_count) \ {0}. */
static void add_mask(int32_t *pt_type, int n, int idx, int mask, int neighbor_count)
{
    pt_type[idx] |= mask;
    if (neighbor_count > 0) {
        for (int i = -neighbor_count; i < neighbor_count; i++) {
            int j = idx + i;
            if (i != 0 && j >= 0 && j < n)
                pt_type[j] |= mask;
        }
    }
}

/* Eigen_math::vector_angle<float>(a, b, 1), EM:25-46, with Eigen's 3-element reduction order
 * e0 + (e1 + e2) for dot() and squaredNorm(); float |.| and acosf (libstdc++ >= 6 overloads). */
static float vector_angle_sharp(const float a[3], const float b[3])
{
    float na = sqrtf(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]));
    float nb = sqrtf(b[0] * b[0] + (b[1] * b[1] + b[2] * b[2]));
    if (na == 0 || nb == 0)
        return 0.0f;
    float d = a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
    return acosf(fabsf(d) / (na * nb));
}

/* compute_features, LFE:361-455 */
static void compute_features(const orc_fe_params *p, const float *xyzi, int n, const int32_t *pt_type,
                             int32_t *pt_label, const float *depth_sq2, float *curvature, float *view_angle)
{
    const int ssd = 2;                                   /* curvature_ssd_size, LFE:364 */
    const int critical_rm_point = ORC_PT_000 | ORC_PT_NAN; /* LFE:365 */
    if (n < 2 * ssd + 1)
        return; /* reference loop bound pts_size-2 underflows for n<2; defined as "no labels" */
    for (int idx = ssd; idx < n - ssd; idx++) {
        if (pt_type[idx] & critical_rm_point)
            continue;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 1; i <= ssd; i++) {
            if ((pt_type[idx + i] & ORC_PT_000) || (pt_type[idx - i] & ORC_PT_000)) {
                if (i == 1)
                    pt_label[idx] |= ORC_LABEL_NEAR_ZERO;
                else
                    pt_label[idx] = ORC_LABEL_INVALID;
                break;
            } else if ((pt_type[idx + i] & ORC_PT_NAN) || (pt_type[idx - i] & ORC_PT_NAN)) {
                if (i == 1)
                    pt_label[idx] |= ORC_LABEL_NEAR_NAN;
                else
                    pt_label[idx] = ORC_LABEL_INVALID;
                break;
            } else {
                acc[0] += xyzi[4 * (idx + i) + 0] + xyzi[4 * (idx - i) + 0];
                acc[1] += xyzi[4 * (idx + i) + 1] + xyzi[4 * (idx - i) + 1];
                acc[2] += xyzi[4 * (idx + i) + 2] + xyzi[4 * (idx - i) + 2];
            }
        }
        if (pt_label[idx] == ORC_LABEL_INVALID)
            continue;
        /* curvature_ssd_size * 2 * x : (size_t)2*2 -> 4, converted to float 4.0f */
        acc[0] -= 4.0f * xyzi[4 * idx + 0];
        acc[1] -= 4.0f * xyzi[4 * idx + 1];
        acc[2] -= 4.0f * xyzi[4 * idx + 2];
        curvature[idx] = acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2];

        float va[3] = {xyzi[4 * idx + 0], xyzi[4 * idx + 1], xyzi[4 * idx + 2]};
        float vb[3] = {xyzi[4 * (idx + ssd) + 0] - xyzi[4 * (idx - ssd) + 0],
                       xyzi[4 * (idx + ssd) + 1] - xyzi[4 * (idx - ssd) + 1],
                       xyzi[4 * (idx + ssd) + 2] - xyzi[4 * (idx - ssd) + 2]};
        /* float angle * 57.3 (double) stored to the float field, LFE:430 */
        view_angle[idx] = (float)((double)vector_angle_sharp(va, vb) * 57.3);

        if (view_angle[idx] > p->minimum_view_angle) {
            if (curvature[idx] < p->thr_surface_curvature)
                pt_label[idx] |= ORC_LABEL_SURFACE;
            float sq2_diff = 0.1f;
            if (curvature[idx] > p->thr_corner_curvature) {
                if (depth_sq2[idx] <= depth_sq2[idx - ssd] && depth_sq2[idx] <= depth_sq2[idx + ssd]) {
                    if (fabsf(depth_sq2[idx] - depth_sq2[idx - ssd]) < sq2_diff * depth_sq2[idx] ||
                        fabsf(depth_sq2[idx] - depth_sq2[idx + ssd]) < sq2_diff * depth_sq2[idx])
                        pt_label[idx] |= ORC_LABEL_CORNER;
                }
            }
        }
    }
}

int orc_fe_extract(const orc_fe_params *p, const float *xyzi, int n, double current_time,
                   int32_t *pt_type, int32_t *pt_label, float *time_stamp, float *polar_angle,
                   int32_t *polar_direction, float *polar_dis_sq2, float *depth_sq2,
                   float *curvature, float *view_angle, float *sigma, float *img2d,
                   int32_t *split_idx, int32_t *n_split, float *last_time_stamp)
{
    const float max_edge_polar_pos = orc_fe_max_edge_polar_pos(p->max_fov);
    int n_splits = 0;
    int n_edge = 0, n_zero = 0;

    /* m_pts_info_vec.clear(); resize(): value-initialised Pt_infos (LFE:118-133,462-463) */
    for (int i = 0; i < n; i++) {
        pt_type[i] = ORC_PT_NORMAL;
        pt_label[i] = ORC_LABEL_UNLABELED;
        time_stamp[i] = 0.0f;
        polar_angle[i] = 0.0f;
        polar_direction[i] = 0;
        polar_dis_sq2[i] = 0.0f;
        depth_sq2[i] = 0.0f;
        curvature[i] = 0.0f;
        view_angle[i] = 0.0f;
        sigma[i] = 0.0f;
        img2d[2 * i] = 0.0f;
        img2d[2 * i + 1] = 0.0f;
    }
    if (last_time_stamp)
        *last_time_stamp = 0.0f;

    for (int idx = 0; idx < n; idx++) { /* LFE:474-564 */
        const float x = xyzi[4 * idx + 0], y = xyzi[4 * idx + 1], z = xyzi[4 * idx + 2];
        const float inten = xyzi[4 * idx + 3];
        /* LFE:481: double + (float*float) -> float */
        time_stamp[idx] = (float)(current_time + (double)(((float)idx) * p->time_internal_pts));
        if (last_time_stamp)
            *last_time_stamp = time_stamp[idx];

        if (!isfinite(x) || !isfinite(y) || !isfinite(z)) { /* LFE:485-491 */
            add_mask(pt_type, n, idx, ORC_PT_NAN, 0);
            continue;
        }
        if (x == 0) { /* LFE:493-512 */
            if (idx == 0) {
                img2d[0] = 0.01f;
                img2d[1] = 0.01f;
                polar_dis_sq2[0] = 0.0001f;
                add_mask(pt_type, n, idx, ORC_PT_000, 0);
                /* falls through (no return/continue in the reference) */
            } else {
                img2d[2 * idx] = img2d[2 * (idx - 1)];
                img2d[2 * idx + 1] = img2d[2 * (idx - 1) + 1];
                polar_dis_sq2[idx] = polar_dis_sq2[idx - 1];
                add_mask(pt_type, n, idx, ORC_PT_000, 0);
                continue;
            }
        }
        depth_sq2[idx] = x * x + y * y + z * z; /* LFE:516,194-198 */
        img2d[2 * idx] = y / x;                 /* LFE:518 */
        img2d[2 * idx + 1] = z / x;
        polar_dis_sq2[idx] = img2d[2 * idx] * img2d[2 * idx] + img2d[2 * idx + 1] * img2d[2 * idx + 1]; /* LFE:519 */

        /* eval_point LFE:343-358 */
        if (depth_sq2[idx] < p->livox_min_allow_dis * p->livox_min_allow_dis)
            add_mask(pt_type, n, idx, ORC_PT_TOO_NEAR, 0);
        sigma[idx] = inten / polar_dis_sq2[idx];
        if (sigma[idx] < p->livox_min_sigma)
            add_mask(pt_type, n, idx, ORC_PT_REFL_LOW, 0);

        if (polar_dis_sq2[idx] > max_edge_polar_pos) /* LFE:523-526 */
            add_mask(pt_type, n, idx, ORC_PT_CIRCLE_EDGE, 2);

        if (idx >= 1) { /* LFE:529-563 */
            float dis_incre = polar_dis_sq2[idx] - polar_dis_sq2[idx - 1];
            if (dis_incre > 0)
                polar_direction[idx] = 1;
            if (dis_incre < 0)
                polar_direction[idx] = -1;
            if (polar_direction[idx] == -1 && polar_direction[idx - 1] == 1) {
                if (n_edge == 0 || (idx - split_idx[n_splits - 1]) > 50) {
                    split_idx[n_splits++] = idx;
                    n_edge++;
                    continue;
                }
            }
            if (polar_direction[idx] == 1 && polar_direction[idx - 1] == -1) {
                if (n_zero == 0 || (idx - split_idx[n_splits - 1]) > 50) {
                    split_idx[n_splits++] = idx;
                    n_zero++;
                    continue;
                }
            }
        }
    }
    split_idx[n_splits++] = n - 1; /* LFE:565 */
    *n_split = n_splits;

    int ret;
    if (n_splits < 6) { /* LFE:572 */
        ret = 0;
    } else {
        int val_index = 0;
        int pt_angle_index = 0;
        float scan_angle = 0;
        int internal_size = 0;
        for (int idx = 0; idx < n; idx++) { /* LFE:575-604 */
            if (val_index < n_splits - 2) {
                if (idx == 0 || idx > split_idx[val_index + 1]) {
                    if (idx > split_idx[val_index + 1])
                        val_index++;
                    internal_size = split_idx[val_index + 1] - split_idx[val_index];
                    if (polar_dis_sq2[split_idx[val_index + 1]] > 10000)
                        pt_angle_index = split_idx[val_index + 1] - (int)(internal_size * 0.20);
                    else
                        pt_angle_index = split_idx[val_index + 1] - (int)(internal_size * 0.80);
                    /* atan2f(float,float) * 57.3 (double) -> float; + 180.0 (double) -> float */
                    scan_angle = (float)((double)atan2f(img2d[2 * pt_angle_index + 1], img2d[2 * pt_angle_index]) * 57.3);
                    scan_angle = (float)((double)scan_angle + 180.0);
                }
            }
            polar_angle[idx] = scan_angle;
        }
        ret = n_splits - 1; /* LFE:606 */
    }

    /* extract_laser_features calls compute_features() regardless of the petal count, LFE:755-756 */
    compute_features(p, xyzi, n, pt_type, pt_label, depth_sq2, curvature, view_angle);
    return ret;
}

/* get_features, LFE:219-272 */
void orc_fe_get_features(int n, const int32_t *pt_type, const int32_t *pt_label, const float *depth_sq2,
                         float minimum_blur, float maximum_blur,
                         int32_t *corner_idx, int32_t *n_corner,
                         int32_t *surf_idx, int32_t *n_surf,
                         int32_t *full_idx, int32_t *n_full)
{
    int corner_num = 0, surface_num = 0, full_num = 0;
    float maximum_idx = maximum_blur * n; /* float * size_t -> float, LFE:227 */
    float minimum_idx = minimum_blur * n;
    const int pt_critical_rm_mask = ORC_PT_000 | ORC_PT_NAN | ORC_PT_TOO_NEAR;
    for (int i = 0; i < n; i++) {
        if ((float)i > maximum_idx || (float)i < minimum_idx) /* idx == i, LFE:232-234 */
            continue;
        if ((pt_type[i] & pt_critical_rm_mask) == 0) {
            if (pt_label[i] & ORC_LABEL_CORNER) {
                /* `continue` here also skips the surface test and the full cloud, LFE:240-241 */
                if (pt_type[i] != ORC_PT_NORMAL)
                    continue;
                if (depth_sq2[i] < 900.0f) /* std::pow(30,2) */
                    corner_idx[corner_num++] = i;
            }
            if (pt_label[i] & ORC_LABEL_SURFACE) {
                if (depth_sq2[i] < 1000000.0f) /* std::pow(1000,2) */
                    surf_idx[surface_num++] = i;
            }
        }
        full_idx[full_num++] = i; /* LFE:263-265 */
    }
    *n_corner = corner_num;
    *n_surf = surface_num;
    *n_full = full_num;
}

/* first index holding the same (x,y,z): the unordered_map keeps the first insertion (LFE:478, PT:23-46) */
static int first_occurrence(const float *xyzi, int idx)
{
    const float x = xyzi[4 * idx], y = xyzi[4 * idx + 1], z = xyzi[4 * idx + 2];
    for (int j = 0; j < idx; j++)
        if (xyzi[4 * j] == x && xyzi[4 * j + 1] == y && xyzi[4 * j + 2] == z)
            return j;
    return idx;
}

/* split_laser_scan, LFE:657-719 */
int orc_fe_split_scan(int n, int clutter_size, const float *xyzi, const int32_t *pt_type,
                      const float *polar_angle, int32_t *first_idx, int32_t *last_idx)
{
    if (clutter_size <= 0 || n <= 0)
        return 0;
    const int remove_mask = ORC_PT_000 | ORC_PT_TOO_NEAR | ORC_PT_NAN; /* LFE:684-688 */
    /* run boundaries: a new petal starts where scan_id_index changes, LFE:672 */
    int *run_start = (int *)malloc(sizeof(int) * (size_t)(clutter_size + 1));
    int scan_idx = 0;
    run_start[0] = 0;
    for (int i = 1; i < n; i++) {
        if (polar_angle[i] != polar_angle[i - 1]) {
            scan_idx++;
            if (scan_idx > clutter_size) { /* cannot happen (see LFE:575-604); guard anyway */
                scan_idx = clutter_size;
                break;
            }
            run_start[scan_idx] = i;
        }
    }
    /* laserCloudScans.resize(scan_idx): the LAST run is discarded, LFE:681 */
    int n_runs = scan_idx;
    int out = 0;
    for (int r = 0; r < n_runs; r++) {
        int b = run_start[r], e = run_start[r + 1]; /* [b,e) */
        int first = -1, last = -1;
        for (int i = b; i < e; i++) {
            if ((pt_type[i] & remove_mask) == 0) {
                /* x==0 with a clean mask cannot occur (assert LFE:702) */
                if (first < 0)
                    first = i;
                last = i;
            }
        }
        if (first >= 0) { /* empty petals are dropped, LFE:713-716 */
            first_idx[out] = first;
            last_idx[out] = last;
            out++;
        }
    }
    free(run_start);
    return out;
}

/* LFX:305-323.  find_pt_info (LFE:206-217) looks the boundary points up by xyz: a point that duplicates an
 * earlier one resolves to the EARLIER index (the unordered_map keeps the first insertion, LFE:478). */
void orc_fe_piecewise(int n, const float *xyzi, int n_petal_clouds, const int32_t *first_idx, const int32_t *last_idx,
                      int pieces, float *piece_start, float *piece_end)
{
    int m_laser_scan_number = n_petal_clouds; /* LFX:90,292: int member */
    for (int i = 0; i < pieces; i++) {
        int start_scans = (m_laser_scan_number * (i)) / pieces; /* integer division, LFX:317-318 */
        int end_scans = (m_laser_scan_number * (i + 1)) / pieces - 1;
        piece_start[i] = ((float)first_occurrence(xyzi, first_idx[start_scans])) / n; /* float / size_t -> float */
        piece_end[i] = ((float)first_occurrence(xyzi, last_idx[end_scans])) / n;
    }
}
