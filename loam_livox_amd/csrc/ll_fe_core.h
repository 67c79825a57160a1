// ll_fe_core.h -- per-point arithmetic of the Livox feature extractor, shared by the HIP kernels
// (ll_fe_kernels.hip) and by the host-side unit checks in tests/hostcheck (which compile these inline
// functions with g++ to test them in the no-GPU tier; the product library itself has no CPU path).
//
// Everything here is fp32 with the reference's evaluation order; compile with -ffp-contract=off
// (the reference is built for baseline x86-64: no FMA contraction).
// Reference: hku-mars/loam_livox source/livox_feature_extractor.hpp (LFE).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__clang__)
#define LL_UNROLL _Pragma("unroll")
#else
#define LL_UNROLL
#endif

#if defined(__HIPCC__)
#define LL_HD __host__ __device__ __forceinline__
#define LL_HD_NOINLINE __host__ __device__ __noinline__ inline
#else
#define LL_HD inline
#define LL_HD_NOINLINE inline
#endif

namespace ll {

enum : int { PT_000 = 1, PT_TOO_NEAR = 2, PT_REFL_LOW = 4, PT_CIRCLE_EDGE = 16, PT_NAN = 32 };
enum : int { LB_INVALID = -1, LB_CORNER = 1, LB_SURFACE = 2, LB_NEAR_NAN = 4, LB_NEAR_ZERO = 8 };

struct FeConst {
    float thr_corner_curvature, thr_surface_curvature, minimum_view_angle;
    float min_dis_sq;          // m_livox_min_allow_dis * m_livox_min_allow_dis (float product, LFE:345)
    float min_sigma;           // LFE:353
    float max_edge_polar_pos;  // LFE:185
    float time_internal_pts;   // LFE:145
    float view_angle_band;     // half width of the acosf ambiguity band around minimum_view_angle (degrees)
};

// What one point contributes on its own (no neighbour information).
struct PointOwn {
    int type_self;    // bits this point sets on itself (LFE:485-526 without the neighbour smear)
    int edge;         // 1 if the point trips the circle-edge test (LFE:523) -> smears 16 onto idx-2, idx-1, idx+1
    int defines;      // 1 if polar_dis_sq2 / pt_2d_img are this point's own values (0: inherited from idx-1, LFE:507-508)
    int reached;      // 1 if the split logic runs for this point (LFE:529), i.e. it was not `continue`d
    float depth_sq2, polar_sq2, img_y, img_z;
};

LL_HD bool ll_isfinite(float v) { return (v - v) == 0.0f; }

// LFE:474-526 for point idx with raw (x,y,z,intensity)
LL_HD PointOwn point_own(float x, float y, float z, float inten, int idx, const FeConst &c)
{
    PointOwn o;
    o.type_self = 0;
    o.edge = 0;
    o.defines = 1;
    o.reached = 0;
    o.depth_sq2 = 0.0f;
    o.polar_sq2 = 0.0f;
    o.img_y = 0.0f;
    o.img_z = 0.0f;
    if (!ll_isfinite(x) || !ll_isfinite(y) || !ll_isfinite(z)) {  // LFE:485-491
        o.type_self = PT_NAN;  // value-initialised fields stay 0 and count as "defined" for inheritance
        return o;
    }
    if (x == 0.0f) {  // LFE:493-512
        o.type_self = PT_000;
        if (idx != 0) {
            o.defines = 0;
            return o;
        }
        // idx == 0 falls through: the placeholder (0.01,0.01)/1e-4 is overwritten below (division by zero)
    }
    o.reached = 1;
    o.depth_sq2 = x * x + y * y + z * z;  // LFE:516
    o.img_y = y / x;                      // LFE:518
    o.img_z = z / x;
    o.polar_sq2 = o.img_y * o.img_y + o.img_z * o.img_z;  // LFE:519
    if (o.depth_sq2 < c.min_dis_sq) o.type_self |= PT_TOO_NEAR;  // LFE:345
    float sigma = inten / o.polar_sq2;                           // LFE:351
    if (sigma < c.min_sigma) o.type_self |= PT_REFL_LOW;         // LFE:353
    if (o.polar_sq2 > c.max_edge_polar_pos) {                    // LFE:523
        o.type_self |= PT_CIRCLE_EDGE;
        o.edge = 1;
    }
    return o;
}

// time stamp of point idx, LFE:481: double + (float*float) stored to float
LL_HD float point_time_stamp(double current_time, int idx, float time_internal_pts)
{
    return (float)(current_time + (double)(((float)idx) * time_internal_pts));
}

// cos-domain argument of Eigen_math::vector_angle<float>(a,b,1) (EM:25-46) with Eigen's 3-element reduction
// order e0 + (e1 + e2).  Returns false when either norm is zero (angle defined as 0).
LL_HD bool view_angle_cos(const float a[3], const float b[3], float *cosv)
{
    float na = sqrtf(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]));
    float nb = sqrtf(b[0] * b[0] + (b[1] * b[1] + b[2] * b[2]));
    if (na == 0.0f || nb == 0.0f) return false;
    float d = a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
    *cosv = fabsf(d) / (na * nb);
    return true;
}

// view angle in degrees from the cosine: float acos, * 57.3 in double, stored to float (LFE:430).
// On the device acosf is evaluated as the double acos rounded to float (correctly rounded except for
// double-rounding ties); the host libm acosf may differ by an ulp, which is what `view_angle_band` guards.
LL_HD float view_angle_from_cos(float cosv)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float ang = (float)acos((double)cosv);
#else
    float ang = acosf(cosv);
#endif
    return (float)((double)ang * 57.3);
}

struct LabelOut {
    int label;
    float curvature, view_angle;
    int ambiguous;  // view angle inside the ambiguity band: label must be re-derived with the host libm
};

// compute_features body for one idx in [2, n-2), LFE:368-454.
// p[k] = raw xyz of idx-2+k (k = 0..4); t[k] = full pt_type of those points; d[k] = depth_sq2 of those points.
LL_HD LabelOut point_label(const float p[5][3], const int t[5], const float d[5], const FeConst &c)
{
    LabelOut o;
    o.label = 0;
    o.curvature = 0.0f;
    o.view_angle = 0.0f;
    o.ambiguous = 0;
    if (t[2] & (PT_000 | PT_NAN)) return o;  // LFE:370
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 1; i <= 2; i++) {  // LFE:380-412
        const int tp = t[2 + i], tm = t[2 - i];
        if ((tp & PT_000) || (tm & PT_000)) {
            if (i == 1)
                o.label |= LB_NEAR_ZERO;
            else
                o.label = LB_INVALID;
            break;
        } else if ((tp & PT_NAN) || (tm & PT_NAN)) {
            if (i == 1)
                o.label |= LB_NEAR_NAN;
            else
                o.label = LB_INVALID;
            break;
        } else {
            acc[0] += p[2 + i][0] + p[2 - i][0];
            acc[1] += p[2 + i][1] + p[2 - i][1];
            acc[2] += p[2 + i][2] + p[2 - i][2];
        }
    }
    if (o.label == LB_INVALID) return o;  // LFE:414
    acc[0] -= 4.0f * p[2][0];             // LFE:419-421
    acc[1] -= 4.0f * p[2][1];
    acc[2] -= 4.0f * p[2][2];
    o.curvature = acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2];  // LFE:422

    float vb[3] = {p[4][0] - p[0][0], p[4][1] - p[0][1], p[4][2] - p[0][2]};  // LFE:427-429
    float cosv;
    if (view_angle_cos(p[2], vb, &cosv))
        o.view_angle = view_angle_from_cos(cosv);
    else
        o.view_angle = 0.0f;
    if (fabsf(o.view_angle - c.minimum_view_angle) <= c.view_angle_band) o.ambiguous = 1;

    if (o.view_angle > c.minimum_view_angle) {  // LFE:433
        if (o.curvature < c.thr_surface_curvature) o.label |= LB_SURFACE;
        const float sq2_diff = 0.1f;
        if (o.curvature > c.thr_corner_curvature) {
            if (d[2] <= d[0] && d[2] <= d[4]) {
                if (fabsf(d[2] - d[0]) < sq2_diff * d[2] || fabsf(d[2] - d[4]) < sq2_diff * d[2]) o.label |= LB_CORNER;
            }
        }
    }
    return o;
}

// get_features predicate for one point, LFE:232-265.  Returns bit0 = corner, bit1 = surface, bit2 = full.
LL_HD int select_point(int idx, int type, int label, float depth_sq2, float minimum_idx, float maximum_idx)
{
    if ((float)idx > maximum_idx || (float)idx < minimum_idx) return 0;
    int r = 0;
    if ((type & (PT_000 | PT_NAN | PT_TOO_NEAR)) == 0) {
        if (label & LB_CORNER) {
            if (type != 0) return 0;  // `continue` skips the surface test and the full cloud, LFE:240-241
            if (depth_sq2 < 900.0f) r |= 1;
        }
        if (label & LB_SURFACE) {
            if (depth_sq2 < 1000000.0f) r |= 2;
        }
    }
    return r | 4;
}

}  // namespace ll
