// ll_reg_query.h -- per-query device helpers of the registrar shared by its kernel files (ll_reg_kernels.hip, ll_knn_kernels.hip):
// pose transform of a feature (pointAssociateToMap, point_cloud_registration.hpp:622-661), storing a search result and its reuse
// record, the per-lane and per-wavefront search of one query slot, and the residual-block constants of one slot
// (point_cloud_registration.hpp:259-323, 357-423).  Device only.
#pragma once
#include <hip/hip_runtime.h>

#include "ll_device.h"
#include "ll_knn_coop.h"
#include "ll_reg_core.h"

namespace ll {

#define KB_THREADS 128
#ifndef RS_THREADS
#define RS_THREADS 512  // threads of a solver workgroup (scan_is_compact pads the plane blocks to whole rounds of it)
#endif

// Loads through an explicit global (address space 1) pointer.  Inside a non-inlined device function the compiler cannot
// tell that a pointer taken from RegDev is global and emits flat_load, which also counts on lgkmcnt: the wait in front
// of every LDS flag read then drained the whole software pipeline of record loads (round-2 profile of solver_eval2).
typedef int ll_v4i __attribute__((ext_vector_type(4)));
typedef float ll_v4f __attribute__((ext_vector_type(4)));
typedef double ll_v2d __attribute__((ext_vector_type(2)));
#define LL_AS_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ int4 gload_i4(const int4 *p)
{
    const ll_v4i v = *(const LL_AS_GLOBAL ll_v4i *)p;
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 gload_f4(const float4 *p)
{
    const ll_v4f v = *(const LL_AS_GLOBAL ll_v4f *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// x, y, z of a float4 record as a 12-byte load: with the 16-byte form the register allocator parks another in-flight value
// in the unused w lane, and the write-after-write hazard on that register drains the load pipeline (solver_eval3)
typedef float ll_v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void gload_f3(const float4 *p, float &x, float &y, float &z)
{
    const ll_v3f v = *(const LL_AS_GLOBAL ll_v3f *)p;
    x = v.x;
    y = v.y;
    z = v.z;
}
__device__ __forceinline__ double2 gload_d2(const double2 *p)
{
    const ll_v2d v = *(const LL_AS_GLOBAL ll_v2d *)p;
    return make_double2(v.x, v.y);
}
// ... and through an explicit LDS (address space 3) pointer: a generic pointer into a __shared__ array that crossed a
// function boundary becomes flat_load / flat_store / flat_atomic, which count on both wait counters
#define LL_AS_LDS __attribute__((address_space(3)))
__device__ __forceinline__ int4 lds_load_i4(const int4 *p)
{
    const ll_v4i v = *(const LL_AS_LDS ll_v4i *)p;
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_store_i4(int4 *p, const int4 &v)
{
    ll_v4i w;
    w.x = v.x;
    w.y = v.y;
    w.z = v.z;
    w.w = v.w;
    *(LL_AS_LDS ll_v4i *)p = w;
}
__device__ __forceinline__ void gstore_f64(double *p, double v) { *(LL_AS_GLOBAL double *)p = v; }
__device__ __forceinline__ unsigned char gload_u8(const unsigned char *p) { return *(const LL_AS_GLOBAL unsigned char *)p; }
__device__ __forceinline__ void gstore_u8(unsigned char *p, unsigned char v) { *(LL_AS_GLOBAL unsigned char *)p = v; }
__device__ __forceinline__ void gstore_u16(unsigned short *p, unsigned short v) { *(LL_AS_GLOBAL unsigned short *)p = v; }
__device__ __forceinline__ void gstore_i4(int4 *p, const int4 &v)
{
    ll_v4i w;
    w.x = v.x;
    w.y = v.y;
    w.z = v.z;
    w.w = v.w;
    *(LL_AS_GLOBAL ll_v4i *)p = w;
}
__device__ __forceinline__ f4 gload_pt(const f4 *p)
{
    const ll_v4f v = *(const LL_AS_GLOBAL ll_v4f *)p;
    f4 o;
    o.x = v.x;
    o.y = v.y;
    o.z = v.z;
    o.w = v.w;
    return o;
}
__device__ __forceinline__ double gload_f64(const double *p) { return *(const LL_AS_GLOBAL double *)p; }

// Block constants of one scan (blk_av, 6 * cap doubles) as three arrays of 16-byte pairs: {a0, v0}[cap], {v1, v2}[cap] and
// {a1, a2}[cap] (line slots only; a plane block folds a' into a0 = n'.a').  A plane block is then one float4 and two
// 16-byte loads per lane instead of one float4 and four 8-byte loads.  cap is even and the base 256-byte aligned.
__device__ __forceinline__ void av_store(double *av, int cap, int slot, bool line, const double a[3], const double v[3])
{
    reinterpret_cast<double2 *>(av)[slot] = make_double2(a[0], v[0]);
    reinterpret_cast<double2 *>(av + (size_t)2 * cap)[slot] = make_double2(v[1], v[2]);
    if (line) reinterpret_cast<double2 *>(av + (size_t)4 * cap)[slot] = make_double2(a[1], a[2]);
}
__device__ __forceinline__ void av_load(const double *av, int cap, int slot, bool line, double &a0, double &a1, double &a2, double &v0,
                                        double &v1, double &v2)
{
    const double2 x = gload_d2(reinterpret_cast<const double2 *>(av) + slot);
    const double2 y = gload_d2(reinterpret_cast<const double2 *>(av + (size_t)2 * cap) + slot);
    a0 = x.x;
    v0 = x.y;
    v1 = y.x;
    v2 = y.y;
    a1 = a2 = 0.0;
    if (line) {
        const double2 z = gload_d2(reinterpret_cast<const double2 *>(av + (size_t)4 * cap) + slot);
        a1 = z.x;
        a2 = z.y;
    }
}

// Scans whose plane blocks live in a per-scan PLANE TABLE built by the solver ({n', c} once per distinct neighbour triple; the build stage
// only decides their flags): planes padded to whole 512-thread rounds + lines within LL_TABLE_MAX_BLOCKS -- 120 rounds, the 128-bit
// activity mask of solve_big (ll_reg_big_path.h) and the 16-bit plane ids.  Within FAST_MAX_BLOCKS and without motion deblur a batch
// takes solve_fast3 (64-bit masks, register tiles); launch_reg_solve decides that per batch.  Larger scans and the force_general
// test switch: per-block constants in HBM (solve_general).
#define FAST_MAX_BLOCKS 24576
#define LL_TABLE_MAX_BLOCKS 61440
__device__ __forceinline__ bool scan_is_compact(const RegDev &rd, const RegConst &rc, int b)
{
    const int nC = rd.n_corner[b], nS = rd.n_surf[b];
    const int nSp = (nS + RS_THREADS - 1) / RS_THREADS * RS_THREADS;
    return !rc.force_general && nSp + nC <= LL_TABLE_MAX_BLOCKS;
}

__device__ __forceinline__ void transform_query(const RegState *st, const RegConst &rc, const float4 &f, float pw[3])
{
    pw[0] = pw[1] = pw[2] = NAN;  // non-finite features are skipped (PCR:242-245; surface: defined deviation)
    if (!(ll_isfinite(f.x) && ll_isfinite(f.y) && ll_isfinite(f.z))) return;
    const float sblur = refine_blur(rc.if_motion_deblur, f.w, rc.min_ts, rc.max_ts);  // PCR:247
    if (rc.if_motion_deblur == 0 || (double)sblur == 1.0) {
        point_to_map(st->pose_curr, f.x, f.y, f.z, pw);  // PCR:629
    } else {
        // Rodrigues interpolation, PCR:641-646
        const double s = (double)sblur;
        const double T[3] = {st->inc[4] * (s * 1.0), st->inc[5] * (s * 1.0), st->inc[6] * (s * 1.0)};
        const double th = st->interp_theta * s;
        const double sn = sin(th), cs1 = 1.0 - cos(th);
        const double pc[3] = {(double)f.x, (double)f.y, (double)f.z};
        double inner[3], o[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double rij = ((i == j) ? 1.0 : 0.0) + sn * st->hat[i * 3 + j] + cs1 * st->hat_sq[i * 3 + j];
                acc += rij * pc[j];
            }
            inner[i] = acc + T[i];
        }
        quat_rot(st->pose_last, inner, o);
        pw[0] = (float)(o[0] + st->pose_last[4]);
        pw[1] = (float)(o[1] + st->pose_last[5]);
        pw[2] = (float)(o[2] + st->pose_last[6]);
    }
}

__device__ __forceinline__ float4 load_feature(const RegDev &rd, int b, int kind, int q)
{
    return kind ? rd.surf_feat[(size_t)b * rd.feat_stride_s + q] : rd.corner_feat[(size_t)b * rd.feat_stride_c + q];
}

__device__ __forceinline__ void knn_store(const RegDev &rd, const RegConst &rc, size_t sb, int slot, int kind, int iter, const Knn5 &r)
{
    // 5 neighbours found inside the match radius  <=>  nearestKSearch == 5 and sq_dis[4] < thr (PCR:249-254,353)
    int4 nn;
    nn.w = (r.count == 5) ? 1 : 0;
    nn.x = r.pos[0];
    nn.y = kind ? r.pos[2] : r.pos[1];  // plane: 0, k/2, k-1 (PCR:416-418); line: 0, 1 (PCR:300-301)
    nn.z = r.pos[4];
    rd.nn[sb + slot] = nn;
    if (rc.debug_knn && iter == rc.debug_knn_iter && rd.dbg_idx) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            rd.dbg_idx[(sb + slot) * 5 + k] = (knn5_idx(r, k) == LL_KNN_EMPTY) ? -1 : knn5_idx(r, k);
            rd.dbg_d2[(sb + slot) * 5 + k] = knn5_d2(r, k);
        }
    }
}

__device__ __forceinline__ void ref_store(const RegDev &rd, size_t sb, int slot, const KnnRef &ref)
{
    rd.ref_q[sb + slot] = make_float4(ref.qx, ref.qy, ref.qz, ref.m_strong);
    rd.ref_p[sb + slot] = make_int4(ref.pos[0], ref.pos[1], ref.pos[2], ref.pos[3]);
    rd.ref_s[sb + slot] = make_float2(__int_as_float(ref.pos[4]), ref.m_set);
}

// Set-stable query (re-query state 1): the same five neighbours, re-evaluated and re-sorted at the new position; the
// reuse record moves there with shrunken budgets (knn5_resort).
__device__ __forceinline__ void resort_one(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int b, int slot, int iter)
{
    const int kind = slot >= rd.cap_c ? 1 : 0;
    const size_t sb = (size_t)b * rd.cap;
    const float4 pw = rd.qw[sb + slot];
    const float4 rq = rd.ref_q[sb + slot];
    const float2 rs = rd.ref_s[sb + slot];
    const int4 rp = rd.ref_p[sb + slot];
    KnnRef ref;
    ref.qx = rq.x;
    ref.qy = rq.y;
    ref.qz = rq.z;
    ref.m_strong = rq.w;
    ref.m_set = rs.y;
    ref.pos[0] = rp.x;
    ref.pos[1] = rp.y;
    ref.pos[2] = rp.z;
    ref.pos[3] = rp.w;
    ref.pos[4] = __float_as_int(rs.x);
    const float delta = knn5_ref_delta(ref, pw.x, pw.y, pw.z);  // the value the re-query kernel classified with
    Knn5 r;
    knn5_resort(kind ? gs : gc, ref, delta, pw.x, pw.y, pw.z, kind ? rc.max_d2_plane : rc.max_d2_line, r);
    ref_store(rd, sb, slot, ref);
    knn_store(rd, rc, sb, slot, kind, iter, r);
}

__device__ __forceinline__ void knn_finish(const RegDev &rd, const RegConst &rc, size_t sb, int slot, int kind, int iter, const float4 &pw,
                                           float max_d2, const Knn5 &r)
{
    if (rc.knn_reuse || rc.check_line_pca || rc.check_plane_pca) {  // the PCA checks need all five positions
        KnnRef ref;
        knn5_make_ref(r, pw.x, pw.y, pw.z, max_d2, ref);
        ref_store(rd, sb, slot, ref);
    }
    knn_store(rd, rc, sb, slot, kind, iter, r);
}

__device__ __forceinline__ void knn_one(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int b, int slot, int iter)
{
    const int kind = slot >= rd.cap_c ? 1 : 0;
    const size_t sb = (size_t)b * rd.cap;
    const float4 pw = rd.qw[sb + slot];
    const float max_d2 = kind ? rc.max_d2_plane : rc.max_d2_line;
    Knn5 r;
    knn5_search(kind ? gs : gc, pw.x, pw.y, pw.z, max_d2, r);  // NaN query -> empty
    knn_finish(rd, rc, sb, slot, kind, iter, pw, max_d2, r);
}

// The same for one query per wavefront (ll_knn_coop.h); lane 0 stores.  All 64 lanes call it with the same arguments.
__device__ __forceinline__ void knn_one_coop(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int b, int slot, int iter)
{
    const int kind = slot >= rd.cap_c ? 1 : 0;
    const size_t sb = (size_t)b * rd.cap;
    const float4 pw = rd.qw[sb + slot];
    const float max_d2 = kind ? rc.max_d2_plane : rc.max_d2_line;
    Knn5 r;
    knn5_search_coop(kind ? gs : gc, pw.x, pw.y, pw.z, max_d2, r);
    if ((threadIdx.x & 63) == 0) knn_finish(rd, rc, sb, slot, kind, iter, pw, max_d2, r);
}

// K6b: residual-block constants (fp64) from the neighbours found by K6a.
__device__ __forceinline__ void build_one(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int b, int slot)
{
    const int kind = slot >= rd.cap_c ? 1 : 0;
    const int q = slot - (kind ? rd.cap_c : 0);
    const RegState *st = rd.state + b;
    const size_t sb = (size_t)b * rd.cap;
    const int4 nn = rd.nn[sb + slot];
    unsigned char flag = BLK_NONE;
    bool feature_ok = true;
    if (nn.w && (kind ? rc.check_plane_pca : rc.check_line_pca)) {
        // K7: PCA check of the five neighbours (PCR:259-292, 357-389); their positions live in the reuse record
        const Grid &g5 = kind ? gs : gc;
        const int4 rp = rd.ref_p[sb + slot];
        const int p5[5] = {rp.x, rp.y, rp.z, rp.w, __float_as_int(rd.ref_s[sb + slot].x)};
        double pts[5][3];
        for (int j = 0; j < 5; j++) {
            const f4 pj = g5.pts[p5[j]];
            pts[j][0] = (double)pj.x;
            pts[j][1] = (double)pj.y;
            pts[j][2] = (double)pj.z;
        }
        feature_ok = pca_check(kind, pts);
    }
    if (nn.w && feature_ok) {
        const Grid &g = kind ? gs : gc;
        double a_out[3], v_out[3];
        const f4 p0 = g.pts[nn.x], p1 = g.pts[nn.y];
        const double pa[3] = {(double)p0.x, (double)p0.y, (double)p0.z};
        const double pb[3] = {(double)p1.x, (double)p1.y, (double)p1.z};
        if (kind == 0) {
            if (rc.icp_line && block_line(st->pose_last, pa, pb, a_out, v_out)) flag = BLK_LINE | BLK_ACTIVE | 8;
        } else {
            flag = 8;  // surf_avail counts even when ICP_PLANE == 0 (PCR:425)
            if (rc.icp_plane) {
                const f4 p2 = g.pts[nn.z];
                const double pc[3] = {(double)p2.x, (double)p2.y, (double)p2.z};
                if (scan_is_compact(rd, rc, b)) {
                    // plane-table path: only the flag is decided here; the solver computes {n', c} once per distinct
                    // (nn0, nn2, nn4) triple from rd.nn (solve_fast3)
                    rd.blk_flag0[sb + slot] = plane_degenerate(pa, pb, pc) ? BLK_NONE : (BLK_PLANE | BLK_ACTIVE | 8);
                    return;
                }
                flag = block_plane(st->pose_last, pa, pb, pc, a_out, v_out) ? (BLK_PLANE | BLK_ACTIVE | 8) : BLK_NONE;
            }
        }
        if (flag & BLK_ACTIVE) {
            const float4 f = load_feature(rd, b, kind, q);
            // (a compact scan's plane blocks returned above: their constants live in the solver's plane table)
            const float sblur = rc.if_motion_deblur ? refine_blur(1, f.w, rc.min_ts, rc.max_ts) : 1.0f;
            rd.blk_f[sb + slot] = make_float4(f.x, f.y, f.z, sblur);
            double *av = rd.blk_av + (size_t)b * 6 * rd.cap;
            av_store(av, rd.cap, slot, kind == 0, a_out, v_out);  // line blocks keep the full a'; plane blocks only the scalar n'.a'
        }
    }
    rd.blk_flag0[sb + slot] = flag;  // the solver works on a copy (LDS, or blk_flag in the general path)
}

}  // namespace ll
