// ll_api.hip -- host side of the C ABI declared in include/loam_livox_hip.h.
// Owns device memory, HIP streams and launch order; all arithmetic of the hot path runs in the kernels of
// ll_fe_kernels.hip / ll_map_kernels.hip / ll_reg_kernels.hip.  There is no CPU fallback: every entry point
// fails with an error string when HIP reports no usable device.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/loam_livox_hip.h"
#include "ll_device.h"
#include "ll_reg_core.h"
#include "ll_cellmap.h"
#include "ll_voxel.h"

using namespace ll;

static thread_local std::string g_err;
static int set_err(const char *where, const char *what)
{
    g_err = std::string(where) + ": " + what;
    return -1;
}
#define HC(call)                                                        \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return set_err(#call, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *ll_last_error(void) { return g_err.c_str(); }
extern "C" const char *ll_version(void) { return "loam_livox_hip 0.1 (gfx950)"; }

extern "C" int ll_runtime_hint_hw_queues(int32_t n)
{
    if (n < 1 || n > 64) return set_err("ll_runtime_hint_hw_queues", "n must be in 1 .. 64");
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;  // the caller's environment wins
    char buf[16];
    snprintf(buf, sizeof(buf), "%d", (int)n);
    if (setenv("GPU_MAX_HW_QUEUES", buf, 0) != 0) return set_err("ll_runtime_hint_hw_queues", "setenv failed");
    return 1;
}

template <typename T>
static int dmalloc(T **p, size_t count)
{
    HC(hipMalloc((void **)p, (count > 0 ? count : 1) * sizeof(T)));
    return 0;
}
#define DM(p, count)                        \
    do {                                    \
        if (dmalloc(&(p), (count)) != 0) return -1; \
    } while (0)

static int check_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return set_err("hipGetDeviceCount", "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= count) return set_err("device", "ordinal out of range");
    HC(hipSetDevice(device));
    return 0;
}

// ============================================================================================== extractor

struct ll_fe {
    ll_fe_params prm;
    FeConst fc;
    FeDev dev;
    hipStream_t stream = nullptr;
    hipEvent_t ev_done = nullptr;
    hipEvent_t ev_staged = nullptr;  // behind the last copies out of hp_npts / hp_time0 (the next upload may rewrite the slots after it)
    bool staged_pending = false;
    int max_n_uploaded = 0;
    // sequential time base of Livox_laser (LFE:150-152)
    double first_receive_time = -1.0, last_maximum_time_stamp = 0.0;
    // mutable device arrays (non-const views of dev.*)
    float4 *d_xyzi = nullptr;
    int *d_npts = nullptr;
    double *d_time0 = nullptr;
    std::vector<int> h_npts;
    // page-locked staging of the per-scan point counts and time bases: an asynchronous copy from pageable memory makes the host
    // wait for everything queued on the stream before it -- behind the 98 MB scan upload that was 1.5 ms per step during which no
    // kernel of the batch in flight could be enqueued
    int *hp_npts = nullptr;
    double *hp_time0 = nullptr;
};

extern "C" void ll_fe_default_params(ll_fe_params *p)
{
    memset(p, 0, sizeof(*p));
    p->thr_corner_curvature = 0.05f;   // LFX:152 default
    p->thr_surface_curvature = 0.01f;  // LFX:153
    p->minimum_view_angle = 10.0f;     // LFX:154
    p->livox_min_allow_dis = 0.1f;     // LFX:854
    p->livox_min_sigma = 7e-4f;        // LFX:859
    p->max_fov = 17.0f;                // LFE:143
    p->time_internal_pts = 1.0e-5f;    // LFE:145
    p->device = 0;
    p->max_points = 24000;
    p->max_scans = 1;
    p->piecewise_number = 3;           // LFX:142
}

static FeConst make_fe_const(const ll_fe_params &p)
{
    FeConst c;
    c.thr_corner_curvature = p.thr_corner_curvature;
    c.thr_surface_curvature = p.thr_surface_curvature;
    c.minimum_view_angle = p.minimum_view_angle;
    c.min_dis_sq = p.livox_min_allow_dis * p.livox_min_allow_dis;
    c.min_sigma = p.livox_min_sigma;
    c.max_edge_polar_pos = (float)pow(tan((double)p.max_fov / 57.3) * 1, 2);  // LFE:185
    c.time_internal_pts = p.time_internal_pts;
    // acosf implementations differ by <= 1 ulp; *57.3 and the float store add < 1 ulp more: 8 ulp of the
    // threshold is a generous band
    c.view_angle_band = 8.0f * (nextafterf(fabsf(p.minimum_view_angle) + 1.0f, INFINITY) - (fabsf(p.minimum_view_angle) + 1.0f));
    return c;
}

extern "C" void ll_fe_destroy(ll_fe *h);
static int fe_create_impl(const ll_fe_params *p, ll_fe *h)
{
    h->prm = *p;
    h->fc = make_fe_const(*p);
    const size_t B = p->max_scans, N = p->max_points, BN = B * N;
    FeDev &d = h->dev;
    memset(&d, 0, sizeof(d));
    d.stride = (int)N;
    d.split_cap = (int)(N / 50 + 8);
    // view-angle ambiguity list (ll_fe_resolve): sized for the whole batch -- 1/16 of the points, at least 4096; a batch that
    // still overflows it makes ll_fe_resolve fail instead of silently keeping device-libm labels
    d.ambig_cap = (int)((BN / 16 > 4096 ? BN / 16 : 4096) < 0x7fffffffull ? (BN / 16 > 4096 ? BN / 16 : 4096) : 0x7fffffffull);
    HC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HC(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&h->ev_staged, hipEventDisableTiming));
    DM(h->d_xyzi, BN);
    DM(h->d_npts, B);
    DM(h->d_time0, B);
    d.xyzi = h->d_xyzi;
    d.npts = h->d_npts;
    d.time0 = h->d_time0;
    DM(d.type, BN);
    DM(d.label, BN);
    DM(d.depth2, BN);
    DM(d.polar2, BN);
    DM(d.curv, BN);
    DM(d.view, BN);
    DM(d.tstamp, BN);
    DM(d.polar_angle, BN);
    DM(d.img, BN);
    DM(d.flags, BN);
    DM(d.cand, BN);
    DM(d.split_idx, B * d.split_cap);
    DM(d.petal_first, B * d.split_cap);
    DM(d.petal_last, B * d.split_cap);
    DM(d.run_angle, B * d.split_cap);
    DM(d.info, B);
    DM(d.corner_idx, BN);
    DM(d.surf_idx, BN);
    DM(d.full_idx, BN);
    DM(d.corner_feat, BN);
    DM(d.surf_feat, BN);
    DM(d.n_corner, B);
    DM(d.n_surf, B);
    DM(d.n_full, B);
    DM(d.n_ambig, 1);
    DM(d.ambig_list, d.ambig_cap);
    // On the handle's own stream, ahead of everything it will ever run.  (A hipMemset on the null stream is asynchronous to the host
    // and NOT ordered with a non-blocking stream: issued behind a busy null stream -- a 5 M-point map upload just before -- the
    // zeroing of d_npts landed after the first upload's copy into it, and the first extraction of a fresh handle saw zero points.)
    HC(hipMemsetAsync(d.n_ambig, 0, sizeof(int), h->stream));
    HC(hipMemsetAsync(h->d_npts, 0, B * sizeof(int), h->stream));
    HC(hipMemsetAsync(d.n_corner, 0, B * sizeof(int), h->stream));
    HC(hipMemsetAsync(d.n_surf, 0, B * sizeof(int), h->stream));
    HC(hipMemsetAsync(d.n_full, 0, B * sizeof(int), h->stream));
    HC(hipMemsetAsync(d.info, 0, B * sizeof(FeScanInfo), h->stream));
    h->h_npts.assign(B, 0);
    HC(hipHostMalloc((void **)&h->hp_npts, B * sizeof(int), hipHostMallocDefault));
    HC(hipHostMalloc((void **)&h->hp_time0, B * sizeof(double), hipHostMallocDefault));
    return 0;
}

extern "C" int ll_fe_create(const ll_fe_params *p, ll_fe **out)
{
    if (!p || !out) return set_err("ll_fe_create", "null argument");
    if (p->max_points < 1 || p->max_scans < 1) return set_err("ll_fe_create", "bad capacity");
    if (p->piecewise_number < 1 || p->piecewise_number > LL_MAX_PIECES) return set_err("ll_fe_create", "piecewise_number out of range");
    if (check_device(p->device)) return -1;
    ll_fe *h = new ll_fe();
    if (fe_create_impl(p, h)) {  // a failed allocation half way: release what was built (fields start out null)
        ll_fe_destroy(h);
        return -1;
    }
    *out = h;
    return 0;
}

extern "C" void ll_fe_destroy(ll_fe *h)
{
    if (!h) return;
    (void)hipSetDevice(h->prm.device);
    FeDev &d = h->dev;
    void *ptrs[] = {h->d_xyzi, h->d_npts, h->d_time0, d.type, d.label, d.depth2, d.polar2, d.curv, d.view, d.tstamp,
                    d.polar_angle, d.img, d.flags, d.cand, d.split_idx, d.petal_first, d.petal_last, d.run_angle, d.info,
                    d.corner_idx, d.surf_idx, d.full_idx, d.corner_feat, d.surf_feat, d.n_corner, d.n_surf, d.n_full,
                    d.n_ambig, d.ambig_list};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->hp_npts) (void)hipHostFree(h->hp_npts);
    if (h->hp_time0) (void)hipHostFree(h->hp_time0);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->ev_staged) (void)hipEventDestroy(h->ev_staged);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" void *ll_fe_stream(ll_fe *h) { return h ? (void *)h->stream : nullptr; }
extern "C" int ll_fe_sync(ll_fe *h)
{
    if (!h) return set_err("ll_fe_sync", "null handle");
    HC(hipSetDevice(h->prm.device));
    HC(hipStreamSynchronize(h->stream));
    return 0;
}

static int fe_upload_impl(ll_fe *h, int32_t first_scan, int32_t n_scans, const float *xyzi, int32_t n_points, const double *current_time,
                          bool wait);
extern "C" int ll_fe_upload(ll_fe *h, int32_t first_scan, int32_t n_scans, const float *xyzi, int32_t n_points,
                            const double *current_time)
{
    return fe_upload_impl(h, first_scan, n_scans, xyzi, n_points, current_time, true);
}
extern "C" int ll_fe_upload_async(ll_fe *h, int32_t first_scan, int32_t n_scans, const float *xyzi, int32_t n_points,
                                  const double *current_time)
{
    return fe_upload_impl(h, first_scan, n_scans, xyzi, n_points, current_time, false);
}
static int fe_upload_impl(ll_fe *h, int32_t first_scan, int32_t n_scans, const float *xyzi, int32_t n_points, const double *current_time,
                          bool wait)
{
    if (!h || !xyzi || !current_time) return set_err("ll_fe_upload", "null argument");
    if (first_scan < 0 || n_scans < 0 || first_scan + n_scans > h->prm.max_scans) return set_err("ll_fe_upload", "scan range exceeds max_scans");
    if (n_points < 0 || n_points > h->prm.max_points) return set_err("ll_fe_upload", "n_points exceeds max_points");
    HC(hipSetDevice(h->prm.device));
    const size_t N = h->prm.max_points;
    // The page-locked staging slots of an upload still in flight are not rewritten: wait for the earlier upload's two small
    // copies (an event right behind them) -- not for the stream, which may hold a whole batch of extraction kernels.
    if (h->staged_pending) {
        HC(hipEventSynchronize(h->ev_staged));
        h->staged_pending = false;
    }
    for (int i = 0; i < n_scans; i++) {
        h->h_npts[first_scan + i] = n_points;
        h->hp_npts[first_scan + i] = n_points;
        h->hp_time0[first_scan + i] = current_time[i];
    }
    HC(hipMemcpyAsync(h->d_npts + first_scan, h->hp_npts + first_scan, n_scans * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HC(hipMemcpyAsync(h->d_time0 + first_scan, h->hp_time0 + first_scan, n_scans * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HC(hipEventRecord(h->ev_staged, h->stream));
    h->staged_pending = true;
    if (n_points > 0 && (size_t)n_points == N)  // full slots: one linear copy (a pitched copy of the same bytes does not run at link speed)
        HC(hipMemcpyAsync(h->d_xyzi + (size_t)first_scan * N, xyzi, (size_t)n_scans * N * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    else if (n_points > 0)
        HC(hipMemcpy2DAsync(h->d_xyzi + (size_t)first_scan * N, N * sizeof(float4), xyzi, (size_t)n_points * sizeof(float4),
                            (size_t)n_points * sizeof(float4), n_scans, hipMemcpyHostToDevice, h->stream));
    if (wait) HC(hipStreamSynchronize(h->stream));  // the caller's buffers may be reused right away
    return 0;
}

extern "C" int ll_fe_extract_batch(ll_fe *h, int32_t n_scans)
{
    if (!h) return set_err("ll_fe_extract_batch", "null handle");
    if (n_scans < 1 || n_scans > h->prm.max_scans) return set_err("ll_fe_extract_batch", "n_scans out of range");
    HC(hipSetDevice(h->prm.device));
    int max_n = 0;
    for (int i = 0; i < n_scans; i++) max_n = h->h_npts[i] > max_n ? h->h_npts[i] : max_n;
    HC(hipMemsetAsync(h->dev.n_ambig, 0, sizeof(int), h->stream));
    if (max_n > 0) launch_fe_point(h->dev, h->fc, n_scans, max_n, h->stream);
    launch_fe_split(h->dev, h->prm.piecewise_number, n_scans, h->stream);
    HC(hipGetLastError());
    return 0;
}

// Re-derive, with the host libm acosf the reference uses, the labels of the (very rare) points whose view angle
// fell inside the acosf ambiguity band.  Synchronises.  Returns the number of such points.
static int fe_resolve_ambiguous(ll_fe *h)
{
    int n_amb = 0;
    HC(hipMemcpyAsync(&n_amb, h->dev.n_ambig, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HC(hipStreamSynchronize(h->stream));
    if (n_amb <= 0) return 0;
    if (n_amb > h->dev.ambig_cap)
        return set_err("ll_fe_resolve", "more points inside the view-angle ambiguity band than the list holds (degenerate minimum_view_angle?)");
    const int n_list = n_amb;
    std::vector<int2> list(n_list);
    HC(hipMemcpy(list.data(), h->dev.ambig_list, n_list * sizeof(int2), hipMemcpyDeviceToHost));
    const size_t N = h->prm.max_points;
    // the re-derived labels are applied behind the loop: copies queued on the handle's stream from staging that outlives them, one wait
    // (a pair of copies and a stream drain per point made a batch with a few thousand flagged points thousands of round trips)
    std::vector<int> fix_label(n_list);
    std::vector<float> fix_view(n_list);
    std::vector<size_t> fix_at;
    fix_at.reserve(n_list);
    for (const int2 &e : list) {
        const int b = e.x, i = e.y, n = h->h_npts[b];
        if (i < 2 || i >= n - 2) continue;
        float4 raw[5];
        HC(hipMemcpy(raw, h->d_xyzi + (size_t)b * N + i - 2, sizeof(raw), hipMemcpyDeviceToHost));
        float p[5][3], d[5];
        int t[5];
        for (int k = 0; k < 5; k++) {
            const PointOwn o = point_own(raw[k].x, raw[k].y, raw[k].z, raw[k].w, i - 2 + k, h->fc);
            p[k][0] = raw[k].x;
            p[k][1] = raw[k].y;
            p[k][2] = raw[k].z;
            t[k] = o.type_self;
            d[k] = o.depth_sq2;
        }
        const LabelOut lo = point_label(p, t, d, h->fc);  // host build: glibc acosf
        fix_label[fix_at.size()] = lo.label;
        fix_view[fix_at.size()] = lo.view_angle;
        fix_at.push_back((size_t)b * N + i);
    }
    // on the handle's stream and waited for: the selection kernel that reads these runs on that stream, and a null-stream copy from
    // pageable memory is not ordered with it
    for (size_t k = 0; k < fix_at.size(); k++) {
        HC(hipMemcpyAsync(h->dev.label + fix_at[k], &fix_label[k], sizeof(int), hipMemcpyHostToDevice, h->stream));
        HC(hipMemcpyAsync(h->dev.view + fix_at[k], &fix_view[k], sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    if (!fix_at.empty()) HC(hipStreamSynchronize(h->stream));
    return n_amb;
}

extern "C" int ll_fe_counts(ll_fe *h, int32_t n_scans, int32_t *n_corner, int32_t *n_surf, int32_t *n_full, int32_t *n_ambiguous)
{
    if (!h) return set_err("ll_fe_counts", "null handle");
    if (n_scans < 1 || n_scans > h->prm.max_scans) return set_err("ll_fe_counts", "n_scans out of range");
    HC(hipSetDevice(h->prm.device));
    HC(hipStreamSynchronize(h->stream));
    if (n_corner) HC(hipMemcpy(n_corner, h->dev.n_corner, n_scans * sizeof(int), hipMemcpyDeviceToHost));
    if (n_surf) HC(hipMemcpy(n_surf, h->dev.n_surf, n_scans * sizeof(int), hipMemcpyDeviceToHost));
    if (n_full) HC(hipMemcpy(n_full, h->dev.n_full, n_scans * sizeof(int), hipMemcpyDeviceToHost));
    if (n_ambiguous) HC(hipMemcpy(n_ambiguous, h->dev.n_ambig, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ll_fe_resolve(ll_fe *h)
{
    if (!h) return set_err("ll_fe_resolve", "null handle");
    HC(hipSetDevice(h->prm.device));
    return fe_resolve_ambiguous(h);
}

extern "C" int ll_fe_extract(ll_fe *h, const float *xyzi, int32_t n, double time_stamp, int32_t *n_petal_clouds)
{
    if (!h || (!xyzi && n > 0)) return set_err("ll_fe_extract", "null argument");
    if (time_stamp < 0.0) return set_err("ll_fe_extract", "time_stamp must be >= 0 (assert at livox_feature_extractor.hpp:724)");
    if (n < 0 || n > h->prm.max_points) return set_err("ll_fe_extract", "n exceeds max_points");
    // LFE:724-736
    double current_time;
    if (time_stamp <= 0.0000001 || (time_stamp < h->last_maximum_time_stamp))
        current_time = h->last_maximum_time_stamp;
    else
        current_time = time_stamp - h->first_receive_time;
    if (h->first_receive_time <= 0) h->first_receive_time = time_stamp;
    static const float dummy[4] = {0, 0, 0, 0};
    if (ll_fe_upload(h, 0, 1, n > 0 ? xyzi : dummy, n, &current_time)) return -1;
    if (ll_fe_extract_batch(h, 1)) return -1;
    if (fe_resolve_ambiguous(h) < 0) return -1;
    if (n > 0) h->last_maximum_time_stamp = (double)point_time_stamp(current_time, n - 1, h->prm.time_internal_pts);  // LFE:482
    FeScanInfo info;
    HC(hipMemcpy(&info, h->dev.info, sizeof(info), hipMemcpyDeviceToHost));
    if (n_petal_clouds) *n_petal_clouds = info.n_petal_clouds;
    return 0;
}

#define D2H_OPT(dst, src, count, type)                                                              \
    do {                                                                                            \
        if (dst) HC(hipMemcpy(dst, src, (size_t)(count) * sizeof(type), hipMemcpyDeviceToHost));    \
    } while (0)

extern "C" int ll_fe_labels(ll_fe *h, int32_t scan, int32_t *pt_type, int32_t *pt_label, float *depth_sq2, float *polar_dis_sq2,
                            float *curvature, float *view_angle, float *time_stamp, float *polar_angle)
{
    if (!h) return set_err("ll_fe_labels", "null handle");
    if (scan < 0 || scan >= h->prm.max_scans) return set_err("ll_fe_labels", "scan out of range");
    HC(hipSetDevice(h->prm.device));
    HC(hipStreamSynchronize(h->stream));
    const size_t off = (size_t)scan * h->prm.max_points;
    const int n = h->h_npts[scan];
    D2H_OPT(pt_type, h->dev.type + off, n, int);
    D2H_OPT(pt_label, h->dev.label + off, n, int);
    D2H_OPT(depth_sq2, h->dev.depth2 + off, n, float);
    D2H_OPT(polar_dis_sq2, h->dev.polar2 + off, n, float);
    D2H_OPT(curvature, h->dev.curv + off, n, float);
    D2H_OPT(view_angle, h->dev.view + off, n, float);
    D2H_OPT(time_stamp, h->dev.tstamp + off, n, float);
    D2H_OPT(polar_angle, h->dev.polar_angle + off, n, float);
    return 0;
}

extern "C" int ll_fe_splits(ll_fe *h, int32_t scan, int32_t *split_idx, int32_t *n_split, int32_t *clutter_size,
                            int32_t *n_petal_clouds, int32_t *first_idx, int32_t *last_idx, float *piece_start, float *piece_end)
{
    if (!h) return set_err("ll_fe_splits", "null handle");
    if (scan < 0 || scan >= h->prm.max_scans) return set_err("ll_fe_splits", "scan out of range");
    HC(hipSetDevice(h->prm.device));
    HC(hipStreamSynchronize(h->stream));
    FeScanInfo info;
    HC(hipMemcpy(&info, h->dev.info + scan, sizeof(info), hipMemcpyDeviceToHost));
    if (n_split) *n_split = info.n_split;
    if (clutter_size) *clutter_size = info.clutter_size;
    if (n_petal_clouds) *n_petal_clouds = info.n_petal_clouds;
    const size_t off = (size_t)scan * h->dev.split_cap;
    D2H_OPT(split_idx, h->dev.split_idx + off, info.n_split, int);
    D2H_OPT(first_idx, h->dev.petal_first + off, info.n_petal_clouds, int);
    D2H_OPT(last_idx, h->dev.petal_last + off, info.n_petal_clouds, int);
    for (int i = 0; i < h->prm.piecewise_number; i++) {
        if (piece_start) piece_start[i] = info.piece_start[i];
        if (piece_end) piece_end[i] = info.piece_end[i];
    }
    return 0;
}

extern "C" int ll_fe_select_batch(ll_fe *h, int32_t n_scans, int32_t piece, float minimum_blur, float maximum_blur)
{
    if (!h) return set_err("ll_fe_select_batch", "null handle");
    if (n_scans < 1 || n_scans > h->prm.max_scans) return set_err("ll_fe_select_batch", "n_scans out of range");
    if (piece >= h->prm.piecewise_number) return set_err("ll_fe_select_batch", "piece out of range");
    HC(hipSetDevice(h->prm.device));
    launch_fe_select(h->dev, n_scans, piece, minimum_blur, maximum_blur, h->stream);
    HC(hipGetLastError());
    return 0;
}

extern "C" int ll_fe_select(ll_fe *h, float minimum_blur, float maximum_blur, int32_t *corner_idx, int32_t *n_corner,
                            int32_t *surf_idx, int32_t *n_surf, int32_t *full_idx, int32_t *n_full, float *corner_xyzi,
                            float *surf_xyzi)
{
    if (ll_fe_select_batch(h, 1, -1, minimum_blur, maximum_blur)) return -1;
    HC(hipStreamSynchronize(h->stream));
    int nc = 0, ns = 0, nf = 0;
    HC(hipMemcpy(&nc, h->dev.n_corner, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&ns, h->dev.n_surf, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&nf, h->dev.n_full, sizeof(int), hipMemcpyDeviceToHost));
    if (n_corner) *n_corner = nc;
    if (n_surf) *n_surf = ns;
    if (n_full) *n_full = nf;
    D2H_OPT(corner_idx, h->dev.corner_idx, nc, int);
    D2H_OPT(surf_idx, h->dev.surf_idx, ns, int);
    D2H_OPT(full_idx, h->dev.full_idx, nf, int);
    D2H_OPT(corner_xyzi, h->dev.corner_feat, nc, float4);
    D2H_OPT(surf_xyzi, h->dev.surf_feat, ns, float4);
    return 0;
}

// the selection ll_fe_select_batch left in slot `scan` (the batched counterpart of ll_fe_select's downloads)
extern "C" int ll_fe_selection(ll_fe *h, int32_t scan, int32_t *corner_idx, int32_t *n_corner, int32_t *surf_idx, int32_t *n_surf,
                               int32_t *full_idx, int32_t *n_full, float *corner_xyzi, float *surf_xyzi)
{
    if (!h) return set_err("ll_fe_selection", "null handle");
    if (scan < 0 || scan >= h->prm.max_scans) return set_err("ll_fe_selection", "scan out of range");
    HC(hipSetDevice(h->prm.device));
    HC(hipStreamSynchronize(h->stream));
    int nc = 0, ns = 0, nf = 0;
    HC(hipMemcpy(&nc, h->dev.n_corner + scan, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&ns, h->dev.n_surf + scan, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&nf, h->dev.n_full + scan, sizeof(int), hipMemcpyDeviceToHost));
    if (n_corner) *n_corner = nc;
    if (n_surf) *n_surf = ns;
    if (n_full) *n_full = nf;
    const size_t off = (size_t)scan * h->dev.stride;
    D2H_OPT(corner_idx, h->dev.corner_idx + off, nc, int);
    D2H_OPT(surf_idx, h->dev.surf_idx + off, ns, int);
    D2H_OPT(full_idx, h->dev.full_idx + off, nf, int);
    D2H_OPT(corner_xyzi, h->dev.corner_feat + off, nc, float4);
    D2H_OPT(surf_xyzi, h->dev.surf_feat + off, ns, float4);
    return 0;
}

// ============================================================================================== map

// A search structure is an IMMUTABLE snapshot once published (the match buffer is refreshed on one thread,
// laser_mapping.hpp:568, while process_new_scan threads register against it, :1737-1742): ll_map_upload /
// ll_history_refresh* build the next grid in buffers nobody else sees and swap the published pointer under the mutex;
// every solve pins the snapshots it was launched with until it has been collected.  A snapshot that only the pool still
// references is recycled for the next build (its buffers keep their capacity).
struct MapSnap {
    MapKind mk;
    int device = 0;
    ~MapSnap()
    {
        (void)hipSetDevice(device);
        map_free(mk);
    }
};
struct ll_map {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::shared_ptr<MapSnap> cur[2];
    std::vector<std::shared_ptr<MapSnap>> pool[2];
    int64_t generation[2] = {0, 0};  // snapshots published so far per kind (ll_map_generation)
};

static std::shared_ptr<MapSnap> map_pin(const ll_map *cm, int kind)
{
    ll_map *m = const_cast<ll_map *>(cm);
    std::lock_guard<std::mutex> lk(m->mu);
    return m->cur[kind];
}
static std::shared_ptr<MapSnap> map_build_target(ll_map *m, int kind)
{
    std::lock_guard<std::mutex> lk(m->mu);
    for (auto &s : m->pool[kind])
        if (s.use_count() == 1) return s;  // referenced by the pool only: not published, not pinned
    std::shared_ptr<MapSnap> s = std::make_shared<MapSnap>();
    s->device = m->device;
    m->pool[kind].push_back(s);
    return s;
}
// returns the generation number this publication got (read under the mutex: a concurrent publisher cannot slip in between)
static int64_t map_publish(ll_map *m, int kind, const std::shared_ptr<MapSnap> &s)
{
    std::lock_guard<std::mutex> lk(m->mu);
    m->cur[kind] = s;
    return ++m->generation[kind];
}
// builds the grid of `n` device-resident points into a fresh snapshot and publishes it
static int map_rebuild(ll_map *m, int kind, const float *d_raw, int stride, int64_t n, float cell, hipStream_t s, const char **err,
                       int64_t *generation = nullptr)
{
    std::shared_ptr<MapSnap> t = map_build_target(m, kind);
    if (map_build(t->mk, d_raw, stride, n, cell, s, err)) return -1;  // returns with the stream drained
    const int64_t g = map_publish(m, kind, t);
    if (generation) *generation = g;
    return 0;
}

extern "C" int ll_map_create(int32_t device, ll_map **out)
{
    if (!out) return set_err("ll_map_create", "null argument");
    if (check_device(device)) return -1;
    ll_map *m = new ll_map();
    m->device = device;
    HC(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    *out = m;
    return 0;
}

extern "C" void ll_map_destroy(ll_map *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (int k = 0; k < 2; k++) {
        m->cur[k].reset();
        m->pool[k].clear();  // snapshots still pinned by a registrar die with its pin
    }
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int ll_map_upload(ll_map *m, int32_t kind, const float *xyz, int32_t stride_floats, int64_t n, float cell_size)
{
    return ll_map_upload_gen(m, kind, xyz, stride_floats, n, cell_size, nullptr);
}

extern "C" int ll_map_upload_gen(ll_map *m, int32_t kind, const float *xyz, int32_t stride_floats, int64_t n, float cell_size, int64_t *generation)
{
    if (!m || (!xyz && n > 0)) return set_err("ll_map_upload", "null argument");
    if (kind != LL_MAP_CORNER && kind != LL_MAP_SURF) return set_err("ll_map_upload", "bad kind");
    if (stride_floats < 3) return set_err("ll_map_upload", "stride_floats must be >= 3");
    if (n < 0 || n > 0x7fffffffLL) return set_err("ll_map_upload", "point count out of range");
    HC(hipSetDevice(m->device));
    if (!(cell_size > 0.f)) cell_size = (kind == LL_MAP_CORNER) ? 1.45f : 0.6f;  // corner: just above the line match radius sqrt(2) m (PCR:89)
    float *d_raw = nullptr;
    const size_t bytes = (size_t)(n > 0 ? n : 1) * stride_floats * sizeof(float);
    HC(hipMalloc(&d_raw, bytes));
    if (n > 0 && hipMemcpyAsync(d_raw, xyz, (size_t)n * stride_floats * sizeof(float), hipMemcpyHostToDevice, m->stream) != hipSuccess) {
        (void)hipFree(d_raw);
        return set_err("ll_map_upload", "host to device copy failed");
    }
    const char *err = nullptr;
    const int rc = map_rebuild(m, kind, d_raw, stride_floats, n, cell_size, m->stream, &err, generation);
    (void)hipFree(d_raw);
    if (rc != 0) return set_err("map_build", err ? err : "failed");
    return 0;
}

extern "C" int ll_map_to_f16(ll_map *m, int32_t kind)
{
    if (!m || kind < 0 || kind > 1) return set_err("ll_map_to_f16", "bad argument");
    HC(hipSetDevice(m->device));
    std::shared_ptr<MapSnap> snap = map_pin(m, kind);
    if (!snap) return set_err("ll_map_to_f16", "map kind not uploaded");
    // converts the published snapshot in place (C5 experiment path): the caller must not have a registration in flight
    if (snap.use_count() > 3) return set_err("ll_map_to_f16", "the snapshot is pinned by a registration in flight");
    const char *err = nullptr;
    if (map_to_f16(snap->mk, m->stream, &err)) return set_err("ll_map_to_f16", err ? err : "failed");
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->generation[kind]++;  // converted in place: not the structure a host-side cache uploaded any more
    }
    return 0;
}

extern "C" int ll_map_dequantized(ll_map *m, int32_t kind, float *xyz, int64_t capacity_points)
{
    if (!m || kind < 0 || kind > 1 || !xyz) return set_err("ll_map_dequantized", "bad argument");
    std::shared_ptr<MapSnap> snap = map_pin(m, kind);
    if (!snap) return set_err("ll_map_dequantized", "map kind not uploaded");
    const MapKind &mk = snap->mk;
    if (capacity_points < mk.n) return set_err("ll_map_dequantized", "buffer too small");
    HC(hipSetDevice(m->device));
    float *d_out = nullptr;
    const size_t bytes = (size_t)(mk.n > 0 ? mk.n : 1) * 3 * sizeof(float);
    HC(hipMalloc(&d_out, bytes));
    HC(hipMemsetAsync(d_out, 0xff, bytes, m->stream));  // NaN pattern for points that were dropped (non-finite input)
    const char *err = nullptr;
    const int rc = map_f16_dequant(mk, d_out, m->stream, &err);
    if (rc == 0 && mk.n > 0) HC(hipMemcpy(xyz, d_out, (size_t)mk.n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_out);
    if (rc) return set_err("ll_map_dequantized", err ? err : "failed");
    return 0;
}

extern "C" int64_t ll_map_generation(const ll_map *cm, int32_t kind)
{
    if (!cm || kind < 0 || kind > 1) return -1;
    ll_map *m = const_cast<ll_map *>(cm);
    std::lock_guard<std::mutex> lk(m->mu);
    return m->generation[kind];
}

extern "C" int64_t ll_map_size(const ll_map *m, int32_t kind)
{
    if (!m || kind < 0 || kind > 1) return -1;
    std::shared_ptr<MapSnap> snap = map_pin(m, kind);
    return snap ? snap->mk.n : 0;
}

// cells of the published grid of `kind` (its cell table holds one 32-bit start per cell + 1): the map's footprint in HBM is
// ll_map_size() records + this table (bench_c5.py's algorithmic bytes)
extern "C" int64_t ll_map_cells(const ll_map *m, int32_t kind)
{
    if (!m || kind < 0 || kind > 1) return -1;
    std::shared_ptr<MapSnap> snap = map_pin(m, kind);
    return snap ? (int64_t)snap->mk.ncell : 0;
}

extern "C" int ll_map_knn5(ll_map *m, int32_t kind, const float *queries_xyz, int32_t n_queries, float max_sq_dis, int32_t *idx5,
                           float *sq_dis5)
{
    if (!m || !queries_xyz || !idx5 || !sq_dis5) return set_err("ll_map_knn5", "null argument");
    std::shared_ptr<MapSnap> snap = (kind < 0 || kind > 1) ? nullptr : map_pin(m, kind);
    if (!snap || (!snap->mk.pts && !snap->mk.pts16)) return set_err("ll_map_knn5", "map kind not uploaded");
    if (n_queries <= 0) return 0;
    HC(hipSetDevice(m->device));
    float *d_q = nullptr, *d_d2 = nullptr;
    int *d_idx = nullptr;
    DM(d_q, (size_t)n_queries * 3);
    DM(d_d2, (size_t)n_queries * 5);
    DM(d_idx, (size_t)n_queries * 5);
    HC(hipMemcpyAsync(d_q, queries_xyz, (size_t)n_queries * 3 * sizeof(float), hipMemcpyHostToDevice, m->stream));
    launch_knn5(snap->mk.grid, d_q, n_queries, max_sq_dis, d_idx, d_d2, m->stream);
    HC(hipGetLastError());
    HC(hipMemcpyAsync(idx5, d_idx, (size_t)n_queries * 5 * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HC(hipMemcpyAsync(sq_dis5, d_d2, (size_t)n_queries * 5 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HC(hipStreamSynchronize(m->stream));
    (void)hipFree(d_q);
    (void)hipFree(d_d2);
    (void)hipFree(d_idx);
    return 0;
}

// The same search with queries and results RESIDENT on the device (pointers from hipMalloc / a torch tensor's data_ptr): nothing crosses
// PCIe, and *kernel_ms (optional) is the search kernel's duration from HIP events on the map's stream -- the figure a roofline needs
// (bench_c5.py).  Synchronous.
extern "C" int ll_map_knn5_device(ll_map *m, int32_t kind, const float *dev_queries_xyz, int64_t n_queries, float max_sq_dis, int32_t *dev_idx5,
                                  float *dev_sq_dis5, float *kernel_ms)
{
    if (!m || !dev_queries_xyz || !dev_idx5 || !dev_sq_dis5) return set_err("ll_map_knn5_device", "null argument");
    if (n_queries < 0 || n_queries > 0x7fffffffLL / 5) return set_err("ll_map_knn5_device", "n_queries out of range");
    std::shared_ptr<MapSnap> snap = (kind < 0 || kind > 1) ? nullptr : map_pin(m, kind);
    if (!snap || (!snap->mk.pts && !snap->mk.pts16)) return set_err("ll_map_knn5_device", "map kind not uploaded");
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_queries == 0) return 0;
    HC(hipSetDevice(m->device));
    struct Events {  // (destroyed on every return path)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    HC(hipEventCreate(&ev.e0));
    HC(hipEventCreate(&ev.e1));
    HC(hipDeviceSynchronize());  // (the caller's buffers may have been written on another stream)
    HC(hipEventRecord(ev.e0, m->stream));
    launch_knn5(snap->mk.grid, dev_queries_xyz, (int)n_queries, max_sq_dis, dev_idx5, dev_sq_dis5, m->stream);
    HC(hipEventRecord(ev.e1, m->stream));
    HC(hipGetLastError());
    HC(hipStreamSynchronize(m->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev.e0, ev.e1);
    if (kernel_ms) *kernel_ms = ms;
    return 0;
}

// ============================================================================================== registrar

struct ll_reg {
    int device = 0;
    int max_scans = 0, max_feat = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_wait = nullptr;
    RegDev dev;
    RegConst rc;
    // own feature storage (host-provided features)
    float4 *d_corner = nullptr, *d_surf = nullptr;
    int *d_nc = nullptr, *d_ns = nullptr;
    std::vector<RegState> h_state;
    std::vector<int> h_nc, h_ns;
    std::shared_ptr<MapSnap> pinned[2];  // map snapshots of the solve in flight (released once it has been collected)
    int debug = 0, profiling = 0;
    int debug_knn_iter = 0;  // ll_reg_set_debug_knn_iteration
    int last_n_scans = 0, last_gated = 0;
    // profiling
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> ev_class;
    float prof_ms[3] = {0, 0, 0};
    int prof_launches[3] = {0, 0, 0};
    double *d_pose_tmp = nullptr;
    int uploaded_scans = 0;  // scans covered by the last ll_reg_upload_features
};

extern "C" void ll_reg_default_params(ll_reg_params *p)
{
    memset(p, 0, sizeof(*p));
    p->if_motion_deblur = 0;              // PCR:60
    p->icp_max_iterations = 20;           // PCR:89
    p->ceres_max_iterations = 100;        // PCR:90
    p->ceres_prerun_times = 2;            // PCR:91
    p->icp_line = 1;                      // PCR:50
    p->icp_plane = 1;                     // PCR:49
    p->if_line_feature_check = 0;         // PCR:46
    p->if_plane_feature_check = 0;        // PCR:48
    p->subsample_seed = 0;                // strict: no sub-sampling, too many features is an error
    p->current_frame_index = 101;
    p->mapping_init_accumulate_frames = 100;  // PCR:84
    p->maximum_allow_residual_block = 100000; // PCR:103
    p->force_all_iterations = 0;
    p->maximum_dis_line_for_match = 2.0;  // PCR:65
    p->maximum_dis_plane_for_match = 50.0; // PCR:64
    p->huber_a = 0.1;                     // PCR:220
    p->inliner_dis = 0.02;                // PCR:97
    p->inlier_ratio = 0.80;               // PCR:98
    p->minimum_icp_R_diff = 0.01;         // PCR:94
    p->minimum_icp_T_diff = 0.01;         // PCR:95
    p->para_max_angular_rate = 200.0f / 50.0f; // PCR:86
    p->para_max_speed = 100.0f / 50.0f;   // PCR:87
    p->max_final_cost = 100.0f;           // PCR:88
    p->minimum_pt_time_stamp = 0.f;       // PCR:92
    p->maximum_pt_time_stamp = 1.0f;      // PCR:93
}

extern "C" void ll_reg_destroy(ll_reg *r);
static int reg_create_impl(int32_t device, int32_t max_scans, int32_t max_features_per_scan, ll_reg *r)
{
    r->device = device;
    r->max_scans = max_scans;
    r->max_feat = max_features_per_scan;
    HC(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    HC(hipEventCreateWithFlags(&r->ev_wait, hipEventDisableTiming));
    RegDev &d = r->dev;
    memset(&d, 0, sizeof(d));
    const size_t B = max_scans, F = max_features_per_scan;
    d.cap_c = (int)F;
    d.cap_s = (int)F;
    d.cap = d.cap_c + d.cap_s;
    int hc = 1;
    while (hc < 2 * d.cap) hc <<= 1;
    d.hash_cap = hc;
    DM(d.state, B);
    DM(d.blk_f, B * d.cap);
    DM(d.blk_av, B * 6 * d.cap);
    DM(d.blk_id, B * d.cap_s);
    {
        const int lim = d.cap_s < 61440 ? d.cap_s : 61440;  // LL_TABLE_MAX_BLOCKS: larger scans never take a plane-table path
        d.tab_cap = (lim + 4095) / 4096 * 4096;             // (<= 61440: private entries count down from tab_cap - 1, below the 16-bit sentinels)
    }
    DM(d.pl_tab, B * (size_t)d.tab_cap * 2);
    DM(d.blk_flag, B * d.cap);
    DM(d.nn, B * d.cap);
    DM(d.qperm, B * d.cap_s);
    DM(d.qsorted, B * d.cap_s);
    DM(d.qperm_c, B * LL_QSORT_CORNER_MAX);
    DM(d.qw, B * d.cap);
    DM(d.ref_q, B * d.cap);
    DM(d.ref_p, B * d.cap);
    DM(d.ref_s, B * d.cap);
    DM(d.blk_flag0, B * d.cap);
    DM(d.work_search, B * d.cap);
    DM(d.work_build, B * d.cap);
    d.n_chunks = (int)((F + 255) / 256);  // must match RQ_THREADS in ll_reg_kernels.hip
    DM(d.work_cnt, B * 4);
    DM(d.work_off, 3 * 2049);  // 3 prefix tables x (RL_MAX_SEG + 1), ll_reg_kernels.hip
    DM(d.grp_ctl, 2 * B + 1);
    DM(d.solve_order, B);
    DM(d.grp_part, B * 2 * LL_GRP * 28);
    DM(d.grp_xch, B * 2 * LL_GRP * 56);
    DM(d.blk_l1, B * d.cap);
    DM(d.hash, B * (size_t)d.hash_cap);
    DM(r->d_corner, B * F);
    DM(r->d_surf, B * F);
    DM(r->d_nc, B);
    DM(r->d_ns, B);
    DM(r->d_pose_tmp, 8);
    HC(hipMemsetAsync(d.blk_flag, 0, B * d.cap, r->stream));  // (on the registrar's stream: a null-stream memset is not ordered with it)
    r->h_state.resize(B);
    r->h_nc.assign(B, 0);
    r->h_ns.assign(B, 0);
    return 0;
}

extern "C" int ll_reg_create(int32_t device, int32_t max_scans, int32_t max_features_per_scan, ll_reg **out)
{
    if (!out) return set_err("ll_reg_create", "null argument");
    if (max_scans < 1 || max_features_per_scan < 1) return set_err("ll_reg_create", "bad capacity");
    if ((int64_t)max_scans * 2 * max_features_per_scan >= 0x7fffffffLL)
        return set_err("ll_reg_create", "max_scans x 2 x max_features_per_scan must stay below 2^31 (work-list entries are 32-bit)");
    if (check_device(device)) return -1;
    ll_reg *r = new ll_reg();
    if (reg_create_impl(device, max_scans, max_features_per_scan, r)) {
        ll_reg_destroy(r);
        return -1;
    }
    *out = r;
    return 0;
}

extern "C" void ll_reg_destroy(ll_reg *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    RegDev &d = r->dev;
    void *ptrs[] = {d.state, d.blk_f, d.blk_av, d.blk_id, d.pl_tab, d.blk_flag, d.nn, d.qperm, d.qsorted, d.qperm_c, d.qw, d.ref_q, d.ref_p, d.ref_s, d.blk_flag0, d.work_search, d.work_build, d.work_cnt, d.work_off, d.grp_ctl, d.solve_order, d.grp_part, d.grp_xch, d.blk_l1, d.hash, d.dbg_idx, d.dbg_d2,
                    r->d_corner, r->d_surf, r->d_nc, r->d_ns, r->d_pose_tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : r->ev) (void)hipEventDestroy(e);
    if (r->ev_wait) (void)hipEventDestroy(r->ev_wait);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

extern "C" void *ll_reg_stream(ll_reg *r) { return r ? (void *)r->stream : nullptr; }

extern "C" int ll_reg_set_debug(ll_reg *r, int32_t enable)
{
    if (!r) return set_err("ll_reg_set_debug", "null handle");
    HC(hipSetDevice(r->device));
    if (enable & (16 | 64))
        return set_err("ll_reg_set_debug", "bits 4 and 6 selected the round-1 / round-2 solver forms, which were retired in round 6 (the plane-table and the "
                                           "general path are the only solver forms)");
    r->debug = enable;
    if ((enable & 1) && !r->dev.dbg_idx) {
        DM(r->dev.dbg_idx, (size_t)r->max_scans * r->dev.cap * 5);
        DM(r->dev.dbg_d2, (size_t)r->max_scans * r->dev.cap * 5);
    }
    return 0;
}

extern "C" int ll_reg_set_debug_knn_iteration(ll_reg *r, int32_t icp_iteration)
{
    if (!r) return set_err("ll_reg_set_debug_knn_iteration", "null handle");
    if (icp_iteration < 0) return set_err("ll_reg_set_debug_knn_iteration", "negative ICP iteration");
    r->debug_knn_iter = icp_iteration;
    return 0;
}

extern "C" int ll_reg_set_profiling(ll_reg *r, int32_t enable)
{
    if (!r) return set_err("ll_reg_set_profiling", "null handle");
    r->profiling = enable ? 1 : 0;
    return 0;
}

static int make_reg_const(const ll_reg_params *p, int debug, RegConst *c)
{
    debug &= ~(16 | 64);  // (retired solver forms; LL_DEBUG_OR cannot ask for them either)
    memset(c, 0, sizeof(*c));
    c->if_motion_deblur = p->if_motion_deblur;
    c->icp_max_iterations = p->icp_max_iterations;
    c->ceres_max_iterations = p->ceres_max_iterations;
    c->ceres_prerun_times = p->ceres_prerun_times;
    c->icp_line = p->icp_line;
    c->icp_plane = p->icp_plane;
    c->check_line_pca = p->if_line_feature_check;
    c->check_plane_pca = p->if_plane_feature_check;
    c->subsample_seed = (unsigned int)p->subsample_seed;
    c->max_blocks = p->maximum_allow_residual_block;
    c->force_all_iterations = p->force_all_iterations;
    c->debug_knn = debug & 1;
    c->force_general = (debug & 2) ? 1 : 0;
    c->knn_reuse = (debug & 4) ? 0 : 1;
    c->knn_reuse_from = (debug & 8) ? 1 : 2;  // bit 3: also try reuse at ICP iteration 1 (test coverage)
    c->solve_group = (debug & 32) ? 1 : 0;    // bit 5: never spread a scan over a group of workgroups (A/B); 0 = decide per batch size
    c->test_group_abort = (debug & 128) ? 1 : 0;  // bit 7: the grouped solver gives up at once (exercises the abort / reject path)
    c->knn_coop = (debug & 256) ? 0 : 1;  // bit 8: corner searches per lane everywhere instead of per wavefront where few (A/B, ll_knn_coop.h)
    c->knn_tile_last_sort = (debug & 8192) ? 0 : ((debug & 16384) ? 2 : 1);  // bits 13 / 14: A/B of the re-sort schedule (sort at iteration 0 only / at 0, 1, 2)
    c->no_solve_order = (debug & 262144) ? 1 : 0;  // bit 18: the small solver's workgroups in scan order (A/B)
    c->no_small_solver = (debug & 32768) ? 1 : 0;  // bit 15: small scans on the 512-thread solver too (A/B)
    c->small_waves = ((debug & 65536) && (debug & 131072)) ? 2 : ((debug & 65536) ? 1 : ((debug & 131072) ? 4 : 0));  // bits 16 / 17: the small solver with one / four wavefronts per scan whatever the batch size (tests)
    c->no_line_cache = (debug & 4096) ? 1 : 0;  // bit 12: no LDS copy of the line blocks in the solver (A/B)
    c->knn_tile = (debug & 512) ? 0 : ((debug & 1024) ? 1 : 2);  // bit 9: no tile search of the surface queries (A/B, ll_knn_tile.h); bit 10: tile
                                                                 // search only where all queries are searched, the reuse machinery for the rest
    c->max_d2_line_d = p->maximum_dis_line_for_match;
    c->max_d2_plane_d = p->maximum_dis_plane_for_match;
    // fp32 distances are compared against the double thresholds (PCR:254,353): d2 < thr  <=>  d2 < ceil_f32(thr)
    float fl = (float)p->maximum_dis_line_for_match, fp = (float)p->maximum_dis_plane_for_match;
    if ((double)fl < p->maximum_dis_line_for_match) fl = nextafterf(fl, INFINITY);
    if ((double)fp < p->maximum_dis_plane_for_match) fp = nextafterf(fp, INFINITY);
    c->max_d2_line = fl;
    c->max_d2_plane = fp;
    c->huber_a = p->huber_a;
    c->inliner_dis = p->inliner_dis;
    c->inlier_ratio = p->inlier_ratio;
    c->minimum_icp_R_diff = p->minimum_icp_R_diff;
    c->minimum_icp_T_diff = p->minimum_icp_T_diff;
    c->bound = (double)p->para_max_speed;
    c->para_max_angular_rate = p->para_max_angular_rate;
    c->max_final_cost = p->max_final_cost;
    c->min_ts = p->minimum_pt_time_stamp;
    c->max_ts = p->maximum_pt_time_stamp;
    return 0;
}

static void prof_begin(ll_reg *r, int cls)
{
    if (!r->profiling) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, r->stream);
    r->ev.push_back(a);
    r->ev.push_back(b);
    r->ev_class.push_back(cls);
}
static void prof_end(ll_reg *r)
{
    if (!r->profiling) return;
    (void)hipEventRecord(r->ev.back(), r->stream);
}

// common launch sequence; the feature pointers in r->dev must be set
static int reg_enqueue(ll_reg *r, const ll_map *map, int n_scans, const ll_reg_params *prm, const double *poses_last,
                       const double *poses_curr, const double *poses_incre)
{
    if (!map || !prm || !poses_last || !poses_curr) return set_err("ll_reg", "null argument");
    if (n_scans < 1 || n_scans > r->max_scans) return set_err("ll_reg", "n_scans out of range");
    if (prm->icp_max_iterations < 0 || prm->ceres_max_iterations < 0 || prm->ceres_prerun_times < 0)
        return set_err("ll_reg", "negative iteration count");
    if (prm->icp_max_iterations > (1 << 19)) return set_err("ll_reg", "icp_max_iterations above 524288");  // (the grouped solver tags its exchanges with the launch number in 20 bits)
    if (map->device != r->device) return set_err("ll_reg", "map lives on another device");
    static const int debug_or = getenv("LL_DEBUG_OR") ? atoi(getenv("LL_DEBUG_OR")) : 0;  // (A/B runs of unmodified drivers: bits of ll_reg_set_debug)
    make_reg_const(prm, r->debug | debug_or, &r->rc);
    r->rc.debug_knn_iter = r->debug_knn_iter;
    // A solve enqueued earlier on this handle and never collected still reads its snapshots: let it finish before its pins are
    // replaced (the snapshots could otherwise be recycled and rebuilt under its kernels by a concurrent ll_map_upload / refresh).
    if (r->pinned[0] || r->pinned[1]) HC(hipStreamSynchronize(r->stream));
    // PCR:199 gate
    // the snapshots this solve runs against, whatever ll_map_upload / ll_history_refresh* publish meanwhile
    r->pinned[0] = map_pin(map, 0);
    r->pinned[1] = map_pin(map, 1);
    struct PinGuard {  // an enqueue that fails after this point must not leave its pins behind
        ll_reg *r;
        bool keep = false;
        ~PinGuard()
        {
            if (!keep) {
                // kernels of this enqueue may already be in flight on the snapshots (a failure inside the ICP loop): they must
                // have drained before the pins go and ll_map_upload / a refresh may recycle the buffers
                (void)hipStreamSynchronize(r->stream);
                r->pinned[0].reset();
                r->pinned[1].reset();
            }
        }
    } pin_guard{r};
    const MapKind empty_kind{};
    const MapKind &mk0 = r->pinned[0] ? r->pinned[0]->mk : empty_kind, &mk1 = r->pinned[1] ? r->pinned[1]->mk : empty_kind;
    const bool run = mk0.n > 0 && mk1.n > 50 && prm->current_frame_index > prm->mapping_init_accumulate_frames;
    r->last_gated = run ? 0 : 1;
    r->last_n_scans = n_scans;
    for (int b = 0; b < n_scans; b++) {
        RegState &s = r->h_state[b];
        memset(&s, 0, sizeof(s));
        for (int i = 0; i < 7; i++) {
            s.pose_last[i] = poses_last[7 * b + i];
            s.pose_curr[i] = poses_curr[7 * b + i];
            s.inc[i] = poses_incre ? poses_incre[7 * b + i] : (i == 3 ? 1.0 : 0.0);
        }
        s.prev_q[3] = 1.0;  // q_last_optimize(1,0,0,0), PCR:204
        s.gated = run ? 0 : 1;
        s.done = run ? 0 : 1;
        s.result = 1;
        s.accepted = 1;
    }
    for (hipEvent_t e : r->ev) (void)hipEventDestroy(e);
    r->ev.clear();
    r->ev_class.clear();
    HC(hipMemcpyAsync(r->dev.state, r->h_state.data(), (size_t)n_scans * sizeof(RegState), hipMemcpyHostToDevice, r->stream));
    // feature counts on the host: launch geometry, and the sub-sampling precondition (the reference's random
    // drop, PCR:232-238,339-345,438-458, is not reproduced)
    HC(hipMemcpyAsync(r->h_nc.data(), r->dev.n_corner, (size_t)n_scans * sizeof(int), hipMemcpyDeviceToHost, r->stream));
    HC(hipMemcpyAsync(r->h_ns.data(), r->dev.n_surf, (size_t)n_scans * sizeof(int), hipMemcpyDeviceToHost, r->stream));
    HC(hipStreamSynchronize(r->stream));
    int max_nc = 0, max_ns = 0;
    for (int b = 0; b < n_scans; b++) {
        max_nc = r->h_nc[b] > max_nc ? r->h_nc[b] : max_nc;
        max_ns = r->h_ns[b] > max_ns ? r->h_ns[b] : max_ns;
    }
    if (max_nc > r->dev.cap_c || max_ns > r->dev.cap_s) return set_err("ll_reg", "feature count exceeds the registrar capacity");
    if (!prm->subsample_seed && (max_nc > prm->maximum_allow_residual_block || max_ns > prm->maximum_allow_residual_block))
        return set_err("ll_reg", "feature count exceeds maximum_allow_residual_block and subsample_seed is 0 (strict mode): raise the "
                                 "limit, or set a seed to get the reference's sub-sampling (point_cloud_registration.hpp:232-238,"
                                 "339-345,438-458) with a reproducible random stream");
    if (prm->subsample_seed && (max_nc > 2 * prm->maximum_allow_residual_block || max_ns > 2 * prm->maximum_allow_residual_block))
        r->rc.knn_reuse = 0;  // skipped features change from iteration to iteration: every iteration searches
    // Scans of thousands of surface queries: the tile search (ll_knn_kernels.hip) takes them, in every ICP iteration -- searching
    // all of them costs less than classifying them against reuse records and searching the lists that leaves
    // (small batches are latency chains, not issue-bound: the wavefront-per-query searches and the short work lists serve them better --
    //  single scan 2.24 ms against 2.49 with a tile launch per iteration; debug bit 11 forces the tile search for tests)
    if (max_ns < LL_KNN_TILE_MIN_SURF || max_ns > LL_KNN_TILE_MAX_SURF || (n_scans <= LL_KNN_COOP_MAX_SCANS && !((r->debug | debug_or) & 2048))) r->rc.knn_tile = 0;
    // Scans sorted in segments (more than LL_KNN_TILE_SEG surface queries: Mid-100) keep the reuse machinery behind the tile search of ICP
    // iterations 0 / 1: measured on C3 (bench_c3.py, k-NN class per step) 6.7 ms against 10.3 ms with a tile search in every iteration and
    // 7.2 ms without the tile search
    if (r->rc.knn_tile == 2 && max_ns > LL_KNN_TILE_SEG) r->rc.knn_tile = 1;
    if (r->rc.knn_tile == 2) r->rc.knn_reuse = 0;
    // Small batches leave most of the chip idle with one workgroup per scan: spread each scan's cost evaluations over a
    // group of LL_GRP workgroups (ll_reg_kernels.hip, group_*).  Compact scans only; the others run on the group's first.
    // A scan whose records (nearly) fit one CU's LDS cache gains nothing from it and pays ~3.5 us per exchange: voxel-filtered
    // clouds of a few thousand features (the mapping loop, Q-pipe) stay on one workgroup.
    r->rc.solve_group = (r->rc.solve_group == 1 || n_scans > LL_GRP_MAX_SCANS || r->rc.if_motion_deblur || r->rc.force_general ||
                         max_nc + max_ns < LL_GRP_MIN_BLOCKS) ? 1 : LL_GRP;
    if (run) {
        if (!mk0.pts || !mk1.pts) return set_err("ll_reg", "map not uploaded (or converted to fp16 points: the registrar needs the fp32 records)");
        // the searches of a registration that reuses neighbours prune with a guard band (ll_knn_core.h Grid::guard): ~8 % more
        // candidates per search, displacement budgets set by the true 6th neighbour, a third fewer searches in the late iterations
        Grid g0 = mk0.grid, g1 = mk1.grid;
        g0.guard = g1.guard = r->rc.knn_reuse ? 0.05f : 0.0f;
        if (r->rc.solve_group > 1)  // the exchange granules carry (launch number, exchange number) tags: none may survive from an earlier registration
            HC(hipMemsetAsync(r->dev.grp_xch, 0, (size_t)n_scans * 2 * LL_GRP * 56 * sizeof(unsigned long long), r->stream));
        for (int it = 0; it < prm->icp_max_iterations; it++) {
            r->rc.xch_epoch = it + 1;
            prof_begin(r, 0);
            launch_reg_knn_build(r->dev, r->rc, g0, g1, n_scans, it, max_nc, max_ns, r->stream);
            prof_end(r);
            prof_begin(r, 1);
            if (r->rc.solve_group > 1) HC(hipMemsetAsync(r->dev.grp_ctl, 0, (size_t)(2 * n_scans + 1) * sizeof(int), r->stream));
            launch_reg_solve(r->dev, r->rc, mk1.grid, n_scans, max_nc, max_ns, it, r->stream);
            prof_end(r);
        }
    }
    prof_begin(r, 2);
    launch_reg_finalize(r->dev, r->rc, n_scans, r->stream);
    prof_end(r);
    HC(hipGetLastError());
    pin_guard.keep = true;  // released by ll_reg_collect
    return 0;
}

extern "C" int ll_reg_collect(ll_reg *r, int32_t n_scans, double *poses_curr, double *poses_incre, ll_reg_report *reports,
                              int32_t *results)
{
    if (!r) return set_err("ll_reg_collect", "null handle");
    if (n_scans < 1 || n_scans > r->max_scans) return set_err("ll_reg_collect", "n_scans out of range");
    HC(hipSetDevice(r->device));
    HC(hipMemcpyAsync(r->h_state.data(), r->dev.state, (size_t)n_scans * sizeof(RegState), hipMemcpyDeviceToHost, r->stream));
    HC(hipStreamSynchronize(r->stream));
    r->pinned[0].reset();  // the solve has left the device: its map snapshots may be recycled
    r->pinned[1].reset();
    int n_aborted = 0;
    for (int b = 0; b < n_scans; b++) {
        const RegState &s = r->h_state[b];
        n_aborted += s.aborted ? 1 : 0;
        for (int i = 0; i < 7; i++) {
            if (poses_curr) poses_curr[7 * b + i] = s.pose_curr[i];
            if (poses_incre) poses_incre[7 * b + i] = s.inc[i];
        }
        if (results) results[b] = s.result;
        if (reports) {
            ll_reg_report &rp = reports[b];
            rp.final_cost = s.final_cost;
            rp.initial_cost = s.initial_cost;
            rp.inlier_threshold = s.inlier_thr;
            rp.angular_diff_deg = s.angular_diff;
            rp.t_diff = s.t_diff;
            rp.icp_iterations = s.icp_iters;
            rp.n_blocks_last = s.n_blocks_last;
            rp.corner_avail = s.corner_avail;
            rp.surf_avail = s.surf_avail;
            rp.lm_iterations_total = s.lm_total;
            rp.accepted = s.accepted;
            rp.gated = s.gated;
            rp.aborted = s.aborted ? 1 : 0;
        }
    }
    if (r->profiling) {
        for (int k = 0; k < 3; k++) {
            r->prof_ms[k] = 0.f;
            r->prof_launches[k] = 0;
        }
        for (size_t i = 0; i < r->ev_class.size(); i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r->ev[2 * i], r->ev[2 * i + 1]) == hipSuccess) {
                r->prof_ms[r->ev_class[i]] += ms;
                r->prof_launches[r->ev_class[i]]++;
            }
        }
    }
    if (n_aborted) {
        // Not an error of the call: every output is filled in, the affected scans come back rejected (result 0, report.aborted 1,
        // pose restored) like any registration the reference rejects, the others are valid.  The count is the return value and
        // ll_last_error() says what happened.
        (void)set_err("ll_reg_collect", "a group barrier of the small-batch solver timed out (device oversubscribed?): the affected scans were rejected");
        return n_aborted;
    }
    return 0;
}

extern "C" int ll_debug_quintic(int32_t device, const double *args10, int32_t n, double *out_sequential, double *out_wavefront)
{
    if (!args10 || !out_sequential || !out_wavefront || n < 0) return set_err("ll_debug_quintic", "bad argument");
    if (check_device(device)) return -1;
    double *d_a = nullptr, *d_s = nullptr, *d_w = nullptr;
    const size_t m = (size_t)(n > 0 ? n : 1);
    HC(hipMalloc((void **)&d_a, m * 10 * sizeof(double)));
    HC(hipMalloc((void **)&d_s, m * sizeof(double)));
    HC(hipMalloc((void **)&d_w, m * sizeof(double)));
    int rc = 0;
    if (hipMemcpy(d_a, args10, (size_t)n * 10 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = -1;
    if (!rc) {
        launch_debug_quintic(d_a, n, d_s, d_w, nullptr);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out_sequential, d_s, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_wavefront, d_w, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = -1;
    }
    (void)hipFree(d_a);
    (void)hipFree(d_s);
    (void)hipFree(d_w);
    return rc ? set_err("ll_debug_quintic", "device error") : 0;
}

extern "C" int ll_reg_debug_cycles(ll_reg *r, int32_t scan, long long out[16])
{
    if (!r || scan < 0 || scan >= r->max_scans) return set_err("ll_reg_debug_cycles", "bad argument");
    for (int i = 0; i < 16; i++) out[i] = r->h_state[scan].dbg_cycles[i];
    return 0;
}

extern "C" int ll_reg_debug_worklists(ll_reg *r, int32_t n_scans, int64_t out[4])
{
    if (!r || !out || n_scans < 1 || n_scans > r->max_scans) return set_err("ll_reg_debug_worklists", "bad argument");
    HC(hipSetDevice(r->device));
    const size_t n = (size_t)n_scans * 4;  // work_cnt: [scan][kind][searched, re-sorted] of the last re-query launch
    std::vector<int> h(n);
    HC(hipStreamSynchronize(r->stream));
    HC(hipMemcpy(h.data(), r->dev.work_cnt, n * sizeof(int), hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = out[3] = 0;
    for (size_t i = 0; i < n; i++) out[i & 3] += h[i];
    return 0;
}

extern "C" int ll_reg_kernel_times(ll_reg *r, float ms[3], int32_t launches[3])
{
    if (!r) return set_err("ll_reg_kernel_times", "null handle");
    for (int k = 0; k < 3; k++) {
        if (ms) ms[k] = r->prof_ms[k];
        if (launches) launches[k] = r->prof_launches[k];
    }
    return 0;
}

extern "C" int ll_reg_enqueue_fe(ll_reg *r, const ll_map *map, ll_fe *fe, int32_t n_scans, const ll_reg_params *prm,
                                 const double *poses_last, const double *poses_curr, const double *poses_incre)
{
    if (!r || !fe) return set_err("ll_reg_enqueue_fe", "null handle");
    if (fe->prm.device != r->device) return set_err("ll_reg_enqueue_fe", "extractor lives on another device");
    if (n_scans > fe->prm.max_scans) return set_err("ll_reg_enqueue_fe", "n_scans exceeds the extractor capacity");
    if (fe->prm.max_points > r->max_feat) return set_err("ll_reg_enqueue_fe", "registrar feature capacity < extractor max_points");
    HC(hipSetDevice(r->device));
    // order after the extractor's stream
    HC(hipEventRecord(r->ev_wait, fe->stream));
    HC(hipStreamWaitEvent(r->stream, r->ev_wait, 0));
    r->dev.corner_feat = fe->dev.corner_feat;
    r->dev.surf_feat = fe->dev.surf_feat;
    r->dev.n_corner = fe->dev.n_corner;
    r->dev.n_surf = fe->dev.n_surf;
    r->dev.feat_stride_c = fe->dev.stride;
    r->dev.feat_stride_s = fe->dev.stride;
    return reg_enqueue(r, map, n_scans, prm, poses_last, poses_curr, poses_incre);
}

// ---------------------------------------------------------------------------------------------------- voxel grid
struct ll_voxel {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;  // the stream the current contents of dev.out were produced on
    hipEvent_t ev = nullptr;
    VoxelDev dev{};
};

extern "C" int ll_voxel_create(int32_t device, int32_t max_clouds, int32_t max_points_per_cloud, ll_voxel **out)
{
    if (!out) return set_err("ll_voxel_create", "null argument");
    if (max_clouds < 1 || max_points_per_cloud < 1) return set_err("ll_voxel_create", "bad capacity");
    if (check_device(device)) return -1;
    ll_voxel *v = new ll_voxel();
    v->device = device;
    HC(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    HC(hipEventCreateWithFlags(&v->ev, hipEventDisableTiming));
    const char *err = nullptr;
    if (voxel_alloc(v->dev, max_clouds, max_points_per_cloud, &err)) {
        voxel_free(v->dev);
        (void)hipStreamDestroy(v->stream);
        (void)hipEventDestroy(v->ev);
        delete v;
        return set_err("ll_voxel_create", err);
    }
    *out = v;
    return 0;
}

extern "C" void ll_voxel_destroy(ll_voxel *v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    voxel_free(v->dev);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    if (v->ev) (void)hipEventDestroy(v->ev);
    delete v;
}

extern "C" int ll_voxel_filter(ll_voxel *v, int32_t n_clouds, const float *xyzi, const int32_t *n_points, int32_t stride_points,
                               const float leaf[3], float *out_xyzi, int32_t *n_out, int32_t *status)
{
    if (!v || !xyzi || !n_points || !leaf || !out_xyzi || !n_out) return set_err("ll_voxel_filter", "null argument");
    if (n_clouds < 1 || n_clouds > v->dev.max_clouds) return set_err("ll_voxel_filter", "n_clouds out of range");
    if (stride_points < 1 || stride_points > v->dev.stride) return set_err("ll_voxel_filter", "stride exceeds max_points_per_cloud");
    for (int b = 0; b < n_clouds; b++)
        if (n_points[b] < 0 || n_points[b] > stride_points) return set_err("ll_voxel_filter", "n_points out of range");
    HC(hipSetDevice(v->device));
    const size_t total = (size_t)n_clouds * stride_points;
    HC(hipMemcpyAsync(v->dev.in, xyzi, total * sizeof(float4), hipMemcpyHostToDevice, v->stream));
    HC(hipMemcpyAsync(v->dev.n, n_points, (size_t)n_clouds * sizeof(int), hipMemcpyHostToDevice, v->stream));
    const char *err = nullptr;
    if (voxel_filter(v->dev, v->dev.in, v->dev.n, stride_points, n_clouds, leaf, v->stream, &err)) return set_err("ll_voxel_filter", err);
    v->last_stream = v->stream;
    HC(hipMemcpyAsync(out_xyzi, v->dev.out, total * sizeof(float4), hipMemcpyDeviceToHost, v->stream));
    HC(hipMemcpyAsync(n_out, v->dev.n_out, (size_t)n_clouds * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    std::vector<int> st(n_clouds);
    HC(hipMemcpyAsync(st.data(), v->dev.status, (size_t)n_clouds * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    HC(hipStreamSynchronize(v->stream));
    if (status)
        for (int b = 0; b < n_clouds; b++) status[b] = st[b];
    return 0;
}

extern "C" int ll_voxel_counts(ll_voxel *v, int32_t n_clouds, int32_t *n_out, int32_t *status)
{
    if (!v || n_clouds < 1 || n_clouds > v->dev.max_clouds) return set_err("ll_voxel_counts", "bad argument");
    HC(hipSetDevice(v->device));
    HC(hipStreamSynchronize(v->stream));
    if (v->last_stream && v->last_stream != v->stream) HC(hipStreamSynchronize(v->last_stream));
    if (n_out) HC(hipMemcpy(n_out, v->dev.n_out, (size_t)n_clouds * sizeof(int), hipMemcpyDeviceToHost));
    if (status) HC(hipMemcpy(status, v->dev.status, (size_t)n_clouds * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ll_reg_enqueue_fe_downsampled(ll_reg *r, const ll_map *map, ll_fe *fe, ll_voxel *vc, ll_voxel *vs, float line_res,
                                             float plane_res, int32_t n_scans, const ll_reg_params *prm, const double *poses_last,
                                             const double *poses_curr, const double *poses_incre)
{
    if (!r || !fe || !vc || !vs) return set_err("ll_reg_enqueue_fe_downsampled", "null handle");
    if (fe->prm.device != r->device || vc->device != r->device || vs->device != r->device)
        return set_err("ll_reg_enqueue_fe_downsampled", "handles live on different devices");
    if (n_scans < 1 || n_scans > fe->prm.max_scans) return set_err("ll_reg_enqueue_fe_downsampled", "n_scans exceeds the extractor capacity");
    if (fe->prm.max_points > r->max_feat) return set_err("ll_reg_enqueue_fe_downsampled", "registrar feature capacity < extractor max_points");
    if (vc == vs) return set_err("ll_reg_enqueue_fe_downsampled", "corner and surface need their own voxel filter handle");
    HC(hipSetDevice(r->device));
    // extractor -> (voxel filters, on the registrar's stream) -> registrar
    HC(hipEventRecord(r->ev_wait, fe->stream));
    HC(hipStreamWaitEvent(r->stream, r->ev_wait, 0));
    const char *err = nullptr;
    const float lc[3] = {line_res, line_res, line_res}, ls[3] = {plane_res, plane_res, plane_res};
    if (voxel_filter(vc->dev, fe->dev.corner_feat, fe->dev.n_corner, fe->dev.stride, n_scans, lc, r->stream, &err))
        return set_err("ll_reg_enqueue_fe_downsampled", err);
    if (voxel_filter(vs->dev, fe->dev.surf_feat, fe->dev.n_surf, fe->dev.stride, n_scans, ls, r->stream, &err))
        return set_err("ll_reg_enqueue_fe_downsampled", err);
    vc->last_stream = vs->last_stream = r->stream;
    r->dev.corner_feat = vc->dev.out;
    r->dev.surf_feat = vs->dev.out;
    r->dev.n_corner = vc->dev.n_out;
    r->dev.n_surf = vs->dev.n_out;
    r->dev.feat_stride_c = vc->dev.out_stride;
    r->dev.feat_stride_s = vs->dev.out_stride;
    return reg_enqueue(r, map, n_scans, prm, poses_last, poses_curr, poses_incre);
}

// ---------------------------------------------------------------------------------------------------- cell map
struct ll_history;
struct ll_cellmap {
    ll_history *owner = nullptr;  // a history's own cell map (ll_history_enable_cell_map): fed by the history, possibly on its service thread
    int device = 0;
    hipStream_t stream = nullptr;
    CellMapDev dev{};
    float4 *d_in = nullptr;   // staging of host clouds (max_points)
    double *d_pose = nullptr;
    CellStats *d_stats = nullptr;  // allocated by the first ll_cellmap_features / ll_cellmap_keyframe_images
    KfOut *d_kf = nullptr;
};

static void cellmap_release(ll_cellmap *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    cellmap_free(c->dev);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_pose) (void)hipFree(c->d_pose);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_kf) (void)hipFree(c->d_kf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// A cell map owned by a history may be fed on that history's service thread (ll_history_set_cell_map_async): every public entry point
// that reads or changes a map first waits for the frames handed over so far (and reports the feeder's failure, if any), so a handle
// borrowed from ll_history_cell_map never sees an append in flight or a map swapped by cellmap_grow under it.
static int history_cells_drain(ll_history *h);
static int cellmap_settle(const ll_cellmap *c) { return (c && c->owner) ? history_cells_drain(c->owner) : 0; }

extern "C" int ll_cellmap_create(int32_t device, int64_t max_points, float resolution, int32_t minimum_revisit_threshold, ll_cellmap **out)
{
    if (!out) return set_err("ll_cellmap_create", "null argument");
    if (max_points < 1 || max_points >= 0x3fffffffLL) return set_err("ll_cellmap_create", "max_points out of range");
    if (!(resolution > 0.f)) return set_err("ll_cellmap_create", "resolution must be positive");
    if (check_device(device)) return -1;
    ll_cellmap *c = new ll_cellmap();
    c->device = device;
    const char *err = nullptr;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || cellmap_alloc(c->dev, (int)max_points, resolution, minimum_revisit_threshold, &err) ||
        hipMalloc((void **)&c->d_in, (size_t)max_points * sizeof(float4)) != hipSuccess || hipMalloc((void **)&c->d_pose, 8 * sizeof(double)) != hipSuccess) {
        cellmap_release(c);
        return set_err("ll_cellmap_create", err ? err : "allocation failed");
    }
    *out = c;
    return 0;
}

extern "C" void ll_cellmap_destroy(ll_cellmap *c) { cellmap_release(c); }

// Points_cloud_map grows on the heap without bound (CMK:619-672); the device map has a capacity: raise it, content kept.
extern "C" int ll_cellmap_reserve(ll_cellmap *c, int64_t max_points)
{
    if (!c) return set_err("ll_cellmap_reserve", "null argument");
    if (cellmap_settle(c)) return -1;
    if (max_points < 1 || max_points >= 0x3fffffffLL) return set_err("ll_cellmap_reserve", "max_points out of range");
    if (max_points <= c->dev.cap) return 0;
    HC(hipSetDevice(c->device));
    const char *err = nullptr;
    if (cellmap_grow(c->dev, (int)max_points, c->stream, &err)) return set_err("ll_cellmap_reserve", err ? err : "allocation failed");
    float4 *d_new = nullptr;
    HC(hipMalloc((void **)&d_new, (size_t)max_points * sizeof(float4)));
    if (c->d_in) (void)hipFree(c->d_in);
    c->d_in = d_new;
    if (c->d_stats) {  // (sized by the capacity: allocated again by the next ll_cellmap_features / ll_cellmap_keyframe_images)
        (void)hipFree(c->d_stats);
        c->d_stats = nullptr;
    }
    return 0;
}

extern "C" int ll_cellmap_append(ll_cellmap *c, const float *xyzi, int32_t n)
{
    if (!c || (n > 0 && !xyzi)) return set_err("ll_cellmap_append", "null argument");
    if (cellmap_settle(c)) return -1;
    if (n < 0 || n > c->dev.cap) return set_err("ll_cellmap_append", "cloud exceeds max_points");
    HC(hipSetDevice(c->device));
    if (n > 0) HC(hipMemcpyAsync(c->d_in, xyzi, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    const char *err = nullptr;
    if (cellmap_append(c->dev, c->d_in, n, c->stream, &err)) return set_err("ll_cellmap_append", err);
    HC(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ll_cellmap_append_touched(ll_cellmap *c, const float *xyzi, int32_t n, int32_t min_points, int32_t *cell_ijk,
                                         int64_t capacity_cells, int64_t *n_touched)
{
    if (!c || (n > 0 && !xyzi) || !n_touched) return set_err("ll_cellmap_append_touched", "null argument");
    if (cellmap_settle(c)) return -1;
    if (n < 0 || n > c->dev.cap) return set_err("ll_cellmap_append_touched", "cloud exceeds max_points");
    HC(hipSetDevice(c->device));
    const bool first = c->dev.n_cells == 0;  // set_point_cloud: every cell that received a point (CMK:596-607)
    const int n_before = c->dev.n_pts;
    if (n > 0) HC(hipMemcpyAsync(c->d_in, xyzi, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    const char *err = nullptr;
    if (cellmap_append(c->dev, c->d_in, n, c->stream, &err)) return set_err("ll_cellmap_append_touched", err);
    if (cellmap_touch_counts(c->dev, n_before, n, c->stream, &err)) return set_err("ll_cellmap_append_touched", err);
    const int nc = c->dev.n_cells;
    std::vector<unsigned int> cnt(nc);
    std::vector<unsigned long long> keys(nc);
    if (nc > 0) {
        HC(hipMemcpyAsync(cnt.data(), c->dev.csel, (size_t)nc * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HC(hipMemcpyAsync(keys.data(), c->dev.ckey, (size_t)nc * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    }
    HC(hipStreamSynchronize(c->stream));
    const unsigned int need = first ? 1u : (unsigned int)(min_points > 1 ? min_points : 1);
    int64_t k = 0;
    for (int i = 0; i < nc; i++) {
        if (cnt[i] < need) continue;
        // (the append is committed by now: a short buffer truncates the list, it does not fail the call -- a retry would append the
        //  cloud a second time.  *n_touched is always the full count; more than capacity_cells means the list was cut.)
        if (cell_ijk && k < capacity_cells) cell_unpack(keys[i], cell_ijk + 3 * (size_t)k);
        k++;
    }
    *n_touched = k;
    return 0;
}

extern "C" int ll_cellmap_query_filter(ll_cellmap *c, const double pose[7], float radius, float maximum_in_fov_angle, float leaf,
                                       int32_t down_sample_replace, int64_t *n_cells_selected, int64_t *n_out)
{
    if (!c || !pose) return set_err("ll_cellmap_query_filter", "null argument");
    if (cellmap_settle(c)) return -1;
    if (!(radius >= 0.f)) return set_err("ll_cellmap_query_filter", "radius must not be negative");
    HC(hipSetDevice(c->device));
    HC(hipMemcpyAsync(c->d_pose, pose, 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const char *err = nullptr;
    if (cellmap_query_filter(c->dev, c->d_pose, radius, maximum_in_fov_angle, leaf, down_sample_replace, c->stream, &err))
        return set_err("ll_cellmap_query_filter", err);
    HC(hipStreamSynchronize(c->stream));
    if (n_cells_selected) *n_cells_selected = c->dev.n_sel;
    if (n_out) *n_out = c->dev.n_filt;
    return 0;
}

extern "C" int64_t ll_cellmap_result(ll_cellmap *c, float *xyzi, int64_t capacity_points)
{
    if (!c) return set_err("ll_cellmap_result", "null argument");
    if (cellmap_settle(c)) return -1;
    const int64_t n = c->dev.n_filt;
    if (!xyzi) return n;
    if (capacity_points < n) return set_err("ll_cellmap_result", "buffer too small");
    if (hipSetDevice(c->device) != hipSuccess) return set_err("ll_cellmap_result", "hipSetDevice failed");
    if (n > 0 && hipMemcpy(xyzi, c->dev.filt, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess)
        return set_err("ll_cellmap_result", "copy failed");
    return n;
}

extern "C" int ll_cellmap_stats(const ll_cellmap *c, int64_t *n_cells, int64_t *n_points, int32_t *frame_idx)
{
    if (!c) return set_err("ll_cellmap_stats", "null argument");
    if (cellmap_settle(c)) return -1;
    if (n_cells) *n_cells = c->dev.n_cells;
    if (n_points) *n_points = c->dev.n_pts;
    if (frame_idx) *frame_idx = c->dev.frame;
    return 0;
}

extern "C" int ll_cellmap_features(ll_cellmap *c, int32_t *feature_type, float *feature_vector, float *mean, float *cov, float *eigen_val,
                                   int64_t capacity_cells)
{
    if (!c) return set_err("ll_cellmap_features", "null argument");
    if (cellmap_settle(c)) return -1;
    const int nc = c->dev.n_cells;
    if (capacity_cells < nc) return set_err("ll_cellmap_features", "buffer too small");
    if (nc == 0) return 0;
    HC(hipSetDevice(c->device));
    if (!c->d_stats) DM(c->d_stats, (size_t)c->dev.cap);  // a cell holds at least one point
    const char *err = nullptr;
    if (cellmap_stats(c->dev, c->d_stats, c->stream, &err)) return set_err("ll_cellmap_features", err);
    std::vector<CellStats> st(nc);
    HC(hipMemcpyAsync(st.data(), c->d_stats, (size_t)nc * sizeof(CellStats), hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nc; i++) {
        if (feature_type) feature_type[i] = st[i].type;
        for (int d = 0; d < 3; d++) {
            if (feature_vector) feature_vector[3 * (size_t)i + d] = st[i].vec[d];
            if (mean) mean[3 * (size_t)i + d] = st[i].mean[d];
            if (eigen_val) eigen_val[3 * (size_t)i + d] = st[i].eval[d];
        }
        if (cov)
            for (int d = 0; d < 6; d++) cov[6 * (size_t)i + d] = st[i].cov[d];
    }
    return 0;
}

extern "C" int ll_cellmap_keyframe_images(ll_cellmap *c, float roi_ratio, float *images, float *ratio_nonzero, float *eigen_R,
                                          int32_t *n_vectors, float *centre_and_range)
{
    if (!c) return set_err("ll_cellmap_keyframe_images", "null argument");
    if (cellmap_settle(c)) return -1;
    if (!(roi_ratio >= 0.f && roi_ratio <= 1.f)) return set_err("ll_cellmap_keyframe_images", "roi_ratio must lie in [0, 1]");
    HC(hipSetDevice(c->device));
    if (!c->d_stats) DM(c->d_stats, (size_t)c->dev.cap);
    if (!c->d_kf) DM(c->d_kf, 1);
    const char *err = nullptr;
    if (cellmap_keyframe_images(c->dev, c->d_stats, roi_ratio, c->d_kf, c->stream, &err)) return set_err("ll_cellmap_keyframe_images", err);
    std::vector<KfOut> h(1);
    HC(hipMemcpyAsync(h.data(), c->d_kf, sizeof(KfOut), hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    const KfOut &k = h[0];
    if (images) memcpy(images, k.img, sizeof(k.img));
    if (ratio_nonzero) memcpy(ratio_nonzero, k.ratio, sizeof(k.ratio));
    if (eigen_R) memcpy(eigen_R, k.R, sizeof(k.R));
    if (n_vectors)
        for (int i = 0; i < 4; i++) n_vectors[i] = k.n_vec[i];
    if (centre_and_range) {
        for (int d = 0; d < 3; d++) centre_and_range[d] = k.centre[d];
        centre_and_range[3] = k.roi_range;
    }
    return 0;
}

extern "C" int ll_keyframe_similarity(int32_t device, const float *img_a, const float *img_b, float *similarity)
{
    if (!img_a || !img_b || !similarity) return set_err("ll_keyframe_similarity", "null argument");
    if (check_device(device)) return -1;
    const size_t n = (size_t)LL_KF_RES * LL_KF_RES;
    float *d = nullptr;
    DM(d, 2 * n + 1);
    int rc = 0;
    const char *err = nullptr;
    if (hipMemcpy(d, img_a, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, img_b, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess || keyframe_similarity(d, d + n, d + 2 * n, nullptr, &err) ||
        hipMemcpy(similarity, d + 2 * n, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        rc = set_err("ll_keyframe_similarity", err ? err : "device copy failed");
    (void)hipFree(d);
    return rc;
}

extern "C" int ll_cellmap_dump(ll_cellmap *c, float *xyzi, int64_t capacity_points, int32_t *cell_ijk, int32_t *cell_start,
                               int32_t *cell_last_update, int64_t capacity_cells)
{
    if (!c) return set_err("ll_cellmap_dump", "null argument");
    if (cellmap_settle(c)) return -1;
    const int np = c->dev.n_pts, nc = c->dev.n_cells;
    if ((xyzi && capacity_points < np) || ((cell_ijk || cell_start || cell_last_update) && capacity_cells < nc))
        return set_err("ll_cellmap_dump", "buffer too small");
    HC(hipSetDevice(c->device));
    if (xyzi && np > 0) HC(hipMemcpy(xyzi, c->dev.pts, (size_t)np * sizeof(float4), hipMemcpyDeviceToHost));
    if (cell_start) {
        if (nc > 0)
            HC(hipMemcpy(cell_start, c->dev.cstart, (size_t)(nc + 1) * sizeof(int), hipMemcpyDeviceToHost));
        else
            cell_start[0] = 0;
    }
    if (cell_last_update && nc > 0) HC(hipMemcpy(cell_last_update, c->dev.clast, (size_t)nc * sizeof(int), hipMemcpyDeviceToHost));
    if (cell_ijk && nc > 0) {
        std::vector<unsigned long long> keys(nc);
        HC(hipMemcpy(keys.data(), c->dev.ckey, (size_t)nc * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < nc; i++) cell_unpack(keys[i], cell_ijk + 3 * (size_t)i);
    }
    return 0;
}

// The map where it lies: device pointers to the stored points ({x, y, z, 0}, ordered by (cell key, insertion order)) and to the 64-bit
// cell key of every point (21 bits per axis of the cell index + 2^20: ll_cellmap_core.h cell_pack).  Valid until the next call that
// changes this map; the handle's stream has been drained.  The input of the multi-GPU cell-map gather (multigpu.gather_cell_maps).
extern "C" int ll_cellmap_device_view(ll_cellmap *c, const float **dev_xyz0, const uint64_t **dev_point_keys, int64_t *n_points, int64_t *n_cells)
{
    if (!c || !dev_xyz0 || !dev_point_keys || !n_points) return set_err("ll_cellmap_device_view", "null argument");
    if (cellmap_settle(c)) return -1;
    HC(hipSetDevice(c->device));
    HC(hipStreamSynchronize(c->stream));
    *dev_xyz0 = (const float *)c->dev.pts;
    *dev_point_keys = (const uint64_t *)c->dev.pkey;
    *n_points = c->dev.n_pts;
    if (n_cells) *n_cells = c->dev.n_cells;
    return 0;
}

// ---------------------------------------------------------------------------------------------------- history
struct ll_history {
    int device = 0;
    hipStream_t stream = nullptr;
    int max_hist = 0, max_pts = 0;
    float res[2] = {0.1f, 0.4f};       // line_res, plane_res
    float4 *frames[2] = {nullptr, nullptr};  // [max_hist + 1][max_pts] ring per kind
    std::vector<int> count[2];         // points per ring slot
    int head = 0, size = 0;            // FIFO window over the ring slots
    float4 *d_in = nullptr, *d_xf = nullptr, *d_concat = nullptr;
    int *d_n = nullptr;
    double *d_pose = nullptr;
    int2 *hp_table = nullptr, *d_table = nullptr;  // [2 kinds][LL_HIST_CONCAT_MAX + 1] segment tables of the concatenation (pinned host / device)
    VoxelDev vox_frame{}, vox_map{};
    double last_q[4] = {0, 0, 0, 1}, last_t[3] = {0, 0, 0};  // m_last_his_add_q / m_last_his_add_t
    double gate[7] = {0, 0, 0, 1, 0, 0, 0};                   // ll_history_set_gate_pose: the node's pose BEFORE the registration
    bool has_gate = false;
    int64_t n_map[2] = {0, 0};
    float4 *d_map[2] = {nullptr, nullptr};   // filtered match buffer of the last refresh
    // m_pt_cell_map_corners / m_pt_cell_map_planes (laser_mapping.hpp:274-275), ll_history_enable_cell_map
    ll_cellmap *cells[2] = {nullptr, nullptr};
    VoxelDev vox_cells{};
    float4 *d_cmap[2] = {nullptr, nullptr};  // match buffer of the last ll_history_refresh_cells
    const float4 *map_src[2] = {nullptr, nullptr};
    // The cell maps fed BESIDE the mapping loop (ll_history_set_cell_map_async): in matching mode 0 nothing reads them between frames
    // (laser_mapping.hpp:1492-1493 only appends), and an append re-sorts the whole stored map -- 0.4 ms per frame once the map holds a
    // couple of million points.  A service thread (the reference runs its map services on threads too, laser_mapping.hpp:568-594) takes the
    // filtered frames from a ring of staging buffers and appends them in order on the cell maps' own streams; every reader drains it first.
    bool cells_async = false;
    std::thread feeder;
    std::mutex mu;
    std::condition_variable cv_job, cv_idle;
    struct FeedJob {
        int kind, slot, n;
        hipEvent_t ready;  // recorded on h->stream behind the copy into the staging slot
    };
    std::deque<FeedJob> jobs;
    int in_flight = 0;        // jobs queued or being appended
    bool stop = false;
    std::string feed_error;   // first failure of the thread (reported by the next drain)
    static constexpr int kStage = 16;
    float4 *stage[2][16] = {};
    int stage_next[2] = {0, 0};
};

// history-owned cell maps grow with the sequence (the reference's cells live on the heap, CMK:619-672): twice the capacity when the
// next cloud would not fit
static int history_cells_append(ll_history *h, int kind, const float4 *d_src, int n, std::string *why)
{
    ll_cellmap *c = h->cells[kind];
    const char *err = nullptr;
    if ((long long)c->dev.n_pts + n > c->dev.cap) {
        long long want = 2LL * c->dev.cap;
        while (want < (long long)c->dev.n_pts + n) want *= 2;
        if (want >= 0x3fffffffLL) {
            *why = "cell map cannot grow further";
            return -1;
        }
        // the staging buffer of the new capacity first: a failure then leaves the map as it was (capacity and staging size agree)
        float4 *d_new = nullptr;
        if (hipMalloc((void **)&d_new, (size_t)want * sizeof(float4)) != hipSuccess) {
            *why = "allocation failed";
            return -1;
        }
        if (cellmap_grow(c->dev, (int)want, c->stream, &err)) {
            (void)hipFree(d_new);
            *why = err ? err : "cell map cannot grow further";
            return -1;
        }
        if (c->d_in) (void)hipFree(c->d_in);
        c->d_in = d_new;
        if (c->d_stats) {
            (void)hipFree(c->d_stats);
            c->d_stats = nullptr;
        }
    }
    if (cellmap_append(c->dev, d_src, n, c->stream, &err)) {
        *why = err ? err : "append failed";
        return -1;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        *why = "stream error";
        return -1;
    }
    return 0;
}

static void history_feeder_main(ll_history *h)
{
    (void)hipSetDevice(h->device);
    for (;;) {
        ll_history::FeedJob job;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv_job.wait(lk, [h] { return h->stop || !h->jobs.empty(); });
            if (h->jobs.empty()) return;  // (stop, and nothing left)
            job = h->jobs.front();
            h->jobs.pop_front();
        }
        std::string why;
        bool failed = false;
        bool skip = false;
        {
            std::lock_guard<std::mutex> lk(h->mu);
            skip = !h->feed_error.empty();  // latched: after a failure nothing more is appended (a map that silently lacks one frame is worse than none)
        }
        if (skip) {
            (void)hipEventSynchronize(job.ready);
        } else if (hipEventSynchronize(job.ready) != hipSuccess) {
            failed = true;
            why = "staging copy failed";
        } else if (history_cells_append(h, job.kind, h->stage[job.kind][job.slot], job.n, &why)) {
            failed = true;
        }
        (void)hipEventDestroy(job.ready);
        {
            std::lock_guard<std::mutex> lk(h->mu);
            if (failed && h->feed_error.empty()) h->feed_error = why;
            h->in_flight--;
        }
        h->cv_idle.notify_all();
    }
}

// every frame handed to the feeder has been appended; 0, or -1 with the feeder's first error.  The error is LATCHED: the feeder stops
// appending at its first failure and every reader / ll_history_add* reports it until ll_history_set_cell_map_async(h, 0) acknowledges
// it (the cell maps then lack the frames from the failing one on; the caller decides whether to go on inline or to start over).
static int history_cells_drain(ll_history *h)
{
    if (!h->cells_async) return 0;
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv_idle.wait(lk, [h] { return h->in_flight == 0; });
    if (!h->feed_error.empty()) return set_err("ll_history (cell-map feeder)", ("feeding stopped at its first failure: " + h->feed_error).c_str());
    return 0;
}

extern "C" void ll_history_destroy(ll_history *h);
static int history_create_impl(int32_t device, int32_t maximum_history_size, int32_t max_points_per_frame, float line_res, float plane_res,
                               ll_history *h);
extern "C" int ll_history_create(int32_t device, int32_t maximum_history_size, int32_t max_points_per_frame, float line_res,
                                 float plane_res, ll_history **out)
{
    if (!out) return set_err("ll_history_create", "null argument");
    if (maximum_history_size < 1 || max_points_per_frame < 1) return set_err("ll_history_create", "bad capacity");
    if (!(line_res > 0.f) || !(plane_res > 0.f)) return set_err("ll_history_create", "resolutions must be positive");
    if ((int64_t)(maximum_history_size + 1) * max_points_per_frame >= 0x7fffffffLL) return set_err("ll_history_create", "history too large");
    if (check_device(device)) return -1;
    ll_history *h = new ll_history();
    if (history_create_impl(device, maximum_history_size, max_points_per_frame, line_res, plane_res, h)) {
        ll_history_destroy(h);
        return -1;
    }
    *out = h;
    return 0;
}

static int history_create_impl(int32_t device, int32_t maximum_history_size, int32_t max_points_per_frame, float line_res, float plane_res,
                               ll_history *h)
{
    h->device = device;
    h->max_hist = maximum_history_size;
    h->max_pts = max_points_per_frame;
    h->res[0] = line_res;
    h->res[1] = plane_res;
    HC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const size_t slots = (size_t)maximum_history_size + 1, cap = slots * max_points_per_frame;
    for (int k = 0; k < 2; k++) {
        DM(h->frames[k], cap);
        DM(h->d_map[k], cap);
        h->count[k].assign(slots, 0);
    }
    DM(h->d_in, (size_t)max_points_per_frame);
    DM(h->d_xf, (size_t)max_points_per_frame);
    DM(h->d_concat, cap);
    DM(h->d_n, 1);
    DM(h->d_pose, 8);
    DM(h->d_table, 2 * (LL_HIST_CONCAT_MAX + 1));
    HC(hipHostMalloc((void **)&h->hp_table, 2 * (LL_HIST_CONCAT_MAX + 1) * sizeof(int2), hipHostMallocDefault));
    const char *err = nullptr;
    if (voxel_alloc(h->vox_frame, 1, max_points_per_frame, &err) || voxel_alloc(h->vox_map, 1, (int)cap, &err))
        return set_err("ll_history_create", err);
    return 0;
}

extern "C" void ll_history_destroy(ll_history *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    voxel_free(h->vox_frame);
    voxel_free(h->vox_map);
    voxel_free(h->vox_cells);
    if (h->feeder.joinable()) {
        {
            std::lock_guard<std::mutex> lk(h->mu);
            h->stop = true;
        }
        h->cv_job.notify_all();
        h->feeder.join();
    }
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < ll_history::kStage; i++)
            if (h->stage[k][i]) (void)hipFree(h->stage[k][i]);
    for (int k = 0; k < 2; k++) cellmap_release(h->cells[k]);
    if (h->hp_table) (void)hipHostFree(h->hp_table);
    void *ptrs[] = {h->frames[0], h->frames[1], h->d_map[0], h->d_map[1], h->d_in, h->d_xf, h->d_concat, h->d_n, h->d_pose, h->d_cmap[0], h->d_cmap[1], h->d_table};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int32_t ll_history_size(const ll_history *h) { return h ? h->size : -1; }

// one kind of one frame: d_src (sensor frame, n points on the device) -> map frame -> VoxelGrid.  The filtered frame stays
// in h->vox_frame.out; *n_out = its size.
static int history_filter_kind(ll_history *h, int kind, const float4 *d_src, int n, int *n_out)
{
    *n_out = 0;
    if (n <= 0) return 0;
    launch_cloud_transform(d_src, h->d_xf, n, h->d_pose, h->stream);  // laser_mapping.hpp:1421-1431
    HC(hipMemcpyAsync(h->d_n, &n, sizeof(int), hipMemcpyHostToDevice, h->stream));
    const float leaf[3] = {h->res[kind], h->res[kind], h->res[kind]};
    const char *err = nullptr;
    if (voxel_filter(h->vox_frame, h->d_xf, h->d_n, n, 1, leaf, h->stream, &err)) return set_err("ll_history_add", err);  // :1434-1437
    HC(hipMemcpyAsync(n_out, h->vox_frame.n_out, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HC(hipStreamSynchronize(h->stream));
    return 0;
}

// ... -> ring slot (when the frame is pushed) and -> cell map (when enabled; every registered frame, :1492-1493)
static int history_push_kind(ll_history *h, int kind, const float4 *d_src, int n, int slot, bool push)
{
    int n_out = 0;
    if (history_filter_kind(h, kind, d_src, n, &n_out)) return -1;
    if (push) {
        if (n_out > 0)
            HC(hipMemcpyAsync(h->frames[kind] + (size_t)slot * h->max_pts, h->vox_frame.out, (size_t)n_out * sizeof(float4),
                              hipMemcpyDeviceToDevice, h->stream));
        h->count[kind][slot] = n_out;
    }
    if (h->cells[kind] && h->cells_async) {
        // hand the filtered frame to the feeder: copy into the next staging slot (free again: at most kStage frames are in flight)
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv_idle.wait(lk, [h] { return h->in_flight < ll_history::kStage; });
            if (!h->feed_error.empty()) return set_err("ll_history_add (cell-map feeder)", ("feeding stopped at its first failure: " + h->feed_error).c_str());
        }
        const int slot = h->stage_next[kind];
        h->stage_next[kind] = (slot + 1) % ll_history::kStage;
        if (n_out > 0)
            HC(hipMemcpyAsync(h->stage[kind][slot], h->vox_frame.out, (size_t)n_out * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        ll_history::FeedJob job{kind, slot, n_out, nullptr};
        HC(hipEventCreateWithFlags(&job.ready, hipEventDisableTiming));
        HC(hipEventRecord(job.ready, h->stream));
        {
            std::lock_guard<std::mutex> lk(h->mu);
            h->jobs.push_back(job);
            h->in_flight++;
        }
        h->cv_job.notify_one();
    } else if (h->cells[kind]) {
        std::string why;
        if (history_cells_append(h, kind, h->vox_frame.out, n_out, &why)) return set_err("ll_history_add (cell map)", why.c_str());
    }
    HC(hipStreamSynchronize(h->stream));
    return 0;
}

static int history_add_common(ll_history *h, const float4 *d_corner, int n_corner, const float4 *d_surf, int n_surf, const double pose[7],
                              double t_step, double angle_step, int32_t *added)
{
    if (n_corner > h->max_pts || n_surf > h->max_pts) return set_err("ll_history_add", "frame exceeds max_points_per_frame");
    // laser_mapping.hpp:1439-1440: distance of the node's m_q_w_curr / m_t_w_curr -- still the pose BEFORE this
    // registration there (it is copied back at :1496-1500) -- from the pose recorded at the last push.  The gate pose is
    // handed over by ll_history_set_gate_pose (one-shot); without it the transform pose gates (identical results while
    // history_add_t_step = history_add_angle_step = 0, the reference's fixed values: every frame is pushed).
    const double *gp = h->has_gate ? h->gate : pose;
    h->has_gate = false;
    const double r_diff = quat_angular_distance(gp, h->last_q) * 57.3;
    const double dt[3] = {gp[4] - h->last_t[0], gp[5] - h->last_t[1], gp[6] - h->last_t[2]};
    const double t_diff = sqrt(dot3(dt, dt));
    const bool push = h->size < h->max_hist || t_diff > t_step || r_diff > angle_step * 57.3;  // :1446-1448
    if (added) *added = push ? 1 : 0;
    if (!push && !h->cells[0]) return 0;
    const int slots = h->max_hist + 1;
    const int slot = (h->head + h->size) % slots;
    HC(hipMemcpyAsync(h->d_pose, pose, 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (history_push_kind(h, 0, d_corner, n_corner, slot, push)) return -1;
    if (history_push_kind(h, 1, d_surf, n_surf, slot, push)) return -1;
    if (!push) return 0;
    for (int i = 0; i < 4; i++) h->last_q[i] = gp[i];  // :1450-1451
    for (int i = 0; i < 3; i++) h->last_t[i] = gp[4 + i];
    h->size++;
    if (h->size > h->max_hist) {  // :1463-1473 pop_front
        h->head = (h->head + 1) % slots;
        h->size--;
    }
    HC(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int ll_history_set_gate_pose(ll_history *h, const double pose[7])
{
    if (!h || !pose) return set_err("ll_history_set_gate_pose", "null argument");
    for (int i = 0; i < 7; i++) h->gate[i] = pose[i];
    h->has_gate = true;
    return 0;
}

extern "C" int ll_history_add(ll_history *h, const float *corner_xyzi, int32_t n_corner, const float *surf_xyzi, int32_t n_surf,
                              const double pose[7], double history_add_t_step, double history_add_angle_step, int32_t *added)
{
    if (!h || !pose || (n_corner > 0 && !corner_xyzi) || (n_surf > 0 && !surf_xyzi)) return set_err("ll_history_add", "null argument");
    if (n_corner < 0 || n_surf < 0 || n_corner > h->max_pts || n_surf > h->max_pts) return set_err("ll_history_add", "frame exceeds max_points_per_frame");
    HC(hipSetDevice(h->device));
    // the two kinds are staged one after the other through d_in: copy the surface cloud to the concat scratch first
    if (n_corner > 0) HC(hipMemcpyAsync(h->d_in, corner_xyzi, (size_t)n_corner * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    if (n_surf > 0) HC(hipMemcpyAsync(h->d_concat, surf_xyzi, (size_t)n_surf * sizeof(float4), hipMemcpyHostToDevice, h->stream));
    return history_add_common(h, h->d_in, n_corner, h->d_concat, n_surf, pose, history_add_t_step, history_add_angle_step, added);
}

extern "C" int ll_history_add_fe(ll_history *h, ll_fe *fe, int32_t scan, const double pose[7], double history_add_t_step,
                                 double history_add_angle_step, int32_t *added)
{
    if (!h || !fe || !pose) return set_err("ll_history_add_fe", "null argument");
    if (fe->prm.device != h->device) return set_err("ll_history_add_fe", "extractor lives on another device");
    if (scan < 0 || scan >= fe->prm.max_scans) return set_err("ll_history_add_fe", "scan slot out of range");
    HC(hipSetDevice(h->device));
    HC(hipStreamSynchronize(fe->stream));
    int nc = 0, ns = 0;
    HC(hipMemcpy(&nc, fe->dev.n_corner + scan, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&ns, fe->dev.n_surf + scan, sizeof(int), hipMemcpyDeviceToHost));
    return history_add_common(h, fe->dev.corner_feat + (size_t)scan * fe->dev.stride, nc, fe->dev.surf_feat + (size_t)scan * fe->dev.stride, ns,
                              pose, history_add_t_step, history_add_angle_step, added);
}

extern "C" int ll_history_add_voxel(ll_history *h, ll_voxel *vc, ll_voxel *vs, int32_t cloud, const double pose[7],
                                    double history_add_t_step, double history_add_angle_step, int32_t *added)
{
    if (!h || !vc || !vs || !pose) return set_err("ll_history_add_voxel", "null argument");
    if (vc->device != h->device || vs->device != h->device) return set_err("ll_history_add_voxel", "handles live on different devices");
    if (cloud < 0 || cloud >= vc->dev.max_clouds || cloud >= vs->dev.max_clouds) return set_err("ll_history_add_voxel", "cloud index out of range");
    HC(hipSetDevice(h->device));
    // wait for the filters' producers only (a device-wide barrier would serialise independent sequences sharing the GPU)
    if (vc->last_stream) HC(hipStreamSynchronize(vc->last_stream));
    if (vs->last_stream && vs->last_stream != vc->last_stream) HC(hipStreamSynchronize(vs->last_stream));
    int nc = 0, ns = 0;
    HC(hipMemcpy(&nc, vc->dev.n_out + cloud, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&ns, vs->dev.n_out + cloud, sizeof(int), hipMemcpyDeviceToHost));
    return history_add_common(h, vc->dev.out + (size_t)cloud * vc->dev.out_stride, nc, vs->dev.out + (size_t)cloud * vs->dev.out_stride, ns, pose,
                              history_add_t_step, history_add_angle_step, added);
}

static float match_cell_size(int kind, float leaf);

extern "C" int ll_history_refresh(ll_history *h, ll_map *map, int64_t *n_map_corner, int64_t *n_map_surf)
{
    if (!h || !map) return set_err("ll_history_refresh", "null argument");
    if (map->device != h->device) return set_err("ll_history_refresh", "map lives on another device");
    HC(hipSetDevice(h->device));
    const int slots = h->max_hist + 1;
    for (int kind = 0; kind < 2; kind++) {
        // laser_mapping.hpp:519-530: concatenate the history, oldest frame first
        int total = 0;
        if (h->size <= LL_HIST_CONCAT_MAX && h->hp_table) {  // one gather launch (the table travels as one small pinned copy)
            int2 *tab = h->hp_table + (size_t)kind * (LL_HIST_CONCAT_MAX + 1);
            int n_seg = 0;
            for (int i = 0; i < h->size; i++) {
                const int slot = (h->head + i) % slots;
                const int c = h->count[kind][slot];
                if (c > 0) tab[n_seg++] = make_int2((int)((size_t)slot * h->max_pts), total);  // (ring size x max_pts < 2^31: ll_history_create)
                total += c;
            }
            tab[n_seg] = make_int2(0, total);
            if (total > 0) {
                int2 *d_tab = h->d_table + (size_t)kind * (LL_HIST_CONCAT_MAX + 1);
                HC(hipMemcpyAsync(d_tab, tab, (size_t)(n_seg + 1) * sizeof(int2), hipMemcpyHostToDevice, h->stream));
                launch_history_concat(h->frames[kind], d_tab, n_seg, total, h->d_concat, h->stream);
            }
        } else {
            for (int i = 0; i < h->size; i++) {
                const int slot = (h->head + i) % slots;
                const int c = h->count[kind][slot];
                if (c > 0)
                    HC(hipMemcpyAsync(h->d_concat + total, h->frames[kind] + (size_t)slot * h->max_pts, (size_t)c * sizeof(float4),
                                      hipMemcpyDeviceToDevice, h->stream));
                total += c;
            }
        }
        int n_out = 0;
        if (total > 0) {
            HC(hipMemcpyAsync(h->d_n, &total, sizeof(int), hipMemcpyHostToDevice, h->stream));
            const float leaf[3] = {h->res[kind], h->res[kind], h->res[kind]};
            const char *err = nullptr;
            if (voxel_filter(h->vox_map, h->d_concat, h->d_n, total, 1, leaf, h->stream, &err)) return set_err("ll_history_refresh", err);  // :533-537
            HC(hipMemcpyAsync(&n_out, h->vox_map.n_out, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HC(hipStreamSynchronize(h->stream));
            HC(hipMemcpyAsync(h->d_map[kind], h->vox_map.out, (size_t)n_out * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        }
        h->n_map[kind] = n_out;
        h->map_src[kind] = h->d_map[kind];
        // the search structure (laser_mapping.hpp:539-546: two KdTreeFLANN::setInputCloud) is the device grid
        const char *err = nullptr;
        const float cell = match_cell_size(kind, h->res[kind]);
        if (map_rebuild(map, kind, (const float *)h->d_map[kind], 4, n_out, cell, h->stream, &err)) return set_err("map_build", err ? err : "failed");
    }
    HC(hipStreamSynchronize(h->stream));
    if (n_map_corner) *n_map_corner = h->n_map[0];
    if (n_map_surf) *n_map_surf = h->n_map[1];
    return 0;
}

extern "C" int64_t ll_history_map_cloud(ll_history *h, int32_t kind, float *xyzi, int64_t capacity_points)
{
    if (!h || kind < 0 || kind > 1) return set_err("ll_history_map_cloud", "bad argument");
    const int64_t n = h->n_map[kind];
    if (!xyzi) return n;
    if (capacity_points < n) return set_err("ll_history_map_cloud", "buffer too small");
    if (hipSetDevice(h->device) != hipSuccess) return set_err("ll_history_map_cloud", "hipSetDevice failed");
    if (n > 0 && hipMemcpy(xyzi, h->map_src[kind], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess)
        return set_err("ll_history_map_cloud", "copy failed");
    return n;
}

extern "C" int ll_history_map_cloud_device(ll_history *h, int32_t kind, const float **dev_xyzi, int64_t *n_points)
{
    if (!h || !dev_xyzi || !n_points || kind < 0 || kind > 1) return set_err("ll_history_map_cloud_device", "bad argument");
    HC(hipSetDevice(h->device));
    HC(hipStreamSynchronize(h->stream));
    *dev_xyzi = (const float *)h->map_src[kind];
    *n_points = h->n_map[kind];
    return 0;
}

static float match_cell_size(int kind, float leaf)
{
    // Cell size from the voxel leaf the buffer has just been filtered with: the points are about one leaf apart (along
    // the edges for the corner cloud, across the surfaces for the other), and a search is fastest with a handful of
    // points per cell.  The default corner cell (1.45 m, sized for a sparse edge map and the sqrt(2) m line radius)
    // would put hundreds of candidates of a dense local edge map into the query's own cells.
    return (kind == LL_MAP_CORNER) ? fminf(fmaxf(4.0f * leaf, 0.4f), 1.45f) : fminf(fmaxf(3.0f * leaf, 0.45f), 1.2f);
}

extern "C" int ll_history_enable_cell_map(ll_history *h, int64_t max_points, float cell_resolution, int32_t threshold_cell_revisit)
{
    if (!h) return set_err("ll_history_enable_cell_map", "null argument");
    if (h->cells[0]) return set_err("ll_history_enable_cell_map", "already enabled");
    if (max_points < h->max_pts) return set_err("ll_history_enable_cell_map", "max_points below max_points_per_frame");
    HC(hipSetDevice(h->device));
    const char *err = nullptr;
    bool ok = true;
    for (int k = 0; k < 2 && ok; k++) {
        // laser_mapping.hpp:620-624: set_resolution( m_pt_cell_resolution ), m_minimum_revisit_threshold
        ok = ll_cellmap_create(h->device, max_points, cell_resolution, threshold_cell_revisit, &h->cells[k]) == 0 &&
             hipMalloc((void **)&h->d_cmap[k], (size_t)max_points * sizeof(float4)) == hipSuccess;
    }
    if (ok && voxel_alloc(h->vox_cells, 1, (int)max_points, &err)) ok = false;
    if (ok) h->cells[0]->owner = h->cells[1]->owner = h;  // (every ll_cellmap_* call on them settles the service thread first, cellmap_settle)
    if (!ok) {  // all or nothing: a half-enabled history would fail later in ll_history_refresh_cells
        const std::string why = err ? std::string(err) : g_err;
        voxel_free(h->vox_cells);
        for (int k = 0; k < 2; k++) {
            cellmap_release(h->cells[k]);
            h->cells[k] = nullptr;
            if (h->d_cmap[k]) (void)hipFree(h->d_cmap[k]);
            h->d_cmap[k] = nullptr;
        }
        return set_err("ll_history_enable_cell_map", why.empty() ? "allocation failed" : why.c_str());
    }
    return 0;
}

// NULL with ll_last_error() set: bad argument, cell maps not enabled, or the service thread failed.  The handle stays the history's: every
// ll_cellmap_* call on it waits for the frames handed to the service thread so far, so it may be kept across ll_history_add*; what
// ll_cellmap_device_view returns for it is valid only until the next ll_history_add* (which may grow and move the map).
extern "C" ll_cellmap *ll_history_cell_map(ll_history *h, int32_t kind)
{
    if (!h || kind < 0 || kind > 1) {
        set_err("ll_history_cell_map", "bad argument");
        return nullptr;
    }
    if (!h->cells[kind]) {
        set_err("ll_history_cell_map", "cell maps are not enabled (ll_history_enable_cell_map)");
        return nullptr;
    }
    if (history_cells_drain(h)) return nullptr;  // (the caller is about to read the map)
    return h->cells[kind];
}

// enable != 0: the frames ll_history_add* receives from now on reach the cell maps through a service thread, in order, beside the caller
// (matching mode 0: nothing reads the cell maps between frames); every entry point that reads them -- ll_history_cell_map,
// ll_history_refresh_cells, ll_history_sync_cell_maps -- waits for the frames handed over so far.  enable == 0: drain and append inline
// again (the default).
extern "C" int ll_history_set_cell_map_async(ll_history *h, int32_t enable)
{
    if (!h) return set_err("ll_history_set_cell_map_async", "null argument");
    if (!h->cells[0]) return set_err("ll_history_set_cell_map_async", "cell maps are not enabled (ll_history_enable_cell_map)");
    HC(hipSetDevice(h->device));
    if (!enable) {
        const int rc = history_cells_drain(h);  // (reports a latched feeder error one last time ...)
        h->cells_async = false;
        std::lock_guard<std::mutex> lk(h->mu);
        h->feed_error.clear();                  // (... and acknowledges it)
        return rc;
    }
    if (h->cells_async) return 0;
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < ll_history::kStage; i++)
            if (!h->stage[k][i]) DM(h->stage[k][i], (size_t)h->max_pts);
    if (!h->feeder.joinable()) h->feeder = std::thread(history_feeder_main, h);
    h->cells_async = true;
    return 0;
}

extern "C" int ll_history_sync_cell_maps(ll_history *h)
{
    if (!h) return set_err("ll_history_sync_cell_maps", "null argument");
    return history_cells_drain(h);
}

// update_buff_for_matching with m_matching_mode == 1 (laser_mapping.hpp:471-546)
extern "C" int ll_history_refresh_cells(ll_history *h, ll_map *map, const double pose[7], float maximum_search_range_corner,
                                        float maximum_search_range_surface, float maximum_in_fov_angle, int32_t down_sample_replace,
                                        int64_t *n_map_corner, int64_t *n_map_surf)
{
    if (!h || !map || !pose) return set_err("ll_history_refresh_cells", "null argument");
    if (!h->cells[0]) return set_err("ll_history_refresh_cells", "cell maps are not enabled (ll_history_enable_cell_map)");
    if (map->device != h->device) return set_err("ll_history_refresh_cells", "map lives on another device");
    if (history_cells_drain(h)) return -1;
    HC(hipSetDevice(h->device));
    const float range[2] = {maximum_search_range_corner, maximum_search_range_surface};
    for (int kind = 0; kind < 2; kind++) {
        ll_cellmap *c = h->cells[kind];
        const float leaf1 = h->res[kind];
        // :475-513: cells in range and in the field of view, each through the VoxelGrid, concatenated
        if (ll_cellmap_query_filter(c, pose, range[kind], maximum_in_fov_angle, leaf1, down_sample_replace, nullptr, nullptr)) return -1;
        const int total = c->dev.n_filt;
        int n_out = 0;
        if (total > 0) {
            HC(hipMemcpyAsync(h->d_n, &total, sizeof(int), hipMemcpyHostToDevice, h->stream));
            const float leaf[3] = {leaf1, leaf1, leaf1};
            const char *err = nullptr;
            if (voxel_filter(h->vox_cells, c->dev.filt, h->d_n, total, 1, leaf, h->stream, &err)) return set_err("ll_history_refresh_cells", err);  // :533-537
            HC(hipMemcpyAsync(&n_out, h->vox_cells.n_out, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HC(hipStreamSynchronize(h->stream));
            HC(hipMemcpyAsync(h->d_cmap[kind], h->vox_cells.out, (size_t)n_out * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
        }
        h->n_map[kind] = n_out;
        h->map_src[kind] = h->d_cmap[kind];
        const char *err = nullptr;
        if (map_rebuild(map, kind, (const float *)h->d_cmap[kind], 4, n_out, match_cell_size(kind, leaf1), h->stream, &err))
            return set_err("map_build", err ? err : "failed");
    }
    HC(hipStreamSynchronize(h->stream));
    if (n_map_corner) *n_map_corner = h->n_map[0];
    if (n_map_surf) *n_map_surf = h->n_map[1];
    return 0;
}

extern "C" int ll_reg_solve_batch_fe(ll_reg *r, const ll_map *map, ll_fe *fe, int32_t n_scans, const ll_reg_params *prm,
                                     const double *poses_last, double *poses_curr, double *poses_incre, ll_reg_report *reports,
                                     int32_t *results)
{
    if (ll_reg_enqueue_fe(r, map, fe, n_scans, prm, poses_last, poses_curr, poses_incre)) return -1;
    return ll_reg_collect(r, n_scans, poses_curr, poses_incre, reports, results);
}

extern "C" int ll_reg_upload_features(ll_reg *r, int32_t n_scans, const float *corner_xyzi, const int32_t *n_corner, int32_t stride_corner,
                                      const float *surf_xyzi, const int32_t *n_surf, int32_t stride_surf)
{
    if (!r || !n_corner || !n_surf) return set_err("ll_reg_upload_features", "null argument");
    if (n_scans < 1 || n_scans > r->max_scans) return set_err("ll_reg_upload_features", "n_scans out of range");
    if (stride_corner < 0 || stride_surf < 0) return set_err("ll_reg_upload_features", "negative stride");
    for (int b = 0; b < n_scans; b++) {  // validate everything before the first copy: a caller mistake must not become a host over-read
        if (n_corner[b] < 0 || n_corner[b] > r->max_feat || n_surf[b] < 0 || n_surf[b] > r->max_feat)
            return set_err("ll_reg_upload_features", "feature count exceeds capacity");
        if ((n_corner[b] > 0 && !corner_xyzi) || (n_surf[b] > 0 && !surf_xyzi))
            return set_err("ll_reg_upload_features", "null feature array with a non-zero count");
        if (n_scans > 1 && (n_corner[b] > stride_corner || n_surf[b] > stride_surf))
            return set_err("ll_reg_upload_features", "feature count exceeds the per-scan stride");
    }
    HC(hipSetDevice(r->device));
    const size_t F = r->max_feat;
    for (int b = 0; b < n_scans; b++) {
        if (n_corner[b] > 0)
            HC(hipMemcpyAsync(r->d_corner + b * F, corner_xyzi + (size_t)b * stride_corner * 4, (size_t)n_corner[b] * sizeof(float4),
                              hipMemcpyHostToDevice, r->stream));
        if (n_surf[b] > 0)
            HC(hipMemcpyAsync(r->d_surf + b * F, surf_xyzi + (size_t)b * stride_surf * 4, (size_t)n_surf[b] * sizeof(float4),
                              hipMemcpyHostToDevice, r->stream));
    }
    HC(hipMemcpyAsync(r->d_nc, n_corner, n_scans * sizeof(int), hipMemcpyHostToDevice, r->stream));
    HC(hipMemcpyAsync(r->d_ns, n_surf, n_scans * sizeof(int), hipMemcpyHostToDevice, r->stream));
    HC(hipStreamSynchronize(r->stream));
    r->uploaded_scans = n_scans;
    return 0;
}

extern "C" int ll_reg_enqueue_fe_merged(ll_reg *r, const ll_map *map, ll_fe *fe, int32_t n_scans, int32_t heads, const ll_reg_params *prm,
                                        const double *poses_last, const double *poses_curr, const double *poses_incre)
{
    if (!r || !fe) return set_err("ll_reg_enqueue_fe_merged", "null handle");
    if (fe->prm.device != r->device) return set_err("ll_reg_enqueue_fe_merged", "extractor lives on another device");
    if (heads < 1 || n_scans < 1 || n_scans > r->max_scans || (int64_t)n_scans * heads > fe->prm.max_scans)
        return set_err("ll_reg_enqueue_fe_merged", "n_scans * heads exceeds the extractor capacity (or n_scans the registrar's)");
    HC(hipSetDevice(r->device));
    HC(hipEventRecord(r->ev_wait, fe->stream));
    HC(hipStreamWaitEvent(r->stream, r->ev_wait, 0));
    const int F = r->max_feat;
    launch_reg_merge_heads(fe->dev.corner_feat, fe->dev.surf_feat, fe->dev.n_corner, fe->dev.n_surf, fe->dev.stride, heads, r->d_corner, r->d_surf,
                           r->d_nc, r->d_ns, F, n_scans, r->stream);
    HC(hipGetLastError());
    r->uploaded_scans = n_scans;
    r->dev.corner_feat = r->d_corner;
    r->dev.surf_feat = r->d_surf;
    r->dev.n_corner = r->d_nc;
    r->dev.n_surf = r->d_ns;
    r->dev.feat_stride_c = F;
    r->dev.feat_stride_s = F;
    // a merged cloud larger than the registrar's capacity shows in the counts: reg_enqueue refuses it
    return reg_enqueue(r, map, n_scans, prm, poses_last, poses_curr, poses_incre);
}

extern "C" int ll_reg_enqueue_uploaded(ll_reg *r, const ll_map *map, int32_t n_scans, const ll_reg_params *prm, const double *poses_last,
                                       const double *poses_curr, const double *poses_incre)
{
    if (!r) return set_err("ll_reg_enqueue_uploaded", "null handle");
    if (n_scans < 1 || n_scans > r->uploaded_scans) return set_err("ll_reg_enqueue_uploaded", "no features uploaded for that many scans");
    HC(hipSetDevice(r->device));
    const size_t F = r->max_feat;
    r->dev.corner_feat = r->d_corner;
    r->dev.surf_feat = r->d_surf;
    r->dev.n_corner = r->d_nc;
    r->dev.n_surf = r->d_ns;
    r->dev.feat_stride_c = (int)F;
    r->dev.feat_stride_s = (int)F;
    return reg_enqueue(r, map, n_scans, prm, poses_last, poses_curr, poses_incre);
}

extern "C" int ll_reg_solve_batch(ll_reg *r, const ll_map *map, int32_t n_scans, const float *corner_xyzi, const int32_t *n_corner,
                                  int32_t stride_corner, const float *surf_xyzi, const int32_t *n_surf, int32_t stride_surf,
                                  const ll_reg_params *prm, const double *poses_last, double *poses_curr, double *poses_incre,
                                  ll_reg_report *reports, int32_t *results)
{
    if (ll_reg_upload_features(r, n_scans, corner_xyzi, n_corner, stride_corner, surf_xyzi, n_surf, stride_surf)) return -1;
    if (ll_reg_enqueue_uploaded(r, map, n_scans, prm, poses_last, poses_curr, poses_incre)) return -1;
    return ll_reg_collect(r, n_scans, poses_curr, poses_incre, reports, results);
}

extern "C" int ll_reg_solve(ll_reg *r, const ll_map *map, const float *scan_corner_xyzi, int32_t n_corner,
                            const float *scan_surf_xyzi, int32_t n_surf, const ll_reg_params *prm, const double pose_last[7],
                            double pose_curr[7], double pose_incre[7], ll_reg_report *rep)
{
    int32_t res = 1;
    double inc_local[7] = {0, 0, 0, 1, 0, 0, 0};
    double *inc = pose_incre ? pose_incre : inc_local;
    const int rc = ll_reg_solve_batch(r, map, 1, scan_corner_xyzi, &n_corner, n_corner, scan_surf_xyzi, &n_surf, n_surf, prm,
                                      pose_last, pose_curr, inc, rep, &res);
    if (rc < 0) return rc;
    return res;
}

extern "C" int ll_reg_debug_knn(ll_reg *r, int32_t scan, int32_t *corner_idx5, float *corner_d25, int32_t *surf_idx5, float *surf_d25)
{
    if (!r) return set_err("ll_reg_debug_knn", "null handle");
    if (!r->dev.dbg_idx) return set_err("ll_reg_debug_knn", "debug taps not enabled (ll_reg_set_debug)");
    if (scan < 0 || scan >= r->max_scans) return set_err("ll_reg_debug_knn", "scan out of range");
    HC(hipSetDevice(r->device));
    HC(hipStreamSynchronize(r->stream));
    int nc = 0, ns = 0;
    HC(hipMemcpy(&nc, r->dev.n_corner + scan, sizeof(int), hipMemcpyDeviceToHost));
    HC(hipMemcpy(&ns, r->dev.n_surf + scan, sizeof(int), hipMemcpyDeviceToHost));
    const size_t base = (size_t)scan * r->dev.cap * 5;
    D2H_OPT(corner_idx5, r->dev.dbg_idx + base, (size_t)nc * 5, int);
    D2H_OPT(corner_d25, r->dev.dbg_d2 + base, (size_t)nc * 5, float);
    D2H_OPT(surf_idx5, r->dev.dbg_idx + base + (size_t)r->dev.cap_c * 5, (size_t)ns * 5, int);
    D2H_OPT(surf_d25, r->dev.dbg_d2 + base + (size_t)r->dev.cap_c * 5, (size_t)ns * 5, float);
    return 0;
}

extern "C" int ll_cloud_transform_fe_device(ll_reg *r, ll_fe *fe, int32_t n_scans, int32_t kind, const int32_t *accept, const double *poses7,
                                            float *dev_out_xyzi, int64_t capacity_points, int64_t *n_points)
{
    if (!r || !fe || !accept || !poses7 || !dev_out_xyzi || !n_points) return set_err("ll_cloud_transform_fe_device", "null argument");
    if (fe->prm.device != r->device) return set_err("ll_cloud_transform_fe_device", "extractor lives on another device");
    if (n_scans < 0 || n_scans > fe->prm.max_scans || kind < 0 || kind > 1 || *n_points < 0)
        return set_err("ll_cloud_transform_fe_device", "bad argument");
    if (n_scans == 0) return 0;
    HC(hipSetDevice(r->device));
    HC(hipStreamSynchronize(fe->stream));
    std::vector<int> cnt((size_t)n_scans);
    HC(hipMemcpy(cnt.data(), kind == 0 ? fe->dev.n_corner : fe->dev.n_surf, (size_t)n_scans * sizeof(int), hipMemcpyDeviceToHost));
    int64_t total = *n_points;
    for (int b = 0; b < n_scans; b++)
        if (accept[b]) total += cnt[(size_t)b];
    if (total > capacity_points) return set_err("ll_cloud_transform_fe_device", "device buffer too small");
    double *d_poses = nullptr;
    DM(d_poses, (size_t)n_scans * 7);
    hipError_t e = hipMemcpyAsync(d_poses, poses7, (size_t)n_scans * 7 * sizeof(double), hipMemcpyHostToDevice, r->stream);
    int64_t at = *n_points;
    const float4 *src = kind == 0 ? fe->dev.corner_feat : fe->dev.surf_feat;
    for (int b = 0; b < n_scans && e == hipSuccess; b++) {
        if (!accept[b] || cnt[(size_t)b] == 0) continue;
        launch_cloud_transform(src + (size_t)b * fe->dev.stride, (float4 *)dev_out_xyzi + at, cnt[(size_t)b], d_poses + (size_t)b * 7, r->stream);
        at += cnt[(size_t)b];
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    (void)hipFree(d_poses);
    if (e != hipSuccess) return set_err("ll_cloud_transform_fe_device", hipGetErrorString(e));
    *n_points = total;
    return 0;
}

extern "C" int ll_cloud_transform(ll_reg *r, const float *in_xyzi, float *out_xyzi, int32_t n, const double pose[7])
{
    if (!r || !pose || (n > 0 && (!in_xyzi || !out_xyzi))) return set_err("ll_cloud_transform", "null argument");
    if (n <= 0) return 0;
    HC(hipSetDevice(r->device));
    float4 *d_in = nullptr, *d_out = nullptr;
    DM(d_in, (size_t)n);
    DM(d_out, (size_t)n);
    HC(hipMemcpyAsync(d_in, in_xyzi, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, r->stream));
    HC(hipMemcpyAsync(r->d_pose_tmp, pose, 7 * sizeof(double), hipMemcpyHostToDevice, r->stream));
    launch_cloud_transform(d_in, d_out, n, r->d_pose_tmp, r->stream);
    HC(hipGetLastError());
    HC(hipMemcpyAsync(out_xyzi, d_out, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, r->stream));
    HC(hipStreamSynchronize(r->stream));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return 0;
}
