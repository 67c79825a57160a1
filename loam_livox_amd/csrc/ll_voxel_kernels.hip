// ll_voxel_kernels.hip -- device VoxelGrid (PCL 1.9 semantics, ll_voxel_core.h) over a batch of independent clouds.
//
// Layout: clouds[B][stride] float4 {x, y, z, intensity}, n[B] points each.  One pass of each kernel covers the
// whole batch:
//   vox_minmax_kernel   finite-point bounding box + count per cloud            (16 B read per point)
//   vox_params_kernel   min_b / divb_mul / status per cloud                    (one thread per cloud)
//   vox_key_kernel      64-bit key = cloud << 32 | leaf index; non-finite points and pass-through clouds get the
//                       sentinel, which sorts last                             (16 B read, 12 B written per point)
//   hipcub radix sort   (key, point index) pairs -- stable, so the points of a voxel stay in input order
//   vox_head_kernel     voxel heads -> per-cloud voxel counts and head positions
//   vox_centroid_kernel one thread per voxel: float sums in input order, divided by the count
//   vox_copy_kernel     pass-through clouds: output = input
// HBM-bound integer/byte work; the sort (4 radix passes over 12-byte pairs) dominates.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <hipcub/hipcub.hpp>

#include "ll_voxel.h"

namespace ll {

#define VOX_SENTINEL 0xffffffffffffffffull

__device__ __forceinline__ unsigned int f2ord(float f)
{
    const unsigned int b = (unsigned int)__float_as_int(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o)
{
    const unsigned int b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __int_as_float((int)b);
}

// mm[b][0..2] = min (ordered encoding), [3..5] = max, [6] = finite count
__global__ __launch_bounds__(256) void vox_minmax_kernel(const float4 *in, const int *n, int stride, unsigned int *mm)
{
    const int b = blockIdx.y;
    const int nb = n[b];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) {
        const float4 p = in[(size_t)b * stride + i];
        if (ll_isfinite(p.x) && ll_isfinite(p.y) && ll_isfinite(p.z)) {
            lo[0] = fminf(lo[0], p.x);
            lo[1] = fminf(lo[1], p.y);
            lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x);
            hi[1] = fmaxf(hi[1], p.y);
            hi[2] = fmaxf(hi[2], p.z);
            cnt++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            lo[d] = fminf(lo[d], __shfl_down(lo[d], off));
            hi[d] = fmaxf(hi[d], __shfl_down(hi[d], off));
        }
        cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0) {  // min / max / integer add: the result does not depend on the order
        unsigned int *m = mm + (size_t)b * 8;
        for (int d = 0; d < 3; d++) {
            atomicMin(&m[d], f2ord(lo[d]));
            atomicMax(&m[3 + d], f2ord(hi[d]));
        }
        atomicAdd(&m[6], (unsigned int)cnt);
    }
}

__global__ void vox_init_kernel(unsigned int *mm, int *n_vox, int n_clouds)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_clouds) return;
    unsigned int *m = mm + (size_t)b * 8;
    m[0] = m[1] = m[2] = 0xffffffffu;  // identities of min / max in the ordered encoding
    m[3] = m[4] = m[5] = 0u;
    m[6] = m[7] = 0u;
    n_vox[b] = 0;
}

__global__ void vox_params_kernel(const unsigned int *mm, int n_clouds, float inv0, float inv1, float inv2, VoxelParams *prm)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_clouds) return;
    const unsigned int *m = mm + (size_t)b * 8;
    const float mn[3] = {ord2f(m[0]), ord2f(m[1]), ord2f(m[2])}, mx[3] = {ord2f(m[3]), ord2f(m[4]), ord2f(m[5])};
    const float inv[3] = {inv0, inv1, inv2};
    voxel_params(mn, mx, (int)m[6], inv, prm[b]);
}

__global__ __launch_bounds__(256) void vox_key_kernel(const float4 *in, const int *n, int stride, const VoxelParams *prm, float inv0,
                                                      float inv1, float inv2, unsigned long long *keys, unsigned int *vals)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= stride) return;
    const size_t g = (size_t)b * stride + i;
    unsigned long long key = VOX_SENTINEL;
    if (i < n[b] && prm[b].status == VOX_OK) {
        const float4 p = in[g];
        if (ll_isfinite(p.x) && ll_isfinite(p.y) && ll_isfinite(p.z)) {
            const float inv[3] = {inv0, inv1, inv2};
            key = ((unsigned long long)b << 32) | (unsigned long long)voxel_index(p.x, p.y, p.z, inv, prm[b]);
        }
    }
    keys[g] = key;
    vals[g] = (unsigned int)i;
}

// heads[b * stride + v] = position (in sorted order) of the first point of voxel v of cloud b.  A cloud's voxels are
// a contiguous run of the sorted array, so its v-th head is found from the cloud's first sorted position.
__global__ __launch_bounds__(256) void vox_head_kernel(const unsigned long long *keys, size_t total, int *n_vox, unsigned int *is_head)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned long long k = keys[i];
    const bool head = k != VOX_SENTINEL && (i == 0 || keys[i - 1] != k);
    is_head[i] = head ? 1u : 0u;
    if (head) atomicAdd(&n_vox[(int)(k >> 32)], 1);  // integer count: order-independent
}

// rank[i] = exclusive prefix of is_head = global voxel id of the head at i
__global__ __launch_bounds__(256) void vox_headpos_kernel(const unsigned int *is_head, const unsigned int *rank, size_t total,
                                                          unsigned int *head_pos)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total && is_head[i]) head_pos[rank[i]] = (unsigned int)i;
}

// One thread per VOXEL (dense: thread t takes the t-th head), not per sorted point: with ~100 points per voxel the latter
// leaves one working lane per wavefront.  vox_off[b] = voxels of the clouds before b; *n_vox_total = all voxels.
__global__ __launch_bounds__(256) void vox_centroid_kernel(const float4 *in, int stride, const unsigned long long *keys,
                                                           const unsigned int *vals, const unsigned int *head_pos,
                                                           const int *vox_off, const int *n_vox_total, size_t total, float4 *out)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)*n_vox_total) return;
    const size_t i = head_pos[t];
    const unsigned long long k = keys[i];
    const int b = (int)(k >> 32);
    const float4 *src = in + (size_t)b * stride;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;  // CentroidPoint<PointXYZI>: AccumulatorXYZ + AccumulatorIntensity
    int cnt = 0;
    // The additions must run in input order, the loads need not: eight keys, then the matching eight indices, then the
    // eight points are fetched as independent loads, so a crowded voxel (hundreds of points on one thread) costs three
    // dependent round trips per eight points instead of per point.
    bool more = true;
    for (size_t j = i; more && j < total; j += 8) {
        bool ok[8];
        unsigned int v[8];
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) ok[u] = (j + u < total) && keys[j + u] == k;
#pragma unroll
        for (int u = 1; u < 8; u++) ok[u] = ok[u] && ok[u - 1];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ok[u] ? vals[j + u] : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = src[v[u]];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (ok[u]) {
                sx = sx + p[u].x;
                sy = sy + p[u].y;
                sz = sz + p[u].z;
                si = si + p[u].w;
                cnt++;
            }
        }
        more = ok[7];
    }
    const float c = (float)cnt;
    out[(size_t)b * stride + ((int)t - vox_off[b])] = make_float4(sx / c, sy / c, sz / c, si / c);
}

// per-cloud voxel offsets (exclusive scan over the clouds, one workgroup), the output counts and the voxel total
__global__ __launch_bounds__(256) void vox_offsets_kernel(const int *n_vox, const int *n, const VoxelParams *prm, int n_clouds, int *vox_off,
                                                          int *n_out, int *status, int *n_vox_total)
{
    __shared__ int s_w[4];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_clouds; base += 256) {
        const int b = base + tid;
        const int c = b < n_clouds ? n_vox[b] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int excl = s_carry + incl - c;
        for (int w = 0; w < wave; w++) excl += s_w[w];
        if (b < n_clouds) {
            vox_off[b] = excl;
            const int st = prm[b].status;
            status[b] = st;
            n_out[b] = st == VOX_PASSTHROUGH ? n[b] : c;
        }
        __syncthreads();
        if (tid == 255) s_carry = excl + c;
        __syncthreads();
    }
    if (tid == 0) *n_vox_total = s_carry;
}

__global__ __launch_bounds__(256) void vox_copy_kernel(const float4 *in, const int *n, int stride, const VoxelParams *prm, float4 *out)
{
    const int b = blockIdx.y;
    if (prm[b].status != VOX_PASSTHROUGH) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n[b]; i += gridDim.x * 256) out[(size_t)b * stride + i] = in[(size_t)b * stride + i];
}

// ---- one workgroup per cloud (round 3) -----------------------------------------------------------------------------------
// The sequential mapping loop filters half a dozen SMALL clouds per frame (a frame's few hundred corner features, its ~17 k
// surface features, the frame and the match-buffer clouds of the history: laser_mapping.hpp:1367-1373, 1434-1437, 533-537),
// one cloud per call -- and the pipeline above costs ~16 launches per call whatever the size (its kernels take 2 - 15 us each;
// the launches and their boundaries are the cost).  For a cloud of up to 1024 x ITEMS points one 1024-thread workgroup does
// the whole filter: bounding box -> leaf indices -> block radix sort of (leaf index, point index) in LDS, stable, so the points
// of a voxel stay in input order -> voxel heads -> one thread per voxel adds its points in input order.  Same arithmetic per
// point and per voxel as the kernels above (voxel_params / voxel_index, float sums in input order, one division per
// component), so the output is bit for bit the general path's (tests/test_gpu_voxel.py runs both).
#define VB_THREADS 1024
template <int ITEMS>
__global__ __launch_bounds__(VB_THREADS) void vox_block_kernel(const float4 *in, const int *n, int stride, float inv0, float inv1, float inv2,
                                                                float4 *out, int *n_out, int *status, int n_lo, int n_hi)
{
    // The instantiation is chosen by the cloud's SIZE, which only the device knows: the host launches the ladder of instantiations its
    // capacity allows, and each takes the clouds with n_lo < n <= n_hi (a 300-point corner cloud in a 24 000-point buffer is then
    // sorted 4 items per thread, not 24: 19 us instead of 70 in the mapping loop)
    {
        const int nb_ = n[blockIdx.x] < stride ? n[blockIdx.x] : stride;
        if (!(nb_ > n_lo && nb_ <= n_hi)) return;
    }
    typedef hipcub::BlockRadixSort<unsigned int, VB_THREADS, ITEMS, unsigned short> Sort;
    constexpr int CAP = VB_THREADS * ITEMS;
    __shared__ union {
        typename Sort::TempStorage sort;
        struct {
            unsigned short vals[CAP];      // point index at every sorted position
            unsigned short head_pos[CAP];  // sorted position of the v-th voxel's first point
        } run;
    } sm;
    __shared__ float s_lo[3][VB_THREADS / 64], s_hi[3][VB_THREADS / 64];
    __shared__ int s_cnt[VB_THREADS / 64], s_heads[VB_THREADS / 64];
    __shared__ unsigned int s_last[VB_THREADS], s_maxkey[VB_THREADS / 64];
    __shared__ VoxelParams s_prm;
    __shared__ int s_bits, s_nvox, s_nvalid;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = n[b] < stride ? n[b] : stride;
    const float4 *src = in + (size_t)b * stride;
    float4 *dst = out + (size_t)b * stride;
    const float inv[3] = {inv0, inv1, inv2};

    // ---- bounding box of the finite points (min / max / count: order-independent) -------------------------------------------
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {  // blocked: thread t owns points t * ITEMS .. (the sort keeps that order inside a voxel)
        const int i = tid * ITEMS + u;
        const float4 p = src[i < nb ? i : 0];
        if (i < nb && ll_isfinite(p.x) && ll_isfinite(p.y) && ll_isfinite(p.z)) {
            lo[0] = fminf(lo[0], p.x);
            lo[1] = fminf(lo[1], p.y);
            lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x);
            hi[1] = fmaxf(hi[1], p.y);
            hi[2] = fmaxf(hi[2], p.z);
            cnt++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            lo[d] = fminf(lo[d], __shfl_down(lo[d], off));
            hi[d] = fmaxf(hi[d], __shfl_down(hi[d], off));
        }
        cnt += __shfl_down(cnt, off);
    }
    if (lane == 0) {
        for (int d = 0; d < 3; d++) {
            s_lo[d][wave] = lo[d];
            s_hi[d][wave] = hi[d];
        }
        s_cnt[wave] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        int c = 0;
        for (int w = 0; w < VB_THREADS / 64; w++) {
            for (int d = 0; d < 3; d++) {
                mn[d] = fminf(mn[d], s_lo[d][w]);
                mx[d] = fmaxf(mx[d], s_hi[d][w]);
            }
            c += s_cnt[w];
        }
        VoxelParams prm;
        voxel_params(mn, mx, c, inv, prm);
        s_prm = prm;
        s_nvalid = c;
    }
    __syncthreads();
    const VoxelParams prm = s_prm;
    if (prm.status != VOX_OK) {  // uniform: "leaf size is too small" -> output = input; no finite point -> empty output
        if (prm.status == VOX_PASSTHROUGH)
            for (int i = tid; i < nb; i += VB_THREADS) dst[i] = src[i];
        if (tid == 0) {
            status[b] = prm.status;
            n_out[b] = prm.status == VOX_PASSTHROUGH ? n[b] : 0;
        }
        return;
    }

    // ---- (leaf index, point index), sorted by leaf index; non-finite points and the padding sort last ------------------------
    unsigned int key[ITEMS];
    unsigned short val[ITEMS];
    unsigned int kmax = 0;
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const int i = tid * ITEMS + u;
        const float4 p = src[i < nb ? i : 0];  // (read again, from L1 / L2: 24 points per thread held in registers spilled)
        const bool ok = i < nb && ll_isfinite(p.x) && ll_isfinite(p.y) && ll_isfinite(p.z);
        key[u] = ok ? voxel_index(p.x, p.y, p.z, inv, prm) : 0xffffffffu;
        val[u] = (unsigned short)i;
        if (ok) kmax = key[u] > kmax ? key[u] : kmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int y = (unsigned int)__shfl_down((int)kmax, off);
        kmax = y > kmax ? y : kmax;
    }
    if (lane == 0) s_maxkey[wave] = kmax;
    __syncthreads();
    if (tid == 0) {
        unsigned int m = 0;
        for (int w = 0; w < VB_THREADS / 64; w++) m = s_maxkey[w] > m ? s_maxkey[w] : m;
        int bits = 1;
        while (bits < 32 && (m + 1u) >> bits) bits++;  // the sentinel becomes m + 1: only these bits need sorting
        s_bits = bits;
        s_maxkey[0] = m;
    }
    __syncthreads();
    const int bits = s_bits;
    const unsigned int sentinel = s_maxkey[0] + 1u;  // (m <= 2^31 - 2: voxel_params caps the grid at 2^31 - 1 leaves)
#pragma unroll
    for (int u = 0; u < ITEMS; u++)
        if (key[u] == 0xffffffffu) key[u] = sentinel;
    __syncthreads();
    Sort(sm.sort).Sort(key, val, 0, bits);
    __syncthreads();  // the sort's storage becomes the run arrays

    // ---- voxel heads: a sorted position whose key differs from its predecessor's ----------------------------------------------
    s_last[tid] = key[ITEMS - 1];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) sm.run.vals[tid * ITEMS + u] = val[u];
    __syncthreads();
    unsigned int prev = tid > 0 ? s_last[tid - 1] : sentinel;  // (position 0 is a head whenever it holds a point)
    int heads = 0;
    unsigned int head_mask = 0;  // ITEMS <= 24
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const bool head = key[u] != sentinel && (tid * ITEMS + u == 0 || key[u] != prev);
        if (head) {
            heads++;
            head_mask |= 1u << u;
        }
        prev = key[u];
    }
    int incl = heads;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) s_heads[wave] = incl;
    __syncthreads();
    int rank = incl - heads;
    for (int w = 0; w < wave; w++) rank += s_heads[w];
    if (tid == VB_THREADS - 1) s_nvox = rank + heads;
#pragma unroll
    for (int u = 0; u < ITEMS; u++)
        if (head_mask & (1u << u)) sm.run.head_pos[rank++] = (unsigned short)(tid * ITEMS + u);
    __syncthreads();

    // ---- one thread per voxel: float sums in input order, one division per component (vox_centroid_kernel) -----------------
    const int n_vox = s_nvox, n_valid = s_nvalid;
    for (int v = tid; v < n_vox; v += VB_THREADS) {
        const int j0 = sm.run.head_pos[v], j1 = v + 1 < n_vox ? (int)sm.run.head_pos[v + 1] : n_valid;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        int c = 0;
        for (int j = j0; j < j1; j += 8) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = src[sm.run.vals[j + u < j1 ? j + u : j0]];  // (eight independent gathers)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (j + u < j1) {
                    sx = sx + q[u].x;
                    sy = sy + q[u].y;
                    sz = sz + q[u].z;
                    si = si + q[u].w;
                    c++;
                }
            }
        }
        const float cf = (float)c;
        dst[v] = make_float4(sx / cf, sy / cf, sz / cf, si / cf);
    }
    if (tid == 0) {
        status[b] = VOX_OK;
        n_out[b] = n_vox;
    }
}

#define VXCHK(x)                              \
    do {                                      \
        hipError_t e_ = (x);                  \
        if (e_ != hipSuccess) {               \
            *err = hipGetErrorString(e_);     \
            return -1;                        \
        }                                     \
    } while (0)

int voxel_alloc(VoxelDev &v, int max_clouds, int stride, const char **err)
{
    memset(&v, 0, sizeof(v));
    v.max_clouds = max_clouds;
    v.stride = stride;
    v.block_path = getenv("LL_VOXEL_GENERAL_PATH") ? 0 : 1;  // (A/B and test switch: every cloud through the multi-kernel pipeline)
    const size_t total = (size_t)max_clouds * stride;
    if (total >= 0x7fffffffull) {
        *err = "max_clouds * max_points_per_cloud must stay below 2^31";
        return -1;
    }
    VXCHK(hipMalloc(&v.in, total * sizeof(float4)));
    VXCHK(hipMalloc(&v.out, total * sizeof(float4)));
    VXCHK(hipMalloc(&v.n, max_clouds * sizeof(int)));
    VXCHK(hipMalloc(&v.n_out, max_clouds * sizeof(int)));
    VXCHK(hipMalloc(&v.status, max_clouds * sizeof(int)));
    VXCHK(hipMalloc(&v.n_vox, max_clouds * sizeof(int)));
    VXCHK(hipMalloc(&v.vox_off, max_clouds * sizeof(int)));
    VXCHK(hipMalloc(&v.mm, (size_t)max_clouds * 8 * sizeof(unsigned int)));
    VXCHK(hipMalloc(&v.prm, max_clouds * sizeof(VoxelParams)));
    VXCHK(hipMalloc(&v.keys, total * sizeof(unsigned long long)));
    VXCHK(hipMalloc(&v.keys2, total * sizeof(unsigned long long)));
    VXCHK(hipMalloc(&v.vals, total * sizeof(unsigned int)));
    VXCHK(hipMalloc(&v.vals2, total * sizeof(unsigned int)));
    VXCHK(hipMalloc(&v.is_head, total * sizeof(unsigned int)));
    VXCHK(hipMalloc(&v.rank, total * sizeof(unsigned int)));
    VXCHK(hipMalloc(&v.head_pos, total * sizeof(unsigned int)));
    VXCHK(hipMalloc(&v.n_vox_total, sizeof(int)));
    size_t t1 = 0, t2 = 0;
    VXCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, v.keys, v.keys2, v.vals, v.vals2, (int)total, 0, 64));
    VXCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, v.is_head, v.rank, (int)total));
    v.tmp_bytes = t1 > t2 ? t1 : t2;
    VXCHK(hipMalloc(&v.tmp, v.tmp_bytes));
    return 0;
}

void voxel_free(VoxelDev &v)
{
    void *ptrs[] = {v.in, v.out, v.n, v.n_out, v.status, v.n_vox, v.vox_off, v.mm, v.prm, v.keys, v.keys2, v.vals, v.vals2, v.is_head, v.rank, v.head_pos, v.n_vox_total, v.tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    memset(&v, 0, sizeof(v));
}

// in / n: device pointers ([n_clouds][in_stride] float4, [n_clouds] int); results in v.out / v.n_out / v.status
int voxel_filter(VoxelDev &v, const float4 *in, const int *n, int in_stride, int n_clouds, const float leaf[3], hipStream_t s,
                 const char **err)
{
    if (n_clouds < 1 || n_clouds > v.max_clouds || in_stride > v.stride || in_stride < 1) {
        *err = "cloud count or stride exceeds the capacity of the voxel filter";
        return -1;
    }
    if (!(leaf[0] > 0.f) || !(leaf[1] > 0.f) || !(leaf[2] > 0.f)) {
        *err = "leaf size must be positive";
        return -1;
    }
    const float inv[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};  // inverse_leaf_size_ = 1 / leaf_size_ (float)
    // clouds of up to 24 576 points: one workgroup per cloud (a ladder of three launches, each taking the clouds of its size class).
    // Any number of clouds: a batch of 2 048 voxel-filtered scans is 8 rounds of 70 us for the surface clouds and one of 19 us for the
    // 300-point corner clouds, where the multi-kernel pipeline sorts the whole [clouds][stride] index space -- 49 M mostly padding
    // entries for the corner clouds (round 4 sent only batches of <= 16 clouds here)
    if (v.block_path && in_stride <= VB_THREADS * 24) {
        hipLaunchKernelGGL(vox_block_kernel<4>, dim3(n_clouds), dim3(VB_THREADS), 0, s, in, n, in_stride, inv[0], inv[1], inv[2], v.out, v.n_out, v.status, -1,
                           VB_THREADS * 4);
        if (in_stride > VB_THREADS * 4)
            hipLaunchKernelGGL(vox_block_kernel<8>, dim3(n_clouds), dim3(VB_THREADS), 0, s, in, n, in_stride, inv[0], inv[1], inv[2], v.out, v.n_out, v.status,
                               VB_THREADS * 4, VB_THREADS * 8);
        if (in_stride > VB_THREADS * 8)
            hipLaunchKernelGGL(vox_block_kernel<24>, dim3(n_clouds), dim3(VB_THREADS), 0, s, in, n, in_stride, inv[0], inv[1], inv[2], v.out, v.n_out, v.status,
                               VB_THREADS * 8, VB_THREADS * 24);
        VXCHK(hipGetLastError());
        v.out_stride = in_stride;
        return 0;
    }
    // the sort works on the compact [n_clouds][in_stride] index space
    const size_t total = (size_t)n_clouds * in_stride;
    hipLaunchKernelGGL(vox_init_kernel, dim3((n_clouds + 63) / 64), dim3(64), 0, s, v.mm, v.n_vox, n_clouds);
    const int gx = (in_stride + 255) / 256;
    const int gmm = gx < 64 ? gx : 64;
    hipLaunchKernelGGL(vox_minmax_kernel, dim3(gmm, n_clouds), dim3(256), 0, s, in, n, in_stride, v.mm);
    hipLaunchKernelGGL(vox_params_kernel, dim3((n_clouds + 63) / 64), dim3(64), 0, s, v.mm, n_clouds, inv[0], inv[1], inv[2], v.prm);
    hipLaunchKernelGGL(vox_key_kernel, dim3(gx, n_clouds), dim3(256), 0, s, in, n, in_stride, v.prm, inv[0], inv[1], inv[2], v.keys, v.vals);
    int cloud_bits = 1;
    while ((1 << cloud_bits) < n_clouds + 1 && cloud_bits < 31) cloud_bits++;
    size_t tb = v.tmp_bytes;
    // all 64 bits when the sentinel is present; the leaf index has at most 31 bits, the cloud id cloud_bits: sorting the
    // low 32 + cloud_bits bits orders everything but the sentinel's upper bits, which are all ones (sort last)
    VXCHK(hipcub::DeviceRadixSort::SortPairs(v.tmp, tb, v.keys, v.keys2, v.vals, v.vals2, (int)total, 0, 32 + cloud_bits, s));
    const int gt = (int)((total + 255) / 256);
    hipLaunchKernelGGL(vox_head_kernel, dim3(gt), dim3(256), 0, s, v.keys2, total, v.n_vox, v.is_head);
    tb = v.tmp_bytes;
    VXCHK(hipcub::DeviceScan::ExclusiveSum(v.tmp, tb, v.is_head, v.rank, (int)total, s));
    hipLaunchKernelGGL(vox_headpos_kernel, dim3(gt), dim3(256), 0, s, v.is_head, v.rank, total, v.head_pos);
    hipLaunchKernelGGL(vox_offsets_kernel, dim3(1), dim3(256), 0, s, v.n_vox, n, v.prm, n_clouds, v.vox_off, v.n_out, v.status, v.n_vox_total);
    hipLaunchKernelGGL(vox_centroid_kernel, dim3(gt), dim3(256), 0, s, in, in_stride, v.keys2, v.vals2, v.head_pos, v.vox_off, v.n_vox_total, total,
                       v.out);
    hipLaunchKernelGGL(vox_copy_kernel, dim3(gmm, n_clouds), dim3(256), 0, s, in, n, in_stride, v.prm, v.out);
    VXCHK(hipGetLastError());
    v.out_stride = in_stride;
    return 0;
}

}  // namespace ll
