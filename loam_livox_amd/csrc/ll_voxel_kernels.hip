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
