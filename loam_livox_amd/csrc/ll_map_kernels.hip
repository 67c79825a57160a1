// ll_map_kernels.hip -- K5: device search-grid build for the match-buffer clouds, and the stand-alone 5-NN
// query kernel.  Replaces pcl::KdTreeFLANN::setInputCloud (hku-mars/loam_livox source/laser_mapping.hpp:544-545,
// source/point_cloud_registration.hpp:596-597): instead of a pointer-chasing k-d tree the map is counting-sorted
// into a uniform cell grid (x fastest), stored as float4 {x,y,z,original index} so that a query streams
// contiguous 16-byte records.  Build traffic ~ 2*(16+8) B/point, once per map refresh.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>

#include "ll_device.h"
#include "ll_knn_coop.h"

namespace ll {

#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t e_ = (x);                       \
        if (e_ != hipSuccess) {                    \
            *err = hipGetErrorString(e_);          \
            return -1;                             \
        }                                          \
    } while (0)

__global__ void aabb_kernel(const float *raw, int stride, int64_t n, float *mn, float *mx)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = raw[i * stride], y = raw[i * stride + 1], z = raw[i * stride + 2];
        if (ll_isfinite(x) && ll_isfinite(y) && ll_isfinite(z)) {
            lo[0] = fminf(lo[0], x);
            lo[1] = fminf(lo[1], y);
            lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x);
            hi[1] = fmaxf(hi[1], y);
            hi[2] = fmaxf(hi[2], z);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_down(lo[d], off));
            hi[d] = fmaxf(hi[d], __shfl_down(hi[d], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        // float atomics on min/max via int ordering tricks are avoided: one atomic per wave on a tiny array,
        // implemented with CAS loops
        for (int d = 0; d < 3; d++) {
            float old = mn[d];
            while (lo[d] < old) {
                const float prev = __int_as_float(atomicCAS((int *)&mn[d], __float_as_int(old), __float_as_int(lo[d])));
                if (prev == old) break;
                old = prev;
            }
            old = mx[d];
            while (hi[d] > old) {
                const float prev = __int_as_float(atomicCAS((int *)&mx[d], __float_as_int(old), __float_as_int(hi[d])));
                if (prev == old) break;
                old = prev;
            }
        }
    }
}

__global__ void cellkey_kernel(const float *raw, int stride, int64_t n, Grid g, unsigned int ncell, unsigned int *keys,
                               int *vals, int *counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = raw[i * stride], y = raw[i * stride + 1], z = raw[i * stride + 2];
    unsigned int key = ncell;  // non-finite points sort to the end and are dropped
    if (ll_isfinite(x) && ll_isfinite(y) && ll_isfinite(z)) {
        int cx = cell_coord(x, g.ox, g.inv_h), cy = cell_coord(y, g.oy, g.inv_h), cz = cell_coord(z, g.oz, g.inv_h);
        cx = min(max(cx, 0), g.nx - 1);
        cy = min(max(cy, 0), g.ny - 1);
        cz = min(max(cz, 0), g.nz - 1);
        key = (unsigned int)((cz * g.ny + cy) * g.nx + cx);
        atomicAdd(&counts[key], 1);
    }
    keys[i] = key;
    vals[i] = (int)i;
}

// (n_valid is read on the device -- cell_start[ncell], the scan's total -- so the host does not have to wait for it between the sort and
//  this launch: one stream drain less per build, and the match buffer is rebuilt twice per frame of the mapping loop)
__global__ void gather_kernel(const float *raw, int stride, const int *d_n_valid, const int *vals_sorted, f4 *pts)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (int64_t)*d_n_valid) return;
    const int i = vals_sorted[j];
    f4 p;
    p.x = raw[(int64_t)i * stride];
    p.y = raw[(int64_t)i * stride + 1];
    p.z = raw[(int64_t)i * stride + 2];
    p.w = __int_as_float(i);
    pts[j] = p;
}

// grow-only device buffer: reallocated (with 25 % headroom) when `need` elements do not fit
template <typename T>
static int grow(T **p, size_t *cap, size_t need, const char **err)
{
    if (need <= *cap && *p) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 16;
    HIPCHK(hipMalloc((void **)p, want * sizeof(T)));
    *cap = want;
    return 0;
}

int map_build(MapKind &mk, const float *d_raw, int stride, int64_t n, float cell, hipStream_t s, const char **err)
{
    if (mk.pts16) (void)hipFree(mk.pts16);  // a previous fp16 conversion does not survive a rebuild
    if (mk.perm) (void)hipFree(mk.perm);
    mk.pts16 = nullptr;
    mk.perm = nullptr;
    mk.n = n;
    mk.n_valid = 0;
    if (!mk.b_mm) HIPCHK(hipMalloc((void **)&mk.b_mm, 6 * sizeof(float)));
    float *d_mm = mk.b_mm;
    const float init[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    HIPCHK(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, s));
    if (n > 0)  // (a match buffer of a few thousand points: 1024 mostly idle workgroups contending on six atomics cost 12 us per rebuild)
        hipLaunchKernelGGL(aabb_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, s, d_raw, stride, n, d_mm, d_mm + 3);
    float mm[6];
    HIPCHK(hipMemcpyAsync(mm, d_mm, sizeof(mm), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    Grid g{};
    if (!(mm[0] <= mm[3])) {  // no finite point
        mm[0] = mm[1] = mm[2] = 0.f;
        mm[3] = mm[4] = mm[5] = 0.f;
    }
    // grow the cell until the dense table fits (<= 2^27 cells)
    float h = cell;
    for (;;) {
        const double nx = floor((double)(mm[3] - mm[0]) / h) + 1, ny = floor((double)(mm[4] - mm[1]) / h) + 1,
                     nz = floor((double)(mm[5] - mm[2]) / h) + 1;
        if (nx * ny * nz <= (double)(1u << 27)) {
            g.nx = (int)nx;
            g.ny = (int)ny;
            g.nz = (int)nz;
            break;
        }
        h *= 1.5f;
    }
    g.h = h;
    g.inv_h = 1.0f / h;
    g.ox = mm[0];
    g.oy = mm[1];
    g.oz = mm[2];
    const float ext = fmaxf(fmaxf(fabsf(mm[0]), fabsf(mm[3])), fmaxf(fmaxf(fabsf(mm[1]), fabsf(mm[4])), fmaxf(fabsf(mm[2]), fabsf(mm[5])))) +
                      fmaxf(mm[3] - mm[0], fmaxf(mm[4] - mm[1], mm[5] - mm[2]));
    g.slack = 1e-3f * h + 2e-6f * ext;
    g.guard = 0.0f;
    const size_t ncell = (size_t)g.nx * g.ny * g.nz;
    mk.ncell = ncell;

    const size_t nn = (size_t)(n > 0 ? n : 1);
    {   // per-point scratch and the cell tables; buffers of one group share a capacity (same need, same growth)
        size_t c[4] = {mk.cap_n, mk.cap_n, mk.cap_n, mk.cap_n}, k[2] = {mk.cap_cells, mk.cap_cells};
        mk.cap_n = mk.cap_cells = 0;  // stays 0 if an allocation fails half way: the next build starts over
        if (grow(&mk.b_keys, &c[0], nn, err) || grow(&mk.b_keys2, &c[1], nn, err) || grow(&mk.b_vals, &c[2], nn, err) ||
            grow(&mk.b_vals2, &c[3], nn, err) || grow(&mk.b_counts, &k[0], ncell + 1, err) || grow(&mk.cell_start, &k[1], ncell + 1, err))
            return -1;
        mk.cap_n = c[0];
        mk.cap_cells = k[0];
    }
    unsigned int *d_keys = mk.b_keys, *d_keys2 = mk.b_keys2;
    int *d_vals = mk.b_vals, *d_vals2 = mk.b_vals2, *d_counts = mk.b_counts;
    HIPCHK(hipMemsetAsync(d_counts, 0, (ncell + 1) * sizeof(int), s));
    if (n > 0)
        hipLaunchKernelGGL(cellkey_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_raw, stride, n, g,
                           (unsigned int)ncell, d_keys, d_vals, d_counts);
    // exclusive scan of counts -> cell_start[0..ncell]
    size_t tmp_bytes = 0, tmp2 = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_counts, mk.cell_start, (int)(ncell + 1), s));
    int end_bit = 1;
    while (((size_t)1 << end_bit) <= ncell && end_bit < 32) end_bit++;
    if (n > 0)
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp2, d_keys, d_keys2, d_vals, d_vals2, (int)n, 0, end_bit, s));
    if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
    {
        char *t = (char *)mk.b_tmp;
        size_t c = mk.cap_tmp;
        if (grow(&t, &c, tmp_bytes > 0 ? tmp_bytes : 16, err)) {
            mk.b_tmp = nullptr;
            mk.cap_tmp = 0;
            return -1;
        }
        mk.b_tmp = t;
        mk.cap_tmp = c;
    }
    void *d_tmp = mk.b_tmp;
    size_t tb = tmp_bytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, d_counts, mk.cell_start, (int)(ncell + 1), s));
    tb = tmp_bytes;
    // stable LSD radix sort: points of one cell stay in ascending original-index order
    if (n > 0) HIPCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, (int)n, 0, end_bit, s));
    int n_valid = 0;
    HIPCHK(hipMemcpyAsync(&n_valid, mk.cell_start + ncell, sizeof(int), hipMemcpyDeviceToHost, s));
    if (grow(&mk.pts, &mk.cap_pts, nn, err)) return -1;  // (room for every point given: n_valid <= n is known only after the drain below)
    if (n > 0)
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_raw, stride, mk.cell_start + ncell, d_vals2, mk.pts);
    HIPCHK(hipStreamSynchronize(s));
    mk.n_valid = n_valid;
    g.pts = mk.pts;
    g.cell_start = mk.cell_start;
    mk.grid = g;
    return 0;
}

void map_free(MapKind &mk)
{
    void *ptrs[] = {mk.pts, mk.cell_start, mk.pts16, mk.perm, mk.b_keys, mk.b_keys2, mk.b_vals, mk.b_vals2, mk.b_counts, mk.b_tmp, mk.b_mm};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    mk = MapKind{};
}

// ---- fp16-point records (BASELINE config C5) ---------------------------------------------------------------------
// A point is stored as its position inside its cell, as a fraction of the cell size in binary16 (error <= 2^-11 cell
// sizes, 0.3 mm at 0.6 m), plus the low 16 bits of the cell's x index; y and z cell indices follow from the row being
// scanned.  Distances are accumulated in fp32 on the dequantised coordinates
//     x^ = ox + ((float)cx + (float)fx) * h          (three separate fp32 operations, no contraction)
// so the result is the exact 5-NN of the dequantised cloud (ll_map_dequantized returns it for the checker).
__device__ __forceinline__ float f16_bits_to_float(unsigned int b) { return __half2float(__ushort_as_half((unsigned short)b)); }

struct PtF16 {
    struct Row {
        float fy, fz;  // (float)cy, (float)cz of the row
    };
    static __device__ __forceinline__ Row row(const Grid &g, int c_lo)
    {
        const int rw = c_lo / g.nx;
        Row r;
        r.fy = (float)(rw % g.ny);
        r.fz = (float)(rw / g.ny);
        return r;
    }
    static __device__ __forceinline__ void load(const Grid &g, const Row &row, int j, float &x, float &y, float &z, int &tok)
    {
        const unsigned long long rec = g.pts16[j];
        const float cx = (float)(unsigned int)((rec >> 48) & 0xffffull);
        x = g.ox + (cx + f16_bits_to_float((unsigned int)(rec & 0xffffull))) * g.h;
        y = g.oy + (row.fy + f16_bits_to_float((unsigned int)((rec >> 16) & 0xffffull))) * g.h;
        z = g.oz + (row.fz + f16_bits_to_float((unsigned int)((rec >> 32) & 0xffffull))) * g.h;
        tok = 0;
    }
    // the original index (the tie-break key) lives in a separate array and is only fetched for candidates that can enter
    static __device__ __forceinline__ void push(const Grid &g, Knn5 &r, float d2, int j, int)
    {
        if (d2 > r.d2[4])
            r.lb2 = fminf(r.lb2, d2);
        else
            knn5_push(r, d2, g.perm[j], j);
    }
};

__global__ void f16_convert_kernel(Grid g, const f4 *pts, int n_valid, unsigned int ncell, unsigned long long *pts16, int *perm)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_valid) return;
    const f4 p = pts[j];
    // the cell the point was sorted into (same arithmetic as cellkey_kernel)
    int cx = cell_coord(p.x, g.ox, g.inv_h), cy = cell_coord(p.y, g.oy, g.inv_h), cz = cell_coord(p.z, g.oz, g.inv_h);
    cx = min(max(cx, 0), g.nx - 1);
    cy = min(max(cy, 0), g.ny - 1);
    cz = min(max(cz, 0), g.nz - 1);
    const float fx = fminf(fmaxf((p.x - g.ox) * g.inv_h - (float)cx, 0.0f), 1.0f);
    const float fy = fminf(fmaxf((p.y - g.oy) * g.inv_h - (float)cy, 0.0f), 1.0f);
    const float fz = fminf(fmaxf((p.z - g.oz) * g.inv_h - (float)cz, 0.0f), 1.0f);
    const unsigned long long hx = __half_as_ushort(__float2half_rn(fx)), hy = __half_as_ushort(__float2half_rn(fy)),
                             hz = __half_as_ushort(__float2half_rn(fz));
    pts16[j] = hx | (hy << 16) | (hz << 32) | ((unsigned long long)(cx & 0xffff) << 48);
    perm[j] = __float_as_int(p.w);
}

__global__ void f16_dequant_kernel(Grid g, int n_valid, float *out_xyz)
{
    // one thread per cell row would know (cy, cz) for free; a binary search over cell_start is simpler and this is a
    // checker path: find the cell of record j
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_valid) return;
    const long long ncell = (long long)g.nx * g.ny * g.nz;
    long long lo = 0, hi = ncell;  // cell_start[lo] <= j < cell_start[hi]
    while (hi - lo > 1) {
        const long long mid = (lo + hi) >> 1;
        if (g.cell_start[mid] <= j) lo = mid; else hi = mid;
    }
    const PtF16::Row row = PtF16::row(g, (int)lo);
    float x, y, z;
    int tok;
    PtF16::load(g, row, j, x, y, z, tok);
    const int i = g.perm[j];
    out_xyz[(size_t)i * 3] = x;
    out_xyz[(size_t)i * 3 + 1] = y;
    out_xyz[(size_t)i * 3 + 2] = z;
}

int map_to_f16(MapKind &mk, hipStream_t s, const char **err)
{
    if (!mk.pts) {
        *err = "map kind not uploaded";
        return -1;
    }
    if (mk.grid.nx > 65536) {
        *err = "fp16-point records hold 16 bits of the cell x index: grid too wide";
        return -1;
    }
    const size_t nv = (size_t)(mk.n_valid > 0 ? mk.n_valid : 1);
    HIPCHK(hipMalloc(&mk.pts16, nv * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&mk.perm, nv * sizeof(int)));
    if (mk.n_valid > 0)
        hipLaunchKernelGGL(f16_convert_kernel, dim3((unsigned)((mk.n_valid + 255) / 256)), dim3(256), 0, s, mk.grid, mk.pts, (int)mk.n_valid,
                           (unsigned int)mk.ncell, mk.pts16, mk.perm);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(mk.pts));
    mk.pts = nullptr;
    mk.cap_pts = 0;
    mk.grid.pts = nullptr;
    mk.grid.pts16 = mk.pts16;
    mk.grid.perm = mk.perm;
    mk.grid.slack += mk.grid.h * 9.8e-4f;  // 2^-10 cell sizes: a dequantised point may sit that far outside its cell
    return 0;
}

int map_f16_dequant(const MapKind &mk, float *d_out_xyz, hipStream_t s, const char **err)
{
    if (!mk.pts16) {
        *err = "not an fp16-point map";
        return -1;
    }
    if (mk.n_valid > 0)
        hipLaunchKernelGGL(f16_dequant_kernel, dim3((unsigned)((mk.n_valid + 255) / 256)), dim3(256), 0, s, mk.grid, (int)mk.n_valid, d_out_xyz);
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

__global__ void knn5_f16_kernel(Grid g, const float *q, int nq, float max_d2, int *idx, float *d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    Knn5 r;
    knn5_search_t<PtF16>(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        idx[5 * i + k] = (r.idx[k] == LL_KNN_EMPTY) ? -1 : r.idx[k];
        d2[5 * i + k] = r.d2[k];
    }
}

__global__ void knn5_kernel(Grid g, const float *q, int nq, float max_d2, int *idx, float *d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    Knn5 r;
    knn5_search(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        idx[5 * i + k] = (r.idx[k] == LL_KNN_EMPTY) ? -1 : r.idx[k];
        d2[5 * i + k] = r.d2[k];
    }
}

// one wavefront per query (ll_knn_coop.h): a small batch of queries is bound by the longest dependent-load chain of a single
// lane's search, not by throughput
__global__ __launch_bounds__(256) void knn5_coop_kernel(Grid g, const float *q, int nq, float max_d2, int *idx, float *d2)
{
    const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (i >= nq) return;  // (whole wavefronts)
    Knn5 r;
    knn5_search_coop(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            idx[5 * i + k] = (r.idx[k] == LL_KNN_EMPTY) ? -1 : r.idx[k];
            d2[5 * i + k] = r.d2[k];
        }
    }
}

void launch_knn5(const Grid &g, const float *d_q, int nq, float max_d2, int *d_idx, float *d_d2, hipStream_t s)
{
    if (nq <= 0) return;
    if (!g.pts16 && nq <= LL_KNN_COOP_MAX_QUERIES)
        hipLaunchKernelGGL(knn5_coop_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, g, d_q, nq, max_d2, d_idx, d_d2);
    else if (g.pts16)
        hipLaunchKernelGGL(knn5_f16_kernel, dim3((nq + 127) / 128), dim3(128), 0, s, g, d_q, nq, max_d2, d_idx, d_d2);
    else
        hipLaunchKernelGGL(knn5_kernel, dim3((nq + 127) / 128), dim3(128), 0, s, g, d_q, nq, max_d2, d_idx, d_d2);
}

}  // namespace ll
