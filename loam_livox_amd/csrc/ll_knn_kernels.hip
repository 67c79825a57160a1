// ll_knn_kernels.hip -- tile search of the surface queries (gfx950, wave64): the 5-NN call site of
// hku-mars/loam_livox source/point_cloud_registration.hpp:351-353 for scans whose queries share map cells by the hundred.
//
//   reg_qsort_kernel    : at ICP iterations 0 and 1 of a registration: the surface queries of a scan, transformed with the
//                         current pose, sorted by the map cell they fall into (one workgroup per scan, block radix sort in LDS
//                         over the bits of the scan's own cell box) -> rd.qperm.  The order is only a grouping: every result
//                         goes to its query's own slot, so nothing depends on it.  After the first pose update -- the large
//                         one -- it is kept (later updates of centimetres move the queries of a cell together; measured at
//                         B = 256: sort at 0 only -> search 3.07 ms per step, at 0 and 1 -> 2.77, at 0, 1, 2 -> 2.83).
//   reg_knn_tile_kernel : one wavefront per 64 consecutive queries of that order: the map points of the cells' common
//                         neighbourhood staged in LDS, every lane offers every staged point to its top five (ll_knn_tile.h),
//                         then the residual-block constants of the same slot.  Lanes the tile cannot settle (0.1 % on the
//                         C2 map) run the per-lane search.  Without motion deblur the pose transform of the query
//                         (pointAssociateToMap, :622-661) happens in here too, and the scan's corner queries (a few hundred,
//                         per-lane ring searches on the sparse corner map) ride in the first workgroups of the same launch:
//                         one launch per ICP iteration instead of transform + corner search + surface search + build.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <limits.h>

#include "ll_knn_tile.h"
#include "ll_reg_query.h"

namespace ll {

#define QS_THREADS 1024
#define QS_BINS 32768  // cells of a scan's box the counting sort takes (128 KB of LDS bins); larger boxes take the radix sort
// the query position the sort and the tile kernel agree on: FUSED = transformed here from the extractor's feature cloud with the
// scan's current pose (no motion deblur), else read from rd.qw (reg_transform_kernel has run)
// transform_query without motion deblur (the only case the fused kernels serve): pointAssociateToMap's plain branch, PCR:629
__device__ __forceinline__ void transform_plain(const RegState *st, const float4 &f, float pw[3])
{
    pw[0] = pw[1] = pw[2] = NAN;  // non-finite features are skipped like in transform_query
    if (!(ll_isfinite(f.x) && ll_isfinite(f.y) && ll_isfinite(f.z))) return;
    point_to_map(st->pose_curr, f.x, f.y, f.z, pw);
}

template <bool FUSED>
__device__ __forceinline__ float4 tile_query_pos(const RegDev &rd, const RegConst &rc, const RegState *st, int b, int q, int nS)
{
    if (!FUSED) return rd.qw[(size_t)b * rd.cap + rd.cap_c + q];
    float pw[3];
    transform_plain(st, load_feature(rd, b, 1, q), pw);
    // a13: a skipped feature is handed on as a non-finite query -- no neighbours, no block (PCR:339-345)
    if (subsample_skip_feature(rc.subsample_seed, 1, st->icp_iters, q, nS, rc.max_blocks)) pw[0] = pw[1] = pw[2] = NAN;
    return make_float4(pw[0], pw[1], pw[2], 0.f);
}

template <int ITEMS, bool FUSED>
__global__ __launch_bounds__(QS_THREADS) void reg_qsort_kernel(RegDev rd, RegConst rc, Grid gs)
{
    typedef hipcub::BlockRadixSort<unsigned int, QS_THREADS, ITEMS, unsigned short> Sort;
    typedef hipcub::BlockScan<int, QS_THREADS> Scan;
    // the radix sort's exchange buffers and the counting sort's bins share the same LDS
    constexpr size_t RAW = sizeof(typename Sort::TempStorage) > (QS_BINS + 1) * sizeof(int) ? sizeof(typename Sort::TempStorage) : (QS_BINS + 1) * sizeof(int);
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RAW];
    typename Sort::TempStorage &sort = *reinterpret_cast<typename Sort::TempStorage *>(s_raw);
    int *s_hist = reinterpret_cast<int *>(s_raw);
    __shared__ typename Scan::TempStorage scan;
    __shared__ int s_lo[3][QS_THREADS / 64], s_hi[3][QS_THREADS / 64];
    __shared__ int s_box[6];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RegState *st = rd.state + b;
    if (st->done) return;
    // Scans of more than LL_KNN_TILE_SEG surface queries (Mid-100: three heads, ~50 k) are sorted in SEGMENTS of that many consecutive
    // queries, one workgroup each (blockIdx.y): the order only has to put the 64 queries of a wavefront into neighbouring cells, and a
    // segment of a sweep is as compact as the sweep.  perm holds segment-relative indices (16 bits); the tile kernel adds the base.
    const int q0 = (int)blockIdx.y * LL_KNN_TILE_SEG;
    const int nS_all = rd.n_surf[b];
    const int nS = nS_all - q0 < LL_KNN_TILE_SEG ? nS_all - q0 : LL_KNN_TILE_SEG;  // queries of this segment
    if (nS <= 0) return;
    // ---- box of the cells the scan's queries fall into (striped reads: coalesced; any initial order is as good as another)
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
    // the cell of every query, kept from this pass when the grid's dimensions fit 10 bits each (every map of a few hundred metres):
    // the pose transform (fp64) and the cell lookup then run once per query instead of once per pass
    const bool packable = gs.nx <= 1024 && gs.ny <= 1024 && gs.nz <= 1024;
    unsigned int key[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const int i = u * QS_THREADS + tid;
        key[u] = 0u;
        if (i < nS) {
            const float4 p = tile_query_pos<FUSED>(rd, rc, st, b, q0 + i, nS_all);
            TileQ tq;
            tile_query(gs, p.x, p.y, p.z, tq);
            if (tq.ingrid) {
                key[u] = 0x80000000u | ((unsigned int)tq.cz << 20) | ((unsigned int)tq.cy << 10) | (unsigned int)tq.cx;
                lo[0] = min(lo[0], tq.cx);
                lo[1] = min(lo[1], tq.cy);
                lo[2] = min(lo[2], tq.cz);
                hi[0] = max(hi[0], tq.cx);
                hi[1] = max(hi[1], tq.cy);
                hi[2] = max(hi[2], tq.cz);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            lo[d] = min(lo[d], __shfl_down(lo[d], off));
            hi[d] = max(hi[d], __shfl_down(hi[d], off));
        }
    }
    if (lane == 0) {
        for (int d = 0; d < 3; d++) {
            s_lo[d][wave] = lo[d];
            s_hi[d][wave] = hi[d];
        }
    }
    __syncthreads();
    if (tid < 3) {
        int mn = INT_MAX, mx = -1;
        for (int w = 0; w < QS_THREADS / 64; w++) {
            mn = min(mn, s_lo[tid][w]);
            mx = max(mx, s_hi[tid][w]);
        }
        s_box[tid] = mn;
        s_box[3 + tid] = mx;
    }
    __syncthreads();
    const int bx0 = s_box[0], by0 = s_box[1], bz0 = s_box[2];
    const long long ex = (long long)s_box[3] - bx0 + 1, ey = (long long)s_box[4] - by0 + 1, ez = (long long)s_box[5] - bz0 + 1;
    const bool any = s_box[3] >= 0;
    // keys are cell indices inside the box, x fastest like the map's own order (neighbours in the order are neighbours in a
    // row); a box too large for 30 bits (a scan scattered over kilometres) degrades to one key: correct, just not grouped
    const bool wide = !any || ex * ey * ez > (1ll << 30);
    const unsigned int kmax = wide ? 0u : (unsigned int)(ex * ey * ez - 1);
    unsigned short val[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const int i = u * QS_THREADS + tid;
        unsigned int k = kmax + 2u;  // padding: behind everything
        if (i < nS) {
            bool ingrid;
            int cx, cy, cz;
            if (packable) {  // (uniform)
                ingrid = (key[u] >> 31) != 0u;
                cx = (int)(key[u] & 1023u);
                cy = (int)((key[u] >> 10) & 1023u);
                cz = (int)((key[u] >> 20) & 1023u);
            } else {
                const float4 p = tile_query_pos<FUSED>(rd, rc, st, b, q0 + i, nS_all);  // (again: ITEMS points held in registers would spill)
                TileQ tq;
                tile_query(gs, p.x, p.y, p.z, tq);
                ingrid = tq.ingrid;
                cx = tq.cx, cy = tq.cy, cz = tq.cz;
            }
            k = kmax + 1u;  // not in the grid / not finite: behind the grouped ones (they take the per-lane search anyway)
            if (ingrid) k = wide ? 0u : (unsigned int)(((long long)(cz - bz0) * ey + (cy - by0)) * ex + (cx - bx0));
        }
        key[u] = k;
        val[u] = (unsigned short)i;
    }
    unsigned short *perm = rd.qperm + (size_t)b * rd.cap_s + q0;
    float4 *qs = rd.qsorted + (size_t)b * rd.cap_s + q0;  // the features in the same order (FUSED: the tile kernel reads them instead of gathering)
    if (kmax + 2u <= (unsigned int)QS_BINS) {
        // ---- counting sort (uniform branch: the scan's cell box has at most QS_BINS cells -- every C2 scan): one LDS histogram over
        //      the box, the atomic's return value is the query's rank inside its cell, one scan over the bins, one scatter.  The
        //      order INSIDE a cell is the order the atomics were served in -- it differs from run to run and nothing depends on it
        //      (the order is a grouping; every result goes to its query's own slot and is exact whatever the grouping).
        const int nbins = (int)kmax + 2;  // cells of the box + one bin for the queries outside the grid
        for (int e = tid; e < nbins; e += QS_THREADS) s_hist[e] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ITEMS; u++) {
            const int i = u * QS_THREADS + tid;
            if (i < nS) val[u] = (unsigned short)atomicAdd(&s_hist[key[u]], 1);
        }
        __syncthreads();
        const int per = (nbins + QS_THREADS - 1) / QS_THREADS, e0 = tid * per;
        int sum = 0;
        for (int e = e0; e < e0 + per && e < nbins; e++) sum += s_hist[e];
        int base;
        Scan(scan).ExclusiveSum(sum, base);
        for (int e = e0; e < e0 + per && e < nbins; e++) {
            const int c = s_hist[e];
            s_hist[e] = base;
            base += c;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ITEMS; u++) {
            const int i = u * QS_THREADS + tid;
            if (i < nS) {
                const int dst = s_hist[key[u]] + (int)val[u];
                perm[dst] = (unsigned short)i;
                if (FUSED) qs[dst] = load_feature(rd, b, 1, q0 + i);
            }
        }
        return;
    }
    int bits = 1;
    while (bits < 32 && ((kmax + 2u) >> bits)) bits++;
    Sort(sort).Sort(key, val, 0, bits);
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const int pos = tid * ITEMS + u;  // (blocked arrangement after the sort)
        if (pos < nS) {
            perm[pos] = val[u];
            if (FUSED) qs[pos] = load_feature(rd, b, 1, q0 + (int)val[u]);
        }
    }
}

// The corner queries of a scan (a few hundred) in the order of the corner map's cells, so that the 64 queries of a wavefront of the lane
// kernel share one or two tiles instead of a dozen (in feature order a wavefront's queries lie all over the room: ten to twenty rounds of
// ~100 candidates each, no faster than a ring search per lane).  One workgroup per scan, block radix sort over the bits of the grid's
// cell count; scans with more than LL_QSORT_CORNER_MAX corner queries keep the feature order (the lane kernel does not read the
// order then).  Same iterations as the surface sort.
#define QC_THREADS 256
#define QC_ITEMS (LL_QSORT_CORNER_MAX / QC_THREADS)
template <bool FUSED>
__global__ __launch_bounds__(QC_THREADS) void reg_qsort_corner_kernel(RegDev rd, RegConst rc, Grid gc)
{
    typedef hipcub::BlockRadixSort<unsigned int, QC_THREADS, QC_ITEMS, unsigned short> Sort;
    __shared__ typename Sort::TempStorage sort;
    const int b = blockIdx.x, tid = threadIdx.x;
    const RegState *st = rd.state + b;
    const int nC = rd.n_corner[b];
    if (st->done || nC > LL_QSORT_CORNER_MAX) return;
    const size_t sb = (size_t)b * rd.cap;
    const long long ncell = (long long)gc.nx * gc.ny * gc.nz;
    unsigned int key[QC_ITEMS];
    unsigned short val[QC_ITEMS];
#pragma unroll
    for (int u = 0; u < QC_ITEMS; u++) {
        const int i = tid * QC_ITEMS + u;  // (blocked arrangement)
        unsigned int k = 0xffffffffu;      // padding: behind everything
        if (i < nC) {
            float4 p;
            if (FUSED) {
                float o[3];
                transform_plain(st, load_feature(rd, b, 0, i), o);
                p = make_float4(o[0], o[1], o[2], 0.f);
            } else {
                p = rd.qw[sb + i];
            }
            TileQ tq;
            tile_query(gc, p.x, p.y, p.z, tq);
            // (a grid of more than 2^31 cells degrades to one key: correct, just not grouped)
            k = !tq.ingrid ? 0xfffffffeu : (ncell > 0x7fffffffll ? 0u : (unsigned int)((tq.cz * gc.ny + tq.cy) * gc.nx + tq.cx));
        }
        key[u] = k;
        val[u] = (unsigned short)i;
    }
    Sort(sort).Sort(key, val);
    unsigned short *perm = rd.qperm_c + (size_t)b * LL_QSORT_CORNER_MAX;
#pragma unroll
    for (int u = 0; u < QC_ITEMS; u++) {
        const int pos = tid * QC_ITEMS + u;
        if (pos < nC) perm[pos] = val[u];
    }
}

#define KT_THREADS 256
// One-dimensional grid, surf_blocks workgroups per scan.  The kernel holds the tile search and nothing else (80 VGPRs: six wavefronts per
// SIMD): what a lane cannot finish here -- the ring search of a query the tile does not settle (sparse surroundings, an exact tie, more
// than a cell outside the grid: a dozen per scan), or the fp64 block constants of a scan without a plane table -- goes onto the scan's work list
// (rd.work_search, surface segment; rd.work_cnt[4 b + 2 + (iter & 1)]) and reg_knn_lane_kernel, the next launch, takes it
// (round 5 ran those lanes in here: 117 VGPRs, four wavefronts per SIMD, and 3 % of the wavefronts executed a ring search for a third
// of their lanes -- nearly all of them queries on the map's outer walls, a centimetre outside the grid: tile_query now adopts the
// nearest cell for them).
#define KT_LIST_BUILD_ONLY 0x80000000u  // list entry: slot | this bit when the neighbours are stored and only build_one is left
template <bool FUSED>
__global__ __launch_bounds__(KT_THREADS) __attribute__((amdgpu_waves_per_eu(FUSED ? 6 : 5, 8)))  // (the gather of rd.qw costs the sixth)
void reg_knn_tile_kernel(RegDev rd, RegConst rc, Grid gs, int iter, int surf_blocks)
{
    __shared__ float4 s_tile[KT_THREADS / 64][LL_TILE_CAP + 4];
    const int bid = blockIdx.x;
    const int b = bid / surf_blocks, sblk = bid - b * surf_blocks;
    const RegState *st = rd.state + b;
    const int i = sblk * KT_THREADS + threadIdx.x;  // position in the scan's cell order
    // The order entry and the sorted feature are read side by side (round 5: order -> feature gather, two dependent round trips), with a
    // clamped index ahead of the bounds test (the arrays hold cap_s entries per scan).  (Hoisting the pose and the done / count words
    // into the same batch with explicit scalar loads was measured: no change at six wavefronts per SIMD.)
    const int done = st->done, nS = rd.n_surf[b], nC = rd.n_corner[b];
    const int ic = i < rd.cap_s ? i : rd.cap_s - 1;
    const int pq = (int)rd.qperm[(size_t)b * rd.cap_s + ic];
    float4 feat = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FUSED) feat = rd.qsorted[(size_t)b * rd.cap_s + ic];
    if (done || (i & ~63) >= nS) return;  // (whole wavefronts)
    const size_t sb = (size_t)b * rd.cap;
    const bool valid = i < nS;
    const bool compact = !rc.force_general && (nS + RS_THREADS - 1) / RS_THREADS * RS_THREADS + nC <= LL_TABLE_MAX_BLOCKS;  // scan_is_compact
#ifdef LL_TILE_TIMING
    // instrumented build: wall clocks of this wavefront per phase, added to the scan's RegState::dbg_cycles by lane 0 --
    // 0 tile_query, 1 round set-up, 2 staging, 3 offers, 4 winners + finish test, 5 query position (order + transform), 6 sum of tile
    // sizes, 7 rounds, 8 store of final lanes, 9 list append, 10 block flag, 11 whole wavefront, 12 wavefronts, 13 wavefronts with a
    // listed lane, 14 listed lanes
    long long tt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = clock64();
    long long tw = t_begin;
#endif
    // segment base + index in the segment; the segment is the same for the whole wavefront (LL_KNN_TILE_SEG is a multiple of 64): scalar
    const int seg_base = (__builtin_amdgcn_readfirstlane(i) / LL_KNN_TILE_SEG) * LL_KNN_TILE_SEG;
    const int q = valid ? seg_base + pq : 0;
    const int slot = rd.cap_c + q;
    float4 pw;
    if (FUSED) {
        // transform_plain + the a13 rule of tile_query_pos, from the values read above
        float o[3];
        transform_plain(st, feat, o);
        if (subsample_skip_feature(rc.subsample_seed, 1, st->icp_iters, q, nS, rc.max_blocks)) o[0] = o[1] = o[2] = NAN;
        pw = make_float4(o[0], o[1], o[2], 0.f);
    } else {
        pw = rd.qw[sb + rd.cap_c + q];
    }
    const float max_d2 = rc.max_d2_plane;
    Knn5 r;
    bool fin;
    int degenerate;  // of the plane through neighbours 0, 2, 4 (1 / 0), -1 = not known (tile of several passes)
    LL_TT(5, tw);
    knn5_tile_wave(gs, valid, pw.x, pw.y, pw.z, max_d2, s_tile[threadIdx.x >> 6], r, fin, degenerate LL_TT_PASS);
#ifdef LL_TILE_TIMING
    tw = clock64();
#endif
    bool listed = valid && !fin;  // sparse surroundings, an exact tie, a query outside the grid or not finite: per-lane search
    if (valid && fin) {
        if (rc.debug_knn && iter == rc.debug_knn_iter) {
#pragma unroll
            for (int k = 0; k < 5; k++) r.idx[k] = as_int(gs.pts[r.pos[k]].w);
        }
        knn_finish(rd, rc, sb, slot, 1, iter, pw, max_d2, r);
        LL_TT(8, tw);
        if (!rc.check_plane_pca && rc.icp_plane && compact) {
            // plane-table path (build_one's early return): only the block's flag is decided here, from the three neighbours the lane
            // has just seen in LDS -- no second look at rd.nn or the map
            // (plane_degenerate: |b - a| == 0 or |c - a| == 0 in double  <=>  the float points coincide)
            if (degenerate < 0) {
                const f4 p0 = gs.pts[r.pos[0]], p1 = gs.pts[r.pos[2]], p2 = gs.pts[r.pos[4]];
                degenerate = ((p1.x == p0.x && p1.y == p0.y && p1.z == p0.z) || (p2.x == p0.x && p2.y == p0.y && p2.z == p0.z)) ? 1 : 0;
            }
            rd.blk_flag0[sb + slot] = degenerate ? BLK_NONE : (BLK_PLANE | BLK_ACTIVE | 8);
        } else {
            listed = true;  // block constants in fp64 (or the PCA check): build_one, in the lane kernel
        }
        LL_TT(10, tw);
    }
    if (listed && !fin && FUSED) rd.qw[sb + slot] = pw;
    const unsigned long long lm = __ballot(listed);
    if (lm) {  // (uniform) one atomic per wavefront reserves its entries
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(&rd.work_cnt[4 * b + 2 + (iter & 1)], (int)__popcll(lm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (listed)
            rd.work_search[sb + rd.cap_c + base + (int)__popcll(lm & ((1ull << lane) - 1ull))] = (int)((unsigned int)slot | (fin ? KT_LIST_BUILD_ONLY : 0u));
    }
#ifdef LL_TILE_TIMING
    LL_TT(9, tw);
    tt[13] = lm ? 1 : 0;
    tt[14] = __popcll(lm);
    tt[11] = clock64() - t_begin;
    tt[12] = 1;
    if ((threadIdx.x & 63) == 0 && LL_TILE_TIMING != 2)
        for (int k_ = 0; k_ < 16; k_++) atomicAdd((unsigned long long *)&rd.state[b].dbg_cycles[k_], (unsigned long long)tt[k_]);
#endif
}

// What the tile kernel does not do.  First corner_blocks workgroups per scan (scan fastest): the scan's CORNER queries, 64 per wavefront in
// the order of the corner map's cells, through the same tile search on the corner map: the line radius (1.41 m) lies inside one cell of that map (1.45 m), so
// the 3 x 3 x 3 tile settles every query whose list has no tie -- with five neighbours or with fewer (tile5_finish) --, a handful of rounds
// per wavefront instead of a ring search per lane (round 5 and the first half of round 6: ~100 dependent loads per lane, 78 us per
// launch whatever the batch); then their line constants (build_one).  Then list_blocks workgroups per scan for the surface list the tile
// kernel wrote: the search where it is still due, then the block constants.  A short list -- a dozen queries with their neighbours
// metres away -- is taken one wavefront per entry (ll_knn_coop.h); a long one (a scan without a plane table lists everything) one lane
// per entry.  (The corner queries in the tile kernel itself, as its first workgroups: 43 spilled VGPRs at its 80-register budget.)
#define KL_THREADS 64
#define KL_COOP_PER 4  // lists of up to this many entries per list workgroup are searched one wavefront per entry
template <bool FUSED>
__global__ __launch_bounds__(KL_THREADS) void reg_knn_lane_kernel(RegDev rd, RegConst rc, Grid gc, Grid gs, int iter, int n_scans, int corner_blocks,
                                                                  int list_blocks)
{
    __shared__ float4 s_tile[LL_TILE_CAP + 4];
    const int bid = blockIdx.x, n_corner_wg = corner_blocks * n_scans;
    if (bid < n_corner_wg) {
        const int b = bid % n_scans, cblk = bid / n_scans;
        const RegState *st = rd.state + b;
        const int i = cblk * KL_THREADS + threadIdx.x, nC = rd.n_corner[b];
        if (st->done || cblk * KL_THREADS >= nC) return;  // (whole wavefronts)
        const size_t sb = (size_t)b * rd.cap;
        const bool valid = i < nC;
        const int slot = !valid ? 0 : (nC > LL_QSORT_CORNER_MAX ? i : (int)rd.qperm_c[(size_t)b * LL_QSORT_CORNER_MAX + i]);  // cell order (reg_qsort_corner_kernel)
        float4 pw;
        if (FUSED) {
            float o[3];
            transform_plain(st, load_feature(rd, b, 0, slot), o);
            if (subsample_skip_feature(rc.subsample_seed, 0, st->icp_iters, slot, nC, rc.max_blocks)) o[0] = o[1] = o[2] = NAN;
            pw = make_float4(o[0], o[1], o[2], 0.f);
        } else {
            pw = rd.qw[sb + slot];
        }
        const float max_d2 = rc.max_d2_line;
        Knn5 r;
        bool fin;
        int degenerate;
#ifdef LL_TILE_TIMING
        long long tt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (-DLL_TILE_TIMING=2: the corner wavefronts are clocked instead of the surface ones)
        const long long t_begin = clock64();
#endif
        knn5_tile_wave<true>(gc, valid, pw.x, pw.y, pw.z, max_d2, s_tile, r, fin, degenerate LL_TT_PASS);
#if defined(LL_TILE_TIMING) && LL_TILE_TIMING == 2
        {
            const long long t_tile = clock64();
            const unsigned long long fb = __ballot(valid && !fin);
            if (valid) {
                if (fin) knn_finish(rd, rc, sb, slot, 0, iter, pw, max_d2, r);
                else { if (FUSED) rd.qw[sb + slot] = pw; knn_one(rd, rc, gc, gs, b, slot, iter); }
            }
            const long long t_search = clock64();
            if (valid) build_one(rd, rc, gc, gs, b, slot);
            const long long t_end = clock64();
            tt[5] = t_tile - t_begin;  // transform + tile wave
            tt[8] = t_search - t_tile; // store / per-lane fall-back
            tt[10] = t_end - t_search; // build_one
            tt[11] = t_end - t_begin;
            tt[12] = 1;
            tt[13] = fb ? 1 : 0;
            tt[14] = __popcll(fb);
            if ((threadIdx.x & 63) == 0)
                for (int k_ = 0; k_ < 16; k_++) atomicAdd((unsigned long long *)&rd.state[b].dbg_cycles[k_], (unsigned long long)tt[k_]);
            return;
        }
#endif
        if (!valid) return;
        if (fin) {
            if (rc.debug_knn && iter == rc.debug_knn_iter) {
#pragma unroll
                for (int k = 0; k < 5; k++)
                    if (k < r.count) r.idx[k] = as_int(gc.pts[r.pos[k]].w);
            }
            knn_finish(rd, rc, sb, slot, 0, iter, pw, max_d2, r);
        } else {  // a tie, a query more than a cell outside the grid, not finite
            if (FUSED) rd.qw[sb + slot] = pw;
            knn_one(rd, rc, gc, gs, b, slot, iter);
        }
        build_one(rd, rc, gc, gs, b, slot);
        return;
    }
    const int b = (bid - n_corner_wg) % n_scans, lblk = (bid - n_corner_wg) / n_scans;
    if (rd.state[b].done) return;
    const size_t sb = (size_t)b * rd.cap;
    const int n = rd.work_cnt[4 * b + 2 + (iter & 1)];
    if (lblk == 0 && threadIdx.x == 0) rd.work_cnt[4 * b + 2 + ((iter & 1) ^ 1)] = 0;  // the other counter, for the next ICP iteration's tile kernel
    if (n <= KL_COOP_PER * list_blocks) {
        for (int k = lblk; k < n; k += list_blocks) {
            const unsigned int e = (unsigned int)rd.work_search[sb + rd.cap_c + k];
            const int slot = (int)(e & ~KT_LIST_BUILD_ONLY);
            if (!(e & KT_LIST_BUILD_ONLY)) knn_one_coop(rd, rc, gc, gs, b, slot, iter);  // (uniform)
            if (threadIdx.x == 0) build_one(rd, rc, gc, gs, b, slot);
        }
        return;
    }
    for (int k = lblk * KL_THREADS + threadIdx.x; k < n; k += list_blocks * KL_THREADS) {
        const unsigned int e = (unsigned int)rd.work_search[sb + rd.cap_c + k];
        const int slot = (int)(e & ~KT_LIST_BUILD_ONLY);
        if (!(e & KT_LIST_BUILD_ONLY)) knn_one(rd, rc, gc, gs, b, slot, iter);
        build_one(rd, rc, gc, gs, b, slot);
    }
}

// max_nc > 0: the corner queries are ordered too (they take the tile search of the lane kernel)
void launch_reg_qsort(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int n_scans, int max_nc, int max_ns, bool fused, hipStream_t s)
{
    if (max_nc > 0) {
        if (fused)
            hipLaunchKernelGGL(reg_qsort_corner_kernel<true>, dim3(n_scans), dim3(QC_THREADS), 0, s, rd, rc, gc);
        else
            hipLaunchKernelGGL(reg_qsort_corner_kernel<false>, dim3(n_scans), dim3(QC_THREADS), 0, s, rd, rc, gc);
    }
    const int nseg = (max_ns + LL_KNN_TILE_SEG - 1) / LL_KNN_TILE_SEG;
#define LL_QSORT(ITEMS)                                                                                                        \
    do {                                                                                                                       \
        if (fused)                                                                                                             \
            hipLaunchKernelGGL((reg_qsort_kernel<ITEMS, true>), dim3(n_scans, nseg), dim3(QS_THREADS), 0, s, rd, rc, gs);     \
        else                                                                                                                   \
            hipLaunchKernelGGL((reg_qsort_kernel<ITEMS, false>), dim3(n_scans, nseg), dim3(QS_THREADS), 0, s, rd, rc, gs);    \
    } while (0)
    if (max_ns <= QS_THREADS * 4)
        LL_QSORT(4);
    else if (max_ns <= QS_THREADS * 8)
        LL_QSORT(8);
    else
        LL_QSORT(24);
#undef LL_QSORT
}

// max_nc > 0: the corner queries are searched (tiles on the corner map) and built by the first workgroups of the lane kernel's launch.
// (Measured and withdrawn, HISTORY.md round 6: the corner searches as a launch of their own on a second stream beside the tile search --
//  one batch at a time 43.3 k -> 44.3 k scans/s, three batches in flight 49.9 k -> 48.0 k: the other batches fill the chip already.)
void launch_reg_knn_tile(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int n_scans, int iter, int max_nc, int max_ns,
                         bool fused, hipStream_t s)
{
    const int sbk = (max_ns + KT_THREADS - 1) / KT_THREADS;
    const int cb = (max_nc + KL_THREADS - 1) / KL_THREADS;
    // list workgroups per scan: a dozen queries are listed in the median scan, a hundred in the worst of a C2 batch (one wavefront per entry
    // up to KL_COOP_PER entries per workgroup: a list of 115 at 17 workgroups fell to the per-lane path, one 60 us chain per lane), a scan
    // in the open may list all of its queries
    int lb = (max_ns + 255) / 256;
    lb = lb < 8 ? 8 : (lb > 64 ? 64 : lb);
    // the list counters alternate between two words per scan (ICP iteration parity): the lane kernel clears the one the next iteration
    // appends to, so only a registration's first search needs a memset (ten fill launches per registration were 1 % of a step)
    if (iter == 0) (void)hipMemsetAsync(rd.work_cnt, 0, (size_t)n_scans * 4 * sizeof(int), s);
    if (fused) {
        hipLaunchKernelGGL(reg_knn_tile_kernel<true>, dim3((unsigned int)(sbk * n_scans)), dim3(KT_THREADS), 0, s, rd, rc, gs, iter, sbk);
        hipLaunchKernelGGL(reg_knn_lane_kernel<true>, dim3((unsigned int)((cb + lb) * n_scans)), dim3(KL_THREADS), 0, s, rd, rc, gc, gs, iter, n_scans, cb, lb);
    } else {
        hipLaunchKernelGGL(reg_knn_tile_kernel<false>, dim3((unsigned int)(sbk * n_scans)), dim3(KT_THREADS), 0, s, rd, rc, gs, iter, sbk);
        hipLaunchKernelGGL(reg_knn_lane_kernel<false>, dim3((unsigned int)((cb + lb) * n_scans)), dim3(KL_THREADS), 0, s, rd, rc, gc, gs, iter, n_scans, cb, lb);
    }
}

}  // namespace ll
