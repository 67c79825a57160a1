// ll_reg_kernels.hip -- HIP kernels (gfx950, wave64) of the scan-to-map registrar.
//
//   K6  reg_transform_kernel, reg_knn_kernel, reg_build_kernel (ICP iterations 0 and 1): per query: transform with the
//                              current pose (pointAssociateToMap, point_cloud_registration.hpp:622-661), exact 5-NN on
//                              the cell grid (:249,351), match-radius tests (:254,353), line / plane block constants
//                              (:300-323, :416-423; ceres_icp.hpp:255-256, 328-334)
//       reg_requery_kernel, reg_list_offsets_kernel, reg_list_kernel (ICP iteration >= 2): exact neighbour reuse -- every
//                              query is classified against two displacement budgets (ll_knn_core.h), the few that need a
//                              new search or a re-sort go to dense per-(scan, kind) work lists, one fused kernel searches /
//                              re-sorts them and rebuilds their blocks
//   K8/K9 reg_solve_kernel   : ONE workgroup per scan (a group of LL_GRP workgroups per scan for batches of <= 16 scans,
//                              group_barrier / group_reduce below) runs what the reference does between :460 and :531:
//                              the 2-iteration prerun solve, the loss-corrected L1 evaluation, the
//                              std::set-deduplicated 80-th percentile inlier threshold (:153-161), the prune,
//                              the final solve and the pose composition -- replacing ceres::Solve /
//                              Problem::Evaluate by a 28-value (21 H + 6 g + 1 cost) workgroup reduction and a
//                              Levenberg-Marquardt controller on lane 0.  No host round trip per iteration.
//                              Two forms: solve_fast3 (plane table in LDS, 18 B streamed per plane block; every compact scan)
//                              and solve_general (> 24 576 blocks or motion deblur)
//        reg_finalize_kernel : accept / reject (:559-573)
//        reg_merge_heads_kernel : Mid-100, the feature clouds of a sweep's heads concatenated on the device
//
// No MFMA: 6x6 systems are reduced, not multiplied.  Plane blocks live in HBM as three 16-byte planes (48 B + 1 flag per
// block), line blocks as 65 B; every cost evaluation streams the part of them that does not fit the LDS record cache.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "ll_reg_query.h"
#include "ll_reg_solve_common.h"

namespace ll {

#define RQ_THREADS 256  // queries per requery workgroup = work-list segment size
#define RQ_WAVES (RQ_THREADS / 64)
#ifndef RS_PREFETCH
#define RS_PREFETCH 2  // blocks in flight per thread in the fast-path sweep (0 = none, 1, 2)
#endif
#define RS_WAVES (RS_THREADS / 64)
#define HASH_EMPTY 0xffffffffffffffffull


// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// Per-iteration query kernels (corner and surface queries share every launch, blockIdx.z = kind).
//
//   ICP iterations 0 and 1 (and every iteration when neighbour reuse is disabled):
//       reg_transform_kernel -> reg_knn_kernel (all queries) -> reg_build_kernel (all queries)
//   ICP iteration >= 2:
//       reg_requery_kernel    : transform + displacement test of every query against its reuse record (ll_knn_core.h):
//                                 stable  -> nothing to do: same neighbours, same order, same residual block;
//                                 re-sort -> the same five neighbours re-evaluated at the new position, slot appended
//                                            to the chunk's re-sort list;
//                                 search  -> slot appended to the chunk's search list
//       reg_list_kernel       : full exact search of the search list + block constants of everything searched or re-sorted
// The two work lists are dense per scan and kind: every re-query workgroup reserves its share of the scan-and-kind's
// segment with one atomicAdd per list (work_cnt; ~94 workgroups per counter -- a single batch-wide counter cost 280 us
// of contention per launch), and the list kernel walks all segments as one dense index space (prefix sums of the 2 B
// counters in LDS, binary search per entry): a small grid of full wavefronts.  Round 1 kept one list segment per
// 256-query chunk and launched one workgroup per chunk: in the late iterations a chunk holds ~3 searches, the launch was
// 48 k workgroups with three busy lanes each, and its ~110 us floor (186 us average) was the cost of scheduling them.
// The order of the entries depends on the order of the atomics; every entry is processed independently, so results do not.

// K6t: pose transform of every query (pointAssociateToMap, fp64 math -> fp32 store like the reference).  A
// kernel of its own so that the double-precision sin/cos of the motion-deblur branch does not set the register
// footprint of the k-NN kernel.
__global__ __launch_bounds__(KB_THREADS) void reg_transform_kernel(RegDev rd, RegConst rc, int skip_kinds)
{
    const int b = blockIdx.y, kind = blockIdx.z;
    if ((skip_kinds >> kind) & 1) return;  // (the tile kernel transforms its own queries)
    const RegState *st = rd.state + b;
    if (st->done) return;
    const int n = kind ? rd.n_surf[b] : rd.n_corner[b];
    const int q = blockIdx.x * KB_THREADS + threadIdx.x;
    if (q >= n) return;
    const int slot = (kind ? rd.cap_c : 0) + q;
    float pw[3];
    transform_query(st, rc, load_feature(rd, b, kind, q), pw);
    // a13: a skipped feature is handed on as a non-finite query -- no neighbours, no block (PCR:232-238, 339-345)
    if (subsample_skip_feature(rc.subsample_seed, kind, st->icp_iters, q, n, rc.max_blocks)) pw[0] = pw[1] = pw[2] = NAN;
    rd.qw[(size_t)b * rd.cap + slot] = make_float4(pw[0], pw[1], pw[2], 0.f);
}

// K6a: one lane per query: exact 5-NN of the transformed point (fp32 only -> small register footprint, so
// occupancy hides the gather latency).  Output per query: positions (cell-sorted order) of the neighbours the
// block needs + "5 found" flag, and the reuse record for the next ICP iteration.
#ifndef KNN_WAVES_PER_EU
#define KNN_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(KB_THREADS) __attribute__((amdgpu_waves_per_eu(KNN_WAVES_PER_EU, 8)))
void reg_knn_kernel(RegDev rd, RegConst rc, Grid gc, Grid gs, int iter, int skip_kinds)
{
    const int b = blockIdx.y, kind = blockIdx.z;
    if ((skip_kinds >> kind) & 1) return;  // reg_knn_coop_kernel has them
    const RegState *st = rd.state + b;
    if (st->done) return;
    const int n = kind ? rd.n_surf[b] : rd.n_corner[b];
    const int q = blockIdx.x * KB_THREADS + threadIdx.x;
    if (q >= n) return;
    knn_one(rd, rc, gc, gs, b, (kind ? rd.cap_c : 0) + q, iter);
}

// K6a for small batches: one query per wavefront -- the corner queries (a few hundred per scan, a quarter of them searching
// rings of the sparse corner map: per lane the longest dependent chain of the launch, header of ll_knn_coop.h), and the
// surface queries too when a scan has few of them (voxel-filtered clouds against a sparse local map: the sequential mapping
// loop, where every search walks rings).  kinds: bit k set = kind k is searched here.
#define KC_THREADS 256
__global__ __launch_bounds__(KC_THREADS) void reg_knn_coop_kernel(RegDev rd, RegConst rc, Grid gc, Grid gs, int iter, int kinds)
{
    const int b = blockIdx.y, kind = blockIdx.z;
    if (!((kinds >> kind) & 1)) return;
    const RegState *st = rd.state + b;
    if (st->done) return;
    const int q = (int)((blockIdx.x * KC_THREADS + threadIdx.x) >> 6);
    if (q >= (kind ? rd.n_surf[b] : rd.n_corner[b])) return;  // (whole wavefronts)
    knn_one_coop(rd, rc, gc, gs, b, (kind ? rd.cap_c : 0) + q, iter);
}

// K6r: transform + reuse test (ICP iteration >= 1)

// One workgroup per chunk of RQ_PER x RQ_THREADS consecutive queries (round 3: four queries per thread -- their eight record loads
// are in flight together, and a workgroup pays its two barriers and two list reservations once per 1024 queries instead of once per
// 256; round 2: 73 us per B = 256 launch for 141 MB of records).  Unstable queries are appended to the dense per-(scan, kind) work
// lists: every thread with state 1 (re-sort) or 2 (search) gets a distinct position in its list; the workgroup reserves one
// contiguous range per list with one round of ballots per query slice, one barrier pair and two independent atomicAdds issued back
// to back.  (Within a list the entries of a workgroup are ordered by slice, then wavefront, then lane; nothing depends on the order.)
#define RQ_PER 4
__global__ __launch_bounds__(RQ_THREADS) void reg_requery_kernel(RegDev rd, RegConst rc, Grid gc, Grid gs, int iter)
{
    const int chunk = blockIdx.x, b = blockIdx.y, kind = blockIdx.z;
    const RegState *st = rd.state + b;
    if (st->done) return;
    const int n = kind ? rd.n_surf[b] : rd.n_corner[b];
    if (chunk * RQ_PER * RQ_THREADS >= n) return;
    const size_t sb = (size_t)b * rd.cap;
    const int koff = kind ? rd.cap_c : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_cnt[2][RQ_PER * RQ_WAVES];  // [list][slice * RQ_WAVES + wave]
    __shared__ int s_base[2];
    float4 ft[RQ_PER], rq[RQ_PER];
    int qq[RQ_PER];  // the query of this thread's u-th place (-1: beyond the scan's queries)
#pragma unroll
    for (int u = 0; u < RQ_PER; u++) {
        const int place = (chunk * RQ_PER + u) * RQ_THREADS + tid;
        qq[u] = place < n ? place : -1;
    }
#pragma unroll
    for (int u = 0; u < RQ_PER; u++) {
        const int qc = qq[u] >= 0 ? qq[u] : 0;
        ft[u] = load_feature(rd, b, kind, qc);
        rq[u] = rd.ref_q[sb + koff + qc];
    }
    int state[RQ_PER];  // 0 = stable or out of range, 1 = re-sorted, 2 = needs a search
    unsigned long long m1[RQ_PER], m2[RQ_PER];
#pragma unroll
    for (int u = 0; u < RQ_PER; u++) {
        const int q = qq[u];
        const int slot = koff + q;
        state[u] = 0;
        if (q >= 0) {
            float pw[3];
            transform_query(st, rc, ft[u], pw);
            KnnRef ref;
            ref.qx = rq[u].x;
            ref.qy = rq[u].y;
            ref.qz = rq[u].z;
            ref.m_strong = rq[u].w;
            const float delta = knn5_ref_delta(ref, pw[0], pw[1], pw[2]);  // NaN for a non-finite query -> search
            if (!(delta < ref.m_strong)) {  // else: same neighbours, same order: nn and the block are unchanged
                const float2 rs = rd.ref_s[sb + slot];
                ref.m_set = rs.y;
                // Both kinds of work are left to the list kernel: the five gathers and the stores of a re-sort in here kept
                // nearly every wavefront alive for three more dependent round trips (73 % of them hold at least one such lane)
                rd.qw[sb + slot] = make_float4(pw[0], pw[1], pw[2], 0.f);
                state[u] = (delta < ref.m_set) ? 1 : 2;
            }
        }
        m1[u] = __ballot(state[u] == 1);
        m2[u] = __ballot(state[u] == 2);
        if (lane == 0) {
            s_cnt[0][u * RQ_WAVES + wave] = __popcll(m1[u]);
            s_cnt[1][u * RQ_WAVES + wave] = __popcll(m2[u]);
        }
    }
    __syncthreads();
    int off1[RQ_PER], off2[RQ_PER], tot1 = 0, tot2 = 0;
#pragma unroll
    for (int u = 0; u < RQ_PER; u++) {
        off1[u] = off2[u] = 0;
        for (int w = 0; w < RQ_WAVES; w++) {
            const int c1 = s_cnt[0][u * RQ_WAVES + w], c2 = s_cnt[1][u * RQ_WAVES + w];
            if (w < wave) off1[u] += c1, off2[u] += c2;
            tot1 += c1, tot2 += c2;
        }
    }
    // (offsets of slice u: everything in the slices before it, then the earlier wavefronts of its own)
    int pre1 = 0, pre2 = 0;
#pragma unroll
    for (int u = 0; u < RQ_PER; u++) {
        int s1 = 0, s2 = 0;
        for (int w = 0; w < RQ_WAVES; w++) s1 += s_cnt[0][u * RQ_WAVES + w], s2 += s_cnt[1][u * RQ_WAVES + w];
        off1[u] += pre1;
        off2[u] += pre2;
        pre1 += s1;
        pre2 += s2;
    }
    int *cnt = rd.work_cnt + ((size_t)b * 2 + kind) * 2;  // [0] search, [1] re-sort
    if (tid == 0) {
        const int b1 = tot1 > 0 ? atomicAdd(cnt + 1, tot1) : 0;
        const int b2 = tot2 > 0 ? atomicAdd(cnt + 0, tot2) : 0;
        s_base[0] = b1;
        s_base[1] = b2;
    }
    __syncthreads();
    const size_t seg = sb + koff;  // the scan-and-kind's own segment of the work arrays
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < RQ_PER; u++) {
        const int slot = koff + qq[u];
        if (state[u] == 1) rd.work_build[seg + s_base[0] + off1[u] + __popcll(m1[u] & below)] = (int)sb + slot;
        if (state[u] == 2) rd.work_search[seg + s_base[1] + off2[u] + __popcll(m2[u] & below)] = (int)sb + slot;
    }
}

__global__ __launch_bounds__(KB_THREADS) void reg_build_kernel(RegDev rd, RegConst rc, Grid gc, Grid gs, int skip_kinds)
{
    const int b = blockIdx.y, kind = blockIdx.z;
    if ((skip_kinds >> kind) & 1) return;  // (the tile kernel builds the blocks of its own slots)
    const RegState *st = rd.state + b;
    if (st->done) return;
    const int n = kind ? rd.n_surf[b] : rd.n_corner[b];
    const int q = blockIdx.x * KB_THREADS + threadIdx.x;
    if (q >= n) return;
    build_one(rd, rc, gc, gs, b, (kind ? rd.cap_c : 0) + q);
}

// ICP iteration >= 2: exact search of the dense search list followed at once by the block constants of the same slot
// (the lane still holds the neighbours), then the block constants of the re-sorted slots.  Grid-stride over the lists.
#define RL_THREADS 128
#define RL_BLOCKS 2048  // x 128 threads; 4096 (every wavefront slot at 64 VGPRs) measured the same, the lists are bound by dependent misses
#define RL_MAX_SEG 2048  // scan-and-kind segments of one offsets table (max_scans <= 1024); larger batches run in slices
#define RL_LOCAL_SEG 64  // up to this many segments (32 scans) the list kernel builds the offsets itself
// exclusive prefix sums of the per-segment list lengths (segment = scan * 2 + kind) -> work_off[list][0 .. n_seg]; one workgroup
__global__ __launch_bounds__(1024) void reg_list_offsets_kernel(RegDev rd, int seg0, int n_seg)
{
    __shared__ int s_wave[3][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sg0 = 2 * tid, sg1 = 2 * tid + 1;  // two segments per thread (n_seg <= 2048): a scan's corner and surface segment
    for (int w = 0; w < 3; w++) {
        // (w = 2: the search list again with every surface segment -- odd: segment = scan * 2 + kind -- counted as empty)
        const int c0 = sg0 < n_seg ? rd.work_cnt[(size_t)(seg0 + sg0) * 2 + (w & 1)] : 0;
        const int c1 = (sg1 < n_seg && w < 2) ? rd.work_cnt[(size_t)(seg0 + sg1) * 2 + w] : 0;
        int incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) s_wave[w][wave] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < wave; k++) base += s_wave[w][k];
        const int excl = base + incl - (c0 + c1);
        int *off_w = rd.work_off + (size_t)w * (RL_MAX_SEG + 1);
        if (sg0 < n_seg) off_w[sg0] = excl;
        if (sg1 < n_seg) off_w[sg1] = excl + c0;
        if (tid == 1023) off_w[n_seg] = base + incl;  // its segments lie beyond n_seg or are the last ones: the grand total
        __syncthreads();
    }
}

// LOCAL: the offsets tables are built in LDS by every workgroup (small batches); a template constant so that the large-batch
// form keeps plain global loads in its binary searches
template <bool LOCAL>
__global__ __launch_bounds__(RL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void reg_list_kernel(RegDev rd, RegConst rc, Grid gc, Grid gs, int iter, int seg0, int n_seg)
{
    // The offsets table (<= 16 KB) is searched where it lies: it stays in L1 / L2, and a copy in LDS would cap the
    // occupancy of this latency-bound kernel (16 KB per 128-thread workgroup: 36 -> 99 us per late iteration at B = 256).
    const int tid = threadIdx.x;
    const int stride = gridDim.x * RL_THREADS;
    // Corner searches first, one per WAVEFRONT while there are few of them (round 3): a late iteration searches a handful of
    // corner queries per scan, each a chain of 100+ dependent loads for a single lane -- the floor of this launch (~100 us at
    // B = 256 for ~1.5 k of them beside 75 k surface searches of ~15 round trips each; 49 us for a single scan).
    // With few searches altogether (a single scan, a small batch, voxel-filtered clouds) every search goes that way.
    // Small batches (<= RL_LOCAL_SEG segments) skip the offsets kernel: every workgroup sums the few counters itself
    // (one launch and one kernel boundary less per ICP iteration: ~6 us of a single scan's ~40 per iteration).
    __shared__ int s_cnt[2][LOCAL ? RL_LOCAL_SEG : 1];
    __shared__ int s_off[3][LOCAL ? RL_LOCAL_SEG + 1 : 1];
    if (LOCAL) {
        if (tid < n_seg) {
            s_cnt[0][tid] = rd.work_cnt[(size_t)(seg0 + tid) * 2 + 0];
            s_cnt[1][tid] = rd.work_cnt[(size_t)(seg0 + tid) * 2 + 1];
        }
        __syncthreads();
        if (tid < 3) {  // (as reg_list_offsets_kernel: searches, re-sorts, the searches of the corner segments alone)
            int acc = 0;
            for (int sg = 0; sg < n_seg; sg++) {
                s_off[tid][sg] = acc;
                acc += tid == 2 ? ((sg & 1) ? 0 : s_cnt[0][sg]) : s_cnt[tid][sg];
            }
            s_off[tid][n_seg] = acc;
        }
        __syncthreads();
    }
    const int *off_s = LOCAL ? s_off[0] : rd.work_off;
    const int *off_c = LOCAL ? s_off[2] : rd.work_off + (size_t)2 * (RL_MAX_SEG + 1);
    const bool coop_all = rc.knn_coop && off_s[n_seg] <= LL_KNN_COOP_MAX_QUERIES;
    const bool coop = rc.knn_coop && off_c[n_seg] <= LL_KNN_COOP_MAX_QUERIES;  // the corner ones at least
    if (coop_all || coop) {
        // (handed out from the LAST wavefront of the grid backwards: the per-lane lists below fill the grid from the front, so
        // with short lists a wavefront has either a cooperative search or per-lane entries and the two chains overlap)
        const int *off_w = coop_all ? off_s : off_c;
        const int total_w = off_w[n_seg];
        const int n_waves = stride >> 6;
        for (int t = n_waves - 1 - (int)((blockIdx.x * RL_THREADS + tid) >> 6); t < total_w; t += n_waves) {
            int lo = 0, hi = n_seg;  // largest segment with off_w[segment] <= t (in off_c the empty surface segments tie with their successor)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off_w[mid] <= t) lo = mid; else hi = mid;
            }
            const int sgg = seg0 + lo, b = sgg >> 1, kind = sgg & 1;
            const int e = rd.work_search[(size_t)b * rd.cap + (kind ? rd.cap_c : 0) + (t - off_w[lo])];
            const int slot = e - b * rd.cap;
            knn_one_coop(rd, rc, gc, gs, b, slot, iter);
            if ((tid & 63) == 0) build_one(rd, rc, gc, gs, b, slot);
        }
    }
    // (one index space over both lists, so that a lane never runs a re-sort after a search, brought the floor from 113 back
    // to 99 us but cost 25 % at the long early lists -- profiles/r02 runs U / V -- and was dropped)
    for (int w = coop_all ? 1 : 0; w < 2; w++) {
        const int *off = LOCAL ? s_off[w] : rd.work_off + (size_t)w * (RL_MAX_SEG + 1);
        const int total = off[n_seg];
        const int *list = w == 0 ? rd.work_search : rd.work_build;
        for (int t = blockIdx.x * RL_THREADS + tid; t < total; t += stride) {
            int lo = 0, hi = n_seg;  // largest segment with off[segment] <= t
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] <= t) lo = mid; else hi = mid;
            }
            const int sgg = seg0 + lo, b = sgg >> 1, kind = sgg & 1;
            if (w == 0 && coop && kind == 0) continue;  // done above
            const int e = list[(size_t)b * rd.cap + (kind ? rd.cap_c : 0) + (t - off[lo])];
            const int slot = e - b * rd.cap;
            if (w == 0) knn_one(rd, rc, gc, gs, b, slot, iter);
            else resort_one(rd, rc, gc, gs, b, slot, iter);
            build_one(rd, rc, gc, gs, b, slot);
        }
    }
}

// Evaluation context as plain locals (R_inc / t_inc for the plain blocks, axis-angle for the motion-deblur ones);
// DEBLUR is a template constant, so the unused half disappears from each kernel instantiation.
#define LL_CTX_DECL(x)                                   \
    double R_[9], t_[3];                                  \
    MbRot mb_;                                            \
    {                                                     \
        const double q_[4] = {(x)[0], (x)[1], (x)[2], (x)[3]}; \
        quat_to_mat(q_, R_);                              \
        t_[0] = (x)[4];                                   \
        t_[1] = (x)[5];                                   \
        t_[2] = (x)[6];                                   \
        if (DEBLUR) mb_prepare(q_, mb_);                  \
    }
#define LL_CTX_ACCUM(kind, ff, a, v, huber_a, acc)                                                        \
    do {                                                                                                  \
        const double f_[3] = {(double)(ff).x, (double)(ff).y, (double)(ff).z};                            \
        if (DEBLUR)                                                                                       \
            block_accumulate_mb((kind), mb_, t_, (double)(ff).w, f_, (a), (v), (huber_a), (acc)); /* ICP:81-233 */ \
        else                                                                                              \
            block_accumulate((kind), R_, t_, f_, (a), (v), (huber_a), (acc)); /* ICP:238-380 */           \
    } while (0)
#define LL_CTX_L1(out, kind, ff, a, v, huber_a, q_last)                                                   \
    do {                                                                                                  \
        const double f_[3] = {(double)(ff).x, (double)(ff).y, (double)(ff).z};                            \
        if (DEBLUR)                                                                                       \
            (out) = block_l1_mb((kind), mb_, t_, (double)(ff).w, f_, (a), (v), (huber_a), (q_last));      \
        else                                                                                              \
            (out) = block_l1((kind), R_, t_, f_, (a), (v), (huber_a), (q_last));                          \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
#ifdef LL_SOLVE_TIMING
#define LL_T0(var) long long var = clock64()
#define LL_TACC(slot, var)                                   \
    do {                                                     \
        if (threadIdx.x == 0) sh.tcyc[slot] += clock64() - var; \
    } while (0)
#else
#define LL_T0(var)
#define LL_TACC(slot, var)
#endif

struct SolveShared {
    long long tcyc[16];
    LmCtl ctl;
    double red[RS_WAVES][LL_NACC];
    double sum[LL_NACC];
    int need;
    int hist[256];
    int sel_bin, sel_cnt, n_cand;
    int isum[RS_WAVES];
    unsigned long long lsum[RS_WAVES];
    unsigned long long sel_prefix;
    int sel_rank;
    int n_active, n_corner_avail, n_surf_avail, n_unique;
    int l1_valid;  // compact path: blk_l1 holds the L1 values at the prerun result (written by its last evaluation)
    int pt_T, pt_Tl, pt_kc, pt_priv;  // plane-table path: distinct triples, table entries in LDS, record rounds cached in LDS, private entries
    double fit[10];                   // arguments of a line search's three-sample fit, from the controller lane to its wavefront
    int pt_nl;                        // ... line blocks of this workgroup's first rounds kept at the top of s_raw (solver_eval3)
    double thr;
    int grp_g, grp_G, grp_seq, grp_abort;  // grouped solver: this workgroup's rank in its scan's group, the group size, barriers passed
    int xch_seq, xch_epoch;                // ... exchanges of partial sums made in this launch, and the launch's number in its registration (granule tags)
    unsigned int xch[LL_GRP][2 * LL_NACC]; // ... the members' partial sums of one exchange, as 32-bit halves
};

__device__ __forceinline__ int slot_of(int j, int nC, int cap_c) { return j < nC ? j : cap_c + (j - nC); }

// ---- grouped solver (small batches) ------------------------------------------------------------------------------------
// With one workgroup per scan a batch of B <= 16 scans keeps B of 256 CUs busy and every cost evaluation re-streams the 85 %
// of the scan's block records that do not fit one CU's LDS.  For such batches a scan is given to a GROUP of LL_GRP
// workgroups: every workgroup runs the whole control flow redundantly on the same numbers (census, LM controller, set
// de-duplication, rank select, prune: all deterministic), but a cost evaluation visits only the workgroup's 1/G share of the
// blocks -- which then fits its LDS record cache, so nothing is streamed after the first evaluation -- and the 28 partial sums
// are exchanged through global memory and added in rank order by every member (identical totals, identical decisions).
// Membership is by ticket (atomic counter, zeroed per launch): the G workgroups of a group are by construction running, so the
// flag barrier below cannot wait for a workgroup that has no CU; a partial group waits only for workgroups that start as CUs
// free up.  Spins are bounded: a barrier that does not complete aborts the scan's registration (RegState::aborted: the
// scan comes back rejected with its pose restored and ll_reg_collect reports the failure) instead of hanging the device.
#define LL_GRP_SPIN_LIMIT (1 << 22)
// FULL: every thread's plain global stores before the barrier (the L1 values) are visible to every member after it, and no
// member keeps stale lines -- an agent-scope release / acquire fence in every wavefront.  Otherwise only data moved with
// agent-scope atomics by the calling threads themselves is exchanged (the partial sums) and the barrier is just the counter.
template <bool FULL>
__device__ __forceinline__ void group_barrier(const RegDev &rd, int b, SolveShared &sh)
{
    if (FULL) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && !sh.grp_abort) {
        const int target = (++sh.grp_seq) * sh.grp_G;
        int *bar = rd.grp_ctl + 1 + 2 * b, *abt = bar + 1;  // the scan's arrival counter and its abort word
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            // a member that gives up tells the others: they must not go on with its stale partial sums, nor wait out their own
            // limit for arrivals that will never come
            if (++spins > LL_GRP_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(abt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                sh.grp_abort = 1;
                __hip_atomic_store(abt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (!sh.grp_abort && __hip_atomic_load(abt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) sh.grp_abort = 1;
    }
    __syncthreads();  // (the partial sums are then read with agent-scope atomic loads, which no cache level may satisfy stale)
    if (FULL) __threadfence();
}

// sh.sum (this workgroup's share) -> sh.sum (the scan's totals), the same value in every member.
// FULL (the one evaluation per launch that also publishes the L1 values): partial sums through rd.grp_part behind the counter
// barrier with its fences.  Otherwise (every other evaluation, ~8 per launch) a fence-free exchange of self-validating
// GRANULES (MI355X_MICROARCH.md, hand-off price list: 8-byte {data, tag} written by one agent-scope store, polled with
// agent-scope loads -- about one memory round trip, against four for store / release / counter / poll / load): every double
// travels as two granules {32-bit half, tag}, tag = launch epoch << 12 | exchange number, so a reader that sees the tag has
// the data and nothing needs ordering.  Slots alternate by exchange parity: a member that posts exchange s + 2 has read all
// of s + 1, which every member posted only after reading all of s.  rd.grp_xch is zeroed per registration and the epoch is the
// ICP iteration, so no tag repeats while a stale granule could still be seen.  Bounded polling like the barrier: a member
// that gives up (or runs out of exchange numbers) raises the scan's abort word, the others see it.
template <bool FULL>
__device__ __forceinline__ void group_reduce(const RegDev &rd, int b, SolveShared &sh)
{
    const int tid = threadIdx.x, G = sh.grp_G;
    if (FULL) {
        double *part = rd.grp_part + ((size_t)b * 2 + (sh.grp_seq & 1)) * LL_GRP * LL_NACC;
        if (tid < LL_NACC) __hip_atomic_store(part + sh.grp_g * LL_NACC + tid, sh.sum[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        group_barrier<true>(rd, b, sh);
        if (tid < LL_NACC) {
            double t = 0.0;
            for (int k = 0; k < G; k++) t += __hip_atomic_load(part + k * LL_NACC + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.sum[tid] = t;
        }
        __syncthreads();
        return;
    }
    constexpr int NG = 2 * LL_NACC;  // granules per member
    static_assert(LL_GRP * NG <= RS_THREADS, "one polling thread per granule");
    if (tid == 0) sh.xch_seq++;
    __syncthreads();  // (sh.sum is complete; grp_abort as the last exchange left it)
    const int seq = sh.xch_seq;
    if (!sh.grp_abort && seq < 4096) {  // uniform
        const unsigned int tag = ((unsigned int)sh.xch_epoch << 12) | (unsigned int)seq;
        unsigned long long *slot = rd.grp_xch + ((size_t)b * 2 + (seq & 1)) * LL_GRP * NG;
        if (tid < NG) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sh.sum[tid >> 1]);
            const unsigned int half = (tid & 1) ? (unsigned int)(bits >> 32) : (unsigned int)bits;
            __hip_atomic_store(slot + sh.grp_g * NG + tid, ((unsigned long long)tag << 32) | (unsigned long long)half, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < G * NG) {
            const int *abt = rd.grp_ctl + 2 + 2 * b;
            unsigned long long v = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ((unsigned int)(v >> 32) != tag) {
                if (++spins > LL_GRP_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(abt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    sh.grp_abort = 1;  // (any thread may raise it; read again behind the barrier below)
                    __hip_atomic_store(const_cast<int *>(abt), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sh.xch[tid / NG][tid % NG] = (unsigned int)v;
        }
    } else if (tid == 0 && !sh.grp_abort) {  // out of exchange numbers: fail loudly rather than reuse a tag
        sh.grp_abort = 1;
        __hip_atomic_store(rd.grp_ctl + 2 + 2 * b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < LL_NACC && !sh.grp_abort) {  // (after an abort the sums stay the member's own: nothing of this launch is kept)
        double t = 0.0;
        for (int k = 0; k < G; k++)
            t += __longlong_as_double((long long)(((unsigned long long)sh.xch[k][2 * tid + 1] << 32) | (unsigned long long)sh.xch[k][2 * tid]));
        sh.sum[tid] = t;
    }
    __syncthreads();
}

// three counts (each < 2^20) in one reduction: a | b << 20 | c << 40
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, SolveShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)__shfl_down((long long)v, off);
    if (lane == 0) sh.lsum[wave] = v;
    __syncthreads();
    unsigned long long s = 0;
    for (int w = 0; w < RS_WAVES; w++) s += sh.lsum[w];
    __syncthreads();
    return s;
}

__device__ int block_sum_int(int v, SolveShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) sh.isum[wave] = v;
    __syncthreads();
    int s = 0;
    for (int w = 0; w < RS_WAVES; w++) s += sh.isum[w];
    __syncthreads();
    return s;
}

__device__ __forceinline__ unsigned long long hash64(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}



// capacities shared by the solver paths (the fast paths are described further down)
#define FAST_MAXK (FAST_MAX_BLOCKS / RS_THREADS)
#define HT_SIZE 16384
#define HT_PART 6144  // keys per de-duplication round (load factor <= 0.375)
// set de-duplication, common case: bitmap + contested-bit set + exact table of the contested keys (all inside s_table)
#define DD_BM_WORDS 16384  // 512 Kbit
#define DD_CB_LOG2 11
#define DD_CB_SIZE (1 << DD_CB_LOG2)
#define DD_EX_SIZE 4096
#define DD_MAX_COLL 900    // contested keys beyond this (heavily duplicated input): hash every key instead
// register-tile de-duplication (inlier_threshold_regs): 2-bit slot states, 64 KB + 32 KB, and a list of twice-contested keys
#define DD2_WORDS 16384
#define DD2_SLOTS (DD2_WORDS * 16)
#define DD2B_WORDS 8192
#define DD2B_SLOTS (DD2B_WORDS * 16)
#define DD2_LIST 2048  // twice-contested keys compared exactly; more (heavily duplicated input): hash every key instead
#define SEL_BINS 4096  // value-range bins of the rank select (must be a multiple of RS_THREADS)
#define SEL_CAND 1024  // keys of the selected bin ranked exactly; more -> radix-select fallback


// std::set de-duplication + rank select of the loss-corrected L1 values held in registers (l1r[k] = value of the
// thread's k-th block, < 0 for "no active block"): sets sh.thr = max(inliner_dis, element at int(ratio * n_distinct))
// (PCR:153-161, 484-485).  `total` only sizes the rounds of the heavily-duplicated fallback.
template <int NK>
__device__ __forceinline__ void inlier_threshold_regs(const double (&l1r)[NK], int total, unsigned long long *s_table, SolveShared &sh,
                                                      const RegConst &rc)
{
    static_assert(NK <= 64 && NK <= FAST_MAXK, "register tile of the fast paths (one bit per entry in the 64-bit masks)");
    const int tid = threadIdx.x;
    LL_T0(t_dd);
    // ---- std::set semantics (PCR:155-160): which values are distinct, and how many ---------------------------------
    // Exact duplicates among the residuals are rare, so the common case is made cheap: every key sets one bit of a
    // 512 Kbit LDS bitmap (atomicOr); only keys whose bit was already set -- true duplicates or hash collisions, a few
    // hundred of ~17 k -- and the keys that share a bit with them go through an exact compare-and-swap table.  Heavily
    // duplicated inputs (more than DD_MAX_COLL such keys) fall back to hashing every key, HT_PART keys per round.
    unsigned long long first_mask = 0;  // bit k: block k of this thread is the first occurrence of its L1 value
    {
        // Common case (round 2 form).  Three short, branch-light passes over the thread's keys instead of a
        // compare-and-swap probe per key (round 1: ~290 instructions per key once the compiler had unrolled the probe loop):
        //   A  every key marks a 2-bit slot state {bit 0: a key landed here, bit 1: a second key landed here} in a
        //      256 K-slot table (atomicOr): a key whose slot never gets bit 1 is distinct from every other key;
        //   B  the keys of contested slots (hash collisions and true duplicates, ~1100 of ~17 k) repeat that in a second
        //      table under an independent hash: distinct keys that merely collided in A almost surely separate here;
        //   C  what is contested twice -- true duplicates plus a stray pair -- is appended to a short list and compared
        //      exactly, each entry against the entries before it.
        // Slot hashes are two 32-bit multiplies.  Heavily duplicated inputs (list overflow) take the table path below.
        int my = 0;
        unsigned int *bmA = (unsigned int *)s_table;                       // [DD2_WORDS]  16 slots x 2 bits per word, 64 KB
        unsigned int *bmB = bmA + DD2_WORDS;                               // [DD2B_WORDS] second table, 32 KB
        unsigned long long *dl = (unsigned long long *)(bmB + DD2B_WORDS); // [DD2_LIST] twice-contested keys
        {
            uint4 *z = (uint4 *)s_table;
            for (int e = tid; e < (DD2_WORDS + DD2B_WORDS) / 4; e += RS_THREADS) z[e] = make_uint4(0u, 0u, 0u, 0u);
            if (tid == 0) sh.n_cand = 0;
        }
        __syncthreads();
        const int kt = (total + RS_THREADS - 1) / RS_THREADS;  // rounds that can hold a block (uniform)
        // slot hashes are recomputed in each pass (two 32-bit multiplies) rather than kept: with l1r[48] live, another 48
        // registers per thread made the compiler spill half of l1r to scratch, and every later pass paid for it
        auto slot_a = [](unsigned long long key) -> unsigned int {
            unsigned int h = (unsigned int)key * 0x9E3779B1u;
            h ^= h >> 15;
            h += (unsigned int)(key >> 32) * 0x85EBCA77u;
            h ^= h >> 13;
            return h & (DD2_SLOTS - 1);
        };
        auto slot_b = [](unsigned long long key) -> unsigned int {
            unsigned int h2 = ((unsigned int)(key >> 32) * 0xC2B2AE3Du) ^ ((unsigned int)key * 0x27D4EB2Fu);
            return (h2 ^ (h2 >> 16)) & (DD2B_SLOTS - 1);
        };
        unsigned long long valid_mask = 0, cont_mask = 0;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            if (k >= kt) continue;  // not `break`: an early exit keeps the loop rolled and the register tile in scratch
            const double l1 = l1r[k];
            const bool valid = l1 >= 0.0;  // inactive slot or NaN (NaN never enters the set)
            const unsigned int slot = slot_a((unsigned long long)__double_as_longlong(l1));
            const unsigned int bit0 = valid ? (1u << ((slot & 15u) * 2u)) : 0u;  // 0: a harmless no-op for padding lanes
            const unsigned int old = atomicOr(&bmA[slot >> 4], bit0);
            atomicOr(&bmA[slot >> 4], (old & bit0) << 1);
            if (valid) valid_mask |= 1ull << k;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NK; k++) {
            if (k >= kt) continue;  // not `break`: an early exit keeps the loop rolled and the register tile in scratch
            const unsigned long long key = (unsigned long long)__double_as_longlong(l1r[k]);
            const unsigned int slot = slot_a(key);
            const bool contested = ((valid_mask >> k) & 1ull) && ((bmA[slot >> 4] >> ((slot & 15u) * 2u)) & 2u);
            const unsigned int h2 = slot_b(key);
            const unsigned int bit0 = contested ? (1u << ((h2 & 15u) * 2u)) : 0u;
            const unsigned int old = atomicOr(&bmB[h2 >> 4], bit0);
            atomicOr(&bmB[h2 >> 4], (old & bit0) << 1);
            if (contested) cont_mask |= 1ull << k;
        }
        __syncthreads();
        unsigned long long twice_mask = 0;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            if (k >= kt) continue;  // not `break`: an early exit keeps the loop rolled and the register tile in scratch
            const unsigned int h2 = slot_b((unsigned long long)__double_as_longlong(l1r[k]));
            if (((cont_mask >> k) & 1ull) && ((bmB[h2 >> 4] >> ((h2 & 15u) * 2u)) & 2u)) twice_mask |= 1ull << k;
        }
        // every valid key that is not contested twice is distinct; the rest go through the exact list
        my = __popcll(valid_mask & ~twice_mask);
        first_mask = valid_mask & ~twice_mask;
        int my_pos[2] = {-1, -1};  // list positions of this thread's twice-contested keys (more than two: overflow)
        int n_twice = __popcll(twice_mask);
        if (twice_mask) {
            int got = 0;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                if (!((twice_mask >> k) & 1ull)) continue;
                const int pos = atomicAdd(&sh.n_cand, 1);
                if (pos < DD2_LIST) dl[pos] = (unsigned long long)__double_as_longlong(l1r[k]);
                if (got < 2) my_pos[got] = pos;
                got++;
            }
        }
        const int over = block_sum_int(n_twice > 2 ? 1 : 0, sh);  // its barriers also publish the list
        const int n_list = sh.n_cand;
        if (n_list <= DD2_LIST && over == 0) {
            if (twice_mask) {
                int got = 0;
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    if (!((twice_mask >> k) & 1ull)) continue;
                    const int pos = my_pos[got < 2 ? got : 1];
                    got++;
                    const unsigned long long key = (unsigned long long)__double_as_longlong(l1r[k]);
                    bool dup = false;
                    for (int j = 0; j < pos; j++) dup |= (dl[j] == key);
                    if (!dup) {
                        first_mask |= 1ull << k;
                        my++;
                    }
                }
            }
        } else {
            first_mask = 0;
            my = 0;
            const int rounds = (total + HT_PART - 1) / HT_PART;
            for (int rnd = 0; rnd < rounds; rnd++) {
                __syncthreads();
                for (int e = tid; e < HT_SIZE; e += RS_THREADS) s_table[e] = HASH_EMPTY;
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const double l1 = l1r[k];
                    if (!(l1 >= 0.0)) continue;
                    const unsigned long long key = (unsigned long long)__double_as_longlong(l1);
                    const unsigned long long hk = hash64(key);
                    if ((int)((hk >> 40) % (unsigned long long)rounds) != rnd) continue;
                    unsigned int h = (unsigned int)hk & (HT_SIZE - 1);
                    for (;;) {
                        const unsigned long long old = atomicCAS(&s_table[h], HASH_EMPTY, key);
                        if (old == HASH_EMPTY) {
                            first_mask |= (1ull << k);
                            my++;
                            break;
                        }
                        if (old == key) break;
                        h = (h + 1u) & (HT_SIZE - 1);
                    }
                }
            }
        }
        const int nu = block_sum_int(my, sh);
        if (tid == 0) {
            sh.n_unique = nu;
            sh.sel_prefix = 0ull;
            int target = (int)(rc.inlier_ratio * (double)nu);  // PCR:160
            if (target > nu - 1) target = nu - 1;
            sh.sel_rank = target;
        }
        __syncthreads();
    }
    LL_TACC(3, t_dd);
    LL_T0(t_sel);
    if (sh.n_unique > 0) {
        // Rank select of the distinct values.  One histogram pass over SEL_BINS value-range bins (a monotone map, so
        // every key in a lower bin is smaller) finds the bin that holds the wanted rank; its few keys are ranked
        // exactly against each other.  An 8-bit radix select over the same keys is the fallback when that bin is
        // crowded (heavily clustered values).
        int *bins = (int *)s_table;                                      // [SEL_BINS]
        unsigned long long *cand = s_table + SEL_BINS / 2;               // [SEL_CAND] keys of the selected bin
        double kmin = INFINITY, kmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < NK; k++)
            if (first_mask & (1ull << k)) {
                kmin = fmin(kmin, l1r[k]);
                kmax = fmax(kmax, l1r[k]);
            }
        for (int off = 32; off > 0; off >>= 1) {
            kmin = fmin(kmin, __shfl_down(kmin, off));
            kmax = fmax(kmax, __shfl_down(kmax, off));
        }
        if ((tid & 63) == 0) {
            sh.red[tid >> 6][0] = kmin;
            sh.red[tid >> 6][1] = kmax;
        }
        for (int e = tid; e < SEL_BINS; e += RS_THREADS) bins[e] = 0;
        __syncthreads();
        double lo = sh.red[0][0], hi = sh.red[0][1];
        for (int w = 1; w < RS_WAVES; w++) {
            lo = fmin(lo, sh.red[w][0]);
            hi = fmax(hi, sh.red[w][1]);
        }
        const double scale = (hi > lo) ? (double)(SEL_BINS - 1) / (hi - lo) : 0.0;
#pragma unroll
        for (int k = 0; k < NK; k++)
            if (first_mask & (1ull << k)) {
                int bi = (int)((l1r[k] - lo) * scale);
                bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                atomicAdd(&bins[bi], 1);
            }
        __syncthreads();
        // locate the bin of the wanted rank: every thread sums its SEL_BINS/RS_THREADS consecutive bins, a workgroup
        // prefix sum (wave scan + the eight wave totals) gives each thread the count below its first bin, and the one
        // thread whose range holds the rank walks its own few bins
        {
            const int per = SEL_BINS / RS_THREADS;
            const int lane = tid & 63, wave = tid >> 6;
            int part = 0;
            for (int e = 0; e < per; e++) part += bins[tid * per + e];
            int incl = part;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if (lane >= off) incl += y;
            }
            if (lane == 63) sh.isum[wave] = incl;
            if (tid == 0) sh.n_cand = 0;
            __syncthreads();
            int below = incl - part;
            for (int w = 0; w < wave; w++) below += sh.isum[w];
            const int rank = sh.sel_rank;
            __syncthreads();  // everyone has read sel_rank / isum before the owner overwrites sel_rank
            const bool last_thread = tid == RS_THREADS - 1;
            if ((rank >= below && rank < below + part) || (last_thread && rank >= below + part)) {
                int cum = below, bi = tid * per;
                for (; bi < tid * per + per - 1; bi++) {
                    if (cum + bins[bi] > rank) break;
                    cum += bins[bi];
                }
                sh.sel_bin = bi;
                sh.sel_rank = rank - cum;  // rank inside the bin
                sh.sel_cnt = bins[bi];
            }
            __syncthreads();
        }
        if (sh.sel_cnt <= SEL_CAND) {
            const int sel_bin = sh.sel_bin;
#pragma unroll
            for (int k = 0; k < NK; k++)
                if (first_mask & (1ull << k)) {
                    int bi = (int)((l1r[k] - lo) * scale);
                    bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                    if (bi == sel_bin) cand[atomicAdd(&sh.n_cand, 1)] = (unsigned long long)__double_as_longlong(l1r[k]);
                }
            __syncthreads();
            const int m = sh.n_cand;  // == sel_cnt
            for (int i = tid; i < m; i += RS_THREADS) {
                const unsigned long long ki = cand[i];
                int rk = 0;
                for (int j = 0; j < m; j++) rk += (cand[j] < ki) ? 1 : 0;  // keys are distinct
                if (rk == sh.sel_rank) sh.sel_prefix = ki;
            }
            __syncthreads();
        } else {
            // crowded bin: MSB-first radix select (8 bits per pass) restricted to the keys of that bin
            const int sel_bin = sh.sel_bin;
            if (tid == 0) sh.sel_prefix = 0ull;
            __syncthreads();
            for (int pass = 0; pass < 8; pass++) {
                const int shift = 56 - 8 * pass;
                for (int e = tid; e < 256; e += RS_THREADS) sh.hist[e] = 0;
                __syncthreads();
                const unsigned long long prefix = sh.sel_prefix;
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    if (!(first_mask & (1ull << k))) continue;
                    int bi = (int)((l1r[k] - lo) * scale);
                    bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                    if (bi != sel_bin) continue;
                    const unsigned long long key = (unsigned long long)__double_as_longlong(l1r[k]);
                    if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&sh.hist[(int)((key >> shift) & 255ull)], 1);
                }
                __syncthreads();
                if (tid == 0) {
                    int rank = sh.sel_rank, d = 0, cum = 0;
                    for (d = 0; d < 256; d++) {
                        if (cum + sh.hist[d] > rank) break;
                        cum += sh.hist[d];
                    }
                    if (d > 255) d = 255;
                    sh.sel_rank = rank - cum;
                    sh.sel_prefix = (prefix << 8) | (unsigned long long)d;
                }
                __syncthreads();
            }
        }
        if (tid == 0) sh.thr = fmax(rc.inliner_dis, __longlong_as_double((long long)sh.sel_prefix));  // PCR:485
    } else {
        if (tid == 0) sh.thr = rc.inliner_dis;  // empty set: defined deviation (PCR:160 would dereference end())
    }
    __syncthreads();
    LL_TACC(4, t_sel);
}

// ---------------------------------------------------------------------------------------------------------
// Fast path (<= FAST_MAX_BLOCKS residual blocks per scan, i.e. every BASELINE Mid-40 configuration): block
// flags live in LDS, the per-block L1 values of the inlier test live in registers, the std::set
// de-duplication runs in an LDS hash table and the rank select reads registers -- the only HBM traffic left is
// one coalesced sweep over the block constants per cost evaluation, software-pipelined one block ahead.
struct BlkRegs {
    float4 f;
    double a0, a1, a2, v0, v1, v2;
};

__device__ __forceinline__ void load_blk(const RegDev &rd, size_t sb, const double *av, int slot, BlkRegs &r)
{
    r.f = gload_f4(rd.blk_f + sb + slot);
    // surface slots hold plane blocks: a' is folded into the scalar a0 = n'.a' (ll_reg_core.h block_plane)
    av_load(av, rd.cap, slot, slot < rd.cap_c, r.a0, r.a1, r.a2, r.v0, r.v1, r.v2);
}





// ---------------------------------------------------------------------------------------------------------
// Round-3 compact path (scan_is_compact(), default): PLANE TABLE.  A scan's ~17 k plane blocks are built from only 2.4 - 4.6 k
// distinct ordered (nn0, nn2, nn4) neighbour triples (the queries of a wall patch share their nearest map points), and every
// block with the same triple carries bit-identical {n', c} -- 32 of the 48 bytes a per-block record form re-streams on each of ~7 cost
// evaluations.  Here the solver workgroup de-duplicates the triples itself at the start of every launch:
//   1. census: block flags -> a 64-bit activity mask per thread (bit k <-> block tid + k * 512 in the planes / padding / lines
//      order; no flag array in LDS);
//   2. every active plane block's triple (rd.nn) goes into an LDS hash table of 8192 16-byte slots: the key is claimed with a
//      64-bit compare-and-swap on {p0, p1} and a 32-bit one on p2 -- whoever loses either moves on to the next slot, nobody
//      ever waits for another lane -- and the slot index is parked in rd.blk_id;
//   3. the occupied slots are numbered densely (prefix sum in slot order), one thread per slot gathers the three map points
//      and computes {n', c} with block_plane() -- the arithmetic reg_build_kernel used per block -- into the scan's table in
//      HBM (rd.pl_tab); blocks that found no slot within PT_MAX_PROBE probes get a private entry at the top of the table;
//   4. rd.blk_id[p] <- dense id; the first PT_TCAP (4864) table entries are copied into LDS, where the hash table was;
//   5. a cost evaluation streams 18 bytes per block (the fp32 feature point straight from the extractor's cloud + the 16-bit
//      id) instead of 48 and reads the plane from LDS (ids beyond the LDS part: one 32-byte gather from the table in L2);
//      whatever LDS the table leaves free caches the first records {f, id} of the scan across the evaluations of a solve.
// Every block still evaluates the numbers the other paths evaluate ({n', c} from block_plane, bit for bit), in the same order;
// results agree with the general path's to rounding (pose 1e-12) and iteration for iteration with the oracle's (tests/test_gpu_reg.py).
#define PT_SLOTS 8192
#define PT_MAX_PROBE 192
#define PT_LDS_BYTES 155648           // s_raw of reg_solve_kernel: 152 KB
#define PT_TCAP (PT_LDS_BYTES / 32)   // table entries that fit LDS
#define LL_LINE_CACHE_MAX 1024        // line blocks of a scan kept in LDS (one workgroup per scan; a group member keeps its first round)
#define PT_EMPTY_A 0xffffffffffffffffull
#define PT_EMPTY_B 0xffffffffu
#define PT_PRIVATE 0xffffu   // rd.blk_id between the two passes: no slot found, the block gets a private table entry
#define PT_INACTIVE 0xfffeu  // ... not an active plane block

struct PtSlot {
    unsigned long long a;  // (p0 << 32) | p1
    unsigned int b;        // p2
    unsigned int id;       // dense plane id (after the compaction)
};

__device__ __forceinline__ unsigned short gload_u16(const unsigned short *p) { return *(const LL_AS_GLOBAL unsigned short *)p; }

__device__ __forceinline__ unsigned int pt_hash(unsigned int p0, unsigned int p1, unsigned int p2)
{
    unsigned int h = p0 * 0x9E3779B1u;
    h ^= h >> 15;
    h += p1 * 0x85EBCA77u;
    h ^= h >> 13;
    h += p2 * 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}

// slot of the triple (inserting it if new), or PT_PRIVATE.  Lock-free and wait-free per probe: a slot belongs to the first
// {p0, p1} that lands on its `a` word AND the first p2 that lands on its `b` word; a lane that loses either race (or finds
// another key) probes on, and every lane with the same triple walks the same probe sequence to the same slot.
template <int MAX_PROBE = PT_MAX_PROBE>
__device__ __forceinline__ unsigned int pt_insert(LL_AS_LDS PtSlot *ht, unsigned int p0, unsigned int p1, unsigned int p2)
{
    const unsigned long long A = ((unsigned long long)p0 << 32) | (unsigned long long)p1;
    const unsigned int hh = pt_hash(p0, p1, p2);
    unsigned int h = hh & (PT_SLOTS - 1);
    // double hashing (odd step, power-of-two table: a full cycle).  A wavefront waits for its slowest lane, and with linear
    // probing some lane of nearly every wavefront sat in one of the long clusters (2 - 3 k cycles per round of 64 inserts
    // at 43 % load)
    const unsigned int step = ((hh >> 13) | 1u) & (PT_SLOTS - 1);
    for (int probe = 0; probe < MAX_PROBE; probe++) {
        // two independent relaxed reads decide the common case (four of five blocks find their triple already there); each
        // word of a slot is written once, by an atomic, so a stale or half-claimed view only sends the lane through the
        // compare-and-swaps.  (Not `volatile`: the memory legalizer brackets a volatile access with waits for every outstanding
        // store, one HBM round trip per insert.)
        unsigned long long a = __hip_atomic_load(&ht[h].a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned int bb = __hip_atomic_load(&ht[h].b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (a == A && bb == p2) return h;
        if (a == PT_EMPTY_A) {  // (on failure the builtin leaves the value it found in `a`)
            if (__hip_atomic_compare_exchange_strong(&ht[h].a, &a, A, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) a = A;
        }
        if (a == A) {
            if (bb == PT_EMPTY_B) {
                if (__hip_atomic_compare_exchange_strong(&ht[h].b, &bb, p2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) bb = p2;
            }
            if (bb == p2) return h;
        }
        h = (h + step) & (PT_SLOTS - 1);
    }
    return PT_PRIVATE;
}

// {n', c} of one neighbour triple -> two int4 (the arithmetic of reg_build_kernel's plane blocks: block_plane)
// SCALED: {m, beta} of ll_reg_core.h plane_scale (solve_fast3) instead of {n', c} (solve_big)
template <bool SCALED = false>
__device__ __forceinline__ void pt_plane(const f4 *map_pts, const double *pose_last, unsigned int i0, unsigned int i1, unsigned int i2, int4 &ob, int4 &oc)
{
    const f4 m0 = gload_pt(map_pts + i0), m1 = gload_pt(map_pts + i1), m2 = gload_pt(map_pts + i2);
    const double pa[3] = {(double)m0.x, (double)m0.y, (double)m0.z};
    const double pb[3] = {(double)m1.x, (double)m1.y, (double)m1.z};
    const double pc[3] = {(double)m2.x, (double)m2.y, (double)m2.z};
    double a_out[3] = {0.0, 0.0, 0.0}, v_out[3] = {0.0, 0.0, 0.0};
    (void)block_plane(pose_last, pa, pb, pc, a_out, v_out);  // degenerate triples never reach the table (build_one clears their flag)
    if (SCALED) {
        double m_[3], beta_;
        plane_scale(v_out, a_out[0], m_, &beta_);
        v_out[0] = m_[0], v_out[1] = m_[1], v_out[2] = m_[2], a_out[0] = beta_;
    }
    ob = make_int4(__double2loint(v_out[0]), __double2hiint(v_out[0]), __double2loint(v_out[1]), __double2hiint(v_out[1]));
    oc = make_int4(__double2loint(v_out[2]), __double2hiint(v_out[2]), __double2loint(a_out[0]), __double2hiint(a_out[0]));
}

__device__ __forceinline__ int4 *pt_table_global(const RegDev &rd, int b, int g, bool grouped)
{
    return rd.pl_tab + ((size_t)b * rd.tab_cap + (grouped ? (size_t)g * (rd.tab_cap / LL_GRP) : 0)) * 2;
}

// steps 1 - 4 above: the census of all blocks and the plane table of this workgroup's share of the plane blocks.  Returns the
// thread's activity mask.  Every loop that touches HBM keeps eight (four) independent loads in flight from clamped addresses.
#define PT_MAP_OFF (PT_SLOTS * 16)  // byte offset in s_raw of the id -> slot map (unsigned short[PT_SLOTS]) used during the build
template <bool GROUPED>
__device__ __noinline__ unsigned long long census_and_plane_table(const RegDev &rd, const RegConst &rc, const f4 *map_pts, int b, const RegState *st,
                                                                  int nC, int nS, uint4 *s_raw, SolveShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int G = GROUPED ? LL_GRP : 1, GS = G * RS_THREADS;
    const int g = GROUPED ? sh.grp_g : 0;
    const int p0 = g * RS_THREADS + tid;
    const int kp = (nS + GS - 1) / GS;
    const int nSp = (nS + RS_THREADS - 1) / RS_THREADS * RS_THREADS;
    const int totp = nSp + nC;
    const size_t sb = (size_t)b * rd.cap;
    LL_AS_LDS PtSlot *ht = (LL_AS_LDS PtSlot *)s_raw;
    LL_AS_LDS unsigned short *slot_of_id = (LL_AS_LDS unsigned short *)((LL_AS_LDS char *)s_raw + PT_MAP_OFF);
    LL_T0(t_census);
    for (int e = tid; e < PT_SLOTS; e += RS_THREADS) lds_store_i4((int4 *)s_raw + e, make_int4(-1, -1, -1, -1));
    if (tid == 0) sh.pt_priv = 0;
    __syncthreads();
    const int4 *nn = rd.nn + sb + rd.cap_c;
    const unsigned char *flag0 = rd.blk_flag0 + sb;
    unsigned short *ids = rd.blk_id + (size_t)b * rd.cap_s;
    // ---- census (PCR:325,425) of all blocks + pass 1 of the own plane blocks: triples -> hash slots -----------------------
    unsigned long long act = 0;
    int na = 0, nca = 0, nsa = 0;
    for (int k0 = 0; k0 * RS_THREADS < totp; k0 += 8) {
        LL_T0(t_trip);
        unsigned char fl8[8];
        int4 t8[GROUPED ? 1 : 8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = tid + (k0 + u) * RS_THREADS;
            const int jc = j < totp ? j : 0;
            const size_t src = jc < nS ? (size_t)rd.cap_c + jc : (jc >= nSp ? (size_t)(jc - nSp) : (size_t)rd.cap_c);
            fl8[u] = gload_u8(flag0 + src);
            if (!GROUPED) t8[u] = gload_i4(nn + (j < nS ? j : 0));
        }
        if (GROUPED) {  // of eight consecutive rounds exactly one (round = g mod 8) is this member's
            const int j = tid + (k0 + g) * RS_THREADS;
            t8[0] = gload_i4(nn + (j < nS ? j : 0));
        }
        unsigned int h8[GROUPED ? 1 : 8];
#ifdef LL_SOLVE_TIMING
        if (fl8[0] == 255 && t8[0].x == -12345) act |= 1ull << 63;  // (forces the loads to have landed before the timer below)
        LL_TACC(10, t_trip);
        LL_T0(t_ins);
#endif
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = tid + (k0 + u) * RS_THREADS;
            const unsigned char fl = (j < totp && (j < nS || j >= nSp)) ? fl8[u] : (unsigned char)0;
            const bool active = (fl & BLK_ACTIVE) != 0;
            if (active) {
                act |= 1ull << (k0 + u);
                na++;
            }
            if (fl & 8) {
                if (j >= nSp) nca++; else nsa++;
            }
            if (!GROUPED || u == g) {
                const int4 t = t8[GROUPED ? 0 : u];
                const unsigned int h = (active && j < nS) ? pt_insert(ht, (unsigned int)t.x, (unsigned int)t.y, (unsigned int)t.z) : PT_INACTIVE;
                if (h == PT_PRIVATE) atomicAdd(&sh.pt_priv, 1);
                h8[GROUPED ? 0 : u] = h;
            }
        }
        // the slot indices leave together at the end of the trip: a store between the inserts makes the wait for the next
        // flag byte a wait for that store (the counters are imprecise behind divergent code), one HBM round trip per block
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = tid + (k0 + u) * RS_THREADS;
            if ((!GROUPED || u == g) && j < nS) gstore_u16(ids + j, (unsigned short)h8[GROUPED ? 0 : u]);
        }
#ifdef LL_SOLVE_TIMING
        LL_TACC(11, t_ins);
#endif
    }
    LL_T0(t_sums);
    {
        const unsigned long long tot = block_sum_u64((unsigned long long)na | ((unsigned long long)nca << 20) | ((unsigned long long)nsa << 40), sh);
        na = (int)(tot & 0xfffffull);
        nca = (int)((tot >> 20) & 0xfffffull);
        nsa = (int)((tot >> 40) & 0xfffffull);
    }
    if (rc.subsample_seed && na > rc.max_blocks) {  // a13 (PCR:438-458); the random stream is indexed by the block's
        int kept = 0;                               // position in the reference's order: corners, then surfaces
        for (int k = 0; k * RS_THREADS < totp; k++) {  // (a dropped block's triple stays in the table, unused)
            if (!((act >> k) & 1ull)) continue;
            const int j = tid + k * RS_THREADS;
            const int jref = j >= nSp ? j - nSp : nC + j;
            if (subsample_drop_block(rc.subsample_seed, st->icp_iters, jref, na, rc.max_blocks))
                act &= ~(1ull << k);
            else
                kept++;
        }
        na = block_sum_int(kept, sh);
    }
    if (tid == 0) {
        sh.n_active = na;
        sh.n_corner_avail = nca;
        sh.n_surf_avail = nsa;
    }
    __syncthreads();  // (also: every insert has landed)
    LL_TACC(12, t_sums);
    LL_TACC(6, t_census);
    LL_T0(t_tab);
    LL_T0(t_cmp);
    // ---- dense ids in slot order; id -> slot map --------------------------------------------------------------------------
    constexpr int SPT = PT_SLOTS / RS_THREADS;  // slots per thread: tid, tid + 512, ... (consecutive slots per thread would put
    unsigned int occ = 0;                       // all 64 lanes of a read on one LDS bank)
#pragma unroll
    for (int i = 0; i < SPT; i++)
        if (ht[tid + i * RS_THREADS].b != PT_EMPTY_B) occ |= 1u << i;
    const int cnt = __popc(occ);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) sh.isum[wave] = incl;
    __syncthreads();
    int base = incl - cnt, T = 0;
    for (int w = 0; w < RS_WAVES; w++) {
        if (w < wave) base += sh.isum[w];
        T += sh.isum[w];
    }
    {
        int rk = base;
#pragma unroll
        for (int i = 0; i < SPT; i++)
            if (occ & (1u << i)) {
                ht[tid + i * RS_THREADS].id = (unsigned int)rk;
                slot_of_id[rk] = (unsigned short)(tid + i * RS_THREADS);
                rk++;
            }
    }
    __syncthreads();
    LL_TACC(13, t_cmp);
    LL_T0(t_pl);
    // ---- plane constants -> the table in HBM: thread t computes ids t, t + 512, ... (four triples' gathers in flight) ------
    int4 *tabG = pt_table_global(rd, b, g, GROUPED);
    double pose_last[7];
#pragma unroll
    for (int i = 0; i < 7; i++) pose_last[i] = gload_f64(st->pose_last + i);
    for (int i0 = tid; i0 < T; i0 += 4 * RS_THREADS) {
        f4 m[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int id = i0 + u * RS_THREADS;
            const unsigned int sl = slot_of_id[id < T ? id : i0];
            const unsigned long long sa = ht[sl].a;
            const unsigned int sb2 = ht[sl].b;
            m[u][0] = gload_pt(map_pts + (unsigned int)(sa >> 32));
            m[u][1] = gload_pt(map_pts + (unsigned int)sa);
            m[u][2] = gload_pt(map_pts + sb2);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int id = i0 + u * RS_THREADS;
            const double pa[3] = {(double)m[u][0].x, (double)m[u][0].y, (double)m[u][0].z};
            const double pb[3] = {(double)m[u][1].x, (double)m[u][1].y, (double)m[u][1].z};
            const double pc[3] = {(double)m[u][2].x, (double)m[u][2].y, (double)m[u][2].z};
            double a_out[3] = {0.0, 0.0, 0.0}, v_out[3] = {0.0, 0.0, 0.0};
            (void)block_plane(pose_last, pa, pb, pc, a_out, v_out);  // degenerate triples never reach the table (build_one clears their flag)
            {  // the table holds the SCALED plane {m = |n'| n', beta = |n'| c} (ll_reg_core.h plane_scale)
                double m_[3], beta_;
                plane_scale(v_out, a_out[0], m_, &beta_);
                v_out[0] = m_[0], v_out[1] = m_[1], v_out[2] = m_[2], a_out[0] = beta_;
            }
            if (id < T) {
                gstore_i4(tabG + 2 * id, make_int4(__double2loint(v_out[0]), __double2hiint(v_out[0]), __double2loint(v_out[1]), __double2hiint(v_out[1])));
                gstore_i4(tabG + 2 * id + 1, make_int4(__double2loint(v_out[2]), __double2hiint(v_out[2]), __double2loint(a_out[0]), __double2hiint(a_out[0])));
            }
        }
    }
    LL_TACC(14, t_pl);
    LL_T0(t_p2);
    // ---- pass 2: slot -> dense id (nothing but arithmetic and LDS reads between the loads and the stores of a trip) ---------
    const int region = GROUPED ? rd.tab_cap / LL_GRP : rd.tab_cap;
    for (int k0 = 0; k0 < kp; k0 += 8) {
        unsigned short h8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int p = p0 + (k0 + u) * GS;
            h8[u] = gload_u16(ids + (p < nS ? p : 0));
        }
        unsigned int id8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned int h = h8[u];
            const unsigned int sid = ht[h < PT_SLOTS ? h : 0u].id;
            id8[u] = h < PT_SLOTS ? sid : (h == PT_PRIVATE ? PT_PRIVATE : 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int p = p0 + (k0 + u) * GS;
            if (p < nS) gstore_u16(ids + p, (unsigned short)id8[u]);
        }
    }
    if (sh.pt_priv > 0) {  // (uniform; rare: the hash table was too crowded around some triples -- those blocks get entries of
        int npriv = 0;     //  their own at the top of the table region, numbered per thread and then across the workgroup)
        for (int k = 0; k < kp; k++) {
            const int p = p0 + k * GS;
            if (p < nS && gload_u16(ids + p) == PT_PRIVATE) npriv++;
        }
        int incl2 = npriv;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl2, off);
            if (lane >= off) incl2 += y;
        }
        __syncthreads();
        if (lane == 63) sh.isum[wave] = incl2;
        __syncthreads();
        int pid = incl2 - npriv;
        for (int w = 0; w < wave; w++) pid += sh.isum[w];
        for (int k = 0; k < kp; k++) {
            const int p = p0 + k * GS;
            if (p >= nS || gload_u16(ids + p) != PT_PRIVATE) continue;
            const unsigned int id = (unsigned int)(region - 1 - pid);
            pid++;
            const int4 t = gload_i4(nn + p);
            int4 ob, oc;
            pt_plane<true>(map_pts, pose_last, (unsigned int)t.x, (unsigned int)t.y, (unsigned int)t.z, ob, oc);
            gstore_i4(tabG + 2 * id, ob);
            gstore_i4(tabG + 2 * id + 1, oc);
            gstore_u16(ids + p, (unsigned short)id);
        }
    }
    __threadfence_block();
    __syncthreads();  // the hash table is dead, the table in HBM complete
    // ---- the first PT_TCAP entries -> LDS; the rest of s_raw caches records ----------------------------------------------
    const int Tl = T < PT_TCAP ? T : PT_TCAP;
    int4 *s_tab = (int4 *)s_raw;
    for (int e0 = tid; e0 < 2 * Tl; e0 += 8 * RS_THREADS) {
        int4 v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * RS_THREADS;
            v8[u] = gload_i4(tabG + (e < 2 * Tl ? e : e0));
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * RS_THREADS;
            if (e < 2 * Tl) lds_store_i4(s_tab + e, v8[u]);
        }
    }
    if (tid == 0) {
        // (T beyond the LDS part, or private entries: the evaluation takes its slower form, solver_eval3)
        // Line blocks (64 B each: sensor point + a', v' in fp64) are few in a Q-full scan and half of a voxel-filtered one; they are
        // not pipelined -- every evaluation that reads them from HBM waits a full memory latency at its end.  The first
        // LL_LINE_CACHE_MAX of them (this group member's first round) stay in LDS above the record cache, written by the
        // FILL evaluation like the plane records.
        int nl = 0;
        if (!rc.no_line_cache && T <= PT_TCAP && sh.pt_priv == 0) {
            int mine = GROUPED ? nC - g * RS_THREADS : nC;
            const int cap = GROUPED ? RS_THREADS : LL_LINE_CACHE_MAX;
            mine = mine < 0 ? 0 : (mine > cap ? cap : mine);
            const int room = (PT_LDS_BYTES / 16 - 2 * Tl) / 4;
            nl = mine < room ? mine : room;
        }
        const int kc = (PT_LDS_BYTES / 16 - 2 * Tl - 4 * nl) / RS_THREADS;
        sh.pt_T = sh.pt_priv > 0 ? PT_TCAP + 1 : T;
        sh.pt_Tl = Tl;
        sh.pt_kc = kc < kp ? kc : kp;
        sh.pt_nl = nl;
    }
    __syncthreads();
    LL_TACC(15, t_p2);
    LL_TACC(8, t_tab);
    return act;
}

// after the inlier phase has used s_raw for its tables
__device__ __forceinline__ void plane_table_reload(const RegDev &rd, int b, bool grouped, uint4 *s_raw, SolveShared &sh)
{
    const int4 *tabG = pt_table_global(rd, b, grouped ? sh.grp_g : 0, grouped);
    int4 *s_tab = (int4 *)s_raw;
    const int n = 2 * sh.pt_Tl;
    for (int e0 = threadIdx.x; e0 < n; e0 += 8 * RS_THREADS) {
        int4 v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * RS_THREADS;
            v8[u] = gload_i4(tabG + (e < n ? e : e0));
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = e0 + u * RS_THREADS;
            if (e < n) lds_store_i4(s_tab + e, v8[u]);
        }
    }
    __syncthreads();
}

struct Rec3 {
    int fx, fy, fz;  // bits of the fp32 feature point (sensor frame)
    unsigned int id;
};
struct Pl3 {
    int4 b, c;  // {n'.x, n'.y}, {n'.z, c}
};

// The plane loop of a cost evaluation, rounds [KB, KE) of this thread's blocks.  Everything the waitcnt pass has to see
// through is straight-line: loads are unconditional (clamped addresses), only arithmetic and LDS stores sit behind
// predicates -- a load behind a branch (a conditional global fallback for the plane, a cached / streamed switch inside the
// loop) made the compiler drain the whole pipeline with vmcnt(0) at every stage (first build of this path: 66 k cycles per
// evaluation against 25 k of fp64 issue).  Records are prefetched four rounds ahead (5 register sets rotating by name), the
// plane of the next record is fetched from the LDS table while the current one is evaluated.
//   FROM_CACHE: records from the LDS record cache (rounds < kc of an evaluation that does not FILL); else 18 B per lane and
//   record from HBM.  TAB_LDS: the scan's whole table is in LDS (T <= PT_TCAP) -- the fast form; otherwise a plane with
//   id >= Tl is gathered from the table in HBM behind a branch (correct, same summation order, not pipelined).
#define LL3_LOAD(R, K)                                                                             \
    {                                                                                              \
        if (FROM_CACHE_) {                                                                         \
            const int kc_ = (K) < kc ? (K) : kc - 1;                                               \
            const int4 c_ = lds_load_i4(cache + tid + kc_ * RS_THREADS);                           \
            R.fx = c_.x;                                                                           \
            R.fy = c_.y;                                                                           \
            R.fz = c_.z;                                                                           \
            R.id = (unsigned int)c_.w;                                                             \
        } else {                                                                                   \
            const int pp_ = p0 + (K) * GS;                                                         \
            const int pc_ = pp_ < nS ? pp_ : 0;                                                    \
            float fx_, fy_, fz_;                                                                   \
            gload_f3(feat + pc_, fx_, fy_, fz_);                                                   \
            R.id = gload_u16(ids + pc_);                                                           \
            R.fx = __float_as_int(fx_);                                                            \
            R.fy = __float_as_int(fy_);                                                            \
            R.fz = __float_as_int(fz_);                                                            \
        }                                                                                          \
    }
#define LL3_PLANE(Q, R)                                                                            \
    {                                                                                              \
        if (TAB_LDS || R.id < (unsigned int)Tl) {                                                  \
            Q.b = lds_load_i4(tabL + 2 * R.id);                                                    \
            Q.c = lds_load_i4(tabL + 2 * R.id + 1);                                                \
        } else {                                                                                   \
            Q.b = gload_i4(tabG + 2 * R.id);                                                       \
            Q.c = gload_i4(tabG + 2 * R.id + 1);                                                   \
        }                                                                                          \
    }
#define LL3_USE(R, Q, K)                                                                                   \
    if ((K) < KE_) {                                                                                       \
        const int pp_ = p0 + (K) * GS;                                                                     \
        if (FILL && (K) < kc) lds_store_i4(cache + tid + (K) * RS_THREADS, make_int4(R.fx, R.fy, R.fz, (int)R.id)); \
        if (pp_ < nS && ((act >> (g + G * (K))) & 1ull)) { /* (a group member's last round may reach into the lines' bits) */ \
            const double f[3] = {(double)__int_as_float(R.fx), (double)__int_as_float(R.fy), (double)__int_as_float(R.fz)}; \
            const double v[3] = {__hiloint2double(Q.b.y, Q.b.x), __hiloint2double(Q.b.w, Q.b.z), __hiloint2double(Q.c.y, Q.c.x)}; \
            const double beta_ = __hiloint2double(Q.c.w, Q.c.z);                                           \
            plane_accumulate_scaled(R_, t_, f, v, beta_, huber_a, acc);                                    \
            if (L1OUT) gstore_f64(l1_planes + pp_, plane_l1_scaled(R_, t_, f, v, beta_, huber_a, q_last)); \
        }                                                                                                  \
    }
#define LL3_PIPE(FROM_CACHE, KB, KE)                                   \
    {                                                                  \
        constexpr bool FROM_CACHE_ = FROM_CACHE;                       \
        const int KB_ = (KB), KE_ = (KE);                              \
        if (KB_ < KE_) {                                               \
            Rec3 r0, r1, r2, r3, r4;                                   \
            Pl3 q0, q1, q2, q3, q4;                                    \
            LL3_LOAD(r0, KB_)                                          \
            LL3_LOAD(r1, KB_ + 1)                                      \
            LL3_LOAD(r2, KB_ + 2)                                      \
            LL3_LOAD(r3, KB_ + 3)                                      \
            LL3_PLANE(q0, r0)                                          \
            for (int k = KB_; k < KE_; k += 5) {                       \
                LL3_LOAD(r4, k + 4)                                    \
                LL3_PLANE(q1, r1)                                      \
                LL3_USE(r0, q0, k)                                     \
                LL3_LOAD(r0, k + 5)                                    \
                LL3_PLANE(q2, r2)                                      \
                LL3_USE(r1, q1, k + 1)                                 \
                LL3_LOAD(r1, k + 6)                                    \
                LL3_PLANE(q3, r3)                                      \
                LL3_USE(r2, q2, k + 2)                                 \
                LL3_LOAD(r2, k + 7)                                    \
                LL3_PLANE(q4, r4)                                      \
                LL3_USE(r3, q3, k + 3)                                 \
                LL3_LOAD(r3, k + 8)                                    \
                LL3_PLANE(q0, r0)                                      \
                LL3_USE(r4, q4, k + 4)                                 \
            }                                                          \
        }                                                              \
    }

// workgroup evaluation of cost / g / H at x over the active blocks -> sh.sum (FILL: this evaluation writes the LDS record cache; L1OUT: it also leaves the L1 values; GROUPED: a group member's share)
template <bool FILL, bool L1OUT, bool GROUPED>
__device__ __noinline__ void solver_eval3(const RegDev &rd, int b, int nC, int nS, const double *x, double huber_a, unsigned long long act,
                                          uint4 *s_raw, const double *q_last_g, SolveShared &sh)
{
    constexpr int DEBLUR = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double q_last[4] = {0.0, 0.0, 0.0, 1.0};
    if (L1OUT) {
        q_last[0] = q_last_g[0];
        q_last[1] = q_last_g[1];
        q_last[2] = q_last_g[2];
        q_last[3] = q_last_g[3];
    }
    double *l1_planes = rd.blk_l1 + (size_t)b * rd.cap + rd.cap_c;
    double *l1_lines = rd.blk_l1 + (size_t)b * rd.cap;
    LL_CTX_DECL(x)
    double acc[LL_NACC];
#pragma unroll
    for (int i = 0; i < LL_NACC; i++) acc[i] = 0.0;
    constexpr int G = GROUPED ? LL_GRP : 1, GS = G * RS_THREADS;
    const int g = GROUPED ? sh.grp_g : 0;
    const int p0 = g * RS_THREADS + tid;
    const int Tl = sh.pt_Tl, kc = sh.pt_kc;
    const int kp = (nS + GS - 1) / GS;
    const int4 *tabL = (const int4 *)s_raw;
    int4 *cache = (int4 *)s_raw + 2 * Tl;
    const int4 *tabG = pt_table_global(rd, b, g, GROUPED);
    const unsigned short *ids = rd.blk_id + (size_t)b * rd.cap_s;
    const float4 *feat = rd.surf_feat + (size_t)b * rd.feat_stride_s;
    if (sh.pt_T <= PT_TCAP) {  // uniform: the whole table is in LDS (every C2 scan)
        constexpr bool TAB_LDS = true;
        if (!FILL) LL3_PIPE(true, 0, kc)
        LL3_PIPE(false, FILL ? 0 : kc, kp)
    } else {
        constexpr bool TAB_LDS = false;
        if (!FILL) LL3_PIPE(true, 0, kc)
        LL3_PIPE(false, FILL ? 0 : kc, kp)
    }
    plane_unfold2(acc);  // the factors 2 of the planes' rotation rows and columns (plane_accumulate_scaled leaves them out), before the lines are added
#ifdef LL_EXP_MOMENT_EVAL
    // EXPERIMENT (-DLL_EXP_MOMENT_EVAL together with -DLL_SOLVE_TIMING, never in the product library): what ONE evaluation in
    // per-plane moment form would cost, run IN ADDITION to the real one so that control flow and evaluation count stay the same -- the
    // difference of the evaluation phase's cycles between this build and the plain timing build is the cost of a moment-form evaluation:
    //   (1) per block: the record (streamed again), the plane from the LDS table, the scalar residual and the Huber test (which blocks
    //       are in the linear region) -- no accumulation;
    //   (2) per distinct plane: the plane + ten moments {N, S1, S2} (80 B, streamed from HBM: 112 B x T does not fit LDS; the scan's unused
    //       plane slots of blk_av stand in for the moment array) and the quadratic-region cost / gradient / Gauss-Newton terms from them.
    // Not included (so this is a LOWER bound of the real thing): the two moment builds per launch and the per-block correction of the
    // linear-region blocks.  The results of (1) and (2) go nowhere (a never-true test keeps them alive).
    {
        double xacc[LL_NACC];
#pragma unroll
        for (int i = 0; i < LL_NACC; i++) xacc[i] = 0.0;
        int n_lin = 0;
        for (int k0 = 0; k0 < kp; k0 += 8) {  // eight rounds' records in flight (the real loop keeps five)
            float fx8[8], fy8[8], fz8[8];
            unsigned int id8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int pp = p0 + (k0 + u) * GS;
                const int pc = pp < nS ? pp : 0;
                gload_f3(feat + pc, fx8[u], fy8[u], fz8[u]);
                id8[u] = gload_u16(ids + pc);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + u, pp = p0 + k * GS;
                const unsigned int idl = id8[u] < (unsigned int)Tl ? id8[u] : 0u;
                const int4 qb = lds_load_i4(tabL + 2 * idl), qc = lds_load_i4(tabL + 2 * idl + 1);
                const double f[3] = {(double)fx8[u], (double)fy8[u], (double)fz8[u]};
                const double m[3] = {__hiloint2double(qb.y, qb.x), __hiloint2double(qb.w, qb.z), __hiloint2double(qc.y, qc.x)};
                const double beta = __hiloint2double(qc.w, qc.z);
                const double q0 = R_[0] * f[0] + R_[1] * f[1] + R_[2] * f[2] + t_[0], q1 = R_[3] * f[0] + R_[4] * f[1] + R_[5] * f[2] + t_[1],
                             q2 = R_[6] * f[0] + R_[7] * f[1] + R_[8] * f[2] + t_[2];
                const double e = m[0] * q0 + m[1] * q1 + m[2] * q2 - beta;
                if (k < kp && pp < nS && ((act >> (g + G * k)) & 1ull) && e * e > huber_a * huber_a) n_lin++;
            }
        }
        const double *mom = rd.blk_av + (size_t)b * 6 * rd.cap + 2 * (size_t)rd.cap_c;  // (stand-in storage: 80 B per plane id)
        for (int id = tid; id < sh.pt_T && id < Tl; id += RS_THREADS) {
            const int4 qb = lds_load_i4(tabL + 2 * id), qc = lds_load_i4(tabL + 2 * id + 1);
            const double m[3] = {__hiloint2double(qb.y, qb.x), __hiloint2double(qb.w, qb.z), __hiloint2double(qc.y, qc.x)};
            const double beta = __hiloint2double(qc.w, qc.z);
            double mo[10];
#pragma unroll
            for (int i = 0; i < 10; i += 2) {
                const double2 v2 = gload_d2(reinterpret_cast<const double2 *>(mom + (size_t)id * 10 + i));
                mo[i] = v2.x;
                mo[i + 1] = v2.y;
            }
            // sensor-frame normal and offset, then the quadratic form in the moments (the algebra of DESIGN section 8's analysis)
            const double ms[3] = {R_[0] * m[0] + R_[3] * m[1] + R_[6] * m[2], R_[1] * m[0] + R_[4] * m[1] + R_[7] * m[2], R_[2] * m[0] + R_[5] * m[1] + R_[8] * m[2]};
            const double bs = m[0] * t_[0] + m[1] * t_[1] + m[2] * t_[2] - beta;
            const double N = mo[0], S1[3] = {mo[1], mo[2], mo[3]};
            const double S2[9] = {mo[4], mo[5], mo[6], mo[5], mo[7], mo[8], mo[6], mo[8], mo[9]};
            double u[3];
#pragma unroll
            for (int i = 0; i < 3; i++) u[i] = S2[3 * i] * ms[0] + S2[3 * i + 1] * ms[1] + S2[3 * i + 2] * ms[2];
            const double l1 = ms[0] * S1[0] + ms[1] * S1[1] + ms[2] * S1[2];
            xacc[27] += 0.5 * (ms[0] * u[0] + ms[1] * u[1] + ms[2] * u[2] + 2.0 * bs * l1 + N * bs * bs);
            const double w[3] = {u[0] + bs * S1[0], u[1] + bs * S1[1], u[2] + bs * S1[2]};
            double gt[3], sx[3];
            cross3(w, ms, gt);
            cross3(S1, ms, sx);
            const double gb = l1 + N * bs;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                xacc[21 + i] += 2.0 * gt[i];
                xacc[24 + i] += gb * ms[i];
            }
            // H_rr = 4 [ms]x S2 [ms]x^T, H_rt = 2 (S1 x ms) ms^T, H_tt = N ms ms^T (sensor frame; the common rotation is applied once at the end)
            double A[9];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double row[3] = {S2[3 * r], S2[3 * r + 1], S2[3 * r + 2]};
                double c[3];
                cross3(row, ms, c);
                A[3 * r] = c[0], A[3 * r + 1] = c[1], A[3 * r + 2] = c[2];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double col[3] = {A[c], A[3 + c], A[6 + c]};
                double h[3];
                cross3(ms, col, h);
#pragma unroll
                for (int r = 0; r <= c; r++) xacc[hidx(r, c)] += 4.0 * h[r];
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    xacc[hidx(i, 3 + j)] += 2.0 * sx[i] * ms[j];
                    if (j >= i) xacc[hidx(3 + i, 3 + j)] += N * ms[i] * ms[j];
                }
        }
        double xs = (double)n_lin;
#pragma unroll
        for (int i = 0; i < LL_NACC; i++) xs += xacc[i];
        if (xs == 1.2345678e300) acc[27] += xs;  // (never: keeps the experiment's arithmetic and loads alive)
    }
#endif
    {
        // line blocks (a few hundred per Mid-40 scan): the 65-byte fp64 form
        const size_t sb = (size_t)b * rd.cap;
        const double *av = rd.blk_av + (size_t)b * 6 * rd.cap;
        const int kpl = (nS + RS_THREADS - 1) / RS_THREADS;  // the lines' first round in the activity mask
        const int nl = sh.pt_nl;
        int4 *lc = (int4 *)s_raw + (PT_LDS_BYTES / 16 - 4 * nl);  // four planes of nl entries: {f}, {a0, v0}, {v1, v2}, {a1, a2}
        int k = 0;
        for (int l = p0; l < nC; l += GS, k++) {
            if (!((act >> (kpl + g + G * k)) & 1ull)) continue;
            BlkRegs br;
            const int j = GROUPED ? (k == 0 ? tid : nl) : l;  // its place in the LDS copy (>= nl: none)
            if (!FILL && j < nl) {
                const int4 c0 = lds_load_i4(lc + j), c1 = lds_load_i4(lc + nl + j), c2 = lds_load_i4(lc + 2 * nl + j), c3 = lds_load_i4(lc + 3 * nl + j);
                br.f = make_float4(__int_as_float(c0.x), __int_as_float(c0.y), __int_as_float(c0.z), __int_as_float(c0.w));
                br.a0 = __hiloint2double(c1.y, c1.x);
                br.v0 = __hiloint2double(c1.w, c1.z);
                br.v1 = __hiloint2double(c2.y, c2.x);
                br.v2 = __hiloint2double(c2.w, c2.z);
                br.a1 = __hiloint2double(c3.y, c3.x);
                br.a2 = __hiloint2double(c3.w, c3.z);
            } else {
                load_blk(rd, sb, av, l, br);
                if (FILL && j < nl) {
                    lds_store_i4(lc + j, make_int4(__float_as_int(br.f.x), __float_as_int(br.f.y), __float_as_int(br.f.z), __float_as_int(br.f.w)));
                    lds_store_i4(lc + nl + j, make_int4(__double2loint(br.a0), __double2hiint(br.a0), __double2loint(br.v0), __double2hiint(br.v0)));
                    lds_store_i4(lc + 2 * nl + j, make_int4(__double2loint(br.v1), __double2hiint(br.v1), __double2loint(br.v2), __double2hiint(br.v2)));
                    lds_store_i4(lc + 3 * nl + j, make_int4(__double2loint(br.a1), __double2hiint(br.a1), __double2loint(br.a2), __double2hiint(br.a2)));
                }
            }
            const double a[3] = {br.a0, br.a1, br.a2};
            const double v[3] = {br.v0, br.v1, br.v2};
            LL_CTX_ACCUM(BLK_LINE, br.f, a, v, huber_a, acc);
            if (L1OUT) {
                double l1;
                LL_CTX_L1(l1, BLK_LINE, br.f, a, v, huber_a, q_last);
                l1_lines[l] = l1;
            }
        }
    }
    // A wavefront none of whose threads owns an active block (voxel-filtered scans of a few hundred blocks: half of the eight)
    // has nothing but +0.0 to add: it writes the zeros instead of running 28 six-step reductions beside the wavefront that
    // shares its SIMD (the 504 data-parallel moves and adds are half of such an evaluation's cycles).  Same sums, bit for bit.
    if (__ballot(act != 0ull) != 0ull) {
        wave_sum_acc(acc, sh.red[wave], lane);
    } else if (lane < LL_NACC) {
        sh.red[wave][lane] = 0.0;
    }
    __syncthreads();
    if (tid < LL_NACC) {
        double s = 0.0;
        for (int w = 0; w < RS_WAVES; w++) s += sh.red[w][tid];
        sh.sum[tid] = s;
    }
    __syncthreads();
    if (GROUPED) {
        LL_T0(t_grp);
        group_reduce<L1OUT>(rd, b, sh);
        LL_TACC(9, t_grp);
    }
}
#undef LL3_LOAD
#undef LL3_PLANE
#undef LL3_USE
#undef LL3_PIPE


// test tap (ll_debug_quintic): the sequential and the wavefront form of the fit on n argument sets, one wavefront each
__global__ __launch_bounds__(64) void debug_quintic_kernel(const double *args, int n, double *out_seq, double *out_wave)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const double *a = args + 10 * (size_t)i;
    const double w = lm_quintic_min_step_wave(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], lane);
    if (lane == 0) {
        out_wave[i] = w;
        out_seq[i] = lm_quintic_min_step(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]);
    }
}
void launch_debug_quintic(const double *args, int n, double *out_seq, double *out_wave, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(debug_quintic_kernel, dim3(n), dim3(64), 0, s, args, n, out_seq, out_wave);
}

// one ceres::Solve on the plane-table layout: starts at x0, leaves the result in sh.ctl
template <bool WANT_L1, bool GROUPED>
__device__ __forceinline__ void solver_lm3(const RegDev &rd, const RegConst &rc, int b, int nC, int nS, const double *x0, int max_iter,
                                           int n_active, unsigned long long act, uint4 *s_raw, const double *q_last, SolveShared &sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        lm_begin(sh.ctl, x0, max_iter, rc.bound);
        sh.l1_valid = 0;
    }
    __syncthreads();
    {
        LL_T0(t0);
        solver_eval3<true, false, GROUPED>(rd, b, nC, nS, sh.ctl.x, rc.huber_a, act, s_raw, q_last, sh);
        LL_TACC(0, t0);
    }
    {
        LL_T0(t1);
        if (tid == 0) sh.need = lm_init(sh.ctl, sh.sum, n_active);
        __syncthreads();
        LL_TACC(1, t1);
    }
    while (sh.need) {
        const bool spec = WANT_L1 && sh.ctl.iteration >= max_iter;  // if this candidate is accepted it is the solve's result
        LL_T0(t0);
        if (spec)
            solver_eval3<false, true, GROUPED>(rd, b, nC, nS, sh.ctl.cand, rc.huber_a, act, s_raw, q_last, sh);
        else
            solver_eval3<false, false, GROUPED>(rd, b, nC, nS, sh.ctl.cand, rc.huber_a, act, s_raw, q_last, sh);
        LL_TACC(0, t0);
        LL_T0(t1);
        if (tid < 64) {  // the controller's wavefront: lane 0 steps the controller, all of it fits a line search's interpolant
            const int need = lm_update_wave(sh.ctl, sh.sum, sh.fit, tid);
            if (tid == 0) {
                sh.need = need;
                sh.l1_valid = (spec && !need && sh.ctl.last_accept == 1) ? 1 : 0;
            }
        }
        __syncthreads();
        LL_TACC(1, t1);
    }
}

// L1 values at the prerun result -> inlier threshold -> prune (PCR:476-499), the activity mask in place of LDS flags.
// Returns the pruned mask.
template <int NK, bool GROUPED>
__device__ __noinline__ unsigned long long inlier_phase3(const RegDev &rd, const RegConst &rc, int b, RegState *st, SolveShared &sh, uint4 *s_raw,
                                                         unsigned long long act, int nC, int nS)
{
    constexpr int DEBLUR = 0;
    const int tid = threadIdx.x;
    const int nSp = (nS + RS_THREADS - 1) / RS_THREADS * RS_THREADS;
    const int totp = nSp + nC;
    const size_t sb = (size_t)b * rd.cap;
    LL_T0(t_l1);
    double *l1g = rd.blk_l1 + sb;
    if (!sh.l1_valid) {
        // rare: the prerun ended on a rejected step (or converged early).  Every workgroup evaluates its own share, from the
        // table in HBM (a plain rolled loop)
        LL_CTX_DECL(sh.ctl.x)
        constexpr int G = GROUPED ? LL_GRP : 1, GS = G * RS_THREADS;
        const int g = GROUPED ? sh.grp_g : 0;
        const int p0 = g * RS_THREADS + tid;
        const int4 *tabG = pt_table_global(rd, b, g, GROUPED);
        const unsigned short *ids = rd.blk_id + (size_t)b * rd.cap_s;
        const float4 *feat = rd.surf_feat + (size_t)b * rd.feat_stride_s;
        const double *av = rd.blk_av + (size_t)b * 6 * rd.cap;
        int k = 0;
        for (int p = p0; p < nS; p += GS, k++) {
            if (!((act >> (g + G * k)) & 1ull)) continue;
            const float4 ff = gload_f4(feat + p);
            const unsigned int id = gload_u16(ids + p);
            const int4 qb = gload_i4(tabG + 2 * id), qc = gload_i4(tabG + 2 * id + 1);
            const double f[3] = {(double)ff.x, (double)ff.y, (double)ff.z};
            const double v[3] = {__hiloint2double(qb.y, qb.x), __hiloint2double(qb.w, qb.z), __hiloint2double(qc.y, qc.x)};
            l1g[rd.cap_c + p] = plane_l1_scaled(R_, t_, f, v, __hiloint2double(qc.w, qc.z), rc.huber_a, st->pose_last);  // (the table holds scaled planes)
        }
        const int kpl = nSp / RS_THREADS;
        k = 0;
        for (int l = p0; l < nC; l += GS, k++) {
            if (!((act >> (kpl + g + G * k)) & 1ull)) continue;
            BlkRegs br;
            load_blk(rd, sb, av, l, br);
            const double a[3] = {br.a0, br.a1, br.a2};
            const double v[3] = {br.v0, br.v1, br.v2};
            double l1;
            LL_CTX_L1(l1, BLK_LINE, br.f, a, v, rc.huber_a, st->pose_last);
            l1g[l] = l1;
        }
        if (GROUPED)
            group_barrier<true>(rd, b, sh);  // the other members' shares
        else
            __syncthreads();
    } else if (tid == 0) {
        sh.tcyc[9] += 1;  // LL_SOLVE_TIMING: how often the shortcut was taken
    }
    double l1r[NK];  // the thread's register tile: block j = tid + k * RS_THREADS
    {
        const int kt = (totp + RS_THREADS - 1) / RS_THREADS;
#pragma unroll
        for (int k = 0; k < NK; k++) {  // unconditional loads from clamped addresses: all in flight together
            const int j = tid + k * RS_THREADS;
            double v = -1.0;
            if (k < kt) {
                const int jc = j < totp ? j : 0;
                const size_t src = jc < nS ? (size_t)rd.cap_c + jc : (jc >= nSp ? (size_t)(jc - nSp) : (size_t)rd.cap_c);
                v = gload_f64(l1g + src);
            }
            l1r[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NK; k++) l1r[k] = ((act >> k) & 1ull) ? l1r[k] : -1.0;
    }
    __syncthreads();
    LL_TACC(2, t_l1);

    inlier_threshold_regs<NK>(l1r, totp, (unsigned long long *)s_raw, sh, rc);  // overwrites the LDS plane table and record cache

    // ---- prune (PCR:487-499) ---------------------------------------------------------------------------------
    LL_T0(t_prune);
    {
        const double thr = sh.thr;
        int na = 0;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            if (!((act >> k) & 1ull)) continue;
            if (l1r[k] > thr)
                act &= ~(1ull << k);
            else
                na++;
        }
        na = block_sum_int(na, sh);
        if (tid == 0) sh.n_active = na;
        __syncthreads();
    }
    LL_TACC(7, t_prune);
    return act;
}

template <bool GROUPED>
__device__ void solve_fast3(const RegDev &rd, const RegConst &rc, const f4 *map_pts, int b, RegState *st, SolveShared &sh, uint4 *s_raw)
{
    constexpr int DEBLUR = 0;
    const int tid = threadIdx.x;
    const int nC = rd.n_corner[b], nS = rd.n_surf[b];
    const int nSp = (nS + RS_THREADS - 1) / RS_THREADS * RS_THREADS;  // lines start at a whole round
    const int totp = nSp + nC;                                        // <= FAST_MAX_BLOCKS (scan_is_compact)
    const size_t sb = (size_t)b * rd.cap;
    if (tid < 16) sh.tcyc[tid] = 0;
    __syncthreads();
    LL_T0(t_total);

    // ---- flags -> the thread's activity mask (bit k: block tid + k * RS_THREADS in the order planes, padding, lines), census
    //      (PCR:325,425), and the scan's plane table (this workgroup's share of it) ------------------------------------------
    unsigned long long act = census_and_plane_table<GROUPED>(rd, rc, map_pts, b, st, nC, nS, s_raw, sh);

    // ---- prerun solve (PCR:463-474); its last evaluation also leaves the per-block L1 values in blk_l1 ------------
    solver_lm3<true, GROUPED>(rd, rc, b, nC, nS, st->inc, rc.ceres_prerun_times, sh.n_active, act, s_raw, st->pose_last, sh);
    int lm_iters = sh.ctl.iteration;

    if (totp <= 36 * RS_THREADS)  // the Mid-40 configurations: a 36-entry register tile per thread
        act = inlier_phase3<36, GROUPED>(rd, rc, b, st, sh, s_raw, act, nC, nS);
    else
        act = inlier_phase3<FAST_MAXK, GROUPED>(rd, rc, b, st, sh, s_raw, act, nC, nS);

    // ---- final solve (PCR:501-508) -----------------------------------------------------------------------------
    {
        __shared__ double x_start_3[7];
        if (tid < 7) x_start_3[tid] = sh.ctl.x[tid];
        plane_table_reload(rd, b, GROUPED, s_raw, sh);  // (its barrier also publishes x_start_3)
        solver_lm3<false, GROUPED>(rd, rc, b, nC, nS, x_start_3, rc.ceres_max_iterations, sh.n_active, act, s_raw, st->pose_last, sh);
    }
    lm_iters += sh.ctl.iteration;
    if (GROUPED) {
        group_barrier<false>(rd, b, sh);  // nobody reads st->inc / st->pose_last any more
        if (sh.grp_g != 0) return;
        if (sh.grp_abort) {  // a barrier timed out: nothing this group computed can be trusted (reg_finalize_kernel rejects the scan)
            if (tid == 0) {
                st->aborted = 1;
                st->done = 1;
                st->icp_iters += 1;
            }
            return;
        }
    }
    solve_epilogue(rc, st, sh, lm_iters);
#ifdef LL_SOLVE_TIMING
    LL_TACC(5, t_total);
    if (tid == 0)
        for (int i = 0; i < 16; i++) st->dbg_cycles[i] += sh.tcyc[i];
#endif
}

// The Mid-40 batches: no motion deblur, every scan within FAST_MAX_BLOCKS (launch_reg_solve decides per batch from the host's feature
// counts; everything else goes to reg_solve_big_kernel, ll_reg_big_path.h).
__global__ __launch_bounds__(RS_THREADS) void reg_solve_kernel(RegDev rd, RegConst rc, const f4 *map_surf)
{
    __shared__ SolveShared sh;
    // 152 KB: hash table -> plane table + record cache; the inlier phase's tables in between
    __shared__ uint4 s_raw[PT_LDS_BYTES / 16];
    int b = blockIdx.x, g = 0, G = 1;
    if (rc.solve_group > 1) {  // grouped launch (n_scans * G workgroups): scan and rank by ticket, see group_barrier
        if (threadIdx.x == 0) sh.grp_seq = atomicAdd(rd.grp_ctl, 1);
        __syncthreads();
        G = rc.solve_group;
        b = sh.grp_seq / G;
        g = sh.grp_seq - b * G;
        __syncthreads();
    }
    RegState *st = rd.state + b;
    if (st->done) return;  // the same answer for every member: the epilogue that sets it runs behind the group's barriers
    {
        const int nS_ = rd.n_surf[b], nC_ = rd.n_corner[b];
        if ((nS_ + RS_THREADS - 1) / RS_THREADS * RS_THREADS + nC_ > FAST_MAX_BLOCKS || !scan_is_compact(rd, rc, b)) {
            // cannot happen (the host launches this kernel only for batches it holds): fail loudly -- rejected and reported -- instead of answering
            if (g == 0 && threadIdx.x == 0) {
                st->aborted = 1;
                st->done = 1;
                st->icp_iters += 1;
            }
            return;
        }
    }
    if (threadIdx.x == 0) {
        sh.grp_g = g;
        sh.grp_G = G;
        sh.grp_seq = 0;
        sh.xch_seq = 0;
        sh.xch_epoch = rc.xch_epoch;
        sh.grp_abort = (rc.test_group_abort && G > 1) ? 1 : 0;  // test switch: behave as if the first barrier had timed out
    }
    __syncthreads();
    if (G > 1)
        solve_fast3<true>(rd, rc, map_surf, b, st, sh, s_raw);
    else
        solve_fast3<false>(rd, rc, map_surf, b, st, sh, s_raw);
}

#include "ll_reg_big_path.h"

__global__ void reg_finalize_kernel(RegDev rd, RegConst rc, int n_scans)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_scans) return;
    RegState *st = rd.state + b;
    st->result = 1;
    st->accepted = 1;
    if (st->gated || st->icp_iters == 0) return;
    st->inlier_thr = st->inlier_thr * st->final_cost / st->initial_cost;  // PCR:559
    const float minimize_cost = (float)st->final_cost;                    // PCR:192,519
    // (an aborted solve, or anything non-finite that reached the pose, is a rejection too: NaN compares false with both limits)
    const bool broken = st->aborted || !((st->angular_diff - st->angular_diff) == 0.0) || !((st->final_cost - st->final_cost) == 0.0);
    if (broken || st->angular_diff > (double)rc.para_max_angular_rate || minimize_cost > rc.max_final_cost) {  // PCR:561
        for (int i = 0; i < 7; i++) st->pose_curr[i] = st->pose_last[i];
        st->result = 0;
        st->accepted = 0;
    }
}

__global__ void cloud_transform_kernel(const float4 *in, float4 *out, int n, const double *pose)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[7];
#pragma unroll
    for (int k = 0; k < 7; k++) p[k] = pose[k];
    const float4 v = in[i];
    float o[3];
    point_to_map(p, v.x, v.y, v.z, o);
    out[i] = make_float4(o[0], o[1], o[2], v.w);  // intensity copied, PCR:659
}

// ---- launch wrappers -------------------------------------------------------------------------------------------
void launch_reg_knn_build(const RegDev &rd, const RegConst &rc, const Grid &gc, const Grid &gs, int n_scans, int iter,
                          int max_nc, int max_ns, hipStream_t s)
{
    if (iter >= rc.knn_reuse_from && rc.knn_reuse) {
        if (max_nc + max_ns <= 0) return;
        const int mx = max_nc > max_ns ? max_nc : max_ns;
        dim3 cgrid((mx + RQ_PER * RQ_THREADS - 1) / (RQ_PER * RQ_THREADS), n_scans, 2);
        (void)hipMemsetAsync(rd.work_cnt, 0, (size_t)n_scans * 4 * sizeof(int), s);
        hipLaunchKernelGGL(reg_requery_kernel, cgrid, dim3(RQ_THREADS), 0, s, rd, rc, gc, gs, iter);
        for (int seg0 = 0; seg0 < 2 * n_scans; seg0 += RL_MAX_SEG) {
            const int n_seg = 2 * n_scans - seg0 < RL_MAX_SEG ? 2 * n_scans - seg0 : RL_MAX_SEG;
            static const int local_seg = getenv("LL_LIST_NO_LOCAL_OFFSETS") ? 0 : RL_LOCAL_SEG;  // (A/B switch)
            if (n_seg <= local_seg) {
                hipLaunchKernelGGL(reg_list_kernel<true>, dim3(256), dim3(RL_THREADS), 0, s, rd, rc, gc, gs, iter, seg0, n_seg);
            } else {
                hipLaunchKernelGGL(reg_list_offsets_kernel, dim3(1), dim3(1024), 0, s, rd, seg0, n_seg);
                hipLaunchKernelGGL(reg_list_kernel<false>, dim3(n_scans >= 64 ? RL_BLOCKS : 256), dim3(RL_THREADS), 0, s, rd, rc, gc, gs, iter, seg0, n_seg);
            }
        }
        return;
    }
    // corner and surface queries share every launch (blockIdx.z = kind): the few hundred corner queries of a scan
    // are latency-bound on their own and would otherwise serialise three more launches per iteration
    const int mx = max_nc > max_ns ? max_nc : max_ns;
    if (mx <= 0) return;
    dim3 grid((mx + KB_THREADS - 1) / KB_THREADS, n_scans, 2);
    // large scans: the surface queries go to the tile kernel (ll_knn_kernels.hip) in the order of the map cells they fall into (sorted at
    // ICP iterations 0 and 1: the first pose update moves the queries by a good part of a cell, the later ones by centimetres); it also
    // builds their blocks and, without motion deblur, transforms them itself
    const bool tile = rc.knn_tile && max_ns >= LL_KNN_TILE_MIN_SURF && max_ns <= LL_KNN_TILE_MAX_SURF;
    const bool fused = tile && !rc.if_motion_deblur;
    // small batches: the corner queries one per wavefront, and the surface queries too when the scans are small
    int coop_kinds = 0;
    if (rc.knn_coop && n_scans <= LL_KNN_COOP_MAX_SCANS) {
        if (max_nc > 0) coop_kinds |= 1;
        if (max_ns > 0 && max_ns <= LL_KNN_COOP_MAX_SURF && !tile) coop_kinds |= 2;
    }
    const bool corner_in_tile = tile && max_nc > 0 && !(coop_kinds & 1);  // ... otherwise they ride in the tile launch
    {
        const int skip = fused ? (corner_in_tile ? 3 : 2) : 0;
        if (skip != 3 && (skip == 0 || max_nc > 0)) {
            const int mt = skip == 2 ? max_nc : mx;
            hipLaunchKernelGGL(reg_transform_kernel, dim3((mt + KB_THREADS - 1) / KB_THREADS, n_scans, 2), dim3(KB_THREADS), 0, s, rd, rc, skip);
        }
    }
    if (tile && iter <= rc.knn_tile_last_sort) launch_reg_qsort(rd, rc, gc, gs, n_scans, corner_in_tile ? max_nc : 0, max_ns, fused, s);
    if (coop_kinds) {
        const int mq = (coop_kinds & 2) ? mx : max_nc;
        hipLaunchKernelGGL(reg_knn_coop_kernel, dim3((mq * 64 + KC_THREADS - 1) / KC_THREADS, n_scans, 2), dim3(KC_THREADS), 0, s, rd, rc, gc, gs, iter, coop_kinds);
    }
    const int done_kinds = coop_kinds | (tile ? 2 : 0) | (corner_in_tile ? 1 : 0);  // kinds that do not need the per-lane kernel
    if ((max_nc > 0 && !(done_kinds & 1)) || (max_ns > 0 && !(done_kinds & 2))) {
        const int mk = (done_kinds & 2) ? max_nc : ((done_kinds & 1) ? max_ns : mx);
        hipLaunchKernelGGL(reg_knn_kernel, dim3((mk + KB_THREADS - 1) / KB_THREADS, n_scans, 2), dim3(KB_THREADS), 0, s, rd, rc, gc, gs, iter, done_kinds);
    }
    if (tile) {
        launch_reg_knn_tile(rd, rc, gc, gs, n_scans, iter, corner_in_tile ? max_nc : 0, max_ns, fused, s);
        if (max_nc > 0 && !corner_in_tile)
            hipLaunchKernelGGL(reg_build_kernel, dim3((max_nc + KB_THREADS - 1) / KB_THREADS, n_scans, 2), dim3(KB_THREADS), 0, s, rd, rc, gc, gs, 2);
    } else {
        hipLaunchKernelGGL(reg_build_kernel, grid, dim3(KB_THREADS), 0, s, rd, rc, gc, gs, 0);
    }
}
// batches reg_solve_kernel holds: no motion deblur, the largest scan within FAST_MAX_BLOCKS (planes padded to whole rounds + lines)
bool reg_solve_fast_eligible(const RegConst &rc, int max_nc, int max_ns)
{
    return !rc.if_motion_deblur && !rc.force_general && (max_ns + RS_THREADS - 1) / RS_THREADS * RS_THREADS + max_nc <= FAST_MAX_BLOCKS;
}
void launch_reg_solve(const RegDev &rd, const RegConst &rc, const Grid &gs, int n_scans, int max_nc, int max_ns, int iter, hipStream_t s)
{
    if (reg_solve_small_eligible(rc, max_nc, max_ns))  // voxel-filtered scans: one or four wavefronts per scan (ll_reg_small_kernels.hip)
        launch_reg_solve_small(rd, rc, gs, n_scans, max_nc, max_ns, iter, s);
    else if (reg_solve_fast_eligible(rc, max_nc, max_ns))  // Mid-40 batches: solve_fast3 (one workgroup per scan, or a group of them for small batches)
        hipLaunchKernelGGL(reg_solve_kernel, dim3(n_scans * (rc.solve_group > 1 ? rc.solve_group : 1)), dim3(RS_THREADS), 0, s, rd, rc, gs.pts);
    else if (rc.if_motion_deblur)
        hipLaunchKernelGGL(reg_solve_big_kernel<1>, dim3(n_scans), dim3(RS_THREADS), 0, s, rd, rc, gs.pts);
    else
        hipLaunchKernelGGL(reg_solve_big_kernel<0>, dim3(n_scans), dim3(RS_THREADS), 0, s, rd, rc, gs.pts);
}
void launch_reg_finalize(const RegDev &rd, const RegConst &rc, int n_scans, hipStream_t s)
{
    hipLaunchKernelGGL(reg_finalize_kernel, dim3((n_scans + 63) / 64), dim3(64), 0, s, rd, rc, n_scans);
}
// Mid-100: the selected features of `heads` consecutive extractor slots (the lidars of one sweep) become ONE registrar scan,
// corner clouds and surface clouds each concatenated in head order (laser_feature_extractor.hpp:348-358), device to device.
// grid (chunks of the extractor stride, n_scans * heads, 2 kinds).  Points beyond the registrar's capacity are not written;
// the counts are, so the host sees the overflow.
__global__ void reg_merge_heads_kernel(const float4 *fe_corner, const float4 *fe_surf, const int *fe_nc, const int *fe_ns, int fe_stride,
                                       int heads, float4 *dst_corner, float4 *dst_surf, int *dst_nc, int *dst_ns, int dst_stride)
{
    const int slot = blockIdx.y, kind = blockIdx.z;
    const int b = slot / heads, h = slot - b * heads;
    const int *cnt = kind ? fe_ns : fe_nc;
    int off = 0;
    for (int k = 0; k < h; k++) off += cnt[b * heads + k];
    const int n = cnt[slot];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && off + i < dst_stride) {
        const float4 *src = (kind ? fe_surf : fe_corner) + (size_t)slot * fe_stride;
        float4 *dst = (kind ? dst_surf : dst_corner) + (size_t)b * dst_stride;
        dst[off + i] = src[i];
    }
    if (i == 0 && h == heads - 1) (kind ? dst_ns : dst_nc)[b] = off + n;
}
void launch_reg_merge_heads(const float4 *fe_corner, const float4 *fe_surf, const int *fe_nc, const int *fe_ns, int fe_stride, int heads,
                            float4 *dst_corner, float4 *dst_surf, int *dst_nc, int *dst_ns, int dst_stride, int n_scans, hipStream_t s)
{
    hipLaunchKernelGGL(reg_merge_heads_kernel, dim3((fe_stride + 255) / 256, n_scans * heads, 2), dim3(256), 0, s, fe_corner, fe_surf, fe_nc,
                       fe_ns, fe_stride, heads, dst_corner, dst_surf, dst_nc, dst_ns, dst_stride);
}
// The history's frames, oldest first, into one cloud (laser_mapping.hpp:519-530): segment g of the table is {first point of the frame in
// `frames`, its first position in `out`}, the table ends with {-, total}.  One launch instead of one device-to-device copy per frame
// (20 frames x 2 kinds per refresh of the match buffer: the copies' launch overhead was a third of the refresh).
__global__ __launch_bounds__(256) void history_concat_kernel(const float4 *frames, const int2 *table, int n_seg, float4 *out)
{
    __shared__ int2 s_tab[LL_HIST_CONCAT_MAX + 1];
    for (int e = threadIdx.x; e <= n_seg; e += 256) s_tab[e] = table[e];
    __syncthreads();
    const int total = s_tab[n_seg].y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int lo = 0, hi = n_seg - 1;  // the last segment whose first output position is <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_tab[mid].y <= i)
                lo = mid;
            else
                hi = mid - 1;
        }
        out[i] = frames[(size_t)s_tab[lo].x + (size_t)(i - s_tab[lo].y)];
    }
}
void launch_history_concat(const float4 *frames, const int2 *d_table, int n_seg, int total, float4 *out, hipStream_t s)
{
    if (n_seg <= 0 || total <= 0) return;
    const int blocks = (total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024;
    hipLaunchKernelGGL(history_concat_kernel, dim3(blocks), dim3(256), 0, s, frames, d_table, n_seg, out);
}

void launch_cloud_transform(const float4 *in, float4 *out, int n, const double *d_pose, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(cloud_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out, n, d_pose);
}

}  // namespace ll
