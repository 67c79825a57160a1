// ll_reg_small_kernels.hip -- the solver for SMALL scans (gfx950, wave64): the reference's real operating point.
//
// laser_mapping.hpp:1367-1373 voxel-filters the feature clouds before every registration (input_downsample_mode, leaf 0.1 / 0.4 m) and
// config/performance_*.yaml caps a problem at maximum_residual_blocks = 200: a registration then has a few hundred residual blocks, not
// the 17 000 of an unfiltered Mid-40 sweep.  reg_solve_kernel (ll_reg_kernels.hip) gives every scan a 512-thread workgroup, 158 KB of LDS
// and therefore a whole CU; on such a scan its eight wavefronts hold two blocks per lane, a third of the launch clears and scans fixed-size
// tables, and a batch lasts as long as its slowest scan's Levenberg-Marquardt controller while 250 CUs idle (profiles/r04c_qpipe_*:
// 772 us per launch of 256 scans).
//
// Here a scan is ONE wavefront (batches of >= LL_SMALL_W1_MIN_SCANS scans: many scans per CU -- the controller of one overlaps the
// evaluations of the others; throughput) or FOUR wavefronts (smaller batches: one scan per CU anyway, so the evaluations are spread over
// its four SIMDs; latency), and everything a registration's inner loop (point_cloud_registration.hpp:460-531) touches more than once lives in
// a few KB of LDS:
//   census    the scan's candidate blocks in the reference's order (corner queries, then surface queries): flags -> active / available
//             counts (PCR:325, 425), the reproducible sub-sampling of PCR:438-458 (ll_reg_core.h subsample_drop_block), and a dense
//             numbering of the kept blocks (ballot + popcount prefix: lines first, then planes);
//   build     line blocks are copied from the 65-byte records the k-NN stage wrote (ll_reg_query.h build_one); a plane block's {n', c}
//             is computed here from its neighbour triple with block_plane() -- the arithmetic of every other path, so the same bits --
//             instead of through a de-duplicating plane table (a few hundred blocks share almost no triples);
//             LDS, structure of arrays: f (3 x fp32), v' (3 x fp64), a0 (a'.x of a line / c of a plane), and a'.y, a'.z for lines:
//             44 B per block + 16 B per line block;
//   solve     cost evaluations read the blocks from LDS (one round ahead), 28 accumulators per lane -> the butterfly reduction of
//             ll_reg_solve_common.h -> the controller lane (the same lm_* code as every other path, the line-search fit on its
//             wavefront);
//   inliers   loss-corrected L1 values in registers; std::set de-duplication (PCR:155-160) and the rank select by a bitonic sort of
//             the 64-bit keys across the registers of a wavefront (no table, no LDS): equal keys end up adjacent, the distinct values are
//             counted and the wanted rank picked by a prefix sum;
//   epilogue  pose composition and the convergence test (solve_epilogue).
// Sums are grouped differently from the 512-thread forms, so results agree with them to rounding (poses ~1e-12, equal block / iteration
// counts: tests/test_gpu_small.py), not bit for bit; a scan's answer does not depend on its slot or on the batch size within one form.
#include <hip/hip_runtime.h>

#include "ll_reg_query.h"
#include "ll_reg_solve_common.h"

namespace ll {

// R_inc / t_inc of the evaluation point x = {q, t}
#define LL_CTX_DECL_SMALL(x)                                   \
    double R_[9], t_[3];                                        \
    {                                                           \
        const double q_[4] = {(x)[0], (x)[1], (x)[2], (x)[3]};  \
        quat_to_mat(q_, R_);                                    \
        t_[0] = (x)[4];                                         \
        t_[1] = (x)[5];                                         \
        t_[2] = (x)[6];                                         \
    }

#ifdef LL_SOLVE_TIMING
#define SM_T0(var) long long var = clock64()
#define SM_TACC(slot, var)                                      \
    do {                                                        \
        if (threadIdx.x == 0) sh.tcyc[slot] += clock64() - var; \
    } while (0)
#else
#define SM_T0(var)
#define SM_TACC(slot, var)
#endif

struct SmallShared {
    long long tcyc[16];  // LL_SOLVE_TIMING builds: evaluations, controller, L1 pass, sort + select, -, total, census, prune, build (RegState::dbg_cycles)
    LmCtl ctl;
    double red[8][LL_NACC];  // per-wavefront sums of an evaluation (W <= 8)
    double sum[LL_NACC];
    double fit[10];
    double thr;
    double x_start[7];
    int need, n_active, n_corner_avail, n_surf_avail;
    int nL, nA;              // kept line blocks, kept blocks (lines first)
    int n_eval;              // cost evaluations of this launch
    int cnt[16][8];          // census: active blocks per (round, wavefront)
    int isum[8];
    unsigned long long lsum[8];
    int hist[256];           // W >= 4: digit histogram of the radix select
    int sel_digit, sel_rank;
};

// what the kernel reads of the registrar's buffers (ll_device.h RegDev holds ~50 pointers: passed whole, the ones a phase keeps live
// crowd the scalar registers of a kernel that already spills them)
struct SmallArgs {
    RegState *state;
    const int *n_corner, *n_surf;
    const int *order;              // scan of workgroup i (longest first, reg_solve_order_kernel), or nullptr
    const unsigned char *blk_flag0;
    const float4 *blk_f;
    const double *blk_av;
    const int4 *nn;
    const float4 *surf_feat;
    const f4 *map_surf;
    int cap_all, cap_c, feat_stride_s;  // RegDev::cap, cap_c, feat_stride_s
    int cap, capl;                      // LDS capacity in blocks / line blocks
};

// the LDS arrays of one scan (dynamic shared memory): see the header
struct SmallBlocks {
    LL_AS_LDS double *v0, *v1, *v2, *a0, *a1, *a2;
    LL_AS_LDS float *fx, *fy, *fz;
};

template <int W>
__device__ __forceinline__ unsigned long long small_sum_u64(unsigned long long v, SmallShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)__shfl_xor((long long)v, off);
    if (W == 1) return v;
    if (lane == 0) sh.lsum[wave] = v;
    __syncthreads();
    unsigned long long s = 0;
#pragma unroll
    for (int w = 0; w < W; w++) s += sh.lsum[w];
    __syncthreads();
    return s;
}

// compare-exchange of two keys held by the same lane
__device__ __forceinline__ void cx_local(unsigned long long &a, unsigned long long &b, bool up)
{
    const bool sw = (b < a) == up;  // (equal keys: swapping them or not is the same)
    const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}

// Bitonic sort (ascending) of 64 * K keys held K per lane, element g = lane * K + r.  Every step is a fixed data-parallel pattern: pairs
// closer than K live in one lane's registers, the others are exchanged through the crossbar.
template <int K>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&key)[K], int lane)
{
    static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16, "keys per lane: a power of two");
    constexpr int N = 64 * K;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= K) {
                const int lj = j / K;  // partner lane = lane ^ lj, same register
                const bool lower = (lane & lj) == 0;
#pragma unroll
                for (int r = 0; r < K; r++) {
                    const int g = lane * K + r;
                    const bool up = (g & k) == 0;
                    const unsigned long long mine = key[r];
                    const unsigned long long other = (unsigned long long)__shfl_xor((long long)mine, lj);
                    const bool keep_min = lower == up;
                    const bool take = (other < mine) == keep_min;  // one compare, no branch (equal keys: taking the partner's is the same)
                    key[r] = take ? other : mine;
                }
            } else {
#pragma unroll
                for (int r = 0; r < K; r++) {
                    if ((r & j) == 0) {
                        const int g = lane * K + r;
                        const bool up = (g & k) == 0;
                        cx_local(key[r], key[r | j], up);
                    }
                }
            }
        }
    }
}

// workgroup evaluation of cost / g / H at x over the active blocks -> sh.sum
template <int W>
__device__ __forceinline__ void small_eval(const SmallBlocks &B, const double *x, double huber_a, unsigned int act, int nL, int nA, SmallShared &sh)
{
    constexpr int DEBLUR = 0;
    constexpr int NT = 64 * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LL_CTX_DECL_SMALL(x)
    double acc[LL_NACC];
#pragma unroll
    for (int i = 0; i < LL_NACC; i++) acc[i] = 0.0;
    if (nA > 0) {
        // one round ahead: the next block's ten values are fetched from LDS (clamped index, unconditional) while this one is evaluated
        int idx = tid < nA ? tid : nA - 1;
        float nfx = B.fx[idx], nfy = B.fy[idx], nfz = B.fz[idx];
        double nv0 = B.v0[idx], nv1 = B.v1[idx], nv2 = B.v2[idx], na0 = B.a0[idx];
        int il = idx < nL ? idx : 0;
        double na1 = B.a1[il], na2 = B.a2[il];
        for (int r = 0; r * NT < nA; r++) {
            const int cur = r * NT + tid;
            const double f[3] = {(double)nfx, (double)nfy, (double)nfz};
            const double v[3] = {nv0, nv1, nv2};
            const double a[3] = {na0, na1, na2};
            {
                const int nx = cur + NT;
                idx = nx < nA ? nx : nA - 1;
                nfx = B.fx[idx], nfy = B.fy[idx], nfz = B.fz[idx];
                nv0 = B.v0[idx], nv1 = B.v1[idx], nv2 = B.v2[idx], na0 = B.a0[idx];
                il = idx < nL ? idx : 0;
                na1 = B.a1[il], na2 = B.a2[il];
            }
            if (cur < nA && ((act >> r) & 1u)) {
                if (cur < nL) {
                    block_accumulate(BLK_LINE, R_, t_, f, a, v, huber_a, acc);  // ICP:238-380
                } else {
                    const double ap[3] = {a[0], 0.0, 0.0};
                    block_accumulate(BLK_PLANE, R_, t_, f, ap, v, huber_a, acc);
                }
            }
        }
    }
    wave_sum_acc(acc, sh.red[wave], lane);
    __syncthreads();
    if (tid < LL_NACC) {
        double s = sh.red[0][tid];
#pragma unroll
        for (int w = 1; w < W; w++) s += sh.red[w][tid];
        sh.sum[tid] = s;
    }
    __syncthreads();
}

// one ceres::Solve: starts at x0, leaves the result in sh.ctl
template <int W>
__device__ __forceinline__ void small_lm(const SmallBlocks &B, const RegConst &rc, const double *x0, int max_iter, unsigned int act, SmallShared &sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) lm_begin(sh.ctl, x0, max_iter, rc.bound);
    __syncthreads();
    {
        SM_T0(t0);
        small_eval<W>(B, sh.ctl.x, rc.huber_a, act, sh.nL, sh.nA, sh);
        SM_TACC(0, t0);
    }
    {
        SM_T0(t1);
        if (tid == 0) sh.need = lm_init(sh.ctl, sh.sum, sh.n_active);
        __syncthreads();
        SM_TACC(1, t1);
    }
    while (sh.need) {
        SM_T0(t0);
        small_eval<W>(B, sh.ctl.cand, rc.huber_a, act, sh.nL, sh.nA, sh);
        SM_TACC(0, t0);
        SM_T0(t1);
        if (tid < 64) {  // the controller's wavefront: lane 0 steps the controller, all of it fits a line search's interpolant
            const int need = lm_update_wave(sh.ctl, sh.sum, sh.fit, tid);
            if (tid == 0) {
                sh.need = need;
                sh.n_eval++;
            }
        }
        __syncthreads();
        SM_TACC(1, t1);
#ifdef LL_SOLVE_TIMING
        if (tid == 0) sh.tcyc[9] += 1;  // evaluations beyond the first of a solve
#endif
    }
}

// W wavefronts per scan, at most M candidate blocks per thread (64 * W * M >= n_corner + n_surf of every scan of the batch)
template <int W, int M>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W == 4 ? 1 : 2, 8)))
void reg_solve_small_kernel(SmallArgs rd, RegConst rc)
{
    const int cap = rd.cap, capl = rd.capl;
    const f4 *map_surf = rd.map_surf;
    constexpr int NT = 64 * W;
    constexpr int K = W <= 2 ? M * W : 1;  // W <= 2: keys per lane of the sorting wavefront (register sort); W >= 4: the sort runs in LDS
    constexpr int NS = W <= 2 ? 1 : (M * NT <= 256 ? 256 : (M * NT <= 512 ? 512 : (M * NT <= 1024 ? 1024 : 2048)));  // ... over this many keys
    static_assert(M <= 16 && K <= 16 && M * NT <= 2048, "census rounds / sort keys per lane / LDS sort size");
    __shared__ SmallShared sh;
    extern __shared__ double s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = rd.order ? rd.order[blockIdx.x] : (int)blockIdx.x;
    RegState *st = rd.state + b;
    if (st->done) return;
    SmallBlocks B;
    {
        LL_AS_LDS double *p = (LL_AS_LDS double *)s_dyn;
        B.v0 = p, B.v1 = p + cap, B.v2 = p + 2 * cap, B.a0 = p + 3 * cap;
        B.a1 = p + 4 * cap, B.a2 = p + 4 * cap + capl;
        LL_AS_LDS float *q = (LL_AS_LDS float *)(p + 4 * cap + 2 * capl + (W >= 4 ? 2 * NS : 0));
        B.fx = q, B.fy = q + cap, B.fz = q + 2 * cap;
    }
    const int nC = rd.n_corner[b], nS = rd.n_surf[b];
    const int ncand = nC + nS;  // <= NT * M (the host chose M)
    if (ncand > NT * M) {       // (a launch that does not hold the scan must not answer for it: rejected and reported, ll_reg_collect)
        if (tid == 0) {
            st->aborted = 1;
            st->done = 1;
            st->icp_iters += 1;
        }
        return;
    }
    const size_t sb = (size_t)b * rd.cap_all;
    const unsigned char *flag0 = rd.blk_flag0 + sb;
#ifdef LL_SOLVE_TIMING
    if (tid < 16) sh.tcyc[tid] = 0;
    __syncthreads();
#endif
    SM_T0(t_total);
    SM_T0(t_census);

    // ---- census (PCR:325, 425) in the reference's order: candidate c < nC is corner query c, else surface query c - nC -------------
    unsigned int act = 0;  // bit r: candidate r * NT + tid is a kept block
    int na = 0, nca = 0, nsa = 0;
#pragma unroll
    for (int r = 0; r < M; r++) {
        const int c = r * NT + tid;
        const int cc = c < ncand ? c : 0;
        const size_t slot = cc < nC ? (size_t)cc : (size_t)rd.cap_c + (cc - nC);
        const unsigned char fl = (c < ncand) ? gload_u8(flag0 + slot) : (unsigned char)0;
        if (fl & BLK_ACTIVE) {
            act |= 1u << r;
            na++;
        }
        if (fl & 8) {
            if (c < nC) nca++; else nsa++;
        }
    }
    {
        const unsigned long long tot = small_sum_u64<W>((unsigned long long)na | ((unsigned long long)nca << 20) | ((unsigned long long)nsa << 40), sh);
        na = (int)(tot & 0xfffffull);
        nca = (int)((tot >> 20) & 0xfffffull);
        nsa = (int)((tot >> 40) & 0xfffffull);
    }
    if (rc.subsample_seed && na > rc.max_blocks) {  // a13 (PCR:438-458): the random stream is indexed by the block's position in the reference's order
#pragma unroll
        for (int r = 0; r < M; r++) {
            if (!((act >> r) & 1u)) continue;
            if (subsample_drop_block(rc.subsample_seed, st->icp_iters, r * NT + tid, na, rc.max_blocks)) act &= ~(1u << r);
        }
    }
    // dense numbering of the kept blocks in candidate order: position = kept blocks of the earlier rounds + of the earlier wavefronts of
    // this round + of the lower lanes
    int pos[M];
    {
        int mine[M];
#pragma unroll
        for (int r = 0; r < M; r++) {
            const unsigned long long bal = __ballot((act >> r) & 1u);
            mine[r] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) sh.cnt[r][wave] = __popcll(bal);
        }
        __syncthreads();
        int base = 0, kept_lines = 0;
#pragma unroll
        for (int r = 0; r < M; r++) {
            int before = 0, round_total = 0;
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int c = sh.cnt[r][w];
                if (w < wave) before += c;
                round_total += c;
            }
            pos[r] = base + before + mine[r];
            base += round_total;
        }
        // kept line blocks: kept candidates below nC (the lines come first in candidate order)
        {
            int kl = 0;
#pragma unroll
            for (int r = 0; r < M; r++)
                if (((act >> r) & 1u) && r * NT + tid < nC) kl++;
            kept_lines = (int)small_sum_u64<W>((unsigned long long)kl, sh);
        }
        if (tid == 0) {
            sh.n_eval = 0;
            sh.n_active = base;
            sh.nA = base;
            sh.nL = kept_lines;
            sh.n_corner_avail = nca;
            sh.n_surf_avail = nsa;
        }
    }
    SM_TACC(6, t_census);
    SM_T0(t_build);
    // ---- build: the kept blocks' constants -> LDS -------------------------------------------------------------------------------------
    {
        double pose_last[7];
#pragma unroll
        for (int i = 0; i < 7; i++) pose_last[i] = gload_f64(st->pose_last + i);
        const double *av = rd.blk_av + (size_t)b * 6 * rd.cap_all;
        const float4 *sfeat = rd.surf_feat + (size_t)b * rd.feat_stride_s;
#pragma unroll
        for (int r = 0; r < M; r++) {
            const int c = r * NT + tid;
            if (!((act >> r) & 1u)) continue;
            const int p = pos[r];
            if (c < nC) {
                const float4 f = gload_f4(rd.blk_f + sb + c);
                double a0, a1, a2, v0, v1, v2;
                av_load(av, rd.cap_all, c, true, a0, a1, a2, v0, v1, v2);
                B.fx[p] = f.x, B.fy[p] = f.y, B.fz[p] = f.z;
                B.v0[p] = v0, B.v1[p] = v1, B.v2[p] = v2;
                B.a0[p] = a0, B.a1[p] = a1, B.a2[p] = a2;
            } else {
                const int q = c - nC;
                const int4 t = gload_i4(rd.nn + sb + rd.cap_c + q);
                const f4 m0 = gload_pt(map_surf + (unsigned int)t.x), m1 = gload_pt(map_surf + (unsigned int)t.y), m2 = gload_pt(map_surf + (unsigned int)t.z);
                float fx, fy, fz;
                gload_f3(sfeat + q, fx, fy, fz);
                const double pa[3] = {(double)m0.x, (double)m0.y, (double)m0.z};
                const double pb[3] = {(double)m1.x, (double)m1.y, (double)m1.z};
                const double pc[3] = {(double)m2.x, (double)m2.y, (double)m2.z};
                double a_out[3] = {0.0, 0.0, 0.0}, v_out[3] = {0.0, 0.0, 0.0};
                (void)block_plane(pose_last, pa, pb, pc, a_out, v_out);  // (degenerate triples were never flagged active: build_one / the tile kernel)
                B.fx[p] = fx, B.fy[p] = fy, B.fz[p] = fz;
                B.v0[p] = v_out[0], B.v1[p] = v_out[1], B.v2[p] = v_out[2];
                B.a0[p] = a_out[0];
            }
        }
    }
    __syncthreads();
    SM_TACC(8, t_build);
    const int nA = sh.nA, nL = sh.nL;
    // from here on a thread's blocks are the DENSE ones r * NT + tid
    unsigned int live = 0;
#pragma unroll
    for (int r = 0; r < M; r++)
        if (r * NT + tid < nA) live |= 1u << r;

    // ---- prerun solve (PCR:463-474) ---------------------------------------------------------------------------------------------------
    small_lm<W>(B, rc, st->inc, rc.ceres_prerun_times, live, sh);
    int lm_iters = sh.ctl.iteration;

    // ---- loss-corrected L1 values at the prerun result (PCR:476-485), in registers -----------------------------------------------------
    double l1[M];
    SM_T0(t_l1);
    {
        constexpr int DEBLUR = 0;
        LL_CTX_DECL_SMALL(sh.ctl.x)
        double q_last[4];
#pragma unroll
        for (int i = 0; i < 4; i++) q_last[i] = gload_f64(st->pose_last + i);
#pragma unroll
        for (int r = 0; r < M; r++) {
            const int idx = r * NT + tid;
            double v1 = -1.0;
            if ((live >> r) & 1u) {
                const double f[3] = {(double)B.fx[idx], (double)B.fy[idx], (double)B.fz[idx]};
                const double v[3] = {B.v0[idx], B.v1[idx], B.v2[idx]};
                if (idx < nL) {
                    const double a[3] = {B.a0[idx], B.a1[idx], B.a2[idx]};
                    v1 = block_l1(BLK_LINE, R_, t_, f, a, v, rc.huber_a, q_last);
                } else {
                    const double a[3] = {B.a0[idx], 0.0, 0.0};
                    v1 = block_l1(BLK_PLANE, R_, t_, f, a, v, rc.huber_a, q_last);
                }
            }
            l1[r] = v1;
        }
    }
    SM_TACC(2, t_l1);
    SM_T0(t_sort);
    // ---- std::set de-duplication + rank select (PCR:153-161): one bitonic sort on the first wavefront --------------------------------
    if (W <= 2) {
        unsigned long long key[K];  // (the first wavefront's: its own M values per lane, then the other wavefront's)
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double v = l1[k < M ? k : 0];
            key[k] = (k < M && v >= 0.0) ? (unsigned long long)__double_as_longlong(v) : 0xffffffffffffffffull;  // inactive slot or NaN (NaN never enters the set)
        }
        if (W > 1) {
            // the other wavefront hands its keys to the first one, 64 at a time through the 512 bytes of one wavefront's partial sums
            // (nothing is being summed now): which lane ends up with which key does not matter to a sort
            unsigned long long *xch = (unsigned long long *)&sh.red[0][0];
            static_assert(sizeof(sh.red) >= 64 * sizeof(unsigned long long), "exchange buffer");
#pragma unroll
            for (int w = 1; w < (W <= 2 ? W : 1); w++) {
#pragma unroll
                for (int r = 0; r < M; r++) {
                    if (wave == w) xch[lane] = key[r];
                    __syncthreads();
                    if (wave == 0) key[(w * M + r) < K ? (w * M + r) : 0] = xch[lane];
                    __syncthreads();
                }
            }
        }
        if (wave == 0) {
            wave_bitonic_sort<K>(key, lane);
            // element g = lane * K + k is the first of its value iff it differs from element g - 1
            const unsigned long long prev_last = (unsigned long long)__shfl_up((long long)key[K - 1], 1);
            unsigned int first = 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const unsigned long long pv = k == 0 ? prev_last : key[k - 1];
                const bool valid = key[k] != 0xffffffffffffffffull;
                if (valid && ((k == 0 && lane == 0) || key[k] != pv)) first |= 1u << k;
            }
            const int cnt = __popc(first);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if (lane >= off) incl += y;
            }
            const int nu = __shfl(incl, 63);
            int target = (int)(rc.inlier_ratio * (double)nu);  // PCR:160
            if (target > nu - 1) target = nu - 1;
            if (nu == 0) {
                if (lane == 0) sh.thr = rc.inliner_dis;  // empty set: defined deviation (PCR:160 would dereference end())
            } else if (target >= incl - cnt && target < incl) {
                int rk = incl - cnt;
                unsigned long long sel = 0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if ((first >> k) & 1u) {
                        if (rk == target) sel = key[k];
                        rk++;
                    }
                }
                sh.thr = fmax(rc.inliner_dis, __longlong_as_double((long long)sel));  // PCR:485
            }
        }
    } else {
        // four / eight wavefronts: no sort.  (A bitonic sort of the keys in LDS -- 66 barrier steps for 2 048 keys -- was a third of a
        // launch of the mapping loop's scans: 44 k of 149 k cycles.)  std::set semantics by an exact LDS hash table -- a key is inserted
        // with one 64-bit compare-and-swap; whoever finds its own key already there is a duplicate, exactly one lane per distinct value is
        // not, whatever the order of the atomics -- then a most-significant-digit-first radix select over the distinct keys, 8 bits per
        // pass: a 256-bin LDS histogram, one wavefront finds the digit that holds the wanted rank.
        LL_AS_LDS unsigned long long *tab = (LL_AS_LDS unsigned long long *)(B.a2 + capl);  // [2 * NS] slots, behind the line arrays
        constexpr unsigned int TS = 2u * NS;
        for (int e = tid; e < (int)TS; e += NT) tab[e] = 0xffffffffffffffffull;
        __syncthreads();
        unsigned int uniq = 0;  // bit r: l1[r] is the first of its value
#pragma unroll
        for (int r = 0; r < M; r++) {
            const double v = l1[r];
            if (!(v >= 0.0)) continue;  // inactive slot or NaN (NaN never enters the set)
            const unsigned long long key = (unsigned long long)__double_as_longlong(v);
            unsigned int hsh = (unsigned int)key * 0x9E3779B1u;
            hsh ^= hsh >> 15;
            hsh += (unsigned int)(key >> 32) * 0x85EBCA77u;
            hsh ^= hsh >> 13;
            unsigned int slot = hsh & (TS - 1u);
            for (;;) {  // (at most half of the slots are ever taken: the probe ends)
                const unsigned long long old = atomicCAS((unsigned long long *)&tab[slot], 0xffffffffffffffffull, key);
                if (old == 0xffffffffffffffffull) {
                    uniq |= 1u << r;
                    break;
                }
                if (old == key) break;
                slot = (slot + 1u) & (TS - 1u);
            }
        }
        const int nu = (int)small_sum_u64<W>((unsigned long long)__popc(uniq), sh);  // (its barriers: every insert has landed)
        int target = (int)(rc.inlier_ratio * (double)nu);  // PCR:160
        if (target > nu - 1) target = nu - 1;
        if (nu == 0) {
            if (tid == 0) sh.thr = rc.inliner_dis;  // empty set: defined deviation (PCR:160 would dereference end())
        } else {
            unsigned long long prefix = 0ull;
            int rank = target;
            for (int pass = 0; pass < 8; pass++) {
                const int shift = 56 - 8 * pass;
                if (tid < 256) sh.hist[tid] = 0;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < M; r++) {
                    if (!((uniq >> r) & 1u)) continue;
                    const unsigned long long key = (unsigned long long)__double_as_longlong(l1[r]);
                    if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&sh.hist[(int)((key >> shift) & 255ull)], 1);
                }
                __syncthreads();
                if (wave == 0) {  // lane l sums bins 4 l .. 4 l + 3; the lane whose range holds the rank walks its four bins
                    const int b0 = sh.hist[4 * lane], b1 = sh.hist[4 * lane + 1], b2 = sh.hist[4 * lane + 2], b3 = sh.hist[4 * lane + 3];
                    const int part = b0 + b1 + b2 + b3;
                    int incl = part;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int y = __shfl_up(incl, off);
                        if (lane >= off) incl += y;
                    }
                    const int below = incl - part;
                    if (rank >= below && rank < incl) {
                        int d = 4 * lane, cum = below;
                        if (cum + b0 <= rank) {
                            cum += b0;
                            d++;
                            if (cum + b1 <= rank) {
                                cum += b1;
                                d++;
                                if (cum + b2 <= rank) {
                                    cum += b2;
                                    d++;
                                }
                            }
                        }
                        sh.sel_digit = d;
                        sh.sel_rank = rank - cum;
                    }
                }
                __syncthreads();
                prefix = (prefix << 8) | (unsigned long long)sh.sel_digit;
                rank = sh.sel_rank;
            }
            if (tid == 0) sh.thr = fmax(rc.inliner_dis, __longlong_as_double((long long)prefix));  // PCR:485
        }
    }
    __syncthreads();
    SM_TACC(3, t_sort);
    SM_T0(t_prune);
    // ---- prune (PCR:487-499) -----------------------------------------------------------------------------------------------------------
    {
        const double thr = sh.thr;
        int keep = 0;
#pragma unroll
        for (int r = 0; r < M; r++) {
            if (!((live >> r) & 1u)) continue;
            if (l1[r] > thr)
                live &= ~(1u << r);
            else
                keep++;
        }
        keep = (int)small_sum_u64<W>((unsigned long long)keep, sh);
        if (tid == 0) sh.n_active = keep;
        if (tid < 7) sh.x_start[tid] = sh.ctl.x[tid];
        __syncthreads();
    }
    SM_TACC(7, t_prune);
    // ---- final solve (PCR:501-508) -------------------------------------------------------------------------------------------------------
    small_lm<W>(B, rc, sh.x_start, rc.ceres_max_iterations, live, sh);
    lm_iters += sh.ctl.iteration;
    solve_epilogue(rc, st, sh, lm_iters);
    if (tid == 0) st->last_work = sh.n_eval;  // next launch: the scans that worked longest start first
#ifdef LL_SOLVE_TIMING
    SM_TACC(5, t_total);
    if (tid == 0)
        for (int i = 0; i < 16; i++) st->dbg_cycles[i] += sh.tcyc[i];
#endif
}

// Longest first: the scans of a batch differ five-fold in the work of a launch (the ones whose step runs into the bound on t_inc contract a
// line search: dozens of extra evaluations), a launch of more scans than the device holds at once ends when its last scan does, and a
// scan's work changes little from one ICP iteration to the next.  One workgroup orders the scans by the evaluations of their previous
// launch, descending (counting sort: the first launch of a registration runs in scan order).  Only the order of dispatch depends on it, no result does.
#define SO_THREADS 1024
#define SO_BINS 256
__global__ __launch_bounds__(SO_THREADS) void reg_solve_order_kernel(const RegState *state, int n_scans, int *order)
{
    __shared__ int hist[SO_BINS], start[SO_BINS];
    const int tid = threadIdx.x;
    for (int e = tid; e < SO_BINS; e += SO_THREADS) hist[e] = 0;
    __syncthreads();
    for (int b = tid; b < n_scans; b += SO_THREADS) {
        int w = state[b].last_work;
        w = w < 0 ? 0 : (w > SO_BINS - 1 ? SO_BINS - 1 : w);
        atomicAdd(&hist[SO_BINS - 1 - w], 1);  // bin 0 = the most work
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int e = 0; e < SO_BINS; e++) {
            start[e] = run;
            run += hist[e];
        }
    }
    __syncthreads();
    for (int b = tid; b < n_scans; b += SO_THREADS) {  // (the order inside a bin is whatever the atomics hand out: it decides nothing but timing)
        int w = state[b].last_work;
        w = w < 0 ? 0 : (w > SO_BINS - 1 ? SO_BINS - 1 : w);
        order[atomicAdd(&start[SO_BINS - 1 - w], 1)] = b;
    }
}

// dynamic LDS of one scan
static size_t small_lds_bytes(int W, int M, int cap, int capl)
{
    const int keys = M * 64 * W;
    const int ns = W <= 2 ? 0 : (keys <= 256 ? 256 : (keys <= 512 ? 512 : (keys <= 1024 ? 1024 : 2048)));  // (NS of the kernel)
    return (size_t)cap * (4 * 8 + 3 * 4) + (size_t)capl * 16 + (size_t)ns * 16;  // (W >= 4: a hash table of 2 NS slots)
}

template <int W, int M>
static void launch_small(const SmallArgs &a, const RegConst &rc, int n_scans, hipStream_t s)
{
    const size_t lds = small_lds_bytes(W, M, a.cap, a.capl);
    static bool attr_set = false;
    if (!attr_set) {  // dynamic LDS beyond the default 64 KB: everything the CU has left beside the kernel's static block
        hipFuncAttributes fa;
        int room = 160 * 1024 - 8 * 1024;
        if (hipFuncGetAttributes(&fa, (const void *)reg_solve_small_kernel<W, M>) == hipSuccess) room = 160 * 1024 - (int)fa.sharedSizeBytes;
        if (hipFuncSetAttribute((const void *)reg_solve_small_kernel<W, M>, hipFuncAttributeMaxDynamicSharedMemorySize, room) != hipSuccess)
            (void)hipGetLastError();  // (not sticky: a launch that needs more than the default then fails on its own)
        attr_set = true;
    }
    hipLaunchKernelGGL((reg_solve_small_kernel<W, M>), dim3(n_scans), dim3(64 * W), lds, s, a, rc);
}

bool reg_solve_small_eligible(const RegConst &rc, int max_nc, int max_ns)
{
    return !rc.if_motion_deblur && !rc.force_general && !rc.no_small_solver && max_nc + max_ns > 0 &&
           max_nc + max_ns <= LL_SMALL_MAX_BLOCKS && max_nc <= 1024;  // (line blocks take 16 B more of LDS each)
}

// wavefronts per scan: four for batches that leave CUs idle anyway (latency); for large batches as many as keep two wavefronts per SIMD
// busy at the number of scans whose blocks fit a CU's LDS together (256 VGPRs per wavefront: eight wavefronts per CU)
int reg_solve_small_waves(const RegConst &rc, int n_scans, int max_nc, int max_ns)
{
    const int total = max_nc + max_ns;
    if (total > 1024) return 8;  // (the only form that holds them)
    if (rc.small_waves) return rc.small_waves;
    if (n_scans < LL_SMALL_W1_MIN_SCANS) return 4;
    const int cap = (total + 63) / 64 * 64, capl = (max_nc + 63) / 64 * 64 + 64;
    const size_t per_scan = small_lds_bytes(1, 1, cap, capl) + sizeof(SmallShared) + 512;
    const int per_cu = (int)((size_t)(160 * 1024) / per_scan);
    return per_cu >= 8 ? 1 : (per_cu >= 4 ? 2 : 4);
}

void launch_reg_solve_small(const RegDev &rd, const RegConst &rc, const Grid &gs, int n_scans, int max_nc, int max_ns, int iter, hipStream_t s)
{
    const int total = max_nc + max_ns;
    SmallArgs a;
    a.state = rd.state, a.n_corner = rd.n_corner, a.n_surf = rd.n_surf, a.blk_flag0 = rd.blk_flag0, a.blk_f = rd.blk_f, a.blk_av = rd.blk_av;
    a.nn = rd.nn, a.surf_feat = rd.surf_feat, a.map_surf = gs.pts, a.cap_all = rd.cap, a.cap_c = rd.cap_c, a.feat_stride_s = rd.feat_stride_s;
    a.cap = (total + 63) / 64 * 64, a.capl = (max_nc + 63) / 64 * 64 + 64;  // (+64: the clamped line index of a scan without lines stays inside)
    const int W = reg_solve_small_waves(rc, n_scans, max_nc, max_ns);
    // more scans than the device runs at once: longest first (from the second launch of a registration on)
    a.order = nullptr;
    if (rd.solve_order && !rc.no_solve_order && n_scans >= LL_SMALL_ORDER_MIN_SCANS && n_scans <= LL_SMALL_ORDER_MAX_SCANS && iter > 0) {
        hipLaunchKernelGGL(reg_solve_order_kernel, dim3(1), dim3(SO_THREADS), 0, s, rd.state, n_scans, rd.solve_order);
        a.order = rd.solve_order;
    }
    const int cls = total <= 256 ? 0 : (total <= 512 ? 1 : 2);  // 64 * W * M >= total (more than 1024: eight wavefronts, M = 4)
    if (W == 1) {
        if (cls == 0) launch_small<1, 4>(a, rc, n_scans, s);
        else if (cls == 1) launch_small<1, 8>(a, rc, n_scans, s);
        else launch_small<1, 16>(a, rc, n_scans, s);
    } else if (W == 2) {
        if (cls == 0) launch_small<2, 2>(a, rc, n_scans, s);
        else if (cls == 1) launch_small<2, 4>(a, rc, n_scans, s);
        else launch_small<2, 8>(a, rc, n_scans, s);
    } else if (W == 4) {
        if (cls == 0) launch_small<4, 1>(a, rc, n_scans, s);
        else if (cls == 1) launch_small<4, 2>(a, rc, n_scans, s);
        else launch_small<4, 4>(a, rc, n_scans, s);
    } else {
        launch_small<8, 4>(a, rc, n_scans, s);
    }
}

}  // namespace ll
