// ll_fe_kernels.hip -- HIP kernels (gfx950, wave64) of the Livox feature extractor.
//
//   K1/K2 fe_point_kernel  : per-point projection, masks, curvature / view-angle labels
//                            (livox_feature_extractor.hpp:474-526 + :343-358 + :322-341 + :361-455)
//   K4    fe_split_kernel  : one workgroup per scan: (0,0,0)-point inheritance, petal split with the 50-point
//                            hysteresis, petal angles, split_laser_scan bookkeeping, piece-wise windows
//                            (livox_feature_extractor.hpp:493-512,529-606,657-719; laser_feature_extractor.hpp:305-323)
//   K3    fe_select_kernel : get_features predicate + order-preserving stream compaction
//                            (livox_feature_extractor.hpp:219-272)
//
// Memory-bound, no MFMA: 16 B/point in (one float4 load per lane, coalesced), ~40 B/point out as SoA planes.
// The 5-point stencil is staged through LDS (256 points + halo 2).  Compiled with -ffp-contract=off: the
// label/index sets must be bit-identical to the reference's fp32 evaluation order.
#include <hip/hip_runtime.h>

#include "ll_device.h"
#include "ll_fe_core.h"

namespace ll {

#define FE_TILE 256
#define FE_HALO 2

__global__ __launch_bounds__(FE_TILE) void fe_point_kernel(FeDev fb, FeConst fc)
{
    const int b = blockIdx.y;
    const int n = fb.npts[b];
    const int tile0 = blockIdx.x * FE_TILE;
    if (tile0 >= n) return;
    const size_t sb = (size_t)b * fb.stride;
    const float4 *__restrict__ pts = fb.xyzi + sb;

    __shared__ float s_x[FE_TILE + 2 * FE_HALO], s_y[FE_TILE + 2 * FE_HALO], s_z[FE_TILE + 2 * FE_HALO];
    __shared__ float s_depth[FE_TILE + 2 * FE_HALO];
    __shared__ int s_type[FE_TILE + 2 * FE_HALO];  // self type | edge<<16

    const int tid = threadIdx.x;
    for (int l = tid; l < FE_TILE + 2 * FE_HALO; l += FE_TILE) {
        const int k = tile0 - FE_HALO + l;
        float x = 0.f, y = 0.f, z = 0.f, dep = 0.f;
        int ty = 0;
        if (k >= 0 && k < n) {
            const float4 p = pts[k];
            const PointOwn o = point_own(p.x, p.y, p.z, p.w, k, fc);
            x = p.x;
            y = p.y;
            z = p.z;
            dep = o.depth_sq2;
            ty = o.type_self | (o.edge << 16);
            if (l >= FE_HALO && l < FE_TILE + FE_HALO) {
                // this lane owns point k: write its own planes
                fb.polar2[sb + k] = o.polar_sq2;
                fb.img[sb + k] = make_float2(o.img_y, o.img_z);
                fb.depth2[sb + k] = o.depth_sq2;
                fb.flags[sb + k] = (signed char)(o.defines | (o.reached << 1));
                fb.tstamp[sb + k] = point_time_stamp(fb.time0[b], k, fc.time_internal_pts);
            }
        }
        s_x[l] = x;
        s_y[l] = y;
        s_z[l] = z;
        s_depth[l] = dep;
        s_type[l] = ty;
    }
    __syncthreads();

    const int i = tile0 + tid;
    if (i >= n) return;
    const int l = tid + FE_HALO;
    // neighbour smear of the circle-edge mask: point k past the edge masks k-2, k-1, k+1 (LFE:330-339)
    const int smear = (s_type[l - 1] | s_type[l + 1] | s_type[l + 2]) >> 16;
    const int type = (s_type[l] & 0xffff) | (smear ? PT_CIRCLE_EDGE : 0);

    LabelOut lo;
    lo.label = 0;
    lo.curvature = 0.f;
    lo.view_angle = 0.f;
    lo.ambiguous = 0;
    if (n >= 5 && i >= 2 && i < n - 2) {
        float p[5][3];
        int t[5];
        float d[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            p[k][0] = s_x[l - 2 + k];
            p[k][1] = s_y[l - 2 + k];
            p[k][2] = s_z[l - 2 + k];
            t[k] = s_type[l - 2 + k] & 0xffff;
            d[k] = s_depth[l - 2 + k];
        }
        lo = point_label(p, t, d, fc);
    }
    fb.type[sb + i] = type;
    fb.label[sb + i] = lo.label;
    fb.curv[sb + i] = lo.curvature;
    fb.view[sb + i] = lo.view_angle;
    if (lo.ambiguous) {
        const int slot = atomicAdd(fb.n_ambig, 1);
        if (slot < fb.ambig_cap) fb.ambig_list[slot] = make_int2(b, i);
    }
}

// ---------------------------------------------------------------------------------------------------------
// One workgroup (1024 threads = 16 waves) per scan.
#define SP_THREADS 1024
#define SP_WAVES (SP_THREADS / 64)
#define SP_MAX_SPLITS 4096  // LDS copy of the split list (n/50 + 8 entries; covers n <= 200k)
#define SP_ITEMS 4          // consecutive points per thread in the two scan passes of fe_split_kernel
#define SP_CAND_CHUNK 2048  // split candidates staged in LDS for the sequential hysteresis walk

__device__ __forceinline__ int wave_incl_max(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        if (lane >= off) v = max(v, t);
    }
    return v;
}

// ordered compaction helper: returns this thread's output slot (or -1) and advances *base (shared) by the
// number of set predicates in the workgroup.  All threads must call it.
__device__ __forceinline__ int block_compact_slot(bool pred, int *s_wave_cnt, int *s_base, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(pred);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = *s_base;
    for (int w = 0; w < wave; w++) off += s_wave_cnt[w];
    const int slot = pred ? off + before : -1;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < SP_WAVES; w++) tot += s_wave_cnt[w];
        *s_base += tot;
    }
    __syncthreads();
    return slot;
}

__device__ __forceinline__ int dir_of(const float *polar2, const signed char *flags, int i)
{
    // polar_direction of point i (LFE:529-541): only points that reach the split code get one
    if (i < 1 || !((flags[i] >> 1) & 1)) return 0;
    const float inc = polar2[i] - polar2[i - 1];
    return inc > 0.f ? 1 : (inc < 0.f ? -1 : 0);
}

__global__ __launch_bounds__(SP_THREADS) void fe_split_kernel(FeDev fb, int pieces)
{
    const int b = blockIdx.x;
    const int n = fb.npts[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t sb = (size_t)b * fb.stride;
    float *polar2 = fb.polar2 + sb;
    float2 *img = fb.img + sb;
    const signed char *flags = fb.flags + sb;
    const int *type = fb.type + sb;
    const float4 *pts = fb.xyzi + sb;
    float *polar_angle = fb.polar_angle + sb;
    int *cand = fb.cand + sb;
    int *splits = fb.split_idx + (size_t)b * fb.split_cap;
    int *pet_first = fb.petal_first + (size_t)b * fb.split_cap;
    int *pet_last = fb.petal_last + (size_t)b * fb.split_cap;
    float *run_angle = fb.run_angle + (size_t)b * fb.split_cap;
    FeScanInfo *info = fb.info + b;

    __shared__ int s_wave[SP_WAVES];
    __shared__ int s_carry, s_base, s_nsplit;
    __shared__ int s_split[SP_MAX_SPLITS];
    __shared__ int s_min[16];
    // Everything a single lane walks sequentially lives in LDS: one lane chasing ~100 dependent global loads per phase
    // (the candidate list, then the per-run angle / first / last arrays) was most of this kernel's 294 us.
    __shared__ int s_cand[SP_CAND_CHUNK];
    __shared__ float s_angle[SP_MAX_SPLITS];
    __shared__ int s_first[SP_MAX_SPLITS], s_last[SP_MAX_SPLITS];
    __shared__ int s_hyst[4];  // ns, n_edge, n_zero, last: the hysteresis state carried across candidate chunks

    // ---- phase A: inheritance of pt_2d_img / polar_dis_sq2 by x==0 points (LFE:507-508) -------------------
    // SP_ITEMS consecutive points per thread: a quarter of the workgroup-wide scan steps (and barriers) per scan
    if (tid == 0) s_carry = -1;
    __syncthreads();
    for (int base = 0; base < n; base += SP_THREADS * SP_ITEMS) {
        const int i0 = base + tid * SP_ITEMS;
        bool defines[SP_ITEMS];
        int m[SP_ITEMS];
        int run = -1;
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++) {
            const int i = i0 + k;
            defines[k] = (i < n) && (flags[i < n ? i : 0] & 1);
            run = max(run, defines[k] ? i : -1);
            m[k] = run;
        }
        const int incl = wave_incl_max(run, lane);
        int excl = __shfl_up(incl, 1);
        if (lane == 0) excl = -1;
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int pre = max(s_carry, excl);
        for (int w = 0; w < wave; w++) pre = max(pre, s_wave[w]);
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++) {
            const int i = i0 + k, v = max(m[k], pre);
            if (i < n && !defines[k] && v >= 0) {
                polar2[i] = polar2[v];
                img[i] = img[v];
            }
        }
        __syncthreads();
        if (tid == SP_THREADS - 1) s_carry = max(pre, run);
        __syncthreads();
    }

    // ---- phase B: direction changes -> candidate list (ascending) ----------------------------------------
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += SP_THREADS * SP_ITEMS) {
        const int i0 = base + tid * SP_ITEMS;
        int kind[SP_ITEMS];
        int cnt = 0;
        int dp = dir_of(polar2, flags, (i0 - 1 < n) ? i0 - 1 : 0);  // dir_of() is 0 for i < 1
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++) {
            const int i = i0 + k;
            kind[k] = 0;
            if (i >= 1 && i < n) {
                const int di = dir_of(polar2, flags, i);
                if (di == -1 && dp == 1) kind[k] = 1;        // edge split candidate, LFE:543
                else if (di == 1 && dp == -1) kind[k] = 2;   // zero split candidate, LFE:553
                dp = di;
            }
            cnt += kind[k] != 0;
        }
        int incl = cnt;  // wave inclusive sum of the per-thread counts
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int slot = s_base + incl - cnt;
        for (int w = 0; w < wave; w++) slot += s_wave[w];
#pragma unroll
        for (int k = 0; k < SP_ITEMS; k++)
            if (kind[k]) cand[slot++] = ((i0 + k) << 2) | kind[k];
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < SP_WAVES; w++) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    const int n_cand = s_base;
    __syncthreads();

    // ---- phase C: 50-point hysteresis, sequential over candidates (LFE:545-562) ----------------------------
    if (tid < 4) s_hyst[tid] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_cand; c0 += SP_CAND_CHUNK) {
        const int m = min(SP_CAND_CHUNK, n_cand - c0);
        for (int k = tid; k < m; k += SP_THREADS) s_cand[k] = cand[c0 + k];
        __syncthreads();
        if (tid == 0) {
            int ns = s_hyst[0], n_edge = s_hyst[1], n_zero = s_hyst[2], last = s_hyst[3];
            for (int c = 0; c < m; c++) {
                const int e = s_cand[c];
                const int i = e >> 2, kind = e & 3;
                bool take;
                if (kind == 1)
                    take = (n_edge == 0) || (i - last > 50);
                else
                    take = (n_zero == 0) || (i - last > 50);
                if (take && ns < fb.split_cap - 1) {
                    if (ns < SP_MAX_SPLITS) s_split[ns] = i;
                    splits[ns++] = i;
                    last = i;
                    if (kind == 1) n_edge++; else n_zero++;
                }
            }
            s_hyst[0] = ns, s_hyst[1] = n_edge, s_hyst[2] = n_zero, s_hyst[3] = last;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int ns = s_hyst[0];
        if (ns < SP_MAX_SPLITS) s_split[ns] = n - 1;
        splits[ns++] = n - 1;  // LFE:565
        s_nsplit = ns;
    }
    __syncthreads();
    const int ns = s_nsplit;

    const bool enough = (ns >= 6) && (ns <= SP_MAX_SPLITS);  // LFE:572
    const int n_runs = enough ? ns - 1 : 0;

    // ---- phase D: petal angles (LFE:575-604) and split_laser_scan bookkeeping (LFE:657-719) ----------------
    for (int v = tid; v < n_runs; v += SP_THREADS) {
        const int s0 = s_split[v], s1 = s_split[v + 1];
        const int internal = s1 - s0;
        int ai;
        if (polar2[s1] > 10000.f)
            ai = s1 - (int)(internal * 0.20);
        else
            ai = s1 - (int)(internal * 0.80);
        const float2 im = img[ai];
        float ang = (float)((double)atan2f(im.y, im.x) * 57.3);
        ang = (float)((double)ang + 180.0);
        run_angle[v] = ang;
        s_angle[v] = ang;
        // first / last surviving point of the run: idx in (s0, s1], run 0 starts at 0.  Nearly every point survives, so
        // the two searches stop after a step or two (a run with no survivor is walked once in full)
        const int lo = (v == 0) ? 0 : s0 + 1;
        int first = -1, lastp = -1;
        for (int i = lo; i <= s1; i++)
            if ((type[i] & (PT_000 | PT_TOO_NEAR | PT_NAN)) == 0) {
                first = i;
                break;
            }
        if (first >= 0)
            for (int i = s1; i >= first; i--)
                if ((type[i] & (PT_000 | PT_TOO_NEAR | PT_NAN)) == 0) {
                    lastp = i;
                    break;
                }
        s_first[v] = first;
        s_last[v] = lastp;
    }
    __syncthreads();
    // per-point petal angle
    for (int i = tid; i < n; i += SP_THREADS) {
        float a = 0.f;
        if (enough) {
            // run v: split[v] < i <= split[v+1] (v >= 1), run 0: i <= split[1]; capped at ns-2
            int lo = 0, hi = ns - 2;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (i <= s_split[mid + 1]) hi = mid; else lo = mid + 1;
            }
            a = s_angle[lo];
        }
        polar_angle[i] = a;
    }
    __syncthreads();
    if (tid == 0) {
        // merge runs whose angle compares equal (a new petal starts only where scan_id_index changes, LFE:672),
        // drop the last petal (LFE:681), drop empty petals (LFE:713-716); compact in place (in LDS).
        int out = 0;
        if (enough) {
            // a run with no points at all (only possible when the closing entry n-1 duplicates the last split)
            // never produces a scan_id_index change: it does not exist for split_laser_scan
            int nr = n_runs;
            while (nr > 1 && s_split[nr - 1] == s_split[nr]) nr--;
            int v = 0;
            while (v < nr) {
                int first = s_first[v], lastp = s_last[v];
                int w = v + 1;
                while (w < nr && s_angle[w] == s_angle[v]) {
                    if (s_first[w] >= 0) {
                        if (first < 0) first = s_first[w];
                        lastp = s_last[w];
                    }
                    w++;
                }
                const bool is_last = (w >= nr);
                if (!is_last && first >= 0) {
                    s_first[out] = first;  // out <= v: never overwrites an entry still to be read
                    s_last[out] = lastp;
                    out++;
                }
                v = w;
            }
        }
        info->n_split = ns;
        info->clutter_size = enough ? ns - 1 : 0;  // LFE:606 / :572
        info->n_petal_clouds = out;
        for (int p = 0; p < LL_MAX_PIECES; p++) {
            info->piece_start[p] = 0.f;
            info->piece_end[p] = 0.f;
        }
        s_base = out;
    }
    __syncthreads();
    for (int v = tid; v < s_base; v += SP_THREADS) {  // the petal table callers read (ll_fe_splits)
        pet_first[v] = s_first[v];
        pet_last[v] = s_last[v];
    }
    // ---- piece-wise windows (laser_feature_extractor.hpp:305-323); the caller only gets here when S > 5 -----
    const int S = s_base;
    if (S > 5 && pieces >= 1 && pieces <= LL_MAX_PIECES && pieces <= S) {
        // find_pt_info(...) returns the FIRST inserted point with the same xyz (unordered_map insert, LFE:478)
        if (tid < 2 * pieces) s_min[tid] = 0x7fffffff;
        __syncthreads();
        // one pass over the points for all 2 * pieces targets; no early exit, so the loads of a thread's iterations overlap
        __shared__ float4 s_tp[2 * LL_MAX_PIECES];
        __shared__ int s_tgt[2 * LL_MAX_PIECES];
        if (tid < 2 * pieces) {
            const int pc = tid >> 1;
            const int start_scans = (S * pc) / pieces, end_scans = (S * (pc + 1)) / pieces - 1;
            const int target = (tid & 1) ? s_last[end_scans] : s_first[start_scans];
            s_tgt[tid] = target;
            s_tp[tid] = pts[target];
        }
        __syncthreads();
        int far = 0;
        for (int q = 0; q < 2 * pieces; q++) far = max(far, s_tgt[q]);
        for (int j = tid; j <= far; j += SP_THREADS) {
            const float4 p = pts[j];
            for (int q = 0; q < 2 * pieces; q++) {
                const float4 tp = s_tp[q];
                if (j <= s_tgt[q] && p.x == tp.x && p.y == tp.y && p.z == tp.z) atomicMin(&s_min[q], j);
            }
        }
        __syncthreads();
        if (tid < pieces) {
            info->piece_start[tid] = ((float)s_min[2 * tid]) / n;     // float / size_t -> float
            info->piece_end[tid] = ((float)s_min[2 * tid + 1]) / n;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS) void fe_select_kernel(FeDev fb, int piece, float min_blur, float max_blur)
{
    const int b = blockIdx.x;
    const int n = fb.npts[b];
    const int tid = threadIdx.x;
    const size_t sb = (size_t)b * fb.stride;
    const int *type = fb.type + sb, *label = fb.label + sb;
    const float *depth2 = fb.depth2 + sb, *tstamp = fb.tstamp + sb;
    const float4 *pts = fb.xyzi + sb;
    __shared__ int s_wave[SP_WAVES];
    __shared__ int s_nc, s_ns, s_nf;
    if (piece >= 0) {
        min_blur = fb.info[b].piece_start[piece];
        max_blur = fb.info[b].piece_end[piece];
    }
    const float maximum_idx = max_blur * n;  // LFE:227-228
    const float minimum_idx = min_blur * n;
    if (tid == 0) {
        s_nc = 0;
        s_ns = 0;
        s_nf = 0;
    }
    __syncthreads();
    for (int base = 0; base < n; base += SP_THREADS) {
        const int i = base + tid;
        int sel = 0;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            sel = select_point(i, type[i], label[i], depth2[i], minimum_idx, maximum_idx);
            if (sel & 3) {
                p = pts[i];
                p.w = tstamp[i];  // intensity := time stamp, LFE:246,255
            }
        }
        const int sc = block_compact_slot((sel & 1) != 0, s_wave, &s_nc, tid);
        if (sc >= 0) {
            fb.corner_idx[sb + sc] = i;
            fb.corner_feat[sb + sc] = p;
        }
        const int ss = block_compact_slot((sel & 2) != 0, s_wave, &s_ns, tid);
        if (ss >= 0) {
            fb.surf_idx[sb + ss] = i;
            fb.surf_feat[sb + ss] = p;
        }
        const int sf = block_compact_slot((sel & 4) != 0, s_wave, &s_nf, tid);
        if (sf >= 0 && fb.full_idx) fb.full_idx[sb + sf] = i;
    }
    if (tid == 0) {
        fb.n_corner[b] = s_nc;
        fb.n_surf[b] = s_ns;
        fb.n_full[b] = s_nf;
    }
}

// ---- launch wrappers (called from ll_api.hip) --------------------------------------------------------------
void launch_fe_point(const FeDev &fb, const FeConst &fc, int n_scans, int max_n, hipStream_t s)
{
    dim3 grid((max_n + FE_TILE - 1) / FE_TILE, n_scans);
    hipLaunchKernelGGL(fe_point_kernel, grid, dim3(FE_TILE), 0, s, fb, fc);
}
void launch_fe_split(const FeDev &fb, int pieces, int n_scans, hipStream_t s)
{
    hipLaunchKernelGGL(fe_split_kernel, dim3(n_scans), dim3(SP_THREADS), 0, s, fb, pieces);
}
void launch_fe_select(const FeDev &fb, int n_scans, int piece, float min_blur, float max_blur, hipStream_t s)
{
    hipLaunchKernelGGL(fe_select_kernel, dim3(n_scans), dim3(SP_THREADS), 0, s, fb, piece, min_blur, max_blur);
}

}  // namespace ll
