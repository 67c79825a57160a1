// ll_knn_coop.h -- the exact 5-NN search of ll_knn_core.h run by ONE WAVEFRONT per query (device only, wave64).
//
// Why: a single lane's search is a chain of dependent loads -- per x-run the two cell_start words, then the candidates four at
// a time, the next run only after the current one has tightened the pruning radius.  A surface query on the C2 map walks ~5 runs
// and ~22 candidates (~15 round trips); a corner query (1.45 m cells, sparse map) ~13 runs and ~110 candidates, and one in four
// goes on into the Chebyshev rings: 100+ round trips.  With tens of thousands of queries per launch other wavefronts hide that;
// the late ICP iterations search a few dozen corner queries per scan and a single scan (the sequential mapping loop) has 340 in
// total -- the launch then lasts exactly as long as its longest chain (49 us at B = 1, ~100 us at B = 256: profiles/r03b).
// Here the 64 lanes of a wavefront split the candidates of ONE query: all nine runs of the 3x3x3 block are looked up at once
// (7 lanes per run), every lane keeps its own ordered top five, and a selection merge (five rounds of a wavefront-wide minimum
// over the lanes' list heads) yields the answer -- a handful of round trips, ~1.5 k instructions.  Rings and the cube sweep hand
// one row segment to each lane.  Costs ~20x the instructions of the per-lane search, so only where latency is the bound.
//
// Same contract as knn5_search: the five smallest (d2, original index) among the points with d2 < max_d2, identical arithmetic
// per candidate (dist2_xyz), and VALID reuse bounds lb2 / out2.  The bounds are not the serial ones bit for bit: nothing is
// pruned inside the 3x3x3 block, so lb2 there is the true 6th-nearest distance instead of a box distance -- the displacement
// budgets only get larger, the neighbour lists are the same.
#pragma once
#include "ll_knn_core.h"

namespace ll {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long coop_dpp_u64(unsigned long long v)
{
    // lanes that receive nothing (row start, masked rows) get the `old` operand: all ones, the neutral element of min
    const int lo = __builtin_amdgcn_update_dpp(-1, (int)(unsigned int)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(-1, (int)(unsigned int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
__device__ __forceinline__ unsigned long long coop_umin(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
// minimum over the wavefront (all 64 lanes active), the same value in every lane
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    v = coop_umin(v, coop_dpp_u64<0x111, 0xf>(v));  // row_shr:1
    v = coop_umin(v, coop_dpp_u64<0x112, 0xf>(v));  // row_shr:2
    v = coop_umin(v, coop_dpp_u64<0x114, 0xf>(v));  // row_shr:4
    v = coop_umin(v, coop_dpp_u64<0x118, 0xf>(v));  // row_shr:8   -> lane 15 of every row
    v = coop_umin(v, coop_dpp_u64<0x142, 0xa>(v));  // row_bcast:15 -> lanes 31 and 63
    v = coop_umin(v, coop_dpp_u64<0x143, 0xc>(v));  // row_bcast:31 -> lane 63
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float coop_dpp_f32_or_inf(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0x7f800000, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_f32(float v)
{
    v = fminf(v, coop_dpp_f32_or_inf<0x111, 0xf>(v));
    v = fminf(v, coop_dpp_f32_or_inf<0x112, 0xf>(v));
    v = fminf(v, coop_dpp_f32_or_inf<0x114, 0xf>(v));
    v = fminf(v, coop_dpp_f32_or_inf<0x118, 0xf>(v));
    v = fminf(v, coop_dpp_f32_or_inf<0x142, 0xa>(v));
    v = fminf(v, coop_dpp_f32_or_inf<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// candidates b + part, b + part + P, ... of the run [b, e) into the lane's own list; four loads in flight like scan_run_t
__device__ __forceinline__ void coop_scan_seg(const Grid &g, int b, int e, int part, int P, float qx, float qy, float qz, float max_d2, Knn5 &loc)
{
    for (int j = b + part; j < e; j += 4 * P) {
        const int j1 = (j + P < e) ? j + P : j, j2 = (j + 2 * P < e) ? j + 2 * P : j, j3 = (j + 3 * P < e) ? j + 3 * P : j;
        const f4 p0 = g.pts[j], p1 = g.pts[j1], p2 = g.pts[j2], p3 = g.pts[j3];
        const float d0 = dist2_xyz(qx, qy, qz, p0.x, p0.y, p0.z);
        const float d1 = dist2_xyz(qx, qy, qz, p1.x, p1.y, p1.z);
        const float d2 = dist2_xyz(qx, qy, qz, p2.x, p2.y, p2.z);
        const float d3 = dist2_xyz(qx, qy, qz, p3.x, p3.y, p3.z);
        if (d0 < max_d2) knn5_push(loc, d0, as_int(p0.w), j); else loc.out2 = fminf(loc.out2, d0);
        if (j1 != j) { if (d1 < max_d2) knn5_push(loc, d1, as_int(p1.w), j1); else loc.out2 = fminf(loc.out2, d1); }
        if (j2 != j) { if (d2 < max_d2) knn5_push(loc, d2, as_int(p2.w), j2); else loc.out2 = fminf(loc.out2, d2); }
        if (j3 != j) { if (d3 < max_d2) knn5_push(loc, d3, as_int(p3.w), j3); else loc.out2 = fminf(loc.out2, d3); }
    }
}

// The lanes' lists -> the wavefront's list (the same in every lane).  Five rounds: the smallest head (d2, index) of all lanes
// wins a place and its lane moves on to its next entry; a point is in exactly one lane's list, so the winner is unique.
// lb2: everything a lane rejected or pushed off its own list, and every entry still standing in some lane's list afterwards.
__device__ __forceinline__ void coop_merge(const Knn5 &loc, Knn5 &r)
{
    float cd[5];
    int ci[5], cp[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        cd[i] = loc.d2[i];
        ci[i] = loc.idx[i];
        cp[i] = loc.pos[i];
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        // squared distances are >= +0: their bit patterns order like the values; an empty head is +inf
        const unsigned long long key = ((unsigned long long)(unsigned int)as_int(cd[0]) << 32) | (unsigned long long)(unsigned int)ci[0];
        const unsigned long long kmin = wave_min_u64(key);
        const bool any = (unsigned int)(kmin >> 32) < 0x7f800000u;
        const bool win = any && key == kmin;
        const unsigned long long wm = __ballot(win);
        const int src = any ? (int)__ffsll((long long)wm) - 1 : 0;
        const int wpos = __builtin_amdgcn_readlane(cp[0], src);
        r.d2[k] = any ? __int_as_float((int)(unsigned int)(kmin >> 32)) : INFINITY;
        r.idx[k] = any ? (int)(unsigned int)kmin : LL_KNN_EMPTY;
        r.pos[k] = any ? wpos : -1;
        cnt += any ? 1 : 0;
        if (win) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                cd[i] = cd[i + 1];
                ci[i] = ci[i + 1];
                cp[i] = cp[i + 1];
            }
            cd[4] = INFINITY;
            ci[4] = LL_KNN_EMPTY;
            cp[4] = -1;
        }
    }
    r.count = cnt;
    r.lb2 = wave_min_f32(fminf(loc.lb2, cd[0]));
    r.out2 = wave_min_f32(loc.out2);
}

// knn5_search for one query per wavefront: every lane passes the same query and receives the same result.  All 64 lanes of
// the wavefront must call it together (blockDim.x a multiple of 64, one-dimensional blocks).
__device__ __forceinline__ void knn5_search_coop(const Grid &g, float qx, float qy, float qz, float max_d2, Knn5 &r)
{
    const int lane = threadIdx.x & 63;
    knn5_init(r);
    if (!ll_isfinite(qx) || !ll_isfinite(qy) || !ll_isfinite(qz)) return;
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    const float rmax_cells = sqrtf(max_d2) * g.inv_h + 2.0f;
    if (fx < -rmax_cells || fy < -rmax_cells || fz < -rmax_cells || fx > (float)g.nx + rmax_cells || fy > (float)g.ny + rmax_cells ||
        fz > (float)g.nz + rmax_cells)
        return;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    const float slack = g.slack;
    const float xm = fmaxf((fx - (float)cx) * g.h - slack, 0.0f), xp = fmaxf(((float)(cx + 1) - fx) * g.h - slack, 0.0f);
    const float ym = fmaxf((fy - (float)cy) * g.h - slack, 0.0f), yp = fmaxf(((float)(cy + 1) - fy) * g.h - slack, 0.0f);
    const float zm = fmaxf((fz - (float)cz) * g.h - slack, 0.0f), zp = fmaxf(((float)(cz + 1) - fz) * g.h - slack, 0.0f);
    Knn5 loc;
    knn5_init(loc);

    // ---- the 3x3x3 block: nine x-runs, seven lanes each, nothing pruned ---------------------------------------------------
    {
        const int ri = lane / 7, part = lane - ri * 7;
        const int y = cy + (ri % 3) - 1, z = cz + (ri / 3) - 1;
        if (ri < 9 && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
            if (x0 <= x1) {
                const int base = (z * g.ny + y) * g.nx;
                const int b = g.cell_start[base + x0], e = g.cell_start[base + x1 + 1];
                coop_scan_seg(g, b, e, part, 7, qx, qy, qz, max_d2, loc);
            }
        }
    }
    coop_merge(loc, r);

    // ---- rings / cube sweep: the control flow of knn5_search_t, one row segment per lane -----------------------------------
    const float m = fminf(fminf(fminf(xm, xp), fminf(ym, yp)), fminf(zm, zp));
    const int kmax = (int)ceilf(sqrtf(max_d2) * g.inv_h) + 1;
    for (int k = 1; k <= kmax; k++) {
        if (k == LL_KNN_CUBE_FROM) {
            int K = kmax;
            if (r.count == 5) {
                const int kd = (int)ceilf(sqrtf(r.d2[4]) * g.inv_h) + 1;
                K = kd < kmax ? kd : kmax;
            }
            if (K < LL_KNN_CUBE_FROM) K = LL_KNN_CUBE_FROM;
            const int in = LL_KNN_CUBE_FROM - 1;
            const int dz_lo = -K > -cz ? -K : -cz, dz_hi = K < g.nz - 1 - cz ? K : g.nz - 1 - cz;
            const int dy_lo = -K > -cy ? -K : -cy, dy_hi = K < g.ny - 1 - cy ? K : g.ny - 1 - cy;
            const int nyr = dy_hi - dy_lo + 1, nzr = dz_hi - dz_lo + 1;
            const int nseg = (nyr > 0 && nzr > 0) ? 2 * nyr * nzr : 0;
            for (int s = lane; s < nseg; s += 64) {
                const int row = s >> 1, seg = s & 1;
                const int dz = dz_lo + row / nyr, dy = dy_lo + row % nyr;
                const bool inner_row = dz >= -in && dz <= in && dy >= -in && dy <= in;
                if (seg == 1 && !inner_row) continue;
                int x0 = (inner_row && seg == 1) ? cx + in + 1 : cx - K;
                int x1 = (inner_row && seg == 0) ? cx - in - 1 : cx + K;
                if (x0 < 0) x0 = 0;
                if (x1 >= g.nx) x1 = g.nx - 1;
                if (x0 > x1) continue;
                const int base = ((cz + dz) * g.ny + (cy + dy)) * g.nx;
                coop_scan_seg(g, g.cell_start[base + x0], g.cell_start[base + x1 + 1], 0, 1, qx, qy, qz, max_d2, loc);
            }
            k = K;
            coop_merge(loc, r);
        } else if (k >= 2) {
            const int dz_lo = -k > -cz ? -k : -cz, dz_hi = k < g.nz - 1 - cz ? k : g.nz - 1 - cz;
            const int dy_lo = -k > -cy ? -k : -cy, dy_hi = k < g.ny - 1 - cy ? k : g.ny - 1 - cy;
            const int nyr = dy_hi - dy_lo + 1, nzr = dz_hi - dz_lo + 1;
            const int nseg = (nyr > 0 && nzr > 0) ? 2 * nyr * nzr : 0;
            for (int s = lane; s < nseg; s += 64) {
                const int row = s >> 1, seg = s & 1;
                const int dz = dz_lo + row / nyr, dy = dy_lo + row % nyr;
                const bool full_row = (dz == -k || dz == k || dy == -k || dy == k);
                if (seg == 1 && full_row) continue;
                int x0 = full_row ? cx - k : (seg == 0 ? cx - k : cx + k);
                int x1 = full_row ? cx + k : x0;
                if (full_row) {
                    if (x0 < 0) x0 = 0;
                    if (x1 >= g.nx) x1 = g.nx - 1;
                } else if (x0 < 0 || x0 >= g.nx) {
                    continue;
                }
                if (x0 > x1) continue;
                const int base = ((cz + dz) * g.ny + (cy + dy)) * g.nx;
                coop_scan_seg(g, g.cell_start[base + x0], g.cell_start[base + x1 + 1], 0, 1, qx, qy, qz, max_d2, loc);
            }
            coop_merge(loc, r);
        }
        if (cx - k <= 0 && cx + k >= g.nx - 1 && cy - k <= 0 && cy + k >= g.ny - 1 && cz - k <= 0 && cz + k >= g.nz - 1) return;
        const float bound = (float)k * g.h + m - ((k >= 2) ? slack : 0.0f);
        const float b2 = bound * bound;
        if (b2 >= max_d2 || (r.count == 5 && r.d2[4] < b2)) {
            r.lb2 = fminf(r.lb2, fminf(b2, max_d2));
            r.out2 = fminf(r.out2, fmaxf(b2, max_d2));
            return;
        }
    }
    r.lb2 = fminf(r.lb2, max_d2);
    r.out2 = fminf(r.out2, max_d2);
}

}  // namespace ll
