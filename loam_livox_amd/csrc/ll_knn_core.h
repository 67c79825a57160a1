// ll_knn_core.h -- exact 5-nearest-neighbour search on a uniform cell grid; the per-query routine shared by
// the HIP kernels and tests/hostcheck.  Stands in for pcl::KdTreeFLANN::nearestKSearch at
// hku-mars/loam_livox source/point_cloud_registration.hpp:249,351: exact k-NN, squared L2 accumulated in
// fp32 in x,y,z order (FLANN L2_Simple<float>), ascending; ties ordered by ascending original index.
//
// Grid layout (built by ll_map_kernels.hip):
//   pts[j]        float4 {x, y, z, bit-cast original index}, sorted by cell key (x fastest, then y, then z)
//   cell_start[c] index in pts of the first point of cell c, c in [0, ncell]; cell_start[ncell] = #points
// Because x is the fastest-varying key digit, the cells (cx-k..cx+k, cy, cz) of one row are one contiguous
// run of pts: a query walks (2k+1)^2 runs per ring instead of (2k+1)^3 cells.
#pragma once
#include "ll_fe_core.h"  // LL_HD

namespace ll {

struct alignas(16) f4 {
    float x, y, z, w;
};

struct Grid {
    const f4 *pts;
    const int *cell_start;
    // fp16-point variant (BASELINE config C5, ll_map_upload_f16): 8-byte records {half fx, fy, fz, uint16 cx} = position of
    // the point inside its cell as a fraction of the cell size + the cell's x index, and the original index of every
    // record.  Only the map k-NN entry point reads them; the registrar needs the fp32 records.
    const unsigned long long *pts16;
    const int *perm;
    float ox, oy, oz;  // origin (min corner)
    float inv_h, h;
    float slack;  // conservative allowance for fp32 rounding of cell assignment (metres)
    float guard;  // reuse guard band (metres, 0 = off): phase 1 prunes a run / end cell only when it lies farther than the running
                  // 5th best PLUS this -- a few more candidates per search, but the lower bound a pruned run leaves behind is then at
                  // least d5 + guard, and the displacement budget of the reuse record (half of lb - d5) is set by the true 6th
                  // neighbour instead of a box distance that exceeds d5 by a hair (the registrar sets 5 cm: half as many late searches)
    int nx, ny, nz;
};

struct Knn5 {
    float d2[5];
    int idx[5];  // original (upload-order) index: the tie-break key and what callers see
    int pos[5];  // position in the cell-sorted array (to fetch the coordinates again)
    int count;
    float out2;  // lower bound on the squared distance of every point at or beyond the match radius (INF if none);
                 // used only while fewer than five neighbours are inside the radius (knn5_reuse_margin)
    float lb2;   // lower bound on the squared distance of every map point that is NOT in the list
                 // (6th best seen, nearest pruned/unvisited cell, match radius) -- lets the next ICP iteration
                 // prove that the neighbour set is unchanged without searching again
};

#define LL_KNN_EMPTY 0x7fffffff
// test-only instrumentation hook (tests/hostcheck counts row look-ups and candidates per query); nothing on the device
#ifndef LL_KNN_STAT
#define LL_KNN_STAT(counter, n)
#endif
#define LL_KNN_CUBE_FROM 5  // ring at which the search stops growing shells and sweeps the remaining cube (knn5_search_t)

LL_HD float knn5_d2(const Knn5 &r, int i) { return r.d2[i]; }
LL_HD int knn5_idx(const Knn5 &r, int i) { return r.idx[i]; }

LL_HD void knn5_init(Knn5 &r)
{
    for (int i = 0; i < 5; i++) {
        r.d2[i] = INFINITY;
        r.idx[i] = LL_KNN_EMPTY;
        r.pos[i] = -1;
    }
    r.count = 0;
    r.lb2 = INFINITY;
    r.out2 = INFINITY;
}

LL_HD int as_int(float f)
{
    union {
        float f;
        int i;
    } u;
    u.f = f;
    return u.i;
}

LL_HD bool lex_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

// Ordered insertion by (d2, idx): carry the displaced element down a fully unrolled compare-swap chain
// (static indices only, so the five slots stay in registers on the GPU).  (Round 3 tried the list as packed 64-bit
// (d2 bits, idx) keys, one compare and selects per place, no branches: 12 % fewer VALU instructions per full search, the same
// time -- 64-bit compares do not issue at the 32-bit rate --, 19 more registers, the late-iteration list kernel 6 - 8 % slower;
// kept as it was.)
LL_HD void knn5_push(Knn5 &r, float d2, int idx, int pos)
{
    if (!lex_less(d2, idx, r.d2[4], r.idx[4])) {
        r.lb2 = fminf(r.lb2, d2);  // rejected: it stays outside the list
        return;
    }
    if (r.count < 5) r.count++;
    float cd = d2;
    int ci = idx, cp = pos;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (lex_less(cd, ci, r.d2[i], r.idx[i])) {
            const float td = r.d2[i];
            const int ti = r.idx[i], tp = r.pos[i];
            r.d2[i] = cd;
            r.idx[i] = ci;
            r.pos[i] = cp;
            cd = td;
            ci = ti;
            cp = tp;
        }
    }
    r.lb2 = fminf(r.lb2, cd);  // whatever fell off the end (INF while the list was not full)
}

// FLANN L2_Simple<float>: result = 0; result += diff*diff for x, y, z.  No FMA contraction.
LL_HD float dist2_xyz(float qx, float qy, float qz, float px, float py, float pz)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    float dx = qx - px, dy = qy - py, dz = qz - pz;
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

LL_HD int cell_coord(float v, float o, float inv_h)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return (int)floorf((v - o) * inv_h);
}

// How the candidate records are read.  PtF32: the 16-byte {x, y, z, index} records every production path uses.
struct PtF32 {
    struct Row {};
    static LL_HD Row row(const Grid &, int) { return Row(); }
    static LL_HD void load(const Grid &g, const Row &, int j, float &x, float &y, float &z, int &tok)
    {
        const f4 p = g.pts[j];
        x = p.x;
        y = p.y;
        z = p.z;
        tok = as_int(p.w);
    }
    static LL_HD void push(const Grid &, Knn5 &r, float d2, int j, int tok) { knn5_push(r, d2, tok, j); }
};

template <class PT>
LL_HD void scan_run_t(const Grid &g, int c_lo, int c_hi /*inclusive cell keys of one x-run*/, float qx, float qy, float qz,
                      float max_d2, Knn5 &r)
{
    const int b = g.cell_start[c_lo], e = g.cell_start[c_hi + 1];
    LL_KNN_STAT(0, 1);
    LL_KNN_STAT(1, e - b);
    const typename PT::Row row = PT::row(g, c_lo);
    // four candidates per trip: the four record loads are independent, so their latencies overlap
    for (int j = b; j < e; j += 4) {
        const int j1 = (j + 1 < e) ? j + 1 : j, j2 = (j + 2 < e) ? j + 2 : j, j3 = (j + 3 < e) ? j + 3 : j;
        float x0, y0, z0, x1, y1, z1, x2, y2, z2, x3, y3, z3;
        int t0, t1, t2, t3;
        PT::load(g, row, j, x0, y0, z0, t0);
        PT::load(g, row, j1, x1, y1, z1, t1);
        PT::load(g, row, j2, x2, y2, z2, t2);
        PT::load(g, row, j3, x3, y3, z3, t3);
        const float d0 = dist2_xyz(qx, qy, qz, x0, y0, z0);
        const float d1 = dist2_xyz(qx, qy, qz, x1, y1, z1);
        const float d2 = dist2_xyz(qx, qy, qz, x2, y2, z2);
        const float d3 = dist2_xyz(qx, qy, qz, x3, y3, z3);
        if (d0 < max_d2) PT::push(g, r, d0, j, t0); else r.out2 = fminf(r.out2, d0);
        if (j1 != j) { if (d1 < max_d2) PT::push(g, r, d1, j1, t1); else r.out2 = fminf(r.out2, d1); }
        if (j2 != j) { if (d2 < max_d2) PT::push(g, r, d2, j2, t2); else r.out2 = fminf(r.out2, d2); }
        if (j3 != j) { if (d3 < max_d2) PT::push(g, r, d3, j3, t3); else r.out2 = fminf(r.out2, d3); }
    }
}

LL_HD void scan_run(const Grid &g, int c_lo, int c_hi, float qx, float qy, float qz, float max_d2, Knn5 &r)
{
    scan_run_t<PtF32>(g, c_lo, c_hi, qx, qy, qz, max_d2, r);
}

// Exact 5-NN of (qx,qy,qz) among points with squared distance < max_d2.
//
// Phase 1 visits the 3x3x3 block of cells around the query as nine x-runs, nearest run first, and prunes with the
// running 5th-best distance: a run (or its outer cells) whose box is farther than the current 5th best cannot
// contribute.  Box distances are shrunk by `slack` so that fp32 rounding of the cell assignment can never
// prune a real candidate; the comparison is strict, so exact-distance ties (ordered by index) are still seen.
// After phase 1 every unvisited point is farther than  bound_1 = h + m  (m = distance from the query to the
// nearest wall of its own cell); if the 5th best is not inside that bound, phase 2 grows Chebyshev rings
// k = 2, 3, ... until it is, or until the bound passes the match radius.
template <class PT>
LL_HD void knn5_search_t(const Grid &g, float qx, float qy, float qz, float max_d2, Knn5 &r)
{
    knn5_init(r);
    if (!ll_isfinite(qx) || !ll_isfinite(qy) || !ll_isfinite(qz)) return;
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    // far outside the grid: nothing can be within the match radius (also keeps the int conversion defined)
    const float rmax_cells = sqrtf(max_d2) * g.inv_h + 2.0f;
    if (fx < -rmax_cells || fy < -rmax_cells || fz < -rmax_cells || fx > (float)g.nx + rmax_cells ||
        fy > (float)g.ny + rmax_cells || fz > (float)g.nz + rmax_cells)
        return;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    const float slack = g.slack;
    // distances (metres, under-estimated by `slack`) from the query to the walls of its own cell
    const float xm = fmaxf((fx - (float)cx) * g.h - slack, 0.0f), xp = fmaxf(((float)(cx + 1) - fx) * g.h - slack, 0.0f);
    const float ym = fmaxf((fy - (float)cy) * g.h - slack, 0.0f), yp = fmaxf(((float)(cy + 1) - fy) * g.h - slack, 0.0f);
    const float zm = fmaxf((fz - (float)cz) * g.h - slack, 0.0f), zp = fmaxf(((float)(cz + 1) - fz) * g.h - slack, 0.0f);
    const float xm2 = xm * xm, xp2 = xp * xp;

    // ---- phase 1: 3x3 runs, own run first, then the 4 face neighbours, then the 4 diagonal ones -----------
    // (dy,dz) order packed two bits per entry (0 -> -1, 1 -> 0, 2 -> +1); kept as a rolled loop so the
    // candidate-scan code exists once (small instruction footprint, fewer live registers)
    const unsigned int DY_CODES = 0x22161u, DZ_CODES = 0x28215u;
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    for (int ri = 0; ri < 9; ri++) {
        const int dyc = (int)((DY_CODES >> (2 * ri)) & 3u) - 1, dzc = (int)((DZ_CODES >> (2 * ri)) & 3u) - 1;
        const int y = cy + dyc, z = cz + dzc;
        if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
        const float dy = dyc < 0 ? ym : (dyc > 0 ? yp : 0.0f);
        const float dz = dzc < 0 ? zm : (dzc > 0 ? zp : 0.0f);
        const float row2 = dy * dy + dz * dz;
        float lim = max_d2;
        if (r.count == 5) {
            lim = r.d2[4];
            if (g.guard > 0.0f) {
                const float d5g = sqrtf(lim) + g.guard;
                lim = fminf(d5g * d5g, max_d2);  // (points at or beyond the match radius never count)
            }
        }
        if (row2 > lim) {
            r.lb2 = fminf(r.lb2, row2);  // everything in this run is at least this far
            if (row2 >= max_d2) r.out2 = fminf(r.out2, row2);
            continue;
        }
        int x0 = cx - 1, x1 = cx + 1;
        if (xm2 + row2 > lim) {
            x0 = cx;
            r.lb2 = fminf(r.lb2, xm2 + row2);
            if (xm2 + row2 >= max_d2) r.out2 = fminf(r.out2, xm2 + row2);
        }
        if (xp2 + row2 > lim) {
            x1 = cx;
            r.lb2 = fminf(r.lb2, xp2 + row2);
            if (xp2 + row2 >= max_d2) r.out2 = fminf(r.out2, xp2 + row2);
        }
        if (x0 < 0) x0 = 0;
        if (x1 >= g.nx) x1 = g.nx - 1;
        if (x0 > x1) continue;
        const int base = (z * g.ny + y) * g.nx;
        LL_KNN_STAT(4, ri);
        scan_run_t<PT>(g, base + x0, base + x1, qx, qy, qz, max_d2, r);
    }

    const float m = fminf(fminf(fminf(xm, xp), fminf(ym, yp)), fminf(zm, zp));  // already shrunk by slack
    const int kmax = (int)ceilf(sqrtf(max_d2) * g.inv_h) + 1;
    for (int k = 1; k <= kmax; k++) {
        if (k == 2) LL_KNN_STAT(2, 1);
        if (k == LL_KNN_CUBE_FROM) {
            LL_KNN_STAT(3, 1);
            // ---- phase 3 (sparse surroundings): four rings have not settled the answer -- the query looks into a part
            // of the map with next to no points (the frontier of a growing local map, a sparse voxel-filtered cloud).
            // Shell by shell, the remaining rings would look up the two end cells of every interior row again and
            // again (~11 k cell lookups out to a 7 m radius at 0.6 m cells); one sweep over the rows of the whole cube
            // that is still needed costs two lookups per row (~1.5 k).  The order of the visits is irrelevant to the
            // result, and the cells of rings 1 .. LL_KNN_CUBE_FROM-1 are left out, so no point is offered twice.
            int K = kmax;
            if (r.count == 5) {
                const int kd = (int)ceilf(sqrtf(r.d2[4]) * g.inv_h) + 1;  // the 5th best can only come closer
                K = kd < kmax ? kd : kmax;
            }
            if (K < LL_KNN_CUBE_FROM) K = LL_KNN_CUBE_FROM;
            const int in = LL_KNN_CUBE_FROM - 1;  // half-width of the cube already scanned
            const int dz_lo = -K > -cz ? -K : -cz, dz_hi = K < g.nz - 1 - cz ? K : g.nz - 1 - cz;
            const int dy_lo = -K > -cy ? -K : -cy, dy_hi = K < g.ny - 1 - cy ? K : g.ny - 1 - cy;
            for (int dz = dz_lo; dz <= dz_hi; dz++) {
                for (int dy = dy_lo; dy <= dy_hi; dy++) {
                    const int base = ((cz + dz) * g.ny + (cy + dy)) * g.nx;
                    const bool inner_row = dz >= -in && dz <= in && dy >= -in && dy <= in;
                    for (int seg = 0; seg < (inner_row ? 2 : 1); seg++) {
                        int x0 = (inner_row && seg == 1) ? cx + in + 1 : cx - K;
                        int x1 = (inner_row && seg == 0) ? cx - in - 1 : cx + K;
                        if (x0 < 0) x0 = 0;
                        if (x1 >= g.nx) x1 = g.nx - 1;
                        if (x0 <= x1) scan_run_t<PT>(g, base + x0, base + x1, qx, qy, qz, max_d2, r);
                    }
                }
            }
            k = K;  // the cube of half-width K has been seen: fall through to the termination test of ring K
        } else if (k >= 2) {
            // ---- phase 2 (rare on dense maps): the full shell at Chebyshev distance k --------------------------
            // The loops only run over the part of the shell that lies inside the grid: a query at the edge of a small
            // local map would otherwise spend its time on (2k+1)^2 empty iterations per ring.
            const int dz_lo = -k > -cz ? -k : -cz, dz_hi = k < g.nz - 1 - cz ? k : g.nz - 1 - cz;
            const int dy_lo = -k > -cy ? -k : -cy, dy_hi = k < g.ny - 1 - cy ? k : g.ny - 1 - cy;
            for (int dz = dz_lo; dz <= dz_hi; dz++) {
                const int z = cz + dz;
                for (int dy = dy_lo; dy <= dy_hi; dy++) {
                    const int y = cy + dy;
                    const int base = (z * g.ny + y) * g.nx;
                    const bool full_row = (dz == -k || dz == k || dy == -k || dy == k);
                    // a full x-run on the shell's faces, otherwise only the two end cells of the row
                    for (int seg = 0; seg < (full_row ? 1 : 2); seg++) {
                        int x0 = full_row ? cx - k : (seg == 0 ? cx - k : cx + k);
                        int x1 = full_row ? cx + k : x0;
                        if (full_row) {  // clamp the run to the grid; single end cells must lie inside it
                            if (x0 < 0) x0 = 0;
                            if (x1 >= g.nx) x1 = g.nx - 1;
                        } else if (x0 < 0 || x0 >= g.nx) {
                            continue;
                        }
                        if (x0 <= x1) scan_run_t<PT>(g, base + x0, base + x1, qx, qy, qz, max_d2, r);
                    }
                }
            }
        }
        // the cube of ring k covers the whole grid: every point has been seen, nothing is left to bound
        if (cx - k <= 0 && cx + k >= g.nx - 1 && cy - k <= 0 && cy + k >= g.ny - 1 && cz - k <= 0 && cz + k >= g.nz - 1) return;
        const float bound = (float)k * g.h + m - ((k >= 2) ? slack : 0.0f);
        const float b2 = bound * bound;
        if (b2 >= max_d2 || (r.count == 5 && r.d2[4] < b2)) {
            // done: every point within the match radius has been seen, or the 5 best cannot be displaced by an
            // unvisited point.  Unvisited points are farther than `bound`; points beyond the radius never count.
            r.lb2 = fminf(r.lb2, fminf(b2, max_d2));
            r.out2 = fminf(r.out2, fmaxf(b2, max_d2));  // unvisited points are beyond `bound` (and beyond the radius
                                                         // when the search was exhaustive inside it)
            return;
        }
    }
    r.lb2 = fminf(r.lb2, max_d2);
    r.out2 = fminf(r.out2, max_d2);
}

LL_HD void knn5_search(const Grid &g, float qx, float qy, float qz, float max_d2, Knn5 &r)
{
    knn5_search_t<PtF32>(g, qx, qy, qz, max_d2, r);
}

// How far the query may move before the result of knn5_search has to be recomputed (metres, conservative):
//   5 found : the set is unchanged while  d5 + delta < lb - delta          ->  (lb - d5) / 2
//   < 5     : still fewer than 5 inside the radius while  lb - delta >= R   ->  lb - R
// where lb = sqrt(lb2).  fp32 rounding of the distances is covered by the subtracted slack.
LL_HD float knn5_reuse_margin(const Knn5 &r, float max_d2)
{
    float lb = sqrtf(r.lb2);
    float mg;
    if (r.count == 5) {
        mg = 0.5f * (lb - sqrtf(r.d2[4]));
    } else {
        // every point inside the radius is in the list (the search was exhaustive there); the state "< 5 inside"
        // persists while no outside point can enter: nearest outside point (or unvisited region) minus the radius.
        // Only meaningful when the ring loop really covered the radius, i.e. out2 >= max_d2.
        lb = sqrtf(r.out2);
        mg = (r.out2 >= max_d2) ? lb - sqrtf(max_d2) : 0.0f;
    }
    mg -= 1e-5f * (1.0f + lb);
    return (mg > 0.0f && ll_isfinite(mg)) ? mg : 0.0f;
}


// ---- neighbour reuse across ICP iterations (exact) -------------------------------------------------------------
// The registrar queries the same feature again after every pose update; late iterations move a query by far less
// than the gaps between its neighbours.  KnnRef remembers where the neighbour list was last established, what it was
// and two displacement budgets (metres, conservative):
//   m_set    : the SET of 5 neighbours is provably unchanged (knn5_reuse_margin);
//   m_strong : additionally their ORDER is unchanged (half the smallest gap between consecutive neighbour
//              distances), i.e. the whole result -- and therefore the residual block -- is bit-for-bit what a new
//              search would give, and nothing has to be recomputed at all.
struct KnnRef {
    float qx, qy, qz;
    float m_strong, m_set;
    int pos[5];  // pos[4] < 0: fewer than 5 neighbours inside the match radius
};

LL_HD float knn5_order_margin(const Knn5 &r)
{
    if (r.count != 5) return INFINITY;  // nothing to order
    float d[5];
    for (int i = 0; i < 5; i++) d[i] = sqrtf(r.d2[i]);
    float g = INFINITY;
    for (int i = 0; i < 4; i++) g = fminf(g, d[i + 1] - d[i]);
    g = 0.5f * g - 1e-5f * (1.0f + d[4]);
    return (g > 0.0f) ? g : 0.0f;
}

LL_HD void knn5_make_ref(const Knn5 &r, float qx, float qy, float qz, float max_d2, KnnRef &ref)
{
    ref.qx = qx;
    ref.qy = qy;
    ref.qz = qz;
    ref.m_set = knn5_reuse_margin(r, max_d2);
    ref.m_strong = fminf(ref.m_set, knn5_order_margin(r));
    for (int i = 0; i < 5; i++) ref.pos[i] = (r.count == 5) ? r.pos[i] : -1;
}

// displacement of the query since the reference was taken, slightly over-estimated
LL_HD float knn5_ref_delta(const KnnRef &ref, float qx, float qy, float qz)
{
    return sqrtf(dist2_xyz(qx, qy, qz, ref.qx, ref.qy, ref.qz)) * 1.000001f + 1e-7f;
}

// Set-stable path: same five neighbours, possibly in another order.  Re-evaluates and re-sorts them at the new
// position (`r` = exactly what knn5_search would return, list part) and moves the reference to the new position
// (budgets shrink by the distance travelled -- triangle inequality -- and the order budget is recomputed).
LL_HD void knn5_resort(const Grid &g, KnnRef &ref, float delta, float qx, float qy, float qz, float max_d2, Knn5 &r)
{
    knn5_init(r);
    if (ref.pos[4] >= 0) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const f4 p = g.pts[ref.pos[i]];
            const float d2 = dist2_xyz(qx, qy, qz, p.x, p.y, p.z);
            if (d2 < max_d2) knn5_push(r, d2, as_int(p.w), ref.pos[i]);
        }
    }
    const float m_set = fmaxf(ref.m_set - delta, 0.0f);
    ref.qx = qx;
    ref.qy = qy;
    ref.qz = qz;
    ref.m_set = m_set;
    if (r.count == 5) {
        for (int i = 0; i < 5; i++) ref.pos[i] = r.pos[i];
        ref.m_strong = fminf(m_set, knn5_order_margin(r));
    } else {
        // a neighbour left the match radius (or there never were five): the query has no valid block; it stays so
        // while the set budget lasts, and the stored positions are kept for the next re-sort
        ref.m_strong = (ref.pos[4] < 0) ? m_set : 0.0f;
    }
}

}  // namespace ll
