// ll_knn_tile.h -- the exact 5-NN search of ll_knn_core.h for a WAVEFRONT OF QUERIES THAT SHARE A MAP TILE.
//
// Stands in for pcl::KdTreeFLANN::nearestKSearch at hku-mars/loam_livox source/point_cloud_registration.hpp:351 (and :249)
// like knn5_search does: exact k-NN, squared L2 accumulated in fp32 in x,y,z order, ties by original index.
//
// Why: one lane per query walking its own nine x-runs (knn5_search_t) keeps the VALU issue port busy at ~50 % lane utilisation --
// a wavefront executes every run any of its lanes needs and every ordered insertion any of its lanes makes.  The workload is
// far more redundant than that schedule can use: the ~17 k surface queries of a Mid-40 scan fall into 60 - 100 cells of the
// 0.6 m grid (hundreds of queries per cell), and a query's whole 27-cell neighbourhood holds ~25 map points.  So the queries
// of a scan are sorted by map cell once per registration (reg_qsort_kernel), a wavefront takes 64 consecutive queries of that
// order -- one or two neighbouring cells --, stages the map points of the cells' common neighbourhood (the TILE: bounding box of
// the lanes' cells +- 1, at most 5 x 5 x 5 cells, a few contiguous x-runs of the cell-sorted array, ~35 points) in LDS with
// coalesced loads, and every lane offers EVERY tile point to its sorted top five through a branch-free min / med3 network on
// 32-bit KEYS {squared distance with its low 8 mantissa bits replaced by the point's place in the tile} (tilek_offer: 6
// instructions per point besides the distance, which packed two-float arithmetic computes for two points at a time; no payload moves, no divergence, no dependent loads); the five winners are then
// evaluated again exactly.  Truncating the distance is harmless unless two of the six smallest keys agree in all their distance
// bits -- then (and for true ties) the lane searches again.  Afterwards a lane's answer is exact iff its 5th best lies
// inside the distance every unvisited point must exceed (one full ring of cells around its own cell: the k = 1 termination test
// of knn5_search_t) and no exact distance tie touches the list; the other lanes (0.1 % on the C2 map: sparse surroundings, ties,
// queries outside the grid) run knn5_search.  Same lists as knn5_search, bit for bit; the reuse bounds are valid and
// tighter (nothing inside the tile is pruned, so lb2 is the true 6th-nearest distance or the ring bound).
#pragma once
#include "ll_knn_core.h"

namespace ll {

// median of three (device: v_med3_f32).  Distances are never NaN here (finite queries, finite map points, +inf padding).
LL_HD float tile_med3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c < hi ? c : hi);
#endif
}

// a lane's running result: the five smallest squared distances seen so far, ascending, with the candidates' positions in the
// cell-sorted array, and the smallest distance of everything that was offered and is not (or no longer) in the list
struct Tile5 {
    float d[5];
    int p[5];
    float lb;
};

LL_HD void tile5_init(Tile5 &t)
{
    for (int i = 0; i < 5; i++) {
        t.d[i] = INFINITY;
        t.p[i] = -1;
    }
    t.lb = INFINITY;
}

// Offer one candidate (squared distance c, position pc).  Equal distances keep their order of arrival (strict compares); the
// caller detects afterwards whether a tie could have mattered (tile5_has_tie) and searches again with the exact tie rule.
// No branches: place i of the new list is med3(d[i-1], c, d[i]) of the old one.
LL_HD void tile5_offer(Tile5 &t, float c, int pc)
{
    const bool m0 = c < t.d[0], m1 = c < t.d[1], m2 = c < t.d[2], m3 = c < t.d[3], m4 = c < t.d[4];
    t.lb = fminf(t.lb, fmaxf(t.d[4], c));  // what falls off the end, or the candidate itself
    const float n4 = tile_med3(t.d[3], c, t.d[4]);
    const float n3 = tile_med3(t.d[2], c, t.d[3]);
    const float n2 = tile_med3(t.d[1], c, t.d[2]);
    const float n1 = tile_med3(t.d[0], c, t.d[1]);
    const float n0 = fminf(t.d[0], c);
    const int q4 = m4 ? (m3 ? t.p[3] : pc) : t.p[4];
    const int q3 = m3 ? (m2 ? t.p[2] : pc) : t.p[3];
    const int q2 = m2 ? (m1 ? t.p[1] : pc) : t.p[2];
    const int q1 = m1 ? (m0 ? t.p[0] : pc) : t.p[1];
    const int q0 = m0 ? pc : t.p[0];
    t.d[0] = n0;
    t.d[1] = n1;
    t.d[2] = n2;
    t.d[3] = n3;
    t.d[4] = n4;
    t.p[0] = q0;
    t.p[1] = q1;
    t.p[2] = q2;
    t.p[3] = q3;
    t.p[4] = q4;
}

// An exact tie among the six smallest distances seen (the five kept + the best one left out): only then can the order of
// arrival differ from the (distance, original index) order of knn5_push.  Ties among +inf (fewer than five candidates) count
// too -- such a lane is not final anyway.
LL_HD bool tile5_has_tie(const Tile5 &t)
{
    return t.d[0] == t.d[1] || t.d[1] == t.d[2] || t.d[2] == t.d[3] || t.d[3] == t.d[4] || t.d[4] == t.lb;
}

// ---- the key network -------------------------------------------------------------------------------------------------------
// One pass covers up to LL_TILE_CAP staged points.  key = bits(d2) with the low LL_TILE_IDX_BITS bits replaced by the point's index
// in the pass: squared distances are >= +0, so their bit patterns order like the values and the keys order by (truncated distance,
// index).  Unsigned integer min / max / med3 throughout: no NaN or denormal rules to think about.
#define LL_TILE_IDX_BITS 8
#define LL_TILE_CAP (1 << LL_TILE_IDX_BITS)
#define LL_TILE_KEY_EMPTY 0xffffffffu

LL_HD unsigned int tile_umin(unsigned int a, unsigned int b) { return a < b ? a : b; }
LL_HD unsigned int tile_umax(unsigned int a, unsigned int b) { return a < b ? b : a; }
// (the compiler turns this shape into v_med3_u32)
LL_HD unsigned int tile_umed3(unsigned int a, unsigned int b, unsigned int c) { return tile_umax(tile_umin(a, b), tile_umin(tile_umax(a, b), c)); }

struct TileK {
    unsigned int k[6];  // the six smallest keys so far, ascending: five for the list, the sixth bounds everything that is not in it
};

LL_HD void tilek_init(TileK &t)
{
    for (int i = 0; i < 6; i++) t.k[i] = LL_TILE_KEY_EMPTY;
}

LL_HD unsigned int tile_key(float d2, int j) { return ((unsigned int)as_int(d2) & ~(unsigned int)(LL_TILE_CAP - 1)) | (unsigned int)j; }

// place i of the new list is med3(k[i-1], c, k[i]) of the old one: six instructions, no branches
LL_HD void tilek_offer(TileK &t, unsigned int c)
{
    const unsigned int n5 = tile_umed3(t.k[4], c, t.k[5]);
    const unsigned int n4 = tile_umed3(t.k[3], c, t.k[4]);
    const unsigned int n3 = tile_umed3(t.k[2], c, t.k[3]);
    const unsigned int n2 = tile_umed3(t.k[1], c, t.k[2]);
    const unsigned int n1 = tile_umed3(t.k[0], c, t.k[1]);
    t.k[0] = tile_umin(t.k[0], c);
    t.k[1] = n1;
    t.k[2] = n2;
    t.k[3] = n3;
    t.k[4] = n4;
    t.k[5] = n5;
}

// two of the six smallest keys share all their distance bits: the truncated order may not be the true one (or it is a true tie).
// Pairs that lie wholly beyond the match radius do not count (their order never reaches a result; the padding entries of a tile with
// fewer than six points are such pairs).
LL_HD bool tilek_collision(const TileK &t, float max_d2)
{
    const int s = LL_TILE_IDX_BITS;
    const unsigned int lim = ((unsigned int)as_int(max_d2) >> s) + 1u;  // truncated distances from here on are > max_d2
    bool c = false;
    for (int i = 0; i < 5; i++) c = c || ((t.k[i] >> s) == (t.k[i + 1] >> s) && (t.k[i] >> s) < lim);
    return c;
}

// a lower bound on the squared distance behind a key (truncation rounds towards zero); +inf for "nothing"
LL_HD float tile_key_lower(unsigned int k)
{
    if (k == LL_TILE_KEY_EMPTY) return INFINITY;
    union {
        unsigned int u;
        float f;
    } v;
    v.u = k & ~(unsigned int)(LL_TILE_CAP - 1);
    return v.f;
}

// Where a query sits in the grid: the quantities knn5_search_t derives at its start, operation for operation -- except that a query
// up to one cell OUTSIDE the grid takes the nearest cell of the grid as its own (the grid is the bounding box of the map's points, so
// every scan point on an outer wall lies a centimetre outside it half of the time: 1 % of the C2 queries).  Its distance to the walls
// of that cell is then 0 on the side it sticks out of, so m = 0 (less the slack) and the finishing bound is h: valid -- a point outside the
// 3 x 3 x 3 block around the adopted cell is at least one cell away along some axis from a query that lies in or beyond that cell.
struct TileQ {
    int cx, cy, cz;
    float m;      // distance (metres, shrunk by slack) from the query to the nearest wall of its own cell
    bool ingrid;  // finite, and its cell (or the cell it adopts) is a cell of the grid (otherwise the query goes to knn5_search)
};

LL_HD void tile_query(const Grid &g, float qx, float qy, float qz, TileQ &o)
{
    o.cx = o.cy = o.cz = 0;
    o.m = 0.0f;
    o.ingrid = false;
    if (!ll_isfinite(qx) || !ll_isfinite(qy) || !ll_isfinite(qz)) return;
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    if (!(fx >= -1.0f && fy >= -1.0f && fz >= -1.0f && fx < (float)g.nx + 1.0f && fy < (float)g.ny + 1.0f && fz < (float)g.nz + 1.0f)) return;
    int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    cx = cx < 0 ? 0 : (cx >= g.nx ? g.nx - 1 : cx);  // (also the rounding of the int -> float conversions above)
    cy = cy < 0 ? 0 : (cy >= g.ny ? g.ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= g.nz ? g.nz - 1 : cz);
    const float slack = g.slack;
    const float xm = fmaxf((fx - (float)cx) * g.h - slack, 0.0f), xp = fmaxf(((float)(cx + 1) - fx) * g.h - slack, 0.0f);
    const float ym = fmaxf((fy - (float)cy) * g.h - slack, 0.0f), yp = fmaxf(((float)(cy + 1) - fy) * g.h - slack, 0.0f);
    const float zm = fmaxf((fz - (float)cz) * g.h - slack, 0.0f), zp = fmaxf(((float)(cz + 1) - fz) * g.h - slack, 0.0f);
    o.cx = cx;
    o.cy = cy;
    o.cz = cz;
    o.m = fminf(fminf(fminf(xm, xp), fminf(ym, yp)), fminf(zm, zp));
    // an adopted cell: the query is at least 0 beyond the wall it sticks out of -- less the slack that the in-cell distances above
    // already carry
    if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)g.nx && fy < (float)g.ny && fz < (float)g.nz)) o.m = -slack;
    o.ingrid = true;
}

// The lane has been offered every point of a tile that contains the 3 x 3 x 3 block of cells around its own cell (clipped to
// the grid).  Returns true and fills r (the list and bounds knn5_search would be allowed to return) when that settles the
// answer: five neighbours inside the match radius, the fifth closer than anything outside the block can be (the k = 1 test of
// knn5_search_t: bound = h + m), no tie -- or fewer than five inside a radius that the block covers.  Otherwise false: search again
// with knn5_search.  r.idx is 0 for the entries of the list (callers that need original indices read pts[pos].w), LL_KNN_EMPTY beyond.
// SHORT_LISTS false: the "fewer than five" case is always left to knn5_search (the surface search of the tile kernel: its radius is 7 m,
// and the code it does not carry keeps the kernel inside 80 registers).
template <bool SHORT_LISTS = true>
LL_HD bool tile5_finish(const Grid &g, const Tile5 &t, const TileQ &tq, float max_d2, Knn5 &r)
{
    const float bound = g.h + tq.m;
    const float b2 = bound * bound;
    if (!(t.d[4] < max_d2)) {
        if (!SHORT_LISTS) return false;
        // Fewer than five inside the radius.  When the radius reaches beyond the block only the rings can tell (7 m at the plane radius);
        // when the block covers it (the line radius, 1.41 m, on the corner map's 1.45 m cells) nothing else can turn up: the answer is the
        // points found so far -- "no five neighbours" for the caller (PCR:249-254) -- with the list and bounds knn5_search leaves behind.
        if (!(max_d2 <= b2)) return false;
        knn5_init(r);
        int n = 0;
        for (int i = 0; i < 4; i++) {
            if (t.d[i] < max_d2) {
                if (t.d[i] == t.d[i + 1]) return false;  // an exact tie inside the radius: the index rule orders it
                r.d2[i] = t.d[i];
                r.pos[i] = t.p[i];
                r.idx[i] = 0;
                n = i + 1;
            }
        }
        float first_out = t.d[4];  // the nearest point at or beyond the radius that was seen (static indices: registers on the device)
        for (int i = 3; i >= 0; i--)
            if (!(t.d[i] < max_d2)) first_out = t.d[i];
        r.count = n;
        r.lb2 = fminf(b2, max_d2);         // (everything outside the list is at or beyond the radius)
        r.out2 = fminf(first_out, b2);     // ... or the block's edge
        return true;
    }
    if (tile5_has_tie(t)) return false;
    // (when the block covers the whole grid every point has been offered and the test below is only conservative)
    if (!(t.d[4] < b2)) return false;
    for (int i = 0; i < 5; i++) {
        r.d2[i] = t.d[i];
        r.pos[i] = t.p[i];
        r.idx[i] = 0;
    }
    r.count = 5;
    r.lb2 = fminf(t.lb, fminf(b2, max_d2));
    r.out2 = fmaxf(b2, max_d2);
    return true;
}

// ---- tile geometry shared by the kernel and its host model --------------------------------------------------------------
// The x-runs of a tile: rows (y, z) in [y0, y1] x [z0, z1], cells x0 .. x1 of each, all inside the grid.
#define LL_TILE_MAX_ROWS 25  // participants of a round lie within +-1 cell of the round's leader: at most 5 x 5 rows of <= 5 cells

#if defined(__HIPCC__)
// ---- device: one round-based search for the 64 queries of a wavefront ------------------------------------------------------
// All 64 lanes call it together (whole wavefronts, one-dimensional blocks).  q*: the lane's query (any value when !active);
// tile: this wavefront's LDS staging buffer, LL_TILE_CAP + 4 entries {x, y, z, bits(position)}.  On return `final` says whether r
// holds the lane's exact result; lanes with active && !final must run knn5_search.
// degenerate: whether neighbours 0, 2, 4 of a final lane coincide as float points (1 / 0), -1 = not known (a tile of several passes).
#define LL_TILE_FAR 1.0e18f  // coordinate of a padding entry: its distance is huge and finite or +inf, never NaN
#ifdef LL_TILE_TIMING  // instrumented builds only (tools/gpu_r6_tile.sh): shader clocks per phase of this wavefront
#define LL_TT_ARG , long long *tt
#define LL_TT_PASS , tt
#define LL_TT(slot, var)                  \
    do {                                  \
        const long long now_ = clock64(); \
        tt[slot] += now_ - var;           \
        var = now_;                       \
    } while (0)
#define LL_TT_ADD(slot, v) tt[slot] += (v)
#else
#define LL_TT_ARG
#define LL_TT_PASS
#define LL_TT(slot, var)
#define LL_TT_ADD(slot, v)
#endif
template <bool SHORT_LISTS = false>
__device__ __forceinline__ void knn5_tile_wave(const Grid &g, bool active, float qx, float qy, float qz, float max_d2, float4 *tile,
                                               Knn5 &r, bool &final, int &degenerate LL_TT_ARG)
{
    degenerate = -1;
#ifdef LL_TILE_TIMING
    long long tw = clock64();
#endif
    const int lane = threadIdx.x & 63;
    TileQ tq;
    tile_query(g, qx, qy, qz, tq);
    const bool ingrid = active && tq.ingrid;
    final = false;
    unsigned long long todo = __ballot(ingrid);
    LL_TT(0, tw);  // tile_query
    while (todo != 0ull) {  // (uniform) one round per group of lanes whose cells lie within +-1 of the leader's: 1.05 rounds on C2
        const int leader = (int)__ffsll((long long)todo) - 1;
        const int lx = __builtin_amdgcn_readlane(tq.cx, leader), ly = __builtin_amdgcn_readlane(tq.cy, leader),
                  lz = __builtin_amdgcn_readlane(tq.cz, leader);
        const int dx = tq.cx - lx, dy = tq.cy - ly, dz = tq.cz - lz;
        const bool part = ingrid && ((todo >> lane) & 1ull) && dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
        todo &= ~__ballot(part);
        // bounding box of the participants' cells, +-1, clipped to the grid (uniform)
        int x0 = lx - 1 - (__ballot(part && dx < 0) ? 1 : 0), x1 = lx + 1 + (__ballot(part && dx > 0) ? 1 : 0);
        int y0 = ly - 1 - (__ballot(part && dy < 0) ? 1 : 0), y1 = ly + 1 + (__ballot(part && dy > 0) ? 1 : 0);
        int z0 = lz - 1 - (__ballot(part && dz < 0) ? 1 : 0), z1 = lz + 1 + (__ballot(part && dz > 0) ? 1 : 0);
        x0 = x0 < 0 ? 0 : x0;
        y0 = y0 < 0 ? 0 : y0;
        z0 = z0 < 0 ? 0 : z0;
        x1 = x1 >= g.nx ? g.nx - 1 : x1;
        y1 = y1 >= g.ny ? g.ny - 1 : y1;
        z1 = z1 >= g.nz ? g.nz - 1 : z1;
        const int nyt = y1 - y0 + 1, nrows = nyt * (z1 - z0 + 1);  // <= LL_TILE_MAX_ROWS
        // row table: lane r holds row r's first candidate and count; inclusive prefix sums over the rows
        int rb = 0, cnt = 0;
        if (lane < nrows) {
            const int rz = (lane >= nyt) + (lane >= 2 * nyt) + (lane >= 3 * nyt) + (lane >= 4 * nyt), ry = lane - rz * nyt;  // lane / nyt, nyt <= 5
            const int base = ((z0 + rz) * g.ny + (y0 + ry)) * g.nx;
            rb = g.cell_start[base + x0];
            cnt = g.cell_start[base + x1 + 1] - rb;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        const int T = __builtin_amdgcn_readlane(incl, LL_TILE_MAX_ROWS - 1);  // (rows beyond nrows count 0)
        const int excl = incl - cnt;
        LL_TT(1, tw);  // round set-up: participants, box, row table
        LL_TT_ADD(6, T);
        LL_TT_ADD(7, 1);
        Tile5 t;  // the round's exact result (participants); every lane starts a round empty
        tile5_init(t);
        bool collided = false;
        // LDS layout of a pass: candidates in PAIRS {x0, x1, y0, y1, z0, z1, bits(pos0), bits(pos1)} (32 bytes), so that a pair's
        // coordinates arrive as the two-float operands of the packed subtract / multiply / add: 8 instructions per two distances
        typedef float ll_tv2 __attribute__((ext_vector_type(2)));
        float *tile_f = reinterpret_cast<float *>(tile);
        const ll_tv2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        unsigned int keep;  // ~(LL_TILE_CAP - 1) in a register: (bits & keep) | index is then one v_and_or_b32 (a literal would not fit beside the scalar index)
        asm volatile("v_mov_b32 %0, 0xffffff00" : "=v"(keep));
        static_assert(LL_TILE_IDX_BITS == 8, "the literal above");
        for (int c0 = 0; c0 < T; c0 += LL_TILE_CAP) {  // (uniform) one pass per LL_TILE_CAP candidates: one pass for every C2 tile
            const int np = (T - c0) < LL_TILE_CAP ? (T - c0) : LL_TILE_CAP;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the previous pass's reads are done)
            for (int s0 = 0; s0 < np; s0 += 64) {  // stage 64 candidates per trip, one per lane, each lane finds its row
                const int j = c0 + s0 + lane;
                int row = 0;
                for (int rr = 0; rr < nrows - 1; rr++) row += (j >= __builtin_amdgcn_readlane(incl, rr)) ? 1 : 0;  // (scalar operand)
                const int addr = __shfl(rb, row) + (j - __shfl(excl, row));
                float ex = LL_TILE_FAR, ey = 0.0f, ez = 0.0f;  // padding: never among five real neighbours
                int ep = -1;
                if (j < T) {
                    const f4 pt = g.pts[addr];
                    ex = pt.x;
                    ey = pt.y;
                    ez = pt.z;
                    ep = addr;
                }
                float *e = tile_f + ((s0 + lane) >> 1) * 8 + (lane & 1);
                e[0] = ex;
                e[2] = ey;
                e[4] = ez;
                e[6] = __int_as_float(ep);
            }
            if (lane < 4) {  // (two more pairs of padding behind a full last trip)
                float *e = tile_f + ((((np + 63) & ~63) + lane) >> 1) * 8 + (lane & 1);
                e[0] = LL_TILE_FAR;
                e[2] = 0.0f;
                e[4] = 0.0f;
                e[6] = __int_as_float(-1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            LL_TT(2, tw);  // staging
            TileK tk;
            tilek_init(tk);
            for (int jj = 0; jj < np; jj += 4) {  // (uniform) two pairs per trip: broadcast reads, four independent offers
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const float4 xy = *reinterpret_cast<const float4 *>(tile_f + ((jj + u) >> 1) * 8);
                    const float2 zz = *reinterpret_cast<const float2 *>(tile_f + ((jj + u) >> 1) * 8 + 4);
                    const ll_tv2 px = {xy.x, xy.y}, py = {xy.z, xy.w}, pz = {zz.x, zz.y};
                    const ll_tv2 dx = qx2 - px, dy = qy2 - py, dz = qz2 - pz;  // dist2_xyz, two candidates at a time
                    ll_tv2 rr2 = dx * dx;
                    rr2 = rr2 + dy * dy;
                    rr2 = rr2 + dz * dz;
                    tilek_offer(tk, ((unsigned int)__float_as_int(rr2.x) & keep) | (unsigned int)(jj + u));
                    tilek_offer(tk, ((unsigned int)__float_as_int(rr2.y) & keep) | (unsigned int)(jj + u + 1));
                }
            }
            LL_TT(3, tw);  // offers
            collided = collided || tilek_collision(tk, max_d2);
            // the five winners again, exactly (their order is the true one unless `collided`)
            const float lbv = tile_key_lower(tk.k[5]);
            float wx[5], wy[5], wz[5];  // (k = 0, 2, 4 are used: the points of the query's plane, PCR:416-418)
#pragma unroll
            for (int k = 0; k < 5; k++) {
                float d = INFINITY;
                int pos = -1;
                wx[k] = wy[k] = wz[k] = 0.0f;
                if (tk.k[k] != LL_TILE_KEY_EMPTY) {
                    const int jw = (int)(tk.k[k] & (LL_TILE_CAP - 1));
                    const float *e = tile_f + (jw >> 1) * 8 + (jw & 1);
                    pos = __float_as_int(e[6]);
                    wx[k] = e[0], wy[k] = e[2], wz[k] = e[4];
                    d = pos >= 0 ? dist2_xyz(qx, qy, qz, wx[k], wy[k], wz[k]) : INFINITY;
                }
                if (c0 == 0) {  // (uniform) the first pass fills the list, later ones (tiles of > LL_TILE_CAP points) merge into it
                    t.d[k] = d;
                    t.p[k] = pos;
                } else {
                    tile5_offer(t, d, pos);
                }
            }
            t.lb = fminf(t.lb, lbv);
            // a one-pass tile: the list is these five, in this order (final lanes only: no collision, five real points)
            if (part && T <= LL_TILE_CAP)
                degenerate = ((wx[2] == wx[0] && wy[2] == wy[0] && wz[2] == wz[0]) || (wx[4] == wx[0] && wy[4] == wy[0] && wz[4] == wz[0])) ? 1 : 0;
        }
        if (part) final = !collided && tile5_finish<SHORT_LISTS>(g, t, tq, max_d2, r);
        LL_TT(4, tw);  // winners again + finish test
    }
}
#endif  // __HIPCC__

}  // namespace ll
