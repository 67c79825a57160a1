// tests/hostcheck/hostcheck.cpp -- TEST-ONLY host build of the device math headers.
//
// The per-thread arithmetic of the HIP kernels lives in loam_livox_amd/csrc/ll_*_core.h as
// __host__ __device__ inline functions.  This file compiles those headers with g++ and drives them with plain
// serial loops that stand in for the kernels' thread indexing, so that the no-GPU test tier can check the
// arithmetic (labels, 5-NN ring search, block construction, analytic Jacobians, LM controller) against the
// oracle.  It is NOT part of the product: libloamlivox_hip.so never links it and has no CPU path.  The
// GPU-only plumbing (LDS staging, wave reductions, hash de-duplication, radix select, stream compaction) is
// covered by the `-m gpu` tests.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

static long long g_knn_stat[4];  // row look-ups, candidates, queries reaching ring 2, queries reaching the cube sweep
static int g_knn_run = -1;       // counter 4: the run of the 3x3x3 block being scanned (per-run candidate counts below)
static int g_knn_run_cands[9];
#define LL_KNN_STAT(counter, n)                                                          \
    do {                                                                                 \
        if ((counter) == 4) g_knn_run = (int)(n);                                        \
        else {                                                                           \
            g_knn_stat[(counter) & 3] += (n);                                            \
            if ((counter) == 1 && g_knn_run >= 0) { g_knn_run_cands[g_knn_run] = (int)(n); g_knn_run = -1; } \
        }                                                                                \
    } while (0)
#include "../../loam_livox_amd/csrc/ll_fe_core.h"
#include "../../loam_livox_amd/csrc/ll_knn_core.h"
#include "../../loam_livox_amd/csrc/ll_knn_tile.h"
#include "../../loam_livox_amd/csrc/ll_reg_core.h"
#include "../../loam_livox_amd/csrc/ll_cellmap_core.h"
#include "../../loam_livox_amd/csrc/ll_voxel_core.h"

using namespace ll;

extern "C" {

struct hc_fe_params {
    float thr_corner_curvature, thr_surface_curvature, minimum_view_angle, livox_min_allow_dis, livox_min_sigma, max_fov,
        time_internal_pts;
};

static FeConst make_const(const hc_fe_params *p)
{
    FeConst c;
    c.thr_corner_curvature = p->thr_corner_curvature;
    c.thr_surface_curvature = p->thr_surface_curvature;
    c.minimum_view_angle = p->minimum_view_angle;
    c.min_dis_sq = p->livox_min_allow_dis * p->livox_min_allow_dis;
    c.min_sigma = p->livox_min_sigma;
    c.max_edge_polar_pos = (float)pow(tan((double)p->max_fov / 57.3) * 1, 2);
    c.time_internal_pts = p->time_internal_pts;
    c.view_angle_band = 0.0f;
    return c;
}

// stands in for fe_point_kernel
int hc_fe_points(const hc_fe_params *p, const float *xyzi, int n, double t0, int32_t *type, int32_t *label, float *depth2,
                 float *curv, float *view, float *tstamp, float *polar2_own, int32_t *flags)
{
    const FeConst c = make_const(p);
    std::vector<PointOwn> own(n);
    for (int i = 0; i < n; i++) {
        own[i] = point_own(xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3], i, c);
        depth2[i] = own[i].depth_sq2;
        polar2_own[i] = own[i].polar_sq2;
        flags[i] = own[i].defines | (own[i].reached << 1);
        tstamp[i] = point_time_stamp(t0, i, c.time_internal_pts);
    }
    auto edge = [&](int k) { return (k >= 0 && k < n) ? own[k].edge : 0; };
    for (int i = 0; i < n; i++) {
        type[i] = own[i].type_self | ((edge(i - 1) | edge(i + 1) | edge(i + 2)) ? PT_CIRCLE_EDGE : 0);
        LabelOut lo;
        lo.label = 0;
        lo.curvature = 0.f;
        lo.view_angle = 0.f;
        if (n >= 5 && i >= 2 && i < n - 2) {
            float pp[5][3], d[5];
            int t[5];
            for (int k = 0; k < 5; k++) {
                pp[k][0] = xyzi[4 * (i - 2 + k)];
                pp[k][1] = xyzi[4 * (i - 2 + k) + 1];
                pp[k][2] = xyzi[4 * (i - 2 + k) + 2];
                t[k] = own[i - 2 + k].type_self;
                d[k] = own[i - 2 + k].depth_sq2;
            }
            lo = point_label(pp, t, d, c);
        }
        label[i] = lo.label;
        curv[i] = lo.curvature;
        view[i] = lo.view_angle;
    }
    return 0;
}

int hc_select(int n, const int32_t *type, const int32_t *label, const float *depth2, float min_blur, float max_blur,
              int32_t *ci, int32_t *nc, int32_t *si, int32_t *ns, int32_t *fi, int32_t *nf)
{
    const float maximum_idx = max_blur * n, minimum_idx = min_blur * n;
    int c = 0, s = 0, f = 0;
    for (int i = 0; i < n; i++) {
        const int sel = select_point(i, type[i], label[i], depth2[i], minimum_idx, maximum_idx);
        if (sel & 1) ci[c++] = i;
        if (sel & 2) si[s++] = i;
        if (sel & 4) fi[f++] = i;
    }
    *nc = c;
    *ns = s;
    *nf = f;
    return 0;
}

// ---- grid (host stand-in for map_build) -----------------------------------------------------------------------
struct hc_grid {
    std::vector<f4> pts;
    std::vector<int> cell_start;
    Grid g;
};

hc_grid *hc_grid_build(const float *xyz, int stride, int64_t n, float cell)
{
    hc_grid *G = new hc_grid();
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; i++) {
        const float *p = xyz + i * stride;
        if (ll_isfinite(p[0]) && ll_isfinite(p[1]) && ll_isfinite(p[2]))
            for (int d = 0; d < 3; d++) {
                mn[d] = fminf(mn[d], p[d]);
                mx[d] = fmaxf(mx[d], p[d]);
            }
    }
    if (!(mn[0] <= mx[0])) mn[0] = mn[1] = mn[2] = mx[0] = mx[1] = mx[2] = 0.f;
    Grid &g = G->g;
    g.h = cell;
    g.inv_h = 1.0f / cell;
    g.ox = mn[0];
    g.oy = mn[1];
    g.oz = mn[2];
    g.nx = (int)floor((double)(mx[0] - mn[0]) / cell) + 1;
    g.ny = (int)floor((double)(mx[1] - mn[1]) / cell) + 1;
    g.nz = (int)floor((double)(mx[2] - mn[2]) / cell) + 1;
    const float ext = fmaxf(fmaxf(fabsf(mn[0]), fabsf(mx[0])), fmaxf(fmaxf(fabsf(mn[1]), fabsf(mx[1])), fmaxf(fabsf(mn[2]), fabsf(mx[2])))) +
                      fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
    g.slack = 1e-3f * cell + 2e-6f * ext;
    g.guard = 0.0f;
    const size_t ncell = (size_t)g.nx * g.ny * g.nz;
    std::vector<std::pair<unsigned, int>> kv;
    kv.reserve(n);
    for (int64_t i = 0; i < n; i++) {
        const float *p = xyz + i * stride;
        if (!(ll_isfinite(p[0]) && ll_isfinite(p[1]) && ll_isfinite(p[2]))) continue;
        int cx = std::min(std::max(cell_coord(p[0], g.ox, g.inv_h), 0), g.nx - 1);
        int cy = std::min(std::max(cell_coord(p[1], g.oy, g.inv_h), 0), g.ny - 1);
        int cz = std::min(std::max(cell_coord(p[2], g.oz, g.inv_h), 0), g.nz - 1);
        kv.emplace_back((unsigned)((cz * g.ny + cy) * g.nx + cx), (int)i);
    }
    std::stable_sort(kv.begin(), kv.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    G->pts.resize(kv.size());
    G->cell_start.assign(ncell + 1, 0);
    for (size_t j = 0; j < kv.size(); j++) {
        const float *p = xyz + (int64_t)kv[j].second * stride;
        f4 q;
        q.x = p[0];
        q.y = p[1];
        q.z = p[2];
        union {
            int i;
            float f;
        } u;
        u.i = kv[j].second;
        q.w = u.f;
        G->pts[j] = q;
        G->cell_start[kv[j].first + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) G->cell_start[c + 1] += G->cell_start[c];
    g.pts = G->pts.data();
    g.cell_start = G->cell_start.data();
    return G;
}
void hc_grid_free(hc_grid *G) { delete G; }
// reuse guard band of the search (ll_knn_core.h Grid::guard; the registrar runs with 0.05 m)
int hc_grid_set_guard(hc_grid *G, float guard)
{
    G->g.guard = guard;
    return 0;
}

/* the three-sample line-search interpolation of ll_reg_core.h, for the check against a dense Vandermonde solve */
// scaled plane blocks (ll_reg_core.h plane_scale / plane_accumulate_scaled / plane_unfold2 / plane_l1_scaled) against the un-scaled forms:
// out[0..27] scaled accumulators (unfolded), out[28..55] block_accumulate's, out[56] scaled L1, out[57] block_l1
void hc_plane_scaled(const double *R, const double *t, const double *f, const double *v, double a0, double huber_a, const double *q_last, double *out)
{
    double m[3], beta, acc[LL_NACC], ref[LL_NACC];
    for (int i = 0; i < LL_NACC; i++) acc[i] = ref[i] = 0.0;
    ll::plane_scale(v, a0, m, &beta);
    ll::plane_accumulate_scaled(R, t, f, m, beta, huber_a, acc);
    ll::plane_unfold2(acc);
    const double a[3] = {a0, 0.0, 0.0};
    ll::block_accumulate(ll::BLK_PLANE, R, t, f, a, v, huber_a, ref);
    for (int i = 0; i < LL_NACC; i++) out[i] = acc[i], out[LL_NACC + i] = ref[i];
    out[2 * LL_NACC] = ll::plane_l1_scaled(R, t, f, m, beta, huber_a, q_last);
    out[2 * LL_NACC + 1] = ll::block_l1(ll::BLK_PLANE, R, t, f, a, v, huber_a, q_last);
}

double hc_quintic_min_step(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo, double hi)
{
    return ll::lm_quintic_min_step(f0, g0, x1, f1, g1, x2, f2, g2, lo, hi);
}

int hc_knn5(const hc_grid *G, const float *q, int nq, float max_d2, int32_t *idx, float *d2)
{
    for (int i = 0; i < nq; i++) {
        Knn5 r;
        knn5_search(G->g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
        for (int k = 0; k < 5; k++) {
            idx[5 * i + k] = (knn5_idx(r, k) == LL_KNN_EMPTY) ? -1 : knn5_idx(r, k);
            d2[5 * i + k] = knn5_d2(r, k);
        }
    }
    return 0;
}

// per-query work of the search: rows looked up, candidates examined, deepest phase reached (1 = 3x3x3 block, 2 = rings, 3 = cube sweep)
int hc_knn5_work(const hc_grid *G, const float *q, int nq, float max_d2, int32_t *rows, int32_t *cands, int32_t *phase)
{
    for (int i = 0; i < nq; i++) {
        long long before[4];
        for (int k = 0; k < 4; k++) before[k] = g_knn_stat[k];
        Knn5 r;
        knn5_search(G->g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
        rows[i] = (int32_t)(g_knn_stat[0] - before[0]);
        cands[i] = (int32_t)(g_knn_stat[1] - before[1]);
        phase[i] = g_knn_stat[3] > before[3] ? 3 : (g_knn_stat[2] > before[2] ? 2 : 1);
    }
    return 0;
}

// candidates examined per run of the 3x3x3 block (-1: run pruned or outside the grid), [nq][9]: input of the SIMT schedule model in
// tools/knn_simt_model.py
int hc_knn5_run_cands(const hc_grid *G, const float *q, int nq, float max_d2, int32_t *cands9)
{
    for (int i = 0; i < nq; i++) {
        for (int k = 0; k < 9; k++) g_knn_run_cands[k] = -1;
        g_knn_run = -1;
        Knn5 r;
        knn5_search(G->g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
        for (int k = 0; k < 9; k++) cands9[9 * i + k] = g_knn_run_cands[k];
    }
    return 0;
}

#define HC_KNN_K 5  // candidates the search keeps (ll_knn_core.h, Knn5)
int hc_knn_k(void) { return HC_KNN_K; }

// the reuse bounds of the same search: the full candidate list, lb2 (lower bound on every point outside the list), out2
// (lower bound on every point at or beyond the match radius) and the two displacement budgets of the reuse record
int hc_knn5_bounds(const hc_grid *G, const float *q, int nq, float max_d2, int32_t *cand, float *lb2, float *out2, float *m_set,
                   float *m_strong)
{
    for (int i = 0; i < nq; i++) {
        Knn5 r;
        knn5_search(G->g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r);
        KnnRef ref;
        knn5_make_ref(r, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, ref);
        for (int k = 0; k < HC_KNN_K; k++) cand[HC_KNN_K * i + k] = (knn5_idx(r, k) == LL_KNN_EMPTY) ? -1 : knn5_idx(r, k);
        lb2[i] = r.lb2;
        out2[i] = r.out2;
        m_set[i] = ref.m_set;
        m_strong[i] = ref.m_strong;
    }
    return 0;
}

// The registrar's reuse logic over a chain of query positions (reg_requery_kernel): path[h][i] = position of query i at
// hop h.  Hop 0 searches; every later hop keeps the result (state 0), re-sorts the stored candidates (1) or searches
// again (2).  idx5 / state: [n_hops][nq][5] / [n_hops][nq]; idx -1 when fewer than five neighbours are inside the radius.
int hc_knn5_reuse_chain(const hc_grid *G, const float *path, int n_hops, int nq, float max_d2, int32_t *idx5, int32_t *state)
{
    for (int i = 0; i < nq; i++) {
        KnnRef ref{};
        int cur[5] = {-1, -1, -1, -1, -1};
        for (int h = 0; h < n_hops; h++) {
            const float *p = path + ((size_t)h * nq + i) * 3;
            int st = 2;
            Knn5 r;
            if (h > 0) {
                const float delta = knn5_ref_delta(ref, p[0], p[1], p[2]);
                if (delta < ref.m_strong) st = 0;
                else if (delta < ref.m_set) st = 1;
                if (st == 1) knn5_resort(G->g, ref, delta, p[0], p[1], p[2], max_d2, r);
            }
            if (st == 2) {
                knn5_search(G->g, p[0], p[1], p[2], max_d2, r);
                knn5_make_ref(r, p[0], p[1], p[2], max_d2, ref);
            }
            if (st != 0)
                for (int k = 0; k < 5; k++) cur[k] = (r.count >= 5) ? knn5_idx(r, k) : -1;
            for (int k = 0; k < 5; k++) idx5[((size_t)h * nq + i) * 5 + k] = cur[k];
            state[(size_t)h * nq + i] = st;
        }
    }
    return 0;
}

// Host model of knn5_tile_wave (ll_knn_tile.h): the queries are taken 64 at a time in the order given (the kernel's order is the
// cell order of reg_qsort_kernel); per "wavefront" the rounds, tiles and candidate order of the device code, the per-lane
// arithmetic from the shared header (tile_query / tile5_offer / tile5_finish); lanes the tile does not settle run knn5_search.
// idx: original indices (-1 where fewer than five inside the radius); stats[0] rounds, [1] candidates staged, [2] lanes that
// fell back, [3] wavefronts.
int hc_knn5_tile(const hc_grid *G, const float *q, int nq, float max_d2, int32_t *idx, float *d2, float *lb2, int64_t *stats)
{
    const Grid &g = G->g;
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    for (int w0 = 0; w0 < nq; w0 += 64) {
        const int nl = nq - w0 < 64 ? nq - w0 : 64;
        stats[3]++;
        TileQ tq[64];
        Tile5 t[64];
        Knn5 r[64];
        bool fin[64], todo[64];
        for (int l = 0; l < nl; l++) {
            tile_query(g, q[3 * (w0 + l)], q[3 * (w0 + l) + 1], q[3 * (w0 + l) + 2], tq[l]);
            tile5_init(t[l]);
            fin[l] = false;
            todo[l] = tq[l].ingrid;
        }
        for (;;) {
            int leader = -1;
            for (int l = 0; l < nl; l++)
                if (todo[l]) { leader = l; break; }
            if (leader < 0) break;
            stats[0]++;
            const int lx = tq[leader].cx, ly = tq[leader].cy, lz = tq[leader].cz;
            bool part[64];
            int x0 = lx - 1, x1 = lx + 1, y0 = ly - 1, y1 = ly + 1, z0 = lz - 1, z1 = lz + 1;
            for (int l = 0; l < nl; l++) {
                const int dx = tq[l].cx - lx, dy = tq[l].cy - ly, dz = tq[l].cz - lz;
                part[l] = todo[l] && dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                if (!part[l]) continue;
                todo[l] = false;
                if (dx < 0) x0 = lx - 2;
                if (dx > 0) x1 = lx + 2;
                if (dy < 0) y0 = ly - 2;
                if (dy > 0) y1 = ly + 2;
                if (dz < 0) z0 = lz - 2;
                if (dz > 0) z1 = lz + 2;
            }
            x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; z0 = z0 < 0 ? 0 : z0;
            x1 = x1 >= g.nx ? g.nx - 1 : x1; y1 = y1 >= g.ny ? g.ny - 1 : y1; z1 = z1 >= g.nz ? g.nz - 1 : z1;
            if ((y1 - y0 + 1) * (z1 - z0 + 1) > LL_TILE_MAX_ROWS) return -1;
            std::vector<f4> cand;  // the tile in the device's staging order: rows y fastest, then z; w = bits(position)
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++) {
                    const int base = (z * g.ny + y) * g.nx;
                    for (int j = g.cell_start[base + x0]; j < g.cell_start[base + x1 + 1]; j++) {
                        f4 e = g.pts[j];
                        union { int i; float f; } u;
                        u.i = j;
                        e.w = u.f;
                        cand.push_back(e);
                    }
                }
            stats[1] += (int64_t)cand.size();
            const int T = (int)cand.size();
            for (int l = 0; l < nl; l++) {
                if (!part[l]) continue;
                const float qx = q[3 * (w0 + l)], qy = q[3 * (w0 + l) + 1], qz = q[3 * (w0 + l) + 2];
                tile5_init(t[l]);
                bool collided = false;
                for (int c0 = 0; c0 < T; c0 += LL_TILE_CAP) {  // passes of LL_TILE_CAP candidates, as on the device
                    const int np = T - c0 < LL_TILE_CAP ? T - c0 : LL_TILE_CAP;
                    TileK tk;
                    tilek_init(tk);
                    const int np4 = (np + 3) & ~3;  // the device offers whole groups of four; the surplus are padding entries
                    for (int jj = 0; jj < np4; jj++) {
                        float dd;
                        if (jj < np) dd = dist2_xyz(qx, qy, qz, cand[c0 + jj].x, cand[c0 + jj].y, cand[c0 + jj].z);
                        else dd = dist2_xyz(qx, qy, qz, 1.0e18f, 0.0f, 0.0f);
                        tilek_offer(tk, tile_key(dd, jj));
                    }
                    collided = collided || tilek_collision(tk, max_d2);
                    const float lbv = tile_key_lower(tk.k[5]);
                    for (int k = 0; k < 5; k++) {
                        float d = INFINITY;
                        int pos = -1;
                        if (tk.k[k] != LL_TILE_KEY_EMPTY) {
                            const int jj = (int)(tk.k[k] & (LL_TILE_CAP - 1));
                            if (jj < np) {
                                pos = as_int(cand[c0 + jj].w);
                                d = dist2_xyz(qx, qy, qz, cand[c0 + jj].x, cand[c0 + jj].y, cand[c0 + jj].z);
                            }
                        }
                        if (c0 == 0) {
                            t[l].d[k] = d;
                            t[l].p[k] = pos;
                        } else {
                            tile5_offer(t[l], d, pos);
                        }
                    }
                    t[l].lb = fminf(t[l].lb, lbv);
                }
                fin[l] = !collided && tile5_finish(g, t[l], tq[l], max_d2, r[l]);
            }
        }
        for (int l = 0; l < nl; l++) {
            const int i = w0 + l;
            if (fin[l]) {
                for (int k = 0; k < r[l].count; k++) r[l].idx[k] = as_int(g.pts[r[l].pos[k]].w);
            } else {
                stats[2]++;
                knn5_search(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], max_d2, r[l]);
            }
            for (int k = 0; k < 5; k++) {
                idx[5 * i + k] = (knn5_idx(r[l], k) == LL_KNN_EMPTY) ? -1 : knn5_idx(r[l], k);
                d2[5 * i + k] = knn5_d2(r[l], k);
            }
            lb2[i] = r[l].lb2;
        }
    }
    return 0;
}

// ---- registration (host stand-in for reg_knn_build_kernel + reg_solve_kernel + reg_finalize_kernel) ----------
struct hc_reg_params {
    int if_motion_deblur, icp_max_iterations, ceres_max_iterations, ceres_prerun_times, icp_line, icp_plane,
        force_all_iterations;
    double max_d2_line, max_d2_plane, huber_a, inliner_dis, inlier_ratio, minimum_icp_R_diff, minimum_icp_T_diff, bound;
    float para_max_angular_rate, max_final_cost, min_ts, max_ts;
    int check_line_pca, check_plane_pca;
    int max_blocks, subsample_seed;  // a13
};

struct hc_blk {
    int kind;
    int active;
    int qidx;  // position of the block's feature in the corner-then-surface order (a13 block stream)
    double f[3], a[3], v[3];
    double s;
};

static int g_deblur = 0;  // set by the entry points below

static void eval_all(const std::vector<hc_blk> &blk, const double x[7], double huber_a, double acc[LL_NACC])
{
    double R[9], t[3] = {x[4], x[5], x[6]};
    quat_to_mat(x, R);
    MbRot mb;
    if (g_deblur) mb_prepare(x, mb);
    for (int i = 0; i < LL_NACC; i++) acc[i] = 0.0;
    for (const hc_blk &b : blk) {
        if (!b.active) continue;
        if (g_deblur)
            block_accumulate_mb(b.kind, mb, t, b.s, b.f, b.a, b.v, huber_a, acc);
        else
            block_accumulate(b.kind, R, t, b.f, b.a, b.v, huber_a, acc);
    }
}

static void lm_run(const std::vector<hc_blk> &blk, const double x0[7], int max_iter, double bound, double huber_a, LmCtl &c)
{
    int n_active = 0;
    for (const hc_blk &b : blk) n_active += b.active;
    double acc[LL_NACC];
    lm_begin(c, x0, max_iter, bound);
    eval_all(blk, c.x, huber_a, acc);
    int need = lm_init(c, acc, n_active);
    while (need) {
        eval_all(blk, c.cand, huber_a, acc);
        need = lm_update(c, acc);
    }
}

// blocks evaluation only (for Jacobian checks): kind[], f[3n], a[3n], v[3n] in the pose_last frame
int hc_eval_blocks(int n, const int32_t *kind, const double *f, const double *a, const double *v, const double *x, double huber_a,
                   double *acc28, int deblur, const double *sblur)
{
    g_deblur = deblur;
    std::vector<hc_blk> blk(n);
    for (int i = 0; i < n; i++) {
        blk[i].kind = kind[i];
        blk[i].active = 1;
        blk[i].s = sblur ? sblur[i] : 1.0;
        for (int k = 0; k < 3; k++) {
            blk[i].f[k] = f[3 * i + k];
            blk[i].a[k] = a[3 * i + k];
            blk[i].v[k] = v[3 * i + k];
        }
        // hc_make_block already returns plane blocks in their stored form (a[0] = n'.a')
    }
    eval_all(blk, x, huber_a, acc28);
    return 0;
}

// world-frame neighbours -> block constants in the pose_last frame
int hc_make_block(int kind, const double *pose_last, const double *pa, const double *pb, const double *pc, double *a_out, double *v_out)
{
    if (kind == BLK_LINE) return block_line(pose_last, pa, pb, a_out, v_out) ? 1 : 0;
    return block_plane(pose_last, pa, pb, pc, a_out, v_out) ? 1 : 0;
}

int hc_pca_check(int is_plane, const float *pts, double *ev_out)
{
    double p5[5][3], center[3] = {0, 0, 0}, cov[9] = {0};
    for (int j = 0; j < 5; j++)
        for (int c = 0; c < 3; c++) {
            p5[j][c] = pts[j * 3 + c];
            center[c] += p5[j][c];
        }
    for (int c = 0; c < 3; c++) center[c] /= 5.0;
    for (int j = 0; j < 5; j++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) cov[r * 3 + c] += (p5[j][r] - center[r]) * (p5[j][c] - center[c]);
    sym3_eigenvalues(cov, ev_out);
    return pca_check(is_plane, p5) ? 1 : 0;
}

int hc_reg_solve(const hc_grid *gc, const hc_grid *gs, const float *corner, int nC, const float *surf, int nS,
                 const hc_reg_params *p, const double *pose_last, double *pose_curr, double *inc, double *report /*[10]*/)
{
    double prev_q[4] = {0, 0, 0, 1}, prev_t[3] = {0, 0, 0};
    std::vector<KnnRef> refs[2];
    refs[0].resize(nC);
    refs[1].resize(nS);
    std::vector<hc_blk> kept[2];
    kept[0].resize(nC);
    kept[1].resize(nS);
    for (auto &v : kept) for (auto &k : v) k.kind = BLK_NONE;
    std::vector<char> kept_avail(nS, 0);
    long n_reused = 0, n_searched = 0;
    g_deblur = p->if_motion_deblur;
    double interp_theta = 0.0, hat[9] = {0}, hat_sq[9] = {0};
    double final_cost = 0, initial_cost = 0, inlier_thr = 0, angular_diff = 0, t_diff = 0;
    int icp_iters = 0, n_blocks_last = 0, corner_avail = 0, surf_avail = 0, lm_total = 0;
    float fl = (float)p->max_d2_line, fp = (float)p->max_d2_plane;
    if ((double)fl < p->max_d2_line) fl = nextafterf(fl, INFINITY);
    if ((double)fp < p->max_d2_plane) fp = nextafterf(fp, INFINITY);
    for (int it = 0; it < p->icp_max_iterations; it++) {
        std::vector<hc_blk> blk;
        corner_avail = surf_avail = 0;
        long st_cnt[3] = {0, 0, 0};
        for (int kind = 0; kind < 2; kind++) {
            const int n = kind ? nS : nC;
            const float *feat = kind ? surf : corner;
            const hc_grid *G = kind ? gs : gc;
            for (int q = 0; q < n; q++) {
                const float *f = feat + 4 * q;
                if (!(ll_isfinite(f[0]) && ll_isfinite(f[1]) && ll_isfinite(f[2]))) continue;
                if (subsample_skip_feature((unsigned int)p->subsample_seed, kind, it, q, n, p->max_blocks)) continue;  // PCR:232-238, 339-345
                float pw[3];
                const float sblur = refine_blur(p->if_motion_deblur, f[3], p->min_ts, p->max_ts);
                if (p->if_motion_deblur == 0 || (double)sblur == 1.0) {
                    point_to_map(pose_curr, f[0], f[1], f[2], pw);
                } else {  // Rodrigues interpolation, PCR:641-646
                    const double sd = (double)sblur;
                    const double T[3] = {inc[4] * (sd * 1.0), inc[5] * (sd * 1.0), inc[6] * (sd * 1.0)};
                    const double th = interp_theta * sd, sn = sin(th), cs1 = 1.0 - cos(th);
                    const double pc3[3] = {(double)f[0], (double)f[1], (double)f[2]};
                    double inner[3], o[3];
                    for (int i = 0; i < 3; i++) {
                        double acc = 0.0;
                        for (int j = 0; j < 3; j++) acc += (((i == j) ? 1.0 : 0.0) + sn * hat[i * 3 + j] + cs1 * hat_sq[i * 3 + j]) * pc3[j];
                        inner[i] = acc + T[i];
                    }
                    quat_rot(pose_last, inner, o);
                    pw[0] = (float)(o[0] + pose_last[4]);
                    pw[1] = (float)(o[1] + pose_last[5]);
                    pw[2] = (float)(o[2] + pose_last[6]);
                }
                Knn5 r;
                const float md2 = kind ? fp : fl;
                KnnRef &ref = refs[kind][q];
                int state = 2;  // 0 = stable (nothing recomputed), 1 = re-sorted, 2 = searched
                if (it > 0) {
                    const float delta = knn5_ref_delta(ref, pw[0], pw[1], pw[2]);
                    if (delta < ref.m_strong) state = 0;
                    else if (delta < ref.m_set) {
                        knn5_resort(G->g, ref, delta, pw[0], pw[1], pw[2], md2, r);
                        state = 1;
                    }
                }
                if (state == 2) {
                    knn5_search(G->g, pw[0], pw[1], pw[2], md2, r);
                    knn5_make_ref(r, pw[0], pw[1], pw[2], md2, ref);
                    n_searched++;
                } else {
                    n_reused++;
                }
                st_cnt[state]++;
                if (state == 0) {
                    // unchanged: re-use the block built earlier for this query (if it had one)
                    if (kept[kind][q].kind != BLK_NONE) {
                        hc_blk kb = kept[kind][q];
                        kb.active = 1;
                        kb.qidx = (kind ? nC : 0) + q;
                        blk.push_back(kb);
                        if (kind == 0) corner_avail++;
                    }
                    if (kind == 1 && kept_avail[q]) surf_avail++;
                    continue;
                }
                kept[kind][q].kind = BLK_NONE;
                if (kind == 1) kept_avail[q] = 0;
                if (r.count != 5) continue;
                if (kind ? p->check_plane_pca : p->check_line_pca) {
                    double pts5[5][3];
                    for (int j = 0; j < 5; j++) {
                        const f4 pj = G->g.pts[r.pos[j]];
                        pts5[j][0] = pj.x;
                        pts5[j][1] = pj.y;
                        pts5[j][2] = pj.z;
                    }
                    if (!pca_check(kind, pts5)) continue;
                }
                hc_blk b;
                b.active = 1;
                b.qidx = (kind ? nC : 0) + q;
                b.s = p->if_motion_deblur ? (double)sblur : 1.0;
                b.f[0] = f[0];
                b.f[1] = f[1];
                b.f[2] = f[2];
                if (kind == 0) {
                    if (!p->icp_line) continue;
                    const f4 p0 = G->g.pts[r.pos[0]], p1 = G->g.pts[r.pos[1]];
                    const double pa[3] = {p0.x, p0.y, p0.z}, pb[3] = {p1.x, p1.y, p1.z};
                    if (!block_line(pose_last, pa, pb, b.a, b.v)) continue;
                    b.kind = BLK_LINE;
                    blk.push_back(b);
                    kept[0][q] = b;
                    corner_avail++;
                } else {
                    if (p->icp_plane) {
                        const f4 p0 = G->g.pts[r.pos[0]], p1 = G->g.pts[r.pos[2]], p2 = G->g.pts[r.pos[4]];
                        const double pa[3] = {p0.x, p0.y, p0.z}, pb[3] = {p1.x, p1.y, p1.z}, pc[3] = {p2.x, p2.y, p2.z};
                        if (!block_plane(pose_last, pa, pb, pc, b.a, b.v)) continue;
                        b.kind = BLK_PLANE;
                        blk.push_back(b);
                        kept[1][q] = b;
                    }
                    kept_avail[q] = 1;
                    surf_avail++;
                }
            }
        }
        if (getenv("HC_KNN_STATS")) fprintf(stderr, "iter %d: stable %ld resort %ld search %ld\n", it, st_cnt[0], st_cnt[1], st_cnt[2]);
        if (p->subsample_seed && (int)blk.size() > p->max_blocks) {  // PCR:438-458
            const int nb = (int)blk.size();
            for (hc_blk &b : blk)
                if (subsample_drop_block((unsigned int)p->subsample_seed, it, b.qidx, nb, p->max_blocks)) b.active = 0;
        }
        LmCtl c;
        lm_run(blk, inc, p->ceres_prerun_times, p->bound, p->huber_a, c);
        int lm_iters = c.iteration;
        // L1 + std::set threshold
        {
            double R[9], t[3] = {c.x[4], c.x[5], c.x[6]};
            quat_to_mat(c.x, R);
            std::vector<double> l1(blk.size());
            MbRot mb;
            if (g_deblur) mb_prepare(c.x, mb);
            for (size_t i = 0; i < blk.size(); i++)
                l1[i] = g_deblur ? block_l1_mb(blk[i].kind, mb, t, blk[i].s, blk[i].f, blk[i].a, blk[i].v, p->huber_a, pose_last)
                                 : block_l1(blk[i].kind, R, t, blk[i].f, blk[i].a, blk[i].v, p->huber_a, pose_last);
            std::vector<double> u;  // blocks dropped by the sub-sampling are no longer part of the problem
            for (size_t i = 0; i < blk.size(); i++)
                if (blk[i].active && l1[i] == l1[i]) u.push_back(l1[i]);
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            double thr = p->inliner_dis;
            if (!u.empty()) {
                int target = (int)(p->inlier_ratio * (double)u.size());
                if (target > (int)u.size() - 1) target = (int)u.size() - 1;
                thr = fmax(p->inliner_dis, u[target]);
            }
            inlier_thr = thr;
            for (size_t i = 0; i < blk.size(); i++)
                if (blk[i].active && l1[i] > thr) blk[i].active = 0;
        }
        double xs[7];
        for (int i = 0; i < 7; i++) xs[i] = c.x[i];
        lm_run(blk, xs, p->ceres_max_iterations, p->bound, p->huber_a, c);
        lm_iters += c.iteration;
        n_blocks_last = 0;
        for (const hc_blk &b : blk) n_blocks_last += b.active;
        for (int i = 0; i < 7; i++) inc[i] = c.x[i];
        if (p->if_motion_deblur) {  // compute_interpolatation_rodrigue, PCR:607-620
            double n = sqrt(inc[0] * inc[0] + inc[1] * inc[1] + inc[2] * inc[2]), axis[3];
            if (inc[3] < 0) n = -n;
            if (n != 0.0) {
                interp_theta = 2.0 * atan2(n, fabs(inc[3]));
                for (int i = 0; i < 3; i++) axis[i] = inc[i] / n;
            } else {
                interp_theta = 0.0;
                axis[0] = 1.0;
                axis[1] = axis[2] = 0.0;
            }
            const double an = sqrt(dot3(axis, axis));
            for (int i = 0; i < 3; i++) axis[i] /= an;
            for (int i = 0; i < 9; i++) hat[i] = 0.0;
            hat[1] = -axis[2]; hat[3] = axis[2]; hat[2] = axis[1]; hat[6] = -axis[1]; hat[5] = -axis[0]; hat[7] = axis[0];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double sacc = 0;
                    for (int k = 0; k < 3; k++) sacc += hat[i * 3 + k] * hat[k * 3 + j];
                    hat_sq[i * 3 + j] = sacc;
                }
        }
        double tw[3], qc[4];
        quat_rot(pose_last, &inc[4], tw);
        pose_curr[4] = tw[0] + pose_last[4];
        pose_curr[5] = tw[1] + pose_last[5];
        pose_curr[6] = tw[2] + pose_last[6];
        quat_mul(pose_last, inc, qc);
        for (int i = 0; i < 4; i++) pose_curr[i] = qc[i];
        angular_diff = (double)((float)quat_angular_distance(qc, pose_last)) * 57.3;
        const double dt[3] = {pose_curr[4] - pose_last[4], pose_curr[5] - pose_last[5], pose_curr[6] - pose_last[6]};
        t_diff = sqrt(dot3(dt, dt));
        final_cost = c.final_cost;
        initial_cost = c.initial_cost;
        lm_total += lm_iters;
        icp_iters++;
        const double dto[3] = {prev_t[0] - inc[4], prev_t[1] - inc[5], prev_t[2] - inc[6]};
        const bool conv = quat_angular_distance(prev_q, inc) < 57.3 * p->minimum_icp_R_diff && sqrt(dot3(dto, dto)) < p->minimum_icp_T_diff;
        if (conv && !p->force_all_iterations) break;
        for (int i = 0; i < 4; i++) prev_q[i] = inc[i];
        for (int i = 0; i < 3; i++) prev_t[i] = inc[4 + i];
    }
    int result = 1;
    if (icp_iters > 0) {
        inlier_thr = inlier_thr * final_cost / initial_cost;
        if (angular_diff > (double)p->para_max_angular_rate || (float)final_cost > p->max_final_cost) {
            for (int i = 0; i < 7; i++) pose_curr[i] = pose_last[i];
            result = 0;
        }
    }
    report[0] = final_cost;
    report[1] = initial_cost;
    report[2] = inlier_thr;
    report[3] = icp_iters;
    report[4] = n_blocks_last;
    report[5] = corner_avail;
    report[6] = surf_avail;
    report[7] = lm_total;
    report[8] = (double)n_reused;
    report[9] = (double)n_searched;
    return result;
}

// ------------------------------------------------------------------------------------------------------ cell map
// Serial stand-in for ll_cellmap_kernels.hip: the same store (points ordered by cell key, insertion order), the same
// key constructions and the same per-thread functions of ll_cellmap_core.h; std::stable_sort plays the radix sort.
struct hc_cellmap {
    CellGeom g;
    int thr, frame;
    std::vector<float> pts;  // xyz0 per point
    std::vector<unsigned long long> pkey, ckey;
    std::vector<int> cstart, clast;
    std::vector<float> filt;
    int n_sel;
};

hc_cellmap *hc_cellmap_create(float resolution, int thr)
{
    hc_cellmap *m = new hc_cellmap();
    m->g = cell_geom(resolution);
    m->thr = thr;
    m->frame = 0;
    m->n_sel = 0;
    m->cstart.push_back(0);
    return m;
}
void hc_cellmap_free(hc_cellmap *m) { delete m; }

static int hc_cell_find(const std::vector<unsigned long long> &ckey, unsigned long long k)
{
    auto it = std::lower_bound(ckey.begin(), ckey.end(), k);
    return (it != ckey.end() && *it == k) ? (int)(it - ckey.begin()) : -1;
}

// cellmap_resort: stable sort by key, drop NONE, rebuild the table; the last n_appended entries stamp their cells
static void hc_cellmap_resort(hc_cellmap *m, std::vector<float> &pts, std::vector<unsigned long long> &pkey, int n_appended)
{
    const int total = (int)pkey.size();
    std::vector<int> order(total);
    for (int i = 0; i < total; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return pkey[a] < pkey[b]; });
    std::vector<float> p2;
    std::vector<unsigned long long> k2, ck;
    std::vector<int> cs, cl;
    for (int i = 0; i < total; i++) {
        const int o = order[i];
        if (pkey[o] == LL_CELL_KEY_NONE) break;
        if (k2.empty() || k2.back() != pkey[o]) {
            ck.push_back(pkey[o]);
            cs.push_back((int)k2.size());
            const int j = hc_cell_find(m->ckey, pkey[o]);
            cl.push_back(j >= 0 ? m->clast[j] : m->frame);
        }
        k2.push_back(pkey[o]);
        for (int d = 0; d < 4; d++) p2.push_back(pts[4 * (size_t)o + d]);
    }
    cs.push_back((int)k2.size());
    for (int i = total - n_appended; i < total; i++) {
        if (pkey[i] == LL_CELL_KEY_NONE) continue;
        const int c = hc_cell_find(ck, pkey[i]);
        if (c >= 0) cl[c] = m->frame;
    }
    m->pts.swap(p2);
    m->pkey.swap(k2);
    m->ckey.swap(ck);
    m->cstart.swap(cs);
    m->clast.swap(cl);
}

int hc_cellmap_append(hc_cellmap *m, const float *xyzi, int n)
{
    std::vector<float> pts = m->pts;
    std::vector<unsigned long long> pkey = m->pkey;
    std::vector<char> reset(m->ckey.size(), 0);
    const int n_old = (int)pkey.size();
    for (int i = 0; i < n; i++) {
        const float *p = xyzi + 4 * (size_t)i;
        int k[3];
        unsigned long long key = LL_CELL_KEY_NONE;
        if (ll_isfinite(p[0]) && ll_isfinite(p[1]) && ll_isfinite(p[2]) && cell_index(p[0], p[1], p[2], m->g, k)) {
            key = cell_pack(k);
            const int c = hc_cell_find(m->ckey, key);
            if (c >= 0 && !(m->frame - m->clast[c] < m->thr)) reset[c] = 1;
        }
        pts.push_back(p[0]);
        pts.push_back(p[1]);
        pts.push_back(p[2]);
        pts.push_back(0.0f);
        pkey.push_back(key);
    }
    for (int i = 0; i < n_old; i++) {
        const int c = hc_cell_find(m->ckey, pkey[i]);
        if (c >= 0 && reset[c]) pkey[i] = LL_CELL_KEY_NONE;
    }
    const bool was_empty = m->ckey.empty();  // the reference bumps its frame counter twice for the first cloud (CMK:615 and :667)
    if (n > 0) hc_cellmap_resort(m, pts, pkey, n);
    m->frame += was_empty ? 2 : 1;
    return 0;
}

// returns the number of filtered points (written to out_xyzi when it fits), -1 on a bad leaf
int hc_cellmap_query_filter(hc_cellmap *m, const double *pose, float radius, float max_fov, float leaf, int replace, float *out_xyzi, int cap,
                            int *n_sel_out)
{
    const float inv_leaf = 1.0f / leaf;
    if (!(cell_leaf_span(m->g, inv_leaf) < 1024.0f)) return -1;
    const int nc = (int)m->ckey.size(), np = (int)m->pkey.size();
    std::vector<unsigned> csel(nc), crank(nc);
    const double q[4] = {pose[0], pose[1], pose[2], pose[3]}, t[3] = {pose[4], pose[5], pose[6]};
    const float sp[3] = {(float)t[0], (float)t[1], (float)t[2]};
    unsigned acc = 0;
    for (int c = 0; c < nc; c++) {
        int k[3];
        cell_unpack(m->ckey[c], k);
        float ctr[3];
        cell_centre(k, m->g, ctr);
        csel[c] = (cell_in_radius(ctr, sp, radius) && cell_in_fov(ctr, q, t, (double)max_fov)) ? 1u : 0u;
        crank[c] = acc;
        acc += csel[c];
    }
    m->n_sel = (int)acc;
    if (n_sel_out) *n_sel_out = (int)acc;
    std::vector<unsigned long long> skey(np);
    std::vector<int> order(np);
    for (int i = 0; i < np; i++) {
        order[i] = i;
        unsigned long long key = LL_CELL_KEY_NONE;
        const int c = hc_cell_find(m->ckey, m->pkey[i]);
        if (c >= 0 && csel[c]) {
            int k[3];
            cell_unpack(m->pkey[i], k);
            int l[3];
            for (int d = 0; d < 3; d++) {
                l[d] = cell_leaf_local(m->pts[4 * (size_t)i + d], k[d], m->g, inv_leaf);
                l[d] = l[d] < 0 ? 0 : (l[d] > 1023 ? 1023 : l[d]);
            }
            key = ((unsigned long long)crank[c] << 30) | ((unsigned long long)l[2] << 20) | ((unsigned long long)l[1] << 10) | (unsigned long long)l[0];
        }
        skey[i] = key;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return skey[a] < skey[b]; });
    std::vector<float> filt;
    std::vector<unsigned long long> fkey;
    for (int i = 0; i < np;) {
        const unsigned long long k = skey[order[i]];
        if (k == LL_CELL_KEY_NONE) break;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        int cnt = 0, j = i;
        for (; j < np && skey[order[j]] == k; j++) {
            const float *p = &m->pts[4 * (size_t)order[j]];
            sx = sx + p[0];
            sy = sy + p[1];
            sz = sz + p[2];
            si = si + p[3];
            cnt++;
        }
        const float c = (float)cnt;
        filt.push_back(sx / c);
        filt.push_back(sy / c);
        filt.push_back(sz / c);
        filt.push_back(si / c);
        fkey.push_back(m->pkey[order[i]]);
        i = j;
    }
    const int nf = (int)fkey.size();
    if (out_xyzi && cap >= nf) memcpy(out_xyzi, filt.data(), filt.size() * sizeof(float));
    if (replace && nf > 0) {
        std::vector<float> pts = m->pts;
        std::vector<unsigned long long> pkey = m->pkey;
        for (int i = 0; i < np; i++) {
            const int c = hc_cell_find(m->ckey, pkey[i]);
            if (c >= 0 && csel[c]) pkey[i] = LL_CELL_KEY_NONE;
        }
        pts.insert(pts.end(), filt.begin(), filt.end());
        pkey.insert(pkey.end(), fkey.begin(), fkey.end());
        hc_cellmap_resort(m, pts, pkey, 0);
    }
    return nf;
}

int hc_cellmap_sizes(const hc_cellmap *m, int *n_cells, int *n_pts, int *frame)
{
    *n_cells = (int)m->ckey.size();
    *n_pts = (int)m->pkey.size();
    *frame = m->frame;
    return 0;
}

// stands in for cm_stats_kernel: out per cell = type, vec[3], mean[3], cov[6], eval[3] (16 floats, type as float)
int hc_cellmap_features(const hc_cellmap *m, float *out16)
{
    for (size_t c = 0; c < m->ckey.size(); c++) {
        int k[3];
        cell_unpack(m->ckey[c], k);
        float ctr[3];
        cell_centre(k, m->g, ctr);
        CellStats s;
        cell_stats(&m->pts[4 * (size_t)m->cstart[c]], 4, m->cstart[c + 1] - m->cstart[c], ctr, m->g.box, s);
        float *o = out16 + 16 * c;
        o[0] = (float)s.type;
        for (int d = 0; d < 3; d++) {
            o[1 + d] = s.vec[d];
            o[4 + d] = s.mean[d];
            o[13 + d] = s.eval[d];
        }
        for (int d = 0; d < 6; d++) o[7 + d] = s.cov[d];
    }
    return 0;
}

// stands in for cm_kf_centre / dist / pick / image kernels (serial; the PCA sums run in cell order)
int hc_cellmap_keyframe(const hc_cellmap *m, float roi_ratio, float *img, float *ratio4, float *R18, int32_t *nvec4, float *centre_range4)
{
    const int nc = (int)m->ckey.size();
    memset(img, 0, sizeof(float) * 4 * LL_KF_RES * LL_KF_RES);
    for (int i = 0; i < 4; i++) ratio4[i] = 0.f, nvec4[i] = 0, centre_range4[i] = 0.f;
    for (int i = 0; i < 18; i++) R18[i] = 0.f;
    if (nc == 0) return 0;
    std::vector<CellStats> st(nc);
    std::vector<float> ctrs(3 * (size_t)nc), dist(nc, 0.f);
    float sum[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < nc; c++) {
        int k[3];
        cell_unpack(m->ckey[c], k);
        cell_centre(k, m->g, &ctrs[3 * (size_t)c]);
        cell_stats(&m->pts[4 * (size_t)m->cstart[c]], 4, m->cstart[c + 1] - m->cstart[c], &ctrs[3 * (size_t)c], m->g.box, st[c]);
        for (int d = 0; d < 3; d++) sum[d] = sum[d] + ctrs[3 * (size_t)c + d];
    }
    const int use_roi = roi_ratio > 0.f;
    float range = 0.f;
    if (use_roi) {
        const float inv = (float)(1.0 / (double)(float)nc);
        float ctr[3];
        for (int d = 0; d < 3; d++) ctr[d] = centre_range4[d] = sum[d] * inv;
        for (int c = 0; c < nc; c++) {
            const float dx = ctrs[3 * (size_t)c] - ctr[0], dy = ctrs[3 * (size_t)c + 1] - ctr[1], dz = ctrs[3 * (size_t)c + 2] - ctr[2];
            dist[c] = sqrtf(dx * dx + dy * dy + dz * dz);
        }
        std::vector<float> srt = dist;
        std::sort(srt.begin(), srt.end());
        srt.erase(std::unique(srt.begin(), srt.end()), srt.end());
        range = srt[(size_t)ceilf((float)(srt.size() - 1) * roi_ratio)];
        centre_range4[3] = range;
    }
    float gk[2 * LL_KF_BLUR + 1];
    kf_gauss_kernel(gk);
    for (int roi = 0; roi < (use_roi ? 2 : 1); roi++) {
        double mm[6] = {1, 0, 0, 1, 0, 1};
        for (int c = 0; c < nc; c++) {
            if (st[c].type != CELL_FEATURE_PLANE || (roi && !(dist[c] < range))) continue;
            const float *v = st[c].vec;
            mm[0] += (double)(v[0] * v[0]);
            mm[1] += (double)(v[0] * v[1]);
            mm[2] += (double)(v[0] * v[2]);
            mm[3] += (double)(v[1] * v[1]);
            mm[4] += (double)(v[1] * v[2]);
            mm[5] += (double)(v[2] * v[2]);
        }
        double val[3], V[9];
        sym3_eigen(mm, val, V);
        float R[9];
        for (int k = 0; k < 3; k++) {
            R[k * 3 + 0] = (float)V[k * 3 + 2];
            R[k * 3 + 1] = (float)V[k * 3 + 1];
        }
        R[0 * 3 + 2] = R[1 * 3 + 0] * R[2 * 3 + 1] - R[2 * 3 + 0] * R[1 * 3 + 1];
        R[1 * 3 + 2] = R[2 * 3 + 0] * R[0 * 3 + 1] - R[0 * 3 + 0] * R[2 * 3 + 1];
        R[2 * 3 + 2] = R[0 * 3 + 0] * R[1 * 3 + 1] - R[1 * 3 + 0] * R[0 * 3 + 1];
        for (int e = 0; e < 9; e++) R18[9 * roi + e] = R[e];
        std::vector<int> hist(2 * LL_KF_RES * LL_KF_RES, 0);
        for (int c = 0; c < nc; c++) {
            const int type = st[c].type;
            if (type == CELL_FEATURE_SPHERE || (roi && !(dist[c] < range))) continue;
            const float *v = st[c].vec;
            float a[3];
            for (int j = 0; j < 3; j++) a[j] = (R[0 * 3 + j] * v[0] + R[1 * 3 + j] * v[1]) + R[2 * 3 + j] * v[2];
            int pi, ti;
            feature_direction(a, &pi, &ti);
            const int which = type == CELL_FEATURE_PLANE ? 1 : 0;
            hist[which * LL_KF_RES * LL_KF_RES + pi * LL_KF_RES + ti]++;
            nvec4[2 * roi + which]++;
        }
        for (int which = 0; which < 2; which++) {
            const int *h = &hist[which * LL_KF_RES * LL_KF_RES];
            int nz = 0;
            std::vector<float> src(LL_KF_RES * LL_KF_RES), tmp(LL_KF_RES * LL_KF_RES);
            for (int e = 0; e < LL_KF_RES * LL_KF_RES; e++) src[e] = (float)h[e], nz += h[e] >= 1;
            ratio4[2 * roi + which] = (float)nz / (float)(LL_KF_RES * LL_KF_RES);
            for (int e = 0; e < LL_KF_RES * LL_KF_RES; e++) {
                const int r = e / LL_KF_RES, col = e % LL_KF_RES;
                float x = 0.f;
                for (int k = 0; k < 2 * LL_KF_BLUR + 1; k++) x = x + gk[k] * src[r * LL_KF_RES + (col + k - LL_KF_BLUR + LL_KF_RES) % LL_KF_RES];
                tmp[e] = x;
            }
            float *dst = img + (size_t)(2 * roi + which) * LL_KF_RES * LL_KF_RES;
            for (int e = 0; e < LL_KF_RES * LL_KF_RES; e++) {
                const int r = e / LL_KF_RES, col = e % LL_KF_RES;
                float x = 0.f;
                for (int k = 0; k < 2 * LL_KF_BLUR + 1; k++) x = x + gk[k] * tmp[((r + k - LL_KF_BLUR + LL_KF_RES) % LL_KF_RES) * LL_KF_RES + col];
                dst[e] = x;
            }
        }
    }
    return 0;
}

int hc_cellmap_dump(const hc_cellmap *m, float *xyzi, int32_t *ijk, int32_t *start, int32_t *last)
{
    memcpy(xyzi, m->pts.data(), m->pts.size() * sizeof(float));
    for (size_t c = 0; c < m->ckey.size(); c++) {
        cell_unpack(m->ckey[c], ijk + 3 * c);
        last[c] = m->clast[c];
    }
    for (size_t c = 0; c < m->cstart.size(); c++) start[c] = m->cstart[c];
    return 0;
}

}  // extern "C"
