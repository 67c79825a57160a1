/*
 * ll_oracle_kdtree.c -- CPU ORACLE (test infrastructure only, see ll_oracle.h).
 *
 * Restates the behaviour of pcl::KdTreeFLANN<PointXYZI>::nearestKSearch as used at
 * point_cloud_registration.hpp:249,351 (PCL is an un-vendored dependency, absent here):
 * exact (eps = 0) k-nearest neighbours in 3-D (intensity ignored), squared L2 distance
 * accumulated in fp32 in x, y, z order (FLANN L2_Simple<float>), results sorted ascending.
 * Exact-distance ties are ordered by ascending point index (documented deviation: FLANN's
 * tie order depends on its private tree layout).
 *
 * The tree itself is ours (median-split k-d tree, leaf buckets); only the result matters.
 */
#include "ll_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_LEAF 12

typedef struct {
    float split;   /* split value (internal nodes) */
    int32_t dim;   /* 0..2 internal, -1 leaf */
    int32_t left;  /* internal: index of left child; leaf: begin in perm */
    int32_t right; /* internal: index of right child; leaf: end in perm */
} orc_node;

struct orc_kdtree {
    const float *xyz;
    int stride;
    int64_t m;
    int32_t *perm;
    orc_node *nodes;
    int32_t n_nodes, cap_nodes;
};

static inline float coord(const orc_kdtree *t, int32_t i, int d) { return t->xyz[(size_t)i * t->stride + d]; }

static int32_t new_node(orc_kdtree *t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (orc_node *)realloc(t->nodes, sizeof(orc_node) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

/* quickselect on perm[lo,hi) so that perm[k] holds the k-th smallest coordinate along d */
static void select_kth(orc_kdtree *t, int32_t lo, int32_t hi, int32_t k, int d)
{
    while (hi - lo > 1) {
        /* median of three pivot */
        int32_t mid = lo + (hi - lo) / 2;
        float a = coord(t, t->perm[lo], d), b = coord(t, t->perm[mid], d), c = coord(t, t->perm[hi - 1], d);
        float pivot = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
        int32_t i = lo, j = hi - 1;
        while (i <= j) {
            while (coord(t, t->perm[i], d) < pivot) i++;
            while (coord(t, t->perm[j], d) > pivot) j--;
            if (i <= j) {
                int32_t tmp = t->perm[i];
                t->perm[i] = t->perm[j];
                t->perm[j] = tmp;
                i++;
                j--;
            }
        }
        if (k <= j)
            hi = j + 1;
        else if (k >= i)
            lo = i;
        else
            return;
    }
}

static int32_t build_rec(orc_kdtree *t, int32_t lo, int32_t hi)
{
    int32_t id = new_node(t);
    if (hi - lo <= ORC_LEAF) {
        t->nodes[id].dim = -1;
        t->nodes[id].left = lo;
        t->nodes[id].right = hi;
        t->nodes[id].split = 0;
        return id;
    }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t i = lo; i < hi; i++)
        for (int d = 0; d < 3; d++) {
            float v = coord(t, t->perm[i], d);
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    int dim = 0;
    float ext = mx[0] - mn[0];
    for (int d = 1; d < 3; d++)
        if (mx[d] - mn[d] > ext) {
            ext = mx[d] - mn[d];
            dim = d;
        }
    int32_t mid = lo + (hi - lo) / 2;
    select_kth(t, lo, hi, mid, dim);
    float split = coord(t, t->perm[mid], dim);
    int32_t l = build_rec(t, lo, mid);
    int32_t r = build_rec(t, mid, hi);
    t->nodes[id].dim = dim;
    t->nodes[id].split = split;
    t->nodes[id].left = l;
    t->nodes[id].right = r;
    return id;
}

orc_kdtree *orc_kdtree_build(const float *xyz, int stride, int64_t m)
{
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    t->xyz = xyz;
    t->stride = stride;
    t->m = m;
    t->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    for (int64_t i = 0; i < m; i++)
        t->perm[i] = (int32_t)i;
    if (m > 0)
        build_rec(t, 0, (int32_t)m);
    return t;
}

void orc_kdtree_free(orc_kdtree *t)
{
    if (!t) return;
    free(t->perm);
    free(t->nodes);
    free(t);
}

typedef struct {
    int k, count;
    int32_t *idx;
    float *d2;
} topk;

/* lexicographic (d2, idx) ordered insertion */
static inline void topk_push(topk *r, float d2, int32_t idx)
{
    if (r->count == r->k) {
        float wd = r->d2[r->k - 1];
        if (d2 > wd || (d2 == wd && idx > r->idx[r->k - 1]))
            return;
    }
    int pos = (r->count < r->k) ? r->count : r->k - 1;
    while (pos > 0 && (r->d2[pos - 1] > d2 || (r->d2[pos - 1] == d2 && r->idx[pos - 1] > idx))) {
        r->d2[pos] = r->d2[pos - 1];
        r->idx[pos] = r->idx[pos - 1];
        pos--;
    }
    r->d2[pos] = d2;
    r->idx[pos] = idx;
    if (r->count < r->k) r->count++;
}

/* FLANN L2_Simple<float>: result = 0; result += diff*diff for x, y, z (fp32) */
static inline float dist2(const float *p, const float q[3])
{
    float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    float r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r;
}

static void search_rec(const orc_kdtree *t, int32_t id, const float q[3], topk *r)
{
    const orc_node *nd = &t->nodes[id];
    if (nd->dim < 0) {
        for (int32_t i = nd->left; i < nd->right; i++) {
            int32_t pi = t->perm[i];
            topk_push(r, dist2(&t->xyz[(size_t)pi * t->stride], q), pi);
        }
        return;
    }
    float diff = q[nd->dim] - nd->split;
    int32_t nearc = diff < 0 ? nd->left : nd->right;
    int32_t farc = diff < 0 ? nd->right : nd->left;
    search_rec(t, nearc, q, r);
    /* explore the far side when the splitting plane is not farther than the current worst
     * (<= keeps lower-index exact ties reachable) */
    if (r->count < r->k || diff * diff <= r->d2[r->k - 1])
        search_rec(t, farc, q, r);
}

int orc_kdtree_knn(const orc_kdtree *t, const float q[3], int k, int32_t *idx, float *d2)
{
    topk r = {k, 0, idx, d2};
    if (t->m > 0)
        search_rec(t, 0, q, &r);
    return r.count;
}

int orc_bruteforce_knn(const float *xyz, int stride, int64_t m, const float q[3], int k, int32_t *idx, float *d2)
{
    topk r = {k, 0, idx, d2};
    for (int64_t i = 0; i < m; i++)
        topk_push(&r, dist2(&xyz[(size_t)i * stride], q), (int32_t)i);
    return r.count;
}
