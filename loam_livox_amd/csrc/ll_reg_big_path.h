// ll_reg_big_path.h -- the solver for scans that the 64-bit activity masks / register tiles of solve_fast3 do not hold: motion-deblur
// scans (ceres_icp_point2plane_mb / point2line_mb, ceres_icp.hpp:81-233) of any size and scans of up to LL_TABLE_MAX_BLOCKS residual
// blocks (Mid-100: three heads, ~50 k blocks), replacing ceres::Solve / Problem::Evaluate of point_cloud_registration.hpp:460-531.
// Included by ll_reg_kernels.hip (one translation unit: the helpers above it are shared).  Round 6; rounds 1 - 5 sent these scans through
// solve_general (below, now the fall-back for scans beyond LL_TABLE_MAX_BLOCKS and the force_general test switch), which streamed
// every block's 49 / 65 bytes on each of ~9 cost evaluations and kept flags and L1 values in HBM: profiles/r05_c3_* -- 10 x the
// algorithmic bytes, 2.45 ms per B = 256 launch, of which (round-6 timing build, gpurun_out/r06a_bench_c3_timing.json) only 37 % were
// cost evaluations: 36 % went into the std::set de-duplication (three hash partitions, each two sweeps over flags + L1 values in HBM),
// 14 % into the rank select, 12 % into a separate L1 sweep.
//
// solve_big keeps what made solve_fast3 fast and drops what ties it to 24 576 blocks:
//   * plane table: a Mid-100 scan's ~50 k plane blocks share ~4 k distinct neighbour triples (tools measurement: 13 blocks per
//     triple against the 20 M-point map), so {n', c} per distinct triple still fits LDS; census + hash inserts + table build as in
//     census_and_plane_table, with a 128-bit activity mask per thread (two registers pairs: 120 rounds of 512 blocks);
//   * a cost evaluation streams 18 B per plane block -- the fp32 feature point with its time stamp straight from the extractor's
//     cloud (the blur ratio s = refine_blur(stamp) is recomputed per block: two float operations) + the 16-bit plane id -- five
//     records deep, plane from the LDS table; the motion-deblur residual and its closed-form Jacobian (ll_reg_core.h
//     block_accumulate_mb) per block;
//   * the prerun's last evaluation leaves the loss-corrected L1 values (one 8-byte store per block), no separate L1 sweep;
//   * the inlier threshold (std::set semantics + rank, PCR:153-161) works on activity / contested / first-occurrence BIT MASKS in
//     registers and reads the L1 values five times in all (mark + range, second-level mark, decide + histogram, candidates,
//     prune) -- no flag bytes in HBM, no hash partitions: the two-level 2-bit slot tables of inlier_threshold_regs are wide enough
//     for 61 k keys (expected twice-contested keys at 50 k: ~700 of the 2 048 the exact list holds).
// Line blocks (a few hundred per scan) are read from HBM in every evaluation (65 B each).  Sums are grouped as in solve_fast3
// (thread-private accumulators over the thread's rounds, one butterfly per wavefront, the wavefront partials in fixed order):
// results agree with the oracle to rounding (pose < 1e-7 with equal ICP / LM / block counts: tests/test_gpu_c3_c5.py, test_gpu_reg.py).

// a triple that finds no slot within this many probes gets a private table entry (a scan with more distinct triples than the 8 192 slots:
// at 192 probes per attempt the census of such a scan took 1.7 M cycles of a 5.8 M-cycle launch)
#define PT_BIG_MAX_PROBE 32

struct Act2 {  // bit k <-> block tid + k * RS_THREADS in the order planes, padding to a whole round, lines
    unsigned long long lo, hi;
};
__device__ __forceinline__ bool act_test(const Act2 &a, int k) { return (((k & 64) ? a.hi : a.lo) >> (k & 63)) & 1ull; }
__device__ __forceinline__ void act_set(Act2 &a, int k)
{
    const unsigned long long bit = 1ull << (k & 63);
    a.lo |= (k & 64) ? 0ull : bit;
    a.hi |= (k & 64) ? bit : 0ull;
}
__device__ __forceinline__ void act_clear(Act2 &a, int k)
{
    const unsigned long long bit = 1ull << (k & 63);
    a.lo &= (k & 64) ? ~0ull : ~bit;
    a.hi &= (k & 64) ? ~bit : ~0ull;
}
__device__ __forceinline__ int act_count(const Act2 &a) { return __popcll(a.lo) + __popcll(a.hi); }

// census of all blocks + the scan's plane table (census_and_plane_table without groups, any number of rounds up to
// LL_TABLE_MAX_BLOCKS / RS_THREADS).  Returns the thread's activity mask.
// Plane ids are RANKED BY USE: while the triples are inserted every hash slot counts the blocks that land on it, and when a scan has
// more distinct triples than LDS holds (PT_TCAP = 4 864; a Mid-100 scan against the 20 M-point map: ~5.3 k) the most used ones stay
// -- the entries left in HBM are the ones hardly any block refers to; the evaluation gathers those through L2 (LB_PLANE).
__device__ __noinline__ Act2 big_census_and_plane_table(const RegDev &rd, const RegConst &rc, const f4 *map_pts, int b, const RegState *st, int nC, int nS,
                                                        uint4 *s_raw, SolveShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kp = (nS + RS_THREADS - 1) / RS_THREADS;
    const int nSp = kp * RS_THREADS;
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
    // ---- census (PCR:325,425) of all blocks + pass 1 of the plane blocks: triples -> hash slots; eight rounds' loads in flight ----
    Act2 act = {0ull, 0ull};
    int na = 0, nca = 0, nsa = 0;
    for (int k0 = 0; k0 * RS_THREADS < totp; k0 += 8) {
        unsigned char fl8[8];
        int4 t8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = tid + (k0 + u) * RS_THREADS;
            const int jc = j < totp ? j : 0;
            const size_t src = jc < nS ? (size_t)rd.cap_c + jc : (jc >= nSp ? (size_t)(jc - nSp) : (size_t)rd.cap_c);
            fl8[u] = gload_u8(flag0 + src);
            t8[u] = gload_i4(nn + (j < nS ? j : 0));
        }
        unsigned int h8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = tid + (k0 + u) * RS_THREADS;
            const unsigned char fl = (j < totp && (j < nS || j >= nSp)) ? fl8[u] : (unsigned char)0;
            const bool active = (fl & BLK_ACTIVE) != 0;
            if (active) {
                act_set(act, k0 + u);
                na++;
            }
            if (fl & 8) {
                if (j >= nSp) nca++; else nsa++;
            }
            const int4 t = t8[u];
            const unsigned int h = (active && j < nS) ? pt_insert<PT_BIG_MAX_PROBE>(ht, (unsigned int)t.x, (unsigned int)t.y, (unsigned int)t.z) : PT_INACTIVE;
            if (h == PT_PRIVATE) atomicAdd(&sh.pt_priv, 1);
            if (h < PT_SLOTS) __hip_atomic_fetch_add(&ht[h].id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // use count - 1 (the slots start at 0xffffffff)
            h8[u] = h;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {  // (the slot indices leave together at the end of the trip, as in census_and_plane_table)
            const int j = tid + (k0 + u) * RS_THREADS;
            if (j < nS) gstore_u16(ids + j, (unsigned short)h8[u]);
        }
    }
    {
        const unsigned long long tot = block_sum_u64((unsigned long long)na | ((unsigned long long)nca << 20) | ((unsigned long long)nsa << 40), sh);
        na = (int)(tot & 0xfffffull);
        nca = (int)((tot >> 20) & 0xfffffull);
        nsa = (int)((tot >> 40) & 0xfffffull);
    }
    if (rc.subsample_seed && na > rc.max_blocks) {  // a13 (PCR:438-458); the random stream is indexed by the block's position in the
        int kept = 0;                               // reference's order: corners, then surfaces
        for (int k = 0; k * RS_THREADS < totp; k++) {
            if (!act_test(act, k)) continue;
            const int j = tid + k * RS_THREADS;
            const int jref = j >= nSp ? j - nSp : nC + j;
            if (subsample_drop_block(rc.subsample_seed, st->icp_iters, jref, na, rc.max_blocks))
                act_clear(act, k);
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
    LL_TACC(6, t_census);
    LL_T0(t_tab);
    // ---- which triples stay in LDS: a deterministic cut through (use class, key bucket); dense ids, the LDS part first; id -> slot map ----
    // 7 classes of use count (>= 32 blocks, 16 - 31, 8 - 15, 4 - 7, 3, 2, 1) x 64 buckets of the key's hash: triples are taken in that
    // order while they fit PT_TCAP.  (Counts and keys do not depend on the order the lanes' inserts landed in, so neither does the cut;
    // the answer would not depend on it anyway -- every block is summed in its own place whichever memory its plane comes from.)
    constexpr int SPT = PT_SLOTS / RS_THREADS;
    constexpr int NCB = 7 * 64;
    LL_AS_LDS int *cb_hist = (LL_AS_LDS int *)((LL_AS_LDS char *)s_raw + PT_MAP_OFF + PT_SLOTS * 2);
    static_assert(PT_MAP_OFF + PT_SLOTS * 2 + NCB * 4 <= PT_LDS_BYTES, "class / bucket histogram behind the id -> slot map");
    if (tid < NCB) cb_hist[tid] = 0;
    __syncthreads();
    auto class_bucket = [&](int slot) -> int {
        const unsigned int cnt = ht[slot].id + 1u;
        const unsigned long long ka = ht[slot].a;
        const unsigned int c = cnt >= 32u ? 0u : (cnt >= 16u ? 1u : (cnt >= 8u ? 2u : (cnt >= 4u ? 3u : (cnt == 3u ? 4u : (cnt == 2u ? 5u : 6u)))));
        return (int)(c * 64u + (pt_hash((unsigned int)(ka >> 32), (unsigned int)ka, ht[slot].b) >> 26));
    };
    unsigned int occ = 0;
#pragma unroll
    for (int i = 0; i < SPT; i++)
        if (ht[tid + i * RS_THREADS].b != PT_EMPTY_B) {
            occ |= 1u << i;
            __hip_atomic_fetch_add(&cb_hist[class_bucket(tid + i * RS_THREADS)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    int cut, Tn, T;  // (class, bucket) cells [0, cut) stay in LDS: Tn triples of T
    {
        const int h = tid < NCB ? cb_hist[tid] : 0;
        int incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) sh.isum[wave] = incl;
        __syncthreads();
        for (int w = 0; w < wave; w++) incl += sh.isum[w];
        __syncthreads();
        const bool fits = tid < NCB && incl <= PT_TCAP;  // (the prefix sums grow: `fits` holds for a leading run of cells)
        const unsigned long long r = block_sum_u64((unsigned long long)(fits ? 1 : 0) | ((unsigned long long)(fits ? h : 0) << 20) | ((unsigned long long)h << 40), sh);
        cut = (int)(r & 0xfffffull);
        Tn = (int)((r >> 20) & 0xfffffull);
        T = (int)((r >> 40) & 0xfffffull);
    }
    unsigned int nearm = 0;  // which of the thread's occupied slots stay in LDS
    int cls_near = 0, cls_far = 0;
#pragma unroll
    for (int i = 0; i < SPT; i++)
        if (occ & (1u << i)) {
            if (class_bucket(tid + i * RS_THREADS) < cut) {
                nearm |= 1u << i;
                cls_near++;
            } else {
                cls_far++;
            }
        }
    __syncthreads();  // (every use count has been read: the id fields may be overwritten)
    {
        const int mine = cls_near | (cls_far << 16);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        if (lane == 63) sh.isum[wave] = incl;
        __syncthreads();
        int before = incl - mine;
        for (int w = 0; w < wave; w++) before += sh.isum[w];
        int rk_near = before & 0xffff, rk_far = Tn + (before >> 16);
#pragma unroll
        for (int i = 0; i < SPT; i++)
            if (occ & (1u << i)) {
                const int id = (nearm & (1u << i)) ? rk_near++ : rk_far++;
                ht[tid + i * RS_THREADS].id = (unsigned int)id;
                slot_of_id[id] = (unsigned short)(tid + i * RS_THREADS);
            }
    }
    __syncthreads();
    // ---- plane constants -> the table in HBM (four triples' gathers in flight) ----
    int4 *tabG = pt_table_global(rd, b, 0, false);
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
            (void)block_plane(pose_last, pa, pb, pc, a_out, v_out);  // degenerate triples never reach the table (the build clears their flag)
            if (id < T) {
                gstore_i4(tabG + 2 * id, make_int4(__double2loint(v_out[0]), __double2hiint(v_out[0]), __double2loint(v_out[1]), __double2hiint(v_out[1])));
                gstore_i4(tabG + 2 * id + 1, make_int4(__double2loint(v_out[2]), __double2hiint(v_out[2]), __double2loint(a_out[0]), __double2hiint(a_out[0])));
            }
        }
    }
    // ---- pass 2: slot -> dense id; which of the thread's blocks have their plane beyond the LDS part ----
    const int region = rd.tab_cap;
    const int Tl = Tn;  // (== T when every triple fits)
    for (int k0 = 0; k0 < kp; k0 += 8) {
        unsigned short h8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int p = tid + (k0 + u) * RS_THREADS;
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
            const int p = tid + (k0 + u) * RS_THREADS;
            if (p < nS) gstore_u16(ids + p, (unsigned short)id8[u]);
        }
    }
    if (sh.pt_priv > 0) {  // (uniform; the hash table was too crowded around some triples -- more distinct triples than a Mid-100 scan against the
        int npriv = 0;     //  synthetic rooms has: those blocks get entries of their own at the top of the table region)
        for (int k = 0; k < kp; k++) {
            const int p = tid + k * RS_THREADS;
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
            const int p = tid + k * RS_THREADS;
            if (p >= nS || gload_u16(ids + p) != PT_PRIVATE) continue;
            const unsigned int id = (unsigned int)(region - 1 - pid);
            pid++;
            const int4 t = gload_i4(nn + p);
            int4 ob, oc;
            pt_plane(map_pts, pose_last, (unsigned int)t.x, (unsigned int)t.y, (unsigned int)t.z, ob, oc);
            gstore_i4(tabG + 2 * id, ob);
            gstore_i4(tabG + 2 * id + 1, oc);
            gstore_u16(ids + p, (unsigned short)id);
        }
    }
    __threadfence_block();
    __syncthreads();  // the hash table is dead, the table in HBM complete
#ifdef LL_SOLVE_TIMING
    if (tid == 0) sh.tcyc[15] += T, sh.tcyc[14] += sh.pt_priv;  // distinct triples in the table, blocks that found no slot
#endif
    if (tid == 0) {
        sh.pt_T = sh.pt_priv > 0 ? PT_TCAP + 1 : T;  // (beyond the LDS part, or private entries: planes with id >= Tl are gathered from HBM)
        sh.pt_Tl = Tl;
        sh.pt_kc = 0;
        sh.pt_nl = 0;
    }
    __syncthreads();
    plane_table_reload(rd, b, false, s_raw, sh);  // the first PT_TCAP entries -> LDS
    LL_TACC(8, t_tab);
    return act;
}

struct RecB {
    int fx, fy, fz, fw;  // bits of the fp32 feature point (sensor frame) and of its time stamp
    unsigned int id;
};

// The plane loop of a cost evaluation, rounds [0, kp) of this thread's blocks: solver_eval3's pipeline (records four rounds ahead in
// five register sets rotating by name, the plane of the next record fetched while the current one is evaluated, every load
// unconditional from a clamped address) without the LDS record cache -- the table of a Mid-100 scan leaves no room for one.
#define LB_LOAD(R, K)                                                                              \
    {                                                                                              \
        const int pp_ = tid + (K) * RS_THREADS;                                                    \
        const int pc_ = pp_ < nS ? pp_ : 0;                                                        \
        R.id = gload_u16(ids + pc_);                                                               \
        if (DEBLUR) {                                                                              \
            const float4 f_ = gload_f4(feat + pc_);                                                \
            R.fx = __float_as_int(f_.x);                                                           \
            R.fy = __float_as_int(f_.y);                                                           \
            R.fz = __float_as_int(f_.z);                                                           \
            R.fw = __float_as_int(f_.w);                                                           \
        } else {                                                                                   \
            float fx_, fy_, fz_;                                                                   \
            gload_f3(feat + pc_, fx_, fy_, fz_);                                                   \
            R.fx = __float_as_int(fx_);                                                            \
            R.fy = __float_as_int(fy_);                                                            \
            R.fz = __float_as_int(fz_);                                                            \
            R.fw = 0;                                                                              \
        }                                                                                          \
    }
/* The plane of a record: from the LDS part of the table (ids below Tl: the most used triples, big_census_and_plane_table) or, for a scan  \
 * with more distinct triples than LDS holds, from the table in HBM.  Both loads are UNCONDITIONAL from clamped addresses -- a block   \
 * whose plane is in LDS gathers entry 0 from HBM (one cached line for the whole wavefront), the others read LDS entry 0 -- and the    \
 * block picks its source when it is evaluated, one round later: no branch inside the pipeline (a load behind a branch makes the    \
 * compiler drain it), and every block is summed in its own place, so a scan's answer does not depend on which triples got LDS.     */ \
#define LB_PLANE(Q, G_, R)                                                                         \
    {                                                                                              \
        const bool near_ = TAB_LDS || R.id < (unsigned int)Tl;                                     \
        const unsigned int idl_ = near_ ? R.id : 0u, idg_ = near_ ? 0u : R.id;                     \
        Q.b = lds_load_i4(tabL + 2 * idl_);                                                        \
        Q.c = lds_load_i4(tabL + 2 * idl_ + 1);                                                    \
        if (!TAB_LDS) {                                                                            \
            G_.b = gload_i4(tabG + 2 * idg_);                                                      \
            G_.c = gload_i4(tabG + 2 * idg_ + 1);                                                  \
        }                                                                                          \
    }
#define LB_USE(R, Q, G_, K)                                                                                \
    if ((K) < kp) {                                                                                        \
        const int pp_ = tid + (K) * RS_THREADS;                                                            \
        if (pp_ < nS && act_test(act, (K))) {                                                              \
            const bool near_ = TAB_LDS || R.id < (unsigned int)Tl;                                         \
            const int4 pb_ = near_ ? Q.b : G_.b, pc_ = near_ ? Q.c : G_.c;                                 \
            const double f[3] = {(double)__int_as_float(R.fx), (double)__int_as_float(R.fy), (double)__int_as_float(R.fz)}; \
            const double v[3] = {__hiloint2double(pb_.y, pb_.x), __hiloint2double(pb_.w, pb_.z), __hiloint2double(pc_.y, pc_.x)}; \
            const double a[3] = {__hiloint2double(pc_.w, pc_.z), 0.0, 0.0};                                \
            if (DEBLUR) {                                                                                  \
                const double s_ = (double)refine_blur(1, __int_as_float(R.fw), min_ts, max_ts); /* PCR:128-141, as build_one stores it for the general path */ \
                block_accumulate_mb(BLK_PLANE, mb_, t_, s_, f, a, v, huber_a, acc);                        \
                if (L1OUT) gstore_f64(l1_planes + pp_, block_l1_mb(BLK_PLANE, mb_, t_, s_, f, a, v, huber_a, q_last)); \
            } else {                                                                                       \
                block_accumulate(BLK_PLANE, R_, t_, f, a, v, huber_a, acc);                                \
                if (L1OUT) gstore_f64(l1_planes + pp_, block_l1(BLK_PLANE, R_, t_, f, a, v, huber_a, q_last)); \
            }                                                                                              \
        }                                                                                                  \
    }
#define LB_PIPE()                                                      \
    {                                                                  \
        if (kp > 0) {                                                  \
            RecB r0, r1, r2, r3, r4;                                   \
            Pl3 q0, q1, g0, g1;                                        \
            LB_LOAD(r0, 0)                                             \
            LB_LOAD(r1, 1)                                             \
            LB_LOAD(r2, 2)                                             \
            LB_LOAD(r3, 3)                                             \
            LB_PLANE(q0, g0, r0)                                       \
            for (int k = 0; k < kp; k += 10) {                         \
                LB_LOAD(r4, k + 4)                                     \
                LB_PLANE(q1, g1, r1)                                   \
                LB_USE(r0, q0, g0, k)                                  \
                LB_LOAD(r0, k + 5)                                     \
                LB_PLANE(q0, g0, r2)                                   \
                LB_USE(r1, q1, g1, k + 1)                              \
                LB_LOAD(r1, k + 6)                                     \
                LB_PLANE(q1, g1, r3)                                   \
                LB_USE(r2, q0, g0, k + 2)                              \
                LB_LOAD(r2, k + 7)                                     \
                LB_PLANE(q0, g0, r4)                                   \
                LB_USE(r3, q1, g1, k + 3)                              \
                LB_LOAD(r3, k + 8)                                     \
                LB_PLANE(q1, g1, r0)                                   \
                LB_USE(r4, q0, g0, k + 4)                              \
                LB_LOAD(r4, k + 9)                                     \
                LB_PLANE(q0, g0, r1)                                   \
                LB_USE(r0, q1, g1, k + 5)                              \
                LB_LOAD(r0, k + 10)                                    \
                LB_PLANE(q1, g1, r2)                                   \
                LB_USE(r1, q0, g0, k + 6)                              \
                LB_LOAD(r1, k + 11)                                    \
                LB_PLANE(q0, g0, r3)                                   \
                LB_USE(r2, q1, g1, k + 7)                              \
                LB_LOAD(r2, k + 12)                                    \
                LB_PLANE(q1, g1, r4)                                   \
                LB_USE(r3, q0, g0, k + 8)                              \
                LB_LOAD(r3, k + 13)                                    \
                LB_PLANE(q0, g0, r0)                                   \
                LB_USE(r4, q1, g1, k + 9)                              \
            }                                                          \
        }                                                              \
    }

// workgroup evaluation of cost / g / H at x over the active blocks -> sh.sum; L1OUT: the evaluation also leaves every active block's
// loss-corrected L1 value in rd.blk_l1 (the prerun's last candidate: if it is accepted, the inlier phase starts from them)
template <bool L1OUT, int DEBLUR>
__device__ __noinline__ void big_eval(const RegDev &rd, const RegConst &rc, int b, int nC, int nS, const double *x, Act2 act, uint4 *s_raw,
                                      const double *q_last_g, SolveShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double huber_a = rc.huber_a;
    const float min_ts = rc.min_ts, max_ts = rc.max_ts;
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
    const int Tl = sh.pt_Tl;
    const int kp = (nS + RS_THREADS - 1) / RS_THREADS;
    const int4 *tabL = (const int4 *)s_raw;
    const int4 *tabG = pt_table_global(rd, b, 0, false);
    const unsigned short *ids = rd.blk_id + (size_t)b * rd.cap_s;
    const float4 *feat = rd.surf_feat + (size_t)b * rd.feat_stride_s;
    if (sh.pt_T <= PT_TCAP) {  // uniform: the whole table is in LDS
        constexpr bool TAB_LDS = true;
        LB_PIPE()
    } else {
        constexpr bool TAB_LDS = false;
        LB_PIPE()
    }
    {
        // line blocks (a few hundred per scan): the 65-byte fp64 form, from HBM
        const size_t sb = (size_t)b * rd.cap;
        const double *av = rd.blk_av + (size_t)b * 6 * rd.cap;
        int k = kp;
        for (int l = tid; l < nC; l += RS_THREADS, k++) {
            if (!act_test(act, k)) continue;
            BlkRegs br;
            load_blk(rd, sb, av, l, br);
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
    wave_sum_acc(acc, sh.red[wave], lane);
    __syncthreads();
    if (tid < LL_NACC) {
        double s = 0.0;
        for (int w = 0; w < RS_WAVES; w++) s += sh.red[w][tid];
        sh.sum[tid] = s;
    }
    __syncthreads();
}
#undef LB_LOAD
#undef LB_PLANE
#undef LB_USE
#undef LB_PIPE

// one ceres::Solve: starts at x0, leaves the result in sh.ctl (solver_lm3 without groups)
template <bool WANT_L1, int DEBLUR>
__device__ __forceinline__ void big_lm(const RegDev &rd, const RegConst &rc, int b, int nC, int nS, const double *x0, int max_iter, int n_active, Act2 act,
                                       uint4 *s_raw, const double *q_last, SolveShared &sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        lm_begin(sh.ctl, x0, max_iter, rc.bound);
        sh.l1_valid = 0;
    }
    __syncthreads();
    {
        LL_T0(t0);
        big_eval<false, DEBLUR>(rd, rc, b, nC, nS, sh.ctl.x, act, s_raw, q_last, sh);
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
            big_eval<true, DEBLUR>(rd, rc, b, nC, nS, sh.ctl.cand, act, s_raw, q_last, sh);
        else
            big_eval<false, DEBLUR>(rd, rc, b, nC, nS, sh.ctl.cand, act, s_raw, q_last, sh);
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

// L1 values at the prerun result -> inlier threshold -> prune (PCR:476-499) for a scan of any size up to LL_TABLE_MAX_BLOCKS: the
// std::set semantics of PCR:153-161 (which L1 values are distinct, the element at int(ratio * n_distinct)) on BIT MASKS in registers
// -- active, valid (not NaN), contested, first occurrence: bit k <-> the thread's k-th block -- with the L1 values read from
// rd.blk_l1 where the prerun's last evaluation left them, five sweeps in all, eight rounds' loads in flight per trip:
//   A  every key marks a 2-bit state {a key landed here, a second key landed here} in a 256 K-slot table; value range of the keys;
//   B  keys of slots that received a second key repeat that in a 128 K-slot table under an independent hash;
//   C  keys contested twice (true duplicates + a stray pair: ~700 of 50 k) go to a list {key, block index} and are the first occurrence of
//      their value iff no entry with the same key has a smaller block index; every other valid key is distinct; the first occurrences
//      fill the value-range histogram of the rank select;
//   D  the keys of the histogram bin that holds the wanted rank are ranked exactly (crowded bin: 8-bit radix select);
//   E  prune.
// (inlier_threshold_regs does A - D on a register tile of at most 48 values per thread; solve_general did them on flag bytes and L1
// values in HBM, in three hash partitions for a 50 k-block scan.)  Returns the pruned mask.
template <int DEBLUR>
__device__ __noinline__ Act2 big_inlier(const RegDev &rd, const RegConst &rc, int b, RegState *st, SolveShared &sh, uint4 *s_raw, Act2 act, int nC, int nS)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kp = (nS + RS_THREADS - 1) / RS_THREADS;
    const int nSp = kp * RS_THREADS;
    const int totp = nSp + nC;
    const int kt = (totp + RS_THREADS - 1) / RS_THREADS;
    const size_t sb = (size_t)b * rd.cap;
    double *l1g = rd.blk_l1 + sb;
    LL_T0(t_l1);
    if (!sh.l1_valid) {
        // rare: the prerun ended on a rejected step (or converged early): every thread evaluates its own blocks, planes from the table in HBM
        LL_CTX_DECL(sh.ctl.x)
        const int4 *tabG = pt_table_global(rd, b, 0, false);
        const unsigned short *ids = rd.blk_id + (size_t)b * rd.cap_s;
        const float4 *feat = rd.surf_feat + (size_t)b * rd.feat_stride_s;
        const double *av = rd.blk_av + (size_t)b * 6 * rd.cap;
        int k = 0;
        for (int p = tid; p < nS; p += RS_THREADS, k++) {
            if (!act_test(act, k)) continue;
            const float4 ff = gload_f4(feat + p);
            const unsigned int id = gload_u16(ids + p);
            const int4 qb = gload_i4(tabG + 2 * id), qc = gload_i4(tabG + 2 * id + 1);
            const double f[3] = {(double)ff.x, (double)ff.y, (double)ff.z};
            const double v[3] = {__hiloint2double(qb.y, qb.x), __hiloint2double(qb.w, qb.z), __hiloint2double(qc.y, qc.x)};
            const double a[3] = {__hiloint2double(qc.w, qc.z), 0.0, 0.0};
            if (DEBLUR)
                l1g[rd.cap_c + p] = block_l1_mb(BLK_PLANE, mb_, t_, (double)refine_blur(1, ff.w, rc.min_ts, rc.max_ts), f, a, v, rc.huber_a, st->pose_last);
            else
                l1g[rd.cap_c + p] = block_l1(BLK_PLANE, R_, t_, f, a, v, rc.huber_a, st->pose_last);
        }
        k = kp;
        for (int l = tid; l < nC; l += RS_THREADS, k++) {
            if (!act_test(act, k)) continue;
            BlkRegs br;
            load_blk(rd, sb, av, l, br);
            const double a[3] = {br.a0, br.a1, br.a2};
            const double v[3] = {br.v0, br.v1, br.v2};
            double l1;
            LL_CTX_L1(l1, BLK_LINE, br.f, a, v, rc.huber_a, st->pose_last);
            l1g[l] = l1;
        }
    } else if (tid == 0) {
        sh.tcyc[9] += 1;  // LL_SOLVE_TIMING: how often the shortcut was taken
    }
    __syncthreads();  // (every thread reads back only what it wrote itself: thread t owns blocks t, t + 512, ... in both phases)
    LL_TACC(2, t_l1);
    LL_T0(t_dd);

    // the L1 values of rounds k0 .. k0 + 7 of this thread (-1: no active block there; NaN stays NaN)
#define LB_KEYS8(V8, K0)                                                                                   \
    {                                                                                                      \
        _Pragma("unroll") for (int u = 0; u < 8; u++)                                                      \
        {                                                                                                  \
            const int j = tid + ((K0) + u) * RS_THREADS;                                                   \
            const int jc = j < totp ? j : 0;                                                               \
            const size_t src = jc < nS ? (size_t)rd.cap_c + jc : (jc >= nSp ? (size_t)(jc - nSp) : (size_t)rd.cap_c); \
            V8[u] = gload_f64(l1g + src);                                                                  \
        }                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 8; u++) V8[u] = ((K0) + u < kt && act_test(act, (K0) + u)) ? V8[u] : -1.0; \
    }
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
    unsigned int *bmA = (unsigned int *)s_raw;                            // [DD2_WORDS]  16 slots x 2 bits per word, 64 KB
    unsigned int *bmB = bmA + DD2_WORDS;                                  // [DD2B_WORDS] second table, 32 KB
    unsigned long long *dlk = (unsigned long long *)(bmB + DD2B_WORDS);   // [DD2_LIST] twice-contested keys ...
    int *dlj = (int *)(dlk + DD2_LIST);                                   // [DD2_LIST] ... and their block indices
    int *bins = dlj + DD2_LIST;                                           // [SEL_BINS] value-range histogram of the distinct keys
    static_assert((DD2_WORDS + DD2B_WORDS) * 4 + DD2_LIST * 12 + SEL_BINS * 4 <= PT_LDS_BYTES, "tables of the inlier phase fit s_raw");
    {
        uint4 *z = (uint4 *)s_raw;
        for (int e = tid; e < (DD2_WORDS + DD2B_WORDS) / 4; e += RS_THREADS) z[e] = make_uint4(0u, 0u, 0u, 0u);
        for (int e = tid; e < SEL_BINS; e += RS_THREADS) bins[e] = 0;
        if (tid == 0) sh.n_cand = 0;
    }
    __syncthreads();
    // ---- A: 2-bit slot states; which blocks hold a key at all; the keys' range ----
    LL_T0(t_a);
    Act2 valid = {0ull, 0ull};
    double kmin = INFINITY, kmax = -INFINITY;
    for (int k0 = 0; k0 < kt; k0 += 8) {
        double v8[8];
        LB_KEYS8(v8, k0)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double l1 = v8[u];
            const bool ok = l1 >= 0.0;  // (inactive slot or NaN: NaN never enters the set)
            const unsigned int slot = slot_a((unsigned long long)__double_as_longlong(l1));
            const unsigned int bit0 = ok ? (1u << ((slot & 15u) * 2u)) : 0u;
            const unsigned int old = atomicOr(&bmA[slot >> 4], bit0);
            atomicOr(&bmA[slot >> 4], (old & bit0) << 1);
            if (ok) {
                act_set(valid, k0 + u);
                kmin = fmin(kmin, l1);
                kmax = fmax(kmax, l1);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        kmin = fmin(kmin, __shfl_down(kmin, off));
        kmax = fmax(kmax, __shfl_down(kmax, off));
    }
    if (lane == 0) {
        sh.red[wave][0] = kmin;
        sh.red[wave][1] = kmax;
    }
    __syncthreads();
    double lo = sh.red[0][0], hi = sh.red[0][1];
    for (int w = 1; w < RS_WAVES; w++) {
        lo = fmin(lo, sh.red[w][0]);
        hi = fmax(hi, sh.red[w][1]);
    }
    const double scale = (hi > lo) ? (double)(SEL_BINS - 1) / (hi - lo) : 0.0;
    LL_TACC(10, t_a);
    LL_T0(t_b);
    // ---- B: keys of contested slots mark the second table ----
    Act2 cont = {0ull, 0ull};
    for (int k0 = 0; k0 < kt; k0 += 8) {
        double v8[8];
        LB_KEYS8(v8, k0)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(v8[u]);
            const unsigned int slot = slot_a(key);
            const bool contested = act_test(valid, k0 + u) && ((bmA[slot >> 4] >> ((slot & 15u) * 2u)) & 2u);
            const unsigned int h2 = slot_b(key);
            const unsigned int bit0 = contested ? (1u << ((h2 & 15u) * 2u)) : 0u;
            const unsigned int old = atomicOr(&bmB[h2 >> 4], bit0);
            atomicOr(&bmB[h2 >> 4], (old & bit0) << 1);
            if (contested) act_set(cont, k0 + u);
        }
    }
    __syncthreads();
    LL_TACC(11, t_b);
    LL_T0(t_c);
    // ---- C: twice-contested keys -> the exact list; everything else that is valid is distinct and goes into the histogram ----
    Act2 first = {0ull, 0ull};
    for (int k0 = 0; k0 < kt; k0 += 8) {
        double v8[8];
        LB_KEYS8(v8, k0)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k0 + u;
            if (!act_test(valid, k)) continue;
            const unsigned long long key = (unsigned long long)__double_as_longlong(v8[u]);
            const unsigned int h2 = slot_b(key);
            if (act_test(cont, k) && ((bmB[h2 >> 4] >> ((h2 & 15u) * 2u)) & 2u)) {
                const int pos = atomicAdd(&sh.n_cand, 1);
                if (pos < DD2_LIST) {
                    dlk[pos] = key;
                    dlj[pos] = tid + k * RS_THREADS;
                }
            } else {
                act_set(first, k);
                int bi = (int)((v8[u] - lo) * scale);
                bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                atomicAdd(&bins[bi], 1);
            }
        }
    }
    __syncthreads();
    LL_TACC(12, t_c);
    LL_T0(t_x);
    const int n_list = sh.n_cand;
#ifdef LL_SOLVE_TIMING
    if (tid == 0) sh.tcyc[13] += 0;  // (n_list: see tcyc[14] of the table build for the private blocks)
#endif
    // The list's entries are decided by the threads of the workgroup side by side (entry e by thread e mod 512: at most four each), not by
    // the blocks' owners -- an owner-side loop ran once per (wavefront, round) that held any listed key, ~600 dependent list scans per
    // wavefront.  An entry is the first occurrence of its value iff no entry with the same key has a smaller block index; the owner
    // never needs to know: the entry's thread counts it, puts it into the histogram and, later, offers it as a candidate of the rank select.
    unsigned int list_first = 0;  // bit q: list entry tid + q * 512 is a first occurrence
    if (n_list <= DD2_LIST) {
        static_assert(DD2_LIST <= 4 * RS_THREADS, "at most four list entries per thread");
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + q * RS_THREADS;
            if (e >= n_list) continue;
            const unsigned long long key = dlk[e];
            const int j = dlj[e];
            bool dup = false;
            int i = 0;
            for (; i + 4 <= n_list; i += 4) {  // (four independent pairs of LDS reads per trip: the scan is bound by their latency)
                const unsigned long long k0_ = dlk[i], k1_ = dlk[i + 1], k2_ = dlk[i + 2], k3_ = dlk[i + 3];
                const int j0_ = dlj[i], j1_ = dlj[i + 1], j2_ = dlj[i + 2], j3_ = dlj[i + 3];
                dup |= (k0_ == key && j0_ < j) | (k1_ == key && j1_ < j) | (k2_ == key && j2_ < j) | (k3_ == key && j3_ < j);
            }
            for (; i < n_list; i++) dup |= (dlk[i] == key && dlj[i] < j);
            if (!dup) {
                list_first |= 1u << q;
                int bi = (int)((__longlong_as_double((long long)key) - lo) * scale);
                bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                atomicAdd(&bins[bi], 1);
            }
        }
    } else {
        // heavily duplicated input: one compare-and-swap table in HBM over every key (solve_general's fall-back)
        __syncthreads();
        first.lo = first.hi = 0ull;
        for (int e = tid; e < SEL_BINS; e += RS_THREADS) bins[e] = 0;
        unsigned long long *table = rd.hash + (size_t)b * rd.hash_cap;
        for (int e = tid; e < rd.hash_cap; e += RS_THREADS) table[e] = HASH_EMPTY;
        __threadfence();
        __syncthreads();
        const unsigned long long mask = (unsigned long long)rd.hash_cap - 1ull;
        for (int k = 0; k < kt; k++) {
            if (!act_test(valid, k)) continue;
            const int j = tid + k * RS_THREADS;
            const size_t src = j < nS ? (size_t)rd.cap_c + j : (size_t)(j - nSp);
            const double l1 = gload_f64(l1g + src);
            const unsigned long long key = (unsigned long long)__double_as_longlong(l1);
            unsigned long long h = hash64(key) & mask;
            for (;;) {
                const unsigned long long old = atomicCAS(&table[h], HASH_EMPTY, key);
                if (old == HASH_EMPTY) {
                    act_set(first, k);
                    int bi = (int)((l1 - lo) * scale);
                    bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                    atomicAdd(&bins[bi], 1);
                    break;
                }
                if (old == key) break;
                h = (h + 1ull) & mask;
            }
        }
    }
    LL_TACC(13, t_x);
    {
        const int nu = block_sum_int(act_count(first) + __popc(list_first), sh);  // (its barriers also complete the histogram)
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
        // ---- D: the bin that holds the wanted rank (a monotone map: every key of a lower bin is smaller), its keys ranked exactly ----
        unsigned long long *cand = (unsigned long long *)s_raw;  // [SEL_CAND] (the slot tables are dead)
        {
            const int per = SEL_BINS / RS_THREADS;
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
        const int sel_bin = sh.sel_bin;
        if (sh.sel_cnt <= SEL_CAND) {
            for (int k0 = 0; k0 < kt; k0 += 8) {
                double v8[8];
                LB_KEYS8(v8, k0)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (!act_test(first, k0 + u)) continue;
                    int bi = (int)((v8[u] - lo) * scale);
                    bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                    if (bi == sel_bin) cand[atomicAdd(&sh.n_cand, 1)] = (unsigned long long)__double_as_longlong(v8[u]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {  // the listed first occurrences
                if (!((list_first >> q) & 1u)) continue;
                const unsigned long long key = dlk[tid + q * RS_THREADS];
                int bi = (int)((__longlong_as_double((long long)key) - lo) * scale);
                bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                if (bi == sel_bin) cand[atomicAdd(&sh.n_cand, 1)] = key;
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
            // crowded bin: MSB-first radix select (8 bits per pass) restricted to the keys of that bin; non-negative doubles order like uint64
            if (tid == 0) sh.sel_prefix = 0ull;
            __syncthreads();
            for (int pass = 0; pass < 8; pass++) {
                const int shift = 56 - 8 * pass;
                for (int e = tid; e < 256; e += RS_THREADS) sh.hist[e] = 0;
                __syncthreads();
                const unsigned long long prefix = sh.sel_prefix;
                for (int k0 = 0; k0 < kt; k0 += 8) {
                    double v8[8];
                    LB_KEYS8(v8, k0)
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (!act_test(first, k0 + u)) continue;
                        int bi = (int)((v8[u] - lo) * scale);
                        bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                        if (bi != sel_bin) continue;
                        const unsigned long long key = (unsigned long long)__double_as_longlong(v8[u]);
                        if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&sh.hist[(int)((key >> shift) & 255ull)], 1);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {  // the listed first occurrences
                    if (!((list_first >> q) & 1u)) continue;
                    const unsigned long long key = dlk[tid + q * RS_THREADS];
                    int bi = (int)((__longlong_as_double((long long)key) - lo) * scale);
                    bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                    if (bi == sel_bin && (pass == 0 || (key >> (shift + 8)) == prefix)) atomicAdd(&sh.hist[(int)((key >> shift) & 255ull)], 1);
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
    // ---- E: prune (PCR:487-499) ----
    LL_T0(t_prune);
    {
        const double thr = sh.thr;
        int na = 0;
        for (int k0 = 0; k0 < kt; k0 += 8) {
            double v8[8];
            LB_KEYS8(v8, k0)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + u;
                if (k >= kt || !act_test(act, k)) continue;
                if (v8[u] > thr)
                    act_clear(act, k);
                else
                    na++;
            }
        }
        na = block_sum_int(na, sh);
        if (tid == 0) sh.n_active = na;
        __syncthreads();
    }
    LL_TACC(7, t_prune);
#undef LB_KEYS8
    return act;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// solve_general: any number of blocks per scan; flags, L1 values and the de-duplication table live in HBM, every cost evaluation
// streams every block's 49 / 65 bytes.  Since round 6 only the fall-back for scans beyond LL_TABLE_MAX_BLOCKS and the force_general
// test switch (ll_reg_set_debug bit 1), which keeps it under test against the oracle and the plane-table paths.
// workgroup evaluation of cost / g / H at x (LDS) over the active blocks -> sh.sum
template <int DEBLUR>
__device__ __noinline__ void solver_eval(const RegDev &rd, int b, int nC, int nS, const double *x, double huber_a, int deblur, SolveShared &sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t sb = (size_t)b * rd.cap;
    const double *av = rd.blk_av + (size_t)b * 6 * rd.cap;
    LL_CTX_DECL(x)
    double acc[LL_NACC];
#pragma unroll
    for (int i = 0; i < LL_NACC; i++) acc[i] = 0.0;
    const int total = nC + nS;
    // software-pipelined one block ahead (flag included): with two waves per SIMD nothing else hides the loads
    int j = tid;
    float4 nf = make_float4(0.f, 0.f, 0.f, 0.f);
    double na0 = 0, na1 = 0, na2 = 0, nv0 = 0, nv1 = 0, nv2 = 0;
    unsigned char nfl = 0;
    if (j < total) {
        const int slot = slot_of(j, nC, rd.cap_c);
        nfl = rd.blk_flag[sb + slot];
        nf = rd.blk_f[sb + slot];
        av_load(av, rd.cap, slot, slot < rd.cap_c, na0, na1, na2, nv0, nv1, nv2);
    }
    while (j < total) {
        const unsigned char fl = nfl;
        const float4 ff = nf;
        const double a[3] = {na0, na1, na2}, v[3] = {nv0, nv1, nv2};
        const int jn = j + RS_THREADS;
        if (jn < total) {
            const int slot = slot_of(jn, nC, rd.cap_c);
            nfl = rd.blk_flag[sb + slot];
            nf = rd.blk_f[sb + slot];
            av_load(av, rd.cap, slot, slot < rd.cap_c, na0, na1, na2, nv0, nv1, nv2);
        }
        if (fl & BLK_ACTIVE) LL_CTX_ACCUM(fl & 3, ff, a, v, huber_a, acc);
        j = jn;
    }
    wave_sum_acc(acc, sh.red[wave], lane);
    __syncthreads();
    if (tid < LL_NACC) {
        double s = 0.0;
        for (int w = 0; w < RS_WAVES; w++) s += sh.red[w][tid];
        sh.sum[tid] = s;
    }
    __syncthreads();
}


// one ceres::Solve: starts at x0 (global/LDS), leaves the result in sh.ctl
template <int DEBLUR>
__device__ void solver_lm(const RegDev &rd, const RegConst &rc, int b, int nC, int nS, const double *x0, int max_iter,
                          int n_active, SolveShared &sh)
{
    const int tid = threadIdx.x;
    if (tid == 0) lm_begin(sh.ctl, x0, max_iter, rc.bound);
    __syncthreads();
    solver_eval<DEBLUR>(rd, b, nC, nS, sh.ctl.x, rc.huber_a, DEBLUR, sh);
    if (tid == 0) sh.need = lm_init(sh.ctl, sh.sum, n_active);
    __syncthreads();
    while (sh.need) {
        solver_eval<DEBLUR>(rd, b, nC, nS, sh.ctl.cand, rc.huber_a, DEBLUR, sh);
        if (tid == 0) sh.need = lm_update(sh.ctl, sh.sum);
        __syncthreads();
    }
}


template <int DEBLUR>
__device__ void solve_general(const RegDev &rd, const RegConst &rc, int b, RegState *st, SolveShared &sh, unsigned long long *s_table)
{
    const int tid = threadIdx.x;
    const int nC = rd.n_corner[b], nS = rd.n_surf[b];
    const int total = nC + nS;
    const size_t sb = (size_t)b * rd.cap;
    const double *av = rd.blk_av + (size_t)b * 6 * rd.cap;

    // ---- census: active blocks, corner_avail / surf_avail (PCR:325,425) -----------------------------------
    {
        int na = 0, nca = 0, nsa = 0;
        for (int j = tid; j < total; j += RS_THREADS) {
            const int slot0 = slot_of(j, nC, rd.cap_c);
            const unsigned char fl = rd.blk_flag0[sb + slot0];
            rd.blk_flag[sb + slot0] = fl;  // working copy: the prune below clears BLK_ACTIVE in place
            na += (fl & BLK_ACTIVE) ? 1 : 0;
            if (fl & 8) {
                if (j < nC) nca++; else nsa++;
            }
        }
        na = block_sum_int(na, sh);
        nca = block_sum_int(nca, sh);
        nsa = block_sum_int(nsa, sh);
        if (rc.subsample_seed && na > rc.max_blocks) {  // a13: "Number of residual blocks too Large, drop them" (PCR:438-458)
            int kept = 0;
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot0 = slot_of(j, nC, rd.cap_c);
                const unsigned char fl = rd.blk_flag[sb + slot0];
                if (!(fl & BLK_ACTIVE)) continue;
                if (subsample_drop_block(rc.subsample_seed, st->icp_iters, j, na, rc.max_blocks))
                    rd.blk_flag[sb + slot0] = fl & ~BLK_ACTIVE;
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
        __syncthreads();
    }

    if (tid < 6) sh.tcyc[tid] = 0;
    __syncthreads();
    LL_T0(t_total);
    // ---- prerun solve (PCR:463-474) -------------------------------------------------------------------------
    {
        LL_T0(t_e);
        solver_lm<DEBLUR>(rd, rc, b, nC, nS, st->inc, rc.ceres_prerun_times, sh.n_active, sh);
        LL_TACC(0, t_e);
    }
    int lm_iters = sh.ctl.iteration;
    LL_T0(t_l1);

    // ---- loss-corrected L1 per block at the prerun result (PCR:476-483) -----------------------------------
    {
        LL_CTX_DECL(sh.ctl.x)
        for (int j = tid; j < total; j += RS_THREADS) {
            const int slot = slot_of(j, nC, rd.cap_c);
            const unsigned char fl = rd.blk_flag[sb + slot];
            if (!(fl & BLK_ACTIVE)) continue;
            const float4 ff = rd.blk_f[sb + slot];
            double a[3], v[3];
            av_load(av, rd.cap, slot, slot < rd.cap_c, a[0], a[1], a[2], v[0], v[1], v[2]);
            double l1v;
            LL_CTX_L1(l1v, fl & 3, ff, a, v, rc.huber_a, st->pose_last);
            rd.blk_l1[sb + slot] = l1v;
        }
    }
    __syncthreads();

    LL_TACC(2, t_l1);
    LL_T0(t_dd);
    // ---- std::set semantics: which L1 values are distinct (first occurrences get flag bit 16), how many ------------
    // Same scheme as the fast path -- LDS bitmap, contested keys through an exact table -- with the keys read back
    // from HBM and split by hash into partitions of at most ~FAST_MAX_BLOCKS keys, so the LDS tables keep their size.
    // Heavily duplicated inputs fall back to one compare-and-swap table in HBM.
    {
        unsigned int *bm = (unsigned int *)s_table;
        unsigned int *cb = bm + DD_BM_WORDS;
        unsigned long long *ex = (unsigned long long *)(cb + DD_CB_SIZE);
        const int parts = (total + FAST_MAX_BLOCKS - 1) / FAST_MAX_BLOCKS;
        int my = 0;
        bool overflow = false;
        for (int part = 0; part < parts && !overflow; part++) {
            __syncthreads();
            for (int e = tid; e < DD_BM_WORDS; e += RS_THREADS) bm[e] = 0u;
            for (int e = tid; e < DD_CB_SIZE; e += RS_THREADS) cb[e] = 0xffffffffu;
            for (int e = tid; e < DD_EX_SIZE; e += RS_THREADS) ex[e] = HASH_EMPTY;
            __syncthreads();
            int ncoll = 0;
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot = slot_of(j, nC, rd.cap_c);
                const unsigned char fl = rd.blk_flag[sb + slot];
                if (!(fl & BLK_ACTIVE)) continue;
                const double l1 = rd.blk_l1[sb + slot];
                if (!(l1 == l1)) continue;  // NaN never enters the set
                const unsigned long long hk = hash64((unsigned long long)__double_as_longlong(l1));
                if ((int)((hk >> 44) % (unsigned long long)parts) != part) continue;
                const unsigned int hb = (unsigned int)hk & (DD_BM_WORDS * 32 - 1);
                const unsigned int bit = 1u << (hb & 31);
                if (atomicOr(&bm[hb >> 5], bit) & bit) {
                    rd.blk_flag[sb + slot] = fl | 32;  // contested bit: its index goes to the set below
                    ncoll++;
                }
            }
            const int total_coll = block_sum_int(ncoll, sh);
            if (total_coll > DD_MAX_COLL) {
                overflow = true;
                break;
            }
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot = slot_of(j, nC, rd.cap_c);
                const unsigned char fl = rd.blk_flag[sb + slot];
                if (!(fl & 32)) continue;
                rd.blk_flag[sb + slot] = fl & ~32;
                const unsigned int hb = (unsigned int)hash64((unsigned long long)__double_as_longlong(rd.blk_l1[sb + slot])) & (DD_BM_WORDS * 32 - 1);
                unsigned int h = (hb * 2654435761u) >> (32 - DD_CB_LOG2);
                for (;;) {
                    const unsigned int old = atomicCAS(&cb[h], 0xffffffffu, hb);
                    if (old == 0xffffffffu || old == hb) break;
                    h = (h + 1u) & (DD_CB_SIZE - 1);
                }
            }
            __syncthreads();
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot = slot_of(j, nC, rd.cap_c);
                const unsigned char fl = rd.blk_flag[sb + slot];
                if (!(fl & BLK_ACTIVE)) continue;
                const double l1 = rd.blk_l1[sb + slot];
                if (!(l1 == l1)) continue;
                const unsigned long long key = (unsigned long long)__double_as_longlong(l1);
                const unsigned long long hk = hash64(key);
                if ((int)((hk >> 44) % (unsigned long long)parts) != part) continue;
                const unsigned int hb = (unsigned int)hk & (DD_BM_WORDS * 32 - 1);
                bool contested = false;
                unsigned int h = (hb * 2654435761u) >> (32 - DD_CB_LOG2);
                for (;;) {
                    const unsigned int c = cb[h];
                    if (c == 0xffffffffu) break;
                    if (c == hb) {
                        contested = true;
                        break;
                    }
                    h = (h + 1u) & (DD_CB_SIZE - 1);
                }
                bool first = !contested;
                if (contested) {
                    unsigned int h2 = (unsigned int)(hk >> 24) & (DD_EX_SIZE - 1);
                    for (;;) {
                        const unsigned long long old = atomicCAS(&ex[h2], HASH_EMPTY, key);
                        if (old == HASH_EMPTY) {
                            first = true;
                            break;
                        }
                        if (old == key) break;
                        h2 = (h2 + 1u) & (DD_EX_SIZE - 1);
                    }
                }
                if (first) {
                    rd.blk_flag[sb + slot] = fl | 16;
                    my++;
                }
            }
        }
        if (overflow) {  // uniform: every thread saw the same total_coll
            __syncthreads();
            my = 0;
            unsigned long long *table = rd.hash + (size_t)b * rd.hash_cap;
            for (int k = tid; k < rd.hash_cap; k += RS_THREADS) table[k] = HASH_EMPTY;
            __syncthreads();
            const unsigned long long mask = (unsigned long long)rd.hash_cap - 1ull;
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot = slot_of(j, nC, rd.cap_c);
                const unsigned char fl0 = rd.blk_flag[sb + slot] & ~(16 | 32);
                rd.blk_flag[sb + slot] = fl0;
                if (!(fl0 & BLK_ACTIVE)) continue;
                const double l1 = rd.blk_l1[sb + slot];
                if (!(l1 == l1)) continue;
                const unsigned long long key = (unsigned long long)__double_as_longlong(l1);
                unsigned long long h = hash64(key) & mask;
                for (;;) {
                    const unsigned long long old = atomicCAS(&table[h], HASH_EMPTY, key);
                    if (old == HASH_EMPTY) {
                        rd.blk_flag[sb + slot] = fl0 | 16;
                        my++;
                        break;
                    }
                    if (old == key) break;
                    h = (h + 1ull) & mask;
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
        // rank select of the distinct values: value-range bins in LDS, then an exact ranking of the selected bin's keys
        // (the fast path's scheme, keys read from HBM); a crowded bin falls back to the radix select below
        int *bins = (int *)s_table;
        unsigned long long *cand = s_table + SEL_BINS / 2;
        const int lane = tid & 63, wave = tid >> 6;
        double kmin = INFINITY, kmax = -INFINITY;
        for (int j = tid; j < total; j += RS_THREADS) {
            const int slot = slot_of(j, nC, rd.cap_c);
            if ((rd.blk_flag[sb + slot] & (BLK_ACTIVE | 16)) != (BLK_ACTIVE | 16)) continue;
            const double l1 = rd.blk_l1[sb + slot];
            kmin = fmin(kmin, l1);
            kmax = fmax(kmax, l1);
        }
        for (int off = 32; off > 0; off >>= 1) {
            kmin = fmin(kmin, __shfl_down(kmin, off));
            kmax = fmax(kmax, __shfl_down(kmax, off));
        }
        __syncthreads();
        if (lane == 0) {
            sh.red[wave][0] = kmin;
            sh.red[wave][1] = kmax;
        }
        for (int e = tid; e < SEL_BINS; e += RS_THREADS) bins[e] = 0;
        __syncthreads();
        double lo = sh.red[0][0], hi = sh.red[0][1];
        for (int w = 1; w < RS_WAVES; w++) {
            lo = fmin(lo, sh.red[w][0]);
            hi = fmax(hi, sh.red[w][1]);
        }
        const double scale = (hi > lo) ? (double)(SEL_BINS - 1) / (hi - lo) : 0.0;
        for (int j = tid; j < total; j += RS_THREADS) {
            const int slot = slot_of(j, nC, rd.cap_c);
            if ((rd.blk_flag[sb + slot] & (BLK_ACTIVE | 16)) != (BLK_ACTIVE | 16)) continue;
            int bi = (int)((rd.blk_l1[sb + slot] - lo) * scale);
            bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
            atomicAdd(&bins[bi], 1);
        }
        __syncthreads();
        {
            const int per = SEL_BINS / RS_THREADS;
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
            __syncthreads();
            const bool last_thread = tid == RS_THREADS - 1;
            if ((rank >= below && rank < below + part) || (last_thread && rank >= below + part)) {
                int cum = below, bi = tid * per;
                for (; bi < tid * per + per - 1; bi++) {
                    if (cum + bins[bi] > rank) break;
                    cum += bins[bi];
                }
                sh.sel_bin = bi;
                sh.sel_rank = rank - cum;
                sh.sel_cnt = bins[bi];
            }
            __syncthreads();
        }
        if (sh.sel_cnt <= SEL_CAND) {
            const int sel_bin = sh.sel_bin;
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot = slot_of(j, nC, rd.cap_c);
                if ((rd.blk_flag[sb + slot] & (BLK_ACTIVE | 16)) != (BLK_ACTIVE | 16)) continue;
                const double l1 = rd.blk_l1[sb + slot];
                int bi = (int)((l1 - lo) * scale);
                bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                if (bi == sel_bin) cand[atomicAdd(&sh.n_cand, 1)] = (unsigned long long)__double_as_longlong(l1);
            }
            __syncthreads();
            const int m = sh.n_cand;
            for (int i = tid; i < m; i += RS_THREADS) {
                const unsigned long long ki = cand[i];
                int rk = 0;
                for (int jj = 0; jj < m; jj++) rk += (cand[jj] < ki) ? 1 : 0;  // keys are distinct
                if (rk == sh.sel_rank) sh.sel_prefix = ki;
            }
            __syncthreads();
        } else {
        // MSB-first radix select (8 bits per pass) over the distinct keys of that bin; non-negative doubles order like uint64
        const int sel_bin = sh.sel_bin;
        if (tid == 0) sh.sel_prefix = 0ull;
        __syncthreads();
        for (int pass = 0; pass < 8; pass++) {
            const int shift = 56 - 8 * pass;
            for (int k = tid; k < 256; k += RS_THREADS) sh.hist[k] = 0;
            __syncthreads();
            const unsigned long long prefix = sh.sel_prefix;
            for (int j = tid; j < total; j += RS_THREADS) {
                const int slot = slot_of(j, nC, rd.cap_c);
                if ((rd.blk_flag[sb + slot] & (BLK_ACTIVE | 16)) != (BLK_ACTIVE | 16)) continue;
                const double l1 = rd.blk_l1[sb + slot];
                int bi = (int)((l1 - lo) * scale);
                bi = bi < 0 ? 0 : (bi > SEL_BINS - 1 ? SEL_BINS - 1 : bi);
                if (bi != sel_bin) continue;
                const unsigned long long key = (unsigned long long)__double_as_longlong(l1);
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
    // ---- prune (PCR:487-499) ---------------------------------------------------------------------------------
    {
        const double thr = sh.thr;
        int na = 0;
        for (int j = tid; j < total; j += RS_THREADS) {
            const int slot = slot_of(j, nC, rd.cap_c);
            unsigned char fl = rd.blk_flag[sb + slot];
            if (!(fl & BLK_ACTIVE)) continue;
            fl &= ~16;
            if (rd.blk_l1[sb + slot] > thr)
                fl &= ~BLK_ACTIVE;
            else
                na++;
            rd.blk_flag[sb + slot] = fl;
        }
        na = block_sum_int(na, sh);
        if (tid == 0) sh.n_active = na;
        __syncthreads();
    }

    // ---- final solve (PCR:501-508) -----------------------------------------------------------------------------
    {
        // the prerun result is the start; copy it out of ctl before lm_begin overwrites ctl.x
        __shared__ double x_start[7];
        if (tid < 7) x_start[tid] = sh.ctl.x[tid];
        __syncthreads();
        LL_TACC(4, t_sel);
        LL_T0(t_e);
        solver_lm<DEBLUR>(rd, rc, b, nC, nS, x_start, rc.ceres_max_iterations, sh.n_active, sh);
        LL_TACC(0, t_e);
    }
    lm_iters += sh.ctl.iteration;

    solve_epilogue(rc, st, sh, lm_iters);
#ifdef LL_SOLVE_TIMING
    LL_TACC(5, t_total);
    if (tid == 0)
        for (int i = 0; i < 6; i++) st->dbg_cycles[i] += sh.tcyc[i];
#endif
}



template <int DEBLUR>
__device__ void solve_big(const RegDev &rd, const RegConst &rc, const f4 *map_pts, int b, RegState *st, SolveShared &sh, uint4 *s_raw)
{
    const int tid = threadIdx.x;
    const int nC = rd.n_corner[b], nS = rd.n_surf[b];
    if (tid < 16) sh.tcyc[tid] = 0;
    __syncthreads();
    LL_T0(t_total);
    // ---- flags -> activity mask, census (PCR:325,425), the scan's plane table ----
    Act2 act = big_census_and_plane_table(rd, rc, map_pts, b, st, nC, nS, s_raw, sh);
    // ---- prerun solve (PCR:463-474); its last evaluation also leaves the per-block L1 values in blk_l1 ----
    big_lm<true, DEBLUR>(rd, rc, b, nC, nS, st->inc, rc.ceres_prerun_times, sh.n_active, act, s_raw, st->pose_last, sh);
    int lm_iters = sh.ctl.iteration;
    // ---- inlier threshold and prune (PCR:476-499); overwrites the LDS plane table ----
    act = big_inlier<DEBLUR>(rd, rc, b, st, sh, s_raw, act, nC, nS);
    // ---- final solve (PCR:501-508) ----
    {
        __shared__ double x_start_b[7];
        if (tid < 7) x_start_b[tid] = sh.ctl.x[tid];
        plane_table_reload(rd, b, false, s_raw, sh);  // (its barrier also publishes x_start_b)
        big_lm<false, DEBLUR>(rd, rc, b, nC, nS, x_start_b, rc.ceres_max_iterations, sh.n_active, act, s_raw, st->pose_last, sh);
    }
    lm_iters += sh.ctl.iteration;
    solve_epilogue(rc, st, sh, lm_iters);
#ifdef LL_SOLVE_TIMING
    LL_TACC(5, t_total);
    if (tid == 0)
        for (int i = 0; i < 16; i++) st->dbg_cycles[i] += sh.tcyc[i];
#endif
}

// One workgroup per scan: motion-deblur batches and batches with a scan beyond solve_fast3's 24 576 blocks (launch_reg_solve decides per
// batch; reg_solve_kernel keeps the Mid-40 batches).  Per scan: the plane-table path above, or solve_general for what it does not hold.
template <int DEBLUR>
__global__ __launch_bounds__(RS_THREADS) void reg_solve_big_kernel(RegDev rd, RegConst rc, const f4 *map_surf)
{
    __shared__ SolveShared sh;
    __shared__ uint4 s_raw[PT_LDS_BYTES / 16];
    static_assert(PT_LDS_BYTES >= HT_SIZE * 8, "s_raw holds the general path's tables");
    const int b = blockIdx.x;
    RegState *st = rd.state + b;
    if (st->done) return;
    if (threadIdx.x == 0) {
        sh.grp_g = 0;
        sh.grp_G = 1;
        sh.grp_seq = 0;
        sh.xch_seq = 0;
        sh.xch_epoch = rc.xch_epoch;
        sh.grp_abort = 0;
    }
    __syncthreads();
    if (scan_is_compact(rd, rc, b))
        solve_big<DEBLUR>(rd, rc, map_surf, b, st, sh, s_raw);
    else
        solve_general<DEBLUR>(rd, rc, b, st, sh, (unsigned long long *)s_raw);
}
