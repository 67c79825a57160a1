// ll_reg_solve_common.h -- device helpers shared by the solver kernels (ll_reg_kernels.hip: one 512-thread workgroup per scan, and
// ll_reg_small_kernels.hip: one wavefront / four wavefronts per small scan): the wavefront reduction of the 28 accumulators of a cost
// evaluation, the Levenberg-Marquardt controller step with the line-search fit on the controller's whole wavefront, and the epilogue of
// one ICP iteration (pose composition, convergence test: point_cloud_registration.hpp:509-531).  Device only.
#pragma once
#include <hip/hip_runtime.h>

#include "ll_reg_query.h"

namespace ll {

// The 28 accumulators of a cost evaluation summed over the wavefront -> red[0 .. 27].  Not 28 six-step trees (168 shifted adds, each
// a pair of DPP moves per 64-bit operand plus the moves that feed them: ~850 VALU instructions per evaluation and wavefront, a
// sixth of the evaluation's issue slots on a C2 scan and half of them on a voxel-filtered one) but ONE butterfly over the whole
// set: at every step a lane keeps the half of its values whose index bit matches its lane bit and hands the other half to the
// partner lane (ds_bpermute: the LDS crossbar, not a VALU slot), so the work halves with the distance -- 16 + 8 + 4 + 2 + 1 + 1 adds.
// Value i ends up in lanes 2i and 2i + 1.  The addition tree is fixed (pairs 32 apart first, then 16, 8, 4, 2, 1), so sums are
// reproducible run to run; a + b and b + a are the same bits, so both lanes of a pair agree.
__device__ __forceinline__ void wave_sum_acc(const double (&acc)[LL_NACC], double *red, int lane)
{
    static_assert(LL_NACC <= 32 && LL_NACC > 16, "butterfly over 32 value slots");
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
    double w16[16], w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const double hi = (16 + i < LL_NACC) ? acc[16 + i] : 0.0;
        const double keep = b5 ? hi : acc[i], send = b5 ? acc[i] : hi;
        w16[i] = keep + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const double keep = b4 ? w16[8 + i] : w16[i], send = b4 ? w16[i] : w16[8 + i];
        w8[i] = keep + __shfl_xor(send, 16);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double keep = b3 ? w8[4 + i] : w8[i], send = b3 ? w8[i] : w8[4 + i];
        w4[i] = keep + __shfl_xor(send, 8);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double keep = b2 ? w4[2 + i] : w4[i], send = b2 ? w4[i] : w4[2 + i];
        w2[i] = keep + __shfl_xor(send, 4);
    }
    const double keep1 = b1 ? w2[1] : w2[0], send1 = b1 ? w2[0] : w2[1];
    const double w1 = keep1 + __shfl_xor(send1, 2);
    const double tot = w1 + __shfl_xor(w1, 1);
    const int idx = (lane >> 1) & 31;
    if (!(lane & 1) && idx < LL_NACC) red[idx] = tot;
}

// compute_interpolatation_rodrigue, PCR:607-620 (Eigen AngleAxis from quaternion)
__device__ inline void compute_interp(const double q[4], RegState *st)
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    double axis[3];
    if (q[3] < 0) n = -n;
    if (n != 0.0) {
        st->interp_theta = 2.0 * atan2(n, fabs(q[3]));
        axis[0] = q[0] / n;
        axis[1] = q[1] / n;
        axis[2] = q[2] / n;
    } else {
        st->interp_theta = 0.0;
        axis[0] = 1.0;
        axis[1] = 0.0;
        axis[2] = 0.0;
    }
    const double an = sqrt(dot3(axis, axis));
    axis[0] /= an;
    axis[1] /= an;
    axis[2] /= an;
    for (int i = 0; i < 9; i++) st->hat[i] = 0.0;
    st->hat[1] = -axis[2];
    st->hat[3] = axis[2];
    st->hat[2] = axis[1];
    st->hat[6] = -axis[1];
    st->hat[5] = -axis[0];
    st->hat[7] = axis[0];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += st->hat[i * 3 + k] * st->hat[k * 3 + j];
            st->hat_sq[i * 3 + j] = s;
        }
}

// pose composition, convergence test and per-iteration report (PCR:509-531); lane 0 only
template <class SH>
__device__ void solve_epilogue(const RegConst &rc, RegState *st, SH &sh, int lm_iters)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        for (int i = 0; i < 7; i++) st->inc[i] = sh.ctl.x[i];
        if (rc.if_motion_deblur) compute_interp(st->inc, st);
        double tw[3];
        quat_rot(st->pose_last, &st->inc[4], tw);  // PCR:514
        st->pose_curr[4] = tw[0] + st->pose_last[4];
        st->pose_curr[5] = tw[1] + st->pose_last[5];
        st->pose_curr[6] = tw[2] + st->pose_last[6];
        double qc[4];
        quat_mul(st->pose_last, st->inc, qc);  // PCR:515
        for (int i = 0; i < 4; i++) st->pose_curr[i] = qc[i];
        st->angular_diff = (double)((float)quat_angular_distance(qc, st->pose_last)) * 57.3;  // PCR:517
        const double dt[3] = {st->pose_curr[4] - st->pose_last[4], st->pose_curr[5] - st->pose_last[5],
                              st->pose_curr[6] - st->pose_last[6]};
        st->t_diff = sqrt(dot3(dt, dt));
        st->final_cost = sh.ctl.final_cost;
        st->initial_cost = sh.ctl.initial_cost;
        st->inlier_thr = sh.thr;
        st->n_blocks_last = sh.n_active;
        st->corner_avail = sh.n_corner_avail;
        st->surf_avail = sh.n_surf_avail;
        st->lm_total += lm_iters;
        st->icp_iters += 1;
        const double dto[3] = {st->prev_t[0] - st->inc[4], st->prev_t[1] - st->inc[5], st->prev_t[2] - st->inc[6]};
        const bool conv = quat_angular_distance(st->prev_q, st->inc) < 57.3 * rc.minimum_icp_R_diff &&
                          sqrt(dot3(dto, dto)) < rc.minimum_icp_T_diff;  // PCR:521-522
        if (conv && !rc.force_all_iterations) {
            st->done = 1;
        } else {
            for (int i = 0; i < 4; i++) st->prev_q[i] = st->inc[i];
            for (int i = 0; i < 3; i++) st->prev_t[i] = st->inc[4 + i];
        }
        if (st->icp_iters >= rc.icp_max_iterations) st->done = 1;
    }
}

// lm_quintic_min_step (ll_reg_core.h) on the controller's wavefront.  The sequential form isolates the roots level by level -- the
// cubic p'' on the (at most three) intervals between the roots of the quadratic p''', then the quartic p' on the (at most four)
// intervals between the roots of p'' -- bisecting one interval after the other: up to seven dependent 60-step bisections on one lane
// while the workgroup -- and the launch, whose length is its slowest scan's -- waits.  Here lane i takes interval i of a level, so a fit is
// two bisections deep; the break points travel by __shfl.  Every lane runs the very functions of the sequential form on the same
// operands, and the roots are gathered in interval order, so the step has the same bits (compared on random and adversarial fits by
// tests/test_gpu_reg.py through ll_debug_quintic).  All 64 lanes must call it.
__device__ __forceinline__ double lm_quintic_min_step_wave(double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2,
                                                           double lo, double hi, int lane)
{
    Quintic q;
    if (!quintic_fit(f0, g0, x1, f1, g1, x2, f2, g2, q)) return fmin(fmax(0.5 * x1, lo), hi);  // (uniform: every lane has the same arguments)
    QuinticChain c;
    quintic_chain(q, c);
    double r3[2] = {hi, hi};
    const int n3 = quintic_quadratic_roots(c.A, c.B, c.C, lo, hi, r3);  // (uniform)
    // ---- p'' on the intervals [lo, r3_0], [r3_0, r3_1], [r3_1, hi] (the ones beyond n3 do not exist) ----
    const double P1 = n3 >= 1 ? r3[0] : hi, P2 = n3 >= 2 ? r3[1] : hi;
    double R0, R1, R2;
    int n2;
    {
        const double a = lane == 0 ? lo : (lane == 1 ? P1 : P2), b = lane == 0 ? P1 : (lane == 1 ? P2 : hi);
        double root = 0.0;
        const bool has = lane <= n3 && lane < 3 && quintic_interval_root(c.d2, a, b, quintic_poly4(c.d2, a), quintic_poly4(c.d2, b), &root);
        const int h0 = __shfl((int)has, 0), h1 = __shfl((int)has, 1), h2 = __shfl((int)has, 2);
        const double y0 = __shfl(root, 0), y1 = __shfl(root, 1), y2 = __shfl(root, 2);
        n2 = h0 + h1 + h2;
        R0 = h0 ? y0 : (h1 ? y1 : y2);
        R1 = (h0 && h1) ? y1 : y2;
        R2 = y2;
        const double last = n2 == 3 ? R2 : (n2 == 2 ? R1 : R0);
        if (n2 > 0 && !(last < hi)) n2--;  // (a break point lies strictly inside; hi closes the last interval anyway)
    }
    // ---- p' on the intervals between lo, the roots of p'' and hi ----
    const double Q1 = n2 >= 1 ? R0 : hi, Q2 = n2 >= 2 ? R1 : hi, Q3 = n2 >= 3 ? R2 : hi;
    const double a = lane == 0 ? lo : (lane == 1 ? Q1 : (lane == 2 ? Q2 : Q3)), b = lane == 0 ? Q1 : (lane == 1 ? Q2 : (lane == 2 ? Q3 : hi));
    double root = 0.0;
    const bool has = lane <= n2 && lane < 4 && quintic_interval_root(c.dq, a, b, quintic_poly4(c.dq, a), quintic_poly4(c.dq, b), &root);
    // ---- MinimizePolynomial's choice (quintic_pick), the roots in interval order: lanes 0 .. 3 evaluate the interpolant at their root, lanes 4 / 5
    //      at the interval's ends -- one sweep for all six values instead of six in a row; the comparisons then run as in the sequential form ----
    const double xe = lane == 4 ? lo : (lane == 5 ? hi : root);
    double ve, de;
    quintic_eval(q, xe, ve, de);
    (void)de;
    double best_x = lo, best_v = __shfl(ve, 4);
    const double vh = __shfl(ve, 5);
    if (!(best_v < vh)) {
        best_v = vh;
        best_x = hi;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int hi_ = __shfl((int)has, i);
        const double xi = __shfl(root, i), v = __shfl(ve, i);
        if (hi_ && v < best_v) {  // (uniform)
            best_v = v;
            best_x = xi;
        }
    }
    return best_x;
}

// the fit with its ten arguments in LDS (written by lane 0).  (Out of line -- a real call inside the solver kernel -- it cost the WHOLE
// kernel half of its speed: 4.6 ms of solver per step against 3.0; the kernel then carries the calling convention's scratch set-up
// and the allocator's call-clobber constraints through every phase.  Inlined, as everything else in this kernel.)
__device__ __forceinline__ double lm_quintic_min_step_wave_call(const double *a, int lane)
{
    return lm_quintic_min_step_wave(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], lane);
}

// lm_update with the three-sample fit on the wavefront: called by every lane of the controller's wavefront (lane 0 holds the controller)
__device__ __forceinline__ int lm_update_wave(LmCtl &c, const double *sum, double *fit, int lane)
{
    int code = 0;
    double e[LL_NACC];  // lane 0's register copy of the evaluation
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < LL_NACC; i++) e[i] = sum[i];
        double cur_cost = 0.0, cg = 0.0;
        code = lm_update_pre(c, e, &cur_cost, &cg);
        if (code == LM_FIT) {
            fit[0] = c.cost, fit[1] = c.gd, fit[2] = c.ls_step, fit[3] = cur_cost, fit[4] = cg, fit[5] = c.ls_prev_x, fit[6] = c.ls_prev_f, fit[7] = c.ls_prev_g;
            fit[8] = 1e-3 * c.ls_step, fit[9] = 0.6 * c.ls_step;
        }
    }
    if (__shfl(code, 0) == LM_FIT) {  // (uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // lane 0's LDS stores before the wavefront's loads (same wavefront: in order)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double step = lm_quintic_min_step_wave_call(fit, lane);
        if (lane == 0) code = lm_update_post(c, fit[3], fit[4], step);
    }
    int r = 0;
    if (lane == 0) r = lm_update_close(c, e, code);
    return r;  // (lane 0's is the answer)
}

}  // namespace ll
