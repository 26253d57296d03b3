// lpd_loss.hip -- lazy triplet / quadruplet loss, forward and gradient in ONE launch.
//
// Replaces loss/pointnetvlad_loss.py:6-97 (best_pos_distance, triplet_loss, quadruplet_loss): the
// reference issues ~25 tiny elementwise/reduction kernels plus their autograd graph per step for a
// [bq, P+Ng+2, 256] tensor; here one 256-thread block computes the squared distances, the best
// positive, the hinges, the lazy max / sum, the mean or hard-count normalisation AND the gradient
// w.r.t. all four inputs.
//
//   d_pos[b][p] = |pos[b][p] - q[b]|^2 ; positive[b] = min_p or max_p            (:6-12,:53-56)
//   x1[b][n] = m1 + positive[b] - |neg[b][n] - q[b]|^2      ; L1 = clamp(x1, 0)    (:64-65)
//   x2[b][n] = m2 + positive[b] - |neg[b][n] - other[b]|^2  ; L2 = clamp(x2, 0)    (:81-82)
//   t[b] = max_n L (lazy) or sum_n L                                               (:67-70,:83-86)
//   loss = mean_b t  or  sum_b t / (count(t > 1e-16) + 1e-16)                      (:72-78,:88-94)
// Gradient conventions follow torch: clamp passes gradient where x >= 0, max routes to the first
// arg-max, min/max over positives to the first arg-min/arg-max, the hard count is a constant.
#include "lpd_common.h"
#include <math.h>

namespace {

struct LossArgs {
    const float* q;      // element (b, 0, d) at q + b*q_sb + d
    const float* pos;    // (b, p, d) at pos + b*pos_sb + p*pos_st + d
    const float* neg;
    const float* other;  // may be null when !quad
    long long q_sb, pos_sb, pos_st, neg_sb, neg_st, other_sb;
    int bq, P, Ng, D;
    float m1, m2;
    int use_min, lazy, ignore_zero, quad;
    float* loss;         // [1]
    float* minmax;       // [2][bq] min_pos then max_pos
    float* gq;           // [bq][D]
    float* gpos;         // [bq][P][D]
    float* gneg;         // [bq][Ng][D]
    float* gother;       // [bq][D] (quad only)
};

__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void metric_loss_kernel(LossArgs a)
{
    extern __shared__ float sm[];
    // layout: dpos[bq][P], dneg[bq][Ng], d2[bq][Ng], t1[bq], t2[bq], w[2], red[4]
    float* dpos = sm;
    float* dneg = dpos + a.bq * a.P;
    float* d2 = dneg + a.bq * a.Ng;
    float* t1 = d2 + a.bq * a.Ng;
    float* t2 = t1 + a.bq;
    float* w = t2 + a.bq;
    float* red = w + 2;
    const int tid = threadIdx.x;

    // ---- pass 1: squared distances ----
    for (int b = 0; b < a.bq; ++b) {
        const float* qb = a.q + b * a.q_sb;
        const float* ob = a.quad ? a.other + b * a.other_sb : nullptr;
        for (int p = 0; p < a.P; ++p) {
            const float* v = a.pos + b * a.pos_sb + p * a.pos_st;
            float s = 0.f;
            for (int d = tid; d < a.D; d += 256) { float e = v[d] - qb[d]; s += e * e; }
            s = block_sum(s, red);
            if (tid == 0) dpos[b * a.P + p] = s;
        }
        for (int n = 0; n < a.Ng; ++n) {
            const float* v = a.neg + b * a.neg_sb + n * a.neg_st;
            float s = 0.f, s2 = 0.f;
            for (int d = tid; d < a.D; d += 256) {
                float e = v[d] - qb[d];
                s += e * e;
                if (ob) { float e2 = v[d] - ob[d]; s2 += e2 * e2; }
            }
            s = block_sum(s, red);
            s2 = block_sum(s2, red);
            if (tid == 0) { dneg[b * a.Ng + n] = s; d2[b * a.Ng + n] = s2; }
        }
    }
    __syncthreads();

    // ---- scalar stage (thread 0): per-query terms, loss, normalisation weights ----
    if (tid == 0) {
        float sum1 = 0.f, sum2 = 0.f, c1 = 0.f, c2 = 0.f;
        for (int b = 0; b < a.bq; ++b) {
            float mn = dpos[b * a.P], mx = dpos[b * a.P];
            for (int p = 1; p < a.P; ++p) { mn = fminf(mn, dpos[b * a.P + p]); mx = fmaxf(mx, dpos[b * a.P + p]); }
            a.minmax[b] = mn;
            a.minmax[a.bq + b] = mx;
            const float positive = a.use_min ? mn : mx;
            float v1 = a.lazy ? -INFINITY : 0.f, v2 = a.lazy ? -INFINITY : 0.f;
            for (int n = 0; n < a.Ng; ++n) {
                float l1 = fmaxf(a.m1 + positive - dneg[b * a.Ng + n], 0.f);
                float l2 = fmaxf(a.m2 + positive - d2[b * a.Ng + n], 0.f);
                v1 = a.lazy ? fmaxf(v1, l1) : v1 + l1;
                v2 = a.lazy ? fmaxf(v2, l2) : v2 + l2;
            }
            t1[b] = v1; t2[b] = v2;
            sum1 += v1; sum2 += v2;
            c1 += v1 > 1e-16f ? 1.f : 0.f;
            c2 += v2 > 1e-16f ? 1.f : 0.f;
        }
        float w1, w2, loss;
        if (a.ignore_zero) { w1 = 1.f / (c1 + 1e-16f); w2 = 1.f / (c2 + 1e-16f); }
        else { w1 = 1.f / a.bq; w2 = 1.f / a.bq; }
        loss = sum1 * w1;
        if (a.quad) loss += sum2 * w2;
        else w2 = 0.f;
        a.loss[0] = loss;
        w[0] = w1; w[1] = w2;
    }
    __syncthreads();
    const float w1 = w[0], w2 = w[1];

    // ---- pass 2: gradients ----
    for (int b = 0; b < a.bq; ++b) {
        // best positive (first arg-min / arg-max)
        int pstar = 0;
        for (int p = 1; p < a.P; ++p) {
            bool better = a.use_min ? dpos[b * a.P + p] < dpos[b * a.P + pstar] : dpos[b * a.P + p] > dpos[b * a.P + pstar];
            if (better) pstar = p;
        }
        const float positive = dpos[b * a.P + pstar];
        // lazy: first arg-max of the clamped hinge
        int n1 = 0, n2 = 0;
        if (a.lazy) {
            float b1 = -INFINITY, b2 = -INFINITY;
            for (int n = 0; n < a.Ng; ++n) {
                float l1 = fmaxf(a.m1 + positive - dneg[b * a.Ng + n], 0.f);
                float l2 = fmaxf(a.m2 + positive - d2[b * a.Ng + n], 0.f);
                if (l1 > b1) { b1 = l1; n1 = n; }
                if (l2 > b2) { b2 = l2; n2 = n; }
            }
        }
        float cpos = 0.f;  // coefficient on d(positive)
        for (int n = 0; n < a.Ng; ++n) {
            bool a1 = (a.m1 + positive - dneg[b * a.Ng + n]) >= 0.f && (!a.lazy || n == n1);
            bool a2 = a.quad && (a.m2 + positive - d2[b * a.Ng + n]) >= 0.f && (!a.lazy || n == n2);
            cpos += (a1 ? w1 : 0.f) + (a2 ? w2 : 0.f);
        }
        const float* qb = a.q + b * a.q_sb;
        const float* ob = a.quad ? a.other + b * a.other_sb : nullptr;
        const float* ps = a.pos + b * a.pos_sb + pstar * a.pos_st;
        for (int d = tid; d < a.D; d += 256) {
            const float qd = qb[d];
            const float dp = 2.f * (ps[d] - qd);
            float gqd = -cpos * dp;
            float god = 0.f;
            for (int p = 0; p < a.P; ++p) a.gpos[((size_t)b * a.P + p) * a.D + d] = (p == pstar) ? cpos * dp : 0.f;
            for (int n = 0; n < a.Ng; ++n) {
                const float nv = a.neg[b * a.neg_sb + n * a.neg_st + d];
                bool a1 = (a.m1 + positive - dneg[b * a.Ng + n]) >= 0.f && (!a.lazy || n == n1);
                bool a2 = a.quad && (a.m2 + positive - d2[b * a.Ng + n]) >= 0.f && (!a.lazy || n == n2);
                float g = 0.f;
                if (a1) { float e = 2.f * (nv - qd); g -= w1 * e; gqd += w1 * e; }
                if (a2) { float e = 2.f * (nv - ob[d]); g -= w2 * e; god += w2 * e; }
                a.gneg[((size_t)b * a.Ng + n) * a.D + d] = g;
            }
            a.gq[(size_t)b * a.D + d] = gqd;
            if (a.quad) a.gother[(size_t)b * a.D + d] = god;
        }
    }
}

// Backward of best_pos_distance (loss/pointnetvlad_loss.py:6-12): min_pos[b] = min_p |pos_p - q|^2, max_pos[b] = max_p ...;
// given the upstream gradients gmin[b], gmax[b]:  d/dq = -2 gmin (pos_amin - q) - 2 gmax (pos_amax - q),
// d/dpos_amin += 2 gmin (pos_amin - q), d/dpos_amax += 2 gmax (pos_amax - q), zero for the other positives.
// First arg-min / arg-max, like torch.min / torch.max.  One block per query tuple.
__global__ __launch_bounds__(256) void best_pos_bwd_kernel(const float* __restrict__ q, long long q_sb, const float* __restrict__ pos,
                                                           long long pos_sb, long long pos_st, const float* __restrict__ gmin,
                                                           const float* __restrict__ gmax, int P, int D, float* __restrict__ gq,
                                                           float* __restrict__ gpos)
{
    __shared__ float d2[64];
    __shared__ float part[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qb = q + b * q_sb;
    for (int p = 0; p < P; ++p) {
        const float* pp = pos + b * pos_sb + p * pos_st;
        float s = 0.f;
        for (int d = tid; d < D; d += 256) { const float t = pp[d] - qb[d]; s += t * t; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (tid == 0) d2[p] = (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
    }
    int amin = 0, amax = 0;
    for (int p = 1; p < P; ++p) {
        if (d2[p] < d2[amin]) amin = p;
        if (d2[p] > d2[amax]) amax = p;
    }
    const float gm = gmin[b], gx = gmax[b];
    for (int d = tid; d < D; d += 256) {
        const float qd = qb[d];
        const float dmin = pos[b * pos_sb + amin * pos_st + d] - qd, dmax = pos[b * pos_sb + amax * pos_st + d] - qd;
        gq[(long long)b * D + d] = -2.0f * (gm * dmin + gx * dmax);
        for (int p = 0; p < P; ++p)
            gpos[((long long)b * P + p) * D + d] = (p == amin ? 2.0f * gm * dmin : 0.0f) + (p == amax ? 2.0f * gx * dmax : 0.0f);
    }
}

}  // namespace

extern "C" int lpd_metric_loss(const float* q, long long q_sb, const float* pos, long long pos_sb, long long pos_st,
                               const float* neg, long long neg_sb, long long neg_st, const float* other,
                               long long other_sb, int bq, int P, int Ng, int D, float m1, float m2, int use_min,
                               int lazy, int ignore_zero, int quad, float* loss, float* minmax, float* gq, float* gpos,
                               float* gneg, float* gother, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(q && pos && neg && loss && minmax && gq && gpos && gneg, "lpd_metric_loss: null pointer");
    LPD_CHECK_ARG(!quad || (other && gother), "lpd_metric_loss: quadruplet form needs other_neg and its gradient buffer");
    LPD_CHECK_ARG(bq > 0 && P > 0 && Ng > 0 && D > 0, "lpd_metric_loss: bad dims bq=%d P=%d Ng=%d D=%d", bq, P, Ng, D);
    size_t floats = (size_t)bq * (P + 2 * Ng) + 2 * (size_t)bq + 2 + 4;
    LPD_CHECK_ARG(floats * sizeof(float) <= 60 * 1024, "lpd_metric_loss: bq*(P+2*Ng)=%d too large for one block", bq * (P + 2 * Ng));
    LossArgs a{q, pos, neg, other, q_sb, pos_sb, pos_st, neg_sb, neg_st, other_sb, bq, P, Ng, D, m1, m2,
               use_min, lazy, ignore_zero, quad, loss, minmax, gq, gpos, gneg, gother};
    hipLaunchKernelGGL(metric_loss_kernel, dim3(1), dim3(256), floats * sizeof(float), stream, a);
    LPD_CHECK_LAUNCH("lpd_metric_loss");
    return LPD_OK;
}

extern "C" int lpd_best_pos_bwd(const float* q, long long q_sb, const float* pos, long long pos_sb, long long pos_st, const float* gmin,
                                const float* gmax, int bq, int P, int D, float* gq, float* gpos, void* stream_)
{
    LPD_CHECK_ARG(q && pos && gmin && gmax && gq && gpos, "lpd_best_pos_bwd: null pointer");
    LPD_CHECK_ARG(bq > 0 && P > 0 && P <= 64 && D > 0, "lpd_best_pos_bwd: bad dims bq=%d P=%d D=%d (P <= 64)", bq, P, D);
    hipLaunchKernelGGL(best_pos_bwd_kernel, dim3(bq), dim3(256), 0, (hipStream_t)stream_, q, q_sb, pos, pos_sb, pos_st, gmin, gmax, P, D,
                       gq, gpos);
    LPD_CHECK_LAUNCH("lpd_best_pos_bwd");
    return LPD_OK;
}
