// lpd_train.hip -- bandwidth-bound kernels of the training path (forward in train mode + backward).
//
// What autograd does implicitly for the reference's `loss.backward()` (train_pointnetvlad.py:129,158)
// through BatchNorm (batch statistics over all B*N points or all B*N*k edges), LeakyReLU, max over k,
// the gather of get_graph_feature (lpdnet_model.py:350-357), softmax and the NetVLAD normalisations
// (PointNetVlad.py:51-74) is written out here as explicit kernels; the dense products of the backward
// pass (dX = dY W, dW = dY^T X) reuse lpd_gemm.hip.
//
// BatchNorm statistics and the backward reductions accumulate in fp64 (per-thread fp64 partials,
// fp64 global atomics): the sums run over up to B*N*k = 3.6 M rows.
#include "lpd_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float act_grad(float pre, int act, float slope)
{
    if (act == 3) {      // uniform; lpd_sigmoid is not speculatable, so the common cases below stay two instructions
        const float s = lpd_sigmoid(pre);
        return s * (1.0f - s);
    }
    return pre > 0.0f ? 1.0f : lpd_neg_slope(act, slope);
}

// the same with the sigmoid case resolved at compile time (SIG) and the negative-side slope formed once (ns = lpd_neg_slope(act, slope)):
// as a run-time test per element the uniform branch splits every element of an unrolled loop into its own basic block
template <bool SIG>
__device__ __forceinline__ float act_grad_t(float pre, float ns)
{
    if constexpr (SIG) {
        const float s = lpd_sigmoid(pre);
        return s * (1.0f - s);
    } else {
        return pre > 0.0f ? 1.0f : ns;
    }
}

// The last step of the column-reduction kernels: thread (o, rg) holds V partial sums for each of two output arrays (channels o V + e), the
// block adds its RG row groups and sends one fp64 atomic per channel and array.  Laid out so that consecutive LANES own consecutive
// CHANNELS: an atomic instruction of a wave then touches 4 cache lines -- issued by the thread that holds the partials (lanes V doubles
// apart) it touched 32 (V = 4) or 64 (V = 8) lines with one or two live lanes each, 2048 single-lane line operations per block.
template <int V>
__device__ __forceinline__ void col_reduce_atomics(double* red /* [2 V][256] */, const double (&a)[V], const double (&b)[V], int O, int C,
                                                   double* __restrict__ out_a, double* __restrict__ out_b)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < V; ++e) { red[e * 256 + tid] = a[e]; red[(V + e) * 256 + tid] = b[e]; }
    __syncthreads();
    const int RG = 256 / O;
    for (int j = tid; j < 2 * C; j += 256) {
        const int half = j >= C, c = half ? j - C : j;
        const int o = c / V, e = c % V;
        const double* src = red + ((half ? V : 0) + e) * 256 + o;
        double v = 0.0;
        for (int g = 0; g < RG; ++g) v += src[g * O];
        atomicAdd((half ? out_b : out_a) + lpd_stat_rofs() + c, v);
    }
}

// ---------------------------------------------------------------------------------------------
// column statistics: sum and sum of squares over R rows.  C/4 <= 256 and 256 % (C/4) == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ X, long long ld, long long R, int C,
                                                       double* __restrict__ sum, double* __restrict__ sumsq)
{
    __shared__ double red[8 * 256];
    const int Q = C >> 2;
    const int RG = 256 / Q;
    const int q = threadIdx.x % Q, rg = threadIdx.x / Q;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    // four rows per trip (eight loads in flight); every element enters the fp64 sums directly -- var = E[x^2] - mean^2 cancels, and a
    // BatchNorm over the 6 rows of a T-Net fc layer with |mean| >> std turned fp32 partial sums (1e-7 of mean^2) into 3e-3 of the output
    const long long S = (long long)gridDim.x * RG;
    for (long long r = (long long)blockIdx.x * RG + rg; r < R; r += 4 * S) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = r + u * S < R ? *reinterpret_cast<const float4*>(X + (r + u * S) * ld + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
            ss[0] += (double)v[u].x * v[u].x; ss[1] += (double)v[u].y * v[u].y; ss[2] += (double)v[u].z * v[u].z; ss[3] += (double)v[u].w * v[u].w;
        }
    }
    col_reduce_atomics<4>(red, s, ss, Q, C, sum, sumsq);
}

// mean / biased var -> scale, shift, mean, invstd; running-stat update (momentum, unbiased var)
__global__ void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq, double count, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sum[c] / count;
    double var = sumsq[c] / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)gamma[c] * is);
    scale[c] = sc;
    shift[c] = (float)((double)beta[c] - m * (double)gamma[c] * is);
    mean[c] = (float)m;
    invstd[c] = (float)is;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// Y = act(scale * X + shift), float4 per thread.  In-place allowed.
// The grid stride (gridDim * 256 threads) is a multiple of the C/4 column quads whenever C/4 divides 256 (all layer
// widths here), so a thread keeps ONE column quad for its whole loop and the per-column constants are loaded once.
// Y16 (or null): a bf16 copy of the result rows ([R][ld16] bf16 elements) next to the fp32 ones.
__global__ void affine_act_kernel(const float* __restrict__ X, long long ldx, float* __restrict__ Y, long long ldy,
                                  long long R, int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                  int act, float slope, uint16_t* __restrict__ Y16 = nullptr, long long ld16 = 0)
{
    const int Q = C >> 2;
    const long long total = R * Q;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool fixed_q = (stride % Q) == 0;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(e % Q);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) { sc = *reinterpret_cast<const float4*>(scale + q * 4); sh = *reinterpret_cast<const float4*>(shift + q * 4); }
    for (; e < total; e += stride) {
        const long long r = e / Q;
        if (!fixed_q) {
            q = (int)(e - r * Q);
            if (scale) { sc = *reinterpret_cast<const float4*>(scale + q * 4); sh = *reinterpret_cast<const float4*>(shift + q * 4); }
        }
        const float4 x = *reinterpret_cast<const float4*>(X + r * ldx + q * 4);
        float4 y;
        y.x = sc.x * x.x + sh.x; y.y = sc.y * x.y + sh.y; y.z = sc.z * x.z + sh.z; y.w = sc.w * x.w + sh.w;
        if (act == 3) {      // uniform: a scalar branch, the exp/divide path is not if-converted into the common case
            y.x = lpd_sigmoid(y.x); y.y = lpd_sigmoid(y.y); y.z = lpd_sigmoid(y.z); y.w = lpd_sigmoid(y.w);
        } else {
            const float ns = lpd_neg_slope(act, slope);
            y.x = lpd_act_pl(y.x, ns); y.y = lpd_act_pl(y.y, ns); y.z = lpd_act_pl(y.z, ns); y.w = lpd_act_pl(y.w, ns);
        }
        if (Y) *reinterpret_cast<float4*>(Y + r * ldy + q * 4) = y;      // uniform (null: the bf16 copy is the only consumer's form)
        if (Y16) {       // uniform
            const unsigned w0 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)y.x) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)y.y) << 16);
            const unsigned w1 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)y.z) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)y.w) << 16);
            *reinterpret_cast<uint2*>(Y16 + r * ld16 + q * 4) = make_uint2(w0, w1);
        }
    }
}

// backward reductions of  Y = act(scale * X + shift):  dbeta = sum dpre, dgamma = sum dpre * xhat
template <bool SIG>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ dY, long long lddy,
                                                                const float* __restrict__ X, long long ldx, long long R,
                                                                int C, const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int act, float slope,
                                                                double* __restrict__ dbeta, double* __restrict__ dgamma)
{
    const float ns = lpd_neg_slope(act, slope);
    __shared__ double red[8 * 256];
    const int Q = C >> 2;
    const int RG = 256 / Q;
    const int q = threadIdx.x % Q, rg = threadIdx.x / Q;
    float sc[4] = {1, 1, 1, 1}, sh[4] = {0, 0, 0, 0}, mu[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (scale) { sc[e] = scale[q * 4 + e]; sh[e] = shift[q * 4 + e]; }
        if (mean) { mu[e] = mean[q * 4 + e]; is[e] = invstd[q * 4 + e]; }
    }
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    // four rows per trip: eight 16-byte loads in flight, the four rows summed in fp32 and the trips in fp64 (with an fp64 add and an fp64
    // fma per element the kernel was bound by them: 378 us for the 1.5 GB of the conv3 map)
    const long long S = (long long)gridDim.x * RG;
    for (long long r = (long long)blockIdx.x * RG + rg; r < R; r += 4 * S) {
        float4 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ru = r + u * S < R ? r + u * S : R - 1;       // past the end: a valid row, its gradient zeroed below
            xv[u] = *reinterpret_cast<const float4*>(X + ru * ldx + q * 4);
            gv[u] = *reinterpret_cast<const float4*>(dY + ru * lddy + q * 4);
        }
        float pb[4] = {0, 0, 0, 0}, pg[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float live = r + u * S < R ? 1.0f : 0.0f;
            const float x[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            const float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dpre = live * g[e] * act_grad_t<SIG>(sc[e] * x[e] + sh[e], ns);
                pb[e] += dpre;
                pg[e] += dpre * ((x[e] - mu[e]) * is[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sb[e] += pb[e]; sg[e] += pg[e]; }
    }
    col_reduce_atomics<4>(red, sb, sg, Q, C, dbeta, dgamma);
}

// dX = scale * (dpre - dbeta/R - xhat * dgamma/R)   (has_bn)   or   dX = dpre   (no BN).  In-place on dY allowed.
// Per-column constants (incl. the two fp64 means) are formed once per thread: see affine_act_kernel.
template <bool SIG>
__global__ void bn_act_bwd_apply_kernel(const float* __restrict__ dY, long long lddy, const float* __restrict__ X,
                                        long long ldx, float* __restrict__ dX, long long lddx, long long R, int C,
                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                        const double* __restrict__ dbeta, const double* __restrict__ dgamma,
                                        double count, int act, float slope, int has_bn)
{
    const float ns = lpd_neg_slope(act, slope);
    const int Q = C >> 2;
    const long long total = R * Q;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool fixed_q = (stride % Q) == 0;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(e % Q);
    float sc[4], sh[4], mu[4], is[4], mb[4], mg[4];
    auto load_consts = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = q * 4 + c;
            sc[c] = scale ? scale[ch] : 1.0f;
            sh[c] = scale ? shift[ch] : 0.0f;
            mu[c] = has_bn ? mean[ch] : 0.0f;
            is[c] = has_bn ? invstd[ch] : 1.0f;
            mb[c] = has_bn ? (float)(dbeta[ch] / count) : 0.0f;
            mg[c] = has_bn ? (float)(dgamma[ch] / count) : 0.0f;
        }
    };
    load_consts();
    auto one = [&](const float4& xv, const float4& gv, long long r) {
        const float x[4] = {xv.x, xv.y, xv.z, xv.w};
        const float g[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float dpre = g[c] * act_grad_t<SIG>(sc[c] * x[c] + sh[c], ns);
            o[c] = has_bn ? sc[c] * (dpre - mb[c] - (x[c] - mu[c]) * is[c] * mg[c]) : dpre;
        }
        *reinterpret_cast<float4*>(dX + r * lddx + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    };
    if (fixed_q) {      // four rows per trip (eight 16-byte loads open per thread), launched at <= 768 blocks: see lpd_bn_act_bwd_bf16
        for (; e < total; e += 4 * stride) {
            float4 xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = (e + u * stride < total ? e + u * stride : e) / Q;
                xv[u] = *reinterpret_cast<const float4*>(X + r * ldx + q * 4);
                gv[u] = *reinterpret_cast<const float4*>(dY + r * lddy + q * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u * stride < total) one(xv[u], gv[u], (e + u * stride) / Q);
        }
        return;
    }
    for (; e < total; e += stride) {
        const long long r = e / Q;
        q = (int)(e - r * Q);
        load_consts();
        one(*reinterpret_cast<const float4*>(X + r * ldx + q * 4), *reinterpret_cast<const float4*>(dY + r * lddy + q * 4), r);
    }
}

// The same two passes on bf16 tensors (the bf16-storage training mode keeps the conv3 map -- y3, its gradient -- in bf16): eight channels
// (16 bytes) per thread; the arithmetic is the fp32 kernels' on the widened values, the result is rounded once on the way out.
__device__ __forceinline__ void bf16x8_widen(const uint4& w, float (&v)[8])
{
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) { v[2 * p] = __uint_as_float(u[p] << 16); v[2 * p + 1] = __uint_as_float(u[p] & 0xffff0000u); }
}

// Four rows per thread and trip (eight 16-byte loads in flight): at 101 registers / 32 KiB of LDS the kernel holds half the waves of the fp32
// one; with one row per trip and an fp64 add + fma per element it ran at 1.55 TB/s (bound by the fp64 operations).  sum dpre xhat is formed as invstd (sum dpre x - mean sum dpre) from fp64 sums.
template <bool SIG>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce16_kernel(const uint16_t* __restrict__ dY, long long lddy, const uint16_t* __restrict__ X,
                                                                  long long ldx, long long R, int C, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, int act, float slope,
                                                                  double* __restrict__ dbeta, double* __restrict__ dgamma)
{
    const float ns = lpd_neg_slope(act, slope);
    __shared__ double red[16 * 256];
    const int O = C >> 3;
    const int RG = 256 / O;
    const int o = threadIdx.x % O, rg = threadIdx.x / O;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale ? scale[o * 8 + e] : 1.0f; sh[e] = scale ? shift[o * 8 + e] : 0.0f; }
    double sb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long S = (long long)gridDim.x * RG;
    for (long long r = (long long)blockIdx.x * RG + rg; r < R; r += 4 * S) {
        uint4 xw[4], gw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ru = r + u * S < R ? r + u * S : R - 1;       // past the end: a valid row, its gradient zeroed below
            xw[u] = *reinterpret_cast<const uint4*>(X + ru * ldx + o * 8);
            gw[u] = *reinterpret_cast<const uint4*>(dY + ru * lddy + o * 8);
        }
        // the four rows are summed in fp32, the trips in fp64 (one fp64 add per four elements: the kernel was bound by them)
        float pb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x[8], g[8];
            bf16x8_widen(xw[u], x);
            bf16x8_widen(gw[u], g);
            const float live = r + u * S < R ? 1.0f : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dpre = live * g[e] * act_grad_t<SIG>(sc[e] * x[e] + sh[e], ns);
                pb[e] += dpre;
                pg[e] += dpre * x[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { sb[e] += pb[e]; sg[e] += pg[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float mu = mean ? mean[o * 8 + e] : 0.0f, is = mean ? invstd[o * 8 + e] : 1.0f;
        sg[e] = (sg[e] - (double)mu * sb[e]) * (double)is;
    }
    col_reduce_atomics<8>(red, sb, sg, O, C, dbeta, dgamma);
}

template <bool SIG>
__global__ void bn_act_bwd_apply16_kernel(const uint16_t* __restrict__ dY, long long lddy, const uint16_t* __restrict__ X, long long ldx,
                                          uint16_t* __restrict__ dX, long long lddx, long long R, int C, const float* __restrict__ scale,
                                          const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
                                          const double* __restrict__ dbeta, const double* __restrict__ dgamma, double count, int act, float slope,
                                          int has_bn)
{
    const float ns = lpd_neg_slope(act, slope);
    const int O = C >> 3;
    const long long total = R * O;
    const long long stride = (long long)gridDim.x * blockDim.x;      // (C / 8 divides 256: a thread keeps its channels)
    const long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int o = (int)(e0 % O);
    // dX = a dpre + b x + c per channel:  a = scale, b = -scale invstd mg, c = -scale (mb - mean invstd mg)   (no BN: a = 1, b = c = 0)
    float sc[8], sh[8], ka[8], kb[8], kc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = o * 8 + c;
        sc[c] = scale ? scale[ch] : 1.0f;
        sh[c] = scale ? shift[ch] : 0.0f;
        const float mu = has_bn ? mean[ch] : 0.0f, is = has_bn ? invstd[ch] : 1.0f;
        const float mb = has_bn ? (float)(dbeta[ch] / count) : 0.0f, mg = has_bn ? (float)(dgamma[ch] / count) : 0.0f;
        ka[c] = has_bn ? sc[c] : 1.0f;
        kb[c] = has_bn ? -sc[c] * is * mg : 0.0f;
        kc[c] = has_bn ? -sc[c] * (mb - mu * is * mg) : 0.0f;
    }
    for (long long e = e0; e < total; e += 4 * stride) {
        uint4 xw[4], gw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long eu = e + u * stride < total ? e + u * stride : e;
            const long long r = eu / O;
            xw[u] = *reinterpret_cast<const uint4*>(X + r * ldx + o * 8);
            gw[u] = *reinterpret_cast<const uint4*>(dY + r * lddy + o * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e + u * stride >= total) break;
            float x[8], g[8];
            bf16x8_widen(xw[u], x);
            bf16x8_widen(gw[u], g);
            unsigned w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = 2 * p + q;
                    const float dpre = g[c] * act_grad_t<SIG>(sc[c] * x[c] + sh[c], ns);
                    v[q] = ka[c] * dpre + kb[c] * x[c] + kc[c];
                }
                w[p] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16);
            }
            *reinterpret_cast<uint4*>(dX + ((e + u * stride) / O) * lddx + o * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// edge tensors: rows (i, t) = i*k + t
// ---------------------------------------------------------------------------------------------
// U[(i,t)] = P[cloud(i)*N + idx[i][t]] + Q[i]      (lpdnet_model.py:350-357 gather + cat, in split form)
// Optionally accumulates the BatchNorm statistics of U (column sums and sums of squares, fp64) while the rows are in
// registers: the separate lpd_colstats pass over the 1.85 / 3.7 GB edge tensor disappears.
template <int LPP>
__global__ __launch_bounds__(256) void edge_build_kernel(const float* __restrict__ P, long long ldp,
                                                         const float* __restrict__ Q, long long ldq,
                                                         const int32_t* __restrict__ idx, float* __restrict__ U,
                                                         long long M, int N, int k, double* __restrict__ sum,
                                                         double* __restrict__ sumsq)
{
    constexpr int PPW = 64 / LPP;
    constexpr int C = LPP * 4;
    __shared__ double red[256][8];
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPP, cl = lane % LPP;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (long long w = wave; w * PPW < M; w += nw) {
        const long long m = w * PPW + sub;
        const bool ok = m < M;
        const long long mm = ok ? m : M - 1;
        const long long base = (mm / N) * N;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (Q) q = *reinterpret_cast<const float4*>(Q + mm * ldq + cl * 4);
        for (int t0 = 0; t0 < k; t0 += LPP) {
            // the point's next LPP neighbour indices: one coalesced load, broadcast by shuffle
            const int my_idx = (t0 + cl < k) ? idx[mm * k + t0 + cl] : 0;
            const int tn = min(k - t0, LPP);
#pragma unroll 5
            for (int t = 0; t < tn; ++t) {
                const int j = __shfl(my_idx, sub * LPP + t, 64);
                float4 p = *reinterpret_cast<const float4*>(P + (base + j) * ldp + cl * 4);
                p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w;
                if (ok) {
                    *reinterpret_cast<float4*>(U + (mm * k + t0 + t) * C + cl * 4) = p;
                    if (sum) {
                        s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
                        ss[0] += (double)p.x * p.x; ss[1] += (double)p.y * p.y; ss[2] += (double)p.z * p.z; ss[3] += (double)p.w * p.w;
                    }
                }
            }
        }
    }
    if (sum) {   // block reduction over the 256 / LPP lane groups that share a column quad, then one fp64 atomic per column
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][4 + e] = ss[e]; }
        __syncthreads();
        if (threadIdx.x < LPP) {
            for (int g = 1; g < 256 / LPP; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[g * LPP + threadIdx.x][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&sum[lpd_stat_rofs() + threadIdx.x * 4 + e], red[threadIdx.x][e]);
                atomicAdd(&sumsq[lpd_stat_rofs() + threadIdx.x * 4 + e], red[threadIdx.x][4 + e]);
            }
        }
    }
}

// out[i][c] = act(scale[c] * sel_t X[(i,t)][c] + shift[c]), arg[i][c] = the selected t
__global__ void group_max_kernel(const float* __restrict__ X, long long ldx, int k, const float* __restrict__ scale,
                                 const float* __restrict__ shift, int act, float slope, float* __restrict__ out,
                                 long long ldo, uint8_t* __restrict__ arg, float* __restrict__ xsel, long long ldsel, long long M,
                                 int C)
{
    const int Q = C >> 2;
    const long long total = M * Q;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / Q;
        const int q = (int)(e - i * Q);
        float mx[4], mn[4];
        int amx[4] = {0, 0, 0, 0}, amn[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) { mx[c] = -INFINITY; mn[c] = INFINITY; }
        for (int t = 0; t < k; ++t) {
            const float4 v4 = *reinterpret_cast<const float4*>(X + (i * k + t) * ldx + q * 4);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (v[c] > mx[c]) { mx[c] = v[c]; amx[c] = t; }   // first arg-max, like torch.max
                if (v[c] < mn[c]) { mn[c] = v[c]; amn[c] = t; }
            }
        }
        float o[4], xs[4];
        uint8_t a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = q * 4 + c;
            const float sc = scale[ch], sh = shift[ch];
            const bool usemax = sc >= 0.0f;
            xs[c] = usemax ? mx[c] : mn[c];
            o[c] = lpd_act(sc * xs[c] + sh, act, slope);
            a[c] = (uint8_t)(usemax ? amx[c] : amn[c]);
        }
        *reinterpret_cast<float4*>(out + i * ldo + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uchar4*>(arg + i * C + q * 4) = make_uchar4(a[0], a[1], a[2], a[3]);
        if (xsel) *reinterpret_cast<float4*>(xsel + i * ldsel + q * 4) = make_float4(xs[0], xs[1], xs[2], xs[3]);
    }
}

// dX[(i,t)][c] (+)= (arg[i][c] == t) ? dOut[i][c] : 0      accumulate = 0 writes every row (zero fill included)
__global__ void group_max_bwd_kernel(const float* __restrict__ dOut, long long ldo, const uint8_t* __restrict__ arg, int k,
                                     float* __restrict__ dX, long long M, int C, int accumulate)
{
    const int Q = C >> 2;
    const long long total = M * Q;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / Q;
        const int q = (int)(e - i * Q);
        const float4 g4 = *reinterpret_cast<const float4*>(dOut + i * ldo + q * 4);
        const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + i * C + q * 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const int a[4] = {a4.x, a4.y, a4.z, a4.w};
        if (accumulate) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dX[(i * k + a[c]) * C + q * 4 + c] += g[c];
        } else {
            for (int t = 0; t < k; ++t) {
                float4 v;
                v.x = a[0] == t ? g[0] : 0.f; v.y = a[1] == t ? g[1] : 0.f;
                v.z = a[2] == t ? g[2] : 0.f; v.w = a[3] == t ? g[3] : 0.f;
                *reinterpret_cast<float4*>(dX + (i * k + t) * C + q * 4) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused backward of  out[i] = max_t act(BN(X[(i,t)]))  (+ an optional dense gradient on the post-activation
// edge values): replaces group_max_bwd -> bn_act_bwd(reduce, apply) -> group_sum on the materialised edge
// tensors.  The arg-max gradient is generated on the fly from (dOut, arg), so the zero-filled [E,C] gradient is
// never written or re-read; the centre-term gradient dQ[i] = sum_t dX[(i,t)] falls out of the apply pass.
// One thread per (point, float4 of channels); column quad fixed per thread (C/4 divides 256).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_bn_bwd_reduce_kernel(const float* __restrict__ dOut, long long ldo,
                                                                 const uint8_t* __restrict__ arg,
                                                                 const float* __restrict__ dDense,   // [E][C] or null
                                                                 const float* __restrict__ X,
                                                                 const float* __restrict__ Xsel,     // [M][ldsel] X at arg, or null
                                                                 long long ldsel, int k, long long M, int C,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 int act, float slope, float inv_ns, double* __restrict__ dbeta,
                                                                 double* __restrict__ dgamma)
{
    __shared__ double red[256][8];
    const int Q = C >> 2;
    const int RG = 256 / Q;
    const int q = threadIdx.x % Q, rg = threadIdx.x / Q;
    float sc[4], sh[4], mu[4], is[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sc[c] = scale[q * 4 + c]; sh[c] = shift[q * 4 + c]; mu[c] = mean[q * 4 + c]; is[c] = invstd[q * 4 + c];
    }
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        const float4 g4 = *reinterpret_cast<const float4*>(dOut + i * ldo + q * 4);
        const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + i * C + q * 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const int a[4] = {a4.x, a4.y, a4.z, a4.w};
        if (dDense) {
            for (int t = 0; t < k; ++t) {
                const float4 xv = *reinterpret_cast<const float4*>(X + (i * k + t) * C + q * 4);
                const float4 dv = *reinterpret_cast<const float4*>(dDense + (i * k + t) * C + q * 4);
                const float x[4] = {xv.x, xv.y, xv.z, xv.w};
                const float d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gy = d[c] + (a[c] == t ? g[c] : 0.0f);
                    // inv_ns > 0: X holds the POST-activation value y = act(pre) (LeakyReLU / identity, negative slope ns = 1 / inv_ns):
                    // pre = y or y / ns, xhat = (pre - beta) / gamma with `mean` = beta and `invstd` = 1 / gamma passed by the host
                    const float pre = inv_ns > 0.0f ? (x[c] > 0.0f ? x[c] : x[c] * inv_ns) : sc[c] * x[c] + sh[c];
                    const float xh = ((inv_ns > 0.0f ? pre : x[c]) - mu[c]) * is[c];
                    const float dpre = gy * act_grad(pre, act, slope);
                    sb[c] += dpre;
                    sg[c] += (double)dpre * xh;
                }
            }
        } else {
            float xs[4] = {0.f, 0.f, 0.f, 0.f};
            if (Xsel) {                      // the selected values kept by the forward: no gather from the edge tensor
                const float4 x4 = *reinterpret_cast<const float4*>(Xsel + i * ldsel + q * 4);
                xs[0] = x4.x; xs[1] = x4.y; xs[2] = x4.z; xs[3] = x4.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {   // only the arg-max edge carries gradient
                const float x = Xsel ? xs[c] : X[(i * k + a[c]) * C + q * 4 + c];
                const float dpre = g[c] * act_grad(sc[c] * x + sh[c], act, slope);
                sb[c] += dpre;
                sg[c] += (double)dpre * ((x - mu[c]) * is[c]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = sb[e]; red[threadIdx.x][4 + e] = sg[e]; }
    __syncthreads();
    if (rg == 0) {
        for (int g2 = 1; g2 < RG; ++g2)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[q][e] += red[g2 * Q + q][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&dbeta[lpd_stat_rofs() + q * 4 + e], red[q][e]);
            atomicAdd(&dgamma[lpd_stat_rofs() + q * 4 + e], red[q][4 + e]);
        }
    }
}

__global__ __launch_bounds__(256) void edge_bn_bwd_apply_kernel(const float* __restrict__ dOut, long long ldo,
                                                                const uint8_t* __restrict__ arg,
                                                                const float* __restrict__ dDense, const float* __restrict__ X,
                                                                float* __restrict__ dX, float* __restrict__ dQ, long long ldq,
                                                                int k, long long M, int C, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const double* __restrict__ dbeta,
                                                                const double* __restrict__ dgamma, double count, int act,
                                                                float slope, float inv_ns)
{
    const int Q = C >> 2;
    const int RG = 256 / Q;
    const int q = threadIdx.x % Q, rg = threadIdx.x / Q;
    float sc[4], sh[4], mu[4], is[4], mb[4], mg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = q * 4 + c;
        sc[c] = scale[ch]; sh[c] = shift[ch]; mu[c] = mean[ch]; is[c] = invstd[ch];
        mb[c] = (float)(dbeta[ch] / count);
        mg[c] = (float)(dgamma[ch] / count);
    }
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        const float4 g4 = *reinterpret_cast<const float4*>(dOut + i * ldo + q * 4);
        const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + i * C + q * 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const int a[4] = {a4.x, a4.y, a4.z, a4.w};
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < k; ++t) {
            const long long off = (i * k + t) * C + q * 4;
            const float4 xv = *reinterpret_cast<const float4*>(X + off);
            float d[4] = {0.f, 0.f, 0.f, 0.f};
            if (dDense) {
                const float4 dv = *reinterpret_cast<const float4*>(dDense + off);
                d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
            }
            const float x[4] = {xv.x, xv.y, xv.z, xv.w};
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gy = d[c] + (a[c] == t ? g[c] : 0.0f);
                const float pre = inv_ns > 0.0f ? (x[c] > 0.0f ? x[c] : x[c] * inv_ns) : sc[c] * x[c] + sh[c];      // see the reduce kernel
                const float xh = ((inv_ns > 0.0f ? pre : x[c]) - mu[c]) * is[c];
                const float dpre = gy * act_grad(pre, act, slope);
                o[c] = sc[c] * (dpre - mb[c] - xh * mg[c]);
                sum[c] += o[c];
            }
            *reinterpret_cast<float4*>(dX + off) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (dQ) *reinterpret_cast<float4*>(dQ + i * ldq + q * 4) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    }
}

// dQ[i] = sum_t dU[(i,t)]
__global__ void group_sum_kernel(const float* __restrict__ dU, int k, float* __restrict__ dQ, long long ldq, long long M, int C)
{
    const int Q = C >> 2;
    const long long total = M * Q;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / Q;
        const int q = (int)(e - i * Q);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < k; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(dU + (i * k + t) * C + q * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dQ + i * ldq + q * 4) = s;
    }
}

// dP[cloud(i)*N + idx[i][t]] += dU[(i,t)]   (transpose of the gather; float atomics, 256 contiguous bytes per wave op)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ dU, const int32_t* __restrict__ idx,
                                                               float* __restrict__ dP, long long ldp, long long M, int N,
                                                               int k, int C)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long i = wave; i < M; i += nw) {
        const long long base = (i / N) * N;
        for (int t = 0; t < k; ++t) {
            const int j = idx[i * k + t];
            const float* src = dU + (i * k + t) * C;
            float* dst = dP + (base + j) * ldp;
            for (int c = lane; c < C; c += 64) atomicAdd(dst + c, src[c]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Transposed kNN graph (CSR): for every point j the list of edges (i,t) with nbr(i,t) = j.  Built once per graph and
// step (three small kernels, int atomics only); the backward of the neighbour gather is then a GATHER-sum per row
// (each dU row read once, 512 B - 1 KiB contiguous) instead of 3.6 M x C float atomics (1.3 TB/s ceiling:
// MI355X_MICROARCH.md "Global float atomics").
// ---------------------------------------------------------------------------------------------
__global__ void csr_count_kernel(const int32_t* __restrict__ idx, int32_t* __restrict__ deg, long long E, int N, int k)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long long i = e / k;
    atomicAdd(deg + (i / N) * N + idx[e], 1);
}

// exclusive scan of the in-degrees of one cloud (block per cloud); rowptr[M] entries + rowptr[M] = E written by cloud B-1
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int32_t* __restrict__ deg, int32_t* __restrict__ rowptr,
                                                        int32_t* __restrict__ cursor, int N, int k, long long M)
{
    __shared__ int part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int lo = min(tid * per, N), hi = min(lo + per, N);
    const int32_t* d = deg + (long long)b * N;
    int sum = 0;
    for (int j = lo; j < hi; ++j) sum += d[j];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = (tid ? part[tid - 1] : 0) + b * N * k;   // every cloud owns exactly N*k edges
    for (int j = lo; j < hi; ++j) {
        rowptr[(long long)b * N + j] = run;
        cursor[(long long)b * N + j] = run;
        run += d[j];
    }
    if (b == gridDim.x - 1 && tid == 0) rowptr[M] = (int)(M * k);
}

__global__ void csr_fill_kernel(const int32_t* __restrict__ idx, int32_t* __restrict__ cursor, int32_t* __restrict__ edges,
                                long long E, int N, int k)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long long i = e / k;
    const int slot = atomicAdd(cursor + (i / N) * N + idx[e], 1);
    edges[slot] = (int32_t)e;
}

// The three kernels above in one, with LDS atomics: 4 blocks per cloud, each owning a quarter of the destination rows (it
// reads all of the cloud's N k indices twice -- L2 -- and counts those below its range for its base offset).  The global-atomic
// version took 175 us per graph at B = 44 (59 us count + 101 us fill with returning atomics on 180 k addresses).
__global__ __launch_bounds__(1024) void csr_build_kernel(const int32_t* __restrict__ idx, int32_t* __restrict__ rowptr,
                                                         int32_t* __restrict__ edges, int N, int k, long long M, int G, int nbmax)
{
    extern __shared__ int csr_sh[];          // bins [nbmax], part [1024]
    __shared__ int s_below;
    int* bins = csr_sh;
    int* part = csr_sh + nbmax;
    const int b = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x;
    const int j0 = (int)((long long)N * g / G), j1 = (int)((long long)N * (g + 1) / G), nb = j1 - j0;
    for (int j = tid; j < nb; j += 1024) bins[j] = 0;
    if (tid == 0) s_below = 0;
    __syncthreads();
    const int Ec = N * k;
    const int32_t* id = idx + (long long)b * Ec;
    int below = 0;
    int e = tid;
    for (; e + 7 * 1024 < Ec; e += 8 * 1024) {      // eight independent index loads in flight
        int j[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = id[e + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j[u] < j0) ++below;
            else if (j[u] < j1) atomicAdd(&bins[j[u] - j0], 1);
        }
    }
    for (; e < Ec; e += 1024) {
        const int j = id[e];
        if (j < j0) ++below;
        else if (j < j1) atomicAdd(&bins[j - j0], 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o, 64);
    if ((tid & 63) == 0) atomicAdd(&s_below, below);
    __syncthreads();
    const int per = (nb + 1023) / 1024;
    const int lo = min(tid * per, nb), hi = min(lo + per, nb);
    int sum = 0;
    for (int j = lo; j < hi; ++j) sum += bins[j];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = (tid ? part[tid - 1] : 0) + s_below + b * Ec;   // every cloud owns exactly N*k edges
    for (int j = lo; j < hi; ++j) {
        const int c = bins[j];
        rowptr[(long long)b * N + j0 + j] = run;
        bins[j] = run;                           // the row's fill cursor
        run += c;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) rowptr[M] = (int)(M * k);
    __syncthreads();
    e = tid;
    for (; e + 7 * 1024 < Ec; e += 8 * 1024) {
        int j[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) j[u] = id[e + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j[u] >= j0 && j[u] < j1) edges[atomicAdd(&bins[j[u] - j0], 1)] = b * Ec + e + u * 1024;
    }
    for (; e < Ec; e += 1024) {
        const int j = id[e];
        if (j >= j0 && j < j1) edges[atomicAdd(&bins[j - j0], 1)] = b * Ec + e;
    }
}

// dP[j] (+)= sum over the incoming edges e of dU[e]; one wave per row, float4 per lane, rows of C = 64 .. 256 floats
template <int LPR>   // lanes per row = C / 4
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const float* __restrict__ dU, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ edges, float* __restrict__ dP,
                                                              long long ldp, long long M, int accumulate)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, cl = lane % LPR;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long r0 = wave * RPW; r0 < M; r0 += nw * RPW) {
        const long long j = r0 + sub;
        if (j >= M) continue;
        const int beg = rowptr[j], end = rowptr[j + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int p = beg;
        for (; p + 3 < end; p += 4) {   // four independent row loads in flight
            const int e0 = edges[p], e1 = edges[p + 1], e2 = edges[p + 2], e3 = edges[p + 3];
            const float4 a = *reinterpret_cast<const float4*>(dU + (long long)e0 * (LPR * 4) + cl * 4);
            const float4 b = *reinterpret_cast<const float4*>(dU + (long long)e1 * (LPR * 4) + cl * 4);
            const float4 c = *reinterpret_cast<const float4*>(dU + (long long)e2 * (LPR * 4) + cl * 4);
            const float4 d = *reinterpret_cast<const float4*>(dU + (long long)e3 * (LPR * 4) + cl * 4);
            acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
            acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; p < end; ++p) {
            const float4 a = *reinterpret_cast<const float4*>(dU + (long long)edges[p] * (LPR * 4) + cl * 4);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
        float4* dst = reinterpret_cast<float4*>(dP + j * ldp + cl * 4);
        if (accumulate) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *dst = acc;
    }
}

// dW[o][c] = sum_m dY[m][o] * X[m][c], Kin <= 8 (first layer, lpdnet_model.py:185).  dW zeroed by the caller.
__global__ __launch_bounds__(256) void dw_smallk_kernel(const float* __restrict__ dY, long long lddy,
                                                        const float* __restrict__ X, long long ldx, long long M, int Co,
                                                        int Kin, float* __restrict__ dW)
{
    // thread = (o, row group); Co <= 256 and 256 % Co == 0
    __shared__ float red[256][8];
    const int RG = 256 / Co;
    const int o = threadIdx.x % Co, rg = threadIdx.x / Co;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long step = (long long)gridDim.x * RG;
    long long m = (long long)blockIdx.x * RG + rg;
    for (; m + 3 * step < M; m += 4 * step) {      // four rows in flight (one row at a time: 107 us for 46 MB at the training step's shape)
        float g[4], x[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g[u] = dY[(m + u * step) * lddy + o];
#pragma unroll
            for (int c = 0; c < 8; ++c) x[u][c] = c < Kin ? X[(m + u * step) * ldx + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += g[u] * x[u][c];
    }
    for (; m < M; m += step) {
        const float g = dY[m * lddy + o];
        for (int c = 0; c < Kin; ++c) acc[c] += g * X[m * ldx + c];
    }
    for (int c = 0; c < 8; ++c) red[threadIdx.x][c] = acc[c];
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < RG; ++g)
            for (int c = 0; c < Kin; ++c) red[o][c] += red[g * Co + o][c];
        for (int c = 0; c < Kin; ++c) atomicAdd(&dW[o * Kin + c], red[o][c]);
    }
}

// ---------------------------------------------------------------------------------------------
// NetVLAD backward pieces
// ---------------------------------------------------------------------------------------------
// dS = A * (g - sum_c A*g),  g = dA + dasum[cloud]    (softmax backward, PointNetVlad.py:58; a_sum path :61)
__global__ void softmax_bwd_kernel(const float* __restrict__ A, const float* __restrict__ dA, const float* __restrict__ dasum,
                                   float* __restrict__ dS, long long rows, int ncols, int rows_per_cloud)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float a = 0.f, g = 0.f;
    if (lane < ncols) {
        a = A[row * ncols + lane];
        g = dA[row * ncols + lane];
        if (dasum) g += dasum[(row / rows_per_cloud) * ncols + lane];
    }
    float dot = a * g;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (lane < ncols) dS[row * ncols + lane] = a * (g - dot);
}

__device__ __forceinline__ float block_sum_256t(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// backward of lpd_vlad_finalize.  v = final normalised vlad [B][F*KC]; inv_c [B][KC], inv_g [B], asum [B][KC] saved
// by the forward.  Outputs dVraw [B][F][KC], dasum [B][KC], dcw2 [F][KC] (atomic accumulation over clouds; zeroed by caller).
template <int KC>
__global__ __launch_bounds__(256) void vlad_finalize_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ v,
                                                                const float* __restrict__ inv_c, const float* __restrict__ inv_g,
                                                                const float* __restrict__ asum, const float* __restrict__ cw2,
                                                                float* __restrict__ dVraw, float* __restrict__ dasum,
                                                                float* __restrict__ dcw2, int F)
{
    __shared__ float s_part[256];
    __shared__ float s_col[KC];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    constexpr int RPB = 256 / KC;
    const int c = tid % KC, rg = tid / KC;
    const float* dv = dOut + (size_t)b * F * KC;
    const float* vv = v + (size_t)b * F * KC;
    float* dr = dVraw + (size_t)b * F * KC;
    const float ig = inv_g[b], ic = inv_c[b * KC + c], as = asum[b * KC + c];
    // <dv, v> over the whole descriptor
    float p = 0.f;
    for (int f = rg; f < F; f += RPB) p += dv[(size_t)f * KC + c] * vv[(size_t)f * KC + c];
    const float dvv = block_sum_256t(p, red);
    // du = ig * (dv - v * <dv,v>);  u = v / ig;  per-column <du, u>
    float pc = 0.f;
    for (int f = rg; f < F; f += RPB) {
        const float vf = vv[(size_t)f * KC + c];
        const float du = ig * (dv[(size_t)f * KC + c] - vf * dvv);
        pc += du * (vf / ig);
    }
    s_part[tid] = pc;
    __syncthreads();
    if (tid < KC) {
        float t = 0.f;
        for (int r = 0; r < RPB; ++r) t += s_part[r * KC + tid];
        s_col[tid] = t;
    }
    __syncthreads();
    const float duu = s_col[c];
    float pa = 0.f;
    for (int f = rg; f < F; f += RPB) {
        const float vf = vv[(size_t)f * KC + c];
        const float u = vf / ig;
        const float du = ig * (dv[(size_t)f * KC + c] - vf * dvv);
        const float d = ic * (du - u * duu);           // d r[f][c]
        dr[(size_t)f * KC + c] = d;
        const float w = cw2[(size_t)f * KC + c];
        pa += d * w;
        atomicAdd(&dcw2[(size_t)f * KC + c], -as * d);
    }
    __syncthreads();
    s_part[tid] = pa;
    __syncthreads();
    if (tid < KC) {
        float t = 0.f;
        for (int r = 0; r < RPB; ++r) t += s_part[r * KC + tid];
        dasum[b * KC + tid] = -t;
    }
}

// The same backward cut into slices of the descriptor rows (grid B x G): the one-block-per-cloud kernel above runs 44 blocks on
// 256 CUs through three dependent passes and adds every element of dcw2 atomically over the clouds (220 us at B = 44).  Here the
// three reductions are three launches over B x G blocks with per-slice partial sums (scratch: the dcw2 buffer, written last),
// and dcw2[f][c] = -sum_b asum[b][c] dVraw[b][f][c] is a plain pass over dVraw.
template <int KC>
__global__ __launch_bounds__(256) void vlad_fbwd_dot_kernel(const float* __restrict__ dOut, const float* __restrict__ v, int F, int G,
                                                            float* __restrict__ part_dvv)
{
    __shared__ float red[4];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int c = tid % KC, rg = tid / KC;
    constexpr int RPB = 256 / KC;
    const int f0 = (int)((long long)F * g / G), f1 = (int)((long long)F * (g + 1) / G);
    const float* dv = dOut + (size_t)b * F * KC;
    const float* vv = v + (size_t)b * F * KC;
    float p = 0.f;
    for (int f = f0 + rg; f < f1; f += RPB) p += dv[(size_t)f * KC + c] * vv[(size_t)f * KC + c];
    const float t = block_sum_256t(p, red);
    if (tid == 0) part_dvv[b * G + g] = t;
}

template <int KC>
__global__ __launch_bounds__(256) void vlad_fbwd_col_kernel(const float* __restrict__ dOut, const float* __restrict__ v,
                                                            const float* __restrict__ inv_g, int F, int G,
                                                            const float* __restrict__ part_dvv, float* __restrict__ part_col)
{
    __shared__ float s_part[256];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int c = tid % KC, rg = tid / KC;
    constexpr int RPB = 256 / KC;
    const int f0 = (int)((long long)F * g / G), f1 = (int)((long long)F * (g + 1) / G);
    const float* dv = dOut + (size_t)b * F * KC;
    const float* vv = v + (size_t)b * F * KC;
    float dvv = 0.f;
    for (int q = 0; q < G; ++q) dvv += part_dvv[b * G + q];
    const float ig = inv_g[b];
    float pc = 0.f;
    for (int f = f0 + rg; f < f1; f += RPB) {
        const float vf = vv[(size_t)f * KC + c];
        const float du = ig * (dv[(size_t)f * KC + c] - vf * dvv);
        pc += du * (vf / ig);
    }
    s_part[tid] = pc;
    __syncthreads();
    if (tid < KC) {
        float t = 0.f;
        for (int r = 0; r < RPB; ++r) t += s_part[r * KC + tid];
        part_col[((size_t)b * G + g) * KC + tid] = t;
    }
}

template <int KC>
__global__ __launch_bounds__(256) void vlad_fbwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ v,
                                                              const float* __restrict__ inv_c, const float* __restrict__ inv_g,
                                                              const float* __restrict__ cw2, int F, int G,
                                                              const float* __restrict__ part_dvv, const float* __restrict__ part_col,
                                                              float* __restrict__ dVraw, float* __restrict__ part_pa)
{
    __shared__ float s_part[256];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int c = tid % KC, rg = tid / KC;
    constexpr int RPB = 256 / KC;
    const int f0 = (int)((long long)F * g / G), f1 = (int)((long long)F * (g + 1) / G);
    const float* dv = dOut + (size_t)b * F * KC;
    const float* vv = v + (size_t)b * F * KC;
    float* dr = dVraw + (size_t)b * F * KC;
    float dvv = 0.f, duu = 0.f;
    for (int q = 0; q < G; ++q) { dvv += part_dvv[b * G + q]; duu += part_col[((size_t)b * G + q) * KC + c]; }
    const float ig = inv_g[b], ic = inv_c[b * KC + c];
    float pa = 0.f;
    for (int f = f0 + rg; f < f1; f += RPB) {
        const float vf = vv[(size_t)f * KC + c];
        const float u = vf / ig;
        const float du = ig * (dv[(size_t)f * KC + c] - vf * dvv);
        const float d = ic * (du - u * duu);           // d r[f][c]
        dr[(size_t)f * KC + c] = d;
        pa += d * cw2[(size_t)f * KC + c];
    }
    s_part[tid] = pa;
    __syncthreads();
    if (tid < KC) {
        float t = 0.f;
        for (int r = 0; r < RPB; ++r) t += s_part[r * KC + tid];
        part_pa[((size_t)b * G + g) * KC + tid] = t;
    }
}

template <int KC>
__global__ void vlad_fbwd_dasum_kernel(const float* __restrict__ part_pa, int B, int G, float* __restrict__ dasum)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * KC) return;
    const int b = e / KC, c = e % KC;
    float t = 0.f;
    for (int q = 0; q < G; ++q) t += part_pa[((size_t)b * G + q) * KC + c];
    dasum[e] = -t;
}

template <int KC>
__global__ void vlad_fbwd_dcw2_kernel(const float* __restrict__ dVraw, const float* __restrict__ asum, int B, int F,
                                      float* __restrict__ dcw2)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // (f, c)
    if (e >= F * KC) return;
    const int c = e % KC;
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += asum[b * KC + c] * dVraw[(size_t)b * F * KC + e];
    dcw2[e] = -t;
}

// ---------------------------------------------------------------------------------------------
// T-Net / STN pieces: per-cloud max over the points with arg-max, its backward, and the gradient of the
// per-cloud k x k alignment matrix for k <= 8 (lpdnet_model.py:229,300; PointNetVlad.py:162,209).
// ---------------------------------------------------------------------------------------------
// in [B][N][ld] -> out [B][C], arg [B][C] (row index inside the cloud, first maximum)
__global__ __launch_bounds__(256) void colmax_arg_kernel(const float* __restrict__ in, long long ld, float* __restrict__ out,
                                                         int32_t* __restrict__ arg, int N, int C)
{
    __shared__ float pv[4][64];
    __shared__ int pi[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const int b = blockIdx.y;
    float m = -INFINITY;
    int am = 0;
    if (c < C) {
        const float* p = in + (size_t)b * N * ld + c;
        for (int n = rg; n < N; n += 4) {
            const float v = p[(size_t)n * ld];
            if (v > m) { m = v; am = n; }
        }
    }
    pv[rg][threadIdx.x & 63] = m;
    pi[rg][threadIdx.x & 63] = am;
    __syncthreads();
    if (rg == 0 && c < C) {
        for (int g = 1; g < 4; ++g) {
            const float v = pv[g][threadIdx.x];
            const int i = pi[g][threadIdx.x];
            if (v > m || (v == m && i < am)) { m = v; am = i; }
        }
        out[(size_t)b * C + c] = m;
        arg[(size_t)b * C + c] = am;
    }
}

// dIn[b*N + arg[b][c]][c] = dOut[b][c]   (dIn zero-filled by the caller)
__global__ void colmax_bwd_kernel(const float* __restrict__ dOut, const int32_t* __restrict__ arg, float* __restrict__ dIn,
                                  long long ld, int B, int N, int C)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * C) return;
    const int b = e / C, c = e - b * C;
    dIn[((size_t)b * N + arg[e]) * ld + c] = dOut[e];
}

// dT[b][i][j] = sum_{m in cloud b} X[m][i] * dY[m][j],  i, j < KD <= 8.  One block per cloud.
__global__ __launch_bounds__(256) void cloud_outer_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ dY,
                                                          long long ldy, float* __restrict__ dT, int N, int KD)
{
    __shared__ float red[4][64];
    const int b = blockIdx.x;
    float acc[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) acc[e] = 0.0f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* x = X + ((size_t)b * N + n) * ldx;
        const float* y = dY + ((size_t)b * N + n) * ldy;
        for (int i = 0; i < KD; ++i)
            for (int j = 0; j < KD; ++j) acc[i * 8 + j] += x[i] * y[j];
    }
    for (int i = 0; i < KD; ++i)
        for (int j = 0; j < KD; ++j) {
            float v = acc[i * 8 + j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i * 8 + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < KD * KD) {
        const int i = threadIdx.x / KD, j = threadIdx.x % KD;
        dT[(size_t)b * KD * KD + threadIdx.x] = red[0][i * 8 + j] + red[1][i * 8 + j] + red[2][i * 8 + j] + red[3][i * 8 + j];
    }
}

inline int grid_for(long long work_items, int per_block)
{
    long long b = (work_items + per_block - 1) / per_block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

inline bool cols_ok(int C) { return C >= 4 && C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0; }

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int lpd_colstats(const float* X, long long ld, long long R, int C, double* sum, double* sumsq, double* stat_ws, void* stream)
{
    LPD_CHECK_ARG(X && sum && sumsq && R > 0, "lpd_colstats: bad arguments");
    LPD_CHECK_ARG(C >= 4 && C % 4 == 0 && ld % 4 == 0, "lpd_colstats: C=%d and ld must be multiples of 4", C);
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_colstats: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    // the kernel takes a column panel of 4*2^n <= 1024 columns; other widths (k*k = 12, 4096 of the T-Nets) go panel by panel
    for (int c0 = 0; c0 < C;) {
        int w = 1024;
        while (w > C - c0) w >>= 1;
        const int RG = 256 / (w / 4);
        hipLaunchKernelGGL(colstats_kernel, dim3(lpd_reduce_grid(grid_for(R, RG * 8))), dim3(256), 0, ST(stream), X + c0, ld, R, w, ws.sum(), ws.sumsq());
        LPD_CHECK_LAUNCH("lpd_colstats");
        if (int rc = lpd_stat_finish(ws, sum + c0, sumsq + c0, w, ST(stream))) return rc;
        c0 += w;
    }
    return LPD_OK;
}

extern "C" int lpd_bn_finalize(const double* sum, const double* sumsq, double count, int C, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* mean, float* invstd, void* stream)
{
    LPD_CHECK_ARG(sum && sumsq && gamma && beta && scale && shift && mean && invstd && C > 0 && count > 0,
                  "lpd_bn_finalize: bad arguments");
    LPD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "lpd_bn_finalize: running stats come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, ST(stream), sum, sumsq, count, C, gamma,
                       beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd);
    LPD_CHECK_LAUNCH("lpd_bn_finalize");
    return LPD_OK;
}

extern "C" int lpd_affine_act(const float* X, long long ldx, float* Y, long long ldy, long long R, int C,
                              const float* scale, const float* shift, int act, float slope, void* stream)
{
    LPD_CHECK_ARG(X && Y && R > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "lpd_affine_act: bad arguments");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_affine_act: scale and shift come in pairs");
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(R * (C / 4), 256 * 4)), dim3(256), 0, ST(stream), X, ldx, Y, ldy, R,
                       C, scale, shift, act, slope);
    LPD_CHECK_LAUNCH("lpd_affine_act");
    return LPD_OK;
}

// lpd_affine_act with a bf16 copy of the result rows beside the fp32 ones (Y16 [R][ld16] bf16 elements, ld16 % 4 == 0): the bf16 storage
// mode's second form of the point features [x1 | x2 | x3] (the conv3 operand and the B operand of its weight gradient).  Y may be null:
// only the bf16 rows are written (x1 and x3, whose fp32 form nothing reads in that mode: 277 MB of stores per step at 44 clouds)
extern "C" int lpd_affine_act2(const float* X, long long ldx, float* Y, long long ldy, void* Y16, long long ld16, long long R, int C,
                               const float* scale, const float* shift, int act, float slope, void* stream)
{
    LPD_CHECK_ARG(X && Y16 && R > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && (!Y || ldy % 4 == 0) && ld16 % 4 == 0 && ((uintptr_t)Y16 & 7) == 0,
                  "lpd_affine_act2: bad arguments");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_affine_act2: scale and shift come in pairs");
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(R * (C / 4), 256 * 4)), dim3(256), 0, ST(stream), X, ldx, Y, ldy, R,
                       C, scale, shift, act, slope, reinterpret_cast<uint16_t*>(Y16), ld16);
    LPD_CHECK_LAUNCH("lpd_affine_act2");
    return LPD_OK;
}

extern "C" int lpd_bn_act_bwd(const float* dY, long long lddy, const float* X, long long ldx, float* dX, long long lddx,
                              long long R, int C, const float* scale, const float* shift, const float* mean,
                              const float* invstd, int act, float slope, int has_bn, double* dbeta, double* dgamma,
                              double* stat_ws, void* stream)
{
    LPD_CHECK_ARG(dY && X && dX && dbeta && dgamma && R > 0, "lpd_bn_act_bwd: bad arguments");
    LPD_CHECK_ARG(cols_ok(C) && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, "lpd_bn_act_bwd: C=%d / leading dims unsupported", C);
    LPD_CHECK_ARG(!has_bn || (scale && shift && mean && invstd), "lpd_bn_act_bwd: BatchNorm form needs scale/shift/mean/invstd");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_bn_act_bwd: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_bn_act_bwd", C);
    const int RG = 256 / (C / 4);
    hipLaunchKernelGGL(act == 3 ? bn_act_bwd_reduce_kernel<true> : bn_act_bwd_reduce_kernel<false>, dim3(lpd_reduce_grid(grid_for(R, RG * 8))), dim3(256), 0, ST(stream),
                       dY, lddy, X, ldx, R, C, scale, shift, has_bn ? mean : nullptr, has_bn ? invstd : nullptr, act, slope, ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_bn_act_bwd(reduce)");
    if (int rc = lpd_stat_finish(ws, dbeta, dgamma, C, ST(stream))) return rc;
    hipLaunchKernelGGL(act == 3 ? bn_act_bwd_apply_kernel<true> : bn_act_bwd_apply_kernel<false>, dim3(lpd_reduce_grid(grid_for(R * (C / 4), 256 * 4))), dim3(256), 0, ST(stream), dY, lddy, X,
                       ldx, dX, lddx, R, C, scale, shift, mean, invstd, dbeta, dgamma, (double)R, act, slope, has_bn);
    LPD_CHECK_LAUNCH("lpd_bn_act_bwd(apply)");
    return LPD_OK;
}

// lpd_bn_act_bwd on bf16 tensors (dY, X, dX: [R][ld] bf16 elements, ld % 8 == 0, C a power of two in 8..2048); dX may alias dY.
extern "C" int lpd_bn_act_bwd_bf16(const void* dY, long long lddy, const void* X, long long ldx, void* dX, long long lddx, long long R, int C,
                                   const float* scale, const float* shift, const float* mean, const float* invstd, int act, float slope,
                                   int has_bn, double* dbeta, double* dgamma, double* stat_ws, void* stream)
{
    LPD_CHECK_ARG(dY && X && dX && dbeta && dgamma && R > 0, "lpd_bn_act_bwd_bf16: bad arguments");
    LPD_CHECK_ARG(C >= 8 && C <= 2048 && (C & (C - 1)) == 0 && lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 &&
                      (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dX) & 15) == 0, "lpd_bn_act_bwd_bf16: C=%d / leading dims / alignment unsupported", C);
    LPD_CHECK_ARG(!has_bn || (scale && shift && mean && invstd), "lpd_bn_act_bwd_bf16: BatchNorm form needs scale/shift/mean/invstd");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_bn_act_bwd_bf16: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    // the reduction goes through the statistics replicas in column panels of at most LPD_STAT_CMAX columns (like lpd_colstats): column c
    // of a replica sits at + c and the second array LPD_STAT_CMAX doubles behind the first, so a 2048-wide layer takes two passes
    for (int c0 = 0; c0 < C; c0 += LPD_STAT_CMAX) {
        const int w = C - c0 < LPD_STAT_CMAX ? C - c0 : LPD_STAT_CMAX;      // a power of two >= 8
        LPD_CHECK_STAT_COLS("lpd_bn_act_bwd_bf16", w);
        const int RG = 256 / (w / 8);
        const int grid = lpd_reduce_grid(grid_for(R, RG * 8));
        hipLaunchKernelGGL(act == 3 ? bn_act_bwd_reduce16_kernel<true> : bn_act_bwd_reduce16_kernel<false>, dim3(grid), dim3(256), 0, ST(stream),
                           reinterpret_cast<const uint16_t*>(dY) + c0, lddy, reinterpret_cast<const uint16_t*>(X) + c0, ldx, R, w, scale ? scale + c0 : nullptr,
                           shift ? shift + c0 : nullptr, has_bn ? mean + c0 : nullptr, has_bn ? invstd + c0 : nullptr, act, slope, ws.sum(), ws.sumsq());
        LPD_CHECK_LAUNCH("lpd_bn_act_bwd_bf16(reduce)");
        if (int rc = lpd_stat_finish(ws, dbeta + c0, dgamma + c0, w, ST(stream))) return rc;
    }
    // (the apply pass keeps eight 16-byte loads open per thread: three blocks per CU stream better than sixteen -- 4096 / 768 / 512 blocks:
    //  the two passes 448 / 393 / 391 us; the one-load-per-trip kernels of this file, lpd_affine_act among them, want the large grids)
    hipLaunchKernelGGL(act == 3 ? bn_act_bwd_apply16_kernel<true> : bn_act_bwd_apply16_kernel<false>, dim3(lpd_reduce_grid(grid_for(R * (C / 8), 256 * 4))), dim3(256), 0, ST(stream), reinterpret_cast<const uint16_t*>(dY),
                       lddy, reinterpret_cast<const uint16_t*>(X), ldx, reinterpret_cast<uint16_t*>(dX), lddx, R, C, scale, shift, mean, invstd, dbeta,
                       dgamma, (double)R, act, slope, has_bn);
    LPD_CHECK_LAUNCH("lpd_bn_act_bwd_bf16(apply)");
    return LPD_OK;
}

extern "C" int lpd_edge_build(const float* P, long long ldp, const float* Q, long long ldq, const int32_t* idx, float* U,
                              long long M, int N, int C, int k, double* sum, double* sumsq, double* stat_ws, void* stream)
{
    LPD_CHECK_ARG((sum == nullptr) == (sumsq == nullptr), "lpd_edge_build: sum and sumsq come in pairs");
    LpdStatWs ws = {nullptr};
    double* usum = sum;
    double* usumsq = sumsq;
    if (sum) {
        ws = lpd_stat_arg(stat_ws);
        LPD_CHECK_ARG(ws.rep, "lpd_edge_build: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
        LPD_CHECK_STAT_COLS("lpd_edge_build", C);
        sum = ws.sum();
        sumsq = ws.sumsq();
    }
    LPD_CHECK_ARG(P && idx && U && M > 0 && N > 0 && k > 0 && M % N == 0, "lpd_edge_build: bad arguments");
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_edge_build: C=%d unsupported (64/128/256)", C);
    LPD_CHECK_ARG(ldp % 4 == 0 && (!Q || ldq % 4 == 0), "lpd_edge_build: leading dims must be multiples of 4");
    const int lpp = C / 4;
    const int g = grid_for((M + 64 / lpp - 1) / (64 / lpp), 4);
    if (lpp == 64) hipLaunchKernelGGL(edge_build_kernel<64>, dim3(g), dim3(256), 0, ST(stream), P, ldp, Q, ldq, idx, U, M, N, k, sum, sumsq);
    else if (lpp == 32) hipLaunchKernelGGL(edge_build_kernel<32>, dim3(g), dim3(256), 0, ST(stream), P, ldp, Q, ldq, idx, U, M, N, k, sum, sumsq);
    else hipLaunchKernelGGL(edge_build_kernel<16>, dim3(g), dim3(256), 0, ST(stream), P, ldp, Q, ldq, idx, U, M, N, k, sum, sumsq);
    LPD_CHECK_LAUNCH("lpd_edge_build");
    if (usum) return lpd_stat_finish(ws, usum, usumsq, C, ST(stream));
    return LPD_OK;
}

static int group_max_impl(const float* X, long long ldx, int k, const float* scale, const float* shift, int act, float slope, float* out,
                          long long ldo, uint8_t* arg, float* xsel, long long ldsel, long long M, int C, void* stream)
{
    LPD_CHECK_ARG(X && scale && shift && out && arg && M > 0 && k > 0 && k <= 255, "lpd_group_max: bad arguments");
    LPD_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!xsel || ldsel % 4 == 0),
                  "lpd_group_max: C and leading dims must be multiples of 4");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_group_max: act=%d unsupported (none/ReLU/LeakyReLU)", act);
    hipLaunchKernelGGL(group_max_kernel, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, ST(stream), X, ldx, k, scale, shift,
                       act, slope, out, ldo, arg, xsel, ldsel, M, C);
    LPD_CHECK_LAUNCH("lpd_group_max");
    return LPD_OK;
}

extern "C" int lpd_group_max(const float* X, long long ldx, int k, const float* scale, const float* shift, int act,
                             float slope, float* out, long long ldo, uint8_t* arg, long long M, int C, void* stream)
{
    return group_max_impl(X, ldx, k, scale, shift, act, slope, out, ldo, arg, nullptr, 0, M, C, stream);
}

extern "C" int lpd_group_max_sel(const float* X, long long ldx, int k, const float* scale, const float* shift, int act,
                                 float slope, float* out, long long ldo, uint8_t* arg, float* xsel, long long ldsel, long long M,
                                 int C, void* stream)
{
    LPD_CHECK_ARG(xsel, "lpd_group_max_sel: xsel is null");
    return group_max_impl(X, ldx, k, scale, shift, act, slope, out, ldo, arg, xsel, ldsel, M, C, stream);
}

extern "C" int lpd_group_max_bwd(const float* dOut, long long ldo, const uint8_t* arg, int k, float* dX, long long M, int C,
                                 int accumulate, void* stream)
{
    LPD_CHECK_ARG(dOut && arg && dX && M > 0 && k > 0 && C % 4 == 0 && ldo % 4 == 0, "lpd_group_max_bwd: bad arguments");
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, ST(stream), dOut, ldo, arg, k, dX,
                       M, C, accumulate);
    LPD_CHECK_LAUNCH("lpd_group_max_bwd");
    return LPD_OK;
}

static int edge_bn_bwd_impl(const float* dOut, long long ldo, const uint8_t* arg, const float* dDense, const float* X,
                            const float* Xsel, long long ldsel, float* dX, float* dQ, long long ldq, int k, long long M, int C,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int act, float slope,
                            float inv_ns, double* dbeta, double* dgamma, double* stat_ws, void* stream)
{
    LPD_CHECK_ARG(inv_ns == 0.0f || (dDense && (act == 0 || act == 2) && inv_ns >= 1.0f),
                  "lpd_edge_bn_bwd: post-activation X needs the dense form and an invertible activation (none / LeakyReLU)");
    LPD_CHECK_ARG(dOut && arg && X && dX && scale && shift && mean && invstd && dbeta && dgamma && M > 0 && k > 0 && k <= 255,
                  "lpd_edge_bn_bwd: bad arguments");
    LPD_CHECK_ARG(cols_ok(C) && ldo % 4 == 0 && (!dQ || ldq % 4 == 0) && (!Xsel || ldsel % 4 == 0),
                  "lpd_edge_bn_bwd: C=%d / leading dims unsupported", C);
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_bn_bwd: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_bn_bwd", C);
    const int RG = 256 / (C / 4);
    hipLaunchKernelGGL(edge_bn_bwd_reduce_kernel, dim3(grid_for(M, RG * 2)), dim3(256), 0, ST(stream), dOut, ldo, arg, dDense, X, Xsel,
                       ldsel, k, M, C, scale, shift, mean, invstd, act, slope, inv_ns, ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_edge_bn_bwd(reduce)");
    if (int rc = lpd_stat_finish(ws, dbeta, dgamma, C, ST(stream))) return rc;
    hipLaunchKernelGGL(edge_bn_bwd_apply_kernel, dim3(grid_for(M, RG)), dim3(256), 0, ST(stream), dOut, ldo, arg, dDense, X, dX, dQ,
                       ldq, k, M, C, scale, shift, mean, invstd, dbeta, dgamma, (double)M * k, act, slope, inv_ns);
    LPD_CHECK_LAUNCH("lpd_edge_bn_bwd(apply)");
    return LPD_OK;
}

extern "C" int lpd_edge_bn_bwd(const float* dOut, long long ldo, const uint8_t* arg, const float* dDense, const float* X,
                               float* dX, float* dQ, long long ldq, int k, long long M, int C, const float* scale,
                               const float* shift, const float* mean, const float* invstd, int act, float slope,
                               float inv_ns, double* dbeta, double* dgamma, double* stat_ws, void* stream)
{
    return edge_bn_bwd_impl(dOut, ldo, arg, dDense, X, nullptr, 0, dX, dQ, ldq, k, M, C, scale, shift, mean, invstd, act, slope, inv_ns,
                            dbeta, dgamma, stat_ws, stream);
}

extern "C" int lpd_edge_bn_bwd_sel(const float* dOut, long long ldo, const uint8_t* arg, const float* X, const float* Xsel,
                                   long long ldsel, float* dX, int k, long long M, int C, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, int act, float slope, double* dbeta, double* dgamma,
                                   double* stat_ws, void* stream)
{
    LPD_CHECK_ARG(Xsel, "lpd_edge_bn_bwd_sel: Xsel is null");
    return edge_bn_bwd_impl(dOut, ldo, arg, nullptr, X, Xsel, ldsel, dX, nullptr, 0, k, M, C, scale, shift, mean, invstd, act, slope,
                            0.0f, dbeta, dgamma, stat_ws, stream);
}

extern "C" int lpd_group_sum(const float* dU, int k, float* dQ, long long ldq, long long M, int C, void* stream)
{
    LPD_CHECK_ARG(dU && dQ && M > 0 && k > 0 && C % 4 == 0 && ldq % 4 == 0, "lpd_group_sum: bad arguments");
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, ST(stream), dU, k, dQ, ldq, M, C);
    LPD_CHECK_LAUNCH("lpd_group_sum");
    return LPD_OK;
}

extern "C" int lpd_scatter_add_rows(const float* dU, const int32_t* idx, float* dP, long long ldp, long long M, int N, int k,
                                    int C, void* stream)
{
    LPD_CHECK_ARG(dU && idx && dP && M > 0 && N > 0 && k > 0 && M % N == 0 && C > 0, "lpd_scatter_add_rows: bad arguments");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(M, 4)), dim3(256), 0, ST(stream), dU, idx, dP, ldp, M, N, k, C);
    LPD_CHECK_LAUNCH("lpd_scatter_add_rows");
    return LPD_OK;
}

extern "C" int lpd_graph_transpose(const int32_t* idx, long long M, int N, int k, int32_t* rowptr, int32_t* edges, int32_t* ws,
                                   void* stream)
{
    LPD_CHECK_ARG(idx && rowptr && edges && ws && M > 0 && N > 0 && k > 0 && M % N == 0, "lpd_graph_transpose: bad arguments");
    LPD_CHECK_ARG(M * k < (1ll << 31), "lpd_graph_transpose: more than 2^31 edges");
    const long long E = M * k;
    if (N <= 32768) {           // one launch, LDS atomics (the row range of a block fits LDS)
        const int G = 4, nbmax = (N + G - 1) / G + 1;
        hipLaunchKernelGGL(csr_build_kernel, dim3((unsigned)(M / N * G)), dim3(1024), (nbmax + 1024) * sizeof(int), ST(stream), idx, rowptr,
                           edges, N, k, M, G, nbmax);
        LPD_CHECK_LAUNCH("lpd_graph_transpose");
        return LPD_OK;
    }
    int32_t* deg = ws;          // [M]
    int32_t* cursor = ws + M;   // [M]
    (void)hipMemsetAsync(deg, 0, sizeof(int32_t) * M, ST(stream));
    hipLaunchKernelGGL(csr_count_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, ST(stream), idx, deg, E, N, k);
    hipLaunchKernelGGL(csr_scan_kernel, dim3((unsigned)(M / N)), dim3(1024), 0, ST(stream), (const int32_t*)deg, rowptr, cursor, N, k, M);
    hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, ST(stream), idx, cursor, edges, E, N, k);
    LPD_CHECK_LAUNCH("lpd_graph_transpose");
    return LPD_OK;
}

extern "C" int lpd_gather_sum_rows(const float* dU, const int32_t* rowptr, const int32_t* edges, float* dP, long long ldp,
                                   long long M, int C, int accumulate, void* stream)
{
    LPD_CHECK_ARG(dU && rowptr && edges && dP && M > 0, "lpd_gather_sum_rows: bad arguments");
    LPD_CHECK_ARG((C == 64 || C == 128 || C == 256) && ldp % 4 == 0, "lpd_gather_sum_rows: C=%d unsupported (64/128/256), ldp %% 4", C);
    const int rpw = 256 / C;   // rows per wave
    const int grid = grid_for((M + rpw - 1) / rpw, 4);
    if (C == 256) hipLaunchKernelGGL(gather_sum_rows_kernel<64>, dim3(grid), dim3(256), 0, ST(stream), dU, rowptr, edges, dP, ldp, M, accumulate);
    else if (C == 128) hipLaunchKernelGGL(gather_sum_rows_kernel<32>, dim3(grid), dim3(256), 0, ST(stream), dU, rowptr, edges, dP, ldp, M, accumulate);
    else hipLaunchKernelGGL(gather_sum_rows_kernel<16>, dim3(grid), dim3(256), 0, ST(stream), dU, rowptr, edges, dP, ldp, M, accumulate);
    LPD_CHECK_LAUNCH("lpd_gather_sum_rows");
    return LPD_OK;
}

extern "C" int lpd_dw_smallk(const float* dY, long long lddy, const float* X, long long ldx, long long M, int Co, int Kin,
                             float* dW, void* stream)
{
    LPD_CHECK_ARG(dY && X && dW && M > 0, "lpd_dw_smallk: bad arguments");
    LPD_CHECK_ARG(Co > 0 && Co <= 256 && 256 % Co == 0 && Kin > 0 && Kin <= 8, "lpd_dw_smallk: Co=%d Kin=%d unsupported", Co, Kin);
    (void)hipMemsetAsync(dW, 0, sizeof(float) * Co * Kin, ST(stream));
    hipLaunchKernelGGL(dw_smallk_kernel, dim3(grid_for(M, (256 / Co) * 64)), dim3(256), 0, ST(stream), dY, lddy, X, ldx, M, Co,
                       Kin, dW);
    LPD_CHECK_LAUNCH("lpd_dw_smallk");
    return LPD_OK;
}

extern "C" int lpd_softmax_bwd(const float* A, const float* dA, const float* dasum, float* dS, long long rows, int ncols,
                               int rows_per_cloud, void* stream)
{
    LPD_CHECK_ARG(A && dA && dS && rows > 0 && ncols > 0 && ncols <= 64 && rows_per_cloud > 0, "lpd_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), A, dA, dasum, dS, rows,
                       ncols, rows_per_cloud);
    LPD_CHECK_LAUNCH("lpd_softmax_bwd");
    return LPD_OK;
}

extern "C" int lpd_vlad_finalize_bwd(const float* dOut, const float* v, const float* inv_c, const float* inv_g,
                                     const float* asum, const float* cw2, float* dVraw, float* dasum, float* dcw2, int B,
                                     int F, int KC, void* stream)
{
    LPD_CHECK_ARG(dOut && v && inv_c && inv_g && asum && cw2 && dVraw && dasum && dcw2 && B > 0 && F > 0,
                  "lpd_vlad_finalize_bwd: bad arguments");
    LPD_CHECK_ARG(KC == 64, "lpd_vlad_finalize_bwd: cluster_size=%d unsupported (64)", KC);
    // slices per cloud: the per-slice partial sums (B * G * (1 + 2 KC) floats) live in the dcw2 buffer until it is written
    int G = 8;
    while (G > 1 && (long long)B * G * (1 + 2 * KC) > (long long)F * KC) G >>= 1;
    if ((long long)B * G * (1 + 2 * KC) <= (long long)F * KC && F >= 8 * G && B <= 65535) {
        float* part_dvv = dcw2;
        float* part_col = dcw2 + (size_t)B * G;
        float* part_pa = part_col + (size_t)B * G * KC;
        hipLaunchKernelGGL(vlad_fbwd_dot_kernel<64>, dim3(B, G), dim3(256), 0, ST(stream), dOut, v, F, G, part_dvv);
        hipLaunchKernelGGL(vlad_fbwd_col_kernel<64>, dim3(B, G), dim3(256), 0, ST(stream), dOut, v, inv_g, F, G, (const float*)part_dvv,
                           part_col);
        hipLaunchKernelGGL(vlad_fbwd_apply_kernel<64>, dim3(B, G), dim3(256), 0, ST(stream), dOut, v, inv_c, inv_g, cw2, F, G,
                           (const float*)part_dvv, (const float*)part_col, dVraw, part_pa);
        hipLaunchKernelGGL(vlad_fbwd_dasum_kernel<64>, dim3((B * KC + 255) / 256), dim3(256), 0, ST(stream), (const float*)part_pa, B, G, dasum);
        hipLaunchKernelGGL(vlad_fbwd_dcw2_kernel<64>, dim3((F * KC + 255) / 256), dim3(256), 0, ST(stream), (const float*)dVraw, asum, B, F, dcw2);
        LPD_CHECK_LAUNCH("lpd_vlad_finalize_bwd");
        return LPD_OK;
    }
    (void)hipMemsetAsync(dcw2, 0, sizeof(float) * (size_t)F * KC, ST(stream));
    hipLaunchKernelGGL(vlad_finalize_bwd_kernel<64>, dim3(B), dim3(256), 0, ST(stream), dOut, v, inv_c, inv_g, asum, cw2, dVraw,
                       dasum, dcw2, F);
    LPD_CHECK_LAUNCH("lpd_vlad_finalize_bwd");
    return LPD_OK;
}

extern "C" int lpd_colmax_arg(const float* in, long long ld, float* out, int32_t* arg, int B, int N, int C, void* stream)
{
    LPD_CHECK_ARG(in && out && arg && B > 0 && B <= 65535 && N > 0 && C > 0, "lpd_colmax_arg: bad arguments");
    hipLaunchKernelGGL(colmax_arg_kernel, dim3((C + 63) / 64, B), dim3(256), 0, ST(stream), in, ld, out, arg, N, C);
    LPD_CHECK_LAUNCH("lpd_colmax_arg");
    return LPD_OK;
}

extern "C" int lpd_colmax_bwd(const float* dOut, const int32_t* arg, float* dIn, long long ld, int B, int N, int C, void* stream)
{
    LPD_CHECK_ARG(dOut && arg && dIn && B > 0 && N > 0 && C > 0, "lpd_colmax_bwd: bad arguments");
    hipLaunchKernelGGL(colmax_bwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, ST(stream), dOut, arg, dIn, ld, B, N, C);
    LPD_CHECK_LAUNCH("lpd_colmax_bwd");
    return LPD_OK;
}

extern "C" int lpd_cloud_outer(const float* X, long long ldx, const float* dY, long long ldy, float* dT, int B, int N, int KD,
                               void* stream)
{
    LPD_CHECK_ARG(X && dY && dT && B > 0 && N > 0 && KD > 0 && KD <= 8, "lpd_cloud_outer: bad arguments (KD <= 8)");
    hipLaunchKernelGGL(cloud_outer_kernel, dim3(B), dim3(256), 0, ST(stream), X, ldx, dY, ldy, dT, N, KD);
    LPD_CHECK_LAUNCH("lpd_cloud_outer");
    return LPD_OK;
}
