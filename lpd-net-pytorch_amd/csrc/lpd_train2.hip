// lpd_train2.hip -- training path, second generation:
//
//  (1) SPLIT-FORM edge stage for x3 = max_k act(BN_train(convSN1(cat(f_j, f_i))))  (util/lpdnet_model.py:256-258) that never
//      materialises the [B*N*k, 256] edge tensor U[(i,t)] = P[nbr(i,t)] + Q[i]:
//        * forward: ONE gather pass gives, per point, S_i = sum_t P_nbr, the selected raw value  usel_i = sel_t P_nbr + Q_i
//          (sel = max where gamma >= 0, min where gamma < 0: the sign of the BatchNorm scale is the sign of gamma, known
//          before the statistics), its slot arg_i, and the batch statistics of U in closed form
//              sum U   = sum_i (S_i + k Q_i),     sum U^2 = sum_i (sum_t P_nbr^2 + 2 Q_i S_i + k Q_i^2)        (fp64);
//          x3 = act(scale * usel + shift) is then an [M, C] elementwise pass.
//        * backward: with dpre_i = dx3_i * act'(pre_i) living on the arg-max edge only and the BatchNorm means
//          m1 = mean(dpre), m2 = mean(dpre * xhat) over all E = M k edges,
//              dU[(i,t)] = s (delta_{t,arg} dpre_i - m1 - xhat[(i,t)] m2),   xhat = (P_nbr + Q_i - mu) invstd
//          sums in closed form to
//              dQ_i = s (dpre_i - k m1 - m2 invstd (S_i + k Q_i - k mu))
//              dP_j = s (A_j - deg_j m1 - m2 invstd (deg_j (P_j - mu) + R_j)),
//              A_j = sum over incoming edges (i,t) with arg_i = t of dpre_i,   R_j = sum over incoming edges of Q_i,
//          one pass over the transposed graph (no float atomics).
//      Replaces edge_build + group_max + edge_bn_bwd + gather_sum_rows on 3.7 GB tensors (B = 44) by [M, C] passes and two
//      gathers.
//
//  (2) bf16 STORAGE of the DG1 -> DG2 chain's edge tensors (BASELINE.json configs[2] "bf16"): the per-edge tensors that
//      must exist (convDG2 consumes every post-activation edge, lpdnet_model.py:249-252) are kept as bf16, every statistic /
//      reduction / accumulation stays fp32-fp64: the two dense products on them run on v_mfma_f32_32x32x16_bf16
//      (lpd_gemm_bf16s: activations bf16 x weights split hi+lo; lpd_gemm_tn_bf16: dW = dZ^T Y).
#include "lpd_common.h"
#include <math.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float act_grad2(float pre, int act, float slope)
{
    return pre > 0.0f ? 1.0f : lpd_neg_slope(act, slope);
}

__device__ __forceinline__ float bf2f(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ float4 ld4_bf16(const uint16_t* p)
{
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf2f(u.x & 0xffffu), bf2f(u.x >> 16), bf2f(u.y & 0xffffu), bf2f(u.y >> 16));
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b)   // RNE (v_cvt_pk_bf16_f32)
{
    const bf16x2v h = __builtin_convertvector((f32x2v){a, b}, bf16x2v);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ void st4_bf16(uint16_t* p, float4 v)
{
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}
__device__ __forceinline__ float rbf(float x) { return bf2f(pack_bf16(x, 0.0f) & 0xffffu); }   // value after bf16 storage

// block reduction of per-thread fp64 partials [8] that belong to column quad (threadIdx.x % LQ) -> fp64 atomics
template <int NT>
__device__ __forceinline__ void reduce_quads(double (&red)[NT][8], const double (&a)[4], const double (&b)[4], int LQ,
                                             double* __restrict__ o0, double* __restrict__ o1)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = a[e]; red[threadIdx.x][4 + e] = b[e]; }
    __syncthreads();
    if ((int)threadIdx.x < LQ) {
        for (int g = 1; g < NT / LQ; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[g * LQ + threadIdx.x][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&o0[lpd_stat_rofs() + threadIdx.x * 4 + e], red[threadIdx.x][e]);
            atomicAdd(&o1[lpd_stat_rofs() + threadIdx.x * 4 + e], red[threadIdx.x][4 + e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// (1) split-form edge stage
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(256) void edge_split_fwd_kernel(const float* __restrict__ P, long long ldp, const float* __restrict__ Q,
                                                             long long ldq, const int32_t* __restrict__ idx,
                                                             const float* __restrict__ gamma, float* __restrict__ S,
                                                             float* __restrict__ usel, uint8_t* __restrict__ arg, long long M,
                                                             int N, int k, double* __restrict__ sum, double* __restrict__ sumsq)
{
    constexpr int PPW = 64 / LPP;
    constexpr int C = LPP * 4;
    __shared__ double red[256][8];
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPP, cl = lane % LPP;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + cl * 4);
    const bool up[4] = {gm.x >= 0.f, gm.y >= 0.f, gm.z >= 0.f, gm.w >= 0.f};
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float kf = (float)k;
    for (long long w = wave; w * PPW < M; w += nw) {
        const long long m = w * PPW + sub;
        const bool ok = m < M;
        const long long mm = ok ? m : M - 1;
        const long long base = (mm / N) * N;
        const float4 q4 = *reinterpret_cast<const float4*>(Q + mm * ldq + cl * 4);
        float sp[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
        float best[4];
        int ab[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) best[c] = up[c] ? -INFINITY : INFINITY;
        for (int t0 = 0; t0 < k; t0 += LPP) {
            const int my_idx = (t0 + cl < k) ? idx[mm * k + t0 + cl] : 0;
            const int tn = min(k - t0, LPP);
#pragma unroll 5
            for (int t = 0; t < tn; ++t) {
                const int j = __shfl(my_idx, sub * LPP + t, 64);
                const float4 p4 = *reinterpret_cast<const float4*>(P + (base + j) * ldp + cl * 4);
                const float p[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sp[c] += p[c];
                    sq[c] = fmaf(p[c], p[c], sq[c]);
                    const bool take = up[c] ? (p[c] > best[c]) : (p[c] < best[c]);   // first extremum, like torch.max
                    best[c] = take ? p[c] : best[c];
                    ab[c] = take ? t0 + t : ab[c];
                }
            }
        }
        if (ok) {
            const float q[4] = {q4.x, q4.y, q4.z, q4.w};
            *reinterpret_cast<float4*>(S + mm * C + cl * 4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
            *reinterpret_cast<float4*>(usel + mm * C + cl * 4) = make_float4(best[0] + q[0], best[1] + q[1], best[2] + q[2], best[3] + q[3]);
            *reinterpret_cast<uchar4*>(arg + mm * C + cl * 4) = make_uchar4((uint8_t)ab[0], (uint8_t)ab[1], (uint8_t)ab[2], (uint8_t)ab[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s[c] += (double)sp[c] + (double)kf * q[c];
                ss[c] += (double)sq[c] + 2.0 * (double)q[c] * sp[c] + (double)kf * q[c] * q[c];
            }
        }
    }
    reduce_quads<256>(red, s, ss, LPP, sum, sumsq);
}

// G = dOut * act'(scale * usel + shift);  dbeta = sum G,  dgamma = sum G * (usel - mean) * invstd    (fp64)
// H (bf16 storage): G is written as bf16 rows, and a bf16 copy of Q behind it ([2][M][C] bf16 in the fp32 buffer's bytes) -- the apply
// kernel gathers both over the transposed graph (k rows each per point: 2.25 KiB per edge in fp32), the sums stay those of the fp32 G
template <bool H>
__global__ __launch_bounds__(256) void edge_split_bwd_reduce_kernel(const float* __restrict__ dOut, long long ldo,
                                                                    const float* __restrict__ usel, float* __restrict__ G,
                                                                    const float* __restrict__ Q, long long ldq,
                                                                    long long M, int C, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, int act, float slope,
                                                                    double* __restrict__ dbeta, double* __restrict__ dgamma)
{
    __shared__ double red[256][8];
    const int LQ = C >> 2;
    const int RG = 256 / LQ;
    const int q = threadIdx.x % LQ, rg = threadIdx.x / LQ;
    float sc[4], sh[4], mu[4], is[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = scale[q * 4 + c]; sh[c] = shift[q * 4 + c]; mu[c] = mean[q * 4 + c]; is[c] = invstd[q * 4 + c]; }
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        const float4 g4 = *reinterpret_cast<const float4*>(dOut + i * ldo + q * 4);
        const float4 u4 = *reinterpret_cast<const float4*>(usel + i * C + q * 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const float u[4] = {u4.x, u4.y, u4.z, u4.w};
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o[c] = g[c] * act_grad2(sc[c] * u[c] + sh[c], act, slope);
            sb[c] += o[c];
            sg[c] += (double)o[c] * ((u[c] - mu[c]) * is[c]);
        }
        if constexpr (H) {
            uint16_t* G16 = reinterpret_cast<uint16_t*>(G);
            st4_bf16(G16 + i * C + q * 4, make_float4(o[0], o[1], o[2], o[3]));
            st4_bf16(G16 + (M + i) * C + q * 4, *reinterpret_cast<const float4*>(Q + i * ldq + q * 4));
        } else {
            *reinterpret_cast<float4*>(G + i * C + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    reduce_quads<256>(red, sb, sg, LQ, dbeta, dgamma);
}

// one (part of a) wave per point j: dQ_j (own row) and dP_j (over the incoming edges of the transposed graph).
// (Tried in round 3 and dropped: the dP slice of a (cloud, 8 channels) accumulated in LDS over the FORWARD graph with ds_add_f32 --
// 923 M lane-adds at B = 44 took 5.3 ms, i.e. ~0.36 lane-adds per clock and CU: LDS float atomics are no bulk accumulator.  The
// forward on an LDS-resident P slice, without the eval K-agg kernel's operand pipeline and persistent blocks: 513 against 559 us --
// the forward now runs on the eval kernel's organisation, lpd_edge.hip edge_split_fwd_cloud16_kernel: 352 us.  Also tried for this
// backward: a block holding the 4-channel slices of Q, dpre and arg of a whole cloud in LDS and walking a degree-sorted, slot-major
// transposed graph (built by LDS atomics + a counting sort in 56 us; heavy rows by a scan kernel): correct, but 8.3 GB of random
// 16 / 16 / 4-byte LDS reads plus the scattered slice fills cost 762 us against 930 here.)
// OUT16 (with H): dP / dQ are written as bf16 rows (lddp / lddq in bf16 elements): the gradient of the SN1 projection in the bf16 mode
template <int LPR, bool H = false, bool OUT16 = false>
__global__ __launch_bounds__(256) void edge_split_bwd_apply_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ edges,
                                                                   const float* __restrict__ G, const uint8_t* __restrict__ arg,
                                                                   const float* __restrict__ S, const float* __restrict__ P,
                                                                   long long ldp, const float* __restrict__ Q, long long ldq,
                                                                   float* __restrict__ dP, long long lddp, float* __restrict__ dQ,
                                                                   long long lddq, long long M, int k,
                                                                   const float* __restrict__ scale, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const double* __restrict__ dbeta,
                                                                   const double* __restrict__ dgamma)
{
    constexpr int RPW = 64 / LPR;
    constexpr int C = LPR * 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, cl = lane % LPR;
    const LpdXcdSweep sweep = lpd_xcd_sweep((M + RPW - 1) / RPW);
    const double E = (double)M * (double)k;
    float sc[4], mu[4], is[4], m1[4], m2i[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = cl * 4 + c;
        sc[c] = scale[ch]; mu[c] = mean[ch]; is[c] = invstd[ch];
        m1[c] = (float)(dbeta[ch] / E);
        m2i[c] = (float)(dgamma[ch] / E) * is[c];
    }
    const float kf = (float)k;
    const unsigned ku = (unsigned)k;
    // H: bf16 rows of G and Q (the reduce kernel's copies), four channels = 8 bytes per lane
    const uint16_t* G16 = reinterpret_cast<const uint16_t*>(G);
    const uint16_t* Q16 = G16 + M * C;
    auto row4 = [&](const float* f32, long long ld, const uint16_t* b16, long long i) -> float4 {
        if constexpr (H) {
            const uint2 w = *reinterpret_cast<const uint2*>(b16 + i * C + cl * 4);
            return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
        } else {
            return *reinterpret_cast<const float4*>(f32 + i * ld + cl * 4);
        }
    };
    for (long long wi = sweep.begin; wi < sweep.end; wi += sweep.step) {
        const long long j = wi * RPW + sub;
        if (j >= M) continue;
        const int beg = rowptr[j], end = rowptr[j + 1];
        float A[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
        int p = beg;
        // (Round 5, tried and dropped: the row's edge ids in ONE load handed out by lane shuffles + eight edges' rows per batch -- the chain
        //  rowptr -> ids -> rows shortened from 11 to 5 dependent round trips at degree 20.  No gain: this kernel 690 -> 933 us in the bf16
        //  form (134 registers, three waves per SIMD), 788 -> 789 us in fp32, the dense counterpart 367 -> 358: the chain is not what the
        //  parked waves wait for; profiles/r05_train_gather_batch8.txt.)
        for (; p + 3 < end; p += 4) {          // four incoming edges at a time: twelve independent row loads in flight (1104 -> 930 us)
            unsigned ii[4], tt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned e = (unsigned)edges[p + u];
                ii[u] = e / ku;
                tt[u] = e - ii[u] * ku;
            }
            float4 q4[4], g4[4];
            uchar4 a4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q4[u] = row4(Q, ldq, Q16, (long long)ii[u]);
                g4[u] = row4(G, C, G16, (long long)ii[u]);
                a4[u] = *reinterpret_cast<const uchar4*>(arg + (long long)ii[u] * C + cl * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // same order of additions as the one-edge loop
                R[0] += q4[u].x; R[1] += q4[u].y; R[2] += q4[u].z; R[3] += q4[u].w;
                A[0] += a4[u].x == tt[u] ? g4[u].x : 0.f; A[1] += a4[u].y == tt[u] ? g4[u].y : 0.f;
                A[2] += a4[u].z == tt[u] ? g4[u].z : 0.f; A[3] += a4[u].w == tt[u] ? g4[u].w : 0.f;
            }
        }
        for (; p < end; ++p) {
            const unsigned e = (unsigned)edges[p];
            const unsigned i = e / ku, t = e - i * ku;
            const float4 q4 = row4(Q, ldq, Q16, (long long)i);
            const float4 g4 = row4(G, C, G16, (long long)i);
            const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + (long long)i * C + cl * 4);
            R[0] += q4.x; R[1] += q4.y; R[2] += q4.z; R[3] += q4.w;
            A[0] += a4.x == t ? g4.x : 0.f; A[1] += a4.y == t ? g4.y : 0.f;
            A[2] += a4.z == t ? g4.z : 0.f; A[3] += a4.w == t ? g4.w : 0.f;
        }
        const float deg = (float)(end - beg);
        const float4 p4 = *reinterpret_cast<const float4*>(P + j * ldp + cl * 4);
        const float4 q4 = *reinterpret_cast<const float4*>(Q + j * ldq + cl * 4);
        const float4 g4 = row4(G, C, G16, j);
        const float4 s4 = *reinterpret_cast<const float4*>(S + j * C + cl * 4);
        const float pj[4] = {p4.x, p4.y, p4.z, p4.w}, qj[4] = {q4.x, q4.y, q4.z, q4.w};
        const float gj[4] = {g4.x, g4.y, g4.z, g4.w}, sj[4] = {s4.x, s4.y, s4.z, s4.w};
        float op[4], oq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            op[c] = sc[c] * (A[c] - deg * m1[c] - m2i[c] * (deg * (pj[c] - mu[c]) + R[c]));
            oq[c] = sc[c] * (gj[c] - kf * m1[c] - m2i[c] * (sj[c] + kf * (qj[c] - mu[c])));
        }
        if constexpr (OUT16) {
            st4_bf16(reinterpret_cast<uint16_t*>(dP) + j * lddp + cl * 4, make_float4(op[0], op[1], op[2], op[3]));
            st4_bf16(reinterpret_cast<uint16_t*>(dQ) + j * lddq + cl * 4, make_float4(oq[0], oq[1], oq[2], oq[3]));
        } else {
            *reinterpret_cast<float4*>(dP + j * lddp + cl * 4) = make_float4(op[0], op[1], op[2], op[3]);
            *reinterpret_cast<float4*>(dQ + j * lddq + cl * 4) = make_float4(oq[0], oq[1], oq[2], oq[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// (2) bf16-storage edge kernels (rows (i,t) = i*k + t, [E][C] bf16, C in {64, 128, 256})
// ---------------------------------------------------------------------------------------------
// U = P[nbr] + Q as bf16, BatchNorm statistics of the STORED (rounded) values accumulated on the way (fp64)
template <int LPP>
__global__ __launch_bounds__(256) void edge_build_bf16_kernel(const float* __restrict__ P, long long ldp, const float* __restrict__ Q,
                                                              long long ldq, const int32_t* __restrict__ idx, uint16_t* __restrict__ U,
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
            const int my_idx = (t0 + cl < k) ? idx[mm * k + t0 + cl] : 0;
            const int tn = min(k - t0, LPP);
#pragma unroll 5
            for (int t = 0; t < tn; ++t) {
                const int j = __shfl(my_idx, sub * LPP + t, 64);
                float4 p = *reinterpret_cast<const float4*>(P + (base + j) * ldp + cl * 4);
                p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w;
                if (ok) {
                    const uint2 pk = make_uint2(pack_bf16(p.x, p.y), pack_bf16(p.z, p.w));
                    *reinterpret_cast<uint2*>(U + (mm * k + t0 + t) * C + cl * 4) = pk;
                    if (sum) {
                        const float r[4] = {bf2f(pk.x & 0xffffu), bf2f(pk.x >> 16), bf2f(pk.y & 0xffffu), bf2f(pk.y >> 16)};
#pragma unroll
                        for (int c = 0; c < 4; ++c) { s[c] += r[c]; ss[c] += (double)r[c] * r[c]; }
                    }
                }
            }
        }
    }
    if (sum) reduce_quads<256>(red, s, ss, LPP, sum, sumsq);
}

// One pass over U (bf16): Y = act(scale * U + shift) as bf16 (the dense consumer's input) AND
// out[i] = act(scale * sel_t U + shift), arg[i] (the max over k of the same values).
__global__ __launch_bounds__(256) void edge_act_max_bf16_kernel(const uint16_t* __restrict__ U, int k, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int act, float slope,
                                                                uint16_t* __restrict__ Y, float* __restrict__ out, long long ldo,
                                                                uint8_t* __restrict__ arg, long long M, int C)
{
    const int LQ = C >> 2;
    const int RG = 256 / LQ;
    const int q = threadIdx.x % LQ, rg = threadIdx.x / LQ;
    const float ns = lpd_neg_slope(act, slope);
    float sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = scale[q * 4 + c]; sh[c] = shift[q * 4 + c]; }
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        float mx[4], mn[4];
        int amx[4] = {0, 0, 0, 0}, amn[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) { mx[c] = -INFINITY; mn[c] = INFINITY; }
        for (int t = 0; t < k; ++t) {
            const long long off = (i * k + t) * C + q * 4;
            const float4 v4 = ld4_bf16(U + off);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (v[c] > mx[c]) { mx[c] = v[c]; amx[c] = t; }
                if (v[c] < mn[c]) { mn[c] = v[c]; amn[c] = t; }
                y[c] = lpd_act_pl(sc[c] * v[c] + sh[c], ns);
            }
            st4_bf16(Y + off, make_float4(y[0], y[1], y[2], y[3]));
        }
        float o[4];
        uint8_t a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool usemax = sc[c] >= 0.0f;
            o[c] = lpd_act_pl(sc[c] * (usemax ? mx[c] : mn[c]) + sh[c], ns);
            a[c] = (uint8_t)(usemax ? amx[c] : amn[c]);
        }
        *reinterpret_cast<float4*>(out + i * ldo + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uchar4*>(arg + i * C + q * 4) = make_uchar4(a[0], a[1], a[2], a[3]);
    }
}

// The same pass for fp32 edge tensors (fp32 storage mode: replaces lpd_group_max + lpd_affine_act, two reads of U, by one)
__global__ __launch_bounds__(256) void edge_act_max_f32_kernel(const float* __restrict__ U, int k, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int act, float slope,
                                                               float* __restrict__ Y, float* __restrict__ out, long long ldo,
                                                               uint8_t* __restrict__ arg, long long M, int C)
{
    const int LQ = C >> 2;
    const int RG = 256 / LQ;
    const int q = threadIdx.x % LQ, rg = threadIdx.x / LQ;
    const float ns = lpd_neg_slope(act, slope);
    float sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = scale[q * 4 + c]; sh[c] = shift[q * 4 + c]; }
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        float mx[4], mn[4];
        int amx[4] = {0, 0, 0, 0}, amn[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) { mx[c] = -INFINITY; mn[c] = INFINITY; }
        for (int t = 0; t < k; ++t) {
            const long long off = (i * k + t) * C + q * 4;
            const float4 v4 = *reinterpret_cast<const float4*>(U + off);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (v[c] > mx[c]) { mx[c] = v[c]; amx[c] = t; }
                if (v[c] < mn[c]) { mn[c] = v[c]; amn[c] = t; }
                y[c] = lpd_act_pl(sc[c] * v[c] + sh[c], ns);
            }
            *reinterpret_cast<float4*>(Y + off) = make_float4(y[0], y[1], y[2], y[3]);
        }
        float o[4];
        uint8_t a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool usemax = sc[c] >= 0.0f;
            o[c] = lpd_act_pl(sc[c] * (usemax ? mx[c] : mn[c]) + sh[c], ns);
            a[c] = (uint8_t)(usemax ? amx[c] : amn[c]);
        }
        *reinterpret_cast<float4*>(out + i * ldo + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uchar4*>(arg + i * C + q * 4) = make_uchar4(a[0], a[1], a[2], a[3]);
    }
}

// One pass over Z (bf16, raw conv output): batch statistics of Z (fp64) and the raw selected value per point
// sel[i] = sel_t Z[(i,t)] (max where gamma >= 0, min where gamma < 0) with its slot; the BatchNorm + activation of the
// selected values is an [M, C] pass afterwards (the sign of the scale is the sign of gamma).
__global__ __launch_bounds__(256) void group_sel_stats_bf16_kernel(const uint16_t* __restrict__ Z, int k, const float* __restrict__ gamma,
                                                                   float* __restrict__ sel, long long lds, uint8_t* __restrict__ arg,
                                                                   long long M, int C, double* __restrict__ sum,
                                                                   double* __restrict__ sumsq)
{
    __shared__ double red[256][8];
    const int LQ = C >> 2;
    const int RG = 256 / LQ;
    const int q = threadIdx.x % LQ, rg = threadIdx.x / LQ;
    bool up[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) up[c] = gamma[q * 4 + c] >= 0.0f;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        float best[4], ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
        int ab[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) best[c] = up[c] ? -INFINITY : INFINITY;
        for (int t = 0; t < k; ++t) {
            const float4 v4 = ld4_bf16(Z + (i * k + t) * C + q * 4);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool take = up[c] ? (v[c] > best[c]) : (v[c] < best[c]);
                best[c] = take ? v[c] : best[c];
                ab[c] = take ? t : ab[c];
                ps[c] += v[c];
                pq[c] = fmaf(v[c], v[c], pq[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[c] += ps[c]; ss[c] += pq[c]; }
        *reinterpret_cast<float4*>(sel + i * lds + q * 4) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<uchar4*>(arg + i * C + q * 4) = make_uchar4((uint8_t)ab[0], (uint8_t)ab[1], (uint8_t)ab[2], (uint8_t)ab[3]);
    }
    reduce_quads<256>(red, s, ss, LQ, sum, sumsq);
}

// fused backward of out[i] = max_t act(BN(X[(i,t)])) (+ optional dense gradient) on bf16 tensors (cf. lpd_edge_bn_bwd)
__global__ __launch_bounds__(256) void edge_bn_bwd_reduce_bf16_kernel(const float* __restrict__ dOut, long long ldo,
                                                                      const uint8_t* __restrict__ arg, const uint16_t* __restrict__ dDense,
                                                                      const uint16_t* __restrict__ X, const float* __restrict__ Xsel,
                                                                      long long ldsel, int k, long long M, int C,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                      int act, float slope, float inv_ns, double* __restrict__ dbeta,
                                                                      double* __restrict__ dgamma)
{
    __shared__ double red[256][8];
    const int LQ = C >> 2;
    const int RG = 256 / LQ;
    const int q = threadIdx.x % LQ, rg = threadIdx.x / LQ;
    float sc[4], sh[4], mu[4], is[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { sc[c] = scale[q * 4 + c]; sh[c] = shift[q * 4 + c]; mu[c] = mean[q * 4 + c]; is[c] = invstd[q * 4 + c]; }
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * RG + rg; i < M; i += (long long)gridDim.x * RG) {
        const float4 g4 = *reinterpret_cast<const float4*>(dOut + i * ldo + q * 4);
        const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + i * C + q * 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const int a[4] = {a4.x, a4.y, a4.z, a4.w};
        if (dDense) {
            for (int t = 0; t < k; ++t) {
                const float4 xv = ld4_bf16(X + (i * k + t) * C + q * 4);
                const float4 dv = ld4_bf16(dDense + (i * k + t) * C + q * 4);
                const float x[4] = {xv.x, xv.y, xv.z, xv.w};
                const float d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gy = d[c] + (a[c] == t ? g[c] : 0.0f);
                    // inv_ns > 0: X holds the POST-activation value y = act(pre) (LeakyReLU / identity, negative slope ns = 1 / inv_ns):
                    // pre = y or y / ns, xhat = (pre - beta) / gamma with `mean` = beta and `invstd` = 1 / gamma passed by the host
                    const float pre = inv_ns > 0.0f ? (x[c] > 0.0f ? x[c] : x[c] * inv_ns) : sc[c] * x[c] + sh[c];
                    const float xh = ((inv_ns > 0.0f ? pre : x[c]) - mu[c]) * is[c];
                    const float dpre = gy * act_grad2(pre, act, slope);
                    sb[c] += dpre;
                    sg[c] += (double)dpre * xh;
                }
            }
        } else {
            float xs[4] = {0.f, 0.f, 0.f, 0.f};
            if (Xsel) {                      // the selected values kept by the forward (lpd_group_sel_stats_bf16): no gather
                const float4 x4 = *reinterpret_cast<const float4*>(Xsel + i * ldsel + q * 4);
                xs[0] = x4.x; xs[1] = x4.y; xs[2] = x4.z; xs[3] = x4.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = Xsel ? xs[c] : bf2f(X[(i * k + a[c]) * C + q * 4 + c]);
                const float dpre = g[c] * act_grad2(sc[c] * x + sh[c], act, slope);
                sb[c] += dpre;
                sg[c] += (double)dpre * ((x - mu[c]) * is[c]);
            }
        }
    }
    reduce_quads<256>(red, sb, sg, LQ, dbeta, dgamma);
}

__global__ __launch_bounds__(256) void edge_bn_bwd_apply_bf16_kernel(const float* __restrict__ dOut, long long ldo,
                                                                     const uint8_t* __restrict__ arg, const uint16_t* __restrict__ dDense,
                                                                     const uint16_t* __restrict__ X, uint16_t* __restrict__ dX,
                                                                     float* __restrict__ dQ, long long ldq, int k, long long M, int C,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const double* __restrict__ dbeta, const double* __restrict__ dgamma,
                                                                     double count, int act, float slope, float inv_ns)
{
    const int LQ = C >> 2;
    const int RG = 256 / LQ;
    const int q = threadIdx.x % LQ, rg = threadIdx.x / LQ;
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
            const float4 xv = ld4_bf16(X + off);
            float d[4] = {0.f, 0.f, 0.f, 0.f};
            if (dDense) {
                const float4 dv = ld4_bf16(dDense + off);
                d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
            }
            const float x[4] = {xv.x, xv.y, xv.z, xv.w};
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gy = d[c] + (a[c] == t ? g[c] : 0.0f);
                const float pre = inv_ns > 0.0f ? (x[c] > 0.0f ? x[c] : x[c] * inv_ns) : sc[c] * x[c] + sh[c];      // see the reduce kernel
                const float xh = ((inv_ns > 0.0f ? pre : x[c]) - mu[c]) * is[c];
                const float dpre = gy * act_grad2(pre, act, slope);
                o[c] = sc[c] * (dpre - mb[c] - xh * mg[c]);
                sum[c] += o[c];      // the centre-term gradient sums the fp32 values, not the rounded ones
            }
            st4_bf16(dX + off, make_float4(o[0], o[1], o[2], o[3]));
        }
        if (dQ) *reinterpret_cast<float4*>(dQ + i * ldq + q * 4) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void gather_sum_rows_bf16_kernel(const uint16_t* __restrict__ dU, const int32_t* __restrict__ rowptr,
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
        for (; p + 3 < end; p += 4) {
            const int e0 = edges[p], e1 = edges[p + 1], e2 = edges[p + 2], e3 = edges[p + 3];
            const float4 a = ld4_bf16(dU + (long long)e0 * (LPR * 4) + cl * 4);
            const float4 b = ld4_bf16(dU + (long long)e1 * (LPR * 4) + cl * 4);
            const float4 c = ld4_bf16(dU + (long long)e2 * (LPR * 4) + cl * 4);
            const float4 d = ld4_bf16(dU + (long long)e3 * (LPR * 4) + cl * 4);
            acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
            acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; p < end; ++p) {
            const float4 a = ld4_bf16(dU + (long long)edges[p] * (LPR * 4) + cl * 4);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
        float4* dst = reinterpret_cast<float4*>(dP + j * ldp + cl * 4);
        if (accumulate) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *dst = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA products on the stored edge tensors
// ---------------------------------------------------------------------------------------------
// C[m][n] (bf16) = sum_k A[m][k] (bf16) * W(n, k)  with W split hi + lo (two bf16 MFMA products, fp32 accumulation).
// One wave owns 32 rows of A and all N <= 128 columns.  The MFMA computes the TRANSPOSED tile (rows = output channels,
// columns = data rows): a lane then holds, for ONE data row, 4 consecutive output channels per accumulator quad and
// stores them as 8 bytes; with its partner half-wave 16 contiguous bytes of the output row.  A row halves are loaded
// straight into registers: lane (row, h) owns the contiguous K/2 * 2 bytes [h * K/2, (h+1) * K/2) of its row, i.e. the
// contraction index is permuted (k = h * K/2 + 8 s + e), the same permutation is applied to the LDS images of W.
template <int K, int N>
__global__ __launch_bounds__(256) void gemm_bf16s_kernel(const uint16_t* __restrict__ A, const float* __restrict__ W, int ldw,
                                                         int b_kmajor, uint16_t* __restrict__ Cout, long long M)
{
    constexpr int KS = K / 16;            // MFMA k-steps
    constexpr int NT = N / 32;            // output-channel tiles
    constexpr int LDW = K + 8;            // bf16 per LDS row (16-byte pad: conflict-free ds_read_b128)
    extern __shared__ __attribute__((aligned(16))) __bf16 wimg[];   // hi image, then lo image: 2 * N * LDW bf16 (68 KiB at 128 x 128)
    __bf16* whi = wimg;
    __bf16* wlo = wimg + N * LDW;
    const int tid = threadIdx.x;
    for (int e = tid; e < N * K; e += 256) {
        const int n = e / K, kk = e % K;
        const float w = b_kmajor ? W[(size_t)kk * ldw + n] : W[(size_t)n * ldw + kk];
        const __bf16 hi = (__bf16)w;
        const __bf16 lo = (__bf16)(w - (float)hi);
        whi[n * LDW + kk] = hi;
        wlo[n * LDW + kk] = lo;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long ntile = (M + 31) / 32;
    const long long stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    bf16x8 a[KS], an[KS];
    auto load_rows = [&](long long t, bf16x8 (&dst)[KS]) {
        const long long row = t * 32 + col;
        const long long rr = row < M ? row : M - 1;
        const uint16_t* ap = A + rr * K + h * (K / 2);
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) dst[s2] = *reinterpret_cast<const bf16x8*>(ap + s2 * 8);
    };
    if (tile < ntile) load_rows(tile, a);
    for (; tile < ntile; tile += stride) {
        const long long row = tile * 32 + col;
        load_rows(tile + stride < ntile ? tile + stride : tile, an);     // the next tile's rows, in flight under this tile's MFMAs
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int off = (j * 32 + col) * LDW + h * (K / 2) + s2 * 8;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(whi + off);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wlo + off);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, a[s2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, a[s2], acc[j], 0, 0, 0);
            }
        }
        if (row < M) {
            uint16_t* cp = Cout + row * N;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // accumulator rows 8g + 4h + {0..3} = output channels j*32 + 8g + 4h + ...
                    const uint2 pk = make_uint2(pack_bf16(acc[j][4 * g], acc[j][4 * g + 1]), pack_bf16(acc[j][4 * g + 2], acc[j][4 * g + 3]));
                    *reinterpret_cast<uint2*>(cp + j * 32 + 8 * g + 4 * h) = pk;
                }
        }
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) a[s2] = an[s2];
    }
}

// dW[a][b] = sum_m A[m][a] * B[m][b]  (A [M][KA], B [M][KB] bf16 row-major; KA, KB in {64, 128}); fp32 slabs per block,
// summed by gemm_tn_reduce_kernel.  A block walks chunks of 64 rows; the MFMA operands are 8 consecutive m of ONE channel,
// so the chunk goes to LDS TRANSPOSED ([channel][row]): a thread loads an 8 (rows) x 8 (channels) patch as eight 16-byte row
// pieces, transposes it in registers (one v_perm_b32 per output dword) and writes eight 16-byte channel pieces; the next
// chunk's patches are requested before the MFMAs of the current one.  (First version: 16-bit scatter stores into the
// transposed image, 1.24 ms for the 3.6 M x 128 x 128 product; the traffic is 1.84 GB.)
template <int KA, int KB>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                           float* __restrict__ slabs, long long M, long long rows_per_block)
{
    constexpr int CH = 64;                 // rows per chunk
    constexpr int LDT = CH + 8;            // bf16 per transposed LDS row (144 B: 16-byte aligned pieces, conflict-free b128 reads)
    constexpr int TA = KA / 32, TB = KB / 32;
    constexpr int TILES = TA * TB;         // 32x32 output tiles; 4 waves share them
    constexpr int TPW = (TILES + 3) / 4;   // tiles per wave
    __shared__ __attribute__((aligned(16))) uint16_t at[KA * LDT];
    __shared__ __attribute__((aligned(16))) uint16_t bt[KB * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long m_begin = (long long)blockIdx.x * rows_per_block;
    const long long m_end = min(M, m_begin + rows_per_block);
    // patch of this thread: matrix (A for tid < KA, B for the next KB threads), channel group c8, row group rg
    const bool isA = tid < KA;
    const int pt = isA ? tid : tid - KA;
    const int ncg = (isA ? KA : KB) / 8;
    const int c8 = pt % ncg, rg = pt / ncg;                       // rg in 0..7 (KA, KB multiples of 64: ncg * 8 = K patches)
    const bool active = tid < KA + KB;
    const uint16_t* src = (isA ? A : B) + c8 * 8;
    const int ldsrc = isA ? KA : KB;
    // column blocks (8 rows of m = 16 bytes) are XOR-swizzled with the channel group: a wave's sixteen lanes with consecutive c8
    // write rows 8 * 144 bytes apart, i.e. onto only two of the sixteen 16-byte bank groups (8-way conflicts on every ds_write_b128
    // unswizzled); the readers apply the same swizzle.  What remains is 2-way on the writes AND on the operand reads
    // (SQ_LDS_BANK_CONFLICT = the conflict-free cycles again; reproduced by the bank model of MI355X_MICROARCH.md).  The
    // conflict-free alternative -- a lane group of 8 = the 8 ROW groups of one channel group, plain images -- was built in round 3
    // and reverted: its loads take 256-byte pieces of 8 rows per instruction instead of 512-byte pieces of 4, and every shape this
    // kernel still serves is bound by those loads (pooling 124 -> 136 us, DG1 weight gradient 78 -> 89, edge products 990 -> 1018).
    uint16_t* dst = (isA ? at : bt) + (c8 * 8) * LDT + ((rg ^ (c8 >> 1)) & 7) * 8;
    uint4 pr[8];
    auto load_patch = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long m = m0 + rg * 8 + i;
            pr[i] = (active && m < m_end) ? *reinterpret_cast<const uint4*>(src + m * ldsrc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_patch = [&]() {
        if (!active) return;
        const uint32_t w[8][4] = {{pr[0].x, pr[0].y, pr[0].z, pr[0].w}, {pr[1].x, pr[1].y, pr[1].z, pr[1].w},
                                  {pr[2].x, pr[2].y, pr[2].z, pr[2].w}, {pr[3].x, pr[3].y, pr[3].z, pr[3].w},
                                  {pr[4].x, pr[4].y, pr[4].z, pr[4].w}, {pr[5].x, pr[5].y, pr[5].z, pr[5].w},
                                  {pr[6].x, pr[6].y, pr[6].z, pr[6].w}, {pr[7].x, pr[7].y, pr[7].z, pr[7].w}};
#pragma unroll
        for (int p = 0; p < 4; ++p) {      // channel pair (2p, 2p+1) of the group
            uint4 lo, hi;                  // channel 2p / 2p+1: rows 0..7 as four dwords (row 2q | row 2q+1 << 16)
            lo.x = __builtin_amdgcn_perm(w[1][p], w[0][p], 0x05040100u); hi.x = __builtin_amdgcn_perm(w[1][p], w[0][p], 0x07060302u);
            lo.y = __builtin_amdgcn_perm(w[3][p], w[2][p], 0x05040100u); hi.y = __builtin_amdgcn_perm(w[3][p], w[2][p], 0x07060302u);
            lo.z = __builtin_amdgcn_perm(w[5][p], w[4][p], 0x05040100u); hi.z = __builtin_amdgcn_perm(w[5][p], w[4][p], 0x07060302u);
            lo.w = __builtin_amdgcn_perm(w[7][p], w[6][p], 0x05040100u); hi.w = __builtin_amdgcn_perm(w[7][p], w[6][p], 0x07060302u);
            *reinterpret_cast<uint4*>(dst + (2 * p) * LDT) = lo;
            *reinterpret_cast<uint4*>(dst + (2 * p + 1) * LDT) = hi;
        }
    };
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    load_patch(m_begin);
    for (long long m0 = m_begin; m0 < m_end; m0 += CH) {
        __syncthreads();               // the previous chunk's MFMA reads are done
        store_patch();
        __syncthreads();
        if (m0 + CH < m_end) load_patch(m0 + CH);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int tile = wave * TPW + j;
            if (tile < TILES) {
                const int ta = tile / TB, tb = tile % TB;
#pragma unroll
                for (int s = 0; s < CH / 16; ++s) {
                    const int ca = ta * 32 + col, cb = tb * 32 + col;
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(at + ca * LDT + (((s * 2 + h) ^ (ca >> 4)) & 7) * 8);
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bt + cb * LDT + (((s * 2 + h) ^ (cb >> 4)) & 7) * 8);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                }
            }
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * KA * KB;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = wave * TPW + j;
        if (tile < TILES) {
            const int ta = tile / TB, tb = tile % TB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)arow * KB + tb * 32 + col] = acc[j][r];
            }
        }
    }
}

// The same product for fp32 operands in split-bf16 form (hi + lo of both operands, three MFMA products per term: fp32-grade,
// lpd_gemm.hip "bf16x3"): dW[a][b] = sum_m A[m][a] * B[m][b], A [M][lda], B [M][ldb] fp32 row-major -- the weight gradients
// dW = dY^T X of the point layers and of the projections (reduction over B*N = 180 224 rows, or over the 3.6 M edges).
// The generic kernel stages k-major operands as 4 x 4 fp32 patches with 8-byte LDS writes and is bound by that staging
// (dW3 [1024 x 512 x 180224]: 1.7 ms with three products AND with one).  Here a block owns a 128 x KBT output tile and an m-range
// (split over grid.z); per 64-row chunk a thread loads an 8 x 8 fp32 patch (sixteen 16-byte loads, requested one chunk
// ahead), converts it to packed bf16 hi / lo pairs, transposes both in registers (v_perm_b32) and writes sixteen 16-byte
// channel pieces into four [channel][row] LDS images.
template <int KBT, bool ABF16 = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B,
                                                         long long ldb, float* __restrict__ slabs, long long M, int KA, int KB,
                                                         long long rows_per_split, int nsplit, long long sA, long long sB)
{
    // ABF16: A holds bf16 rows (lda / sA in bf16 elements): its patches ARE the hi image, there is no lo image and the a_lo b_hi product
    // is skipped (the bf16-storage training mode's conv3-map tensors: pooling, the assignment and conv3 weight gradients)
    constexpr int CH = 64, LDT = CH + 8;
    constexpr int TB = KBT / 32;               // output tiles along b; 4 along a
    constexpr int TPW = TB;                    // tiles per wave: wave w owns a-tile w, all b-tiles
    __shared__ __attribute__((aligned(16))) uint16_t ah[128 * LDT];
    __shared__ __attribute__((aligned(16))) uint16_t al[128 * LDT];
    __shared__ __attribute__((aligned(16))) uint16_t bh[KBT * LDT];
    __shared__ __attribute__((aligned(16))) uint16_t bl[KBT * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    // XCD-aware order: the output tiles of ONE m-range run next to each other on one XCD, so the rows they all read (A re-read by
    // every b-tile, B by every a-tile) come from that XCD's L2 once (in (x, y, z) grid order the 32 tiles of a range were dealt
    // to all eight XCDs: 5.9 GB through the fabric for the 1.1 GB dW3 operands, 1.12 ms)
    const int ntile_a = KA / 128, ntile_b = KB / KBT;
    const int lin = lpd_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lin % (ntile_a * ntile_b), zall = lin / (ntile_a * ntile_b);      // zall = batch * nsplit + split
    const int zsplit = zall % nsplit, zb = zall / nsplit;
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(A) + (long long)zb * sA;
    A += (long long)zb * sA;
    B += (long long)zb * sB;
    const int a0 = (tile % ntile_a) * 128, b0 = (tile / ntile_a) * KBT;
    const long long m_begin = (long long)zsplit * rows_per_split;
    const long long m_end = min(M, m_begin + rows_per_split);
    const bool isA = tid < 128;
    const int pt = isA ? tid : tid - 128;
    const int ncg = (isA ? 128 : KBT) / 8;
    const int c8 = pt % ncg, rg = pt / ncg;
    const bool active = isA || pt < KBT;       // KBT = 64: threads 192.. have no patch
    const float* src = isA ? A + a0 + c8 * 8 : B + b0 + c8 * 8;
    const long long ldsrc = isA ? lda : ldb;
    const int swz = ((rg ^ (c8 >> 1)) & 7) * 8;      // XOR-swizzled column block (see gemm_tn_bf16_kernel)
    uint16_t* dhi = (isA ? ah : bh) + (c8 * 8) * LDT + swz;
    uint16_t* dlo = (isA ? al : bl) + (c8 * 8) * LDT + swz;
    float4 pr[16];
    auto load_patch = [&](long long m0) {
        if (ABF16 && isA) {           // wave-uniform: eight bf16 values = one 16-byte piece per row, kept as raw bits
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long m = m0 + rg * 8 + i;
                pr[2 * i] = m < m_end ? *reinterpret_cast<const float4*>(A16 + a0 + c8 * 8 + m * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
                pr[2 * i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long m = m0 + rg * 8 + i;
            const bool ok = active && m < m_end;
            pr[2 * i] = ok ? *reinterpret_cast<const float4*>(src + m * ldsrc) : make_float4(0.f, 0.f, 0.f, 0.f);
            pr[2 * i + 1] = ok ? *reinterpret_cast<const float4*>(src + m * ldsrc + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&]() {
        if (!active) return;
        // one channel pair at a time (its eight rows converted, transposed, written): 16 temporaries live instead of 64 -- with all
        // four pairs converted first the <128> kernel needed 264 registers, i.e. ONE block per CU
        if (ABF16 && isA) {                    // wave-uniform: the rows are the hi image already (word p of a row = channels 2p, 2p+1)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t wh[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 q = pr[2 * i];
                    wh[i] = __float_as_uint(p == 0 ? q.x : p == 1 ? q.y : p == 2 ? q.z : q.w);
                }
                uint4 lo, hi;
                lo.x = __builtin_amdgcn_perm(wh[1], wh[0], 0x05040100u); hi.x = __builtin_amdgcn_perm(wh[1], wh[0], 0x07060302u);
                lo.y = __builtin_amdgcn_perm(wh[3], wh[2], 0x05040100u); hi.y = __builtin_amdgcn_perm(wh[3], wh[2], 0x07060302u);
                lo.z = __builtin_amdgcn_perm(wh[5], wh[4], 0x05040100u); hi.z = __builtin_amdgcn_perm(wh[5], wh[4], 0x07060302u);
                lo.w = __builtin_amdgcn_perm(wh[7], wh[6], 0x05040100u); hi.w = __builtin_amdgcn_perm(wh[7], wh[6], 0x07060302u);
                *reinterpret_cast<uint4*>(dhi + (2 * p) * LDT) = lo;
                *reinterpret_cast<uint4*>(dhi + (2 * p + 1) * LDT) = hi;
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t wh[8], wl[8];             // row i: packed bf16 (channel 2p | channel 2p+1 << 16)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 q = pr[2 * i + (p >> 1)];
                const float v0 = (p & 1) ? q.z : q.x, v1 = (p & 1) ? q.w : q.y;
                const uint32_t hp = pack_bf16(v0, v1);
                wh[i] = hp;
                wl[i] = pack_bf16(v0 - __uint_as_float(hp << 16), v1 - __uint_as_float(hp & 0xffff0000u));
            }
            uint4 lo, hi;
            lo.x = __builtin_amdgcn_perm(wh[1], wh[0], 0x05040100u); hi.x = __builtin_amdgcn_perm(wh[1], wh[0], 0x07060302u);
            lo.y = __builtin_amdgcn_perm(wh[3], wh[2], 0x05040100u); hi.y = __builtin_amdgcn_perm(wh[3], wh[2], 0x07060302u);
            lo.z = __builtin_amdgcn_perm(wh[5], wh[4], 0x05040100u); hi.z = __builtin_amdgcn_perm(wh[5], wh[4], 0x07060302u);
            lo.w = __builtin_amdgcn_perm(wh[7], wh[6], 0x05040100u); hi.w = __builtin_amdgcn_perm(wh[7], wh[6], 0x07060302u);
            *reinterpret_cast<uint4*>(dhi + (2 * p) * LDT) = lo;
            *reinterpret_cast<uint4*>(dhi + (2 * p + 1) * LDT) = hi;
            lo.x = __builtin_amdgcn_perm(wl[1], wl[0], 0x05040100u); hi.x = __builtin_amdgcn_perm(wl[1], wl[0], 0x07060302u);
            lo.y = __builtin_amdgcn_perm(wl[3], wl[2], 0x05040100u); hi.y = __builtin_amdgcn_perm(wl[3], wl[2], 0x07060302u);
            lo.z = __builtin_amdgcn_perm(wl[5], wl[4], 0x05040100u); hi.z = __builtin_amdgcn_perm(wl[5], wl[4], 0x07060302u);
            lo.w = __builtin_amdgcn_perm(wl[7], wl[6], 0x05040100u); hi.w = __builtin_amdgcn_perm(wl[7], wl[6], 0x07060302u);
            *reinterpret_cast<uint4*>(dlo + (2 * p) * LDT) = lo;
            *reinterpret_cast<uint4*>(dlo + (2 * p + 1) * LDT) = hi;
        }
    };
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    load_patch(m_begin);
    for (long long m0 = m_begin; m0 < m_end; m0 += CH) {
        __syncthreads();
        store_patch();
        __syncthreads();
        if (m0 + CH < m_end) load_patch(m0 + CH);
#pragma unroll
        for (int s = 0; s < CH / 16; ++s) {
            const int ca = wave * 32 + col;
            const int oa = ca * LDT + (((s * 2 + h) ^ (ca >> 4)) & 7) * 8;
            const bf16x8 avh = *reinterpret_cast<const bf16x8*>(ah + oa);
            const bf16x8 avl = *reinterpret_cast<const bf16x8*>(al + oa);
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int cb = j * 32 + col;
                const int ob = cb * LDT + (((s * 2 + h) ^ (cb >> 4)) & 7) * 8;
                const bf16x8 bvh = *reinterpret_cast<const bf16x8*>(bh + ob);
                const bf16x8 bvl = *reinterpret_cast<const bf16x8*>(bl + ob);
                if constexpr (!ABF16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl, bvh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, bvl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, bvh, acc[j], 0, 0, 0);
            }
        }
    }
    float* slab = slabs + (size_t)zall * KA * KB;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int arow = a0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[(size_t)arow * KB + b0 + j * 32 + col] = acc[j][r];
        }
}

// The wide weight gradient (conv3: dW3 = dZ^T X, 1024 x 512 over 180 224 rows) on 256 x 256 output tiles.  gemm_tn_x3_kernel
// above runs that shape at 26 % of the matrix peak: its 128 x 128 tiles move 5.9 GB through L2 for 1.1 GB of operands, a chunk's
// conversion (305 vector instructions per wave) and its 48 MFMAs alternate between two barriers, and the loads of the next chunk
// have only the MFMA phase to land in (PMC: 45 % of the wave cycles issue-stalled, 24 % parked).  Here: 512 threads, a wave owns
// 64 x 128 of the tile (2 x 4 accumulator tiles), 32-row chunks in TWO halves of the same [channel][row] images (channel rows
// of 144 bytes: 64 rows + 16 bytes of padding -- the operand reads are conflict-free on 36-dword strides, and a store group
// of 8 lanes = 4 row groups x 2 channel quads 576 bytes apart covers 128 contiguous bytes of banks), ONE barrier per chunk:
// while the MFMAs of chunk n read one half, the same waves convert chunk n+1 (requested a whole iteration earlier) and write
// it to the other half, and request chunk n+2.  All loads are unconditional (M % 32 == 0, ranges of whole chunks).
template <bool ABF16>
__global__ __launch_bounds__(512, 2) void gemm_tn_x3_256_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B,
                                                                long long ldb, float* __restrict__ slabs, int KA, int KB,
                                                                long long rows_per_split, long long M)
{
    // ABF16: A holds bf16 rows (lda in bf16 elements): hi image only, the a_lo b_hi product skipped (cf. gemm_tn_x3_kernel)
    constexpr int LDB = 144;                       // bytes per channel row of an image
    extern __shared__ __attribute__((aligned(16))) unsigned char tn3_lds[];
    unsigned char* const img_hi = tn3_lds;                  // [512 channels: A 0..255, B 256..511][144]
    unsigned char* const img_lo = tn3_lds + 512 * LDB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int ntile_a = KA / 256, ntile_b = KB / 256;
    const int lin = lpd_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lin % (ntile_a * ntile_b), split = lin / (ntile_a * ntile_b);
    const int a0 = (tile % ntile_a) * 256, b0 = (tile / ntile_a) * 256;
    const long long m_begin = (long long)split * rows_per_split;
    const long long m_end = min(M, m_begin + rows_per_split);
    const int nchunk = (int)((m_end - m_begin) / 32);
    // this thread's patch: 8 rows (row group rg) x 4 channels (quad); waves 0..3 stage A, waves 4..7 stage B
    const int rg = tid & 3, quad = tid >> 2;
    const bool isA = quad < 64;
    const bool raw = ABF16 && isA;                 // wave-uniform (waves 0..3 stage A)
    const float* src = isA ? A + a0 + quad * 4 : B + b0 + (quad - 64) * 4;
    const long long ldsrc = isA ? lda : ldb;
    src += (m_begin + rg * 8) * ldsrc;
    const uint16_t* src16 = reinterpret_cast<const uint16_t*>(A) + a0 + quad * 4 + (m_begin + rg * 8) * lda;
    const long long chunk_step = 32 * ldsrc;
    const int wr_off = (quad * 4) * LDB + rg * 16;             // + 64 * half + channel * 144
    const int wa = wave & 3, wb = wave >> 2;                  // wave's tiles: A channels 64 wa .. +63, B channels 128 wb .. +127
    const int rd_a = (wa * 64 + col) * LDB + h * 16;          // + 64 * half + 32 * kstep + 32 * 144 * tile
    const int rd_b = (256 + wb * 128 + col) * LDB + h * 16;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 r0[8], r1[8];
    auto load = [&](float4 (&r)[8], int chunk) {
        if (ABF16 && raw) {                        // four bf16 values per row piece, kept as raw bits in .x / .y
            const uint16_t* p16 = src16 + (long long)chunk * 32 * lda;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint2 w = *reinterpret_cast<const uint2*>(p16 + i * lda);
                r[i] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), 0.0f, 0.0f);
            }
        } else {
            const float* p = src + (long long)chunk * chunk_step;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const float4*>(p + i * ldsrc);
        }
    };
    // conversion of a patch in eight pieces (channel pair p = piece / 4: rows 0..3, rows 4..7, the two hi stores, the two lo stores)
    uint32_t wh[8], wl[8];
    auto conv_piece = [&](const float4 (&r)[8], int half, int piece) {
        const int p = piece >> 2, sub = piece & 3;
        if (sub < 2) {
#pragma unroll
            for (int i = 4 * sub; i < 4 * sub + 4; ++i) {
                const float v0 = p ? r[i].z : r[i].x, v1 = p ? r[i].w : r[i].y;
                const uint32_t hp = pack_bf16(v0, v1);
                wh[i] = (ABF16 && raw) ? __float_as_uint(p ? r[i].y : r[i].x) : hp;
                wl[i] = pack_bf16(v0 - __uint_as_float(hp << 16), v1 - __uint_as_float(hp & 0xffff0000u));
            }
        } else {
            if (ABF16 && raw && sub == 3) return;  // no lo image
            const uint32_t* w = sub == 2 ? wh : wl;
            uint4 e, o;                            // even / odd channel of the pair: rows 0..7 as four dwords
            e.x = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u); o.x = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u);
            e.y = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u); o.y = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
            e.z = __builtin_amdgcn_perm(w[5], w[4], 0x05040100u); o.z = __builtin_amdgcn_perm(w[5], w[4], 0x07060302u);
            e.w = __builtin_amdgcn_perm(w[7], w[6], 0x05040100u); o.w = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u);
            unsigned char* img = sub == 2 ? img_hi : img_lo;
            const int off = wr_off + half * 64 + (2 * p) * LDB;
            *reinterpret_cast<uint4*>(img + off) = e;
            *reinterpret_cast<uint4*>(img + off + LDB) = o;
        }
    };
    bf16x8 ah[2], al[2], bh[2], bl[2];
    auto read_a = [&](int half, int ks) {
        const int oa = rd_a + half * 64 + ks * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(img_hi + oa + i * 32 * LDB);
            al[i] = *reinterpret_cast<const bf16x8*>(img_lo + oa + i * 32 * LDB);
        }
    };
    auto read_b = [&](int slot, int half, int ks, int j) {
        const int ob = rd_b + half * 64 + ks * 32 + j * 32 * LDB;
        bh[slot] = *reinterpret_cast<const bf16x8*>(img_hi + ob);
        bl[slot] = *reinterpret_cast<const bf16x8*>(img_lo + ob);
    };
    // one chunk: eight blocks of [operand reads of the NEXT block | 6 MFMAs | one piece of the next chunk's conversion].  (Pinning
    // the blocks with sched_barrier(0), or leaving the conversion in one lump between the two k-steps, times the same: 725..730 us
    // for dW3 against 993 us for the 128-wide kernel on the same box; PMC: no LDS conflicts, MFMA-busy 0.37..0.40.)
    auto chunk = [&](int n, const float4 (&cur)[8], float4 (&nxt)[8]) {
        const int half = n & 1;
        load(nxt, n + 2 < nchunk ? n + 2 : nchunk - 1);        // unconditional: the last chunks are re-read and dropped
        read_a(half, 0);
        read_b(0, half, 0, 0);
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int j = step & 3, sl = step & 1;
            if (step < 7) read_b(sl ^ 1, half, (step + 1) >> 2, (step + 1) & 3);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (!ABF16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[sl], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[sl], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[sl], acc[i][j], 0, 0, 0);
            }
            if (step == 3) read_a(half, 1);
            conv_piece(cur, half ^ 1, step);       // on the last chunk: a re-read chunk into the half nobody reads again
        }
        __syncthreads();
    };

    if (nchunk > 0) {          // (a split past the end of the rows writes a zero slab)
        load(r0, 0);
#pragma unroll
        for (int piece = 0; piece < 8; ++piece) conv_piece(r0, 0, piece);
        load(r0, nchunk > 1 ? 1 : 0);
        __syncthreads();
        for (int n = 0; n < nchunk; n += 2) {
            chunk(n, r0, r1);
            if (n + 1 < nchunk) chunk(n + 1, r1, r0);
        }
    }
    float* slab = slabs + (size_t)split * KA * KB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = a0 + wa * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)arow * KB + b0 + wb * 128 + j * 32 + col] = acc[i][j][r];
            }
}

// the same sum with 16 slab groups per block (16 threads x 4 elements each; n % 64 == 0): with many slabs of a small product (256 x 64 over
// 256 row ranges) one thread per element walked the 256 slabs one dependent load at a time
__global__ __launch_bounds__(256) void gemm_tn_reduce16_kernel(const float* __restrict__ slabs, float* __restrict__ out, int n, int nslabs)
{
    __shared__ double red[16][16][4];
    const int t = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = (blockIdx.x * 16 + t) * 4;
    slabs += (size_t)blockIdx.y * nslabs * n;
    double s[4] = {0, 0, 0, 0};
    for (int b = grp; b < nslabs; b += 64) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = b + 16 * u < nslabs ? *reinterpret_cast<const float4*>(slabs + (size_t)(b + 16 * u) * n + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) red[grp][t][c] = s[c];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int tt = threadIdx.x >> 2, c = threadIdx.x & 3;
        double v = 0.0;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) v += red[g2][tt][c];
        out[(size_t)blockIdx.y * n + (blockIdx.x * 16 + tt) * 4 + c] = (float)v;
    }
}

__global__ void gemm_tn_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int n, int nslabs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // blockIdx.y = problem of a batch (its slabs are consecutive)
    if (e >= n) return;
    slabs += (size_t)blockIdx.y * nslabs * n;
    double s = 0.0;
    for (int b = 0; b < nslabs; ++b) s += slabs[(size_t)b * n + e];
    out[(size_t)blockIdx.y * n + e] = (float)s;
}

inline int grid_for(long long items, int per_block, int cap = 4096)
{
    long long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}


// Dense counterpart of edge_split_bwd_apply_kernel: the gradient G [E][C] in front of a train-mode BatchNorm over the edges
// U[(i,t)] = P[nbr(i,t)] + Q[i] is DENSE (the DG1 stage: every post-activation edge feeds convDG2).  With m1 = mean(G), m2 = mean(G xhat),
//   dU[(i,t)] = s (G[(i,t)] - m1 - xhat[(i,t)] m2),   xhat = (P_j + Q_i - mu) invstd,
// summed in closed form over the incoming edges of row j (transposed graph) and over the k outgoing edges of row j:
//   dP_j = s (A_j - deg_j m1 - m2 invstd (deg_j (P_j - mu) + R_j)),      A_j = sum G[e], R_j = sum Q_i over the incoming edges e = (i, t)
//   dQ_j = s (gsum_j - k m1 - m2 invstd (S_j + k (Q_j - mu))),           gsum_j = sum_t G[(j,t)] (lpd_edge_mlp_train_bwd), S_j = sum_t P[nbr(j,t)]
// -- one gather pass (a G row and a Q row per edge), no dU tensor, no float atomics.  GT: uint16_t (bf16 rows) or float.
template <int LPR, typename GT>
__global__ __launch_bounds__(256) void edge_dense_bwd_apply_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ edges,
                                                                   const GT* __restrict__ G, const float* __restrict__ gsum,
                                                                   const float* __restrict__ S, const float* __restrict__ P,
                                                                   long long ldp, const float* __restrict__ Q, long long ldq,
                                                                   float* __restrict__ dP, long long lddp, float* __restrict__ dQ,
                                                                   long long lddq, long long M, int k,
                                                                   const float* __restrict__ scale, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const double* __restrict__ dbeta,
                                                                   const double* __restrict__ dgamma)
{
    constexpr int RPW = 64 / LPR;
    constexpr int C = LPR * 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, cl = lane % LPR;
    const LpdXcdSweep sweep = lpd_xcd_sweep((M + RPW - 1) / RPW);
    const double E = (double)M * (double)k;
    float sc[4], mu[4], m1[4], m2i[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = cl * 4 + c;
        sc[c] = scale[ch]; mu[c] = mean[ch];
        m1[c] = (float)(dbeta[ch] / E);
        m2i[c] = (float)(dgamma[ch] / E) * invstd[ch];
    }
    const float kf = (float)k;
    const unsigned ku = (unsigned)k;
    auto ldg = [&](unsigned e) -> float4 {
        if constexpr (sizeof(GT) == 2) return ld4_bf16(reinterpret_cast<const uint16_t*>(G) + (long long)e * C + cl * 4);
        else return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(G) + (long long)e * C + cl * 4);
    };
    for (long long wi = sweep.begin; wi < sweep.end; wi += sweep.step) {
        const long long j = wi * RPW + sub;
        if (j >= M) continue;
        const int beg = rowptr[j], end = rowptr[j + 1];
        float A[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
        int p = beg;
        for (; p + 3 < end; p += 4) {          // four incoming edges (eight row loads) in flight
            unsigned ee[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ee[u] = (unsigned)edges[p + u];
            float4 q4[4], g4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q4[u] = *reinterpret_cast<const float4*>(Q + (long long)(ee[u] / ku) * ldq + cl * 4);
                g4[u] = ldg(ee[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // same order of additions as the one-edge loop
                R[0] += q4[u].x; R[1] += q4[u].y; R[2] += q4[u].z; R[3] += q4[u].w;
                A[0] += g4[u].x; A[1] += g4[u].y; A[2] += g4[u].z; A[3] += g4[u].w;
            }
        }
        for (; p < end; ++p) {
            const unsigned e = (unsigned)edges[p];
            const float4 q4 = *reinterpret_cast<const float4*>(Q + (long long)(e / ku) * ldq + cl * 4);
            const float4 g4 = ldg(e);
            R[0] += q4.x; R[1] += q4.y; R[2] += q4.z; R[3] += q4.w;
            A[0] += g4.x; A[1] += g4.y; A[2] += g4.z; A[3] += g4.w;
        }
        const float deg = (float)(end - beg);
        const float4 p4 = *reinterpret_cast<const float4*>(P + j * ldp + cl * 4);
        const float4 q4 = *reinterpret_cast<const float4*>(Q + j * ldq + cl * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(gsum + j * C + cl * 4);
        const float4 s4 = *reinterpret_cast<const float4*>(S + j * C + cl * 4);
        const float pj[4] = {p4.x, p4.y, p4.z, p4.w}, qj[4] = {q4.x, q4.y, q4.z, q4.w};
        const float gj[4] = {g4.x, g4.y, g4.z, g4.w}, sj[4] = {s4.x, s4.y, s4.z, s4.w};
        float op[4], oq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            op[c] = sc[c] * (A[c] - deg * m1[c] - m2i[c] * (deg * (pj[c] - mu[c]) + R[c]));
            oq[c] = sc[c] * (gj[c] - kf * m1[c] - m2i[c] * (sj[c] + kf * (qj[c] - mu[c])));
        }
        *reinterpret_cast<float4*>(dP + j * lddp + cl * 4) = make_float4(op[0], op[1], op[2], op[3]);
        *reinterpret_cast<float4*>(dQ + j * lddq + cl * 4) = make_float4(oq[0], oq[1], oq[2], oq[3]);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int lpd_edge_split_fwd(const float* P, long long ldp, const float* Q, long long ldq, const int32_t* idx,
                                  const float* gamma, float* S, float* usel, uint8_t* arg, long long M, int N, int C, int k,
                                  double* sum, double* sumsq, double* stat_ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && Q && idx && gamma && S && usel && arg && sum && sumsq, "lpd_edge_split_fwd: null pointer");
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_edge_split_fwd: C=%d unsupported", C);
    LPD_CHECK_ARG(M > 0 && N > 0 && k > 0 && k <= 255 && M % N == 0 && ldp % 4 == 0 && ldq % 4 == 0, "lpd_edge_split_fwd: bad dims");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_split_fwd: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_split_fwd", C);
    const int lpp = C / 4;
    const int grid = grid_for(M, 4 * (64 / lpp) * 4, 2048);
    if (C == 64) hipLaunchKernelGGL(edge_split_fwd_kernel<16>, dim3(grid), dim3(256), 0, stream, P, ldp, Q, ldq, idx, gamma, S, usel, arg, M, N, k, ws.sum(), ws.sumsq());
    else if (C == 128) hipLaunchKernelGGL(edge_split_fwd_kernel<32>, dim3(grid), dim3(256), 0, stream, P, ldp, Q, ldq, idx, gamma, S, usel, arg, M, N, k, ws.sum(), ws.sumsq());
    else hipLaunchKernelGGL(edge_split_fwd_kernel<64>, dim3(grid), dim3(256), 0, stream, P, ldp, Q, ldq, idx, gamma, S, usel, arg, M, N, k, ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_edge_split_fwd");
    return lpd_stat_finish(ws, sum, sumsq, C, stream);
}

extern "C" int lpd_edge_split_bwd(const float* dOut, long long ldo, const float* usel, const uint8_t* arg, const float* S,
                                  const float* P, long long ldp, const float* Q, long long ldq, const int32_t* rowptr,
                                  const int32_t* edges, float* G, float* dP, long long lddp, float* dQ, long long lddq, long long M,
                                  int C, int k, const float* scale, const float* shift, const float* mean, const float* invstd,
                                  int act, float slope, int half, double* dbeta, double* dgamma, double* stat_ws, void* stream_)
{
    // half != 0 (bf16 storage): the workspace G holds bf16 rows of G and of Q ([2][M][C] bf16 = the same M C floats); C = 256 only
    // half & 2: dP / dQ are bf16 rows too (lddp / lddq in bf16 elements)
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(dOut && usel && arg && S && P && Q && rowptr && edges && G && dP && dQ && dbeta && dgamma,
                  "lpd_edge_split_bwd: null pointer");
    LPD_CHECK_ARG(!half || C == 256, "lpd_edge_split_bwd: the bf16 form is built for C = 256");
    LPD_CHECK_ARG(!(half & 2) || (half & 1), "lpd_edge_split_bwd: bf16 outputs go with the bf16 gather rows");
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_edge_split_bwd: C=%d unsupported", C);
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_split_bwd: activation %d unsupported", act);
    LPD_CHECK_ARG(ldo % 4 == 0 && ldp % 4 == 0 && ldq % 4 == 0 && lddp % 4 == 0 && lddq % 4 == 0, "lpd_edge_split_bwd: leading dims % 4");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_split_bwd: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_split_bwd", C);
    const int rg = 256 / (C / 4);
    if (half) hipLaunchKernelGGL(edge_split_bwd_reduce_kernel<true>, dim3(lpd_reduce_grid(grid_for(M, rg * 8, 2048))), dim3(256), 0, stream, dOut, ldo, usel, G, Q, ldq, M, C,
                                 scale, shift, mean, invstd, act, slope, ws.sum(), ws.sumsq());
    else hipLaunchKernelGGL(edge_split_bwd_reduce_kernel<false>, dim3(lpd_reduce_grid(grid_for(M, rg * 8, 2048))), dim3(256), 0, stream, dOut, ldo, usel, G, Q, ldq, M, C,
                            scale, shift, mean, invstd, act, slope, ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_edge_split_bwd(reduce)");
    if (int rc = lpd_stat_finish(ws, dbeta, dgamma, C, stream)) return rc;
    const int lpr = C / 4;
    const int grid = grid_for(M, 4 * (64 / lpr) * 2, 8192);
    if (half & 2) hipLaunchKernelGGL((edge_split_bwd_apply_kernel<64, true, true>), dim3(grid), dim3(256), 0, stream, rowptr, edges, (const float*)G, arg, S, P, ldp, Q, ldq, dP, lddp, dQ, lddq, M, k, scale, mean, invstd, (const double*)dbeta, (const double*)dgamma);
    else if (half) hipLaunchKernelGGL((edge_split_bwd_apply_kernel<64, true>), dim3(grid), dim3(256), 0, stream, rowptr, edges, (const float*)G, arg, S, P, ldp, Q, ldq, dP, lddp, dQ, lddq, M, k, scale, mean, invstd, (const double*)dbeta, (const double*)dgamma);
    else if (C == 64) hipLaunchKernelGGL(edge_split_bwd_apply_kernel<16>, dim3(grid), dim3(256), 0, stream, rowptr, edges, (const float*)G, arg, S, P, ldp, Q, ldq, dP, lddp, dQ, lddq, M, k, scale, mean, invstd, (const double*)dbeta, (const double*)dgamma);
    else if (C == 128) hipLaunchKernelGGL(edge_split_bwd_apply_kernel<32>, dim3(grid), dim3(256), 0, stream, rowptr, edges, (const float*)G, arg, S, P, ldp, Q, ldq, dP, lddp, dQ, lddq, M, k, scale, mean, invstd, (const double*)dbeta, (const double*)dgamma);
    else hipLaunchKernelGGL(edge_split_bwd_apply_kernel<64>, dim3(grid), dim3(256), 0, stream, rowptr, edges, (const float*)G, arg, S, P, ldp, Q, ldq, dP, lddp, dQ, lddq, M, k, scale, mean, invstd, (const double*)dbeta, (const double*)dgamma);
    LPD_CHECK_LAUNCH("lpd_edge_split_bwd(apply)");
    return LPD_OK;
}

// Closed-form backward of a train-mode BatchNorm over the edges with a DENSE incoming gradient (edge_dense_bwd_apply_kernel): dP, dQ
// from G [M k][C] (bf16 != 0: bf16 rows), gsum [M][C] = sum_t G, S [M][C] = sum_t P[nbr] (lpd_edge_split_fwd), the transposed graph
// (lpd_graph_transpose) and the sums dbeta = sum G, dgamma = sum G xhat (lpd_edge_mlp_train_bwd).  C in {64, 128, 256}.
extern "C" int lpd_edge_dense_bwd_apply(const void* G, int bf16, const float* gsum, const float* S, const float* P, long long ldp, const float* Q,
                                        long long ldq, const int32_t* rowptr, const int32_t* edges, float* dP, long long lddp, float* dQ,
                                        long long lddq, long long M, int C, int k, const float* scale, const float* mean, const float* invstd,
                                        const double* dbeta, const double* dgamma, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(G && gsum && S && P && Q && rowptr && edges && dP && dQ && scale && mean && invstd && dbeta && dgamma,
                  "lpd_edge_dense_bwd_apply: null pointer");
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_edge_dense_bwd_apply: C=%d unsupported", C);
    LPD_CHECK_ARG(M > 0 && k > 0 && ldp % 4 == 0 && ldq % 4 == 0 && lddp % 4 == 0 && lddq % 4 == 0, "lpd_edge_dense_bwd_apply: bad dims");
    const int lpr = C / 4;
    const int grid = grid_for(M, 4 * (64 / lpr) * 2, 8192);
#define LPD_DENSE_APPLY(LPR, T)                                                                                                      \
    hipLaunchKernelGGL((edge_dense_bwd_apply_kernel<LPR, T>), dim3(grid), dim3(256), 0, stream, rowptr, edges, (const T*)G, gsum, S, P, ldp, \
                       Q, ldq, dP, lddp, dQ, lddq, M, k, scale, mean, invstd, dbeta, dgamma)
    if (bf16) {
        if (C == 64) LPD_DENSE_APPLY(16, uint16_t); else if (C == 128) LPD_DENSE_APPLY(32, uint16_t); else LPD_DENSE_APPLY(64, uint16_t);
    } else {
        if (C == 64) LPD_DENSE_APPLY(16, float); else if (C == 128) LPD_DENSE_APPLY(32, float); else LPD_DENSE_APPLY(64, float);
    }
#undef LPD_DENSE_APPLY
    LPD_CHECK_LAUNCH("lpd_edge_dense_bwd_apply");
    return LPD_OK;
}

extern "C" int lpd_edge_build_bf16(const float* P, long long ldp, const float* Q, long long ldq, const int32_t* idx, uint16_t* U,
                                   long long M, int N, int C, int k, double* sum, double* sumsq, double* stat_ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx && U, "lpd_edge_build_bf16: null pointer");
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_edge_build_bf16: C=%d unsupported", C);
    LPD_CHECK_ARG((sum == nullptr) == (sumsq == nullptr), "lpd_edge_build_bf16: sum and sumsq come together");
    LpdStatWs ws = {nullptr};
    double* usum = sum;
    double* usumsq = sumsq;
    if (sum) {
        ws = lpd_stat_arg(stat_ws);
        LPD_CHECK_ARG(ws.rep, "lpd_edge_build_bf16: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
        LPD_CHECK_STAT_COLS("lpd_edge_build_bf16", C);
        sum = ws.sum();
        sumsq = ws.sumsq();
    }
    const int lpp = C / 4;
    const int grid = grid_for(M, 4 * (64 / lpp) * 4, 2048);
    if (C == 64) hipLaunchKernelGGL(edge_build_bf16_kernel<16>, dim3(grid), dim3(256), 0, stream, P, ldp, Q, ldq, idx, U, M, N, k, sum, sumsq);
    else if (C == 128) hipLaunchKernelGGL(edge_build_bf16_kernel<32>, dim3(grid), dim3(256), 0, stream, P, ldp, Q, ldq, idx, U, M, N, k, sum, sumsq);
    else hipLaunchKernelGGL(edge_build_bf16_kernel<64>, dim3(grid), dim3(256), 0, stream, P, ldp, Q, ldq, idx, U, M, N, k, sum, sumsq);
    LPD_CHECK_LAUNCH("lpd_edge_build_bf16");
    if (usum) return lpd_stat_finish(ws, usum, usumsq, C, stream);
    return LPD_OK;
}

extern "C" int lpd_edge_act_max_bf16(const uint16_t* U, int k, const float* scale, const float* shift, int act, float slope,
                                     uint16_t* Y, float* out, long long ldo, uint8_t* arg, long long M, int C, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(U && scale && shift && Y && out && arg, "lpd_edge_act_max_bf16: null pointer");
    LPD_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && k <= 255 && ldo % 4 == 0, "lpd_edge_act_max_bf16: bad dims");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_act_max_bf16: activation %d unsupported", act);
    const int rg = 256 / (C / 4);
    hipLaunchKernelGGL(edge_act_max_bf16_kernel, dim3(grid_for(M, rg, 8192)), dim3(256), 0, stream, U, k, scale, shift, act, slope, Y, out,
                       ldo, arg, M, C);
    LPD_CHECK_LAUNCH("lpd_edge_act_max_bf16");
    return LPD_OK;
}

// fp32 storage: Y [M*k][C] = act(scale * U + shift), out [M][ldo] = its max over the k rows of a point, arg [M][C] the slot -- one pass
// over U (util/lpdnet_model.py:249-250: convDG1's BatchNorm + activation and x1 = max over k)
extern "C" int lpd_edge_act_max(const float* U, int k, const float* scale, const float* shift, int act, float slope, float* Y, float* out,
                                long long ldo, uint8_t* arg, long long M, int C, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(U && scale && shift && Y && out && arg, "lpd_edge_act_max: null pointer");
    LPD_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && k <= 255 && ldo % 4 == 0, "lpd_edge_act_max: bad dims");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_act_max: activation %d unsupported", act);
    const int rg = 256 / (C / 4);
    hipLaunchKernelGGL(edge_act_max_f32_kernel, dim3(grid_for(M, rg, 8192)), dim3(256), 0, stream, U, k, scale, shift, act, slope, Y, out,
                       ldo, arg, M, C);
    LPD_CHECK_LAUNCH("lpd_edge_act_max");
    return LPD_OK;
}

extern "C" int lpd_group_sel_stats_bf16(const uint16_t* Z, int k, const float* gamma, float* sel, long long lds, uint8_t* arg,
                                        long long M, int C, double* sum, double* sumsq, double* stat_ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(Z && gamma && sel && arg && sum && sumsq, "lpd_group_sel_stats_bf16: null pointer");
    LPD_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && k <= 255 && lds % 4 == 0, "lpd_group_sel_stats_bf16: bad dims");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_group_sel_stats_bf16: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_group_sel_stats_bf16", C);
    const int rg = 256 / (C / 4);
    hipLaunchKernelGGL(group_sel_stats_bf16_kernel, dim3(grid_for(M, rg * 4, 4096)), dim3(256), 0, stream, Z, k, gamma, sel, lds, arg, M, C,
                       ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_group_sel_stats_bf16");
    return lpd_stat_finish(ws, sum, sumsq, C, stream);
}

static int edge_bn_bwd_bf16_impl(const float* dOut, long long ldo, const uint8_t* arg, const uint16_t* dDense, const uint16_t* X,
                                 const float* Xsel, long long ldsel, uint16_t* dX, float* dQ, long long ldq, int k, long long M, int C,
                                 const float* scale, const float* shift, const float* mean, const float* invstd, int act, float slope,
                                 float inv_ns, double* dbeta, double* dgamma, double* stat_ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(inv_ns == 0.0f || (dDense && (act == 0 || act == 2) && inv_ns >= 1.0f),
                  "lpd_edge_bn_bwd_bf16: post-activation X needs the dense form and an invertible activation (none / LeakyReLU)");
    LPD_CHECK_ARG(dOut && arg && X && dX && scale && shift && mean && invstd && dbeta && dgamma, "lpd_edge_bn_bwd_bf16: null pointer");
    LPD_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && k <= 255 && ldo % 4 == 0 && ldq % 4 == 0 && ldsel % 4 == 0,
                  "lpd_edge_bn_bwd_bf16: bad dims");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_bn_bwd_bf16: activation %d unsupported", act);
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_bn_bwd_bf16: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_bn_bwd_bf16", C);
    const int rg = 256 / (C / 4);
    hipLaunchKernelGGL(edge_bn_bwd_reduce_bf16_kernel, dim3(grid_for(M, rg * 4, 4096)), dim3(256), 0, stream, dOut, ldo, arg, dDense, X, Xsel,
                       ldsel, k, M, C, scale, shift, mean, invstd, act, slope, inv_ns, ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_edge_bn_bwd_bf16(reduce)");
    if (int rc = lpd_stat_finish(ws, dbeta, dgamma, C, stream)) return rc;
    hipLaunchKernelGGL(edge_bn_bwd_apply_bf16_kernel, dim3(grid_for(M, rg, 8192)), dim3(256), 0, stream, dOut, ldo, arg, dDense, X, dX, dQ,
                       ldq, k, M, C, scale, shift, mean, invstd, (const double*)dbeta, (const double*)dgamma, (double)M * (double)k, act,
                       slope, inv_ns);
    LPD_CHECK_LAUNCH("lpd_edge_bn_bwd_bf16(apply)");
    return LPD_OK;
}

extern "C" int lpd_edge_bn_bwd_bf16(const float* dOut, long long ldo, const uint8_t* arg, const uint16_t* dDense, const uint16_t* X,
                                    uint16_t* dX, float* dQ, long long ldq, int k, long long M, int C, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, int act, float slope, float inv_ns,
                                    double* dbeta, double* dgamma, double* stat_ws, void* stream)
{
    return edge_bn_bwd_bf16_impl(dOut, ldo, arg, dDense, X, nullptr, 0, dX, dQ, ldq, k, M, C, scale, shift, mean, invstd, act, slope, inv_ns,
                                 dbeta, dgamma, stat_ws, stream);
}

extern "C" int lpd_edge_bn_bwd_bf16_sel(const float* dOut, long long ldo, const uint8_t* arg, const uint16_t* X, const float* Xsel,
                                        long long ldsel, uint16_t* dX, int k, long long M, int C, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, int act, float slope, double* dbeta, double* dgamma,
                                        double* stat_ws, void* stream)
{
    LPD_CHECK_ARG(Xsel, "lpd_edge_bn_bwd_bf16_sel: Xsel is null");
    return edge_bn_bwd_bf16_impl(dOut, ldo, arg, nullptr, X, Xsel, ldsel, dX, nullptr, 0, k, M, C, scale, shift, mean, invstd, act, slope,
                                 0.0f, dbeta, dgamma, stat_ws, stream);
}

extern "C" int lpd_gather_sum_rows_bf16(const uint16_t* dU, const int32_t* rowptr, const int32_t* edges, float* dP, long long ldp,
                                        long long M, int C, int accumulate, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(dU && rowptr && edges && dP, "lpd_gather_sum_rows_bf16: null pointer");
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_gather_sum_rows_bf16: C=%d unsupported", C);
    const int lpr = C / 4;
    const int grid = grid_for(M, 4 * (64 / lpr) * 2, 8192);
    if (C == 64) hipLaunchKernelGGL(gather_sum_rows_bf16_kernel<16>, dim3(grid), dim3(256), 0, stream, dU, rowptr, edges, dP, ldp, M, accumulate);
    else if (C == 128) hipLaunchKernelGGL(gather_sum_rows_bf16_kernel<32>, dim3(grid), dim3(256), 0, stream, dU, rowptr, edges, dP, ldp, M, accumulate);
    else hipLaunchKernelGGL(gather_sum_rows_bf16_kernel<64>, dim3(grid), dim3(256), 0, stream, dU, rowptr, edges, dP, ldp, M, accumulate);
    LPD_CHECK_LAUNCH("lpd_gather_sum_rows_bf16");
    return LPD_OK;
}

extern "C" int lpd_gemm_bf16s(const uint16_t* A, const float* W, int ldw, int b_kmajor, uint16_t* C, long long M, int N, int K,
                              void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(A && W && C && M > 0, "lpd_gemm_bf16s: null pointer");
    LPD_CHECK_ARG((N == 128 && K == 128) || (N == 64 && K == 64), "lpd_gemm_bf16s: built for (N, K) = (128, 128) and (64, 64), got (%d, %d)", N, K);
    const int grid = grid_for((M + 31) / 32, 4 * 8, 2048);
    const size_t lds = (size_t)2 * N * (K + 8) * sizeof(uint16_t);
    if (N == 128) {
        auto kern = gemm_bf16s_kernel<128, 128>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, A, W, ldw, b_kmajor, C, M);
    } else {
        auto kern = gemm_bf16s_kernel<64, 64>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, A, W, ldw, b_kmajor, C, M);
    }
    LPD_CHECK_LAUNCH("lpd_gemm_bf16s");
    return LPD_OK;
}

extern "C" long long lpd_gemm_tn_bf16_ws_floats(long long M, int KA, int KB)
{
    const long long blocks = M / 2048 > 1024 ? 1024 : (M / 2048 < 1 ? 1 : M / 2048);
    return blocks * KA * KB;
}

extern "C" int lpd_gemm_tn_bf16(const uint16_t* A, const uint16_t* B, float* dW, float* ws, long long M, int KA, int KB, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(A && B && dW && ws && M > 0, "lpd_gemm_tn_bf16: null pointer");
    LPD_CHECK_ARG((KA == 128 && KB == 128) || (KA == 64 && KB == 64), "lpd_gemm_tn_bf16: built for 128 x 128 and 64 x 64");
    const long long blocks = lpd_gemm_tn_bf16_ws_floats(M, KA, KB) / ((long long)KA * KB);
    long long rpb = (M + blocks - 1) / blocks;
    rpb = (rpb + 63) / 64 * 64;
    if (KA == 128) hipLaunchKernelGGL((gemm_tn_bf16_kernel<128, 128>), dim3((unsigned)blocks), dim3(256), 0, stream, A, B, ws, M, rpb);
    else hipLaunchKernelGGL((gemm_tn_bf16_kernel<64, 64>), dim3((unsigned)blocks), dim3(256), 0, stream, A, B, ws, M, rpb);
    LPD_CHECK_LAUNCH("lpd_gemm_tn_bf16");
    const int n = KA * KB;
    if (n % 64 == 0) hipLaunchKernelGGL(gemm_tn_reduce16_kernel, dim3(n / 64), dim3(256), 0, stream, (const float*)ws, dW, n, (int)blocks);
    else hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)ws, dW, n, (int)blocks);
    LPD_CHECK_LAUNCH("lpd_gemm_tn_bf16(reduce)");
    return LPD_OK;
}

namespace {

// ---------------------------------------------------------------------------------------------
// The wide weight gradient on ROW-MAJOR LDS images read with ds_read_b64_tr_b16 (gfx950's transposed LDS read): the reduction runs
// over the ROWS of both operands, so every MFMA operand is a column slice (one channel, 8 consecutive rows) of a row-major tensor.
// The kernels above transpose in registers on the way into [channel][row] images (v_perm + 16-byte writes per 8 x 8 patch; dW3 ran at
// 24 % of the MFMA time its products need, and bf16 rows as A changed nothing: the staging, not the operands, set the time).  Here the
// staged rows keep their layout -- bf16 rows are copied, fp32 rows split into a hi and a lo image, 8-byte writes -- and the hardware
// transposes on the read: per 16-lane group a block of 4 rows x 16 channels comes back channel-major, two reads give a lane its 8
// rows of one channel = the 32x32x16 operand (lanes 0..31 channels 0..31 rows 0..7, lanes 32..63 rows 8..15).
// Image rows are 512 + 64 bytes apart: the 4 rows of a read then sit on 4 x 64 bytes of different banks (conflict-free; on 512-byte
// rows all four would share their banks).  256 x 256 output tile, 8 waves (a wave: 64 x 128 = 2 x 4 accumulator tiles), 32-row
// chunks double-buffered, one barrier per chunk, the next chunk's rows requested before the MFMAs of this one.
// A16: A holds bf16 rows (hi image only, two products); else fp32 rows (hi + lo, three products).
// ---------------------------------------------------------------------------------------------
typedef short tr_i16x4 __attribute__((ext_vector_type(4)));
typedef short tr_i16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tr_u32x4 __attribute__((ext_vector_type(4)));     // (HIP's uint4 as a lambda-captured array element stayed in scratch memory)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1)
{
    typedef tr_i16x4 __attribute__((address_space(3))) * lds_ptr;
    const tr_i16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const tr_i16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    const tr_i16x8 v = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// TB: b-channels per block (256 / 128 / 64; a-channels: 256).  Wave tiles: TB = 256: 64 x 128 (2 x 4 accumulator tiles), 128: 64 x 64,
// 64: 32 x 64 -- narrow products (64 clusters, 64-channel layers) are bound by streaming A, whose rows every block reads exactly once.
// Batched over consecutive problems (sA / sB elements apart): the NetVLAD pooling, one problem per cloud.
// a_scale / a_shift (or null): the rows of A are act(a_scale[a] A[m][a] + a_shift[a]) -- a train-mode BatchNorm affine + activation
// applied where the raw map is staged (multiply, then add: the bits of lpd_affine_act; bf16 rows: rounded to bf16 again = the map a
// bf16-storing pass would have written), so that the activated [B N, 1024] map need not exist.
// B16: B holds bf16 rows too (ldb / sB in bf16 elements; TB = 256): copied like A's, no lo image -- with bf16 rows on both sides a term
// is ONE product, exact in the operands (dW3 of the bf16 storage mode: the bf16 gradient of the conv3 map against the bf16 point features).
template <bool A16, int TB, bool ATR = false, bool B16 = false>
__global__ __launch_bounds__(512, (A16 && TB != 256) ? 4 : 2) void gemm_tn_tr_kernel(const void* __restrict__ A_, long long lda, const float* __restrict__ B,
                                                                            long long ldb, float* __restrict__ slabs, int KA, int KB,
                                                                            long long rows_per_split, int nsplit, long long M, long long sA,
                                                                            long long sB, const float* __restrict__ a_scale,
                                                                            const float* __restrict__ a_shift, float a_ns)
{
    constexpr int ROWA = 576;                      // bytes per row of an A image (256 channels x 2 + 64)
    constexpr int ROWB = TB * 2 + 64;              // ... of a B image: 576 / 320 / 192, all = 64 mod 128 -> the 4 rows of a read on 4 bank groups
    constexpr int IMGA = 32 * ROWA, IMGB = 32 * ROWB;
    static_assert(!B16 || TB == 256, "bf16 rows as B: the 256-wide tile");
    constexpr int BUF = (A16 ? 1 : 2) * IMGA + (B16 ? 1 : 2) * IMGB;
    constexpr int A_HI = 0, A_LO = IMGA, B_HI = (A16 ? 1 : 2) * IMGA, B_LO = B_HI + IMGB;
    constexpr int NI = TB == 64 ? 1 : 2, NJ = TB == 256 ? 4 : 2;        // accumulator tiles of a wave
    constexpr int WGA = 256 / (NI * 32);                                // wave groups along a (4 or 8)
    constexpr int BQ = TB / 4;                                          // float4 per B row (64 / 32 / 16)
    constexpr int BP = TB * 32 / 4 / 512;                               // float4 of B per thread and chunk (4 / 2 / 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char trl[];      // [2 buffers][A hi [A lo] B hi B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int ntile_a = KA / 256, ntile_b = KB / TB;
    const int lin = lpd_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lin % (ntile_a * ntile_b), zall = lin / (ntile_a * ntile_b);      // zall = batch * nsplit + split
    const int split = zall % nsplit, zb = zall / nsplit;
    const int a0 = (tile % ntile_a) * 256, b0 = (tile / ntile_a) * TB;
    const long long m_begin = (long long)split * rows_per_split;
    const long long m_end = min(M, m_begin + rows_per_split);
    const int nchunk = m_end > m_begin ? (int)((m_end - m_begin) / 32) : 0;
    // staging: fp32 rows of A -- 4 passes of (row tid / 64 + 8 p, channels 4 (tid % 64) ..); bf16 rows -- 2 passes of (row tid / 32 + 16 p,
    // channels 8 (tid % 32) ..); B -- BP passes of (row tid / BQ + (512 / BQ) p, channels 4 (tid % BQ) ..): whole contiguous row pieces per wave
    const float* srcB = B + zb * sB + b0 + (tid % BQ) * 4 + (m_begin + tid / BQ) * ldb;
    const uint16_t* srcB16 = reinterpret_cast<const uint16_t*>(B) + zb * sB + b0 + (tid & 31) * 8 + (m_begin + (tid >> 5)) * ldb;      // B16: as A16
    const float* srcA = reinterpret_cast<const float*>(A_) + zb * sA + a0 + (tid & 63) * 4 + (m_begin + (tid >> 6)) * lda;
    const uint16_t* srcA16 = reinterpret_cast<const uint16_t*>(A_) + zb * sA + a0 + (tid & 31) * 8 + (m_begin + (tid >> 5)) * lda;
    const int wrb = (tid / BQ) * ROWB + (tid % BQ) * 8;        // + (512 / BQ) p rows
    const int wr32 = (tid >> 6) * ROWA + (tid & 63) * 8;       // + 8 p rows
    const int wr16 = (tid >> 5) * ROWA + (tid & 31) * 16;      // + 16 p rows
    float4 rb[BP], ra[4];
    tr_u32x4 ra16[2], rb16[2];
    // this thread's channels of A are the same in every chunk: their affine, once
    float asc[ATR ? 8 : 1], ash[ATR ? 8 : 1];
    if constexpr (ATR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = A16 ? a0 + (tid & 31) * 8 + e : a0 + (tid & 63) * 4 + (e & 3);
            asc[e] = a_scale[ch];
            ash[e] = a_shift[ch];
        }
    }
    auto tf = [&](float v, int e) { v = asc[ATR ? e : 0] * v + ash[ATR ? e : 0]; return fmaxf(v, 0.0f) + a_ns * fminf(v, 0.0f); };
    auto request = [&](int c) {
        const long long r = (long long)c * 32;
        if constexpr (B16) {
#pragma unroll
            for (int p = 0; p < 2; ++p) rb16[p] = *reinterpret_cast<const tr_u32x4*>(srcB16 + (r + 16 * p) * ldb);
        } else {
#pragma unroll
            for (int p = 0; p < BP; ++p) rb[p] = *reinterpret_cast<const float4*>(srcB + (r + (512 / BQ) * p) * ldb);
        }
        if constexpr (A16) {
#pragma unroll
            for (int p = 0; p < 2; ++p) ra16[p] = *reinterpret_cast<const tr_u32x4*>(srcA16 + (r + 16 * p) * lda);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const float4*>(srcA + (r + 8 * p) * lda);
        }
    };
    auto split_store = [&](unsigned char* hi, unsigned char* lo, const float4& v) {
        const uint32_t h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
        const uint32_t l01 = pack_bf16(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
        const uint32_t l23 = pack_bf16(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
        *reinterpret_cast<uint2*>(hi) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(lo) = make_uint2(l01, l23);
    };
    auto stage = [&](int buf) {
        unsigned char* base = trl + buf * BUF;
        if constexpr (B16) {
#pragma unroll
            for (int p = 0; p < 2; ++p) *reinterpret_cast<tr_u32x4*>(base + B_HI + wr16 + 16 * p * ROWB) = rb16[p];      // (ROWB == ROWA at TB = 256)
        } else {
#pragma unroll
            for (int p = 0; p < BP; ++p)
                split_store(base + B_HI + wrb + (512 / BQ) * p * ROWB, base + B_LO + wrb + (512 / BQ) * p * ROWB, rb[p]);
        }
        if constexpr (A16) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                tr_u32x4 w = ra16[p];
                if constexpr (ATR) {      // widen, transform, round
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        w[d] = pack_bf16(tf(__uint_as_float(w[d] << 16), 2 * d), tf(__uint_as_float(w[d] & 0xffff0000u), 2 * d + 1));
                }
                *reinterpret_cast<tr_u32x4*>(base + A_HI + wr16 + 16 * p * ROWA) = w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = ra[p];
                if constexpr (ATR) v = make_float4(tf(v.x, 0), tf(v.y, 1), tf(v.z, 2), tf(v.w, 3));
                split_store(base + A_HI + wr32 + 8 * p * ROWA, base + A_LO + wr32 + 8 * p * ROWA, v);
            }
        }
    };
    // transposed reads: lane 4 q + pp of 16-lane group g addresses row q, channels 4 pp .. of the group's 4 x 16 block; groups 0 / 1 take
    // channels 0..15 / 16..31 of the tile, groups 2 / 3 the same channels 8 rows further (the operand's second k half)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int wa = wave % WGA, wb = wave / WGA;    // the wave's tiles: A channels 32 NI wa .., B channels 32 NJ wb ..
    const int rd_a = ((g >> 1) * 8 + q) * ROWA + ((g & 1) * 16 + pp * 4 + wa * NI * 32) * 2;
    const int rd_b = ((g >> 1) * 8 + q) * ROWB + ((g & 1) * 16 + pp * 4 + wb * NJ * 32) * 2;
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if (nchunk > 0) {
        request(0);
        stage(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const unsigned char* base = trl + (c & 1) * BUF;
        if (c + 1 < nchunk) request(c + 1);
        // 2 NJ stages (k-step s, b tile j), software-pipelined by one: the transposed reads of stage t + 1 are issued in front of the
        // MFMAs of stage t (with all reads of a k-step in front of its MFMAs both waves of a SIMD -- in step behind the barrier -- read,
        // then multiply: the compute phase took twice the MFMA time)
        bf16x8 ah[2][NI], al[2][NI], bh[2], bl[2];
        auto load_a = [&](int s) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const unsigned char* pa = base + rd_a + s * 16 * ROWA + i * 64;
                ah[s & 1][i] = tr_frag(pa + A_HI, pa + A_HI + 4 * ROWA);
                if constexpr (!A16) al[s & 1][i] = tr_frag(pa + A_LO, pa + A_LO + 4 * ROWA);
            }
        };
        auto load_b = [&](int t) {
            const int s = t / NJ, j = t % NJ;
            const unsigned char* pb = base + rd_b + s * 16 * ROWB + j * 64;
            bh[t & 1] = tr_frag(pb + B_HI, pb + B_HI + 4 * ROWB);
            if constexpr (!B16) bl[t & 1] = tr_frag(pb + B_LO, pb + B_LO + 4 * ROWB);
        };
        load_a(0);
        load_b(0);
#pragma unroll
        for (int t = 0; t < 2 * NJ; ++t) {
            const int s = t / NJ, j = t % NJ;
            if (t + 1 < 2 * NJ) {
                if ((t + 1) % NJ == 0) load_a((t + 1) / NJ);
                load_b(t + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if constexpr (!A16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s & 1][i], bh[t & 1], acc[i][j], 0, 0, 0);
                if constexpr (!B16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s & 1][i], bl[t & 1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s & 1][i], bh[t & 1], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < nchunk) stage((c + 1) & 1);
        __syncthreads();
    }
    float* slab = slabs + (size_t)zall * KA * KB;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = a0 + (wa * NI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)arow * KB + b0 + (wb * NJ + j) * 32 + col] = acc[i][j][r];
            }
}

}  // namespace

// the 256 x 256-tile kernel: wide products over whole 32-row chunks (LPD_DEBUG=tn256=0 keeps the 128-wide kernel)
static bool gemm_tn_256(long long M, int KA, int KB, int batch)
{
    static const bool on = lpd_debug("tn256", 1) != 0;
    return on && batch == 1 && KA % 256 == 0 && KB % 256 == 0 && M % 32 == 0 && M >= 8192;
}

// the transposed-read kernel (gemm_tn_tr_kernel): b-channels per block, or 0 where it is not built (LPD_DEBUG=tn-tr=0: never)
static int tn_tr_tb(long long M, int KA, int KB, int batch, bool a_bf16, bool act = false)      // act: with the operand transform (this kernel only)
{
    static const bool on = lpd_debug("tn-tr", 1) != 0;
    if (KA % 256 != 0 || KB % 64 != 0 || M % 32 != 0 || M < 2048) return 0;
    if (act) return KB % 256 == 0 ? 256 : (KB % 128 == 0 ? 128 : 64);
    if (!on) return 0;
    // fp32 rows as A leave room for ONE block per CU: the pooling's 4 x 44 tiles of 4096 rows then run as 176 or 352 blocks on 256 CUs
    // (208 us against 190 on the register-transposing kernel, which holds several blocks per CU)
    if (!a_bf16 && batch > 1) return 0;
    return KB % 256 == 0 ? 256 : (KB % 128 == 0 ? 128 : 64);
}

static long long gemm_tn_splits(long long M, int KA, int KB, int batch, bool a_bf16 = true, bool act = false)
{
    if (const int tb = tn_tr_tb(M, KA, KB, batch, a_bf16, act)) {
        // blocks per CU by LDS: two with bf16 rows as A and narrow B tiles (61 / 82 KiB), else one
        const long long tiles = (long long)(KA / 256) * (KB / tb) * batch, cap = (tb == 256 || !a_bf16) ? 256 : 512;
        long long splits = 1;
        while (tiles * splits * 2 <= cap && M / (splits * 2) >= 256) splits *= 2;
        return splits;
    }
    if (gemm_tn_256(M, KA, KB, batch)) {
        const long long tiles = (long long)(KA / 256) * (KB / 256);
        long long splits = 1;
        while (tiles * splits * 2 <= 256 && M / (splits * 2) >= 1024) splits *= 2;
        return splits;
    }
    long long tiles = (long long)(KA / 128) * ((KB + 127) / 128) * batch;
    long long splits = 1;
    static const int cap = lpd_debug("tn-blocks", 512);      // one resident round of blocks (2 per CU): half the slab traffic of 1024 (eval pooling 147 -> 138 us)
    while (tiles * splits * 2 <= cap && M / (splits * 2) >= 1024) splits *= 2;
    return splits;
}

extern "C" long long lpd_gemm_tn_ws_floats(long long M, int KA, int KB, int batch)
{
    if (batch < 1) batch = 1;
    const long long a = gemm_tn_splits(M, KA, KB, batch, true), b = gemm_tn_splits(M, KA, KB, batch, false);      // either operand type
    return (a > b ? a : b) * batch * KA * KB;
}

extern "C" long long lpd_gemm_tn_act_ws_floats(long long M, int KA, int KB, int batch, int a_bf16);

// lpd_gemm_tn with the operand transform of gemm_tn_tr_kernel: the rows of A are act(a_scale[a] A[m][a] + a_shift[a]).  Built on the
// transposed-read kernel only (KA % 256 == 0, M % 32 == 0, M >= 2048).
extern "C" int lpd_gemm_tn_act(const void* A_, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB,
                               int batch, long long sA, long long sB, int a_bf16, const float* a_scale, const float* a_shift, int a_act,
                               float a_slope, void* stream_);

static int gemm_tn_impl(const void* A_, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB,
                        int batch, long long sA, long long sB, int a_bf16, const float* a_scale, const float* a_shift, float a_ns, void* stream_);

extern "C" long long lpd_gemm_tn_act_ws_floats(long long M, int KA, int KB, int batch, int a_bf16)
{
    if (batch < 1) batch = 1;
    return gemm_tn_splits(M, KA, KB, batch, a_bf16 != 0, true) * batch * KA * KB;
}

extern "C" int lpd_gemm_tn(const void* A_, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB,
                           int batch, long long sA, long long sB, int bf16_rows, void* stream_)
{
    return gemm_tn_impl(A_, lda, B, ldb, dW, ws, M, KA, KB, batch, sA, sB, bf16_rows, nullptr, nullptr, 1.0f, stream_);
}

extern "C" int lpd_gemm_tn_act(const void* A_, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB,
                               int batch, long long sA, long long sB, int a_bf16, const float* a_scale, const float* a_shift, int a_act,
                               float a_slope, void* stream_)
{
    LPD_CHECK_ARG(a_scale && a_shift && a_act >= 0 && a_act <= 2, "lpd_gemm_tn_act: scale / shift / activation");
    return gemm_tn_impl(A_, lda, B, ldb, dW, ws, M, KA, KB, batch, sA, sB, a_bf16, a_scale, a_shift,
                        a_act == 0 ? 1.0f : (a_act == 1 ? 0.0f : a_slope), stream_);
}

static int gemm_tn_impl(const void* A_, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB,
                        int batch, long long sA, long long sB, int ab_flags, const float* a_scale, const float* a_shift, float a_ns, void* stream_)
{
    // ab_flags: bit 0 = A holds bf16 rows, bit 1 = B holds bf16 rows (with bit 0, KB % 256 == 0, the transposed-read kernel)
    const int a_bf16 = ab_flags & 1, b_bf16 = (ab_flags >> 1) & 1;
    hipStream_t stream = (hipStream_t)stream_;
    const float* A = reinterpret_cast<const float*>(A_);      // a_bf16: bf16 rows (lda, sA in bf16 elements)
    LPD_CHECK_ARG(A && B && dW && ws && M > 0 && batch >= 1, "lpd_gemm_tn: bad arguments");
    LPD_CHECK_ARG(KA > 0 && KA % 128 == 0 && KB > 0 && KB % 64 == 0, "lpd_gemm_tn: KA %% 128 and KB %% 64 required (KA=%d KB=%d)", KA, KB);
    LPD_CHECK_ARG(lda % (a_bf16 ? 8 : 4) == 0 && ldb % (b_bf16 ? 8 : 4) == 0 && sA % (a_bf16 ? 8 : 4) == 0 && sB % (b_bf16 ? 8 : 4) == 0 &&
                      (((uintptr_t)A | (uintptr_t)B) & 15) == 0, "lpd_gemm_tn: operands must be 16-byte aligned rows");
    LPD_CHECK_ARG(!b_bf16 || (a_bf16 && !a_scale && batch == 1 && tn_tr_tb(M, KA, KB, batch, true, true) == 256),
                  "lpd_gemm_tn: bf16 rows as B are built with bf16 rows as A, KA %% 256 == 0, KB %% 256 == 0, M %% 32 == 0, M >= 2048");
    const bool act = a_scale != nullptr || b_bf16;      // (both forms exist on the transposed-read kernel only)
    LPD_CHECK_ARG(!act || tn_tr_tb(M, KA, KB, batch, a_bf16 != 0, true),
                  "lpd_gemm_tn_act: the operand transform is built for KA %% 256 == 0, M %% 32 == 0, M >= 2048 (M=%lld KA=%d)", M, KA);
    const long long splits = gemm_tn_splits(M, KA, KB, batch, a_bf16 != 0, act);
    long long rps = (M + splits - 1) / splits;
    rps = (rps + 63) / 64 * 64;
    if (const int tb = tn_tr_tb(M, KA, KB, batch, a_bf16 != 0, act)) {
        rps = (M + splits - 1) / splits;
        rps = (rps + 31) / 32 * 32;
        const long long blocks = (long long)(KA / 256) * (KB / tb) * splits * batch;
        LPD_CHECK_ARG(blocks < (1ll << 31), "lpd_gemm_tn: too many blocks");
        const int lds_tr = 2 * ((a_bf16 ? 1 : 2) * 32 * 576 + (b_bf16 ? 1 : 2) * 32 * (tb * 2 + 64));
#define LPD_TN_TR_LAUNCH(AB_, TB_, TR_)                                                                                                     \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)gemm_tn_tr_kernel<AB_, TB_, TR_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_tr);       \
        hipLaunchKernelGGL((gemm_tn_tr_kernel<AB_, TB_, TR_>), dim3((unsigned)blocks), dim3(512), lds_tr, stream, A_, lda, B, ldb, ws, KA, KB, \
                           rps, (int)splits, M, sA, sB, a_scale, a_shift, a_ns);                                                            \
    } while (0)
        if (b_bf16) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_tr_kernel<true, 256, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_tr);
            hipLaunchKernelGGL((gemm_tn_tr_kernel<true, 256, false, true>), dim3((unsigned)blocks), dim3(512), lds_tr, stream, A_, lda, B, ldb, ws, KA, KB,
                               rps, (int)splits, M, sA, sB, a_scale, a_shift, a_ns);
        } else if (act) {      // the operand transform: the 64-wide tile (NetVLAD pooling and assignment weight gradient)
            LPD_CHECK_ARG(tb == 64, "lpd_gemm_tn_act: built for KB %% 128 != 0 (64-wide b tiles), KB=%d", KB);
            if (a_bf16) LPD_TN_TR_LAUNCH(true, 64, true); else LPD_TN_TR_LAUNCH(false, 64, true);
        } else if (a_bf16) { if (tb == 256) LPD_TN_TR_LAUNCH(true, 256, false); else if (tb == 128) LPD_TN_TR_LAUNCH(true, 128, false); else LPD_TN_TR_LAUNCH(true, 64, false); }
        else { if (tb == 256) LPD_TN_TR_LAUNCH(false, 256, false); else if (tb == 128) LPD_TN_TR_LAUNCH(false, 128, false); else LPD_TN_TR_LAUNCH(false, 64, false); }
#undef LPD_TN_TR_LAUNCH
        LPD_CHECK_LAUNCH("lpd_gemm_tn(tr)");
        const int n = KA * KB;
        hipLaunchKernelGGL(gemm_tn_reduce16_kernel, dim3(n / 64, batch), dim3(256), 0, stream, (const float*)ws, dW, n, (int)splits);
        LPD_CHECK_LAUNCH("lpd_gemm_tn(reduce)");
        return LPD_OK;
    }
    if (gemm_tn_256(M, KA, KB, batch)) {
        const long long blocks = (long long)(KA / 256) * (KB / 256) * splits;
        constexpr int lds = 2 * 512 * 144;
        if (a_bf16) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_x3_256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL(gemm_tn_x3_256_kernel<true>, dim3((unsigned)blocks), dim3(512), lds, stream, A, lda, B, ldb, ws, KA, KB, rps, M);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_tn_x3_256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL(gemm_tn_x3_256_kernel<false>, dim3((unsigned)blocks), dim3(512), lds, stream, A, lda, B, ldb, ws, KA, KB, rps, M);
        }
        LPD_CHECK_LAUNCH("lpd_gemm_tn(256)");
        const int n = KA * KB;
        if (n % 64 == 0) hipLaunchKernelGGL(gemm_tn_reduce16_kernel, dim3(n / 64, 1), dim3(256), 0, stream, (const float*)ws, dW, n, (int)splits);
        else hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((n + 255) / 256, 1), dim3(256), 0, stream, (const float*)ws, dW, n, (int)splits);
        LPD_CHECK_LAUNCH("lpd_gemm_tn(reduce)");
        return LPD_OK;
    }
    const int kbt = KB % 128 == 0 ? 128 : 64;
    const long long blocks = (long long)(KA / 128) * (KB / kbt) * splits * batch;
    LPD_CHECK_ARG(blocks < (1ll << 31), "lpd_gemm_tn: too many blocks");
#define LPD_TN_LAUNCH(KBT_, AB_) hipLaunchKernelGGL((gemm_tn_x3_kernel<KBT_, AB_>), dim3((unsigned)blocks), dim3(256), 0, stream, A, lda, B, ldb, ws, M, KA, KB, rps, (int)splits, sA, sB)
    if (kbt == 128) { if (a_bf16) LPD_TN_LAUNCH(128, true); else LPD_TN_LAUNCH(128, false); }
    else { if (a_bf16) LPD_TN_LAUNCH(64, true); else LPD_TN_LAUNCH(64, false); }
#undef LPD_TN_LAUNCH
    LPD_CHECK_LAUNCH("lpd_gemm_tn");
    const int n = KA * KB;
    if (n % 64 == 0) hipLaunchKernelGGL(gemm_tn_reduce16_kernel, dim3(n / 64, batch), dim3(256), 0, stream, (const float*)ws, dW, n, (int)splits);
    else hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, stream, (const float*)ws, dW, n, (int)splits);
    LPD_CHECK_LAUNCH("lpd_gemm_tn(reduce)");
    return LPD_OK;
}
