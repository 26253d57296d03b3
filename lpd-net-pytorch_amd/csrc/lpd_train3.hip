// lpd_train3.hip -- backward of the DG2 stage (x2 = max_k act(BN_train(convDG2(y1e))), util/lpdnet_model.py:251-252) on bf16 edge
// tensors WITHOUT the [E, 128] gradient tensor dZ.
//
// With dpre_i[c] = dx2_i[c] * act'(pre) living on the arg-max edge of (point i, channel c) only, m1 = mean(dpre), m2 = mean(dpre xhat)
// over all E = M k edges and xhat = (z - mu) invstd, the BatchNorm backward is
//     dZ[(i,t)][c] = s_c (delta_{t,arg_ic} dpre_i[c] - m1_c - xhat[(i,t)][c] m2_c)                                   (dense: E x 128).
// lpd_train2.hip wrote dZ (bf16, 0.92 GB at B = 44), read it for dW2 = dZ^T Y1e and again for dY1e = dZ W2: 467 + 699 + 558 us.
// Here:
//   (1) lpd_bn_sel_bwd_reduce: an [M, C] pass over (dx2, the selected raw values kept by the forward): dpre (bf16), sum dpre, sum dpre xhat.
//   (2) lpd_edge_dw_sel_bf16: ONE pass over Y1e computes S = D^T Y1e (D = the arg-max matrix delta dpre, built from [M, C] data while
//       staging), the Gram matrix G = Y1e^T Y1e and the column sums s = 1^T Y1e; then, because z = Y1e W2^T,
//           dW2[c][:] = s_c (S[c][:] - m1_c s - m2_c invstd_c (W2[c][:] G - mu_c s))                     (lpd_dw2_finish, fp64).
//   (3) lpd_gemm_bf16s_bnbwd: dY1e = dZ W2 with dZ generated in the A-operand loader from z (bf16), (arg, dpre) of the row's point and
//       two constants per channel: A = z a1 + a0 + delta dpre, the factor s_c folded into the staged weight rows.
#include "lpd_common.h"
#include <math.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b)   // RNE (v_cvt_pk_bf16_f32)
{
    const bf16x2v h = __builtin_convertvector((f32x2v){a, b}, bf16x2v);
    return __builtin_bit_cast(uint32_t, h);
}

inline int grid_for(long long items, int per_block, int cap = 4096)
{
    long long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ------------------------------------------------------------------------------------------ (1)
// dpre16[i][c] = bf16(dOut[i][c] * act'(scale x + shift)), dbeta += dpre, dgamma += dpre xhat (fp32 values, fp64 sums; the same
// arithmetic and loop order as edge_bn_bwd_reduce_bf16_kernel's arg-max branch)
__global__ __launch_bounds__(256) void bn_sel_bwd_reduce_kernel(const float* __restrict__ dOut, long long ldo, const float* __restrict__ Xsel,
                                                                long long ldsel, long long M, int C, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float ns, uint16_t* __restrict__ dpre16,
                                                                float* __restrict__ dpre32, double* __restrict__ dbeta,
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
        const float4 x4 = *reinterpret_cast<const float4*>(Xsel + i * ldsel + q * 4);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const float x[4] = {x4.x, x4.y, x4.z, x4.w};
        float dp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dp[c] = g[c] * (sc[c] * x[c] + sh[c] > 0.0f ? 1.0f : ns);
            sb[c] += dp[c];
            sg[c] += (double)dp[c] * ((x[c] - mu[c]) * is[c]);
        }
        if (dpre16) *reinterpret_cast<uint2*>(dpre16 + i * C + q * 4) = make_uint2(pack_bf16(dp[0], dp[1]), pack_bf16(dp[2], dp[3]));
        if (dpre32) *reinterpret_cast<float4*>(dpre32 + i * C + q * 4) = make_float4(dp[0], dp[1], dp[2], dp[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = sb[e]; red[threadIdx.x][4 + e] = sg[e]; }
    __syncthreads();
    if ((int)threadIdx.x < LQ) {
        for (int g2 = 1; g2 < RG; ++g2)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[g2 * LQ + threadIdx.x][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&dbeta[lpd_stat_rofs() + threadIdx.x * 4 + e], red[threadIdx.x][e]);
            atomicAdd(&dgamma[lpd_stat_rofs() + threadIdx.x * 4 + e], red[threadIdx.x][4 + e]);
        }
    }
}

// ------------------------------------------------------------------------------------------ (2)
// slab[block] = [ S = D^T Y (128 x 128) | G = Y^T Y (128 x 128) | s = column sums of Y (128) ] over the block's rows of Y [E][128] bf16.
// 512 threads, 128-row chunks in the two halves of [channel][row] LDS images (channel rows of 528 bytes: 2 x 128 rows + 16 bytes
// of padding -- operand reads on 132-dword strides and store groups of 8 consecutive row groups are conflict-free, cf.
// gemm_tn_bf16_kernel), one barrier per chunk: waves 0..3 stage Y patches (8 rows x 8 channels, register transpose), waves 4..7
// build the D patches from (arg, dpre) of the one or two points their 8 rows belong to; the loads of chunk n+2 are requested at
// the top of chunk n.  Wave (wa, wb) owns output row tiles wa (of S) and 4 + wa (of G), column tiles 2 wb, 2 wb + 1.
// (First version: 256 threads, 64-row chunks, 8 accumulator tiles per wave: 34 spilled registers, loads one chunk ahead, 518 us.)
__global__ __launch_bounds__(512, 2) void edge_dw_sel_bf16_kernel(const uint16_t* __restrict__ Y, const uint8_t* __restrict__ arg,
                                                                  const uint16_t* __restrict__ dpre16, int k, long long E, long long Mp,
                                                                  long long rows_per_block, float* __restrict__ slabs)
{
    constexpr int LDB = 528;                       // bytes per channel row of an image
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_lds[];
    unsigned char* const img_d = dw_lds;
    unsigned char* const img_y = dw_lds + 128 * LDB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long m_begin = (long long)blockIdx.x * rows_per_block;
    const long long m_end = min(E, m_begin + rows_per_block);
    const int nchunk = m_end > m_begin ? (int)((m_end - m_begin) / 128) : 0;     // whole chunks: E % 128 == 0 (host check)
    const bool isY = __builtin_amdgcn_readfirstlane(tid) < 256;      // wave-uniform: waves 0..3 stage Y; waves 4..7 build D
    const int wv = wave & 3;
    const int rg = (lane & 7) + 8 * (wv >> 1), c8 = (lane >> 3) + 8 * (wv & 1);   // row group (8 rows) 0..15, channel group (8 channels) 0..15
    unsigned char* const img = isY ? img_y : img_d;
    const int wr_off = (c8 * 8) * LDB + rg * 16;   // + 256 * half + channel * LDB
    const int wa = wave & 3, wb = wave >> 2;
    const int rd_a = (wa * 32 + col) * LDB + h * 16;             // + 256 * half + 32 * kstep   (D image and Y image)
    const int rd_b = (wb * 64 + col) * LDB + h * 16;             // + 32 * LDB * j

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // raw registers of a chunk: Y threads 8 row pieces; D threads (dpre, dpre, arg | arg) of two points in r[0..2], the first row's slot in ta
    struct Raw { uint4 r[8]; int ta; };
    Raw r0, r1;
    auto load = [&](Raw& R, int chunk) {
        const long long row0 = m_begin + (long long)chunk * 128 + rg * 8;
        if (isY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) R.r[i] = *reinterpret_cast<const uint4*>(Y + (row0 + i) * 128 + c8 * 8);
        } else {
            const long long ia = row0 / k;
            const long long ib = ia + 1 < Mp ? ia + 1 : Mp - 1;
            R.ta = (int)(row0 - ia * k);
            const uint2 aa = *reinterpret_cast<const uint2*>(arg + ia * 128 + c8 * 8);
            const uint2 ab = *reinterpret_cast<const uint2*>(arg + ib * 128 + c8 * 8);
            R.r[0] = *reinterpret_cast<const uint4*>(dpre16 + ia * 128 + c8 * 8);
            R.r[1] = *reinterpret_cast<const uint4*>(dpre16 + ib * 128 + c8 * 8);
            R.r[2] = make_uint4(aa.x, aa.y, ab.x, ab.y);
        }
    };
    auto transpose_store = [&](const uint32_t (&w)[8][4], int half) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {              // channel pair (2p, 2p+1) of the group: rows 0..7 as four dwords each
            uint4 e, o;
            e.x = __builtin_amdgcn_perm(w[1][p], w[0][p], 0x05040100u); o.x = __builtin_amdgcn_perm(w[1][p], w[0][p], 0x07060302u);
            e.y = __builtin_amdgcn_perm(w[3][p], w[2][p], 0x05040100u); o.y = __builtin_amdgcn_perm(w[3][p], w[2][p], 0x07060302u);
            e.z = __builtin_amdgcn_perm(w[5][p], w[4][p], 0x05040100u); o.z = __builtin_amdgcn_perm(w[5][p], w[4][p], 0x07060302u);
            e.w = __builtin_amdgcn_perm(w[7][p], w[6][p], 0x05040100u); o.w = __builtin_amdgcn_perm(w[7][p], w[6][p], 0x07060302u);
            const int off = wr_off + half * 256 + (2 * p) * LDB;
            *reinterpret_cast<uint4*>(img + off) = e;
            *reinterpret_cast<uint4*>(img + off + LDB) = o;
        }
    };
    auto convert_store = [&](const Raw& R, int half) {
        uint32_t w[8][4];
        if (isY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w[i][0] = R.r[i].x; w[i][1] = R.r[i].y; w[i][2] = R.r[i].z; w[i][3] = R.r[i].w;
#pragma unroll
                for (int d = 0; d < 4; ++d) {      // column sums of the staged rows (fp32 per thread, fp64 across blocks)
                    cs[2 * d] += __uint_as_float(w[i][d] << 16);
                    cs[2 * d + 1] += __uint_as_float(w[i][d] & 0xffff0000u);
                }
            }
        } else {
            const uint32_t dpa[4] = {R.r[0].x, R.r[0].y, R.r[0].z, R.r[0].w}, dpb[4] = {R.r[1].x, R.r[1].y, R.r[1].z, R.r[1].w};
            const uint32_t aa[2] = {R.r[2].x, R.r[2].y}, ab[2] = {R.r[2].z, R.r[2].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tt = R.ta + i;
                const bool second = tt >= k;
                const uint32_t t = (uint32_t)(second ? tt - k : tt);
#pragma unroll
                for (int d = 0; d < 4; ++d) {      // channels 2d, 2d + 1
                    const uint32_t a2 = ((second ? ab[d >> 1] : aa[d >> 1]) >> (16 * (d & 1))) & 0xffffu;
                    const uint32_t dp = second ? dpb[d] : dpa[d];
                    const uint32_t mask = ((a2 & 0xffu) == t ? 0xffffu : 0u) | ((a2 >> 8) == t ? 0xffff0000u : 0u);
                    w[i][d] = dp & mask;
                }
            }
        }
        transpose_store(w, half);
    };
    auto mma = [&](int half, int ks0) {            // four of the chunk's eight k-steps
#pragma unroll
        for (int ks = ks0; ks < ks0 + 4; ++ks) {
            const int o = half * 256 + ks * 32;
            const bf16x8 ad = *reinterpret_cast<const bf16x8*>(img_d + rd_a + o);
            const bf16x8 ay = *reinterpret_cast<const bf16x8*>(img_y + rd_a + o);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(img_y + rd_b + o + j * 32 * LDB);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, b, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay, b, acc[1][j], 0, 0, 0);
            }
        }
    };
    auto chunk = [&](int n, const Raw& cur, Raw& nxt) {
        load(nxt, n + 2 < nchunk ? n + 2 : nchunk - 1);        // unconditional: the last chunks are re-read and dropped
        mma(n & 1, 0);
        if (n + 1 < nchunk) convert_store(cur, (n + 1) & 1);
        mma(n & 1, 4);
        __syncthreads();
    };
    if (nchunk > 0) {
        load(r0, 0);
        convert_store(r0, 0);
        load(r0, nchunk > 1 ? 1 : 0);
        __syncthreads();
        for (int n = 0; n < nchunk; n += 2) {
            chunk(n, r0, r1);
            if (n + 1 < nchunk) chunk(n + 1, r1, r0);
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * (256 * 128 + 128);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = i * 128 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)arow * 128 + wb * 64 + j * 32 + col] = acc[i][j][r];
            }
    // column sums: the 16 row-group threads of a channel group, summed in a fixed order through LDS (the images are free now)
    float* tmp = reinterpret_cast<float*>(img_d);
    if (isY)
#pragma unroll
        for (int e = 0; e < 8; ++e) tmp[(c8 * 16 + rg) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128) {
        const int g8 = tid >> 3, e = tid & 7;
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += tmp[(g8 * 16 + r) * 8 + e];
        slab[256 * 128 + tid] = t;
    }
}

// out[e] = sum over the slabs, fp64.  16 slab groups per block (16 threads x 4 elements each; n % 64 == 0): one thread per element walked
// the 256 slabs one dependent load at a time (129 blocks on 256 CUs).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, double* __restrict__ out, int n, int nslabs)
{
    __shared__ double red[16][16][4];
    const int t = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = (blockIdx.x * 16 + t) * 4;
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
        out[(blockIdx.x * 16 + tt) * 4 + c] = v;
    }
}

// dW2[c][n] = s_c (S[c][n] - m1_c s[n] - m2_c invstd_c (sum_m W2[c][m] G[m][n] - mu_c s[n]))      one block per c, one thread per n
__global__ __launch_bounds__(128) void dw2_finish_kernel(const double* __restrict__ R, const float* __restrict__ W2, long long ldw,
                                                         const float* __restrict__ scale, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const double* __restrict__ dbeta,
                                                         const double* __restrict__ dgamma, double count, float* __restrict__ dW2)
{
    const int c = blockIdx.x, n = threadIdx.x;
    const double* S = R;
    const double* G = R + 128 * 128;
    const double* s = R + 2 * 128 * 128;
    double zy = 0.0;
    for (int m = 0; m < 128; ++m) zy += (double)W2[(size_t)c * ldw + m] * G[m * 128 + n];
    const double m1 = dbeta[c] / count, m2 = dgamma[c] / count;
    dW2[c * 128 + n] = (float)((double)scale[c] * (S[c * 128 + n] - m1 * s[n] - m2 * (double)invstd[c] * (zy - (double)mean[c] * s[n])));
}

// ------------------------------------------------------------------------------------------ (3)
// dY[e][n] (bf16) = sum_c dZ[e][c] W2[c][n], dZ[e][c] = s_c (delta dpre - m1_c - xhat m2_c) generated in the loader:
// A[e][c] = z[e][c] a1_c + a0_c + (arg[i][c] == t ? dpre[i][c] : 0), a1 = -invstd m2, a0 = mu invstd m2 - m1, e = (i, t);
// staged weight rows s_c W2[c][:] (hi + lo).  Structure of gemm_bf16s_kernel (one wave: 32 rows x all 128 columns, transposed
// MFMA tile, contraction index permuted k = h 64 + 8 s + e).  A tile's 32 rows belong to at most three points (k >= 16): their
// arg / dpre rows (128 + 256 bytes each) go through a wave-private LDS buffer, requested one tile ahead and BEFORE the next
// tile's z rows (vmcnt counts in order: the wait for them leaves the z rows in flight); as per-lane registers the two
// would be 48 more registers per tile in flight (first version: 333 spilled registers).
__global__ __launch_bounds__(256, 2) void gemm_bf16s_bnbwd_kernel(const uint16_t* __restrict__ Z, const uint8_t* __restrict__ arg,
                                                                   const uint16_t* __restrict__ dpre16, int k, const float* __restrict__ W2,
                                                                   int ldw, const float* __restrict__ scale, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const double* __restrict__ dbeta,
                                                                   const double* __restrict__ dgamma, double count,
                                                                   uint16_t* __restrict__ Cout, long long E, long long Mp)
{
    constexpr int K = 128, N = 128, KS = K / 16, NT = N / 32, LDW = K + 8;
    constexpr int PB = 384;                         // bytes of one point in the side buffer: dpre (256) | arg (128)
    extern __shared__ __attribute__((aligned(16))) __bf16 wimg3[];   // hi image, lo image (2 * N * LDW bf16), float2 consts [K], side buffers
    __bf16* whi = wimg3;
    __bf16* wlo = wimg3 + N * LDW;
    float2* cst = reinterpret_cast<float2*>(wimg3 + 2 * N * LDW);
    unsigned char* side = reinterpret_cast<unsigned char*>(cst + K) + (threadIdx.x >> 6) * (2 * 3 * PB);   // this wave's two buffers
    const int tid = threadIdx.x;
    for (int e = tid; e < N * K; e += 256) {       // W(n, kk) = s_kk W2[kk][n]
        const int n = e % N, kk = e / N;
        const float w = W2[(size_t)kk * ldw + n] * scale[kk];
        const __bf16 hi = (__bf16)w;
        const __bf16 lo = (__bf16)(w - (float)hi);
        whi[n * LDW + kk] = hi;
        wlo[n * LDW + kk] = lo;
    }
    if (tid < K) {
        const float m1 = (float)(dbeta[tid] / count), m2 = (float)(dgamma[tid] / count);
        cst[tid] = make_float2(-invstd[tid] * m2, mean[tid] * invstd[tid] * m2 - m1);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long ntile = (E + 31) / 32;
    const long long stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    uint4 zc[KS], zn[KS], sd, sa;
    // side data of a tile: lane L stages dpre piece (point L / 16, 16 bytes L % 16) and arg piece (point (L & 31) / 8, 16 bytes L % 8)
    const int sd_p = (lane >> 4) < 3 ? (lane >> 4) : 2, sd_q = lane & 15;
    const int sa_p = ((lane & 31) >> 3) < 3 ? ((lane & 31) >> 3) : 2, sa_q = lane & 7;
    auto load_side = [&](long long tl) {
        const long long i0 = (tl * 32 < E ? tl * 32 : E - 1) / k;
        const long long pd = i0 + sd_p < Mp ? i0 + sd_p : Mp - 1, pa = i0 + sa_p < Mp ? i0 + sa_p : Mp - 1;
        sd = *reinterpret_cast<const uint4*>(dpre16 + pd * K + sd_q * 8);
        sa = *reinterpret_cast<const uint4*>(arg + pa * K + sa_q * 16);
    };
    auto store_side = [&](int buf) {
        unsigned char* b = side + buf * (3 * PB);
        if (lane < 48) *reinterpret_cast<uint4*>(b + sd_p * PB + sd_q * 16) = sd;
        if (lane < 24) *reinterpret_cast<uint4*>(b + sa_p * PB + 256 + sa_q * 16) = sa;
    };
    auto load_z = [&](long long tl, uint4 (&z)[KS]) {
        const long long row = tl * 32 + col;
        const uint16_t* zp = Z + (row < E ? row : E - 1) * K + h * (K / 2);
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) z[s2] = *reinterpret_cast<const uint4*>(zp + s2 * 8);
    };
    int buf = 0;
    if (tile < ntile) {
        load_side(tile);
        load_z(tile, zc);
        store_side(0);
    }
    for (; tile < ntile; tile += stride, buf ^= 1) {
        const long long row = tile * 32 + col;
        const long long rr = row < E ? row : E - 1;
        const long long i0 = (tile * 32 < E ? tile * 32 : E - 1) / k;
        const long long ip = rr / k;
        const uint32_t t = (uint32_t)(rr - ip * k);
        const unsigned char* sb = side + buf * (3 * PB) + (int)(ip - i0) * PB + h * 128;     // this lane's point: dpre half, arg half at + 256 - h * 64
        const long long tn = tile + stride < ntile ? tile + stride : tile;
        load_side(tn);                              // first (see above), then the z rows of the next tile
        load_z(tn, zn);
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            const uint32_t zw[4] = {zc[s2].x, zc[s2].y, zc[s2].z, zc[s2].w};
            const uint4 d4 = *reinterpret_cast<const uint4*>(sb + s2 * 16);
            const uint2 a2v = *reinterpret_cast<const uint2*>(sb + 256 - h * 64 + s2 * 8);
            const uint32_t dw[4] = {d4.x, d4.y, d4.z, d4.w};
            const uint32_t aw[2] = {a2v.x, a2v.y};
            uint32_t op[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {          // channels h 64 + 8 s2 + 2 d, + 1
                const float4 c2 = *reinterpret_cast<const float4*>(&cst[h * (K / 2) + s2 * 8 + 2 * d]);   // (a1, a0) of both channels
                const uint32_t a2 = (aw[d >> 1] >> (16 * (d & 1))) & 0xffffu;
                const float v0 = fmaf(__uint_as_float(zw[d] << 16), c2.x, c2.y) + ((a2 & 0xffu) == t ? __uint_as_float(dw[d] << 16) : 0.0f);
                const float v1 = fmaf(__uint_as_float(zw[d] & 0xffff0000u), c2.z, c2.w) +
                                 ((a2 >> 8) == t ? __uint_as_float(dw[d] & 0xffff0000u) : 0.0f);
                op[d] = pack_bf16(v0, v1);
            }
            const bf16x8 av = __builtin_bit_cast(bf16x8, make_uint4(op[0], op[1], op[2], op[3]));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int off = (j * 32 + col) * LDW + h * (K / 2) + s2 * 8;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(whi + off);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wlo + off);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, av, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, av, acc[j], 0, 0, 0);
            }
        }
        store_side(buf ^ 1);                        // the next tile's side data (this wave's own buffer: no barrier)
        if (row < E) {
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
        for (int s2 = 0; s2 < KS; ++s2) zc[s2] = zn[s2];
    }
}

// ------------------------------------------------------------------------------------------ fp32 storage: the same two kernels
// (2f) slab[block] = [ S = D^T Y | G = Y^T Y | column sums of Y ] for fp32 Y [E][128] and fp32 dpre [M][128], split-bf16 products (three
// per term).  512 threads, 64-row chunks in the two halves of FOUR images (D hi / lo, Y hi / lo: [channel][2 x 64 rows] bf16, 272-byte
// channel rows); a staging thread owns an 8-row x 4-channel patch (waves 0..3: Y, waves 4..7: D), loads two chunks ahead.
__global__ __launch_bounds__(512, 2) void edge_dw_sel_f32_kernel(const float* __restrict__ Y, const uint8_t* __restrict__ arg,
                                                                 const float* __restrict__ dpre, int k, long long E, long long Mp,
                                                                 long long rows_per_block, float* __restrict__ slabs)
{
    constexpr int LDB = 272, IMG = 128 * LDB;
    extern __shared__ __attribute__((aligned(16))) unsigned char dwf_lds[];      // D hi | D lo | Y hi | Y lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long m_begin = (long long)blockIdx.x * rows_per_block;
    const long long m_end = min(E, m_begin + rows_per_block);
    const int nchunk = m_end > m_begin ? (int)((m_end - m_begin) / 64) : 0;      // whole chunks: E % 64 == 0 (host check)
    const bool isY = __builtin_amdgcn_readfirstlane(tid) < 256;
    const int pt = tid & 255;
    const int rg = pt & 7, c4 = pt >> 3;           // row group (8 rows) 0..7, channel quad 0..31
    unsigned char* const img_hi = dwf_lds + (isY ? 2 * IMG : 0);
    unsigned char* const img_lo = img_hi + IMG;
    const int wr_off = (c4 * 4) * LDB + rg * 16;   // + 128 * half + channel * LDB
    const int wa = wave & 3, wb = wave >> 2;
    const int rd_a = (wa * 32 + col) * LDB + h * 16;             // + 128 * half + 32 * kstep
    const int rd_b = (wb * 64 + col) * LDB + h * 16;             // + 32 * LDB * j

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    struct Raw { float4 r[8]; };     // Y: 8 row pieces; D: dpre of two points in r[0], r[1]; r[2] = (arg quad a, arg quad b, first slot) as bits
    Raw r0, r1;
    auto load = [&](Raw& R, int chunk) {
        const long long row0 = m_begin + (long long)chunk * 64 + rg * 8;
        if (isY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) R.r[i] = *reinterpret_cast<const float4*>(Y + (row0 + i) * 128 + c4 * 4);
        } else {
            const long long ia = row0 / k;
            const long long ib = ia + 1 < Mp ? ia + 1 : Mp - 1;
            const uint32_t aa = *reinterpret_cast<const uint32_t*>(arg + ia * 128 + c4 * 4);
            const uint32_t ab = *reinterpret_cast<const uint32_t*>(arg + ib * 128 + c4 * 4);
            R.r[0] = *reinterpret_cast<const float4*>(dpre + ia * 128 + c4 * 4);
            R.r[1] = *reinterpret_cast<const float4*>(dpre + ib * 128 + c4 * 4);
            R.r[2] = make_float4(__uint_as_float(aa), __uint_as_float(ab), __int_as_float((int)(row0 - ia * k)), 0.0f);
        }
    };
    auto convert_store = [&](const Raw& R, int half) {
        float v[8][4];
        if (isY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i][0] = R.r[i].x; v[i][1] = R.r[i].y; v[i][2] = R.r[i].z; v[i][3] = R.r[i].w;
#pragma unroll
                for (int c = 0; c < 4; ++c) cs[c] += v[i][c];
            }
        } else {
            const float da[4] = {R.r[0].x, R.r[0].y, R.r[0].z, R.r[0].w}, db[4] = {R.r[1].x, R.r[1].y, R.r[1].z, R.r[1].w};
            const uint32_t aa = __float_as_uint(R.r[2].x), ab = __float_as_uint(R.r[2].y);
            const int ta = __float_as_int(R.r[2].z);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tt = ta + i;
                const bool second = tt >= k;
                const uint32_t t = (uint32_t)(second ? tt - k : tt);
                const uint32_t a = second ? ab : aa;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[i][c] = ((a >> (8 * c)) & 0xffu) == t ? (second ? db[c] : da[c]) : 0.0f;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {              // channel pair (2p, 2p+1) of the quad
            uint32_t wh[8], wl[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v0 = v[i][2 * p], v1 = v[i][2 * p + 1];
                const uint32_t hp = pack_bf16(v0, v1);
                wh[i] = hp;
                wl[i] = pack_bf16(v0 - __uint_as_float(hp << 16), v1 - __uint_as_float(hp & 0xffff0000u));
            }
            uint4 e, o;
            e.x = __builtin_amdgcn_perm(wh[1], wh[0], 0x05040100u); o.x = __builtin_amdgcn_perm(wh[1], wh[0], 0x07060302u);
            e.y = __builtin_amdgcn_perm(wh[3], wh[2], 0x05040100u); o.y = __builtin_amdgcn_perm(wh[3], wh[2], 0x07060302u);
            e.z = __builtin_amdgcn_perm(wh[5], wh[4], 0x05040100u); o.z = __builtin_amdgcn_perm(wh[5], wh[4], 0x07060302u);
            e.w = __builtin_amdgcn_perm(wh[7], wh[6], 0x05040100u); o.w = __builtin_amdgcn_perm(wh[7], wh[6], 0x07060302u);
            const int off = wr_off + half * 128 + (2 * p) * LDB;
            *reinterpret_cast<uint4*>(img_hi + off) = e;
            *reinterpret_cast<uint4*>(img_hi + off + LDB) = o;
            e.x = __builtin_amdgcn_perm(wl[1], wl[0], 0x05040100u); o.x = __builtin_amdgcn_perm(wl[1], wl[0], 0x07060302u);
            e.y = __builtin_amdgcn_perm(wl[3], wl[2], 0x05040100u); o.y = __builtin_amdgcn_perm(wl[3], wl[2], 0x07060302u);
            e.z = __builtin_amdgcn_perm(wl[5], wl[4], 0x05040100u); o.z = __builtin_amdgcn_perm(wl[5], wl[4], 0x07060302u);
            e.w = __builtin_amdgcn_perm(wl[7], wl[6], 0x05040100u); o.w = __builtin_amdgcn_perm(wl[7], wl[6], 0x07060302u);
            *reinterpret_cast<uint4*>(img_lo + off) = e;
            *reinterpret_cast<uint4*>(img_lo + off + LDB) = o;
        }
    };
    const unsigned char* const dh = dwf_lds, * const dl = dwf_lds + IMG, * const yh = dwf_lds + 2 * IMG, * const yl = dwf_lds + 3 * IMG;
    auto mma = [&](int half, int ks0) {            // two of the chunk's four k-steps
#pragma unroll
        for (int ks = ks0; ks < ks0 + 2; ++ks) {
            const int o = half * 128 + ks * 32;
            const bf16x8 adh = *reinterpret_cast<const bf16x8*>(dh + rd_a + o), adl = *reinterpret_cast<const bf16x8*>(dl + rd_a + o);
            const bf16x8 ayh = *reinterpret_cast<const bf16x8*>(yh + rd_a + o), ayl = *reinterpret_cast<const bf16x8*>(yl + rd_a + o);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(yh + rd_b + o + j * 32 * LDB);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(yl + rd_b + o + j * 32 * LDB);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adl, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ayl, bh, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adh, bl, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ayh, bl, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adh, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ayh, bh, acc[1][j], 0, 0, 0);
            }
        }
    };
    auto chunk = [&](int n, const Raw& cur, Raw& nxt) {
        load(nxt, n + 2 < nchunk ? n + 2 : nchunk - 1);        // unconditional: the last chunks are re-read and dropped
        mma(n & 1, 0);
        if (n + 1 < nchunk) convert_store(cur, (n + 1) & 1);
        mma(n & 1, 2);
        __syncthreads();
    };
    if (nchunk > 0) {
        load(r0, 0);
        convert_store(r0, 0);
        load(r0, nchunk > 1 ? 1 : 0);
        __syncthreads();
        for (int n = 0; n < nchunk; n += 2) {
            chunk(n, r0, r1);
            if (n + 1 < nchunk) chunk(n + 1, r1, r0);
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * (256 * 128 + 128);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = i * 128 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)arow * 128 + wb * 64 + j * 32 + col] = acc[i][j][r];
            }
    float* tmp = reinterpret_cast<float*>(dwf_lds);      // column sums: the 8 row-group threads of a channel quad, in a fixed order
    if (isY)
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[(c4 * 8 + rg) * 4 + e] = cs[e];
    __syncthreads();
    if (tid < 128) {
        const int q4 = tid >> 2, e = tid & 3;
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += tmp[(q4 * 8 + r) * 4 + e];
        slab[256 * 128 + tid] = t;
    }
}

// ------------------------------------------------------------------------------------------ (2t)
// The same slab (S = D^T Y | G = Y^T Y | column sums of Y) on ROW-MAJOR LDS images read with ds_read_b64_tr_b16 (cf. gemm_tn_tr_kernel,
// lpd_train2.hip): the rows of Y are copied (bf16) or split into hi / lo (fp32) where they are staged, D is built row by row from (arg,
// dpre) of the row's point -- no register transposes -- and every MFMA operand (one channel, 8 consecutive rows) comes out of two
// transposed reads.  Image rows 256 + 64 bytes apart (the 4 rows of a read on 4 x 64 bytes of different banks), 32-row chunks in two
// buffers, one barrier per chunk, the rows of chunk n + 2 requested at the top of chunk n.  H: bf16 tensors, one product per term; else
// fp32 tensors, three.  Wave (wa, wb): row tiles wa of S and of G, column tiles 2 wb, 2 wb + 1 (the slab layout of the kernels above).
typedef short dw_i16x4 __attribute__((ext_vector_type(4)));
typedef short dw_i16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dw_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 dw_tr_frag(const unsigned char* p0, const unsigned char* p1)
{
    typedef dw_i16x4 __attribute__((address_space(3))) * lds_ptr;
    const dw_i16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const dw_i16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    const dw_i16x8 v = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <bool H>
__global__ __launch_bounds__(512, 2) void edge_dw_sel_tr_kernel(const void* __restrict__ Y_, const uint8_t* __restrict__ arg,
                                                                const void* __restrict__ dpre_, int k, long long E, long long Mp,
                                                                long long rows_per_block, float* __restrict__ slabs)
{
    constexpr int ROWB = 320, IMG = 32 * ROWB;
    constexpr int BUF = (H ? 2 : 4) * IMG;
    constexpr int Y_HI = 0, Y_LO = IMG, D_HI = (H ? 1 : 2) * IMG, D_LO = D_HI + IMG;
    constexpr int NR = H ? 1 : 2;                  // rows of a chunk per staging thread
    extern __shared__ __attribute__((aligned(16))) unsigned char dwt[];       // [2 buffers][Y hi [Y lo] D hi [D lo]][32][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long m_begin = (long long)blockIdx.x * rows_per_block;
    const long long m_end = min(E, m_begin + rows_per_block);
    const int nchunk = m_end > m_begin ? (int)((m_end - m_begin) / 32) : 0;      // whole chunks (host check)
    // staging: fp32 -- rows tid / 32 + 16 p, channel quad tid % 32; bf16 -- row tid / 16, channel octet tid % 16
    const int srow = H ? tid >> 4 : tid >> 5, sq = H ? tid & 15 : tid & 31;
    const float* Y32 = reinterpret_cast<const float*>(Y_);
    const uint16_t* Y16 = reinterpret_cast<const uint16_t*>(Y_);
    const float* dp32 = reinterpret_cast<const float*>(dpre_);
    const uint16_t* dp16 = reinterpret_cast<const uint16_t*>(dpre_);
    const unsigned ku = (unsigned)k;
    struct Raw { float4 y[NR]; dw_u32x4 d[NR]; dw_u32x2 a[NR]; unsigned t[NR]; };      // rows of Y; dpre and arg of the row's point; its slot
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load = [&](Raw& R, int chunk) {
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const long long e = m_begin + (long long)chunk * 32 + srow + 16 * p;
            const unsigned i = (unsigned)e / ku;
            R.t[p] = (unsigned)e - i * ku;
            if constexpr (H) {
                R.y[p] = *reinterpret_cast<const float4*>(Y16 + e * 128 + sq * 8);                        // eight bf16 values as raw bits
                R.d[p] = *reinterpret_cast<const dw_u32x4*>(dp16 + (long long)i * 128 + sq * 8);
                R.a[p] = *reinterpret_cast<const dw_u32x2*>(arg + (long long)i * 128 + sq * 8);
            } else {
                R.y[p] = *reinterpret_cast<const float4*>(Y32 + e * 128 + sq * 4);
                R.d[p] = *reinterpret_cast<const dw_u32x4*>(dp32 + (long long)i * 128 + sq * 4);
                R.a[p][0] = *reinterpret_cast<const unsigned*>(arg + (long long)i * 128 + sq * 4);
                R.a[p][1] = 0u;
            }
        }
    };
    auto split_store = [&](unsigned char* hi, unsigned char* lo, float v0, float v1, float v2, float v3) {
        const uint32_t h01 = pack_bf16(v0, v1), h23 = pack_bf16(v2, v3);
        const uint32_t l01 = pack_bf16(v0 - __uint_as_float(h01 << 16), v1 - __uint_as_float(h01 & 0xffff0000u));
        const uint32_t l23 = pack_bf16(v2 - __uint_as_float(h23 << 16), v3 - __uint_as_float(h23 & 0xffff0000u));
        *reinterpret_cast<uint2*>(hi) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(lo) = make_uint2(l01, l23);
    };
    auto stage = [&](const Raw& R, int buf) {
        unsigned char* base = dwt + buf * BUF;
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int row = srow + 16 * p;
            const unsigned t = R.t[p];
            if constexpr (H) {
                const dw_u32x4 yw = __builtin_bit_cast(dw_u32x4, R.y[p]);
                dw_u32x4 dw;
#pragma unroll
                for (int d = 0; d < 4; ++d) {      // channels 2 d, 2 d + 1 of the octet: bytes 2 d, 2 d + 1 of the arg octet
                    const unsigned a = R.a[p][d >> 1] >> (16 * (d & 1));
                    const unsigned keep = ((a & 0xffu) == t ? 0x0000ffffu : 0u) | (((a >> 8) & 0xffu) == t ? 0xffff0000u : 0u);
                    dw[d] = R.d[p][d] & keep;
                    cs[2 * d] += __uint_as_float(yw[d] << 16);
                    cs[2 * d + 1] += __uint_as_float(yw[d] & 0xffff0000u);
                }
                *reinterpret_cast<dw_u32x4*>(base + Y_HI + row * ROWB + sq * 16) = yw;
                *reinterpret_cast<dw_u32x4*>(base + D_HI + row * ROWB + sq * 16) = dw;
            } else {
                const float4 y = R.y[p];
                cs[0] += y.x; cs[1] += y.y; cs[2] += y.z; cs[3] += y.w;
                split_store(base + Y_HI + row * ROWB + sq * 8, base + Y_LO + row * ROWB + sq * 8, y.x, y.y, y.z, y.w);
                const unsigned a = R.a[p][0];
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ((a >> (8 * c)) & 0xffu) == t ? __uint_as_float(R.d[p][c]) : 0.0f;
                split_store(base + D_HI + row * ROWB + sq * 8, base + D_LO + row * ROWB + sq * 8, v[0], v[1], v[2], v[3]);
            }
        }
    };
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int rd = ((g >> 1) * 8 + q) * ROWB + ((g & 1) * 16 + pp * 4) * 2;
    const int wa = wave & 3, wb = wave >> 2;
    const int rd_a = rd + wa * 64, rd_b = rd + wb * 128;          // + 64 j
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto mma = [&](int buf, int s) {
        const unsigned char* base = dwt + buf * BUF + s * 16 * ROWB;
        const bf16x8 adh = dw_tr_frag(base + D_HI + rd_a, base + D_HI + rd_a + 4 * ROWB);
        const bf16x8 ayh = dw_tr_frag(base + Y_HI + rd_a, base + Y_HI + rd_a + 4 * ROWB);
        bf16x8 adl, ayl;
        if constexpr (!H) {
            adl = dw_tr_frag(base + D_LO + rd_a, base + D_LO + rd_a + 4 * ROWB);
            ayl = dw_tr_frag(base + Y_LO + rd_a, base + Y_LO + rd_a + 4 * ROWB);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 bh = dw_tr_frag(base + Y_HI + rd_b + 64 * j, base + Y_HI + rd_b + 64 * j + 4 * ROWB);
            if constexpr (!H) {
                const bf16x8 bl = dw_tr_frag(base + Y_LO + rd_b + 64 * j, base + Y_LO + rd_b + 64 * j + 4 * ROWB);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adl, bh, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ayl, bh, acc[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adh, bl, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ayh, bl, acc[1][j], 0, 0, 0);
            }
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adh, bh, acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ayh, bh, acc[1][j], 0, 0, 0);
        }
    };
    Raw r0, r1;
    auto chunk = [&](int n, const Raw& cur, Raw& nxt) {
        load(nxt, n + 2 < nchunk ? n + 2 : nchunk - 1);        // unconditional: the last chunks are re-read and dropped
        mma(n & 1, 0);
        if (n + 1 < nchunk) stage(cur, (n + 1) & 1);
        mma(n & 1, 1);
        __syncthreads();
    };
    if (nchunk > 0) {
        load(r0, 0);
        stage(r0, 0);
        load(r0, nchunk > 1 ? 1 : 0);
        __syncthreads();
        for (int n = 0; n < nchunk; n += 2) {
            chunk(n, r0, r1);
            if (n + 1 < nchunk) chunk(n + 1, r1, r0);
        }
    }
    float* slab = slabs + (size_t)blockIdx.x * (256 * 128 + 128);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int arow = i * 128 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[(size_t)arow * 128 + wb * 64 + j * 32 + col] = acc[i][j][r];
            }
    // column sums of Y: the threads that share a channel group, in a fixed order
    __syncthreads();
    float* tmp = reinterpret_cast<float*>(dwt);
    constexpr int CPT = H ? 8 : 4, NG = H ? 16 : 32, NRG = 512 / NG;      // channels per thread, channel groups, threads per group
#pragma unroll
    for (int e = 0; e < CPT; ++e) tmp[(sq * NRG + srow) * CPT + e] = cs[e];
    __syncthreads();
    if (tid < 128) {
        const int grp = tid / CPT, e = tid % CPT;
        float t = 0.0f;
        for (int r = 0; r < NRG; ++r) t += tmp[(grp * NRG + r) * CPT + e];
        slab[256 * 128 + tid] = t;
    }
}

// (3f) dY [E][128] (fp32) = dZ W2 with dZ generated in the loader from fp32 z: A = z a1 + a0 + delta dpre, split hi + lo, three products
// with the staged weight rows s_c W2[c][:] (hi + lo).  512 threads share the weight images; a wave owns 32 rows and all 128 columns
// (transposed tile); the result leaves through a wave-private LDS tile as whole 128-byte rows (cf. lpd_gemm_x3t_rows).
__global__ __launch_bounds__(512, 2) void gemm_f32s_bnbwd_kernel(const float* __restrict__ Z, const uint8_t* __restrict__ arg,
                                                                 const float* __restrict__ dpre, int k, const float* __restrict__ W2, int ldw,
                                                                 const float* __restrict__ scale, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const double* __restrict__ dbeta,
                                                                 const double* __restrict__ dgamma, double count, float* __restrict__ Cout,
                                                                 long long E, long long Mp)
{
    constexpr int K = 128, N = 128, KS = K / 16, NT = N / 32, LDW = K + 8;
    constexpr int PB = 640;                         // bytes of one point in the side buffer: dpre (512) | arg (128)
    extern __shared__ __attribute__((aligned(16))) __bf16 wimg4[];   // hi, lo images | float2 consts [K] | per wave: side buffers, staging tile
    __bf16* whi = wimg4;
    __bf16* wlo = wimg4 + N * LDW;
    float2* cst = reinterpret_cast<float2*>(wimg4 + 2 * N * LDW);
    unsigned char* wbase = reinterpret_cast<unsigned char*>(cst + K) + (threadIdx.x >> 6) * (2 * 3 * PB + 32 * 36 * 4);
    unsigned char* side = wbase;
    float* st = reinterpret_cast<float*>(wbase + 2 * 3 * PB);
    const int tid = threadIdx.x;
    for (int e = tid; e < N * K; e += 512) {       // W(n, kk) = s_kk W2[kk][n]
        const int n = e % N, kk = e / N;
        const float w = W2[(size_t)kk * ldw + n] * scale[kk];
        const __bf16 hi = (__bf16)w;
        const __bf16 lo = (__bf16)(w - (float)hi);
        whi[n * LDW + kk] = hi;
        wlo[n * LDW + kk] = lo;
    }
    if (tid < K) {
        const float m1 = (float)(dbeta[tid] / count), m2 = (float)(dgamma[tid] / count);
        cst[tid] = make_float2(-invstd[tid] * m2, mean[tid] * invstd[tid] * m2 - m1);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const long long ntile = (E + 31) / 32;
    const long long stride = (long long)gridDim.x * 8;
    long long tile = (long long)blockIdx.x * 8 + wave;
    float4 zc[16], zn[16], sd[2], sa;
    // side data of a tile: lanes 0..47 stage dpre (point L / 16, 32 bytes L % 16), lanes 0..23 arg (point L / 8, 16 bytes L % 8)
    const int sd_p = (lane >> 4) < 3 ? (lane >> 4) : 2, sd_q = lane & 15;
    const int sa_p = ((lane & 31) >> 3) < 3 ? ((lane & 31) >> 3) : 2, sa_q = lane & 7;
    auto load_side = [&](long long tl) {
        const long long i0 = (tl * 32 < E ? tl * 32 : E - 1) / k;
        const long long pd = i0 + sd_p < Mp ? i0 + sd_p : Mp - 1, pa = i0 + sa_p < Mp ? i0 + sa_p : Mp - 1;
        sd[0] = *reinterpret_cast<const float4*>(dpre + pd * K + sd_q * 8);
        sd[1] = *reinterpret_cast<const float4*>(dpre + pd * K + sd_q * 8 + 4);
        sa = *reinterpret_cast<const float4*>(arg + pa * K + sa_q * 16);
    };
    auto store_side = [&](int buf) {
        unsigned char* b = side + buf * (3 * PB);
        if (lane < 48) {
            *reinterpret_cast<float4*>(b + sd_p * PB + sd_q * 32) = sd[0];
            *reinterpret_cast<float4*>(b + sd_p * PB + sd_q * 32 + 16) = sd[1];
        }
        if (lane < 24) *reinterpret_cast<float4*>(b + sa_p * PB + 512 + sa_q * 16) = sa;
    };
    auto load_z = [&](long long tl, float4 (&z)[16]) {
        const long long row = tl * 32 + col;
        const float* zp = Z + (row < E ? row : E - 1) * K + h * (K / 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) z[u] = *reinterpret_cast<const float4*>(zp + u * 4);
    };
    int buf = 0;
    if (tile < ntile) {
        load_side(tile);
        load_z(tile, zc);
        store_side(0);
    }
    float* const stw = st + col * 36 + 4 * h;
    const float* const str = st + (lane >> 3) * 36 + (lane & 7) * 4;
    for (; tile < ntile; tile += stride, buf ^= 1) {
        const long long row = tile * 32 + col;
        const long long rr = row < E ? row : E - 1;
        const long long i0 = (tile * 32 < E ? tile * 32 : E - 1) / k;
        const long long ip = rr / k;
        const uint32_t t = (uint32_t)(rr - ip * k);
        const unsigned char* sb = side + buf * (3 * PB) + (int)(ip - i0) * PB;     // this lane's point: dpre [128] fp32, arg [128] at + 512
        const long long tn = tile + stride < ntile ? tile + stride : tile;
        load_side(tn);                              // first, then the z rows of the next tile (in-order vmcnt, see gemm_bf16s_bnbwd_kernel)
        load_z(tn, zn);
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {           // channels h 64 + 8 s2 .. + 7
            const float zv[8] = {zc[2 * s2].x, zc[2 * s2].y, zc[2 * s2].z, zc[2 * s2].w, zc[2 * s2 + 1].x, zc[2 * s2 + 1].y, zc[2 * s2 + 1].z, zc[2 * s2 + 1].w};
            const int c0 = h * (K / 2) + s2 * 8;
            const float4 d0 = *reinterpret_cast<const float4*>(sb + c0 * 4), d1 = *reinterpret_cast<const float4*>(sb + c0 * 4 + 16);
            const uint2 a2 = *reinterpret_cast<const uint2*>(sb + 512 + c0);
            const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            uint32_t oh[4], ol[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float4 c2 = *reinterpret_cast<const float4*>(&cst[c0 + 2 * d]);       // (a1, a0) of channels 2d, 2d + 1
                const uint32_t ab = ((d < 2 ? a2.x : a2.y) >> (16 * (d & 1))) & 0xffffu;
                const float v0 = fmaf(zv[2 * d], c2.x, c2.y) + ((ab & 0xffu) == t ? dv[2 * d] : 0.0f);
                const float v1 = fmaf(zv[2 * d + 1], c2.z, c2.w) + ((ab >> 8) == t ? dv[2 * d + 1] : 0.0f);
                const uint32_t hp = pack_bf16(v0, v1);
                oh[d] = hp;
                ol[d] = pack_bf16(v0 - __uint_as_float(hp << 16), v1 - __uint_as_float(hp & 0xffff0000u));
            }
            const bf16x8 avh = __builtin_bit_cast(bf16x8, make_uint4(oh[0], oh[1], oh[2], oh[3]));
            const bf16x8 avl = __builtin_bit_cast(bf16x8, make_uint4(ol[0], ol[1], ol[2], ol[3]));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int off = (j * 32 + col) * LDW + c0;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(whi + off);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wlo + off);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, avh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, avl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, avh, acc[j], 0, 0, 0);
            }
        }
        store_side(buf ^ 1);                        // the next tile's side data (this wave's own buffer: no barrier)
        // rows of the tile through the wave's LDS tile: lane (row col, h) holds columns 32 j + 8 g + 4 h .. + 3
        const long long r0w = tile * 32 + (lane >> 3);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stw + g * 8) = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const long long rw = r0w + p * 8;
                const float4 v = *reinterpret_cast<const float4*>(str + p * 8 * 36);
                if (rw < E) *reinterpret_cast<float4*>(Cout + rw * N + j * 32 + (lane & 7) * 4) = v;
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) zc[u] = zn[u];
    }
}

}  // namespace

static int bn_sel_bwd_reduce_impl(const float* dOut, long long ldo, const float* Xsel, long long ldsel, long long M, int C, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, int act, float slope, uint16_t* dpre16,
                                  float* dpre32, double* dbeta, double* dgamma, double* stat_ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(dOut && Xsel && scale && shift && mean && invstd && (dpre16 || dpre32) && dbeta && dgamma && M > 0, "lpd_bn_sel_bwd_reduce: null pointer");
    LPD_CHECK_ARG(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && ldo % 4 == 0 && ldsel % 4 == 0, "lpd_bn_sel_bwd_reduce: bad dims");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_bn_sel_bwd_reduce: activation %d unsupported", act);
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_bn_sel_bwd_reduce: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_bn_sel_bwd_reduce", C);
    const int rg = 256 / (C / 4);
    hipLaunchKernelGGL(bn_sel_bwd_reduce_kernel, dim3(lpd_reduce_grid(grid_for(M, rg * 4, 4096))), dim3(256), 0, stream, dOut, ldo, Xsel, ldsel, M, C, scale,
                       shift, mean, invstd, act == 0 ? 1.0f : (act == 1 ? 0.0f : slope), dpre16, dpre32, ws.sum(), ws.sumsq());
    LPD_CHECK_LAUNCH("lpd_bn_sel_bwd_reduce");
    return lpd_stat_finish(ws, dbeta, dgamma, C, stream);
}

extern "C" int lpd_bn_sel_bwd_reduce(const float* dOut, long long ldo, const float* Xsel, long long ldsel, long long M, int C,
                                     const float* scale, const float* shift, const float* mean, const float* invstd, int act, float slope,
                                     uint16_t* dpre16, double* dbeta, double* dgamma, double* stat_ws, void* stream)
{
    return bn_sel_bwd_reduce_impl(dOut, ldo, Xsel, ldsel, M, C, scale, shift, mean, invstd, act, slope, dpre16, nullptr, dbeta, dgamma, stat_ws, stream);
}

extern "C" int lpd_bn_sel_bwd_reduce_f32(const float* dOut, long long ldo, const float* Xsel, long long ldsel, long long M, int C,
                                         const float* scale, const float* shift, const float* mean, const float* invstd, int act, float slope,
                                         float* dpre, double* dbeta, double* dgamma, double* stat_ws, void* stream)
{
    return bn_sel_bwd_reduce_impl(dOut, ldo, Xsel, ldsel, M, C, scale, shift, mean, invstd, act, slope, nullptr, dpre, dbeta, dgamma, stat_ws, stream);
}

// LPD_DEBUG=dw-sel-tr=0: the register-transposing kernels
static bool edge_dw_sel_tr()
{
    static const bool on = lpd_debug("dw-sel-tr", 1) != 0;
    return on;
}

static long long edge_dw_sel_blocks(long long E)
{
    long long b = E / 4096;
    return b > 256 ? 256 : (b < 1 ? 1 : b);       // one 512-thread block per CU
}

extern "C" long long lpd_edge_dw_sel_bf16_ws_bytes(long long E)
{
    return edge_dw_sel_blocks(E) * (256 * 128 + 128) * (long long)sizeof(float) + (256 * 128 + 128) * (long long)sizeof(double);
}

extern "C" int lpd_edge_dw_sel_bf16(const uint16_t* Y, const uint8_t* arg, const uint16_t* dpre16, int k, long long M, const float* W2,
                                    long long ldw, const float* scale, const float* mean, const float* invstd, const double* dbeta,
                                    const double* dgamma, float* dW2, void* ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(Y && arg && dpre16 && W2 && scale && mean && invstd && dbeta && dgamma && dW2 && ws && M > 0, "lpd_edge_dw_sel_bf16: null pointer");
    LPD_CHECK_ARG(k >= 8 && k <= 255 && ldw >= 128, "lpd_edge_dw_sel_bf16: 8 <= k <= 255 and ldw >= 128 required");
    LPD_CHECK_ARG((M * k) % 128 == 0, "lpd_edge_dw_sel_bf16: M * k must be a multiple of 128 (got %lld)", M * k);
    LPD_CHECK_ARG((((uintptr_t)Y | (uintptr_t)dpre16 | (uintptr_t)arg | (uintptr_t)ws) & 15) == 0, "lpd_edge_dw_sel_bf16: pointers must be 16-byte aligned");
    const long long E = M * k;
    const long long blocks = edge_dw_sel_blocks(E);
    long long rpb = (E + blocks - 1) / blocks;
    rpb = (rpb + 127) / 128 * 128;
    constexpr int n = 256 * 128 + 128;
    constexpr int lds = 2 * 128 * 528;
    double* red = reinterpret_cast<double*>(ws);
    float* slabs = reinterpret_cast<float*>(red + n);
    // (bf16 tensors: measured 343 us against 326 on the register-transposing kernel -- one product per term, the staging is copies
    //  either way; the transposed-read kernel runs for LPD_DEBUG=dw-sel-tr=2 only)
    static const bool tr16 = lpd_debug("dw-sel-tr", 1) == 2;
    if (tr16 && E < (1ll << 32)) {
        constexpr int lds_tr = 2 * 2 * 32 * 320;
        (void)hipFuncSetAttribute((const void*)edge_dw_sel_tr_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_tr);
        hipLaunchKernelGGL(edge_dw_sel_tr_kernel<true>, dim3((unsigned)blocks), dim3(512), lds_tr, stream, Y, arg, dpre16, k, E, M, rpb, slabs);
    } else {
        (void)hipFuncSetAttribute((const void*)edge_dw_sel_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(edge_dw_sel_bf16_kernel, dim3((unsigned)blocks), dim3(512), lds, stream, Y, arg, dpre16, k, E, M, rpb, slabs);
    }
    LPD_CHECK_LAUNCH("lpd_edge_dw_sel_bf16");
    static_assert(n % 64 == 0, "slab_reduce_kernel: 64 elements per block");
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(n / 64), dim3(256), 0, stream, (const float*)slabs, red, n, (int)blocks);
    LPD_CHECK_LAUNCH("lpd_edge_dw_sel_bf16(reduce)");
    hipLaunchKernelGGL(dw2_finish_kernel, dim3(128), dim3(128), 0, stream, (const double*)red, W2, ldw, scale, mean, invstd, dbeta, dgamma,
                       (double)M * (double)k, dW2);
    LPD_CHECK_LAUNCH("lpd_edge_dw_sel_bf16(finish)");
    return LPD_OK;
}

extern "C" int lpd_gemm_bf16s_bnbwd(const uint16_t* Z, const uint8_t* arg, const uint16_t* dpre16, int k, long long M, const float* W2, int ldw,
                                    const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma,
                                    uint16_t* dY, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(Z && arg && dpre16 && W2 && scale && mean && invstd && dbeta && dgamma && dY && M > 0, "lpd_gemm_bf16s_bnbwd: null pointer");
    LPD_CHECK_ARG(k >= 16 && k <= 255 && ldw >= 128, "lpd_gemm_bf16s_bnbwd: 16 <= k <= 255 and ldw >= 128 required");
    LPD_CHECK_ARG((((uintptr_t)Z | (uintptr_t)dpre16 | (uintptr_t)arg | (uintptr_t)dY) & 15) == 0, "lpd_gemm_bf16s_bnbwd: pointers must be 16-byte aligned");
    const long long E = M * k;
    constexpr int lds = 2 * 128 * (128 + 8) * 2 + 128 * 8 + 4 * 2 * 3 * 384;
    (void)hipFuncSetAttribute((const void*)gemm_bf16s_bnbwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long tiles = (E + 31) / 32;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(gemm_bf16s_bnbwd_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, Z, arg, dpre16, k, W2, ldw, scale, mean, invstd,
                       dbeta, dgamma, (double)M * (double)k, dY, E, M);
    LPD_CHECK_LAUNCH("lpd_gemm_bf16s_bnbwd");
    return LPD_OK;
}

// fp32 storage: the same two steps on fp32 tensors (split-bf16 products).  (M * k) % 64 == 0, 16 <= k <= 255.
extern "C" int lpd_edge_dw_sel_f32(const float* Y, const uint8_t* arg, const float* dpre, int k, long long M, const float* W2, long long ldw,
                                   const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma,
                                   float* dW2, void* ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(Y && arg && dpre && W2 && scale && mean && invstd && dbeta && dgamma && dW2 && ws && M > 0, "lpd_edge_dw_sel_f32: null pointer");
    LPD_CHECK_ARG(k >= 8 && k <= 255 && ldw >= 128, "lpd_edge_dw_sel_f32: 8 <= k <= 255 and ldw >= 128 required");
    LPD_CHECK_ARG((M * k) % 64 == 0, "lpd_edge_dw_sel_f32: M * k must be a multiple of 64 (got %lld)", M * k);
    LPD_CHECK_ARG((((uintptr_t)Y | (uintptr_t)dpre | (uintptr_t)arg | (uintptr_t)ws) & 15) == 0, "lpd_edge_dw_sel_f32: pointers must be 16-byte aligned");
    const long long E = M * k;
    const long long blocks = edge_dw_sel_blocks(E);
    long long rpb = (E + blocks - 1) / blocks;
    rpb = (rpb + 63) / 64 * 64;
    constexpr int n = 256 * 128 + 128;
    constexpr int lds = 4 * 128 * 272;
    double* red = reinterpret_cast<double*>(ws);
    float* slabs = reinterpret_cast<float*>(red + n);
    if (edge_dw_sel_tr() && E < (1ll << 32)) {
        constexpr int lds_tr = 2 * 4 * 32 * 320;
        (void)hipFuncSetAttribute((const void*)edge_dw_sel_tr_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_tr);
        hipLaunchKernelGGL(edge_dw_sel_tr_kernel<false>, dim3((unsigned)blocks), dim3(512), lds_tr, stream, Y, arg, dpre, k, E, M, rpb, slabs);
    } else {
        (void)hipFuncSetAttribute((const void*)edge_dw_sel_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(edge_dw_sel_f32_kernel, dim3((unsigned)blocks), dim3(512), lds, stream, Y, arg, dpre, k, E, M, rpb, slabs);
    }
    LPD_CHECK_LAUNCH("lpd_edge_dw_sel_f32");
    static_assert(n % 64 == 0, "slab_reduce_kernel: 64 elements per block");
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(n / 64), dim3(256), 0, stream, (const float*)slabs, red, n, (int)blocks);
    LPD_CHECK_LAUNCH("lpd_edge_dw_sel_f32(reduce)");
    hipLaunchKernelGGL(dw2_finish_kernel, dim3(128), dim3(128), 0, stream, (const double*)red, W2, ldw, scale, mean, invstd, dbeta, dgamma,
                       (double)M * (double)k, dW2);
    LPD_CHECK_LAUNCH("lpd_edge_dw_sel_f32(finish)");
    return LPD_OK;
}

extern "C" int lpd_gemm_f32s_bnbwd(const float* Z, const uint8_t* arg, const float* dpre, int k, long long M, const float* W2, int ldw,
                                   const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma,
                                   float* dY, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(Z && arg && dpre && W2 && scale && mean && invstd && dbeta && dgamma && dY && M > 0, "lpd_gemm_f32s_bnbwd: null pointer");
    LPD_CHECK_ARG(k >= 16 && k <= 255 && ldw >= 128, "lpd_gemm_f32s_bnbwd: 16 <= k <= 255 and ldw >= 128 required");
    LPD_CHECK_ARG((((uintptr_t)Z | (uintptr_t)dpre | (uintptr_t)arg | (uintptr_t)dY) & 15) == 0, "lpd_gemm_f32s_bnbwd: pointers must be 16-byte aligned");
    const long long E = M * k;
    constexpr int lds = 2 * 128 * (128 + 8) * 2 + 128 * 8 + 8 * (2 * 3 * 640 + 32 * 36 * 4);
    (void)hipFuncSetAttribute((const void*)gemm_f32s_bnbwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long tiles = (E + 31) / 32;
    long long blocks = (tiles + 7) / 8;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(gemm_f32s_bnbwd_kernel, dim3((unsigned)blocks), dim3(512), lds, stream, Z, arg, dpre, k, W2, ldw, scale, mean, invstd,
                       dbeta, dgamma, (double)M * (double)k, dY, E, M);
    LPD_CHECK_LAUNCH("lpd_gemm_f32s_bnbwd");
    return LPD_OK;
}
