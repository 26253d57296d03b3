// lpd_gemm_t.hip -- split-bf16 product of cloud-panel activations with a small weight matrix, computed TRANSPOSED:
//     C_panels[M][N] = act((A_panels[M][K] . W[N][K]^T + bias) * scale + shift),   K = 64 or 128.
//
// The neighbour / centre projection of the split SN1 edge convolution (util/lpdnet_model.py:257: 128 -> 2 x 256 columns for
// every point) writes four times the bytes it reads and multiplies over a SHORT reduction: on the 128 x 128 block tile of
// lpd_gemm_bf16x3 it spends its time in the two barriers per 32-deep k-tile (four k-tiles in all), in re-splitting A for
// each of the four column blocks and in an epilogue of 64 single-float stores per lane that fill a 128-byte line in four
// instructions (164 us at 32 clouds, 15 % MFMA busy, HBM floor 42 us).
//
// Here the MFMA computes C^T: the WEIGHTS are the A operand (rows of the MFMA tile = output columns n; the fragments
// lpd_gemm_prep_b writes have exactly that shape) and the DATA rows are the B operand (columns of the tile = points m).
//   * result: lane (m, h) holds columns 8 q + 4 h + {0..3} of row m -- four consecutive floats of panel q.  One float4 store
//     per lane and (tile, q); the 64 lanes of the instruction write 32 rows x 32 bytes = ONE contiguous KiB of the panel.
//   * the workgroup's 128 rows x K are read once (a lane's float4 of every panel), split hi / lo and kept in LDS for all N
//     columns ([row][K + 8] bf16 images, conflict-free ds_read_b128 operand fetches); ONE barrier in the whole kernel.
//   * each wave owns every fourth 32-column tile and holds that tile's weight fragments (K / 16 k-steps, hi and lo) in
//     registers while it multiplies them with the four row tiles; the next tile's fragments are requested before.  Weight
//     traffic from L2: N K 4 bytes per workgroup.  (First version: a wave kept its 32 data rows in registers instead and
//     streamed ALL fragments, no LDS at all: 121 us, of which 50 us were the 1.07 GB of fragment reads from L2 -- with the
//     fragment address pinned the same kernel took 72 us.)
// Summation order: k ascending, per term lo.hi, hi.lo, hi.hi (the order of the other split-bf16 kernels' k-steps).
#include "lpd_common.h"

namespace {

typedef __bf16 xt_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float xt_f32x2 __attribute__((ext_vector_type(2)));

// hi / lo split of eight consecutive channels on packed pairs (2.5 instructions per element; see lpd_edge.hip)
__device__ __forceinline__ void xt_split8(const float4& a, const float4& b, xt_bf16x8& hi, xt_bf16x8& lo)
{
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned hu[4], lu[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const xt_bf16x2 h2 = __builtin_convertvector((xt_f32x2){x[2 * p], x[2 * p + 1]}, xt_bf16x2);
        const unsigned u = __builtin_bit_cast(unsigned, h2);
        const float r0 = x[2 * p] - __uint_as_float(u << 16), r1 = x[2 * p + 1] - __uint_as_float(u & 0xffff0000u);
        const xt_bf16x2 l2 = __builtin_convertvector((xt_f32x2){r0, r1}, xt_bf16x2);
        hu[p] = u;
        lu[p] = __builtin_bit_cast(unsigned, l2);
    }
    const uint4 hv = make_uint4(hu[0], hu[1], hu[2], hu[3]), lv = make_uint4(lu[0], lu[1], lu[2], lu[3]);
    hi = __builtin_bit_cast(xt_bf16x8, hv);
    lo = __builtin_bit_cast(xt_bf16x8, lv);
}

struct X3tArgs {
    const float* A;
    const __bf16* fhi;
    const __bf16* flo;
    float* C;
    int M, N;
    const float* bias;
    const float* scale;
    const float* shift;
    float ns;                      // negative-side slope of the piecewise-linear activation (1 none, 0 ReLU, slope LeakyReLU)
    long long a_cloud, c_cloud;    // floats between clouds
    int panel_n, panel_ld;
    long long a_lo;                // != 0: A is the hi plane of a pair of split bf16 planes (a_cloud in elements), lo plane a_lo behind
    long long lda, ldc;            // ROWS form: A [M][lda], C [M][ldc] row-major fp32 (no panels)
    long long sA, sC, sF;          // ROWS form, batched over blockIdx.y: floats between the problems' A / C, bf16 elements between their fragments
    int c_bf16;                    // ROWS form: C holds bf16 values (ldc / sC in bf16 elements)
};

constexpr int XT_THREADS = 256;    // 128 rows per workgroup (a panel cloud is a multiple of 128 rows)

// hi / lo split of four consecutive channels (one float4 of a panel row)
__device__ __forceinline__ void xt_split4(const float4& a, uint2& hi, uint2& lo)
{
    const xt_bf16x2 h01 = __builtin_convertvector((xt_f32x2){a.x, a.y}, xt_bf16x2), h23 = __builtin_convertvector((xt_f32x2){a.z, a.w}, xt_bf16x2);
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    const float r0 = a.x - __uint_as_float(u01 << 16), r1 = a.y - __uint_as_float(u01 & 0xffff0000u);
    const float r2 = a.z - __uint_as_float(u23 << 16), r3 = a.w - __uint_as_float(u23 & 0xffff0000u);
    const xt_bf16x2 l01 = __builtin_convertvector((xt_f32x2){r0, r1}, xt_bf16x2), l23 = __builtin_convertvector((xt_f32x2){r2, r3}, xt_bf16x2);
    hi = make_uint2(u01, u23);
    lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

// RB (ROWS form only): rows per workgroup.  At K = 128 the two 128-row images (68 KiB) and the staging tiles (18 KiB) left ONE
// workgroup = one wave per SIMD on a CU: the NetVLAD backward's gradient of the conv3 map ([4096 x 1024 x 128] x 44, bf16 result)
// took 184 us for 461 MB.  64-row workgroups (53 KiB, 208 registers) run two per CU: 164 us.
template <int KS, bool ROWS, int RB = 128>
__global__ __launch_bounds__(XT_THREADS, 2) void gemm_x3t_kernel(X3tArgs g)
{
    static_assert(RB == 128 || (ROWS && RB == 64), "panel operands: 128 rows per workgroup");
    constexpr int RT = RB / 32;                  // 32-row tiles per workgroup (each wave multiplies all of them by its column tiles)
    constexpr int K = KS * 16, LDK = K + 8, IMG = RB * LDK;
    extern __shared__ __attribute__((aligned(16))) __bf16 img[];     // [hi | lo][128 rows][LDK] (+ ROWS: a [32][36] fp32 staging tile per wave)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int col = lane & 31;
    const int m0 = lpd_xcd_remap(blockIdx.x, gridDim.x) * RB;
    const int cloud = ROWS ? 0 : m0 / g.panel_n;
    const int mc0 = ROWS ? m0 : m0 - cloud * g.panel_n;  // first row of the workgroup inside its cloud (ROWS: in the matrix)
    const float* A = ROWS ? g.A + (long long)blockIdx.y * g.sA : g.A + (long long)cloud * g.a_cloud;
    float* C = ROWS ? g.C + (long long)blockIdx.y * g.sC + (long long)(mc0 + col) * g.ldc + 4 * h
                    : g.C + (long long)cloud * g.c_cloud + (long long)(mc0 + col) * 8 + 4 * h;

    // ---- weight fragments of this wave's first tile: fragment (tile nt, k-step s) = 1 KiB at ((nt * KS + s) * 64 + lane) * 8 ----
    const int NT = g.N >> 5;
    const __bf16* fh = g.fhi + (ROWS ? (long long)blockIdx.y * g.sF : 0) + (long long)lane * 8;
    const __bf16* fl = g.flo + (ROWS ? (long long)blockIdx.y * g.sF : 0) + (long long)lane * 8;
    xt_bf16x8 w_hi[KS], w_lo[KS];
    auto load_w = [&](int nt, int s) {       // one register set: step s of the NEXT tile is requested right behind the MFMAs of step s
        nt = nt < NT ? nt : NT - 1;           // past the end: any valid tile (never multiplied)
        const long long off = ((long long)nt * KS + s) * 512;
        w_hi[s] = *reinterpret_cast<const xt_bf16x8*>(fh + off);
        w_lo[s] = *reinterpret_cast<const xt_bf16x8*>(fl + off);
    };
#pragma unroll
    for (int s = 0; s < KS; ++s) load_w(wave, s);

    // ---- the 128 rows, split once: thread -> (row tid / 2, float4 half tid % 2) of every panel ----
    if constexpr (ROWS) {         // row-major A: a wave's lanes along the row (K / 4 float4 per row), 256 / (K / 4) rows per pass
        constexpr int Q4 = K / 4, RPP = XT_THREADS / Q4;
        const int c4 = tid % Q4, r0 = tid / Q4;
        const float* src = A + (long long)(mc0 + r0) * g.lda + c4 * 4;
        float4 v[RB / RPP];
#pragma unroll
        for (int p = 0; p < RB / RPP; ++p) v[p] = *reinterpret_cast<const float4*>(src + (long long)p * RPP * g.lda);
#pragma unroll
        for (int p = 0; p < RB / RPP; ++p) {
            uint2 hh, ll;
            xt_split4(v[p], hh, ll);
            *reinterpret_cast<uint2*>(img + (p * RPP + r0) * LDK + c4 * 4) = hh;
            *reinterpret_cast<uint2*>(img + IMG + (p * RPP + r0) * LDK + c4 * 4) = ll;
        }
    } else if (g.a_lo) {          // pre-split planes (the producer wrote hi / lo): straight copies
        const int row = tid >> 1, half = tid & 1;
        const __bf16* src = reinterpret_cast<const __bf16*>(g.A) + (long long)cloud * g.a_cloud + (long long)(mc0 + row) * 8 + half * 4;
        uint2 vh[K / 8], vl[K / 8];
#pragma unroll
        for (int p = 0; p < K / 8; ++p) {
            vh[p] = *reinterpret_cast<const uint2*>(src + (long long)p * g.panel_ld * 8);
            vl[p] = *reinterpret_cast<const uint2*>(src + g.a_lo + (long long)p * g.panel_ld * 8);
        }
#pragma unroll
        for (int p = 0; p < K / 8; ++p) {
            *reinterpret_cast<uint2*>(img + row * LDK + p * 8 + half * 4) = vh[p];
            *reinterpret_cast<uint2*>(img + IMG + row * LDK + p * 8 + half * 4) = vl[p];
        }
    } else {
        const int row = tid >> 1, half = tid & 1;
        const float* src = A + (long long)(mc0 + row) * 8 + half * 4;
        float4 v[K / 8];
#pragma unroll
        for (int p = 0; p < K / 8; ++p) v[p] = *reinterpret_cast<const float4*>(src + (long long)p * g.panel_ld * 8);
#pragma unroll
        for (int p = 0; p < K / 8; ++p) {
            uint2 hh, ll;
            xt_split4(v[p], hh, ll);
            *reinterpret_cast<uint2*>(img + row * LDK + p * 8 + half * 4) = hh;
            *reinterpret_cast<uint2*>(img + IMG + row * LDK + p * 8 + half * 4) = ll;
        }
    }
    __syncthreads();

    const __bf16* dh = img + col * LDK + h * 8;
    const bool plain = !g.bias && !g.scale && g.ns == 1.0f;
    // ROWS: a row of a 32 x 32 result tile sits in two lanes, four columns at a time -- stored directly, an instruction writes 32-byte
    // pieces of 32 rows (180224 x 512 x 128: 217 us, 1.7 TB/s of writes).  The tile goes through a wave-private LDS tile instead and
    // leaves as 8 whole 128-byte rows per instruction.
    float* const st = reinterpret_cast<float*>(img + 2 * IMG) + wave * (32 * 36);
    float* const stw = st + col * 36 + 4 * h;                 // this lane's columns 8 q + 4 h .. + 3 of row col
    const float* const str = st + (lane >> 3) * 36 + (lane & 7) * 4;
    float* const Crow = ROWS ? g.C + (long long)blockIdx.y * g.sC + (long long)(mc0 + (lane >> 3)) * g.ldc + (lane & 7) * 4 : nullptr;
    // bf16 C: a row of the tile is 64 bytes -- a lane takes 8 columns (16 bytes) of row lane / 4, an instruction writes 16 whole rows
    const float* const str16 = st + (lane >> 2) * 36 + (lane & 3) * 8;
    uint16_t* const Crow16 = ROWS ? reinterpret_cast<uint16_t*>(g.C) + (long long)blockIdx.y * g.sC + (long long)(mc0 + (lane >> 2)) * g.ldc + (lane & 3) * 8
                                  : nullptr;
    auto rows_out = [&](int nt, int i) {
        if (g.c_bf16) {                       // uniform
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float4 lo4 = *reinterpret_cast<const float4*>(str16 + p * 16 * 36), hi4 = *reinterpret_cast<const float4*>(str16 + p * 16 * 36 + 4);
                const xt_bf16x2 w0 = __builtin_convertvector((xt_f32x2){lo4.x, lo4.y}, xt_bf16x2), w1 = __builtin_convertvector((xt_f32x2){lo4.z, lo4.w}, xt_bf16x2);
                const xt_bf16x2 w2 = __builtin_convertvector((xt_f32x2){hi4.x, hi4.y}, xt_bf16x2), w3 = __builtin_convertvector((xt_f32x2){hi4.z, hi4.w}, xt_bf16x2);
                *reinterpret_cast<uint4*>(Crow16 + (long long)(i * 32 + p * 16) * g.ldc + nt * 32) =
                    make_uint4(__builtin_bit_cast(unsigned, w0), __builtin_bit_cast(unsigned, w1), __builtin_bit_cast(unsigned, w2), __builtin_bit_cast(unsigned, w3));
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(Crow + (long long)(i * 32 + p * 8) * g.ldc + nt * 32) = *reinterpret_cast<const float4*>(str + p * 8 * 36);
    };
    for (int nt = wave; nt < NT; nt += 4) {
        // (the operand fetches do not depend on the tile: left visible, the compiler hoists all of them out of this loop -- 256
        //  registers of data fragments, spilled)
        int opq = 0;
        asm volatile("" : "+v"(opq));            // (an opaque OFFSET: an opaque pointer loses the LDS address space -> flat loads)
        const __bf16* dhi = dh + opq;
        const __bf16* dlo = dhi + IMG;
        f32x16 acc[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xt_bf16x8 d_hi[RT], d_lo[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                d_hi[i] = *reinterpret_cast<const xt_bf16x8*>(dhi + i * 32 * LDK + s * 16);
                d_lo[i] = *reinterpret_cast<const xt_bf16x8*>(dlo + i * 32 * LDK + s * 16);
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_lo[s], d_hi[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[s], d_lo[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_hi[s], d_hi[i], acc[i], 0, 0, 0);
            load_w(nt + 4, s);
        }
        // ---- epilogue: registers 4 q .. 4 q + 3 of row tile i are columns 32 nt + 8 q + 4 h + {0..3} of row 32 i + col ----
        if (plain) {                              // uniform: the bare product (the edge projections)
            if constexpr (ROWS) {
#pragma unroll
                for (int i = 0; i < RT; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(stw + q * 8) = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
                    rows_out(nt, i);
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* dst = C + (long long)(nt * 4 + q) * g.panel_ld * 8;
#pragma unroll
                for (int i = 0; i < RT; ++i)
                    *reinterpret_cast<float4*>(dst + i * 32 * 8) = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nt * 32 + 8 * q + 4 * h;
            float4 bi = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bi;
            if (g.bias) bi = *reinterpret_cast<const float4*>(g.bias + n);
            if (g.scale) { sc = *reinterpret_cast<const float4*>(g.scale + n); sh = *reinterpret_cast<const float4*>(g.shift + n); }
            float* dst = C + (long long)(nt * 4 + q) * g.panel_ld * 8;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float4 v = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
                v.x = (v.x + bi.x) * sc.x + sh.x; v.y = (v.y + bi.y) * sc.y + sh.y;
                v.z = (v.z + bi.z) * sc.z + sh.z; v.w = (v.w + bi.w) * sc.w + sh.w;
                v.x = lpd_act_pl(v.x, g.ns); v.y = lpd_act_pl(v.y, g.ns); v.z = lpd_act_pl(v.z, g.ns); v.w = lpd_act_pl(v.w, g.ns);
                if constexpr (ROWS) acc[i][4 * q] = v.x, acc[i][4 * q + 1] = v.y, acc[i][4 * q + 2] = v.z, acc[i][4 * q + 3] = v.w;
                else *reinterpret_cast<float4*>(dst + i * 32 * 8) = v;
            }
        }
        if constexpr (ROWS) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stw + q * 8) = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
                rows_out(nt, i);
            }
        }
    }
}

template <int KS, bool ROWS = false, int RB = 128>
void x3t_launch(const X3tArgs& g, hipStream_t stream, int batch = 1)
{
    const size_t lds = (size_t)2 * RB * (KS * 16 + 8) * sizeof(__bf16) + (ROWS ? 4 * 32 * 36 * sizeof(float) : 0);
    auto kern = gemm_x3t_kernel<KS, ROWS, RB>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(g.M / RB, batch), dim3(XT_THREADS), lds, stream, g);
}

}  // namespace

extern "C" int lpd_gemm_x3t_applies(int M, int N, int K, int act, long long a_cloud, long long c_cloud, int panel_n)
{
    return a_cloud != 0 && c_cloud != 0 && (K == 64 || K == 128) && N > 0 && N % 32 == 0 && M > 0 && M % 128 == 0 &&
           panel_n > 0 && panel_n % 128 == 0 && M % panel_n == 0 && act >= 0 && act <= 2;
}

static int gemm_x3t_impl(const float* A, const void* frags, float* C, int M, int N, int K, const float* bias, const float* scale,
                         const float* shift, int act, float slope, long long a_cloud, long long c_cloud, int panel_n, int panel_ld,
                         long long a_lo, void* stream_)
{
    LPD_CHECK_ARG(A && frags && C, "lpd_gemm_x3t: null pointer");
    LPD_CHECK_ARG(lpd_gemm_x3t_applies(M, N, K, act, a_cloud, c_cloud, panel_n),
                  "lpd_gemm_x3t: built for cloud-panel A and C, K in {64, 128}, N %% 32 == 0, clouds of a multiple of 128 points, "
                  "act none / ReLU / LeakyReLU (M=%d N=%d K=%d act=%d)", M, N, K, act);
    LPD_CHECK_ARG(panel_ld >= panel_n, "lpd_gemm_x3t: panel_ld < panel_n");
    LPD_CHECK_ARG(act != 2 || (slope >= 0.0f && slope <= 1.0f), "lpd_gemm_x3t: LeakyReLU slope %g outside [0, 1]", (double)slope);
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_gemm_x3t: scale and shift must be given together");
    LPD_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)frags & 15) == 0 && ((uintptr_t)C & 15) == 0 &&
                      ((uintptr_t)bias & 15) == 0 && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
                  "lpd_gemm_x3t: pointers must be 16-byte aligned");
    const int KS = K / 16, NT = N / 32;
    const __bf16* fhi = reinterpret_cast<const __bf16*>(frags);
    X3tArgs g{A, fhi, fhi + (long long)NT * KS * 512, C, M, N, bias, scale, shift, act == 0 ? 1.0f : (act == 1 ? 0.0f : slope), a_cloud, c_cloud, panel_n, panel_ld, a_lo,
              0, 0, 0, 0, 0, 0};
    hipStream_t stream = (hipStream_t)stream_;
    if (KS == 8) x3t_launch<8>(g, stream);
    else x3t_launch<4>(g, stream);
    LPD_CHECK_LAUNCH("lpd_gemm_x3t");
    return LPD_OK;
}

extern "C" int lpd_gemm_x3t(const float* A, const void* frags, float* C, int M, int N, int K, const float* bias, const float* scale,
                            const float* shift, int act, float slope, long long a_cloud, long long c_cloud, int panel_n, int panel_ld,
                            void* stream_)
{
    return gemm_x3t_impl(A, frags, C, M, N, K, bias, scale, shift, act, slope, a_cloud, c_cloud, panel_n, panel_ld, 0, stream_);
}

// the same product with A given as SPLIT bf16 planes (a_hi: hi plane [cloud][K/8][panel_ld][8], a_cloud elements between clouds; the lo
// plane a_lo elements behind it): the x2 block as the fused edge MLP writes it for conv3 (lpd_edge_mlp_bf16x3s)
extern "C" int lpd_gemm_x3ts(const void* a_hi, long long a_lo, const void* frags, float* C, int M, int N, int K, const float* bias,
                             const float* scale, const float* shift, int act, float slope, long long a_cloud, long long c_cloud,
                             int panel_n, int panel_ld, void* stream_)
{
    LPD_CHECK_ARG(a_lo != 0 && a_lo % 8 == 0, "lpd_gemm_x3ts: lo-plane offset");
    return gemm_x3t_impl(reinterpret_cast<const float*>(a_hi), frags, C, M, N, K, bias, scale, shift, act, slope, a_cloud, c_cloud, panel_n,
                         panel_ld, a_lo, stream_);
}

// The same transposed product on ROW-MAJOR fp32 operands (no panels): C [M][ldc] = act((A [M][lda] . W^T + bias) * scale + shift), K = 64
// or 128, N % 32 == 0, M % 128 == 0; `batch` problems with their own A, C and fragments (strides sA, sC in floats; frag_bytes between the
// fragment sets as lpd_gemm_prep_b wrote them).  The generic 128 x 128 block kernel spends a short reduction in barriers, re-splitting
// and 4-byte stores: SN1 projection of the training step (180224 x 512 x 128) 211 us there.
extern "C" int lpd_gemm_x3t_rows_applies(int M, int N, int K, int act, long long lda, long long ldc)
{
    return (K == 64 || K == 128) && N > 0 && N % 32 == 0 && M > 0 && M % 128 == 0 && act >= 0 && act <= 2 && lda % 4 == 0 && ldc % 4 == 0 &&
           lda >= K && ldc >= N;
}

// c_bf16: C receives bf16 values (ldc / sC in bf16 elements, ldc % 8 == 0).
extern "C" int lpd_gemm_x3t_rows(const float* A, long long lda, const void* frags, void* C_, long long ldc, int c_bf16, int M, int N, int K,
                                 const float* bias, const float* scale, const float* shift, int act, float slope, int batch, long long sA, long long sC,
                                 long long frag_bytes, void* stream_)
{
    float* C = reinterpret_cast<float*>(C_);
    LPD_CHECK_ARG(!c_bf16 || (ldc % 8 == 0 && sC % 8 == 0), "lpd_gemm_x3t_rows: bf16 C needs ldc %% 8 == 0");
    LPD_CHECK_ARG(A && frags && C && batch >= 1 && batch <= 65535, "lpd_gemm_x3t_rows: bad arguments");
    LPD_CHECK_ARG(lpd_gemm_x3t_rows_applies(M, N, K, act, lda, ldc),
                  "lpd_gemm_x3t_rows: K in {64, 128}, N %% 32 == 0, M %% 128 == 0, act none / ReLU / LeakyReLU (M=%d N=%d K=%d act=%d)", M, N, K, act);
    LPD_CHECK_ARG(act != 2 || (slope >= 0.0f && slope <= 1.0f), "lpd_gemm_x3t_rows: LeakyReLU slope %g outside [0, 1]", (double)slope);
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_gemm_x3t_rows: scale and shift must be given together");
    LPD_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)frags & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
                      ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0 && sA % 4 == 0 && sC % 4 == 0 && frag_bytes % 16 == 0,
                  "lpd_gemm_x3t_rows: pointers / strides must be 16-byte aligned");
    const int KS = K / 16, NT = N / 32;
    const __bf16* fhi = reinterpret_cast<const __bf16*>(frags);
    X3tArgs g{A, fhi, fhi + (long long)NT * KS * 512, C, M, N, bias, scale, shift, act == 0 ? 1.0f : (act == 1 ? 0.0f : slope), 0, 0, 0, 0, 0,
              lda, ldc, sA, sC, frag_bytes / 2, c_bf16 ? 1 : 0};
    hipStream_t stream = (hipStream_t)stream_;
    static const int rb = lpd_debug("x3t-rb", 64);
    if (KS == 8 && rb == 64 && c_bf16) x3t_launch<8, true, 64>(g, stream, batch);      // 64-row workgroups: two per CU instead of one (see the kernel; fp32 results: 116 us either way)
    else if (KS == 8) x3t_launch<8, true>(g, stream, batch);
    else x3t_launch<4, true>(g, stream, batch);
    LPD_CHECK_LAUNCH("lpd_gemm_x3t_rows");
    return LPD_OK;
}
