// lpd_gemm.hip -- fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32) with a fused
// per-column affine + activation epilogue, batched and split-K capable.
//
// This one kernel family carries every dense contraction of the hot path:
//   * per-point 1x1 convolutions + eval-mode BatchNorm + activation
//       (util/lpdnet_model.py:231-232,262 conv{1,2,3}_lpd; util/PointNetVlad.py:213-230 conv1..5;
//        T-Net / STN stacks lpdnet_model.py:297-299, PointNetVlad.py:152-160)
//   * the neighbour/centre projections of the split edge convolutions (lpdnet_model.py:249,257)
//   * NetVLAD soft-assignment x @ cluster_weights (PointNetVlad.py:48) and the residual pooling
//     act^T @ x (PointNetVlad.py:66) as a batched "A stored k-major" GEMM
//   * the 65536->256 hidden projection (PointNetVlad.py:76) via split-K (weight-bandwidth bound).
//
// Why f32 MFMA and not bf16: descriptors must match the fp32 reference to 1e-4 (norm-relative);
// the f32-input MFMA is exact fp32 (one rounding per FMA) at the full fp32 vector rate
// (157 TFLOP/s peak, MI355X_MICROARCH.md "Matrix cores") and needs one VGPR per operand.
//
// Tiling (cdna_hip_programming.md section 3 "FP32-input MFMA" reports 122 TF for this shape):
// 256 threads = 2x2 waves, block tile 128 x (64|128) x 32, each wave TM x TN tiles of 32x32.
// Both operands sit in LDS k-major ([k][m] / [k][n]) so every MFMA operand fetch is a
// conflict-free ds_read_b32 of 32 consecutive floats per half-wave; operands that are stored
// k-contiguous in memory are transposed while being staged (row pad 1 => 2-way, free).
#include "lpd_common.h"

namespace {

constexpr int GEMM_THREADS = 256;
constexpr int GEMM_BK = 32;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;          // logical sizes; K is the per-split depth handled by one block (multiple of 32)
    int Ktot;             // true reduction length: k >= Ktot reads as zero (ragged K, e.g. N points not % 32)
    int lda, ldb, ldc;
    long long sA, sB, sC;  // batch strides (elements)
    int splits;            // grid.z = batch * splits
    long long sCsplit;     // slab stride between splits (elements)
    // epilogue (ignored when splits > 1: raw partial sums are written)
    const float* bias;    // [N] or null: v += bias
    const float* scale;   // [N] or null: v = v * scale + shift
    const float* shift;   // [N] or null
    int act;
    float slope;
    int accumulate;       // C += result (applied after the epilogue)
};

__device__ __forceinline__ float4 ld4_guard(const float* row, int i, int limit)
{
    // row is 16-byte aligned and i % 4 == 0; elements >= limit read as zero
    if (i + 3 < limit) return *reinterpret_cast<const float4*>(row + i);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < limit) v.x = row[i];
    if (i + 1 < limit) v.y = row[i + 1];
    if (i + 2 < limit) v.z = row[i + 2];
    return v;
}

// TN: 32-wide n-tiles per wave (1 => BN = 64, 2 => BN = 128).  TM fixed at 2 (BM = 128).
// KTAIL: the reduction length is not a multiple of 32 (ragged point counts); only then are the operand loads k-guarded
// (the guards cost ~25 % on the big GEMMs when compiled in unconditionally).
template <bool A_KMAJOR, bool B_KMAJOR, int TN, bool KTAIL>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(GemmArgs g)
{
    constexpr int BM = 128;
    constexpr int BN = 64 * TN;
    constexpr int BK = GEMM_BK;
    // leading dims of the k-major LDS images: +1 pad for transposing stores, +4 for straight copies
    constexpr int LDA = A_KMAJOR ? BM + 4 : BM + 1;
    constexpr int LDB = B_KMAJOR ? BN + 4 : BN + 1;
    constexpr int A_F4 = BM * BK / 4 / GEMM_THREADS;  // float4 per thread per tile = 4
    constexpr int B_F4 = BN * BK / 4 / GEMM_THREADS;  // 2 or 4

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                   // [2][BK][LDA]
    float* Bs = smem + 2 * BK * LDA;    // [2][BK][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    const int z = blockIdx.z;
    const int batch = z / g.splits;
    const int split = z - batch * g.splits;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int kbase = split * g.K;

    const float* A = g.A + (long long)batch * g.sA;
    const float* B = g.B + (long long)batch * g.sB;
    float* C = g.C + (long long)batch * g.sC + (long long)split * g.sCsplit;

    float4 ra[A_F4], rb[B_F4];

    auto load_tiles = [&](int kt) {
        const int k0 = kbase + kt * BK;
#pragma unroll
        for (int e = 0; e < A_F4; ++e) {
            int f = e * GEMM_THREADS + tid;
            if constexpr (A_KMAJOR) {  // memory [k][m]: 32 float4 per k-row
                int kk = f / (BM / 4), mq = f % (BM / 4);
                ra[e] = (!KTAIL || k0 + kk < g.Ktot) ? ld4_guard(A + (long long)(k0 + kk) * g.lda, m0 + mq * 4, g.M)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {                   // memory [m][k]: 8 float4 per m-row
                int mm = f / (BK / 4), kq = f % (BK / 4);
                int m = m0 + mm;
                m = m < g.M ? m : g.M - 1;  // clamp: rows >= M are never stored
                if constexpr (KTAIL) ra[e] = ld4_guard(A + (long long)m * g.lda, k0 + kq * 4, g.Ktot);
                else ra[e] = *reinterpret_cast<const float4*>(A + (long long)m * g.lda + k0 + kq * 4);
            }
        }
#pragma unroll
        for (int e = 0; e < B_F4; ++e) {
            int f = e * GEMM_THREADS + tid;
            if constexpr (B_KMAJOR) {  // memory [k][n]
                int kk = f / (BN / 4), nq = f % (BN / 4);
                rb[e] = (!KTAIL || k0 + kk < g.Ktot) ? ld4_guard(B + (long long)(k0 + kk) * g.ldb, n0 + nq * 4, g.N)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {                   // memory [n][k]
                int nn = f / (BK / 4), kq = f % (BK / 4);
                int n = n0 + nn;
                n = n < g.N ? n : g.N - 1;
                if constexpr (KTAIL) rb[e] = ld4_guard(B + (long long)n * g.ldb, k0 + kq * 4, g.Ktot);
                else rb[e] = *reinterpret_cast<const float4*>(B + (long long)n * g.ldb + k0 + kq * 4);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float* as = As + buf * BK * LDA;
        float* bs = Bs + buf * BK * LDB;
#pragma unroll
        for (int e = 0; e < A_F4; ++e) {
            int f = e * GEMM_THREADS + tid;
            if constexpr (A_KMAJOR) {
                int kk = f / (BM / 4), mq = f % (BM / 4);
                *reinterpret_cast<float4*>(as + kk * LDA + mq * 4) = ra[e];
            } else {
                int mm = f / (BK / 4), kq = f % (BK / 4);
                as[(kq * 4 + 0) * LDA + mm] = ra[e].x;
                as[(kq * 4 + 1) * LDA + mm] = ra[e].y;
                as[(kq * 4 + 2) * LDA + mm] = ra[e].z;
                as[(kq * 4 + 3) * LDA + mm] = ra[e].w;
            }
        }
#pragma unroll
        for (int e = 0; e < B_F4; ++e) {
            int f = e * GEMM_THREADS + tid;
            if constexpr (B_KMAJOR) {
                int kk = f / (BN / 4), nq = f % (BN / 4);
                *reinterpret_cast<float4*>(bs + kk * LDB + nq * 4) = rb[e];
            } else {
                int nn = f / (BK / 4), kq = f % (BK / 4);
                bs[(kq * 4 + 0) * LDB + nn] = rb[e].x;
                bs[(kq * 4 + 1) * LDB + nn] = rb[e].y;
                bs[(kq * 4 + 2) * LDB + nn] = rb[e].z;
                bs[(kq * 4 + 3) * LDB + nn] = rb[e].w;
            }
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = g.K / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float* as = As + buf * BK * LDA + wm * 64 + col;
        const float* bs = Bs + buf * BK * LDB + wn * (32 * TN) + col;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[2], b[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = as[(2 * s + h) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(2 * s + h) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    const bool raw = g.splits > 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (32 * TN) + j * 32 + col;
        if (n >= g.N) continue;
        float bi = 0.f, sc = 1.f, sh = 0.f;
        if (!raw) {
            if (g.bias) bi = g.bias[n];
            if (g.scale) { sc = g.scale[n]; sh = g.shift[n]; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                if (!raw) {
                    v += bi;
                    v = v * sc + sh;
                    v = lpd_act(v, g.act, g.slope);
                    if (g.accumulate) v += C[(long long)m * g.ldc + n];
                }
                C[(long long)m * g.ldc + n] = v;
            }
        }
    }
}

// sums split-K slabs and applies the epilogue.  one thread per output element.
__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N,
                                          int ldc, int splits, long long slab_stride, long long sWs_batch,
                                          long long sC_batch, const float* bias, const float* scale,
                                          const float* shift, int act, float slope, int accumulate)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    const int batch = blockIdx.z;
    if (n >= N) return;
    const float* p = slabs + (long long)batch * sWs_batch + (long long)m * N + n;
    float v = 0.0f;
    for (int s = 0; s < splits; ++s) v += p[(long long)s * slab_stride];
    if (bias) v += bias[n];
    if (scale) v = v * scale[n] + shift[n];
    v = lpd_act(v, act, slope);
    float* dst = C + (long long)batch * sC_batch + (long long)m * ldc + n;
    if (accumulate) v += *dst;
    *dst = v;
}

template <bool AK, bool BK_, int TN, bool KTAIL>
int gemm_launch_t(const GemmArgs& g, int batch, hipStream_t stream);

template <bool AK, bool BK_, int TN>
int gemm_launch(const GemmArgs& g, int batch, hipStream_t stream)
{
    // every split covers whole 32-deep k-tiles of real data <=> Ktot is a multiple of 32 and splits divide it evenly
    const bool ktail = (g.Ktot % GEMM_BK) != 0 || (long long)g.K * g.splits != g.Ktot;
    return ktail ? gemm_launch_t<AK, BK_, TN, true>(g, batch, stream) : gemm_launch_t<AK, BK_, TN, false>(g, batch, stream);
}

template <bool AK, bool BK_, int TN, bool KTAIL>
int gemm_launch_t(const GemmArgs& g, int batch, hipStream_t stream)
{
    constexpr int BM = 128, BN = 64 * TN;
    constexpr int LDA = AK ? BM + 4 : BM + 1;
    constexpr int LDB = BK_ ? BN + 4 : BN + 1;
    size_t lds = (size_t)2 * GEMM_BK * (LDA + LDB) * sizeof(float);
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch * g.splits);
    auto kern = gemm_f32_kernel<AK, BK_, TN, KTAIL>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, stream, g);
    LPD_CHECK_LAUNCH("lpd_gemm");
    return LPD_OK;
}

}  // namespace

// C-ABI: see include/lpd_hip.h for the contract.
extern "C" int lpd_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                        int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC,
                        int splits, float* splitk_ws, const float* bias, const float* scale, const float* shift,
                        int act, float slope, int accumulate, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(A && B && C, "lpd_gemm: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "lpd_gemm: bad dims M=%d N=%d K=%d batch=%d", M, N, K, batch);
    LPD_CHECK_ARG(splits >= 1, "lpd_gemm: splits=%d", splits);
    LPD_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0, "lpd_gemm: lda/ldb must be multiples of 4 (lda=%d ldb=%d)", lda, ldb);
    LPD_CHECK_ARG(sA % 4 == 0 && sB % 4 == 0, "lpd_gemm: batch strides of A/B must be multiples of 4");
    LPD_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "lpd_gemm: A/B must be 16-byte aligned");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_gemm: scale and shift must be given together");
    LPD_CHECK_ARG(splits == 1 || splitk_ws, "lpd_gemm: split-K needs a workspace of batch*splits*M*N floats");
    LPD_CHECK_ARG((long long)batch * splits <= 65535, "lpd_gemm: batch*splits exceeds grid.z");

    GemmArgs g;
    g.A = A; g.B = B;
    g.M = M; g.N = N;
    g.K = ((K + splits - 1) / splits + GEMM_BK - 1) / GEMM_BK * GEMM_BK;   // per-split depth, rounded up to the k-tile
    g.Ktot = K;
    g.lda = lda; g.ldb = ldb;
    g.sA = sA; g.sB = sB;
    g.splits = splits;
    g.bias = bias; g.scale = scale; g.shift = shift; g.act = act; g.slope = slope;
    g.accumulate = accumulate;
    if (splits > 1) {
        g.C = splitk_ws; g.ldc = N; g.sC = (long long)splits * M * N; g.sCsplit = (long long)M * N;
    } else {
        g.C = C; g.ldc = ldc; g.sC = sC; g.sCsplit = 0;
    }
    const int tn = N > 64 ? 2 : 1;
    int rc;
#define LPD_GEMM_CASE(AK, BK_)                                                       \
    rc = (tn == 2) ? gemm_launch<AK, BK_, 2>(g, batch, stream) : gemm_launch<AK, BK_, 1>(g, batch, stream)
    if (a_kmajor && b_kmajor) LPD_GEMM_CASE(true, true);
    else if (a_kmajor) LPD_GEMM_CASE(true, false);
    else if (b_kmajor) LPD_GEMM_CASE(false, true);
    else LPD_GEMM_CASE(false, false);
#undef LPD_GEMM_CASE
    if (rc != LPD_OK) return rc;
    if (splits > 1) {
        dim3 grid((N + 255) / 256, M, batch);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, grid, dim3(256), 0, stream, (const float*)splitk_ws, C, M, N,
                           ldc, splits, (long long)M * N, (long long)splits * M * N, sC, bias, scale, shift, act, slope, accumulate);
        LPD_CHECK_LAUNCH("lpd_gemm(splitk reduce)");
    }
    return LPD_OK;
}
