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
    // Cloud-panel operands [cloud][cols/8][Np][8] (the layout the cloud-resident K-agg kernel streams: an 8-channel slice
    // of one cloud is one contiguous 32*Np-byte run, a whole cloud one contiguous block): a_cloud / c_cloud = floats
    // between clouds (the buffer may hold more panels than this operand uses), panel_n = Np = points per cloud
    // (a multiple of the 128-row block tile, so a block's rows share a cloud).  0 = row-major.
    long long a_cloud, c_cloud;
    int panel_n;
    int panel_ld;   // rows allotted to one panel (>= panel_n; the pad keeps consecutive panels off the same HBM channels)
    int prods;      // split-bf16 kernel: 3 = a_lo b_hi + a_hi b_lo + a_hi b_hi (fp32-grade), 1 = a_hi b_hi only (plain bf16 operands:
                    // the bf16-storage training mode, lpd_gemm_bf16x1)
};

// PANELS (template): bit 0 = A panel-major, bit 1 = C panel-major; 0 compiles to the plain row-major addressing
template <int PANELS>
__device__ __forceinline__ float* gemm_c_ptr(const GemmArgs& g, float* C, int m, int n)
{
    if constexpr (PANELS & 2) {   // C already points at the block's cloud; m is the row inside the cloud
        return C + ((long long)(n >> 3) * g.panel_ld + m) * 8 + (n & 7);
    } else return C + (long long)m * g.ldc + n;
}

// float4 of A at (row m, k0 + 4 kq .. +3); k0 is the (uniform) start of a 32-deep k-tile.  Panel form: A points at the
// block's cloud, m is the row inside the cloud; split so that the k-tile term is scalar and the rest loop-invariant.
template <int PANELS>
__device__ __forceinline__ const float* gemm_a_ptr(const float* A, int lda, int panel_ld, int m, int k0, int kq)
{
    if constexpr (PANELS & 1)
        return A + (long long)(k0 >> 3) * panel_ld * 8 + (((kq >> 1) * panel_ld + m) * 8 + (kq & 1) * 4);
    else return A + (long long)m * lda + k0 + kq * 4;
}

// Tile coordinates of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin by linear id, x fastest: with the
// plain blockIdx mapping the column blocks of one row tile land on 8 different XCDs and every XCD's L2 pulls the whole
// of A across the fabric (conv3: 8 x 268 MB).  Remapped so that each XCD owns a contiguous range of (z, row tile,
// column block) triples: the column blocks of a row tile run next to each other on ONE L2 and A crosses the fabric once.
__device__ __forceinline__ void gemm_block_coords(int& bx, int& by, int& bz)
{
    const int gx = gridDim.x, gy = gridDim.y;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int v = lpd_xcd_remap(lin, gx * gy * gridDim.z);
    bx = v % gx;
    const int t = v / gx;
    by = t % gy;
    bz = t / gy;
}

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

// Epilogue of one wave's 64 x (32*TN) accumulator block (32x32 MFMA tile layout: column = lane & 31, row =
// (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).  The activation code is uniform: the sigmoid (exp + divide per element)
// lives behind a scalar branch, the other three are the branch-free  max(v,0) + ns * min(v,0).
template <int TN, bool SIGMOID, int PANELS>
__device__ __forceinline__ void gemm_epilogue_t(const GemmArgs& g, const f32x16 (&acc)[2][TN], float* C, int mw, int nw, int h, int col,
                                                int m_cloud0)
{
    const bool raw = g.splits > 1;
    const float ns = g.act == 0 ? 1.0f : (g.act == 1 ? 0.0f : g.slope);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + col;
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
                const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                if (!raw) {
                    v += bi;
                    v = v * sc + sh;
                    if constexpr (SIGMOID) v = 1.0f / (1.0f + __expf(-v));
                    else v = fmaxf(v, 0.0f) + ns * fminf(v, 0.0f);
                    if (g.accumulate) v += *gemm_c_ptr<PANELS>(g, C, (PANELS & 2) ? m - m_cloud0 : m, n);
                }
                *gemm_c_ptr<PANELS>(g, C, (PANELS & 2) ? m - m_cloud0 : m, n) = v;
            }
        }
    }
}

template <int TN, int PANELS>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][TN], float* C, int mw, int nw, int h, int col,
                                              int m_cloud0)
{
    if (g.act == 3 && g.splits == 1) gemm_epilogue_t<TN, true, PANELS>(g, acc, C, mw, nw, h, col, m_cloud0);
    else gemm_epilogue_t<TN, false, PANELS>(g, acc, C, mw, nw, h, col, m_cloud0);
}

// TN: 32-wide n-tiles per wave (1 => BN = 64, 2 => BN = 128).  TM fixed at 2 (BM = 128).
// KTAIL: the reduction length is not a multiple of 32 (ragged point counts); only then are the operand loads k-guarded
// (the guards cost ~25 % on the big GEMMs when compiled in unconditionally).
template <bool A_KMAJOR, bool B_KMAJOR, int TN, bool KTAIL, int PANELS = 0>
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

    int bx, by, z;
    gemm_block_coords(bx, by, z);
    const int batch = z / g.splits;
    const int split = z - batch * g.splits;
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int kbase = split * g.K;

    // cloud-panel operands: the block's 128 rows lie in one cloud (panel_n % 128 == 0)
    const int m_cloud0 = (PANELS != 0) ? (m0 / g.panel_n) * g.panel_n : 0;
    const float* A = g.A + (long long)batch * g.sA + ((PANELS & 1) ? (long long)(m0 / g.panel_n) * g.a_cloud : 0);
    const float* B = g.B + (long long)batch * g.sB;
    float* C = g.C + (long long)batch * g.sC + (long long)split * g.sCsplit + ((PANELS & 2) ? (long long)(m0 / g.panel_n) * g.c_cloud : 0);

    float4 ra[A_F4], rb[B_F4];

    // k-tiles are walked from `kskew` on (and wrap): workgroups that sweep the reduction in step over operands with a
    // power-of-two row stride keep meeting on the same HBM channels otherwise (NetVLAD pooling: 4-KiB rows)
    const int nkt = g.K / GEMM_BK;
    const int kskew = (by * 5 + bx * 3) % nkt;
    auto load_tiles = [&](int kt) {
        kt += kskew;
        kt = kt >= nkt ? kt - nkt : kt;
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
                if constexpr (KTAIL) ra[e] = ld4_guard(A + (long long)m * g.lda, k0 + kq * 4, g.Ktot);   // (ragged K: row-major only)
                else ra[e] = *reinterpret_cast<const float4*>(gemm_a_ptr<PANELS>(A, g.lda, g.panel_ld, (PANELS & 1) ? m - m_cloud0 : m, k0, kq));
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
    gemm_epilogue<TN, PANELS>(g, acc, C, m0 + wm * 64, n0 + wn * (32 * TN), h, col, m_cloud0);
}


// ---------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") GEMM: fp32 operands, fp32 accumulation, three bf16 MFMA products per term.
//
// Each fp32 operand is split while it is staged into LDS:  x = hi + lo + O(2^-17 |x|),  hi = bf16(x),
// lo = bf16(x - hi), and  a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on v_mfma_f32_32x32x16_bf16 (products of bf16
// values are exact in fp32; the dropped lo*lo term and the split residual are ~2^-16 relative per product).
// Measured effect on the descriptors of the full network: 2e-6 norm-relative against the fp32/fp64 reference, 45x
// inside the 1e-4 bar (DESIGN.md "GEMM precision"); the bf16 MFMA runs at 16x the rate of the f32-input MFMA, so
// three products still leave > 5x.  Not used where bits matter: the kNN distance tiles and the layers that feed the
// feature-space kNN (conv1/conv2) stay on the f32-input MFMA / fp32 FMA paths.
//
// LDS images are [row][k] with k contiguous (row stride 40 bf16 = 80 B: conflict-free ds_read_b128 for the 32x32x16
// operand: lane (r = lane & 31, h = lane >> 5) reads k = 8h .. 8h+7 of row r).  Operands stored k-major in memory
// are transposed in registers: a thread loads a 4(k) x 4(m) patch and writes four 8-byte k-runs.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// hi = bf16(x) (RNE), lo = bf16(x - hi), on packed pairs: v_cvt_pk_bf16_f32 for two elements at a time, the hi halves
// widened back to fp32 by a shift / mask (written element by element the compiler converts some elements singly and
// re-packs them with perm / alignbit / mov; see lpd_edge.hip).  Same values either way.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, bf16x4& hi, bf16x4& lo)
{
    const bf16x2 h01 = __builtin_convertvector((f32x2_){x0, x1}, bf16x2), h23 = __builtin_convertvector((f32x2_){x2, x3}, bf16x2);
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    const float r0 = x0 - __uint_as_float(u01 << 16), r1 = x1 - __uint_as_float(u01 & 0xffff0000u);
    const float r2 = x2 - __uint_as_float(u23 << 16), r3 = x3 - __uint_as_float(u23 & 0xffff0000u);
    const bf16x2 l01 = __builtin_convertvector((f32x2_){r0, r1}, bf16x2), l23 = __builtin_convertvector((f32x2_){r2, r3}, bf16x2);
    const uint2 hp = make_uint2(u01, u23), lp = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    hi = __builtin_bit_cast(bf16x4, hp);
    lo = __builtin_bit_cast(bf16x4, lp);
}

// Register staging of one operand tile (ROWS x 32): NR = max(ROWS / 32, 4 for k-major with ROWS = 64) float4 per thread.
// k-major operands are held as 4(k) x 4(row) patches: patch id = p * 256 + tid, kq = id / (ROWS/4), rq = id % (ROWS/4).
template <int ROWS>
struct X3Regs { static constexpr int NR = ROWS >= 128 ? ROWS / 32 : 4; };

template <bool KMAJOR, int ROWS>
__device__ __forceinline__ void x3_store(const float4 (&r)[X3Regs<ROWS>::NR], __bf16* hi_img, __bf16* lo_img, int tid)
{
    constexpr int LDK = GEMM_BK + 8;
    if constexpr (KMAJOR) {
        constexpr int NP = (2 * ROWS + GEMM_THREADS - 1) / GEMM_THREADS;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int id = p * GEMM_THREADS + tid;
            const int kq = id / (ROWS / 4), rq = id % (ROWS / 4);
            if (2 * ROWS % GEMM_THREADS == 0 || kq < GEMM_BK / 4) {
                const float4 &r0 = r[p * 4], &r1 = r[p * 4 + 1], &r2 = r[p * 4 + 2], &r3 = r[p * 4 + 3];
                bf16x4 h, l;
                split4(r0.x, r1.x, r2.x, r3.x, h, l);
                *reinterpret_cast<bf16x4*>(hi_img + (rq * 4 + 0) * LDK + kq * 4) = h;
                *reinterpret_cast<bf16x4*>(lo_img + (rq * 4 + 0) * LDK + kq * 4) = l;
                split4(r0.y, r1.y, r2.y, r3.y, h, l);
                *reinterpret_cast<bf16x4*>(hi_img + (rq * 4 + 1) * LDK + kq * 4) = h;
                *reinterpret_cast<bf16x4*>(lo_img + (rq * 4 + 1) * LDK + kq * 4) = l;
                split4(r0.z, r1.z, r2.z, r3.z, h, l);
                *reinterpret_cast<bf16x4*>(hi_img + (rq * 4 + 2) * LDK + kq * 4) = h;
                *reinterpret_cast<bf16x4*>(lo_img + (rq * 4 + 2) * LDK + kq * 4) = l;
                split4(r0.w, r1.w, r2.w, r3.w, h, l);
                *reinterpret_cast<bf16x4*>(hi_img + (rq * 4 + 3) * LDK + kq * 4) = h;
                *reinterpret_cast<bf16x4*>(lo_img + (rq * 4 + 3) * LDK + kq * 4) = l;
            }
        }
    } else {                       // register e: row (e*256 + tid) / 8, k quad (e*256 + tid) % 8
        constexpr int NF4 = ROWS * GEMM_BK / 4 / GEMM_THREADS;
#pragma unroll
        for (int e = 0; e < NF4; ++e) {
            const int f = e * GEMM_THREADS + tid;
            const int rr = f / (GEMM_BK / 4), kq = f % (GEMM_BK / 4);
            bf16x4 h, l;
            split4(r[e].x, r[e].y, r[e].z, r[e].w, h, l);
            *reinterpret_cast<bf16x4*>(hi_img + rr * LDK + kq * 4) = h;
            *reinterpret_cast<bf16x4*>(lo_img + rr * LDK + kq * 4) = l;
        }
    }
}

// load one operand tile (ROWS x 32 at row0, k0) into r[]; `mem` is the operand base, ld its leading dim
template <bool KMAJOR, int ROWS, bool KTAIL, int PANELS = 0>
__device__ __forceinline__ void x3_load(float4 (&r)[X3Regs<ROWS>::NR], const float* mem, int ld, int row0, int nrows, int k0,
                                        int Ktot, int tid, int panel_ld = 0, int cloud_row0 = 0)
{
    if constexpr (KMAJOR) {        // memory [k][row]
        constexpr int NP = (2 * ROWS + GEMM_THREADS - 1) / GEMM_THREADS;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int id = p * GEMM_THREADS + tid;
            const int kq = id / (ROWS / 4), rq = id % (ROWS / 4);
            if (2 * ROWS % GEMM_THREADS == 0 || kq < GEMM_BK / 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = k0 + kq * 4 + e;
                    r[p * 4 + e] = (!KTAIL || kk < Ktot) ? ld4_guard(mem + (long long)kk * ld, row0 + rq * 4, nrows)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    } else {                       // memory [row][k]
        constexpr int NF4 = ROWS * GEMM_BK / 4 / GEMM_THREADS;
#pragma unroll
        for (int e = 0; e < NF4; ++e) {
            const int f = e * GEMM_THREADS + tid;
            const int rr = f / (GEMM_BK / 4), kq = f % (GEMM_BK / 4);
            int row = row0 + rr;
            row = row < nrows ? row : nrows - 1;   // clamp: rows past the end are never stored
            if constexpr (KTAIL) r[e] = ld4_guard(mem + (long long)row * ld, k0 + kq * 4, Ktot);
            else r[e] = *reinterpret_cast<const float4*>(gemm_a_ptr<PANELS>(mem, ld, panel_ld, (PANELS & 1) ? row - cloud_row0 : row, k0, kq));
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, int TN, bool KTAIL, int PANELS = 0>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_bf16x3_kernel(GemmArgs g)
{
    constexpr int BM = 128;
    constexpr int BN = 64 * TN;
    constexpr int BK = GEMM_BK;
    constexpr int LDK = BK + 8;                         // bf16 elements per LDS row (80 B)
    constexpr int A_IMG = BM * LDK, B_IMG = BN * LDK;   // elements per image (hi or lo)
    // LDS: A_hi | A_lo | B_hi | B_lo

    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    int bx, by, z;
    gemm_block_coords(bx, by, z);
    const int batch = z / g.splits;
    const int split = z - batch * g.splits;
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int kbase = split * g.K;

    // cloud-panel operands: the block's 128 rows lie in one cloud (panel_n % 128 == 0)
    const int m_cloud0 = (PANELS != 0) ? (m0 / g.panel_n) * g.panel_n : 0;
    const float* A = g.A + (long long)batch * g.sA + ((PANELS & 1) ? (long long)(m0 / g.panel_n) * g.a_cloud : 0);
    const float* B = g.B + (long long)batch * g.sB;
    float* C = g.C + (long long)batch * g.sC + (long long)split * g.sCsplit + ((PANELS & 2) ? (long long)(m0 / g.panel_n) * g.c_cloud : 0);

    float4 ra[X3Regs<BM>::NR], rb[X3Regs<BN>::NR];
    // k-tiles are walked from `kskew` on (and wrap): workgroups that sweep the reduction in step over operands with a
    // power-of-two row stride keep meeting on the same HBM channels otherwise (NetVLAD pooling: 4-KiB rows)
    const int nkt = g.K / GEMM_BK;
    const int kskew = (by * 5 + bx * 3) % nkt;
    auto load_tiles = [&](int kt) {
        kt += kskew;
        kt = kt >= nkt ? kt - nkt : kt;
        const int k0 = kbase + kt * BK;
        x3_load<A_KMAJOR, BM, KTAIL, PANELS & 1>(ra, A, g.lda, m0, g.M, k0, g.Ktot, tid, g.panel_ld, m_cloud0);
        x3_load<B_KMAJOR, BN, KTAIL>(rb, B, g.ldb, n0, g.N, k0, g.Ktot, tid);
    };
    auto store_tiles = [&]() {
        x3_store<A_KMAJOR, BM>(ra, smem16, smem16 + A_IMG, tid);
        x3_store<B_KMAJOR, BN>(rb, smem16 + 2 * A_IMG, smem16 + 2 * A_IMG + B_IMG, tid);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // One LDS buffer (40 KiB at TN = 2 -> four workgroups per CU), the next tile prefetched into registers while the
    // MFMAs of this one run: the split (VALU) and the MFMA phases of different workgroups overlap on a SIMD.
    const int nk = g.K / BK;
    const __bf16* ah = smem16 + (wm * 64 + col) * LDK + h * 8;
    const __bf16* al = ah + A_IMG;
    const __bf16* bh = smem16 + 2 * A_IMG + (wn * (32 * TN) + col) * LDK + h * 8;
    const __bf16* bl = bh + B_IMG;
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();          // every wave is done reading the previous tile
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a_hi[2], a_lo[2], b_hi[TN], b_lo[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) a_hi[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * LDK + s * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) b_hi[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * LDK + s * 16);
            if (g.prods == 1) {      // uniform: plain bf16 operands
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) a_lo[i] = *reinterpret_cast<const bf16x8*>(al + i * 32 * LDK + s * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) b_lo[j] = *reinterpret_cast<const bf16x8*>(bl + j * 32 * LDK + s * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue (accumulator layout identical to the f32-input 32x32 tile) ----
    gemm_epilogue<TN, PANELS>(g, acc, C, m0 + wm * 64, n0 + wn * (32 * TN), h, col, m_cloud0);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 GEMM with PRE-ARRANGED B fragments ("x3w"): C = act((A.B + bias) * scale + shift), A [M][K] row-major
// activations, B a (small) weight matrix.  lpd_gemm_prep_b splits B once into hi/lo bf16 and stores it in MFMA
// fragment order -- fragment (n-tile, k-step) = 64 lanes x 8 bf16 = 1 KiB contiguous -- so the main kernel never
// stages B through LDS and never splits it: each wave owns 32 WN columns of the output and streams their fragments
// straight from L2 into registers (one coalesced 16-B load per lane and fragment), while the A tile (128 rows, split
// hi/lo on the fly) is the only LDS traffic.  Same scheme as edge_mlp_x3_kernel; the generic kernel above re-splits
// and re-stages B in every block.
// ---------------------------------------------------------------------------------------------
// fragment order: frag[((nt * KS + ks) * 64 + lane) * 8 + j] = B[k = 16 ks + 8 (lane >> 5) + j][n = 32 nt + (lane & 31)]
__global__ void gemm_prep_b_kernel(const float* __restrict__ B, int ldb, int b_kmajor, int N, int K, int KS,
                                   __bf16* __restrict__ fhi, __bf16* __restrict__ flo, long long total, long long sB = 0)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (nt, ks, lane); blockIdx.y: matrix of a batch
    if (t >= total) return;
    B += (long long)blockIdx.y * sB;
    fhi += (long long)blockIdx.y * total * 16;      // a matrix's fragment set: hi array, then lo array
    flo += (long long)blockIdx.y * total * 16;
    const int lane = (int)(t & 63);
    const long long f = t >> 6;
    const int ks = (int)(f % KS), nt = (int)(f / KS);
    const int n = nt * 32 + (lane & 31);
    const int k0 = ks * 16 + (lane >> 5) * 8;
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        float v = 0.0f;
        if (n < N && k < K) v = b_kmajor ? B[(long long)k * ldb + n] : B[(long long)n * ldb + k];
        h[j] = (__bf16)v;
        l[j] = (__bf16)(v - (float)h[j]);
    }
    *reinterpret_cast<bf16x8*>(fhi + t * 8) = h;
    *reinterpret_cast<bf16x8*>(flo + t * 8) = l;
}

struct X3wArgs {
    const float* A;
    const __bf16* fhi;
    const __bf16* flo;
    float* C;
    int M, N, K, KS;       // KS = ceil(K / 16) fragments per n-tile
    int lda, ldc;
    const float* bias;
    const float* scale;
    const float* shift;
    int act;
    float slope;
    int accumulate;
    long long a_cloud, c_cloud;   // cloud-panel operands (see GemmArgs); 0 = row-major
    int panel_n, panel_ld;
    int rotate;                   // start the reduction of row tile t at chunk 5 t mod chunks (see the kernel)
    int prods;                    // 3 = split-bf16 (fp32-grade), 1 = a_hi b_hi only (bf16-storage training mode)
    double* stat_sum;             // train-mode BatchNorm statistics of the RAW product (+ bias), accumulated in the epilogue:
    double* stat_sumsq;           // column sums / sums of squares over the M rows (fp32 over a block's 128 rows, fp64 atomics); or null
    // A-operand transform (row-major A, ONE column block: N <= 128): the product takes act(a_scale[k] A[m][k] + a_shift[k]) and the
    // transformed rows are stored to a_out [M][a_ld] on the way -- a train-mode BatchNorm affine + activation applied where its only
    // dense consumer reads the raw tensor (conv3 -> bn3 -> NetVLAD assignment), instead of a pass of its own
    const float* a_scale;
    const float* a_shift;
    float* a_out;
    int a_ld;
    float a_ns;                   // negative slope of the activation (1 = none, 0 = ReLU)
    // per-problem weights over consecutive row ranges (lpd_gemm_x3w_batched): rows [b batch_rows, (b + 1) batch_rows) take the fragment
    // set b, frag_stride bf16 elements behind set b - 1 (hi and lo arrays alike); 0 = one weight for all rows
    int batch_rows;
    long long frag_stride;
    // a_bf16 != 0: A holds bf16 rows (lda in bf16 elements; row-major, K % 32 == 0): the bf16-storage training mode's conv3-map tensors.
    // Without an operand transform the values ARE the hi image (no lo image, prods = 2: a_hi b_lo + a_hi b_hi); with a_scale they are
    // widened, transformed and split like fp32 rows.
    int a_bf16;
    int a_out_bf16;               // the transformed rows are stored as bf16 (a_ld in bf16 elements)
    int c_bf16;                   // C receives bf16 values (ldc in bf16 elements; row-major, N % 32 == 0, no accumulate): the statistics are
                                  // still those of the fp32 accumulators
};

// ---------------------------------------------------------------------------------------------
// The kernel: block tile 128 x (128 WN), each wave a 128 x (32 WN) strip (4 x WN MFMA tiles), 32-deep chunks; WN = 2 for
// N >= 256.  Why: the issue slots of a SIMD are shared by its waves and an MFMA's 32 cycles hide at most
// ~6 other vector / LDS / memory instructions (MI355X_MICROARCH.md "vector-instruction ISSUE cost").  Per 16-deep k-step
// a wave issues 8 ds_read_b128 (A hi/lo, 4 row tiles) + 2 WN fragment loads for 12 WN MFMAs; at WN = 2 that is 0.5
// operand fetches per MFMA (the 128 x 128 kernels: 0.67 from LDS alone, plus their share of staging B).
// Memory waits are counted, never drained (vmcnt retires in order): per chunk the queue is
//     B(kc, step 1) | A(kc + 2) ............ | B(kc + 1, step 0)
// B fragments (L2 hits) are requested one k-step ahead; the fp32 rows of A (HBM / remote L2) TWO chunks ahead into one
// of two register sets, so the wait in front of the split + LDS store of chunk kc + 1 only covers loads issued a whole
// chunk (48 WN MFMAs) earlier and leaves the younger ones in flight.  (The first version requested the fragments right
// where it consumed them and drained A behind them every chunk: 640 us on conv3 against 505 us now, generic kernel 660.)
// A may be cloud panels (PANELS & 1), C too (PANELS & 2): a block's 128 rows lie in one cloud.
// ---------------------------------------------------------------------------------------------
// KC: k per chunk (2 or 4 MFMA k-steps); 32 everywhere (64-deep chunks -- 256 instead of 128 bytes per visit of a
// row-major A row -- changed nothing on the NetVLAD assignment).
// MODE (row-major operands only): bit 0 = A holds bf16 rows (X3wArgs::a_bf16; a bf16 a_out needs it), bit 1 = C receives bf16 values.
// Compile-time: as run-time branches the fp32 instantiations went from 9 to ~1000 spilled registers.
template <int WN, bool KTAIL, int PANELS, int KC, bool TALL = false, int MODE = 0>
__global__ __launch_bounds__(GEMM_THREADS, TALL ? 4 : 2) void gemm_x3w_wide_kernel(X3wArgs g)
{
    static_assert(MODE == 0 || PANELS == 0, "bf16 operands are row-major");
    constexpr bool A16 = (MODE & 1) != 0, C16 = (MODE & 2) != 0;
    constexpr bool A16T = A16 && (MODE & 4) != 0;      // bf16 rows behind an operand transform (a_scale): widened, transformed, split
    constexpr bool A16P = A16 && !A16T;                // plain bf16 rows: they ARE the hi image, no lo image, two products
    // TALL (N <= 64, WN = 1): two column tiles only -- the waves split the ROWS as well (wave = column tile + 2 x row half,
    // two row tiles each) instead of two of the four multiplying tiles that are never stored.
    constexpr int RT = TALL ? 2 : 4;            // row tiles per wave
    static_assert(!TALL || WN == 1, "tall blocks are one column tile per wave");
    constexpr int X3V_KC = KC;
    constexpr int X3V_LDK = X3V_KC + 8;        // 80- / 144-byte LDS rows: conflict-free ds_read_b128
    constexpr int X3V_IMG = 128 * X3V_LDK;     // one image (hi or lo)
    constexpr int NF4 = 128 * KC / 4 / GEMM_THREADS;   // float4 of A per thread and chunk (4 or 8)
    constexpr int KSC = KC / 16;               // k-steps per chunk
    constexpr int QR = KC / 4;                 // float4 per row of the chunk
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];   // [2 buffers][hi | lo][128][LDK]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int col = lane & 31;
    int bx, by, bz;
    gemm_block_coords(bx, by, bz);
    const int m0 = by * 128;
    const int nt0 = TALL ? bx * 2 + (wave & 1) : (bx * 4 + wave) * WN;   // this wave's first 32-column tile
    const int rt0 = TALL ? (wave >> 1) * 2 : 0;                          // ... and first 32-row tile
    const int NT = (g.N + 31) >> 5;
    const int nchunks = (g.K + X3V_KC - 1) / X3V_KC;
    // Row tile `by` walks the reduction from chunk `skew` on (and wraps): with a row-major A whose row stride is a power
    // of two (the 4-KiB rows of the 1024-wide feature map) the 128 row pieces of a chunk share their low address bits, and
    // workgroups that all sweep k from 0 together keep landing on the same HBM channels.
    const int skew = g.rotate ? (by * 5) % nchunks : 0;
    auto phys = [&](int l) { const int c = l + skew; return c >= nchunks ? c - nchunks : c; };   // logical -> actual chunk

    const int cloud = (PANELS != 0) ? m0 / g.panel_n : 0;
    const int m_cloud0 = cloud * g.panel_n;
    const float* A = g.A + ((PANELS & 1) ? (long long)cloud * g.a_cloud : 0);
    float* C = g.C + ((PANELS & 2) ? (long long)cloud * g.c_cloud : 0);

    // A staging: 128 x 32 fp32 per chunk = 4 float4 per thread, two sets (chunk parity).
    //   row-major: register e holds row (e*256 + tid) / 8, k quad (e*256 + tid) % 8 (a row's 128 B are 8 lanes)
    //   panels   : register e holds panel e of the chunk (8 channels), row tid / 2, half tid % 2 (one panel's 128 rows
    //              x 32 B = 4 KiB contiguous per instruction)
    float4 ra[2][NF4];
    float4 rsc[2], rsh[2];      // a_scale / a_shift quad of the chunk in flight (row-major staging: a thread's k quad is f % QR for every e)
    float4 rsc8[2], rsh8[2];    // ... the second quad of a bf16 piece (eight channels per thread)
    constexpr int QR8 = KC / 8, NF8 = NF4 / 2;      // bf16 rows: 16-byte pieces per row of the chunk, pieces per thread
    int rk0[2] = {0, 0};
    const float* a_base;        // loop-invariant part of this thread's A addresses
    {
        if constexpr (PANELS & 1) {
            int row = m0 + (tid >> 1);
            a_base = A + (long long)(row - m_cloud0) * 8 + (tid & 1) * 4;
        } else a_base = A;
    }
    auto load_a = [&](int kc, float4 (&r)[NF4], int set) {
        const int k0 = kc * X3V_KC;
        if constexpr (!(PANELS & 1)) {
            if (A16T || (!A16P && g.a_scale)) {      // uniform
                rk0[set] = k0;
                if constexpr (A16) {      // (bf16 rows: a thread's piece is EIGHT channels, k0 + (tid % QR8) * 8 .. + 7; K % 32 == 0)
                    rsc[set] = *reinterpret_cast<const float4*>(g.a_scale + k0 + (tid % QR8) * 8);
                    rsh[set] = *reinterpret_cast<const float4*>(g.a_shift + k0 + (tid % QR8) * 8);
                    rsc8[set] = *reinterpret_cast<const float4*>(g.a_scale + k0 + (tid % QR8) * 8 + 4);
                    rsh8[set] = *reinterpret_cast<const float4*>(g.a_shift + k0 + (tid % QR8) * 8 + 4);
                } else {
                    rsc[set] = ld4_guard(g.a_scale, k0 + (tid % QR) * 4, g.K);
                    rsh[set] = ld4_guard(g.a_shift, k0 + (tid % QR) * 4, g.K);
                }
            }
        }
        if constexpr (A16) {
            // bf16 rows (round 6): one 16-byte piece = eight channels per load (8-byte pieces before: twice the load instructions for the
            // same bytes); register e holds row (e * 256 + tid) / QR8, piece (e * 256 + tid) % QR8 as raw bits
#pragma unroll
            for (int e = 0; e < NF8; ++e) {
                const int f = e * GEMM_THREADS + tid;
                int row = m0 + f / QR8;
                row = row < g.M ? row : g.M - 1;
                const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a_base) + (long long)row * g.lda + k0 + (f % QR8) * 8);
                r[e] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < NF4; ++e) {
            if constexpr (PANELS & 1) {
                const int kp = (k0 >> 3) + e;          // panel index (K % 8 == 0)
                if (!KTAIL || kp * 8 < g.K) r[e] = *reinterpret_cast<const float4*>(a_base + (long long)kp * g.panel_ld * 8);
                else r[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int f = e * GEMM_THREADS + tid;
                int row = m0 + f / QR;
                row = row < g.M ? row : g.M - 1;
                if constexpr (KTAIL) r[e] = ld4_guard(a_base + (long long)row * g.lda, k0 + (f % QR) * 4, g.K);
                else r[e] = *reinterpret_cast<const float4*>(a_base + (long long)row * g.lda + k0 + (f % QR) * 4);
            }
        }
    };
    auto store_a = [&](int buf, const float4 (&r)[NF4], int set) {
        __bf16* hi_img = smem16 + buf * 2 * X3V_IMG;
        __bf16* lo_img = hi_img + X3V_IMG;
        if constexpr (A16) {
#pragma unroll
            for (int e = 0; e < NF8; ++e) {
                const int f = e * GEMM_THREADS + tid;
                const int rr = f / QR8, k8 = f % QR8;
                const unsigned w[4] = {__float_as_uint(r[e].x), __float_as_uint(r[e].y), __float_as_uint(r[e].z), __float_as_uint(r[e].w)};
                if constexpr (A16P) {      // the rows ARE the hi image
                    *reinterpret_cast<uint4*>(hi_img + rr * X3V_LDK + k8 * 8) = make_uint4(w[0], w[1], w[2], w[3]);
                    continue;
                }
                // behind an operand transform: widened, transformed (multiply, then add -- not fused: the bits of lpd_affine_act), stored
                const float scv[8] = {rsc[set].x, rsc[set].y, rsc[set].z, rsc[set].w, rsc8[set].x, rsc8[set].y, rsc8[set].z, rsc8[set].w};
                const float shv[8] = {rsh[set].x, rsh[set].y, rsh[set].z, rsh[set].w, rsh8[set].x, rsh8[set].y, rsh8[set].z, rsh8[set].w};
                float v[8];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) { v[2 * p2] = __uint_as_float(w[p2] << 16); v[2 * p2 + 1] = __uint_as_float(w[p2] & 0xffff0000u); }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    v[c] = scv[c] * v[c] + shv[c];
                    v[c] = fmaxf(v[c], 0.0f) + g.a_ns * fminf(v[c], 0.0f);
                }
                const int m = m0 + rr, kk = rk0[set] + k8 * 8;
                const bool put = g.a_out && m < g.M;
                if (g.a_out_bf16) {
                    // the transformed map IS a bf16 tensor (stored here, or re-formed by every consumer's loader): the product takes the
                    // rounded values, so that all of them see the same map; they ARE the hi image, the lo image is never read (prods = 2)
                    bf16x8 ob;
#pragma unroll
                    for (int c = 0; c < 8; ++c) ob[c] = (__bf16)v[c];
                    if (put) *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(g.a_out) + (long long)m * g.a_ld + kk) = ob;
                    *reinterpret_cast<bf16x8*>(hi_img + rr * X3V_LDK + k8 * 8) = ob;
                    continue;
                }
                if (put) {
                    *reinterpret_cast<float4*>(g.a_out + (long long)m * g.a_ld + kk) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(g.a_out + (long long)m * g.a_ld + kk + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                bf16x4 hh, ll;
                split4(v[0], v[1], v[2], v[3], hh, ll);
                *reinterpret_cast<bf16x4*>(hi_img + rr * X3V_LDK + k8 * 8) = hh;
                *reinterpret_cast<bf16x4*>(lo_img + rr * X3V_LDK + k8 * 8) = ll;
                split4(v[4], v[5], v[6], v[7], hh, ll);
                *reinterpret_cast<bf16x4*>(hi_img + rr * X3V_LDK + k8 * 8 + 4) = hh;
                *reinterpret_cast<bf16x4*>(lo_img + rr * X3V_LDK + k8 * 8 + 4) = ll;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < NF4; ++e) {
            int rr, k4;
            if constexpr (PANELS & 1) { rr = tid >> 1; k4 = e * 2 + (tid & 1); }
            else { const int f = e * GEMM_THREADS + tid; rr = f / QR; k4 = f % QR; }
            float4 v = r[e];
            if constexpr (!(PANELS & 1)) {
                if constexpr (A16) {
                    const unsigned w0 = __float_as_uint(v.x), w1 = __float_as_uint(v.y);
                    if constexpr (A16P) {      // the rows ARE the hi image
                        *reinterpret_cast<uint2*>(hi_img + rr * X3V_LDK + k4 * 4) = make_uint2(w0, w1);
                        continue;
                    }
                    v = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
                }
                if (A16T || (!A16P && g.a_scale)) {      // uniform: the transformed operand, stored on the way
                    const float4 sc = rsc[set], sh = rsh[set];
                    // (multiply, then add -- not fused: the same bits as lpd_affine_act)
                    v.x = sc.x * v.x + sh.x; v.y = sc.y * v.y + sh.y; v.z = sc.z * v.z + sh.z; v.w = sc.w * v.w + sh.w;
                    v.x = fmaxf(v.x, 0.0f) + g.a_ns * fminf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f) + g.a_ns * fminf(v.y, 0.0f);
                    v.z = fmaxf(v.z, 0.0f) + g.a_ns * fminf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f) + g.a_ns * fminf(v.w, 0.0f);
                    const int m = m0 + rr, kk = rk0[set] + k4 * 4;
                    const bool put = g.a_out && m < g.M && (!KTAIL || kk + 3 < g.K);
                    if (A16 && g.a_out_bf16) {
                        // the transformed map IS a bf16 tensor (stored here, or re-formed by every consumer's loader): the product takes
                        // the rounded values, so that all of them see the same map
                        bf16x4 ob;
                        ob[0] = (__bf16)v.x; ob[1] = (__bf16)v.y; ob[2] = (__bf16)v.z; ob[3] = (__bf16)v.w;
                        if (put) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.a_out) + (long long)m * g.a_ld + kk) = ob;
                        // (the rounded values ARE the hi image and the lo image is zero: prods = 2 never reads it -- no split, no second store)
                        *reinterpret_cast<bf16x4*>(hi_img + rr * X3V_LDK + k4 * 4) = ob;
                        continue;
                    } else if (put) {
                        *reinterpret_cast<float4*>(g.a_out + (long long)m * g.a_ld + kk) = v;
                    }
                }
            }
            bf16x4 hh, ll;
            split4(v.x, v.y, v.z, v.w, hh, ll);
            *reinterpret_cast<bf16x4*>(hi_img + rr * X3V_LDK + k4 * 4) = hh;
            *reinterpret_cast<bf16x4*>(lo_img + rr * X3V_LDK + k4 * 4) = ll;
        }
    };

    // B fragments: set 0 = the chunk's first k-step, set 1 = its second
    bf16x8 b_hi[2][WN], b_lo[2][WN];
    const __bf16* fh[WN];
    const __bf16* fl[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int nt = nt0 + j < NT ? nt0 + j : 0;   // strips past N read tile 0 (never stored)
        const long long fb = g.batch_rows ? (long long)(m0 / g.batch_rows) * g.frag_stride : 0;      // this row tile's problem
        fh[j] = g.fhi + fb + ((long long)nt * g.KS * 64 + lane) * 8;
        fl[j] = g.flo + fb + ((long long)nt * g.KS * 64 + lane) * 8;
    }
    auto load_b = [&](int ks, int set) {
        ks = min(ks, g.KS - 1);                       // past the end: any valid fragment (its A columns are zero)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            b_hi[set][j] = *reinterpret_cast<const bf16x8*>(fh[j] + (long long)ks * 512);
            b_lo[set][j] = *reinterpret_cast<const bf16x8*>(fl[j] + (long long)ks * 512);
        }
    };

    f32x16 acc[RT][WN];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto kstep = [&](const __bf16* ah, const __bf16* al, int s, int set) {
        bf16x8 a_hi[RT], a_lo[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            a_hi[i] = *reinterpret_cast<const bf16x8*>(ah + (rt0 + i) * 32 * X3V_LDK + s * 16);
            if constexpr (!A16P) a_lo[i] = *reinterpret_cast<const bf16x8*>(al + (rt0 + i) * 32 * X3V_LDK + s * 16);
        }
        if constexpr (A16P) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[set][j], acc[i][j], 0, 0, 0);
        } else if constexpr (A16T) {
            if (g.prods == 3) {      // uniform (a transformed map that is NOT rounded to bf16 again has a lo image)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], b_hi[set][j], acc[i][j], 0, 0, 0);
            }
            if (g.prods != 1) {      // uniform
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[set][j], acc[i][j], 0, 0, 0);
            }
        } else if (g.prods != 1) {      // uniform
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], b_hi[set][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[set][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[set][j], acc[i][j], 0, 0, 0);
    };
    // one chunk; `cur` / `nxt`: the register sets holding A(kc + 1) (loaded during the previous chunk) / receiving A(kc + 2)
    auto chunk = [&](int kc, float4 (&nxt)[NF4], const float4 (&cur)[NF4], int nset) {
        const int buf = kc & 1;
        const __bf16* ah = smem16 + buf * 2 * X3V_IMG + col * X3V_LDK + h * 8;
        const __bf16* al = ah + X3V_IMG;
        load_b(phys(kc) * KSC + 1, 1);
        load_a(phys(min(kc + 2, nchunks - 1)), nxt, nset);   // unconditional (the tail re-reads the last chunk): a branch here makes the
                                                 // compiler count vmcnt for the path without these loads and over-wait
#pragma unroll
        for (int s = 0; s < KSC; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            kstep(ah, al, s, s & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < KSC) load_b((s + 2 < KSC ? phys(kc) : phys(min(kc + 1, nchunks - 1))) * KSC + (s + 2) % KSC, s & 1);   // the set just consumed receives the step after next
        }
        if (kc + 1 < nchunks) store_a(buf ^ 1, cur, nset ^ 1);
        __syncthreads();
    };

    load_b(phys(0) * KSC, 0);
    load_a(phys(0), ra[0], 0);
    store_a(0, ra[0], 0);                                 // (before set 0's scale / shift quad is overwritten below)
    if (nchunks > 1) load_a(phys(1), ra[1], 1);
    __syncthreads();
    for (int kc = 0; kc < nchunks; kc += 2) {
        chunk(kc, ra[0], ra[1], 0);                       // A(kc + 1) sits in set 1, A(kc + 2) goes to set 0
        if (kc + 1 < nchunks) chunk(kc + 1, ra[1], ra[0], 1);
    }

    const float ns = g.act == 0 ? 1.0f : (g.act == 1 ? 0.0f : g.slope);
    // Round 6: the RAW result of a whole row tile (no scale / activation / accumulate: the train-mode products whose BatchNorm follows)
    // leaves without the generic epilogue's per-element affine, activation, row guard and 64-bit address arithmetic -- that epilogue
    // cost conv3 of the bf16 training step 130 of its 640 us (the same product without statistics and with fp32 stores: 505 us)
    const bool raw = !(PANELS & 2) && !g.scale && g.act == 0 && !g.accumulate && m0 + 128 <= g.M;      // uniform
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = (nt0 + j) * 32 + col;
        if (n >= g.N) continue;
        float bi = 0.f, sc = 1.f, sh = 0.f;
        if (g.bias) bi = g.bias[n];
        if (raw) {
            float su = 0.f, sq = 0.f;
            const long long row0 = (long long)(m0 + rt0 * 32 + 4 * h) * g.ldc + n;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = i * 32 + (r & 3) + 8 * (r >> 2);      // row inside this lane's share (wave-uniform)
                    const float v = acc[i][j][r] + bi;
                    su += v;
                    sq += v * v;
                    if constexpr (C16) {      // the lane pair (n, n + 1) leaves as one 4-byte word, written by the even lane
                        const unsigned mine = __builtin_bit_cast(unsigned short, (__bf16)v);
                        const unsigned other = lpd_lane_xor1(mine);
                        if (!(col & 1)) *reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(C) + row0 + (long long)ro * g.ldc) = mine | (other << 16);
                    } else C[row0 + (long long)ro * g.ldc] = v;
                }
            if (g.stat_sum) {        // uniform
                su += __shfl_xor(su, 32, 64);
                sq += __shfl_xor(sq, 32, 64);
                if (h == 0) {
                    const size_t rofs = (size_t)((m0 >> 7) % LPD_STAT_REPLICAS) * 2 * LPD_STAT_CMAX;
                    atomicAdd(g.stat_sum + rofs + n, (double)su);
                    atomicAdd(g.stat_sumsq + rofs + n, (double)sq);
                }
            }
            continue;
        }
        if (g.scale) { sc = g.scale[n]; sh = g.shift[n]; }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (rt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bi;
                v = v * sc + sh;
                if (g.act == 3) v = lpd_sigmoid(v);
                else v = fmaxf(v, 0.0f) + ns * fminf(v, 0.0f);
                float* dst;
                if constexpr (PANELS & 2) dst = C + ((long long)(n >> 3) * g.panel_ld + (m - m_cloud0)) * 8 + (n & 7);
                else {
                    if constexpr (C16) {      // the lane pair (n, n + 1) leaves as one 4-byte word, written by the even lane
                        const unsigned mine = __builtin_bit_cast(unsigned short, (__bf16)v);
                        const unsigned other = lpd_lane_xor1(mine);
                        if (!(col & 1)) *reinterpret_cast<unsigned*>(reinterpret_cast<uint16_t*>(C) + (long long)m * g.ldc + n) = mine | (other << 16);
                        continue;
                    }
                    dst = C + (long long)m * g.ldc + n;
                }
                if (g.accumulate) v += *dst;
                *dst = v;
            }
        if (g.stat_sum) {        // uniform.  This lane: column n, up to 16 RT rows; its partner (lane ^ 32) the other rows of the tiles
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (rt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = m < g.M ? acc[i][j][r] + bi : 0.f;
                    su += v;
                    sq += v * v;
                }
            su += __shfl_xor(su, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (h == 0) {        // into the replica of this row block (lpd_common.h, column statistics across blocks)
                const size_t rofs = (size_t)((m0 >> 7) % LPD_STAT_REPLICAS) * 2 * LPD_STAT_CMAX;
                atomicAdd(g.stat_sum + rofs + n, (double)su);
                atomicAdd(g.stat_sumsq + rofs + n, (double)sq);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split-K partial products for a FEW rows (M <= 64) against a k-major weight matrix: the NetVLAD hidden projection
// [B, 65536] x [65536, 256] (util/PointNetVlad.py:76) and the T-Net fully-connected layers.  The MFMA kernel fills 32 of
// its 128 tile rows and runs 128 workgroups, each streaming its weight slice through a 2-barrier LDS pipeline
// (82 us at B = 32: 1.1 TB/s over the 67 MB of weights).  Here one workgroup owns one split (a KC-deep slice of K) and
// 256 columns: the X slice sits in LDS ([M][KC], read as broadcast float4), every thread streams ONE column of W
// (a wave reads 256 contiguous bytes per k) and keeps 16 rows of partial sums in registers; row groups of 16 share the
// columns.  fp32 FMA chains in k order.  Slab layout = the MFMA kernel's ([split][M][N]), same reduce kernel.
// ---------------------------------------------------------------------------------------------
constexpr int SMALLM_KC = 128;   // k per split (256: 52 us for the NetVLAD hidden projection at 32 rows, 128: 43 us, 64: 50 us)
typedef float smallm_f2 __attribute__((ext_vector_type(2)));
typedef float smallm_f4 __attribute__((ext_vector_type(4)));

// RG row groups of 16 rows.  LDS image: row PAIRS interleaved per k ([pair][k][2]) so that one ds_read_b128 brings two
// k steps of two rows and the partial sums advance two rows per v_pk_fma_f32.
template <int RG>
__global__ __launch_bounds__(256 * RG) void linear_smallm_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                                  float* __restrict__ slabs, int M, int N, int K)
{
    constexpr int KC = SMALLM_KC;
    extern __shared__ __attribute__((aligned(16))) float xs_[];   // [8 RG pairs][KC][2]
    const int tid = threadIdx.x;
    const int c = tid & 255, rg = tid >> 8;
    constexpr int nrows = RG * 16;
    const int split = blockIdx.x;
    const int k0 = split * KC;
    for (int i = tid; i < nrows * KC; i += 256 * RG) {
        const int r = i / KC, k = i % KC;
        xs_[((r >> 1) * KC + k) * 2 + (r & 1)] = (r < M && k0 + k < K) ? X[(long long)r * ldx + k0 + k] : 0.0f;
    }
    __syncthreads();
    const int n = blockIdx.y * 256 + c;
    const float* wp = W + min(n, N - 1);
    const float* xr = xs_ + rg * 8 * KC * 2;
    smallm_f2 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = (smallm_f2)(0.0f);
    // 16 weight rows in flight per thread (4 KiB per wave): with 4 the kernel ran at the latency-bound 1 TB/s
    auto wrow = [&](int k) { return wp[(long long)min(k0 + k, K - 1) * ldw]; };   // rows past K meet zeros of X
    float wn[16], wc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) wn[j] = wrow(j);
    for (int k = 0; k < KC; k += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) wc[j] = wn[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) wn[j] = wrow(min(k + 16 + j, KC - 1));   // (the last step re-reads: no branch around loads)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const smallm_f4 x = *reinterpret_cast<const smallm_f4*>(xr + (p * KC + k + 2 * q) * 2);
                acc[p] = __builtin_elementwise_fma(x.lo, (smallm_f2)(wc[2 * q]), acc[p]);
                acc[p] = __builtin_elementwise_fma(x.hi, (smallm_f2)(wc[2 * q + 1]), acc[p]);
            }
    }
    if (n < N) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int m = rg * 16 + 2 * p;
            if (m < M) slabs[((long long)split * M + m) * N + n] = acc[p].x;
            if (m + 1 < M) slabs[((long long)split * M + m + 1) * N + n] = acc[p].y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5: the same product as a weight STREAM for M <= 4 rows and a deep reduction (the NetVLAD hidden projection [B, 65536] x
// [65536, 256]: 67 MB of weights, the whole cost).  linear_smallm_kernel gives every thread one weight COLUMN -- 4-byte loads, 256
// bytes per wave-instruction, 2.3 TB/s (29 us at 32 rows; 33 us at ONE row, where the forward's critical path has nothing else to
// hide it behind; at 32 rows the column form stays: see the dispatch).  Here a wave reads whole 1-KiB weight ROWS, 16 bytes per lane (lane l: columns 4 l .. 4 l + 3), FR_DEPTH of them in
// flight; the four waves of a block take the k-rows of the block's slice in turn, every lane keeps MT rows x 4 columns of partial
// sums, the X slice sits in LDS as [k][MT] (one broadcast read per four rows), and the waves' partial sums are added through LDS
// before ONE slab per block is written (K / FR_KPER slabs instead of K / 128).  fp32 FMA chains; the slab reduce is the existing one.
// ---------------------------------------------------------------------------------------------
constexpr int FR_KPER = 256;     // k-rows per block (K = 65536: 256 blocks, one per CU; 128: 26.5 us against 23.4 at one row)
constexpr int FR_DEPTH = 8;      // weight rows in flight per wave (8 KiB)

template <int MT>
__global__ __launch_bounds__(256) void linear_fewrows_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                              float* __restrict__ slabs, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) float fr_lds[];      // [FR_KPER][MT] X slice, then the reduction buffer
    float* xs = fr_lds;
    float* red = fr_lds + FR_KPER * MT;                                   // [2][MT][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * FR_KPER;
    const int kn = min(FR_KPER, K - k0);                                  // k-rows of this block
    for (int i = tid; i < FR_KPER * MT; i += 256) {
        const int k = i / MT, m = i - k * MT;
        xs[i] = (m < M && k < kn) ? X[(long long)m * ldx + k0 + k] : 0.0f;
    }
    __syncthreads();
    const int n0 = blockIdx.y * 256 + lane * 4;
    const float* wp = W + n0;
    float4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto wrow = [&](int k) { return *reinterpret_cast<const float4*>(wp + (long long)(k0 + min(k, kn - 1)) * ldw); };   // past the slice: re-read (X is zero there)
    float4 wn[FR_DEPTH];
#pragma unroll
    for (int d = 0; d < FR_DEPTH; ++d) wn[d] = wrow(wave + 4 * d);
    for (int kk = wave; kk < kn; kk += 4 * FR_DEPTH) {
        float4 wc[FR_DEPTH];
#pragma unroll
        for (int d = 0; d < FR_DEPTH; ++d) wc[d] = wn[d];
#pragma unroll
        for (int d = 0; d < FR_DEPTH; ++d) wn[d] = wrow(kk + 4 * (FR_DEPTH + d));
#pragma unroll
        for (int d = 0; d < FR_DEPTH; ++d) {
            const int k = min(kk + 4 * d, FR_KPER - 1);                   // rows past kn meet zeros of X
            const float* xr = xs + k * MT;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float x = xr[m];
                acc[m].x = fmaf(x, wc[d].x, acc[m].x); acc[m].y = fmaf(x, wc[d].y, acc[m].y);
                acc[m].z = fmaf(x, wc[d].z, acc[m].z); acc[m].w = fmaf(x, wc[d].w, acc[m].w);
            }
        }
    }
    // waves 2, 3 -> LDS, waves 0, 1 add; wave 1 -> LDS, wave 0 adds and stores the block's slab
    __syncthreads();                                                      // (xs is dead: red may alias nothing of it, but keep the phases apart)
    if (wave >= 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<float4*>(red + ((wave - 2) * MT + m) * 256 + lane * 4) = acc[m];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 o = *reinterpret_cast<const float4*>(red + (wave * MT + m) * 256 + lane * 4);
            acc[m].x += o.x; acc[m].y += o.y; acc[m].z += o.z; acc[m].w += o.w;
        }
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) *reinterpret_cast<float4*>(red + m * 256 + lane * 4) = acc[m];
    }
    __syncthreads();
    if (wave == 0 && n0 < N) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < M) {
                const float4 o = *reinterpret_cast<const float4*>(red + m * 256 + lane * 4);
                *reinterpret_cast<float4*>(slabs + ((long long)blockIdx.x * M + m) * N + n0) =
                    make_float4(acc[m].x + o.x, acc[m].y + o.y, acc[m].z + o.z, acc[m].w + o.w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6: the same product for 5 .. 32 rows on the f32-input MFMA (the NetVLAD hidden projection of an eval slice: [32, 65536] x
// [65536, 256], 67 MB of weights read once).  linear_smallm_kernel spends 1.07 GFLOP on the VALU from 4-byte column loads and writes
// 512 slabs (29 us + 13 us of slab reduction at 32 rows = 0.20 of the HBM roofline for the weights).  Here the weights are a STREAM of
// whole rows through the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32 = the k-ordered fmaf chain):
//   * block = 256 k-rows x 256 columns (one block per CU at K = 65536), X slice [256][32] in LDS (zero rows past M);
//   * wave (column half ch, k phase kq of 4): per k-pair ONE 16-byte load per lane -- lane (c, h) reads W[k0 + 2 i + h][128 ch + 4 c .. + 3],
//     512 contiguous bytes per half-wave -- feeds FOUR MFMAs (tile j holds columns 128 ch + 4 c + j: a permutation of the columns, undone
//     by the float4 stores of the epilogue); R32_DEPTH loads in flight per lane; 4 x 64 cycles of MFMA per KiB and wave = 9 TB/s for
//     the chip: the matrix cores stay just ahead of HBM;
//   * the four k phases of a column half are added through LDS, ONE slab per block (K / 256 slabs), the existing slab reduce.
// ---------------------------------------------------------------------------------------------
constexpr int R32_KPER = 256;    // k-rows per block
constexpr int R32_DEPTH = 16;    // weight loads in flight per lane (16 KiB per wave, 128 KiB per block: the stream is latency-bound below that)
constexpr int R32_XLD = 33;      // LDS row stride of the X slice (k-major rows of 32 floats, padded: staging writes walk k)
constexpr int R32_PH = 4;        // k phases: wave (column half ch, phase kq) takes the k-pairs i = kq, kq + 4, ...
constexpr int R32_LDS_FLOATS = 2 * 2 * 4 * 16 * 64;      // partner exchange: [stage slot 0..1][column half][tile][r][lane]

__global__ __launch_bounds__(512) void linear_rows32_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                             float* __restrict__ slabs, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) float r32_lds[];      // X slice [k][m] (33 KiB); afterwards the partner exchange (64 KiB)
    float* xs = r32_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = wave & 1, kq = wave >> 1;
    const int k0 = blockIdx.x * R32_KPER;
    const int kn = min(R32_KPER, K - k0);
    const int c = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * 256 + ch * 128 + 4 * c;
    const float* wp = W + n0;
    // this wave's k-pairs: i = kq + 4 t, t = 0 .. NT - 1; its rows 2 i + h (rows past the slice are re-reads of its last row: X is zero there)
    auto wrow = [&](int t) { return *reinterpret_cast<const float4*>(wp + (long long)(k0 + min(2 * (kq + R32_PH * t) + h, kn - 1)) * ldw); };
    constexpr int NT = R32_KPER / 2 / R32_PH;      // 32 k-pairs per wave
    static_assert(NT == 2 * R32_DEPTH, "two rounds of R32_DEPTH loads");
    float4 wn[R32_DEPTH];
    {   // X slice: 16 values per thread, all requested before the first is stored (a rolled loop waited for every load in turn:
        // 16 round trips to a cold X, 20 of the kernel's 31 us) and BEFORE the weight rows: loads return in order, so the slice is
        // staged while the first round of weights is still on its way
        constexpr int NX = 32 * R32_KPER / 512;
        float xv[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 512 * j, m = i / R32_KPER, k = i - m * R32_KPER;
            xv[j] = X[(long long)min(m, M - 1) * ldx + k0 + min(k, kn - 1)];      // (no branch around the load; masked below)
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int d = 0; d < R32_DEPTH; ++d) wn[d] = wrow(d);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 512 * j, m = i / R32_KPER, k = i - m * R32_KPER;
            xv[j] = (m < M && k < kn) ? xv[j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 512 * j, m = i / R32_KPER, k = i - m * R32_KPER;
            xs[k * R32_XLD + m] = xv[j];
        }
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int d = 0; d < R32_DEPTH; ++d) {
            const int t = rd * R32_DEPTH + d;
            const float a = xs[(2 * (kq + R32_PH * t) + h) * R32_XLD + c];
            const float4 w = wn[d];
            if (rd == 0) {                                     // the register set is free again: request the second round's row NOW
                wn[d] = wrow(R32_DEPTH + d);
                asm volatile("" ::: "memory");                 // (the scheduler sank these loads behind the round's 64 MFMAs)
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.w, acc[3], 0, 0, 0);
        }
    }
    // phases 2, 3 -> LDS, phases 0, 1 add; phase 1 -> LDS, phase 0 adds and stores the block's slab.  Accumulator element r of lane
    // (c, h): row (r & 3) + 8 (r >> 2) + 4 h, columns n0 .. n0 + 3 from the four tiles
    __syncthreads();                                               // the X slice is dead
    float* red = r32_lds + ((kq & 1) * 2 + ch) * (4 * 16 * 64);
    if (kq >= 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (kq < 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] += red[(j * 16 + r) * 64 + lane];
    }
    __syncthreads();
    if (kq == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (kq == 0 && n0 < N) {
        const float* r1 = r32_lds + (1 * 2 + ch) * (4 * 16 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) {
                float4 o;
                o.x = acc[0][r] + r1[(0 * 16 + r) * 64 + lane];
                o.y = acc[1][r] + r1[(1 * 16 + r) * 64 + lane];
                o.z = acc[2][r] + r1[(2 * 16 + r) * 64 + lane];
                o.w = acc[3][r] + r1[(3 * 16 + r) * 64 + lane];
                *reinterpret_cast<float4*>(slabs + ((long long)blockIdx.x * M + m) * N + n0) = o;
            }
        }
    }
}

// sums split-K slabs and applies the epilogue.  one thread per output element.
__global__ void gemm_splitk_reduce_kernel(const float* slabs, float* C, int M, int N,   // (C may be the slabs: staged reduction)
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
    v = lpd_act_any(v, act, slope);
    float* dst = C + (long long)batch * sC_batch + (long long)m * ldc + n;
    if (accumulate) v += *dst;
    *dst = v;
}

template <bool AK, bool BK_, int TN, bool KTAIL>
int gemm_launch_t(const GemmArgs& g, int batch, hipStream_t stream);
template <bool AK, bool BK_, int TN, bool KTAIL>
int gemm_x3_launch_t(const GemmArgs& g, int batch, hipStream_t stream);

// panel-major A and/or C: row-major-A kernels only, 128-column tiles, whole k-tiles
template <bool BK_, int PANELS>
int gemm_panel_launch(const GemmArgs& g, bool x3, hipStream_t stream)
{
    constexpr int BM = 128, BN = 128;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, 1);
    if (x3) {
        constexpr int LDK = GEMM_BK + 8;
        size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(__bf16);
        auto kern = gemm_bf16x3_kernel<false, BK_, 2, false, PANELS>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, stream, g);
    } else {
        constexpr int LDA = BM + 1, LDB = BK_ ? BN + 4 : BN + 1;
        size_t lds = (size_t)2 * GEMM_BK * (LDA + LDB) * sizeof(float);
        auto kern = gemm_f32_kernel<false, BK_, 2, false, PANELS>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, stream, g);
    }
    LPD_CHECK_LAUNCH("lpd_gemm(panels)");
    return LPD_OK;
}

template <bool AK, bool BK_, int TN>
int gemm_launch(const GemmArgs& g, int batch, bool x3, hipStream_t stream)
{
    // every split covers whole 32-deep k-tiles of real data <=> Ktot is a multiple of 32 and splits divide it evenly
    const bool ktail = (g.Ktot % GEMM_BK) != 0 || (long long)g.K * g.splits != g.Ktot;
    if (x3) {
        return ktail ? gemm_x3_launch_t<AK, BK_, TN, true>(g, batch, stream) : gemm_x3_launch_t<AK, BK_, TN, false>(g, batch, stream);
    }
    return ktail ? gemm_launch_t<AK, BK_, TN, true>(g, batch, stream) : gemm_launch_t<AK, BK_, TN, false>(g, batch, stream);
}

template <bool AK, bool BK_, int TN, bool KTAIL>
int gemm_x3_launch_t(const GemmArgs& g, int batch, hipStream_t stream)
{
    constexpr int BM = 128, BN = 64 * TN, LDK = GEMM_BK + 8;
    size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(__bf16);
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, batch * g.splits);
    auto kern = gemm_bf16x3_kernel<AK, BK_, TN, KTAIL>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, stream, g);
    LPD_CHECK_LAUNCH("lpd_gemm(bf16x3)");
    return LPD_OK;
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
static int gemm_entry(int mode /* 0 f32-input MFMA, 3 split-bf16, 1 plain bf16 */, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC,
                      int splits, float* splitk_ws, const float* bias, const float* scale, const float* shift,
                      int act, float slope, int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream_)
{
    const bool x3 = mode != 0;
    const bool a_panels = a_cloud != 0, c_panels = c_cloud != 0;
    LPD_CHECK_ARG(!(a_panels || c_panels) || (!a_kmajor && batch == 1 && splits == 1 && K % 32 == 0 && panel_n > 0 &&
                                              panel_n % 128 == 0 && M % panel_n == 0 && panel_ld >= panel_n),
                  "lpd_gemm: cloud-panel A / C need a_kmajor = 0, batch = 1, splits = 1, K %% 32 == 0, points per cloud %% 128 == 0");
    if (a_panels) lda = 4;   // unused; keeps the alignment checks below happy
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
    g.a_cloud = a_cloud;
    g.c_cloud = c_cloud;
    g.panel_n = panel_n > 0 ? panel_n : 1;
    g.panel_ld = panel_ld;
    g.prods = mode == 1 ? 1 : 3;
    if (splits > 1) {
        g.C = splitk_ws; g.ldc = N; g.sC = (long long)splits * M * N; g.sCsplit = (long long)M * N;
    } else {
        g.C = C; g.ldc = ldc; g.sC = sC; g.sCsplit = 0;
    }
    if (a_panels || c_panels) {
        const int pm = (a_panels ? 1 : 0) | (c_panels ? 2 : 0);
        if (b_kmajor) return pm == 1 ? gemm_panel_launch<true, 1>(g, x3, stream) : pm == 2 ? gemm_panel_launch<true, 2>(g, x3, stream)
                                                                                            : gemm_panel_launch<true, 3>(g, x3, stream);
        return pm == 1 ? gemm_panel_launch<false, 1>(g, x3, stream) : pm == 2 ? gemm_panel_launch<false, 2>(g, x3, stream)
                                                                              : gemm_panel_launch<false, 3>(g, x3, stream);
    }
    const int tn = N > 64 ? 2 : 1;
    int rc;
    static const bool fewrows_on = lpd_debug("fewrows", 1) != 0;
    static const bool rows32_on = fewrows_on;
    // (measured, the hidden projection on one stream: 1 row 32.8 -> 23.4 us, 4 rows 36.5 -> 30 us, 8 rows no gain; 32 rows 42.4 -> 61.4 us
    //  -- 128 partial sums per lane make the wide form FMA- and register-bound -- so the stream form serves up to FR_MAXM rows)
    constexpr int FR_MAXM = 4;
    if (fewrows_on && splits > 1 && !x3 && !a_kmajor && b_kmajor && batch == 1 && M <= FR_MAXM && N % 256 == 0 && K >= 8192 && ldb % 4 == 0 &&
        (((uintptr_t)B | (uintptr_t)splitk_ws) & 15) == 0 && (long long)((K + FR_KPER - 1) / FR_KPER) <= (long long)splits) {
        // weight-stream form (linear_fewrows_kernel): K / FR_KPER slabs of the caller's `splits` are used
        const int nsl = (K + FR_KPER - 1) / FR_KPER;
        const dim3 grid(nsl, N / 256);
#define LPD_FEWROWS(MT_)                                                                                                          \
        do {                                                                                                                      \
            const size_t lds = (size_t)(FR_KPER * MT_ + 2 * MT_ * 256) * sizeof(float);                                           \
            auto kern = linear_fewrows_kernel<MT_>;                                                                              \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, A, lda, B, ldb, splitk_ws, M, N, K);                           \
        } while (0)
        if (M <= 1) LPD_FEWROWS(1);
        else if (M <= 2) LPD_FEWROWS(2);
        else LPD_FEWROWS(4);
#undef LPD_FEWROWS
        LPD_CHECK_LAUNCH("lpd_gemm(few rows, weight stream)");
        splits = nsl;
        g.sCsplit = (long long)M * N;
        rc = LPD_OK;
    } else
    if (rows32_on && splits > 1 && !x3 && !a_kmajor && b_kmajor && batch == 1 && M > FR_MAXM && M <= 32 && N % 256 == 0 && K >= 8192 && ldb % 4 == 0 &&
        (((uintptr_t)B | (uintptr_t)splitk_ws) & 15) == 0 && (long long)((K + R32_KPER - 1) / R32_KPER) <= (long long)splits) {
        // 5 .. 32 rows: the weight stream through the f32-input MFMA (linear_rows32_kernel), K / R32_KPER slabs
        const int nsl = (K + R32_KPER - 1) / R32_KPER;
        const size_t lds = (size_t)R32_LDS_FLOATS * sizeof(float);      // 64 KiB >= the 33-KiB X slice
        static_assert(R32_LDS_FLOATS >= R32_KPER * R32_XLD, "the exchange region covers the X slice");
        (void)hipFuncSetAttribute((const void*)linear_rows32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(linear_rows32_kernel, dim3(nsl, N / 256), dim3(512), lds, stream, A, lda, B, ldb, splitk_ws, M, N, K);
        LPD_CHECK_LAUNCH("lpd_gemm(rows32, weight stream)");
        splits = nsl;
        g.sCsplit = (long long)M * N;
        rc = LPD_OK;
    } else
    if (splits > 1 && !x3 && !a_kmajor && b_kmajor && batch == 1 && M <= 64 && g.K == SMALLM_KC) {
        const int rgs = (M + 15) / 16;
        const size_t lds = (size_t)rgs * 16 * SMALLM_KC * sizeof(float);
        const dim3 grid(splits, (N + 255) / 256);
#define LPD_SMALLM(RG_)                                                                                                     \
        hipLaunchKernelGGL(linear_smallm_kernel<RG_>, grid, dim3(256 * RG_), lds, stream, A, lda, B, ldb, splitk_ws, M, N, K)
        if (rgs == 1) LPD_SMALLM(1);
        else if (rgs == 2) LPD_SMALLM(2);
        else if (rgs == 3) LPD_SMALLM(3);
        else LPD_SMALLM(4);
#undef LPD_SMALLM
        LPD_CHECK_LAUNCH("lpd_gemm(small-M split-K)");
        rc = LPD_OK;
    } else {
#define LPD_GEMM_CASE(AK, BK_)                                                       \
    rc = (tn == 2) ? gemm_launch<AK, BK_, 2>(g, batch, x3, stream) : gemm_launch<AK, BK_, 1>(g, batch, x3, stream)
    if (a_kmajor && b_kmajor) LPD_GEMM_CASE(true, true);
    else if (a_kmajor) LPD_GEMM_CASE(true, false);
    else if (b_kmajor) LPD_GEMM_CASE(false, true);
    else LPD_GEMM_CASE(false, false);
#undef LPD_GEMM_CASE
    }
    if (rc != LPD_OK) return rc;
    if (splits > 1) {
        int nslab = splits;
        long long stride = (long long)M * N;
        if (batch == 1 && splits >= 64 && splits % 16 == 0) {
            // many slabs, few outputs (M * N / 256 workgroups): first sum groups of 16 slabs in place (grid z = group; a
            // thread overwrites the element of its group's first slab it has just read), then the group sums
            dim3 grid1((N + 255) / 256, M, splits / 16);
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, grid1, dim3(256), 0, stream, (const float*)splitk_ws, splitk_ws, M, N, N, 16,
                               stride, 16 * stride, 16 * stride, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0,
                               0.0f, 0);
            nslab = splits / 16;
            stride *= 16;
        }
        dim3 grid((N + 255) / 256, M, batch);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, grid, dim3(256), 0, stream, (const float*)splitk_ws, C, M, N,
                           ldc, nslab, stride, (long long)splits * M * N, sC, bias, scale, shift, act, slope, accumulate);
        LPD_CHECK_LAUNCH("lpd_gemm(splitk reduce)");
    }
    return LPD_OK;
}

extern "C" int lpd_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                        int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC,
                        int splits, float* splitk_ws, const float* bias, const float* scale, const float* shift,
                        int act, float slope, int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream)
{
    return gemm_entry(0, A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, batch, sA, sB, sC, splits, splitk_ws, bias,
                      scale, shift, act, slope, accumulate, a_cloud, c_cloud, panel_n, panel_ld, stream);
}

extern "C" int lpd_gemm_bf16x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                               int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC,
                               int splits, float* splitk_ws, const float* bias, const float* scale, const float* shift,
                               int act, float slope, int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream)
{
    return gemm_entry(3, A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, batch, sA, sB, sC, splits, splitk_ws, bias,
                      scale, shift, act, slope, accumulate, a_cloud, c_cloud, panel_n, panel_ld, stream);
}

extern "C" int lpd_gemm_bf16x1(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                               int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC,
                               int splits, float* splitk_ws, const float* bias, const float* scale, const float* shift,
                               int act, float slope, int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream)
{
    return gemm_entry(1, A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, batch, sA, sB, sC, splits, splitk_ws, bias,
                      scale, shift, act, slope, accumulate, a_cloud, c_cloud, panel_n, panel_ld, stream);
}

// bytes of the fragment buffer for an N x K weight: hi and lo arrays of ceil(N/32) * ceil(K/16) fragments of 1 KiB
extern "C" long long lpd_gemm_prep_b_bytes(int N, int K)
{
    if (N <= 0 || K <= 0) return 0;
    return 2ll * ((N + 31) / 32) * ((K + 15) / 16) * 1024;
}

extern "C" int lpd_gemm_prep_b(const float* B, int ldb, int b_kmajor, int N, int K, void* frags, void* stream)
{
    LPD_CHECK_ARG(B && frags && N > 0 && K > 0, "lpd_gemm_prep_b: bad arguments");
    LPD_CHECK_ARG(((uintptr_t)frags & 15) == 0, "lpd_gemm_prep_b: frags must be 16-byte aligned");
    const int KS = (K + 15) / 16, NT = (N + 31) / 32;
    const long long total = (long long)NT * KS * 64;
    __bf16* fhi = reinterpret_cast<__bf16*>(frags);
    __bf16* flo = fhi + total * 8;
    hipLaunchKernelGGL(gemm_prep_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, ldb, b_kmajor, N,
                       K, KS, fhi, flo, total);
    LPD_CHECK_LAUNCH("lpd_gemm_prep_b");
    return LPD_OK;
}

// `batch` matrices sB floats apart -> `batch` fragment sets, lpd_gemm_prep_b_bytes(N, K) bytes apart (lpd_gemm_x3t_rows, batched)
extern "C" int lpd_gemm_prep_b_batch(const float* B, int ldb, int b_kmajor, int N, int K, int batch, long long sB, void* frags, void* stream)
{
    LPD_CHECK_ARG(B && frags && N > 0 && K > 0 && batch >= 1 && batch <= 65535, "lpd_gemm_prep_b_batch: bad arguments");
    LPD_CHECK_ARG(((uintptr_t)frags & 15) == 0, "lpd_gemm_prep_b_batch: frags must be 16-byte aligned");
    const int KS = (K + 15) / 16, NT = (N + 31) / 32;
    const long long total = (long long)NT * KS * 64;
    __bf16* fhi = reinterpret_cast<__bf16*>(frags);
    __bf16* flo = fhi + total * 8;
    hipLaunchKernelGGL(gemm_prep_b_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream, B, ldb, b_kmajor,
                       N, K, KS, fhi, flo, total, sB);
    LPD_CHECK_LAUNCH("lpd_gemm_prep_b_batch");
    return LPD_OK;
}

template <int WN, int PANELS, int KC, bool TALL = false, int MODE = 0>
static void x3w_wide_launch_kc(const X3wArgs& g, int NT, hipStream_t stream)
{
    const size_t lds = (size_t)4 * 128 * (KC + 8) * sizeof(__bf16);
    dim3 grid(TALL ? (NT + 1) / 2 : (NT + 4 * WN - 1) / (4 * WN), (g.M + 127) / 128);
    if (g.K % KC) {
        auto kern = gemm_x3w_wide_kernel<WN, true, PANELS, KC, TALL, MODE>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, stream, g);
    } else {
        auto kern = gemm_x3w_wide_kernel<WN, false, PANELS, KC, TALL, MODE>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds, stream, g);
    }
}

template <int WN, int PANELS>
static void x3w_wide_launch(const X3wArgs& g, int NT, hipStream_t stream)
{
    if constexpr (WN == 1) {
        if (NT <= 2) { x3w_wide_launch_kc<1, PANELS, 32, true>(g, NT, stream); return; }   // N <= 64: the NetVLAD assignment
    }
    static const int kc64 = lpd_debug("x3w-kc", 0) == 64;   // experiment: 64-deep chunks
    if (kc64 && g.K % 64 == 0) { x3w_wide_launch_kc<WN, PANELS, 64>(g, NT, stream); return; }
    x3w_wide_launch_kc<WN, PANELS, 32>(g, NT, stream);
}

static int gemm_x3w_impl(const float* A, int lda, const void* frags, float* C, int ldc, int M, int N, int K, const float* bias,
                         const float* scale, const float* shift, int act, float slope, int accumulate,
                         long long a_cloud, long long c_cloud, int panel_n, int panel_ld, int impl, double* stat_sum, double* stat_sumsq,
                         double* stat_ws, void* stream_, const float* a_scale = nullptr, const float* a_shift = nullptr, float* a_out = nullptr,
                         int a_ld = 0, float a_ns = 1.0f, int batch_rows = 0, long long frag_bytes = 0, int a_flags = 0)
{
    hipStream_t stream = (hipStream_t)stream_;
    const bool a_panels = a_cloud != 0, c_panels = c_cloud != 0;
    LPD_CHECK_ARG(A && frags && C && M > 0 && N > 0 && K > 0, "lpd_gemm_x3w: bad arguments");
    LPD_CHECK_ARG(a_panels || (lda % 4 == 0), "lpd_gemm_x3w: lda %% 4 != 0");
    LPD_CHECK_ARG(((uintptr_t)A & 7) == 0 && ((uintptr_t)frags & 15) == 0, "lpd_gemm_x3w: A (8 bytes) and the fragments (16) must be aligned");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_gemm_x3w: scale and shift must be given together");
    LPD_CHECK_ARG(!(a_panels || c_panels) || (panel_n > 0 && panel_n % 128 == 0 && M % panel_n == 0 && panel_ld >= panel_n),
                  "lpd_gemm_x3w: cloud-panel operands need points per cloud %% 128 == 0");
    LPD_CHECK_ARG(!a_panels || K % 8 == 0, "lpd_gemm_x3w: cloud-panel A needs K %% 8 == 0");
    LPD_CHECK_ARG(!c_panels || N % 8 == 0, "lpd_gemm_x3w: cloud-panel C needs N %% 8 == 0");
    const int prods = (impl & 16) ? 1 : 3;   // impl | 16: plain bf16 operands (a_hi b_hi only)
    impl &= 15;
    LPD_CHECK_ARG(impl == 0 || impl == 2 || impl == 3, "lpd_gemm_x3w: impl=%d", impl);
    const int KS = (K + 15) / 16, NT = (N + 31) / 32;
    const __bf16* fhi = reinterpret_cast<const __bf16*>(frags);
    LpdStatWs sws = {nullptr};
    if (stat_sum) {
        LPD_CHECK_ARG(N <= LPD_STAT_CMAX, "lpd_gemm_x3w_stats: N=%d exceeds %d columns", N, LPD_STAT_CMAX);
        sws = lpd_stat_arg(stat_ws);
        LPD_CHECK_ARG(sws.rep, "lpd_gemm_x3w_stats: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    }
    X3wArgs g{A, fhi, fhi + (long long)NT * KS * 512, C, M, N, K, KS, lda, ldc, bias, scale, shift, act, slope, accumulate,
              a_cloud, c_cloud, panel_n, panel_ld, 0, prods, stat_sum ? sws.sum() : nullptr, stat_sum ? sws.sumsq() : nullptr,
              a_scale, a_shift, a_out, a_ld, a_ns, batch_rows, 0, a_flags & 1, (a_flags >> 1) & 1, (a_flags >> 2) & 1};
    LPD_CHECK_ARG(!(a_flags & 4) || (!c_panels && !accumulate && N % 32 == 0 && ldc % 2 == 0 && ((uintptr_t)C & 3) == 0),
                  "lpd_gemm_x3w: bf16 C needs row-major rows, N %% 32 == 0, no accumulate");
    if (a_flags & 1) {      // bf16 rows: row-major, whole chunks, 8-byte aligned pieces; the plain product takes them as the hi image
        LPD_CHECK_ARG(!a_panels && K % 32 == 0 && lda % 4 == 0 && ((uintptr_t)A & 7) == 0 && impl != 1, "lpd_gemm_x3w: bf16 A needs row-major rows, K %% 32 == 0");
        if ((!a_scale || (a_flags & 2)) && g.prods == 3) g.prods = 2;      // plain bf16 rows, or a transformed map rounded to bf16: no lo image
    }
    LPD_CHECK_ARG(!(a_flags & 2) || ((a_flags & 1) && a_scale && (!a_out || a_ld % 4 == 0)), "lpd_gemm_x3w_act: a bf16 map is built for bf16 rows A");
    if (batch_rows) {      // per-problem fragment sets, each lpd_gemm_prep_b_bytes(N, K) bytes (hi half, then lo half): the lo array of set 0
        LPD_CHECK_ARG(batch_rows % 128 == 0 && M % batch_rows == 0 && frag_bytes == lpd_gemm_prep_b_bytes(N, K),      // starts where a single set's would
                      "lpd_gemm_x3w_batched: batch_rows %% 128 == 0, M %% batch_rows == 0, frag_bytes = lpd_gemm_prep_b_bytes(N, K)");
        g.frag_stride = frag_bytes / 2;      // bf16 elements between consecutive sets (a set = hi array + lo array)
    }
    LPD_CHECK_ARG(!a_scale || (a_shift && !a_panels && N <= 128 && impl != 3 && (!a_out || (a_ld % 4 == 0 && ((uintptr_t)a_out & 7) == 0))
                               && (((uintptr_t)a_scale | (uintptr_t)a_shift) & 15) == 0),
                  "lpd_gemm_x3w_act: the operand transform needs a row-major A and ONE column block (N <= 128)");
    if (a_scale) impl = 2;
    {   // (it matters for a row-major A with a power-of-two row stride; applied to every layout so that the summation
        //  order -- and with it every bit of the result -- does not depend on the layout of A)
        static const int rot = lpd_debug("x3w-rotate", 1);
        g.rotate = rot;
    }
    // impl: 0 = by shape, 2 = 128 x 128 blocks, 3 = 128 x 256 blocks
    if (impl == 0) impl = N >= 256 ? 3 : 2;
    const int panels = (a_panels ? 1 : 0) | (c_panels ? 2 : 0);
    if (a_flags & 5) {      // bf16 rows in (whole chunks: K % 32 == 0, checked above) or bf16 values out: their own instantiations
        const int mode = (a_flags & 1) | ((a_flags & 4) ? 2 : 0) | (((a_flags & 1) && a_scale) ? 4 : 0);
        LPD_CHECK_ARG(K % 32 == 0 && panels == 0, "lpd_gemm_x3w: bf16 operands need K %% 32 == 0 and row-major operands");
        if (mode == 2 && impl == 3) x3w_wide_launch_kc<2, 0, 32, false, 2>(g, NT, stream);          // conv3 + statistics -> bf16 map
        else if (mode == 3 && impl == 3) x3w_wide_launch_kc<2, 0, 32, false, 3>(g, NT, stream);     // ... from bf16 point features
        else if (mode == 1 && impl == 3) x3w_wide_launch_kc<2, 0, 32, false, 1>(g, NT, stream);     // dX = dY W on the bf16 gradient
        else if (mode == 1 && impl == 2 && NT <= 2) x3w_wide_launch_kc<1, 0, 32, true, 1>(g, NT, stream);   // dA of the NetVLAD backward
        else if (mode == 1 && impl == 2) x3w_wide_launch_kc<1, 0, 32, false, 1>(g, NT, stream);
        else if (mode == 5 && impl == 2 && NT <= 2) x3w_wide_launch_kc<1, 0, 32, true, 5>(g, NT, stream);   // assignment + bn3 / activation
        else if (mode == 5 && impl == 2) x3w_wide_launch_kc<1, 0, 32, false, 5>(g, NT, stream);
        else LPD_CHECK_ARG(false, "lpd_gemm_x3w: bf16 operand combination %d with %d-wide blocks is not built", mode, impl == 3 ? 256 : 128);
        LPD_CHECK_LAUNCH("lpd_gemm_x3w(bf16)");
        if (stat_sum) return lpd_stat_finish(sws, stat_sum, stat_sumsq, N, stream);
        return LPD_OK;
    }
    if (impl == 2) {
        switch (panels) {
            case 0: x3w_wide_launch<1, 0>(g, NT, stream); break;
            case 1: x3w_wide_launch<1, 1>(g, NT, stream); break;
            case 2: x3w_wide_launch<1, 2>(g, NT, stream); break;
            default: x3w_wide_launch<1, 3>(g, NT, stream); break;
        }
    } else {
        switch (panels) {
            case 0: x3w_wide_launch<2, 0>(g, NT, stream); break;
            case 1: x3w_wide_launch<2, 1>(g, NT, stream); break;
            case 2: x3w_wide_launch<2, 2>(g, NT, stream); break;
            default: x3w_wide_launch<2, 3>(g, NT, stream); break;
        }
    }
    LPD_CHECK_LAUNCH("lpd_gemm_x3w");
    if (stat_sum) return lpd_stat_finish(sws, stat_sum, stat_sumsq, N, stream);
    return LPD_OK;
}

extern "C" int lpd_gemm_x3w(const float* A, int lda, const void* frags, float* C, int ldc, int M, int N, int K, const float* bias,
                            const float* scale, const float* shift, int act, float slope, int accumulate,
                            long long a_cloud, long long c_cloud, int panel_n, int panel_ld, int impl, void* stream_)
{
    return gemm_x3w_impl(A, lda, frags, C, ldc, M, N, K, bias, scale, shift, act, slope, accumulate, a_cloud, c_cloud, panel_n, panel_ld, impl,
                         nullptr, nullptr, nullptr, stream_);
}

// The bare product C = A W^T (+ bias) of a TRAIN-mode layer together with the statistics its BatchNorm needs: stat_sum[n] /
// stat_sumsq[n] (fp64, zeroed here) receive the column sums / sums of squares of C over the M rows from the kernel's epilogue
// (fp32 over a block's 128 rows, then fp64 atomics) -- what lpd_colstats computes with a second pass over C
// (util/lpdnet_model.py:262: conv3_lpd + bn3_lpd over all B*N points; :251 convDG2 over all B*N*k edges).
// lpd_gemm_x3w with per-problem weights: rows [b batch_rows, (b + 1) batch_rows) of A are multiplied by weight b, whose fragments are set b
// of lpd_gemm_prep_b_batch (frag_bytes = lpd_gemm_prep_b_bytes(N, K) apart).  Row-major A / C over all problems (M = batch x batch_rows).
// The NetVLAD backward's dA[b] = x[b] . dV[b] (util/PointNetVlad.py:64-67 transposed: [N, 1024] x [1024, 64] per cloud) ran on the generic
// batched kernel at 2.6 TB/s of its 738-MB operand.
// a_scale / a_shift (or null): the rows of A are act(a_scale[k] A[m][k] + a_shift[k]) (lpd_gemm_x3w_act's operand transform, nothing
// stored; bf16 rows: the transformed values rounded to bf16, i.e. the map a bf16-storing lpd_gemm_x3w_act would have written).
extern "C" int lpd_gemm_x3w_batched(const void* A, int lda, int a_bf16, const void* frags, long long frag_bytes, int batch_rows, float* C, int ldc,
                                    int M, int N, int K, const float* a_scale, const float* a_shift, int a_act, float a_slope, int impl,
                                    void* stream_)
{
    LPD_CHECK_ARG(batch_rows > 0 && frag_bytes > 0, "lpd_gemm_x3w_batched: batch_rows / frag_bytes");
    LPD_CHECK_ARG((a_scale == nullptr) == (a_shift == nullptr) && a_act >= 0 && a_act <= 2, "lpd_gemm_x3w_batched: operand transform");
    return gemm_x3w_impl(reinterpret_cast<const float*>(A), lda, frags, C, ldc, M, N, K, nullptr, nullptr, nullptr, 0, 0.0f, 0, 0, 0, 0, 0, impl, nullptr,
                         nullptr, nullptr, stream_, a_scale, a_shift, nullptr, 0, a_act == 0 ? 1.0f : (a_act == 1 ? 0.0f : a_slope), batch_rows,
                         frag_bytes, a_bf16 ? (a_scale ? 3 : 1) : 0);
}

// C = A16 W^T (+= C with accumulate) for bf16 rows A16 [M][lda] (bf16 elements): the rows are the hi image, two MFMA products against
// the split weight.  The bf16-storage training mode's dX = dY W on the bf16 gradient of the conv3 map.
extern "C" int lpd_gemm_x3w_bf16a(const void* A16, int lda, const void* frags, float* C, int ldc, int M, int N, int K, int accumulate, int impl,
                                  void* stream_)
{
    return gemm_x3w_impl(reinterpret_cast<const float*>(A16), lda, frags, C, ldc, M, N, K, nullptr, nullptr, nullptr, 0, 0.0f, accumulate, 0, 0, 0, 0,
                         impl, nullptr, nullptr, nullptr, stream_, nullptr, nullptr, nullptr, 0, 1.0f, 0, 0, 1);
}

// c_bf16: C receives bf16 values ([M][ldc] bf16 elements; N % 32 == 0) -- the statistics stay those of the fp32 accumulators
// c_bf16 & 2: A holds bf16 rows as well (lda in bf16 elements, K % 32 == 0; built with a bf16 C and N >= 256)
extern "C" int lpd_gemm_x3w_stats(const float* A, int lda, const void* frags, void* C, int ldc, int c_bf16, int M, int N, int K, const float* bias,
                                  double* stat_sum, double* stat_sumsq, int impl, double* stat_ws, void* stream_)
{
    LPD_CHECK_ARG(stat_sum && stat_sumsq, "lpd_gemm_x3w_stats: null statistics");
    return gemm_x3w_impl(A, lda, frags, reinterpret_cast<float*>(C), ldc, M, N, K, bias, nullptr, nullptr, 0, 0.0f, 0, 0, 0, 0, 0, impl, stat_sum,
                         stat_sumsq, stat_ws, stream_, nullptr, nullptr, nullptr, 0, 1.0f, 0, 0, ((c_bf16 & 1) ? 4 : 0) | ((c_bf16 & 2) ? 1 : 0));
}

// C = act_a(a_scale[k] A[m][k] + a_shift[k]) W^T (+ bias): the train-mode BatchNorm affine + activation of the layer in front applied in
// the operand loader, the transformed rows stored to a_out [M][a_ld] on the way (may be null).  Row-major A, N <= 128 (one column
// block: every A element is staged exactly once).  util/lpdnet_model.py:262 (bn3 + act) -> util/PointNetVlad.py:48 (assignment).
extern "C" int lpd_gemm_x3w_act(const void* A, int lda, const void* frags, float* C, int ldc, int M, int N, int K, const float* bias,
                                const float* a_scale, const float* a_shift, int a_act, float a_slope, void* a_out, int a_ld, int flags, int impl,
                                void* stream_)
{
    LPD_CHECK_ARG(a_scale && a_shift, "lpd_gemm_x3w_act: null scale / shift");
    LPD_CHECK_ARG(a_act >= 0 && a_act <= 2, "lpd_gemm_x3w_act: activation %d unsupported", a_act);
    return gemm_x3w_impl(reinterpret_cast<const float*>(A), lda, frags, C, ldc, M, N, K, bias, nullptr, nullptr, 0, 0.0f, 0, 0, 0, 0, 0, impl & 16, nullptr,
                         nullptr, nullptr, stream_, a_scale, a_shift, reinterpret_cast<float*>(a_out), a_ld,
                         a_act == 0 ? 1.0f : (a_act == 1 ? 0.0f : a_slope), 0, 0, flags & 3);
}
