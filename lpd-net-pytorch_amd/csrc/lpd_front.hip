// lpd_front.hip -- the layers in front of the feature-space kNN of LPD-Net, in one kernel:
//     F1 = act(BN1(conv1_lpd(xyz)))          3 -> 64       (util/lpdnet_model.py:231)
//     F0 = act(BN2(conv2_lpd(F1)))           64 -> 64      (util/lpdnet_model.py:232)
//     squared norms + packed operand image (+ bf16 operand image) of F0 for lpd_knn_pm
// As separate launches these were lpd_linear_smallk (32 us at 32 x 4096 points: one thread per output element), an exact f32
// MFMA lpd_gemm (30 us) and the kNN's prep pass (17 us): 79 us to move 1.5 MB in and 33 MB out three times over.  Here a
// workgroup takes 128 points: conv1 straight into the k-major LDS image of the MFMA's A operand, conv2 on
// v_mfma_f32_32x32x2_f32 (exact fp32: one rounding per fma, k ascending), the F0 tile to global AND back into LDS, where four
// lanes per point build the kNN operands exactly as knn_prep_pm4_kernel does (same summation order of the squared norm).
// Everything stays exact fp32: these features feed the kNN, whose indices must be those of the fp32 reference.
#include "lpd_common.h"

namespace {

typedef __bf16 fr_bf16x8 __attribute__((ext_vector_type(8)));

struct FrontArgs {
    const float* xyz;
    int ldx;
    const float* W1;   // [64][3]
    const float* s1;
    const float* b1;
    const float* W2;   // [64][64], [out][in]
    const float* s2;
    const float* b2;
    float ns;          // negative-side slope of the activation
    float* F0;         // [M][64]
    float* xx;         // [M] or null: no kNN operands
    float* xp;         // [M][2][32]
    __bf16* xb;        // fragment-major bf16 image or null
    float* tiles;      // tile statistics of the best-first kNN ([B*nt][64] centroids, then |c|^2, radius, max |x|^2: [B*nt] each) or null
    long long M;
    int N, nt;
};

constexpr int FR_PTS = 128;
constexpr int FR_LDA = 136;                 // k-major A image [64][136] = 34816 B; later the F0 tile [128][68]
constexpr int FR_LDB = 68;                  // k-major W2 image [64][68]
constexpr int FR_LDF = 68;

__global__ __launch_bounds__(256) void lpdnet_front_kernel(FrontArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[64 * FR_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[64 * FR_LDB];
    __shared__ float xs[FR_PTS * 3];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const long long m0 = (long long)blockIdx.x * FR_PTS;

    // ---- W2 -> k-major LDS image; the block's coordinates ----
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int f = e * 256 + tid;             // float4 f: row n = f / 16, k quad f % 16
        const int n = f >> 4, kq = f & 15;
        const float4 w = *reinterpret_cast<const float4*>(g.W2 + n * 64 + kq * 4);
        Bs[(kq * 4 + 0) * FR_LDB + n] = w.x;
        Bs[(kq * 4 + 1) * FR_LDB + n] = w.y;
        Bs[(kq * 4 + 2) * FR_LDB + n] = w.z;
        Bs[(kq * 4 + 3) * FR_LDB + n] = w.w;
    }
    for (int i = tid; i < FR_PTS * 3; i += 256) {
        const int p = i / 3, c = i - p * 3;
        xs[i] = g.xyz[(m0 + p) * g.ldx + c];
    }
    __syncthreads();

    // ---- conv1 + BN + activation -> As[k = channel][point]: lanes along the points, the channel uniform per wave ----
    {
        const int p = tid & 127;
        const float x0 = xs[p * 3], x1 = xs[p * 3 + 1], x2 = xs[p * 3 + 2];
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int c = (tid >> 7) + 2 * i;
            float v = fmaf(x0, g.W1[c * 3], 0.0f);
            v = fmaf(x1, g.W1[c * 3 + 1], v);
            v = fmaf(x2, g.W1[c * 3 + 2], v);
            v = v * g.s1[c] + g.b1[c];
            As[c * FR_LDA + p] = lpd_act_pl(v, g.ns);
        }
    }
    __syncthreads();

    // ---- conv2 on the f32-input MFMA: wave w = rows 32 w .. + 31, both 32-column tiles ----
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    {
        const float* as = As + wave * 32 + col;
        const float* bs = Bs + col;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a = as[(2 * s + h) * FR_LDA];
            const float b0 = bs[(2 * s + h) * FR_LDB], b1 = bs[(2 * s + h) * FR_LDB + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
    }
    __syncthreads();          // every wave is done with the A image: its memory becomes the F0 tile

    // ---- BN + activation, F0 rows to global and into LDS ----
    float* Fs = As;           // [128][FR_LDF]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = 32 * j + col;
        const float sc = g.s2[n], sh = g.b2[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[j][r] * sc + sh;
            v = lpd_act_pl(v, g.ns);
            if (!g.xx) g.F0[(m0 + m) * 64 + n] = v;      // (with kNN operands the rows leave from the LDS tile below: 16-byte stores)
            Fs[m * FR_LDF + n] = v;
        }
    }
    if (!g.xx) return;
    __syncthreads();

    // ---- kNN operands, four lanes per point (knn_prep_pm4_kernel on the LDS tile) ----
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 64 + (tid >> 2);
        const int j = tid & 3;
        const long long m = m0 + p;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(Fs + p * FR_LDF + 16 * j + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            // the F0 row: four lanes x 64 bytes (round 5; from the accumulators it was 32 four-byte stores per lane: store-issue-bound)
            *reinterpret_cast<float4*>(g.F0 + m * 64 + 16 * j + 4 * q) = t;
        }
        // (Round 6: kept scalar.  The compiler packed this chain into v_pk_mul_f32 / v_pk_add_f32 with op_sel half-selects, one s_nop
        //  between dependent instructions; with an MFMA-heavy kernel of ANOTHER stream co-resident on the SIMD the chain returned wrong
        //  sums for whole 4-point groups -- squared norms and tile maxima off, F0 and the operand image right -- in 37 of 40 launches
        //  (profiles/r06_concurrency_packed_f32.txt).  The empty asm keeps every step a plain v_mul_f32 / v_add_f32.)
        float sq = __fmul_rn(v[0], v[0]);
        asm volatile("" : "+v"(sq));
#pragma unroll
        for (int c = 1; c < 16; ++c) {
            float m = __fmul_rn(v[c], v[c]);
            asm volatile("" : "+v"(m));
            sq = __fadd_rn(sq, m);
            asm volatile("" : "+v"(sq));
        }
        const int lane0 = lane & ~3;
        float total = __shfl(sq, lane0, 64);
        total = __fadd_rn(total, __shfl(sq, lane0 + 1, 64));
        total = __fadd_rn(total, __shfl(sq, lane0 + 2, 64));
        total = __fadd_rn(total, __shfl(sq, lane0 + 3, 64));
        if (j == 0) g.xx[m] = total;
        float* dst = g.xp + m * 64 + 8 * j;
        const long long b = m / g.N;
        const int pt = (int)(m - b * g.N);
        __bf16* fb = g.xb ? g.xb + ((((size_t)b * g.nt + (pt >> 5)) * 5 + j) * 64 + (pt & 31)) * 8 : nullptr;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            *reinterpret_cast<float4*>(dst + hh * 32) = make_float4(v[hh], v[2 + hh], v[4 + hh], v[6 + hh]);
            *reinterpret_cast<float4*>(dst + hh * 32 + 4) = make_float4(v[8 + hh], v[10 + hh], v[12 + hh], v[14 + hh]);
            if (fb) {
                fr_bf16x8 e;
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = (__bf16)v[2 * i + hh];
                *reinterpret_cast<fr_bf16x8*>(fb + hh * 32 * 8) = e;
            }
        }
        if (fb && j < 2) {     // fifth k-step: (hi, lo) of -xx / 2 in half 0, zeros in half 1
            fr_bf16x8 e;
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = (__bf16)0.0f;
            if (j == 0) {
                const float w = -0.5f * total;
                const __bf16 hi = (__bf16)w;
                e[0] = hi;
                e[1] = (__bf16)(w - (float)hi);
            }
            __bf16* f4 = g.xb + ((((size_t)b * g.nt + (pt >> 5)) * 5 + 4) * 64 + (pt & 31) + 32 * j) * 8;
            *reinterpret_cast<fr_bf16x8*>(f4) = e;
        }
        if (j == 0) xs[p] = total;
    }
    if (!g.tiles) return;
    __syncthreads();

    // ---- tile statistics of the best-first kNN (knn7_tile_stats_kernel, same arithmetic): wave w = tile w of the block ----
    {
        const long long nbt = (g.M / g.N) * g.nt;                   // tiles in all clouds
        const long long t = (m0 >> 5) + wave;                       // N % 128 == 0: tile index over all clouds = point / 32
        const int pc = (lane & 31) * 2 + (lane >> 5);               // packed column `lane` is channel 2 s + h
        const float* tile = Fs + wave * 32 * FR_LDF;
        float sum = 0.f;
        for (int i = 0; i < 32; ++i) sum += tile[i * FR_LDF + pc];
        const float cm = sum / 32.0f;
        g.tiles[t * 64 + lane] = cm;
        float* cen = Bs + wave * 64;                                // (the W2 image is not needed any more: every wave passed the
        cen[lane] = cm;                                             //  barrier behind the MFMAs) broadcast reads instead of 64 shuffles
        float d2 = 0.f, nx = 0.f;
        const int pr = lane & 31;                                   // both half-waves walk the same 32 points (max is idempotent)
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            const float d = tile[pr * FR_LDF + (c & 31) * 2 + (c >> 5)] - cen[c];
            d2 = fmaf(d, d, d2);
        }
        nx = xs[wave * 32 + pr];
        const bool bad = __any(!(fabsf(nx) <= 3.0e38f));
        float cn = fmaf(cm, cm, 0.f);
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) {
            d2 = fmaxf(d2, __shfl_xor(d2, mm, 64));
            nx = fmaxf(nx, __shfl_xor(nx, mm, 64));
            cn += __shfl_xor(cn, mm, 64);
        }
        if (lane == 0) {
            float* cnorm = g.tiles + nbt * 64;
            cnorm[t] = cn;
            cnorm[nbt + t] = sqrtf(d2) * 1.0001f + 1e-30f;          // radius
            cnorm[2 * nbt + t] = bad ? INFINITY : nx;              // max |x|^2
        }
    }
}

}  // namespace

extern "C" int lpd_lpdnet_front(const float* xyz, int ldx, const float* W1, const float* s1, const float* b1, const float* W2,
                                const float* s2, const float* b2, int act, float slope, float* F0, int B, int N, int k, float* knn_ws,
                                void* stream_)
{
    LPD_CHECK_ARG(xyz && W1 && s1 && b1 && W2 && s2 && b2 && F0, "lpd_lpdnet_front: null pointer");
    LPD_CHECK_ARG(B > 0 && N > 0 && N % FR_PTS == 0 && ldx >= 3, "lpd_lpdnet_front: needs N %% 128 == 0 (B=%d N=%d ldx=%d)", B, N, ldx);
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_lpdnet_front: act=%d unsupported (none/ReLU/LeakyReLU)", act);
    LPD_CHECK_ARG(act != 2 || (slope >= 0.0f && slope <= 1.0f), "lpd_lpdnet_front: LeakyReLU slope %g outside [0, 1]", (double)slope);
    LPD_CHECK_ARG((((uintptr_t)W2 | (uintptr_t)F0) & 15) == 0, "lpd_lpdnet_front: W2 and F0 must be 16-byte aligned");
    FrontArgs g{xyz, ldx, W1, s1, b1, W2, s2, b2, act == 0 ? 1.0f : (act == 1 ? 0.0f : slope), F0, nullptr, nullptr, nullptr, nullptr,
                (long long)B * N, N, (N + 31) / 32};
    if (knn_ws) {
        void* xb = nullptr;
        const int rc = lpd_knn_pm_layout(B, 64, N, k, knn_ws, &g.xx, &g.xp, &xb, &g.tiles);
        if (rc != LPD_OK) return rc;
        g.xb = reinterpret_cast<__bf16*>(xb);
    }
    hipLaunchKernelGGL(lpdnet_front_kernel, dim3((unsigned)(g.M / FR_PTS)), dim3(256), 0, (hipStream_t)stream_, g);
    LPD_CHECK_LAUNCH("lpd_lpdnet_front");
    return LPD_OK;
}
