// lpd_knn.hip -- fused streaming kNN (distance + top-k), bit-exact with the reference CPU path.
//
// Replaces util/lpdnet_model.py:317-326 (`knn`): the reference materialises three [B,N,N] fp32
// tensors (inner, pairwise_distance twice) and calls topk; here the [N,N] matrix never exists.
//
// Arithmetic contract (what torch's CPU path does, restated in oracle/lpd_oracle.c):
//   dot_ij = fma chain over channels c = 0..C-1 starting from +0        (lpdnet_model.py:318)
//   xx_i   = non-fused squares, sequential inside 16-channel blocks,
//            block partials added sequentially                          (lpdnet_model.py:320)
//   pd_ij  = ((-xx_j) - (-2 * dot_ij)) - xx_i                           (lpdnet_model.py:322,324)
//   idx    = k largest pd per row i, descending                         (lpdnet_model.py:325)
//   ties: equal pd -> lower index first (torch leaves this unspecified).
//
// gfx950 mapping: the dot products run on the f32-input MFMA (v_mfma_f32_32x32x2_f32), whose
// result is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md section 3, "FP32-input
// MFMA"), so one 32x32 tile of pd costs C/2 MFMAs and zero VALU FMAs.  MFMA rows = candidates j,
// columns = queries i: a lane owns ONE query (column lane&31) and receives 16 candidates per tile
// in its accumulator registers, so top-k selection is lane-local (sorted list in VGPRs); lanes
// l and l+32 hold disjoint candidate subsets of the same query and are merged once at the end.
// Candidates stream through LDS in [channel][candidate] order (the reference's own [B,C,N] layout
// is already channel-major, so staging is a straight copy and every LDS read is conflict-free).
#include "lpd_common.h"
#include <math.h>

namespace {

constexpr int KNN_WAVES = 4;
constexpr int KNN_THREADS = KNN_WAVES * 64;
constexpr int KNN_QPB = KNN_WAVES * 32;  // queries per block

template <int CP>
struct KnnCfg {
    // candidates per LDS chunk: keep the double-buffered chunk at <= 64 KiB
    static constexpr int CHUNK = (CP <= 32) ? 128 : (CP <= 64 ? 64 : 32);
    static constexpr int ROWS = 2 * CP;                       // padded channel count
    static constexpr int FLOATS = ROWS * CHUNK;               // per buffer
    static constexpr int PER_THREAD = FLOATS / KNN_THREADS;   // floats staged per thread
};

// sum of squares in torch's CPU reduction order (oracle_sumsq in oracle/lpd_oracle.c)
__global__ void knn_sumsq_kernel(const float* __restrict__ x, float* __restrict__ xx, int C, int N)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (n >= N) return;
    const float* p = x + (size_t)b * C * N + n;
    float total = 0.0f;
    for (int c0 = 0; c0 < C; c0 += 16) {
        int c1 = c0 + 16 < C ? c0 + 16 : C;
        float v0 = p[(size_t)c0 * N];
        float acc = __fmul_rn(v0, v0);
        for (int c = c0 + 1; c < c1; ++c) {
            float v = p[(size_t)c * N];
            acc = __fadd_rn(acc, __fmul_rn(v, v));
        }
        total = (c0 == 0) ? acc : __fadd_rn(total, acc);
    }
    xx[(size_t)b * N + n] = total;
}

// Insert (pd, j) into a list sorted by (value descending, index ascending).
// Precondition: j is larger than every index already in the list (each lane scans j ascending),
// so the candidate ranks after every element with value >= pd.
template <int KMAX>
__device__ __forceinline__ void knn_insert(float (&v)[KMAX], int (&id)[KMAX], float pd, int j)
{
#pragma unroll
    for (int s = KMAX - 1; s >= 1; --s) {
        bool ge_cur = v[s] >= pd;
        bool ge_prev = v[s - 1] >= pd;
        float nv = ge_prev ? pd : v[s - 1];
        int ni = ge_prev ? j : id[s - 1];
        v[s] = ge_cur ? v[s] : nv;
        id[s] = ge_cur ? id[s] : ni;
    }
    bool ge0 = v[0] >= pd;
    v[0] = ge0 ? v[0] : pd;
    id[0] = ge0 ? id[0] : j;
}

// General insert (no ordering assumption on j): an element ranks before the candidate when its
// value is larger, or equal with a smaller index.
template <int KMAX>
__device__ __forceinline__ void knn_insert_any(float (&v)[KMAX], int (&id)[KMAX], float pd, int j)
{
#pragma unroll
    for (int s = KMAX - 1; s >= 1; --s) {
        bool bf_cur = (v[s] > pd) || (v[s] == pd && id[s] < j);
        bool bf_prev = (v[s - 1] > pd) || (v[s - 1] == pd && id[s - 1] < j);
        float nv = bf_prev ? pd : v[s - 1];
        int ni = bf_prev ? j : id[s - 1];
        v[s] = bf_cur ? v[s] : nv;
        id[s] = bf_cur ? id[s] : ni;
    }
    bool bf0 = (v[0] > pd) || (v[0] == pd && id[0] < j);
    v[0] = bf0 ? v[0] : pd;
    id[0] = bf0 ? id[0] : j;
}

template <int CP, int KMAX, bool USE_MFMA>
__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float* __restrict__ x,   // [B][C][N]
                                                           const float* __restrict__ xx,  // [B][N]
                                                           int32_t* __restrict__ idx,     // [B][N][k]
                                                           int C, int N, int k)
{
    using Cfg = KnnCfg<CP>;
    constexpr int CHUNK = Cfg::CHUNK;
    constexpr int ROWS = Cfg::ROWS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: xs[2][ROWS][CHUNK], xxs[2][CHUNK]; the merge phase reuses xs.
    float* xs = smem;
    float* xxs = smem + 2 * Cfg::FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;    // k-half of the MFMA operand layout
    const int col = lane & 31;  // query column inside the wave's tile
    const int b = blockIdx.y;
    const int q = blockIdx.x * KNN_QPB + wave * 32 + col;  // this lane's query
    const bool q_ok = q < N;
    const float* xb = x + (size_t)b * C * N;
    const float* xxb = xx + (size_t)b * N;

    // query operand: B[k = 2s+h][col] = x[2s+h][q]  (zero beyond C or N)
    float qreg[CP];
#pragma unroll
    for (int s = 0; s < CP; ++s) {
        int c = 2 * s + h;
        qreg[s] = (q_ok && c < C) ? xb[(size_t)c * N + q] : 0.0f;
    }
    const float xq = q_ok ? xxb[q] : 0.0f;

    float lv[KMAX];
    int li[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        lv[s] = -INFINITY;
        li[s] = 0x7fffffff;
    }

    const int nchunks = (N + CHUNK - 1) / CHUNK;

    // ---- chunk staging: global -> registers -> LDS (double-buffered) ----
    float stage[Cfg::PER_THREAD];
    float stage_xx = INFINITY;
    auto load_chunk = [&](int ch) {
        const int j0 = ch * CHUNK;
#pragma unroll
        for (int e = 0; e < Cfg::PER_THREAD; ++e) {
            int f = e * KNN_THREADS + tid;  // flat index in [ROWS][CHUNK]
            int c = f / CHUNK, jj = f % CHUNK;
            int j = j0 + jj;
            stage[e] = (c < C && j < N) ? xb[(size_t)c * N + j] : 0.0f;
        }
        if (tid < CHUNK) {
            int j = j0 + tid;
            stage_xx = j < N ? xxb[j] : INFINITY;  // +inf => pd = -inf for padded candidates
        }
    };
    auto store_chunk = [&](int buf) {
        float* dst = xs + buf * Cfg::FLOATS;
#pragma unroll
        for (int e = 0; e < Cfg::PER_THREAD; ++e) dst[e * KNN_THREADS + tid] = stage[e];
        if (tid < CHUNK) xxs[buf * CHUNK + tid] = stage_xx;
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk(ch + 1);  // in flight during this chunk's math

        const float* cx = xs + buf * Cfg::FLOATS;
        const float* cxx = xxs + buf * CHUNK;
        const int j0 = ch * CHUNK;
#pragma unroll 1
        for (int t = 0; t < CHUNK / 32; ++t) {
            if (j0 + t * 32 >= N) break;
            float d[16];
            if constexpr (USE_MFMA) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < CP; ++s) {
                    float a = cx[(2 * s + h) * CHUNK + t * 32 + col];  // A[row=col][k=2s+h]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = acc[r];
            } else {
                // VALU fallback with the same tile/lane mapping: each lane recomputes its 16 dots
                // as an explicit fmaf chain.  Needs the full query vector, so it re-reads it from
                // global memory (slow; kept as an on-device cross-check of the MFMA numerics).
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    float dot = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        float qv = q_ok ? xb[(size_t)c * N + q] : 0.0f;
                        dot = __fmaf_rn(cx[c * CHUNK + t * 32 + row], qv, dot);
                    }
                    d[r] = dot;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // MFMA C/D row of register r
                const int jj = t * 32 + row;
                const float xxj = cxx[jj];
                const float inner = __fmul_rn(-2.0f, d[r]);
                const float tt = __fsub_rn(-xxj, inner);
                const float pd = __fsub_rn(tt, xq);
                if (pd > lv[KMAX - 1]) knn_insert<KMAX>(lv, li, pd, j0 + jj);
            }
        }

        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- merge the two half-lists of each query (lanes l and l+32) ----
    // The upper half-wave publishes its lists through LDS (xs is free now); the lower half-wave
    // folds them into its own register list with the full (value, index) comparator.
    float* mv = smem + wave * (2 * 32 * KMAX);
    int* mi = reinterpret_cast<int*>(mv + 32 * KMAX);
    if (h == 1) {
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            mv[col * KMAX + s] = lv[s];
            mi[col * KMAX + s] = li[s];
        }
    }
    __syncthreads();
    if (h == 0) {
        for (int e = 0; e < KMAX; ++e) {
            const float pv = mv[col * KMAX + e];
            const int pj = mi[col * KMAX + e];
            // partner list is sorted: once an element cannot enter, none of the rest can
            const bool enters = (pv > lv[KMAX - 1]) || (pv == lv[KMAX - 1] && pj < li[KMAX - 1]);
            if (!enters) break;
            knn_insert_any<KMAX>(lv, li, pv, pj);
        }
        if (q_ok) {
            int32_t* out = idx + ((size_t)b * N + q) * k;
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
                if (s < k) out[s] = li[s];
        }
    }
}

template <int CP, int KMAX>
int knn_launch(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, int impl,
               hipStream_t stream)
{
    using Cfg = KnnCfg<CP>;
    size_t stage_bytes = (size_t)(2 * Cfg::FLOATS + 2 * Cfg::CHUNK) * sizeof(float);
    size_t merge_bytes = (size_t)KNN_WAVES * 2 * 32 * KMAX * sizeof(float);
    size_t lds = stage_bytes > merge_bytes ? stage_bytes : merge_bytes;
    dim3 grid((N + KNN_QPB - 1) / KNN_QPB, B);
    if (impl == 1) {
        // VALU cross-check path: only built for the k <= 20 list size
        if constexpr (KMAX == 20) {
            auto kern = knn_kernel<CP, KMAX, false>;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, grid, dim3(KNN_THREADS), lds, stream, x, xx, idx, C, N, k);
        } else {
            lpd_set_error("lpd_knn: impl=1 (VALU cross-check) supports k <= 20 only");
            return LPD_ERR_UNSUPPORTED;
        }
    } else {
        auto kern = knn_kernel<CP, KMAX, true>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(KNN_THREADS), lds, stream, x, xx, idx, C, N, k);
    }
    LPD_CHECK_LAUNCH("lpd_knn");
    return LPD_OK;
}

template <int CP>
int knn_dispatch_k(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, int impl,
                   hipStream_t stream)
{
    if (k <= 20) return knn_launch<CP, 20>(x, xx, idx, B, C, N, k, impl, stream);
    if (k <= 32) return knn_launch<CP, 32>(x, xx, idx, B, C, N, k, impl, stream);
    if (k <= 64) return knn_launch<CP, 64>(x, xx, idx, B, C, N, k, impl, stream);
    lpd_set_error("lpd_knn: k=%d > 64 unsupported", k);
    return LPD_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int lpd_knn(const float* x, int B, int C, int N, int k, int32_t* idx, float* xx_ws, int impl,
                       void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(x && idx && xx_ws, "lpd_knn: null pointer");
    LPD_CHECK_ARG(B > 0 && C > 0 && N > 0, "lpd_knn: bad dims B=%d C=%d N=%d", B, C, N);
    LPD_CHECK_ARG(k > 0 && k <= N, "lpd_knn: need 0 < k <= N (k=%d N=%d)", k, N);
    LPD_CHECK_ARG(B <= 65535, "lpd_knn: B=%d exceeds grid.y", B);
    hipLaunchKernelGGL(knn_sumsq_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, x, xx_ws, C, N);
    LPD_CHECK_LAUNCH("lpd_knn(sumsq)");
    if (C <= 4) return knn_dispatch_k<2>(x, xx_ws, idx, B, C, N, k, impl, stream);
    if (C <= 64) return knn_dispatch_k<32>(x, xx_ws, idx, B, C, N, k, impl, stream);
    if (C <= 256) return knn_dispatch_k<128>(x, xx_ws, idx, B, C, N, k, impl, stream);
    lpd_set_error("lpd_knn: C=%d > 256 unsupported", C);
    return LPD_ERR_UNSUPPORTED;
}
