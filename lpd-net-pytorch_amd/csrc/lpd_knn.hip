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

// The same insertion written so that every list register is updated IN PLACE.  In the streaming kernel the lists are
// loop-carried through the queue-drain loop; with the C++ form above the register allocator failed to coalesce the
// loop PHIs and every drain iteration carried 161 v_mov_b32 on top of the ~120 useful instructions (ISA dump,
// DESIGN.md 3.1).  Tied "+v" operands pin each list element to one register; per slot: one compare (reused as the next
// slot's "current" mask), a median for the value -- new v[s] = med3(v[s-1], v[s], x) because v[s-1] >= v[s] -- and two
// selects for the index.  The s_nop covers the VALU-writes-SGPR -> VALU-reads-it wait states the compiler would insert.
template <int KMAX>
__device__ __forceinline__ void knn_insert_inplace(float (&v)[KMAX], int (&id)[KMAX], float x, int j)
{
    unsigned long long cc, cp;
    int ti;
    asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(cc) : "v"(v[KMAX - 1]), "v"(x));   // current slot keeps its entry?
#pragma unroll
    for (int s = KMAX - 1; s >= 1; --s) {
        asm volatile(
            "v_cmp_ge_f32_e64 %[cp], %[vp], %[x]\n\t"            // previous slot ranks before x?
            "v_med3_f32 %[vs], %[vp], %[vs], %[x]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[ti], %[ip], %[j], %[cp]\n\t"    // incoming index: j if the previous slot stays, else its index
            "v_cndmask_b32_e64 %[is], %[ti], %[is], %[cc]"       // keep the own index while the own value stays
            : [vs] "+v"(v[s]), [is] "+v"(id[s]), [cp] "=&s"(cp), [ti] "=&v"(ti)
            : [vp] "v"(v[s - 1]), [ip] "v"(id[s - 1]), [x] "v"(x), [j] "v"(j), [cc] "s"(cc));
        cc = cp;
    }
    asm volatile(
        "v_cndmask_b32_e64 %[i0], %[j], %[i0], %[cc]\n\t"
        "v_max_f32 %[v0], %[v0], %[x]"
        : [v0] "+v"(v[0]), [i0] "+v"(id[0])
        : [x] "v"(x), [j] "v"(j), [cc] "s"(cc));
}

// values only (phase A: the admission threshold needs no indices)
template <int KMAX>
__device__ __forceinline__ void knn_insert_values(float (&v)[KMAX], float x)
{
#pragma unroll
    for (int s = KMAX - 1; s >= 1; --s) v[s] = __builtin_amdgcn_fmed3f(v[s - 1], v[s], x);
    v[0] = fmaxf(v[0], x);
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

// ---------------------------------------------------------------------------------------------
// Product path (impl 0): wave-independent streaming kNN.
//
// Lessons from the first two generations (profiles/r01a, r01b): (1) inserting into the sorted register list
// inside the scan executes the ~100-instruction insertion on almost every candidate step, because the
// probability that ANY of the 64 lanes accepts is ~1 even when each lane accepts 5 %; (2) sharing staged
// candidate chunks through LDS couples the waves of a block at a barrier every chunk, so one wave's list
// maintenance stalls all others.  This kernel therefore
//   * gives every wave its own 32 queries and lets it stream the candidates straight from L2 into MFMA operand
//     registers (channel-major input => each operand fetch is two coalesced 128-B segments); no block barrier
//     in the scan; blocks are remapped so that one XCD walks whole clouds and their features stay in its L2;
//   * phase A: exact top-k of the ~96 candidates around the query's own tile (clouds are Z-ordered by
//     lpd_morton.hip, so these are spatial neighbours) gives a per-query admission threshold T0 that at least
//     k candidates are known to reach;
//   * phase B: one ascending scan of all candidates; the fast path per candidate is 3 VALU for pd, a compare
//     against max(T0, current k-th best) and an exec-masked 8-byte append to the lane's LDS queue; queues are
//     drained into the register lists only when a lane could overflow on the next tile (wave vote).
// Results are identical to the reference arithmetic (bit-exact pd, ties -> lower index first).
// ---------------------------------------------------------------------------------------------
constexpr int KNN3_WAVES = 4;
constexpr int KNN3_THREADS = KNN3_WAVES * 64;
constexpr int KNN3_QCAP = 40;   // queue slots per lane; a tile can add 16
// floats of LDS per wave: the queue (QCAP float2 per lane), reused at the end to merge the two half-lists (2 * 64 * KMAX)
template <int KMAX>
constexpr int knn3_wave_floats() { return KNN3_QCAP * 128 > 2 * 64 * KMAX ? KNN3_QCAP * 128 : 2 * 64 * KMAX; }

// ---- packed operand layout -------------------------------------------------------------------------------
// Ablations on MI355X (round 1: compile-time variants behind impl 10..41, removed in round 5; HISTORY.md 3.1) showed the scan is INSTRUCTION-ISSUE bound, not MFMA- or
// memory-bound: with channel-major operands every MFMA needed its own dword load plus ~6 scalar address
// instructions (each channel row is N floats away), and the per-candidate exec-masked appends cost ~100 executed
// instructions per tile even when nothing is admitted -- ~650 instructions per 32x32 tile against 32 MFMAs.
// A pre-pass therefore repacks the cloud as xp[b][n][h][s] = x[b][2s+h][n] (h = k-half of the MFMA operand
// layout): a lane's whole operand column is CP contiguous floats = CP/4 dwordx4 loads with immediate offsets.
template <int CP>
__global__ void knn_pack_kernel(const float* __restrict__ x, float* __restrict__ xp, int C, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const float* src = x + (size_t)b * C * N + n;
    float* dst = xp + ((size_t)b * N + n) * (2 * CP);
#pragma unroll 4
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < CP; ++s) {
            const int c = 2 * s + h;
            dst[h * CP + s] = c < C ? src[(size_t)c * N] : 0.0f;
        }
}

// Point-major input ([B*N][ld] rows, the layout the rest of the pipeline uses): squared norms (same summation order as
// knn_sumsq_kernel) and the packed operand image in one pass, one thread per point, no transposes.
template <int CP>
__global__ void knn_prep_pm_kernel(const float* __restrict__ xpm, int ld, float* __restrict__ xx, float* __restrict__ xp, int C,
                                   long long M)
{
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* row = xpm + m * ld;
    float v[2 * CP];
    if constexpr (CP == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = c < C ? row[c] : 0.0f;
    } else {
        const bool vec = (ld & 3) == 0 && (C & 3) == 0;
#pragma unroll
        for (int g = 0; g < 2 * CP / 4; ++g) {
            if (vec && 4 * g < C) {
                const float4 t = *reinterpret_cast<const float4*>(row + 4 * g);
                v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = (4 * g + e < C) ? row[4 * g + e] : 0.0f;
            }
        }
    }
    float total = 0.0f;
#pragma unroll
    for (int c0 = 0; c0 < 2 * CP; c0 += 16) {
        if (c0 < C) {
            float acc = __fmul_rn(v[c0], v[c0]);
#pragma unroll
            for (int c = c0 + 1; c < c0 + 16 && c < 2 * CP; ++c)
                if (c < C) acc = __fadd_rn(acc, __fmul_rn(v[c], v[c]));
            total = (c0 == 0) ? acc : __fadd_rn(total, acc);
        }
    }
    xx[m] = total;
    float* dst = xp + m * (2 * CP);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s4 = 0; s4 < CP; s4 += (CP >= 4 ? 4 : CP)) {
            if constexpr (CP >= 4) {
                *reinterpret_cast<float4*>(dst + h * CP + s4) =
                    make_float4(v[2 * s4 + h], v[2 * (s4 + 1) + h], v[2 * (s4 + 2) + h], v[2 * (s4 + 3) + h]);
            } else {
                *reinterpret_cast<float2*>(dst + h * CP) = make_float2(v[h], v[2 + h]);
            }
        }
}

// The same for exactly 64 channels with FOUR lanes per point (lane j: channels 16 j .. 16 j + 15): a row is read as one
// contiguous 256 bytes per four lanes and the packed image written in 128-byte runs (one thread per point read 16 strided
// float4 and ran at 2 TB/s: 31 us at 32 x 4096 points); the partial sums of the four 16-channel blocks are combined in the
// order knn_sumsq_kernel uses, so xx is bit-identical.  When the best-first kernel will want the bf16 image of the operands
// (xb != nullptr: tabulated low-precision bounds, whole tiles only), the lane writes its k-step of it from the same registers:
// lane j's eight packed values of half h ARE fragment (tile, k-step j) of lane (point, h).
typedef __bf16 knn_prep_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void knn_prep_pm4_kernel(const float* __restrict__ xpm, int ld, float* __restrict__ xx, float* __restrict__ xp,
                                                           __bf16* __restrict__ xb, long long M, int N, int nt)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = t >> 2;
    const int j = (int)(t & 3);
    const bool live = m < M;
    const float* row = xpm + (live ? m : M - 1) * ld + 16 * j;
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 q = *reinterpret_cast<const float4*>(row + 4 * g);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
    float acc = __fmul_rn(v[0], v[0]);
#pragma unroll
    for (int c = 1; c < 16; ++c) acc = __fadd_rn(acc, __fmul_rn(v[c], v[c]));
    const int lane0 = (threadIdx.x & 63) & ~3;
    float total = __shfl(acc, lane0, 64);
    total = __fadd_rn(total, __shfl(acc, lane0 + 1, 64));
    total = __fadd_rn(total, __shfl(acc, lane0 + 2, 64));
    total = __fadd_rn(total, __shfl(acc, lane0 + 3, 64));
    if (!live) return;
    if (j == 0) xx[m] = total;
    float* dst = xp + m * 64 + 8 * j;
    const long long b = m / N;
    const int pt = (int)(m - b * N);
    __bf16* fb = xb ? xb + ((((size_t)b * nt + (pt >> 5)) * 5 + j) * 64 + (pt & 31)) * 8 : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<float4*>(dst + h * 32) = make_float4(v[h], v[2 + h], v[4 + h], v[6 + h]);
        *reinterpret_cast<float4*>(dst + h * 32 + 4) = make_float4(v[8 + h], v[10 + h], v[12 + h], v[14 + h]);
        if (fb) {
            knn_prep_bf16x8 e;
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = (__bf16)v[2 * i + h];
            *reinterpret_cast<knn_prep_bf16x8*>(fb + h * 32 * 8) = e;
        }
    }
    if (fb && j < 2) {     // fifth k-step: (hi, lo) of -xx / 2 in half 0, zeros in half 1
        knn_prep_bf16x8 e;
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (__bf16)0.0f;
        if (j == 0) {
            const float w = -0.5f * total;
            const __bf16 hi = (__bf16)w;
            e[0] = hi;
            e[1] = (__bf16)(w - (float)hi);
        }
        __bf16* f4 = xb + ((((size_t)b * nt + (pt >> 5)) * 5 + 4) * 64 + (pt & 31) + 32 * j) * 8;
        *reinterpret_cast<knn_prep_bf16x8*>(f4) = e;
    }
}

// operand column of point j (clamped to N-1: padded candidates are neutralised through xx = NaN)
template <int CP>
__device__ __forceinline__ void knn3_ld_ops(const float* __restrict__ xpb, int N, int j, int h, float (&a)[CP])
{
    const float* row = xpb + ((size_t)min(j, N - 1) * 2 + h) * CP;
    if constexpr (CP % 4 == 0) {
#pragma unroll
        for (int g = 0; g < CP / 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(row + 4 * g);
            a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
        }
    } else {
        const float2 v = *reinterpret_cast<const float2*>(row);
        a[0] = v.x; a[1] = v.y;
    }
}

// squared norms of the 16 candidate rows this lane receives from the MFMA (rows 8g+4h+{0..3}); NaN for padding
__device__ __forceinline__ void knn3_ld_xx(const float* __restrict__ xxb, int N, int j0, int h, bool vec_ok, float4 (&x4)[4])
{
    if (vec_ok && j0 + 32 <= N) {
#pragma unroll
        for (int g = 0; g < 4; ++g) x4[g] = *reinterpret_cast<const float4*>(xxb + j0 + 8 * g + 4 * h);
        return;
    }
    const float nanv = __builtin_nanf("");   // padded candidates: pd = NaN, never admitted
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r0 = j0 + 8 * g + 4 * h;
        x4[g].x = r0 + 0 < N ? xxb[r0 + 0] : nanv;
        x4[g].y = r0 + 1 < N ? xxb[r0 + 1] : nanv;
        x4[g].z = r0 + 2 < N ? xxb[r0 + 2] : nanv;
        x4[g].w = r0 + 3 < N ? xxb[r0 + 3] : nanv;
    }
}

// One tile: pd of 32 candidates x 32 queries on the MFMA; operand registers are refilled with the NEXT tile's
// values four at a time right after the MFMAs that consumed them have issued.
//   pd = ((-xx_j) - (-2 dot)) - xx_i ;  (-xx_j) - (-2 dot) == fma(2, dot, -xx_j) bit for bit (2*dot is exact).
// ALWAYS (clouds of whole 32-point tiles, N % 32 == 0): the refills are UNCONDITIONAL and the squared norms one vector load each -- no
// branch inside the tile.  With `if (have_next)` around the refills the compiler resolves the vmcnt state at every join by waiting
// for vmcnt(0) -- in the middle of the tile's MFMA sequence and again in front of the norm loads: two L2 round trips per visited
// tile, the ~3 k cycles of 'advance' per tile that no re-coding of the walk could remove.  Callers pass a valid
// tile (any) when there is no next one; its values are never used.
template <int CP, bool ALWAYS = false>
__device__ __forceinline__ void knn3_tile(float (&a)[CP], float4 (&x4)[4], const float (&qreg)[CP], float xq,
                                          const float* __restrict__ xpb, const float* __restrict__ xxb, int N, int j_next,
                                          bool have_next, int col, int h, bool vec_ok, float (&pd)[16])
{
    if constexpr (ALWAYS) have_next = true;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* row = xpb + ((size_t)min(j_next + col, N - 1) * 2 + h) * CP;
    if constexpr (CP % 4 == 0) {
#pragma unroll
        for (int g = 0; g < CP / 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + e], qreg[4 * g + e], acc, 0, 0, 0);
            if (have_next) {
                const float4 v = *reinterpret_cast<const float4*>(row + 4 * g);
                a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < CP; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qreg[s], acc, 0, 0, 0);
        if (have_next) {
            const float2 v = *reinterpret_cast<const float2*>(row);
            a[0] = v.x; a[1] = v.y;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float4 xv = x4[r >> 2];
        const float xxj = (r & 3) == 0 ? xv.x : (r & 3) == 1 ? xv.y : (r & 3) == 2 ? xv.z : xv.w;
        pd[r] = __fsub_rn(__fmaf_rn(2.0f, acc[r], -xxj), xq);
    }
    if constexpr (ALWAYS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) x4[g] = *reinterpret_cast<const float4*>(xxb + j_next + 8 * g + 4 * h);
    } else if (have_next) knn3_ld_xx(xxb, N, j_next, h, vec_ok, x4);
}

template <int CP, int KMAX>
__global__ __launch_bounds__(KNN3_THREADS, ((KMAX > 32 || (KMAX == 32 && CP == 32)) ? 1 : 2))   // long lists: one wave per SIMD
void knn3_kernel(const float* __restrict__ xp, const float* __restrict__ xx, int32_t* __restrict__ idx, int N, int k,
                 int blocks_per_cloud, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    // XCD-aware: each XCD takes a contiguous range of (cloud, query-block) items => whole clouds per L2
    const int vb = lpd_xcd_remap(blockIdx.x, gridDim.x);
    const int b = vb / blocks_per_cloud;
    const int qb = vb - b * blocks_per_cloud;
    const int q0 = qb * (KNN3_WAVES * 32) + wave * 32;   // first query of this wave
    const int q = q0 + col;
    const bool q_ok = q < N;
    const float* xpb = xp + (size_t)b * N * (2 * CP);
    const float* xxb = xx + (size_t)b * N;
    const bool vec_ok = (N & 3) == 0;
    float2* myq = reinterpret_cast<float2*>(smem + (size_t)wave * knn3_wave_floats<KMAX>()) + lane;   // slot s at myq[s*64]

    float qreg[CP];
    knn3_ld_ops<CP>(xpb, N, q, h, qreg);
    const float xq = xxb[min(q, N - 1)];

    float lv[KMAX];
    int li[KMAX];
    auto reset_lists = [&]() {
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            lv[s] = -INFINITY;
            li[s] = 0x7fffffff;
        }
    };
    reset_lists();

    const int ntiles = (N + 31) / 32;
    float a[CP];
    float4 x4[4];
    float pd[16];

    // ---- phase A: admission threshold from the tiles around the wave's own tile ----
    {
        const int t_own = q0 / 32;
        int t_lo = t_own - 1, t_hi = t_own + 2;                 // 3 tiles = 96 candidates
        while ((t_hi - t_lo) * 16 < KMAX) { --t_lo; ++t_hi; }   // each half-wave must see >= KMAX candidates
        if (t_lo < 0) { t_hi -= t_lo; t_lo = 0; }
        if (t_hi > ntiles) { t_lo -= t_hi - ntiles; t_hi = ntiles; }
        if (t_lo < 0) t_lo = 0;
        knn3_ld_ops<CP>(xpb, N, t_lo * 32 + col, h, a);
        knn3_ld_xx(xxb, N, t_lo * 32, h, vec_ok, x4);
        for (int t = t_lo; t < t_hi; ++t) {
            knn3_tile<CP>(a, x4, qreg, xq, xpb, xxb, N, (t + 1) * 32, t + 1 < t_hi, col, h, vec_ok, pd);
            // unconditional, branch-free insertion: a value that does not beat the K-th best (or NaN padding) is replaced
            // by -inf, which leaves the list unchanged.  No divergent region => the compiler keeps the lists in place
            // (inserting under `if` cost ~4x the useful instructions in v_mov copies at the control-flow join).
#pragma unroll
            for (int r = 0; r < 16; ++r) knn_insert_values<KMAX>(lv, pd[r] > lv[KMAX - 1] ? pd[r] : -INFINITY);   // NaN padding -> -inf
        }
    }
    // k-th best of my half; the larger of the two halves' values is reached by >= k candidates overall
    float t0 = k <= KMAX ? lv[KMAX - 1] : -INFINITY;
    {
        // lists hold KMAX >= k entries: use the k-th (index k-1) when k < KMAX would be tighter, but k is a runtime
        // value and the list is in registers; the KMAX-th is a valid (slightly looser) bound
        const float other = __shfl_xor(t0, 32, 64);
        t0 = fmaxf(t0, other);
        if (!(t0 > -INFINITY)) t0 = -INFINITY;   // fewer than KMAX local candidates (tiny N): admit everything
    }
    reset_lists();

    // ---- phase B: ascending scan with queued selection ----
    int cnt = 0;
    int stat_adm = 0, stat_it = 0, stat_pass = 0;   // diagnostics (dbg&64): admissions of this lane, drain iterations / passing tiles of the wave
    auto drain = [&]() {
        stat_adm += cnt;
        // uniform trip count = the fullest queue of the wave, computed ONCE: a counted loop keeps the lists in place
        // (with `__any(e < cnt)` as the loop condition every iteration carried 160 register copies)
        int nmax = cnt;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        for (int e = 0; e < nmax; ++e) {
            ++stat_it;
            const float2 ent = myq[(e < cnt ? e : 0) * 64];
            const float pv = (e < cnt && ent.x > lv[KMAX - 1]) ? ent.x : -INFINITY;   // -inf: no-op insert (branch-free)
            knn_insert_inplace<KMAX>(lv, li, pv, __float_as_int(ent.y));               // FIFO => j ascending
        }
        cnt = 0;
    };
    knn3_ld_ops<CP>(xpb, N, col, h, a);
    knn3_ld_xx(xxb, N, 0, h, vec_ok, x4);
    for (int t = 0; t < ntiles; ++t) {
        knn3_tile<CP>(a, x4, qreg, xq, xpb, xxb, N, (t + 1) * 32, (t + 1 < ntiles), col, h, vec_ok, pd);
        const float thr = fmaxf(t0, lv[KMAX - 1]);
        const bool list_full = lv[KMAX - 1] > -INFINITY;
        // tile-level reject: most tiles are far from all 32 (Z-ordered, hence clustered) queries of the wave; one max
        // over the 16 values and a wave vote replace 16 exec-masked append sequences (~100 executed instructions)
        float mx = fmaxf(fmaxf(fmaxf(pd[0], pd[1]), fmaxf(pd[2], pd[3])), fmaxf(fmaxf(pd[4], pd[5]), fmaxf(pd[6], pd[7])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(pd[8], pd[9]), fmaxf(pd[10], pd[11])), fmaxf(fmaxf(pd[12], pd[13]), fmaxf(pd[14], pd[15]))));
        if (!__any(list_full ? (mx > thr) : (mx >= thr))) continue;
        ++stat_pass;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // before the list is full every candidate reaching T0 is admitted; afterwards it must beat the k-th best
            const bool admit = list_full ? (pd[r] > thr) : (pd[r] >= thr);
            if (admit) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                myq[cnt * 64] = make_float2(pd[r], __int_as_float(t * 32 + row));
                ++cnt;
            }
        }
        if (__any(cnt > KNN3_QCAP - 16)) drain();
    }
    drain();

    // ---- merge the two half-lists: both halves publish their sorted lists in this wave's (now free) queue region and
    // the lower half-wave walks the two lists with a 2-pointer merge (k steps of two LDS reads and a compare) ----
    float* mv = smem + (size_t)wave * knn3_wave_floats<KMAX>();   // wave region >= 2*64*KMAX floats
    int* mi = reinterpret_cast<int*>(mv + 64 * KMAX);
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        mv[lane * KMAX + s] = lv[s];
        mi[lane * KMAX + s] = li[s];
    }
    __syncthreads();
    if ((dbg & 64) && q_ok) {   // diagnostics instead of indices: [admitted by this half, drain iterations, passing tiles, T0 bits]
        if (h == 0) { int32_t* o = idx + ((size_t)b * N + q) * k; o[0] = stat_adm; o[1] = stat_it; o[2] = stat_pass; o[3] = __float_as_int(t0); }
        else { int32_t* o = idx + ((size_t)b * N + q) * k; o[4] = stat_adm; }
    } else
    if (h == 0 && q_ok) {
        const float* av = mv + lane * KMAX;
        const int* ai = mi + lane * KMAX;
        const float* bv = mv + (lane + 32) * KMAX;
        const int* bi = mi + (lane + 32) * KMAX;
        int pa = 0, pb = 0;
        int32_t* out = idx + ((size_t)b * N + q) * k;
        for (int s = 0; s < k; ++s) {
            bool take_a;
            if (pa >= KMAX) take_a = false;
            else if (pb >= KMAX) take_a = true;
            else {
                const float va = av[pa], vb2 = bv[pb];
                take_a = (va > vb2) || (va == vb2 && ai[pa] < bi[pb]);
            }
            if (take_a) out[s] = ai[pa++];
            else out[s] = bi[pb++];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Best-first kNN (product path for N <= 4096, k <= 20, C <= 64).
//
// The ascending scan above computes all N/32 distance tiles of every wave and rejects ~75 % of them afterwards; its
// lanes with a poor admission threshold (queries at Z-curve jumps) insert ~K ln(N/K) candidates and set the trip count
// of every queue drain.  Here each wave (32 queries = tile W of the Z-ordered cloud)
//   * knows, per candidate tile T (32 consecutive points: centroid c_T, radius r_T from a tiny pre-pass), an UPPER bound
//     of pd for each of its queries: ub[i][T] = -(max(0, |x_i - c_T| - r_T))^2 (+ fp32 slack), computed for all T by four
//     MFMA tiles against the 128 centroids and kept in LDS as bf16 (rounded up);
//   * visits the tiles outwards along the Z-curve (W, W+1, W-1, W+2, ...): the nearest tiles fill the lists at once, so
//     the k-th-best thresholds are tight from the second tile on (a pre-sorted centroid-distance order visited the same
//     ~40 of 128 tiles and cost a sort kernel);
//   * skips a tile, before loading or multiplying anything, when no query of the wave can gain from it
//     (ub[i][T] < threshold_i for all i): one LDS read, a compare and a wave vote.
// Exactness: a tile is skipped only if every pd in it is provably below the thresholds (the slack covers the rounding
// of the fp32 pd evaluation), so the admitted SET equals that of the full scan; exact value ties are ranked by index
// through the interval property of the walk (knn7_insert).
// ---------------------------------------------------------------------------------------------
constexpr bool KNN7_DEFAULT = true;    // impl 0 = best-first where it is built (N <= 4096, k <= 20, C <= 64); 4 forces the ascending kernel
constexpr int KNN7_QCAP = 24;      // queue slots per lane
constexpr int KNN7_MAXT = 128;     // candidate tiles per cloud (N <= 4096)
constexpr int KNN7_WAVE_LDS = KNN7_MAXT * 32 * 2 + KNN7_QCAP * 64 * 8;   // bf16 bound table + queue = 20 KiB per wave
// Three coordinates (CP = 2): the bound of a tile is recomputed from its centroid where it is tested (a dozen instructions)
// instead of being tabulated, and the queue holds 20 slots: 10 KiB per wave, so that LDS admits the four waves per SIMD
// the kernel's 120 VGPRs allow (the search is wait-bound: at 20 KiB it ran two).
constexpr int KNN7_QCAP_XYZ = 20;
constexpr int KNN7_QCAP_BIG = 48;
constexpr int KNN7_WAVE_LDS_XYZ = KNN7_QCAP_XYZ * 64 * 8;
// The same for 64 channels (a 64-term distance per tested tile, half per half-lane) measured WORSE than the table:
// 627 us against 566 at two waves per SIMD, 703 us at three (168 VGPRs, spills) -- off.
constexpr bool KNN7_ONFLY64 = false;
// Per-wave LDS of the best-first kernel.  ONFLY: the bound of a tile is recomputed from its centroid where the walk tests it
// instead of being tabulated -- always for three coordinates (a dozen instructions), and for 64 channels when the table
// ([tiles][32] bf16) would not fit: N > 4096.  KMAX > 20 (k up to 64: the stress configuration): 128 list registers per lane,
// one wave per SIMD, and the wave's region must hold the two half-lists for the final merge (64 * KMAX * 8 bytes).
template <int CP, int KMAX, bool ONFLY> struct Knn7Cfg {
    // 64-entry lists: the two half-lists are merged IN REGISTERS (knn_merge_halves_bitonic: no LDS region), so the wave's LDS is its
    // queue alone.  64 channels run one wave per SIMD (304 registers) and take a deep queue (fewer, better balanced drains: a drain
    // runs max-over-lanes iterations of a 64-slot insertion); three coordinates fit two waves per SIMD (240 registers) with 16 KiB each.
    static constexpr int QCAP = KMAX > 20 ? (CP == 2 ? 32 : KNN7_QCAP_BIG) : ((CP == 2) ? KNN7_QCAP_XYZ : KNN7_QCAP);
    static constexpr int QBYTES = QCAP * 64 * 8;
    static constexpr int TBYTES = ONFLY ? 0 : KNN7_MAXT * 32 * 2;
    static constexpr int MERGE = KMAX > 20 ? 0 : 64 * KMAX * 8;
    static constexpr int WAVE = (TBYTES + QBYTES) > MERGE ? (TBYTES + QBYTES) : MERGE;
    static constexpr int QOFF = TBYTES;      // byte offset of the queue inside the wave's region
    static constexpr int WAVES_PER_SIMD = KMAX > 20 ? (CP == 2 ? 2 : 1) : (CP == 2 ? 4 : 2);
};

// Merge of the two half-lists of every query (lanes l and l + 32 hold disjoint candidate subsets, each sorted by (value descending,
// index ascending)) without LDS: a bitonic merge on registers.  Stage 1 (half-cleaner against the partner's REVERSED list, read by
// lane shuffles) leaves the 64 best of the union as a bitonic sequence in both half-lanes; six compare-exchange stages on static
// register indices sort it.  Indices are distinct across the halves, so the order is total and the result is the reference's.
__device__ __forceinline__ bool knn_before(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

__device__ __forceinline__ void knn_merge_halves_bitonic(float (&v)[64], int (&id)[64])
{
#pragma unroll
    for (int s = 0; s < 32; ++s) {       // pairs (s, 63 - s): both partner values are read before either of mine is replaced
        const int r = 63 - s;
        const float pv_r = __shfl_xor(v[r], 32, 64), pv_s = __shfl_xor(v[s], 32, 64);
        const int pi_r = __shfl_xor(id[r], 32, 64), pi_s = __shfl_xor(id[s], 32, 64);
        const bool ks = knn_before(v[s], id[s], pv_r, pi_r);      // slot s: the better of mine[s] and partner[63 - s]
        const bool kr = knn_before(v[r], id[r], pv_s, pi_s);      // slot 63 - s: the better of mine[63 - s] and partner[s]
        v[s] = ks ? v[s] : pv_r; id[s] = ks ? id[s] : pi_r;
        v[r] = kr ? v[r] : pv_s; id[r] = kr ? id[r] : pi_s;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int s = 0; s < 64; ++s)
            if ((s & d) == 0) {
                const bool sw = knn_before(v[s + d], id[s + d], v[s], id[s]);
                const float tv = v[s]; const int ti = id[s];
                v[s] = sw ? v[s + d] : tv; id[s] = sw ? id[s + d] : ti;
                v[s + d] = sw ? tv : v[s + d]; id[s + d] = sw ? ti : id[s + d];
            }
}

// ---- tight tile bounds from a low-precision pass (64 channels, N <= 4096) -----------------------------------------------
// The centroid / radius bound lets ~47 of a cloud's 128 candidate tiles through for the average wave, but only ~17 hold a
// candidate that beats some query's k-th best (tools/knn_prefilter_study.py): in 64 dimensions a ball around the centroid
// is a loose description of 32 points.  And a visited tile is expensive far beyond its 32 f32 MFMAs: ~10 k cycles of a
// wave's life, of which ~2.7 k are the exposed latency of the operand prefetch (s_memtime per phase, tools/knn7_stats.py).
// The SAME distance tile on bf16 operands is five 32x32x16 MFMAs (160 matrix cycles against 2048) and bounds every exact
// value of the tile from above:
//     S(q, c) = sum_i bf16(q_i) bf16(c_i) + hi(-xx_c / 2) + lo(-xx_c / 2)     (fp32 accumulation; the last two terms ride in
//                                                                              a fifth k-step against a query operand of ones)
//     | S - (q.c - xx_c / 2) | <= 7.9e-3 |q| |c| + xx_c 2^-17                  (bf16 unit roundoff u = 2^-8: (1 + u)^2 - 1 = 7.83e-3
//                                                                              per product, Cauchy-Schwarz, + fp32 accumulation;
//                                                                              two-term split of xx_c / 2: u^2 / 2)
//     computed pd(q, c) <= 2 S - xx_q + 2 (7.9e-3 |q| |c|_max + xx_max 2^-17) + 2 E0        (E0: slack of the fp32 evaluation)
// (The first version used 2^-7.8: half the true bound -- u is 2^-8, not 2^-9.  Random clouds never noticed; a cloud of identical
//  points, where every product errs by the maximum in the same direction, did: tests/test_ops_gpu.py 'identical', C = 64.)
// knn7_bound_kernel evaluates the right-hand side for ALL (query, candidate tile) pairs of a cloud -- a dense N x N x 64 bf16
// product with a max over each tile's 32 candidates in the epilogue, no data-dependent control -- and writes the table the
// best-first kernel keeps in LDS ([query tile][candidate tile][32 queries], bf16 rounded up), in place of the centroid
// bounds: ~20 of the 47 tiles remain.  Non-finite inputs make the tile's xx_max (and with it the bound) +inf or NaN: the tile
// is visited.  Image xb: per (point, k-half) 32 channels in packed operand order + 8 extras (hi, lo, 0 ...; zero in the
// second half) = the lane's five 16-byte MFMA operands, stored fragment-major ([tile][k-step][lane][8]).
constexpr int KNN7_XB = 80;   // bf16 per point: 2 halves x 40
constexpr int KNN7_PRE_MAXT = 512;   // the low-precision pass runs for clouds of up to 512 tiles (N <= 16384: 16.8 MB of bounds per cloud)
typedef __bf16 knn_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void knn7_bf16_kernel(const float* __restrict__ xp, const float* __restrict__ xx, __bf16* __restrict__ xb,
                                                        int N, int nt)
{
    // one thread per (tile, lane) of cloud blockIdx.y; lane = (point of the tile, k-half).  Fragment-major image: the operand
    // of (tile T, k-step s) is 64 lanes x 16 bytes = ONE contiguous KiB (a lane-strided image -- 80 bytes per (point, half) --
    // made every operand load touch 40 cache lines for its 1 KiB and the bound kernel L1-bound: 113 us)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt * 64) return;
    const int b = blockIdx.y;
    const int T = t >> 6, lane = t & 63, h = lane >> 5;
    const int pt = min(T * 32 + (lane & 31), N - 1);                  // rows past the cloud repeat its last point (a valid candidate)
    const float* src = xp + (((size_t)b * N + pt) * 2 + h) * 32;
    __bf16* dst = xb + (((size_t)b * nt + T) * 5 * 64 + lane) * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(src + 8 * g), c = *reinterpret_cast<const float4*>(src + 8 * g + 4);
        knn_bf16x8 v;
        v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
        v[4] = (__bf16)c.x; v[5] = (__bf16)c.y; v[6] = (__bf16)c.z; v[7] = (__bf16)c.w;
        *reinterpret_cast<knn_bf16x8*>(dst + (size_t)g * 512) = v;
    }
    knn_bf16x8 e;
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = (__bf16)0.0f;
    if (h == 0) {
        const float v = -0.5f * xx[(size_t)b * N + pt];
        const __bf16 hi = (__bf16)v;
        e[0] = hi;
        e[1] = (__bf16)(v - (float)hi);
    }
    *reinterpret_cast<knn_bf16x8*>(dst + (size_t)4 * 512) = e;
}

// grid (ceil(nt / 8), B), 256 threads: wave w of block x bounds query tiles 8 x + 2 w, + 1 against every candidate tile; two
// waves per SIMD (one multiplies while the other finishes a tile: max over its candidates, bound, bf16 round-up, store).
// (Four query tiles per wave at one wave per SIMD, software-pipelined by hand, was slower: 133 us against 113.)
constexpr int KNN7_BQT = 2;
__global__ __launch_bounds__(256, 2) void knn7_bound_kernel(const __bf16* __restrict__ xb, const float* __restrict__ xx,
                                                            const float* __restrict__ txmax, uint16_t* __restrict__ ubq, int N, int nt, int C,
                                                            int tchunk)
{
    constexpr int QT = KNN7_BQT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, col = lane & 31;
    const int b = blockIdx.y;
    const int W0 = (blockIdx.x * 4 + wave) * QT;
    const __bf16* xbb = xb + ((size_t)b * nt * 5 * 64 + lane) * 8;
    const float* txb = txmax + (size_t)b * nt;
    float smax = 0.f;
    for (int t = lane; t < nt; t += 64) smax = fmaxf(smax, txb[t]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) smax = fmaxf(smax, __shfl_xor(smax, m, 64));
    const float E0 = 8.0f * (float)(C + 8) * 1.1920929e-7f * smax + 1e-30f;       // the best-first kernel's slack (same expression)

    knn_bf16x8 qop[QT][4], ones;
    float xq[QT], kq[QT];
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)((h == 0 && e < 2) ? 1.0f : 0.0f);
#pragma unroll
    for (int i = 0; i < QT; ++i) {
        const int Wq = min(W0 + i, nt - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qop[i][s] = *reinterpret_cast<const knn_bf16x8*>(xbb + ((size_t)Wq * 5 + s) * 512);
        xq[i] = xx[(size_t)b * N + min(Wq * 32 + col, N - 1)];
        kq[i] = (sqrtf(xq[i]) * 1.0001f + 1e-30f) * 7.9e-3f;                      // |q| (2 u + u^2 + accumulation), u = 2^-8
    }
    float pinf = INFINITY;
    asm volatile("" : "+v"(pinf));
    // max as v_med3(a, b, +inf) with the +inf in a register the optimiser cannot see through: fmaxf on MFMA results costs three
    // v_max each (two canonicalising self-maxes)
    auto mx2 = [&](float a_, float b_) { return __builtin_amdgcn_fmed3f(a_, b_, pinf); };
    // candidate operands: a ring of RING register sets, requested RING - 1 tiles ahead (one tile ahead -- a copy at the end of
    // the trip -- made every trip as long as an L2 round trip: 105 us, 1640 cycles per tile for 10 MFMAs)
    constexpr int RING = 4;
    knn_bf16x8 cop[RING][5];
    auto load_c = [&](int T, knn_bf16x8 (&c)[5]) {
        T = T < nt ? T : nt - 1;
#pragma unroll
        for (int s = 0; s < 5; ++s) c[s] = *reinterpret_cast<const knn_bf16x8*>(xbb + ((size_t)T * 5 + s) * 512);
    };
    __shared__ float stx[KNN7_PRE_MAXT];                 // xx_max of every tile (a global load per tile would be a dependent round trip)
    for (int t = tid; t < nt; t += 256) stx[t] = txb[t];
    __syncthreads();
    if (W0 >= nt) return;
    auto one_tile = [&](int T, const knn_bf16x8 (&c)[5]) {
        const float tx = stx[T];
        const float ncT = sqrtf(tx) * 1.0001f;
        const float cT = 2.0f * (tx * 7.62939453125e-6f + E0);
#pragma unroll
        for (int i = 0; i < QT; ++i) {
            f32x16 sa = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c[s], qop[i][s], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c[4], ones, sa, 0, 0, 0);
            float m = mx2(mx2(mx2(sa[0], sa[1]), mx2(sa[2], sa[3])), mx2(mx2(sa[4], sa[5]), mx2(sa[6], sa[7])));
            m = mx2(m, mx2(mx2(mx2(sa[8], sa[9]), mx2(sa[10], sa[11])), mx2(mx2(sa[12], sa[13]), mx2(sa[14], sa[15]))));
            m = mx2(m, __shfl_xor(m, 32, 64));                                    // all 32 candidates of the tile
            float ub = 2.0f * m - xq[i] + (2.0f * (kq[i] * ncT) + cT);
            ub += fabsf(ub) * 9.5367431640625e-7f;                                // the rounding of this expression itself
            uint32_t bits = __float_as_uint(ub);
            bits = (ub <= 0.0f) ? (bits >> 16) : 0x7f80u;                         // truncation rounds a negative value up; else (or NaN) +inf
            if (h == 0 && W0 + i < nt) ubq[(((size_t)b * nt + (W0 + i)) * nt + T) * 32 + col] = (uint16_t)bits;
        }
    };
    // blockIdx.z: this block's range of candidate tiles (tchunk, a multiple of RING).  One range for large batches; a small batch
    // (one cloud = 16 blocks of a 256-CU chip, each walking all 128 candidate tiles: 85 us of latency) is cut into up to 8 ranges
    const int tbeg = blockIdx.z * tchunk, tend = min(nt, tbeg + tchunk);
#pragma unroll
    for (int d = 0; d < RING - 1; ++d) load_c(tbeg + d, cop[d]);
    for (int T0 = tbeg; T0 < tend; T0 += RING) {
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            load_c(T0 + d + RING - 1, cop[(d + RING - 1) % RING]);      // the set tile T0 + d - 1 has just released
            if (T0 + d < tend) one_tile(T0 + d, cop[d]);
        }
    }
}

// tile statistics: centroid (packed operand layout), |c|^2, radius (inflated), max |x|^2.  One wave per tile.
template <int CP>
__global__ __launch_bounds__(64) void knn7_tile_stats_kernel(const float* __restrict__ xp, const float* __restrict__ xx,
                                                            float* __restrict__ cenp, float* __restrict__ cnorm,
                                                            float* __restrict__ rad, float* __restrict__ txmax, int N, int nt)
{
    constexpr int CH = 2 * CP;
    __shared__ float cen[CH];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cnt = min(32, N - t * 32);
    const float* base = xp + ((size_t)b * N + (size_t)t * 32) * CH;
    for (int c = tid; c < CH; c += 64) {
        float sum = 0.f;
        for (int i = 0; i < cnt; ++i) sum += base[(size_t)i * CH + c];
        const float m = sum / (float)cnt;
        cen[c] = m;
        cenp[((size_t)b * nt + t) * CH + c] = m;
    }
    __syncthreads();
    float d2 = 0.f, nx = 0.f, cn = 0.f;
    if (tid < cnt) {
        for (int c = 0; c < CH; ++c) {
            const float d = base[(size_t)tid * CH + c] - cen[c];
            d2 = fmaf(d, d, d2);
        }
        nx = xx[(size_t)b * N + t * 32 + tid];
    }
    for (int c = tid; c < CH; c += 64) cn = fmaf(cen[c], cen[c], cn);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        d2 = fmaxf(d2, __shfl_xor(d2, m, 64));
        nx = fmaxf(nx, __shfl_xor(nx, m, 64));
        cn += __shfl_xor(cn, m, 64);
    }
    const bool bad = __any(tid < cnt && !(fabsf(xx[(size_t)b * N + t * 32 + tid]) <= 3.0e38f));   // NaN / inf in the tile
    if (tid == 0) {
        rad[(size_t)b * nt + t] = sqrtf(d2) * 1.0001f + 1e-30f;   // upper bound of max |x_i - c| (the fmaf chain is good to ~1e-6)
        cnorm[(size_t)b * nt + t] = cn;
        txmax[(size_t)b * nt + t] = bad ? INFINITY : nx;          // (every bound that uses it becomes +inf: the tile is visited)
    }
}

// ---- longest-first launch order --------------------------------------------------------------------------------------
// The waves of the best-first kernel visit very different numbers of candidate tiles (C = 64: mean 51, p90 69, max 89;
// tools/knn_dist.py) and a launch is two rounds of waves over the chip's slots (B = 32: 4096 waves, 2048 slots): in
// index order a long wave that starts in the second round runs on alone (list-scheduling simulation on measured tile
// counts, tools/knn_predict.py: makespan 141 / 168 tile units at C = 3 / 64 against 98 / 120 for a perfect balance).
// A tile-level estimate predicts the count well enough (correlation 0.72..0.77) to launch the long waves first:
//   pred(W) = #{T : max(0, |c_W - c_T| - r_W - r_T) <= r_W / 2}        (centroids c, radii r of the 32-point tiles)
// and each XCD's contiguous range of work items (lpd_xcd_remap) is sorted by it, descending, in chunks of 1024.
template <int CP>
__global__ __launch_bounds__(128) void knn7_predict_kernel(const float* __restrict__ cenp, const float* __restrict__ rad,
                                                           int32_t* __restrict__ pred, int nt)
{
    constexpr int CH = 2 * CP;
    __shared__ float cw[CH];
    __shared__ int part[2];
    const int W = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* cb = cenp + (size_t)b * nt * CH;
    for (int c = tid; c < CH; c += 128) cw[c] = cb[(size_t)W * CH + c];
    __syncthreads();
    const float rw = rad[(size_t)b * nt + W];
    int n = 0;
    for (int T = tid; T < nt; T += 128) {
        const float* ct = cb + (size_t)T * CH;
        float d2 = 0.f;
        if constexpr (CH % 4 == 0) {
#pragma unroll 4
            for (int c = 0; c < CH; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(ct + c);
                const float d0 = v.x - cw[c], d1 = v.y - cw[c + 1], d2_ = v.z - cw[c + 2], d3 = v.w - cw[c + 3];
                d2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2_, d2_, fmaf(d3, d3, d2))));
            }
        } else {
            for (int c = 0; c < CH; ++c) { const float d = ct[c] - cw[c]; d2 = fmaf(d, d, d2); }
        }
        const float gap = fmaxf(sqrtf(d2) - rw - rad[(size_t)b * nt + T], 0.0f);
        n += gap <= 0.5f * rw ? 1 : 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m, 64);
    if ((tid & 63) == 0) part[tid >> 6] = n;
    __syncthreads();
    if (tid == 0) pred[(size_t)b * nt + W] = part[0] + part[1];
}

// The same estimate with the cloud's centroids staged in LDS (round 5): knn7_predict_kernel reads every candidate centroid row from
// global memory per lane -- 64 lanes on 64 different 256-byte rows per load -- and took 27 us at 32 clouds x 64 channels for 0.27 GFLOP.
// One block = 16 query tiles of one cloud; all nt centroids in LDS (row stride 2 CP + 1 floats: the 16 lanes of a query tile read 16
// different rows of one column conflict-free); a lane takes every 16th candidate tile, the 16 lanes of a query tile add up their
// counts.  Same formula; built for nt (2 CP + 1) 4 <= 64 KiB (N <= 8192 at 64 channels), the global-memory form serves the rest.
constexpr int KNN7_PRED_WPB = 16;
template <int CP>
__global__ __launch_bounds__(256) void knn7_predict_lds_kernel(const float* __restrict__ cenp, const float* __restrict__ rad,
                                                               int32_t* __restrict__ pred, int nt)
{
    constexpr int CH = 2 * CP, LD = CH + 1;
    extern __shared__ float pcen[];                 // [nt][LD], then rad [nt]
    float* prad = pcen + (size_t)nt * LD;
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* cb = cenp + (size_t)b * nt * CH;
    for (int i = tid; i < nt * CH; i += 256) pcen[(i / CH) * LD + (i % CH)] = cb[i];
    for (int i = tid; i < nt; i += 256) prad[i] = rad[(size_t)b * nt + i];
    __syncthreads();
    const int W = blockIdx.x * KNN7_PRED_WPB + (tid >> 4), l = tid & 15;
    const int Wc = min(W, nt - 1);
    const float* cw = pcen + (size_t)Wc * LD;
    const float rw = prad[Wc];
    int n = 0;
    for (int T = l; T < nt; T += 16) {
        const float* ct = pcen + (size_t)T * LD;
        float d2 = 0.f;
#pragma unroll 16
        for (int c = 0; c < CH; ++c) { const float d = ct[c] - cw[c]; d2 = fmaf(d, d, d2); }
        const float gap = fmaxf(sqrtf(d2) - rw - prad[T], 0.0f);
        n += gap <= 0.5f * rw ? 1 : 0;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) n += __shfl_xor(n, m, 64);
    if (l == 0 && W < nt) pred[(size_t)b * nt + W] = n;
}

// order[range of XCD x] = that range's item ids sorted by pred descending (ties: id ascending), chunk by chunk.
// grid (chunks, 8), 1024 threads: one key per thread; strides below 64 by lane shuffles, the rest through LDS.
__global__ __launch_bounds__(1024) void knn7_order_kernel(const int32_t* __restrict__ pred, int32_t* __restrict__ order, int nitems)
{
    // 32-bit sort words (round 5; 64-bit (pred, item) pairs before: two shuffles and a 64-bit compare per exchange, 17 us per launch):
    // (0xffff - pred) << 16 | position inside the chunk.  pred <= tiles per cloud <= 2048, position < 1024.
    __shared__ unsigned sk[1024];
    const int tid = threadIdx.x, xcd = blockIdx.y;
    const int q = nitems / 8, r = nitems % 8;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int len = q + (xcd < r ? 1 : 0);
    const int c0 = blockIdx.x * 1024;
    if (c0 >= len) return;
    const int cl = min(1024, len - c0);
    unsigned v = ~0u;                                          // padding sorts to the end
    if (tid < cl) v = ((0xffffu - (unsigned)min(pred[base + c0 + tid], 0xfffe)) << 16) | (unsigned)tid;
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned o;
            if (j < 64) o = __shfl_xor(v, j, 64);
            else {
                sk[tid] = v;
                __syncthreads();
                o = sk[tid ^ j];
                __syncthreads();
            }
            const bool lower = (tid & j) == 0, up = (tid & k) == 0;
            const unsigned mn = v < o ? v : o, mx = v < o ? o : v;
            v = (lower == up) ? mn : mx;
        }
    }
    if (tid < cl) order[base + c0 + tid] = base + c0 + (int)(v & 0xffffu);
}

// In-place insertion for the best-first kernel.  The tiles are visited W, W+1, W-1, W+2, ...: the set of visited tiles
// is always an interval around W, so a candidate from a tile above W has a larger index than everything in the list
// (ranks AFTER equal values: reference rule, lower index first) and one from a tile below W a smaller index than
// everything (ranks BEFORE equal values; rows of such a tile are queued in descending order).  Both are ONE compare per
// slot:  element ranks before the candidate  <=>  v >= y,  with y = x (after equals) or y = nextabove(x) (before
// equals).  x = y = -inf is the no-op.  (Exact value ties are common: pd is quantised at ulp(|x|^2), ~1 % of the
// queries see one among their admitted candidates.)
template <int KMAX>
__device__ __forceinline__ void knn7_insert(float (&v)[KMAX], int (&id)[KMAX], float x, float y, int j)
{
    // (Round 5, tried and dropped: leaving the unrolled chain at the first check -- every 2 / 4 / 8 slots -- at which no lane is still
    //  shifting.  A shift is over for a lane at the first element that ranks before its candidate, but some lane of the 64 nearly always
    //  inserts deep: 32 x 4096 x k20 405 / 198 us with the checks against 393 / 198 without, k = 64 lists 3.82 / 1.77 ms against 3.77 / 1.64.)
    unsigned long long cc, cp;
    int ti;
    asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(cc) : "v"(v[KMAX - 1]), "v"(y));
#pragma unroll
    for (int s = KMAX - 1; s >= 1; --s) {
        asm volatile(
            "v_cmp_ge_f32_e64 %[cp], %[vp], %[y]\n\t"
            "v_med3_f32 %[vs], %[vp], %[vs], %[x]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 %[ti], %[ip], %[j], %[cp]\n\t"
            "v_cndmask_b32_e64 %[is], %[ti], %[is], %[cc]"
            : [vs] "+v"(v[s]), [is] "+v"(id[s]), [cp] "=&s"(cp), [ti] "=&v"(ti)
            : [vp] "v"(v[s - 1]), [ip] "v"(id[s - 1]), [x] "v"(x), [y] "v"(y), [j] "v"(j), [cc] "s"(cc));
        cc = cp;
    }
    asm volatile(
        "v_cndmask_b32_e64 %[i0], %[j], %[i0], %[cc]\n\t"
        "v_max_f32 %[v0], %[v0], %[x]"
        : [v0] "+v"(v[0]), [i0] "+v"(id[0])
        : [x] "v"(x), [j] "v"(j), [cc] "s"(cc));
}

// WAVES: waves per workgroup.  A workgroup's LDS and wave slots stay taken until its slowest wave is done and the tile
// counts of neighbouring waves differ (C = 64: mean 51, p90 69, max 89 tiles): single-wave workgroups at C = 64
// (638 -> 607 us), four waves at C = 3 (shorter waves; the larger groups launch faster: 278 vs 287 us).
// SPLIT (round 5): SPLIT waves of one workgroup share ONE query tile W and divide its walk among them -- the visited tiles in walk
// order, dealt round-robin (each wave still meets the tiles above W in ascending and the tiles below W in descending order, which is
// all the one-compare insertion needs) -- each with its own pair of half-lists; the waves exchange thresholds through LDS (no barrier:
// a stale threshold is a looser one) and the 2 * SPLIT half-lists of a query are merged at the end.  A cloud is 128 query tiles: at one
// wave per tile a small batch leaves most of the chip idle and a search lasts as long as its longest wave's walk over ~50 candidate
// tiles (one cloud: 195 + 138 us of a 0.55 ms forward); with four waves per tile the walk is a quarter as long.  The bound table of the
// tile is ONE LDS copy per workgroup.  Bounds on the union's KMAX-th best that every wave may use: the best wave-level bound, and the
// minimum over all 2 * SPLIT half-lists of their ceil(KMAX / (2 SPLIT))-th best (then the lists together hold KMAX candidates >= it).
// Large batches keep SPLIT = 1: a split launch does ~1.6x the work of the single-wave one.  Not because every wave fills an empty list
// on its first tile -- letting wave 0 take the query tile alone first, the others starting from its bound, changed nothing (16 clouds
// 444 us against 421) -- but because a drain lasts as long as its fullest lane, and a wave that sees a quarter of the candidates has
// the same maximum for a quarter of the average.
template <int CP, int KMAX, int WAVES, bool ONFLY, bool FULLT = false, int SPLIT = 1>
__global__ __launch_bounds__(WAVES * SPLIT * 64, ((SPLIT > 1 && CP == 2) ? 3 : Knn7Cfg<CP, KMAX, ONFLY>::WAVES_PER_SIMD)) void knn7_kernel(const float* __restrict__ xp, const float* __restrict__ xx,
                                                             const float* __restrict__ cenp, const float* __restrict__ cnorm,
                                                             const float* __restrict__ rad, const float* __restrict__ txmax,
                                                             int32_t* __restrict__ idx, const int32_t* __restrict__ order,
                                                             int N, int k, int nt, int C, int blocks_per_cloud, int dbg,
                                                             const uint16_t* __restrict__ ubq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
    if constexpr (FULLT) dbg = 0;      // diagnostics (statistics, phase clocks) live in the branchy variant only: they cost ~17 registers
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too
    const int h = lane >> 5;
    const int col = lane & 31;
    int vb = lpd_xcd_remap(blockIdx.x, gridDim.x);
    if (order) vb = order[vb];                   // longest-first inside the XCD's range (single-wave workgroups)
    const int b = vb / blocks_per_cloud;
    const int qb = vb - b * blocks_per_cloud;
    static_assert(SPLIT == 1 || (WAVES == 1 && FULLT && KMAX <= 20 && !(ONFLY && CP == 32)), "SPLIT: whole tiles, short lists, bounds from the LDS table or from centroids");
    const int sw = SPLIT > 1 ? wave : 0;               // this wave's share of the walk
    const int q0 = SPLIT > 1 ? qb * 32 : qb * (WAVES * 32) + wave * 32;
    const int W = q0 >> 5;
    const bool wave_ok = q0 < N;                 // the last block of a cloud may hold waves without queries
    const int q = q0 + col;
    const bool q_ok = q < N;
    const float* xpb = xp + (size_t)b * N * (2 * CP);
    const float* xxb = xx + (size_t)b * N;
    const float* cenb = cenp + (size_t)b * nt * (2 * CP);
    const float* cnb = cnorm + (size_t)b * nt;
    const float* radb = rad + (size_t)b * nt;
    const bool vec_ok = (N & 3) == 0;
    const bool vec_ok_t = (nt & 3) == 0;
    using L = Knn7Cfg<CP, KMAX, ONFLY>;
    static_assert(CP != 2 || ONFLY, "three coordinates: bounds on the fly");
    // LDS: SPLIT == 1: per wave [table | queue]; SPLIT > 1: [table (shared)] [SPLIT queues] [thresholds: SPLIT x 2 x 64 floats]
    uint16_t* ubt = reinterpret_cast<uint16_t*>(smem7 + (SPLIT > 1 ? (size_t)0 : (size_t)wave * L::WAVE));   // [MAXT][32] (CP > 2)
    unsigned char* qreg_base = SPLIT > 1 ? smem7 + L::TBYTES + (size_t)wave * L::QBYTES : smem7 + (size_t)wave * L::WAVE + L::QOFF;
    float2* myq = reinterpret_cast<float2*>(qreg_base) + lane;                                            // slot s at myq[s*64]
    volatile float* thr_sh = reinterpret_cast<volatile float*>(smem7 + L::TBYTES + (size_t)SPLIT * L::QBYTES);    // SPLIT > 1 only
    static_assert(SPLIT == 1 || L::QBYTES >= 64 * KMAX * 8, "SPLIT: a wave's queue region holds its two half-lists for the merge");
    if constexpr (SPLIT > 1) {
        thr_sh[(sw * 2 + 0) * 64 + lane] = -INFINITY;
        thr_sh[(sw * 2 + 1) * 64 + lane] = -INFINITY;
    }

    float qreg[CP];
    knn3_ld_ops<CP>(xpb, N, q, h, qreg);
    const float xq = xxb[min(q, N - 1)];

    // slack of the fp32 pd evaluation: |pd_computed - pd_true| <= E0 (fma chain of C terms, squared norms, two final ops;
    // (|x| + |y|)^2 <= 4 max|x|^2)
    float smax = 0.f;
    for (int t = lane; t < nt; t += 64) smax = fmaxf(smax, txmax[(size_t)b * nt + t]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) smax = fmaxf(smax, __shfl_xor(smax, m, 64));
    const float E0 = 8.0f * (float)(C + 8) * 1.1920929e-7f * smax + 1e-30f;
    float4 qc3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (CP == 2) qc3 = *reinterpret_cast<const float4*>(xpb + (size_t)min(q, N - 1) * 4);

    float a[CP];
    float4 x4[4];
    float pd[16];

    // ---- bound table: from the low-precision pass (knn7_bound_kernel) when it ran ... ----
    if (wave_ok && !ONFLY && ubq) {
        const uint16_t* src = ubq + ((size_t)b * nt + W) * nt * 32;
        for (int e = (SPLIT > 1 ? tid : lane) * 8; e < nt * 32; e += SPLIT * 64 * 8)       // 16 bytes per lane and trip (nt * 32 % 8 == 0)
            *reinterpret_cast<uint4*>(ubt + e) = *reinterpret_cast<const uint4*>(src + e);
    }
    // ---- ... else pd of every query against every tile centroid (the centroids are a 'cloud' of nt points) ----
    if (wave_ok && !ONFLY && !ubq) {
        const int nct = (nt + 31) / 32;
        for (int ct = sw; ct < nct; ct += SPLIT) {      // (SPLIT > 1: the waves fill the shared table together)
            float4 r4[4];
            knn3_ld_ops<CP>(cenb, nt, ct * 32 + col, h, a);
            knn3_ld_xx(cnb, nt, ct * 32, h, vec_ok_t, x4);
            knn3_ld_xx(radb, nt, ct * 32, h, vec_ok_t, r4);
            knn3_tile<CP>(a, x4, qreg, xq, cenb, cnb, nt, 0, false, col, h, vec_ok_t, pd);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int T = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float4 rv = r4[r >> 2];
                const float rT = (r & 3) == 0 ? rv.x : (r & 3) == 1 ? rv.y : (r & 3) == 2 ? rv.z : rv.w;
                const float dq = sqrtf(fmaxf(-pd[r] - E0, 0.0f)) * 0.99999f;    // lower bound of |x_i - c_T|
                const float lb = fmaxf(dq - rT, 0.0f);
                const float ub = -(lb * lb) * 0.99999f + E0;                     // upper bound of every computed pd in tile T
                uint32_t bits = __float_as_uint(ub);
                bits = (ub <= 0.0f) ? (bits >> 16) : 0x7f80u;                    // truncation rounds a negative value up; else +inf
                if (T < nt) ubt[T * 32 + col] = (uint16_t)bits;
            }
        }
    }

    if constexpr (SPLIT > 1) __syncthreads();      // the shared bound table and the threshold slots are in place

    float lv[KMAX];
    int li[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        lv[s] = -INFINITY;
        li[s] = 0x7fffffff;
    }
    float thrv = q_ok ? -INFINITY : INFINITY;   // k-th best so far (admission + skip threshold); padded queries admit nothing
    int cnt = 0;
    // SPLIT > 1: thresholds of the other waves of the workgroup (see the kernel's comment): the union's KMAX-th best is at least the
    // best wave-level bound and at least the smallest M3-th best of the 2 * SPLIT half-lists
    constexpr int M3 = (KMAX + 2 * SPLIT - 1) / (2 * SPLIT);
    auto refresh = [&]() {
        if constexpr (SPLIT > 1) {
            float best = thr_sh[(sw * 2 + 0) * 64 + lane];      // this wave's own bound (published by its last drain)
            const float t3 = lv[M3 - 1];
            float m3 = fminf(t3, __shfl_xor(t3, 32, 64));
#pragma unroll
            for (int o = 1; o < SPLIT; ++o) {
                const int s2 = (sw + o) % SPLIT;
                best = fmaxf(best, thr_sh[(s2 * 2 + 0) * 64 + lane]);
                m3 = fminf(m3, fminf(thr_sh[(s2 * 2 + 1) * 64 + lane], thr_sh[(s2 * 2 + 1) * 64 + (lane ^ 32)]));
            }
            thrv = q_ok ? fmaxf(best, m3) : INFINITY;
        }
    };
    int stat_tiles = 0, stat_it = 0, stat_adm = 0, stat_drains = 0, stat_fn = 0;   // diagnostics (dbg): visited tiles, drain iterations, admitted, drains

    // visiting order: outwards along the Z-curve, W, W+1, W-1, W+2, ... (the visited tiles always form an interval around
    // W, which is what makes the one-compare insertion exact)
    int spos = 0;
    int vcount = 0;                        // SPLIT > 1: tiles of the cloud met so far in walk order (dealt round-robin to the waves)
    auto find_next_walk = [&]() -> int {   // next tile in order that some query of the wave can still gain from; -1 at the end
        while (spos < 2 * nt) {
            const int T = (spos & 1) ? W + ((spos + 1) >> 1) : W - (spos >> 1);
            ++spos;
            if (T < 0 || T >= nt) continue;
            if constexpr (SPLIT > 1) {
                const bool mine = (vcount % SPLIT) == sw;
                ++vcount;
                if (!mine) continue;
            }
            if constexpr ((CP == 32) && !ONFLY && !FULLT) ++stat_fn;
            float ub;
            if constexpr (CP == 2) {   // centroid in operand order (c0, c2 | c1, 0), like the query's own row
                const float4 cen = *reinterpret_cast<const float4*>(cenb + (size_t)T * 4);
                const float e0 = qc3.x - cen.x, e1 = qc3.y - cen.y, e2 = qc3.z - cen.z;
                const float d2 = fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
                const float dq = sqrtf(fmaxf(d2 - E0, 0.0f)) * 0.99999f;          // lower bound of |x_i - c_T|
                const float lb = fmaxf(dq - radb[T], 0.0f);
                ub = -(lb * lb) * 0.99999f + E0;                                  // upper bound of every computed pd in tile T
            } else if constexpr (ONFLY) {   // this half-lane's CP channels, the other half by shuffle
                if (ubq) {      // ... unless the low-precision pass tabulated the bounds (too large for LDS here: read where tested)
                    ub = __uint_as_float((uint32_t)ubq[(((size_t)b * nt + W) * nt + T) * 32 + col] << 16);
                    if (__any(ub >= thrv)) return T;
                    continue;
                }
                const float* ct = cenb + (size_t)T * (2 * CP) + h * CP;
                float part = 0.0f;
#pragma unroll
                for (int g4 = 0; g4 < CP / 4; ++g4) {
                    const float4 cv = *reinterpret_cast<const float4*>(ct + 4 * g4);
                    const float e0 = qreg[4 * g4] - cv.x, e1 = qreg[4 * g4 + 1] - cv.y, e2 = qreg[4 * g4 + 2] - cv.z, e3 = qreg[4 * g4 + 3] - cv.w;
                    part = fmaf(e3, e3, fmaf(e2, e2, fmaf(e1, e1, fmaf(e0, e0, part))));
                }
                const float d2 = part + __shfl_xor(part, 32, 64);
                const float dq = sqrtf(fmaxf(d2 - E0, 0.0f)) * 0.99999f;
                const float lb = fmaxf(dq - radb[T], 0.0f);
                ub = -(lb * lb) * 0.99999f + E0;
            } else ub = __uint_as_float((uint32_t)ubt[T * 32 + col] << 16);
            if (__any(ub >= thrv)) return T;
        }
        return -1;
    };
    // Large clouds with tabulated bounds (ONFLY && ubq): one dependent 2-byte global read per TESTED tile -- 512 of them per wave at
    // N = 16384, an L2 round trip each -- was the walk's cost.  The bounds of eight consecutive walk positions are now requested
    // together (eight independent loads, packed two per register) and tested as a bit mask against the current threshold: one round
    // trip per eight tiles.  The bounds do not depend on the threshold, so testing a cached bound later is the same test.
    uint32_t ubp[4] = {0, 0, 0, 0};
    uint32_t cvalid = 0;               // bit j: cached position j is a tile of the cloud (the first threshold is -inf: a sentinel bound would pass)
    int cbase = -8;                    // walk position of the first cached bound (a multiple of 8); spos in [cbase, cbase + 8) is cached
    auto walk_tile = [&](int pos) -> int { return (pos & 1) ? W + ((pos + 1) >> 1) : W - (pos >> 1); };
    auto find_next_tab = [&]() -> int {
        while (spos < 2 * nt) {
            if (spos >= cbase + 8) {
                cbase = spos & ~7;
                cvalid = 0;
                uint32_t raw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int T = walk_tile(cbase + j);
                    const bool ok = T >= 0 && T < nt;                                   // uniform
                    cvalid |= (ok ? 1u : 0u) << j;
                    raw[j] = (uint32_t)ubq[(((size_t)b * nt + W) * nt + (ok ? T : W)) * 32 + col];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ubp[j] = raw[2 * j] | (raw[2 * j + 1] << 16);
            }
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ub = __uint_as_float((j & 1) ? (ubp[j >> 1] & 0xffff0000u) : (ubp[j >> 1] << 16));
                m |= (__any(ub >= thrv) ? 1u : 0u) << j;
            }
            m &= cvalid & ~((1u << (spos - cbase)) - 1u);          // tiles of the cloud, not yet consumed
            if (m == 0) { spos = cbase + 8; continue; }
            const int j = __builtin_ctz(m);
            spos = cbase + j + 1;
            return walk_tile(cbase + j);
        }
        return -1;
    };
    // Round 6: the same eight-at-a-time test for the table in LDS (N <= 4096, one wave per tile).  find_next_walk reads ONE bound, votes
    // and branches per candidate tile -- a dependent LDS round trip for each of the cloud's 128 tiles: 65 k of a wave's 276 k cycles went
    // into the walk (tools/knn7_stats.py) for 23 visited tiles.  Eight independent ds_read_u16 per trip, the votes on cached bounds.
    auto find_next_lds8 = [&]() -> int {
        while (spos < 2 * nt) {
            if (spos >= cbase + 8) {
                cbase = spos & ~7;
                cvalid = 0;
                uint32_t raw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int T = walk_tile(cbase + j);
                    const bool ok = T >= 0 && T < nt;                                   // uniform
                    cvalid |= (ok ? 1u : 0u) << j;
                    raw[j] = (uint32_t)ubt[(ok ? T : W) * 32 + col];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ubp[j] = raw[2 * j] | (raw[2 * j + 1] << 16);
            }
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ub = __uint_as_float((j & 1) ? (ubp[j >> 1] & 0xffff0000u) : (ubp[j >> 1] << 16));
                m |= (__any(ub >= thrv) ? 1u : 0u) << j;
            }
            m &= cvalid & ~((1u << (spos - cbase)) - 1u);          // tiles of the cloud, not yet consumed
            if (m == 0) { spos = cbase + 8; continue; }
            const int j = __builtin_ctz(m);
            spos = cbase + j + 1;
            if constexpr ((CP == 32) && !ONFLY && !FULLT) stat_fn += j + 1;
            return walk_tile(cbase + j);
        }
        return -1;
    };
    // (The same batching for three coordinates -- centroids and radii of eight walk positions requested together, eight bounds cached
    //  -- made the xyz walk SLOWER, 177 -> 201 us at 32 clouds: its per-tile test is a scalar load and a dozen vector instructions, not a
    //  dependent LDS round trip, and the cached form re-votes eight bounds per call.)
    auto find_next = [&]() -> int {
        if constexpr (ONFLY && CP == 32) { if (ubq) return find_next_tab(); }
        if constexpr (!ONFLY && CP == 32 && SPLIT == 1) return find_next_lds8();
        return find_next_walk();
    };
    auto drain = [&]() {
        int nmax = cnt;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        if constexpr (!FULLT) { stat_it += nmax; stat_adm += cnt; ++stat_drains; }
        for (int e = 0; e < nmax; ++e) {
            const float2 ent = myq[(e < cnt ? e : 0) * 64];
            const float pv = (e < cnt && ent.x >= lv[KMAX - 1]) ? ent.x : -INFINITY;   // -inf: no-op insert (branch-free)
            const int j = __float_as_int(ent.y);
            // candidates from tiles below the wave's own rank before equal values: compare against nextabove(pv)
            const int pb = __float_as_int(pv);
            int nb = pb < 0 ? pb - 1 : pb + 1;
            nb = (pb & 0x7fffffff) == 0 ? 1 : nb;                           // +-0 -> smallest positive
            const float y = (j >= q0 || pv == -INFINITY) ? pv : __int_as_float(nb);
            knn7_insert<KMAX>(lv, li, pv, y, j);
        }
        cnt = 0;
        // Threshold of the QUERY, not of this half: the two half-lanes of a query each keep a top-K of their own 16 rows
        // per tile.  The query's K-th best is at least either half's K-th best, and at least the smaller of the two
        // ceil(K/2)-th bests (then both halves hold ceil(K/2) candidates >= it).  With neighbours split about evenly
        // between the halves the latter is close to the true K-th best, and it exists after the first tile.
        constexpr int HALF = (KMAX + 1) / 2;
        const float tK = lv[KMAX - 1], tH = lv[HALF - 1];
        const float pK = __shfl_xor(tK, 32, 64), pH = __shfl_xor(tH, 32, 64);
        const float thr = fmaxf(fmaxf(tK, pK), fminf(tH, pH));
        thrv = q_ok ? thr : INFINITY;
        if constexpr (SPLIT > 1) {      // publish this wave's bounds, then take the others' into account
            thr_sh[(sw * 2 + 0) * 64 + lane] = thr;
            thr_sh[(sw * 2 + 1) * 64 + lane] = lv[M3 - 1];
            refresh();
        }
    };

    long long tm_adv = 0, tm_tile = 0, tm_sel = 0, tm_drain = 0, tm0 = 0, tm_start = 0;   // dbg: cycles per phase (s_memtime)
    constexpr bool TIMERS = (CP == 32) && !ONFLY && !FULLT;     // (the phase clocks cost the xyz kernel registers it does not have: 128 at four waves per SIMD)
    auto tick = [&]() -> long long { return (TIMERS && dbg) ? (long long)__builtin_readcyclecounter() : 0; };
    tm_start = tick();
    // Operands two tiles ahead (TWO register sets, 64 channels with short lists only: 226 registers): a tile's operands are
    // requested while the tile BEFORE the previous one is multiplied.  One tile ahead, the copy of the loop-carried registers at
    // the back-edge waited ~2.7 k cycles per tile for the youngest loads (an L2 round trip under load is longer than the
    // selection that follows the MFMAs): a quarter of a wave's life (s_memtime per phase, tools/knn7_stats.py).
    constexpr bool TWO_AHEAD = (CP == 32) && KMAX <= 20 && !ONFLY;
    float a2[TWO_AHEAD ? CP : 1];
    float4 x42[4];
    int cur = wave_ok ? find_next() : -1;
    int nxt = cur >= 0 ? find_next() : -1;
    int nxt2 = (TWO_AHEAD && nxt >= 0) ? find_next() : -1;
    tm_adv += tick() - tm_start;
    if (cur >= 0) {
        knn3_ld_ops<CP>(xpb, N, cur * 32 + col, h, a);
        knn3_ld_xx(xxb, N, cur * 32, h, vec_ok, x4);
    }
    if constexpr (TWO_AHEAD) {
        if (nxt >= 0) {
            knn3_ld_ops<CP>(xpb, N, nxt * 32 + col, h, a2);
            knn3_ld_xx(xxb, N, nxt * 32, h, vec_ok, x42);
        }
    }
    // one tile: `cur` sits in (aa, xx4), which the product refills with the operands of the tile TWO_AHEAD ? nxt2 : nxt
    auto visit = [&](auto& aa, float4 (&xx4)[4]) {
        if constexpr (!FULLT) ++stat_tiles;
        tm0 = tick();
        const int pf = TWO_AHEAD ? nxt2 : nxt;
        knn3_tile<CP, FULLT>(aa, xx4, qreg, xq, xpb, xxb, N, (pf >= 0 ? pf : 0) * 32, pf >= 0, col, h, vec_ok, pd);
        float mx = fmaxf(fmaxf(fmaxf(pd[0], pd[1]), fmaxf(pd[2], pd[3])), fmaxf(fmaxf(pd[4], pd[5]), fmaxf(pd[6], pd[7])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(pd[8], pd[9]), fmaxf(pd[10], pd[11])), fmaxf(fmaxf(pd[12], pd[13]), fmaxf(pd[14], pd[15]))));
        if (TIMERS && dbg) { const long long t = tick(); tm_tile += t - tm0; tm0 = t; }
        refresh();      // (SPLIT > 1: what the other waves of the workgroup have found meanwhile)
        if (__any(mx >= thrv)) {
            if (cur >= W) {   // rows ascending: later arrivals have larger indices
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (pd[r] >= thrv) {   // ties with the k-th best are admitted: the list decides (NaN padding never passes)
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                        myq[cnt * 64] = make_float2(pd[r], __int_as_float(cur * 32 + row));
                        ++cnt;
                    }
                }
            } else {          // tile below the wave's own: rows descending, every arrival has the smallest index so far
#pragma unroll
                for (int r = 15; r >= 0; --r) {
                    if (pd[r] >= thrv) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                        myq[cnt * 64] = make_float2(pd[r], __int_as_float(cur * 32 + row));
                        ++cnt;
                    }
                }
            }
            if (TIMERS && dbg) { const long long t = tick(); tm_sel += t - tm0; tm0 = t; }
            if (__any(cnt > L::QCAP - 16)) drain();
            if (TIMERS && dbg) { const long long t = tick(); tm_drain += t - tm0; tm0 = t; }
        }
        cur = nxt;
        if constexpr (TWO_AHEAD) {
            nxt = nxt2;
            nxt2 = nxt >= 0 ? find_next() : -1;
        } else nxt = cur >= 0 ? find_next() : -1;
        if (TIMERS && dbg) { const long long t = tick(); tm_adv += t - tm0; tm0 = t; }
    };
    if constexpr (TWO_AHEAD) {
        while (cur >= 0) {
            visit(a, x4);
            if (cur < 0) break;
            visit(a2, x42);          // the sets swap roles: static names, no copies of registers with loads in flight
        }
    } else {
        while (cur >= 0) {
            if constexpr (!FULLT) ++stat_tiles;
            tm0 = tick();
            const int pf = nxt;
            knn3_tile<CP, FULLT>(a, x4, qreg, xq, xpb, xxb, N, (pf >= 0 ? pf : 0) * 32, pf >= 0, col, h, vec_ok, pd);
            float mx = fmaxf(fmaxf(fmaxf(pd[0], pd[1]), fmaxf(pd[2], pd[3])), fmaxf(fmaxf(pd[4], pd[5]), fmaxf(pd[6], pd[7])));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(pd[8], pd[9]), fmaxf(pd[10], pd[11])), fmaxf(fmaxf(pd[12], pd[13]), fmaxf(pd[14], pd[15]))));
            if (TIMERS && dbg) { const long long t = tick(); tm_tile += t - tm0; tm0 = t; }
            refresh();      // (SPLIT > 1: what the other waves of the workgroup have found meanwhile)
            if (__any(mx >= thrv)) {
                if (cur >= W) {   // rows ascending: later arrivals have larger indices
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (pd[r] >= thrv) {   // ties with the k-th best are admitted: the list decides (NaN padding never passes)
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                            myq[cnt * 64] = make_float2(pd[r], __int_as_float(cur * 32 + row));
                            ++cnt;
                        }
                    }
                } else {          // tile below the wave's own: rows descending, every arrival has the smallest index so far
    #pragma unroll
                    for (int r = 15; r >= 0; --r) {
                        if (pd[r] >= thrv) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                            myq[cnt * 64] = make_float2(pd[r], __int_as_float(cur * 32 + row));
                            ++cnt;
                        }
                    }
                }
                if (TIMERS && dbg) { const long long t = tick(); tm_sel += t - tm0; tm0 = t; }
                if (__any(cnt > L::QCAP - 16)) drain();
                if (TIMERS && dbg) { const long long t = tick(); tm_drain += t - tm0; tm0 = t; }
            }
            cur = nxt;
            nxt = cur >= 0 ? find_next() : -1;
            if (TIMERS && dbg) { const long long t = tick(); tm_adv += t - tm0; tm0 = t; }
    }
    }
    tm0 = tick();
    drain();
    tm_drain += tick() - tm0;
    if constexpr (KMAX == 64) {
        // ---- merge the two half-lists in registers (no LDS region) ----
        if (dbg && q_ok) {   // diagnostics instead of indices
            int32_t* o = idx + ((size_t)b * N + q) * k;
            if (h == 0) { o[0] = stat_tiles; o[1] = stat_it; o[2] = stat_adm; o[3] = stat_drains; }
            else o[4] = stat_adm;
            return;
        }
        knn_merge_halves_bitonic(lv, li);
        if (h == 0 && q_ok) {
            int32_t* out = idx + ((size_t)b * N + q) * k;
#pragma unroll
            for (int s = 0; s < 64; ++s)
                if (s < k) out[s] = li[s];
        }
        return;
    } else {
    // ---- merge the two half-lists (same as the ascending kernel); region: this wave's table + queue (20 KiB >= 64*KMAX*8) ----
    float* mv = reinterpret_cast<float*>(SPLIT > 1 ? qreg_base : smem7 + (size_t)wave * L::WAVE);
    int* mi = reinterpret_cast<int*>(mv + 64 * KMAX);
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        mv[lane * KMAX + s] = lv[s];
        mi[lane * KMAX + s] = li[s];
    }
    __syncthreads();
    if constexpr (SPLIT > 1) {
        // stage 1, every wave: its two half-lists -> one sorted list of KMAX per query, in the registers of the h == 0 lanes (the merged
        // list replaces the half-lists in the wave's region: lanes of one wave run in program order, nobody else reads this region yet)
        {
            const float* av = mv + lane * KMAX;
            const int* ai = mi + lane * KMAX;
            const float* bv = mv + (lane ^ 32) * KMAX;
            const int* bi = mi + (lane ^ 32) * KMAX;
            int pa = 0, pb = 0;
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {      // (both half-lanes run it: the h == 1 copy is not used)
                const float va = av[pa], vb2 = bv[pb];
                const int ia = ai[pa], ib = bi[pb];
                const bool take_a = knn_before(va, ia, vb2, ib);      // -inf / 0x7fffffff padding ranks last; indices are distinct
                lv[s] = take_a ? va : vb2;
                li[s] = take_a ? ia : ib;
                pa += take_a ? 1 : 0;
                pb += take_a ? 0 : 1;
                pa = min(pa, KMAX - 1);           // (a list that has given all KMAX entries is never asked again: s < KMAX)
                pb = min(pb, KMAX - 1);
            }
        }
        float* gv = mv;                           // merged lists of this wave: [32 queries][KMAX] values, then indices
        int* gi = reinterpret_cast<int*>(gv + 32 * KMAX);
        if (h == 0) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                gv[col * KMAX + s] = lv[s];
                gi[col * KMAX + s] = li[s];
            }
        }
        __syncthreads();
        // stage 2, wave 0: SPLIT-way merge of the waves' lists (distinct indices: the order is total and it is the reference's)
        if (wave == 0 && h == 0 && q_ok) {
            float hv[SPLIT];
            int hi[SPLIT], pos[SPLIT];
            const float* mbase = reinterpret_cast<const float*>(smem7 + L::TBYTES) + col * KMAX;      // wave w2's merged lists at + w2 * QBYTES / 4
            constexpr int WSTRIDE = L::QBYTES / 4, IOFF = 32 * KMAX;
#pragma unroll
            for (int w2 = 0; w2 < SPLIT; ++w2) {
                hv[w2] = mbase[w2 * WSTRIDE];
                hi[w2] = __float_as_int(mbase[w2 * WSTRIDE + IOFF]);
                pos[w2] = 0;
            }
            int32_t* out = idx + ((size_t)b * N + q) * k;
            for (int s = 0; s < k; ++s) {
                int bw = 0;
                float bvv = hv[0];
                int bii = hi[0];
#pragma unroll
                for (int w2 = 1; w2 < SPLIT; ++w2) {
                    const bool better = knn_before(hv[w2], hi[w2], bvv, bii);
                    bw = better ? w2 : bw;
                    bvv = better ? hv[w2] : bvv;
                    bii = better ? hi[w2] : bii;
                }
                out[s] = bii;
#pragma unroll
                for (int w2 = 0; w2 < SPLIT; ++w2) {
                    if (bw == w2) {
                        pos[w2] += 1;
                        const bool more = pos[w2] < KMAX;
                        const int pp = pos[w2] < KMAX ? pos[w2] : KMAX - 1;
                        hv[w2] = more ? mbase[w2 * WSTRIDE + pp] : -INFINITY;
                        hi[w2] = more ? __float_as_int(mbase[w2 * WSTRIDE + IOFF + pp]) : 0x7fffffff;
                    }
                }
            }
        }
        return;
    }
    if (dbg && q_ok) {   // diagnostics instead of indices
        int32_t* o = idx + ((size_t)b * N + q) * k;
        if (h == 0) {   // (the phase clocks are not ordered against vector work: the wait for the operand prefetch lands in 'advance')
            o[0] = stat_tiles; o[1] = stat_it; o[2] = stat_adm; o[3] = stat_drains;
            if (k >= 12) { o[5] = 0; o[6] = (int)tm_adv; o[7] = (int)tm_tile; o[8] = (int)tm_sel; o[9] = (int)tm_drain; o[10] = (int)(tick() - tm_start); o[11] = stat_fn; }
        } else o[4] = stat_adm;
    } else
    if (h == 0 && q_ok) {
        const float* av = mv + lane * KMAX;
        const int* ai = mi + lane * KMAX;
        const float* bv = mv + (lane + 32) * KMAX;
        const int* bi = mi + (lane + 32) * KMAX;
        int pa = 0, pb = 0;
        int32_t* out = idx + ((size_t)b * N + q) * k;
        for (int s = 0; s < k; ++s) {
            bool take_a;
            if (pa >= KMAX) take_a = false;
            else if (pb >= KMAX) take_a = true;
            else {
                const float va = av[pa], vb2 = bv[pb];
                take_a = (va > vb2) || (va == vb2 && ai[pa] < bi[pb]);
            }
            if (take_a) out[s] = ai[pa++];
            else out[s] = bi[pb++];
        }
    }
    }   // KMAX != 64
}

template <int CP, int KMAX>
int knn3_launch(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, hipStream_t stream, int dbg = 0)
{
    // packed operands live behind the squared norms in the caller's workspace: [xx: B*N][xp: B*N*2*CP]
    float* xp = const_cast<float*>(xx) + (size_t)B * N;
    if (x) {   // x == nullptr: already packed (point-major entry)
        hipLaunchKernelGGL(knn_pack_kernel<CP>, dim3((N + 255) / 256, B), dim3(256), 0, stream, x, xp, C, N);
        LPD_CHECK_LAUNCH("lpd_knn(pack)");
    }
    size_t lds = (size_t)KNN3_WAVES * knn3_wave_floats<KMAX>() * sizeof(float);
    const int bpc = (N + KNN3_WAVES * 32 - 1) / (KNN3_WAVES * 32);
    auto kern = knn3_kernel<CP, KMAX>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(bpc * B), dim3(KNN3_THREADS), lds, stream, (const float*)xp, xx, idx, N, k, bpc, dbg);
    LPD_CHECK_LAUNCH("lpd_knn");
    return LPD_OK;
}

// workspace behind [xx | xp]: centroids, |c|^2, radii, max |x|^2 per 32-point tile
inline bool knn7_tight();
inline size_t knn7_extra_floats(int B, int N, int CP)
{
    const size_t nt = (size_t)(N + 31) / 32;
    size_t n = (size_t)B * nt * (2 * CP + 3 + 2) + 16;   // + predicted tile counts and launch order (int32 each)
    if (CP == 32 && nt <= (size_t)KNN7_PRE_MAXT && knn7_tight())   // low-precision pass: bf16 image of the operands + the bound table
                                                                    // (B nt^2 16 floats: 1.07 GB at B = 64, N = 16384 -- not reserved when the pass is off)
        n += (size_t)B * nt * 32 * (KNN7_XB / 2) + (size_t)B * nt * nt * 16 + 16;
    return n;
}

// the low-precision bound pass (LPD_DEBUG=knn-pre=0: centroid / radius bounds) and where its bf16 operand image lives in the workspace
inline bool knn7_tight()
{
    static const bool tight = lpd_debug("knn-pre", 1) != 0;
    return tight;
}
inline __bf16* knn7_xb_of(const float* xx, int B, int N)      // 64 channels (CP = 32), one wave per workgroup
{
    const int nt = (N + 31) / 32;
    const float* txmax = xx + (size_t)B * N * (1 + 64) + (size_t)B * nt * (64 + 2);
    const int32_t* order = reinterpret_cast<const int32_t*>(txmax + (size_t)B * nt + 4) + (size_t)B * nt;
    const uintptr_t a0 = (reinterpret_cast<uintptr_t>(order + (size_t)nt * B + 4) + 15) & ~(uintptr_t)15;
    return reinterpret_cast<__bf16*>(a0);
}

template <int CP, int KMAX, bool ONFLY>
int knn7_launch(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, hipStream_t stream, int dbg = 0, bool xb_ready = false,
                bool stats_ready = false)
{
    using L = Knn7Cfg<CP, KMAX, ONFLY>;
    static_assert(L::WAVE >= L::MERGE, "merge region must fit the wave's LDS region");
    const int nt = (N + 31) / 32;
    float* xp = const_cast<float*>(xx) + (size_t)B * N;
    float* cenp = xp + (size_t)B * N * 2 * CP;
    float* cnorm = cenp + (size_t)B * nt * 2 * CP;
    float* rad = cnorm + (size_t)B * nt;
    float* txmax = rad + (size_t)B * nt;
    if (x) hipLaunchKernelGGL(knn_pack_kernel<CP>, dim3((N + 255) / 256, B), dim3(256), 0, stream, x, xp, C, N);
    if (!stats_ready)
        hipLaunchKernelGGL(knn7_tile_stats_kernel<CP>, dim3(nt, B), dim3(64), 0, stream, (const float*)xp, xx, cenp, cnorm, rad, txmax, N, nt);
    LPD_CHECK_LAUNCH("lpd_knn(tile pre-pass)");
    constexpr int WAVES = 1;
    const int bpc = (N + WAVES * 32 - 1) / (WAVES * 32);
    static const bool lpt = lpd_debug("knn-order", 1) != 0;
    int32_t* pred = reinterpret_cast<int32_t*>(txmax + (size_t)B * nt + 4);
    int32_t* order = pred + (size_t)B * nt;
    const int nitems = bpc * B;
    uint16_t* ubq = nullptr;
    if constexpr (CP == 32) {
        if (knn7_tight() && nt <= KNN7_PRE_MAXT) {
            __bf16* xb = knn7_xb_of(xx, B, N);
            ubq = reinterpret_cast<uint16_t*>(xb + (size_t)B * nt * 32 * KNN7_XB);
            if (!xb_ready) hipLaunchKernelGGL(knn7_bf16_kernel, dim3((nt * 64 + 255) / 256, B), dim3(256), 0, stream, (const float*)xp, xx, xb, N, nt);
            const int bx = (nt + 4 * KNN7_BQT - 1) / (4 * KNN7_BQT);
            int zs = 1;                                  // ranges of candidate tiles per query block: until the grid has ~512 blocks
            while (zs < 8 && (long long)bx * B * zs < 512 && (nt + 2 * zs - 1) / (2 * zs) >= 8) zs *= 2;
            const int tchunk = (((nt + zs - 1) / zs) + 3) & ~3;
            hipLaunchKernelGGL(knn7_bound_kernel, dim3(bx, B, (nt + tchunk - 1) / tchunk), dim3(256), 0, stream, (const __bf16*)xb, xx,
                               (const float*)txmax, ubq, N, nt, C, tchunk);
            LPD_CHECK_LAUNCH("lpd_knn(low-precision bounds)");
        }
    }
    static const bool branchy = lpd_debug("knn-fullt", 1) == 0;     // 0: the branchy tile body for every N
    // Small batches: KNN7_SPLIT waves per query tile (see knn7_kernel).  LPD_DEBUG=knn-split=0 never, =1 always (where it is built); default:
    // while the split grid fits the wave slots of the chip (256 CUs x 4 SIMDs x waves per SIMD).
    constexpr int KNN7_SPLIT = 4;
    constexpr bool split_built = KMAX <= 20 && !(ONFLY && CP == 32);
    static const int split_env = lpd_debug("knn-split", -1);
    // Measured (tools/knn_split_bench.py, one stream, N = 4096): 1 cloud 193 -> 144 us (64 channels) / 136 -> 92 us (xyz); 6 clouds
    // 221 -> 239 / 142 -> 116; 10 clouds 233 -> 313 / 155 -> 186; 16 clouds 262 -> 421 / 158 -> 259: the split launch does MORE work
    // (every wave pays the first tile's 16-entry drain, the early thresholds are looser, the lists are merged twice), so it pays
    // exactly while all of its waves are resident at once -- it then is latency, not throughput, that sets the launch time.
    const int split_wps = CP == 2 ? 3 : L::WAVES_PER_SIMD;      // (the split xyz kernel holds 146 registers: three waves per SIMD)
    const bool split = split_built && N % 32 == 0 && !branchy && !dbg && split_env != 0 &&
                       (split_env == 1 || (long long)nitems * KNN7_SPLIT <= 1024ll * split_wps);
    // (longest-first order: pointless while every workgroup of the launch is resident at once)
    const bool use_order = lpt && !(split && (long long)nitems * KNN7_SPLIT <= 1024ll * split_wps);
    if (use_order) {
        const size_t plds = ((size_t)nt * (2 * CP + 1) + nt) * sizeof(float);
        if (plds <= 65536) {
            auto pk = knn7_predict_lds_kernel<CP>;
            (void)hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
            hipLaunchKernelGGL(pk, dim3((nt + KNN7_PRED_WPB - 1) / KNN7_PRED_WPB, B), dim3(256), plds, stream, (const float*)cenp, (const float*)rad, pred, nt);
        } else
            hipLaunchKernelGGL(knn7_predict_kernel<CP>, dim3(nt, B), dim3(128), 0, stream, (const float*)cenp, (const float*)rad, pred, nt);
        hipLaunchKernelGGL(knn7_order_kernel, dim3((nitems / 8 + 1 + 1023) / 1024, 8), dim3(1024), 0, stream, (const int32_t*)pred, order, nitems);
        LPD_CHECK_LAUNCH("lpd_knn(launch order)");
    }
    {
        size_t lds = (size_t)WAVES * L::WAVE;
        int threads = WAVES * 64;
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(bpc * B), dim3(threads), lds, stream, (const float*)xp, xx, (const float*)cenp,
                               (const float*)cnorm, (const float*)rad, (const float*)txmax, idx, (const int32_t*)(use_order ? order : nullptr),
                               N, k, nt, C, bpc, dbg, (const uint16_t*)ubq);
        };
        if constexpr (split_built) {
            if (split) {
                lds = (size_t)L::TBYTES + (size_t)KNN7_SPLIT * L::QBYTES + (size_t)KNN7_SPLIT * 2 * 64 * sizeof(float);
                threads = KNN7_SPLIT * 64;
                go(knn7_kernel<CP, KMAX, WAVES, ONFLY, true, KNN7_SPLIT>);
                LPD_CHECK_LAUNCH("lpd_knn(best-first, split)");
                return LPD_OK;
            }
        }
        if (N % 32 == 0 && !branchy && !dbg) go(knn7_kernel<CP, KMAX, WAVES, ONFLY, true>);      // whole tiles: branch-free tile body
        else go(knn7_kernel<CP, KMAX, WAVES, ONFLY, false>);
        LPD_CHECK_LAUNCH("lpd_knn(best-first)");
    }
    return LPD_OK;
}

// best-first dispatch: the tabulated-bound kernel for N <= 4096, k <= 20 (round 1); bounds on the fly and 64-entry lists for
// larger clouds and neighbourhoods (BASELINE configs[4]: N = 16384, k = 64)
constexpr int KNN7_MAXN = 65536;
inline bool knn7_applies(int C, int N, int k) { return C <= 64 && k <= 64 && N <= KNN7_MAXN; }
inline int knn7_dispatch(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, hipStream_t stream, int dbg, bool xb_ready = false,
                         bool stats_ready = false)
{
    const bool small = k <= 20 && N <= KNN7_MAXT * 32;
    if (C <= 4) return k <= 20 ? knn7_launch<2, 20, true>(x, xx, idx, B, C, N, k, stream, dbg) : knn7_launch<2, 64, true>(x, xx, idx, B, C, N, k, stream, dbg);
    if (small) return knn7_launch<32, 20, false>(x, xx, idx, B, C, N, k, stream, dbg, xb_ready, stats_ready);
    return k <= 20 ? knn7_launch<32, 20, true>(x, xx, idx, B, C, N, k, stream, dbg, xb_ready, stats_ready)
                   : knn7_launch<32, 64, true>(x, xx, idx, B, C, N, k, stream, dbg, xb_ready, stats_ready);
}

template <int CP>
int knn_dispatch_k(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, int impl, hipStream_t stream);

template <int CP>
int knn3_dispatch_k(const float* x, const float* xx, int32_t* idx, int B, int C, int N, int k, hipStream_t stream)
{
    if (k <= 20) return knn3_launch<CP, 20>(x, xx, idx, B, C, N, k, stream);
    if (k <= 32) return knn3_launch<CP, 32>(x, xx, idx, B, C, N, k, stream);
    if (k <= 64) return knn3_launch<CP, 64>(x, xx, idx, B, C, N, k, stream);   // 64-entry lists: one wave per SIMD (stress config K = 64)
    if (!x) { lpd_set_error("lpd_knn_pm: k=%d > 64 unsupported on the point-major entry", k); return LPD_ERR_UNSUPPORTED; }
    return knn_dispatch_k<CP>(x, xx, idx, B, C, N, k, 0, stream);
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

extern "C" long long lpd_knn_workspace_floats(int B, int C, int N, int k)
{
    (void)k;
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const int cp = C <= 4 ? 2 : (C <= 64 ? 32 : 0);
    long long n = (long long)B * N * (1 + 2 * cp);
    if (cp) n += (long long)knn7_extra_floats(B, N, cp);   // tile statistics / visiting orders / repair list of the best-first path
    return n;
}

extern "C" int lpd_knn(const float* x, int B, int C, int N, int k, int32_t* idx, float* xx_ws, int impl,
                       void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(x && idx && xx_ws, "lpd_knn: null pointer");
    LPD_CHECK_ARG(B > 0 && C > 0 && N > 0, "lpd_knn: bad dims B=%d C=%d N=%d", B, C, N);
    LPD_CHECK_ARG(k > 0 && k <= N, "lpd_knn: need 0 < k <= N (k=%d N=%d)", k, N);
    LPD_CHECK_ARG(B <= 65535, "lpd_knn: B=%d exceeds grid.y", B);
    LPD_CHECK_ARG(impl == 0 || impl == 1 || impl == 2 || impl == 4 || impl == 5 || impl == 6, "lpd_knn: impl=%d (0 product; 4 / 6 force the ascending / best-first kernel; 5 statistics; 1 VALU cross-check; 2 first-generation kernel)", impl);
    hipLaunchKernelGGL(knn_sumsq_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, x, xx_ws, C, N);
    LPD_CHECK_LAUNCH("lpd_knn(sumsq)");
    if (((impl == 0 && KNN7_DEFAULT) || (impl == 5 || impl == 6)) && knn7_applies(C, N, k))   // best-first (5: statistics)
        return knn7_dispatch(x, xx_ws, idx, B, C, N, k, stream, impl == 5);
    if (impl == 0 || impl == 4) {   // ascending scan (larger clouds, k > 20; impl 4: forced, for A/B timing)
        if (C <= 4) return knn3_dispatch_k<2>(x, xx_ws, idx, B, C, N, k, stream);
        if (C <= 64) return knn3_dispatch_k<32>(x, xx_ws, idx, B, C, N, k, stream);
        if (C <= 256) return knn_dispatch_k<128>(x, xx_ws, idx, B, C, N, k, 0, stream);   // wide features: v1
    } else {
        const int v1impl = impl == 1 ? 1 : 0;   // 1: VALU cross-check, 2: v1 MFMA + in-scan insertion
        if (C <= 4) return knn_dispatch_k<2>(x, xx_ws, idx, B, C, N, k, v1impl, stream);
        if (C <= 64) return knn_dispatch_k<32>(x, xx_ws, idx, B, C, N, k, v1impl, stream);
        if (C <= 256) return knn_dispatch_k<128>(x, xx_ws, idx, B, C, N, k, v1impl, stream);
    }
    lpd_set_error("lpd_knn: C=%d > 256 unsupported", C);
    return LPD_ERR_UNSUPPORTED;
}

namespace {
// does lpd_knn_pm(impl) run the low-precision bound pass on these sizes (and so want the bf16 operand image)?
inline bool knn_pm_wants_xb(int C, int N, int k, int impl)
{
    const bool best_first = ((impl == 0 && KNN7_DEFAULT) || (impl == 5 || impl == 6)) && knn7_applies(C, N, k);
    return best_first && C == 64 && knn7_tight() && N <= KNN7_PRE_MAXT * 32 && N % 32 == 0;
}
}  // namespace

extern "C" int lpd_knn_pm_layout(int B, int C, int N, int k, float* ws, float** xx, float** xp, void** xb, float** tiles)
{
    LPD_CHECK_ARG(ws && xx && xp && xb && tiles && B > 0 && N > 0 && C > 0 && C <= 64, "lpd_knn_pm_layout: bad arguments");
    const int cp = C <= 4 ? 2 : 32;
    *xx = ws;
    *xp = ws + (size_t)B * N;
    *xb = (C == 64 && knn_pm_wants_xb(C, N, k, 0)) ? (void*)knn7_xb_of(ws, B, N) : nullptr;
    *tiles = (KNN7_DEFAULT && knn7_applies(C, N, k)) ? *xp + (size_t)B * N * 2 * cp : nullptr;
    return LPD_OK;
}

// Point-major entry: x_pm [B*N][ld] rows (C <= 64 channels used).  Same results as lpd_knn on the transposed input; skips
// the channel-major round trip (transpose + pack) that the pipeline would otherwise pay for each graph.
extern "C" int lpd_knn_pm(const float* x_pm, int ld, int B, int C, int N, int k, int32_t* idx, float* ws, int impl, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG((x_pm || (impl & LPD_KNN_PM_PREPARED)) && idx && ws, "lpd_knn_pm: null pointer");
    LPD_CHECK_ARG(B > 0 && C > 0 && N > 0 && ld >= C, "lpd_knn_pm: bad dims B=%d C=%d N=%d ld=%d", B, C, N, ld);
    LPD_CHECK_ARG(k > 0 && k <= N, "lpd_knn_pm: need 0 < k <= N (k=%d N=%d)", k, N);
    LPD_CHECK_ARG(C <= 64 && k <= 64, "lpd_knn_pm: built for C <= 64, k <= 64 (got C=%d k=%d); use lpd_knn on the channel-major tensor", C, k);
    LPD_CHECK_ARG(C <= 4 || (impl & LPD_KNN_PM_PREPARED) || ((uintptr_t)x_pm & 15) == 0, "lpd_knn_pm: x_pm must be 16-byte aligned");
    const long long M = (long long)B * N;
    float* xp = ws + M;
    const bool prepped = (impl & LPD_KNN_PM_PREPARED) != 0;     // the operands are in ws already (lpd_lpdnet_front)
    impl &= ~LPD_KNN_PM_PREPARED;
    LPD_CHECK_ARG(impl == 0 || impl == 4 || impl == 5 || impl == 6, "lpd_knn_pm: impl=%d (0 product; 4 / 6 force the ascending / best-first kernel; 5 statistics)", impl);
    const bool best_first = ((impl == 0 && KNN7_DEFAULT) || (impl == 5 || impl == 6)) && knn7_applies(C, N, k);
    // the bf16 operand image of the low-precision bound pass is written with the operands when that pass will run
    const bool xb_ready = knn_pm_wants_xb(C, N, k, impl) && (prepped || ld % 4 == 0);
    LPD_CHECK_ARG(!prepped || C == 64, "lpd_knn_pm: prepared operands are a 64-channel affair (C=%d)", C);
    if (prepped) { /* nothing to do */ }
    else if (C <= 4) hipLaunchKernelGGL(knn_prep_pm_kernel<2>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, x_pm, ld, ws, xp, C, M);
    else if (C == 64 && ld % 4 == 0) {      // four lanes per point
        hipLaunchKernelGGL(knn_prep_pm4_kernel, dim3((unsigned)((4 * M + 255) / 256)), dim3(256), 0, stream, x_pm, ld, ws, xp,
                           xb_ready ? knn7_xb_of(ws, B, N) : (__bf16*)nullptr, M, N, (N + 31) / 32);
    } else hipLaunchKernelGGL(knn_prep_pm_kernel<32>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, x_pm, ld, ws, xp, C, M);
    LPD_CHECK_LAUNCH("lpd_knn_pm(prep)");
    if (best_first) return knn7_dispatch(nullptr, ws, idx, B, C, N, k, stream, impl == 5, xb_ready, prepped);
    if (C <= 4) return knn3_dispatch_k<2>(nullptr, ws, idx, B, C, N, k, stream);
    return knn3_dispatch_k<32>(nullptr, ws, idx, B, C, N, k, stream);
}
