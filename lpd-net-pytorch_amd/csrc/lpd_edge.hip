// lpd_edge.hip -- kNN-graph neighbour aggregation kernels.
//
// Replaces util/lpdnet_model.py:331-363 (`get_graph_feature`: gather + repeat + cat + permute, which
// materialises a [B,2C,N,k] tensor) fused with the edge convolutions that consume it
// (lpdnet_model.py:249-258: convDG1 -> max, convDG2 -> max, convSN1 -> max; and the
// LPDNetOrign variants lpdnet_model.py:96-107).
//
// Algebra: a 1x1 conv on cat(neighbour_j, centre_i) is
//     W[:, :C] f_j + W[:, C:] f_i  =  P_j + Q_i,
// BatchNorm is a per-channel affine (scale s, shift b) and LeakyReLU/ReLU is monotone, so
//     max_j act(s (P_j + Q_i) + b) = act(s * sel_j P_j + s Q_i + b),  sel = max if s >= 0 else min.
// P and Q come from ONE dense GEMM per stage (lpd_gemm.hip); the [B,2C,N,k] edge tensor never
// exists.  Two kernels:
//
//  * edge_gather_max  (K-agg, the HBM-bound kNN-aggregation kernel of BASELINE.json):
//      out[i][c] = act(s[c] * (sel_j P[idx[i][j]][c] + Q[i][c]) + b[c])
//    one (part of a) wavefront per point, 16 B per lane, k independent row loads in flight.
//    Algorithmic bytes per point (fp32, counted once, gathers not replayed):
//      C*4 (P row) + C*4 (Q row) + 4*k (idx) + C*4 (out)  = 3152 B at C = 256, k = 20.
//
//  * edge_mlp  (DG1 -> DG2 chain, lpdnet_model.py:249-252): the second conv consumes the
//    post-activation per-edge tensor, so it cannot be split; it is the one genuine per-edge dense
//    contraction.  Per block: 64 points; for each neighbour slot t the tile
//      Y1_t[p][c] = act(s1[c] (P[idx[p][t]][c] + Q[p][c]) + b1[c])
//    is built in LDS straight from gathered P rows and multiplied by W2 on the f32 MFMA; running
//    per-point max/min of the raw product stay in registers across the k slots, so the
//    [B,C,N,k] intermediate never reaches HBM.  Output: act(s2 * sel_t z_t + b2).
#include "lpd_common.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------
// K-agg: gather-max aggregation
// ------------------------------------------------------------------------------------------
struct GatherArgs {
    const float* P;      // [M][ldp] neighbour projections
    const float* Q;      // [M][ldq] centre term or null
    const int32_t* idx;  // [M][k] neighbour indices, local to the cloud
    float* out;          // [M][ldo]
    const float* scale;  // [C] or null (=> 1)
    const float* shift;  // [C] or null (=> 0)
    int M, N, C, k;
    int ldp, ldq, ldo;
    int act;
    float slope;
};

// LPP = lanes per point = C / 4 (16, 32 or 64)
template <int LPP>
__global__ __launch_bounds__(256) void edge_gather_max_kernel(GatherArgs g)
{
    constexpr int PPW = 64 / LPP;  // points per wave
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int sub = lane / LPP;         // which point of the wave
    const int cl = lane % LPP;          // float4 column of the point's row
    const int waves_per_block = blockDim.x >> 6;
    const int nwork = (g.M + PPW - 1) / PPW;  // wave-sized work items
    // Work distribution for L2 locality (profiles/r01b: with one contiguous chunk per block the 256 blocks resident on
    // an XCD touched 256 places 64 points apart -- a 16 MB span of P against a 4 MB L2, hit rate 35 %).
    // Blocks b, b+8, b+16, ... share an XCD (round-robin dispatch).  Each XCD owns one contiguous range of work items
    // and its resident blocks sweep that range TOGETHER: at any time they cover one window of (blocks per XCD * waves)
    // consecutive points, whose neighbour rows (Z-ordered clouds) fit the XCD's L2.
    const int nx = 8;
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const int blocks_per_xcd = (gridDim.x + nx - 1) / nx;   // slots; the last ones may be missing on some XCDs
    const int items_per_xcd = (nwork + nx - 1) / nx;
    const int x_begin = xcd * items_per_xcd;
    const int x_end = min(x_begin + items_per_xcd, nwork);
    const int sweep = blocks_per_xcd * waves_per_block;

    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.scale) sc = *reinterpret_cast<const float4*>(g.scale + cl * 4);
    if (g.shift) sh = *reinterpret_cast<const float4*>(g.shift + cl * 4);

    for (int w = x_begin + slot * waves_per_block + wave_in_block; w < x_end; w += sweep) {
        const int m = w * PPW + sub;
        const bool ok = m < g.M;
        const int mm = ok ? m : g.M - 1;
        const int cloud_base = (mm / g.N) * g.N;
        // one coalesced load of the point's k indices, broadcast by shuffle below
        int my_idx = 0;
        if (cl < g.k) my_idx = g.idx[(size_t)mm * g.k + cl];
        float4 vmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        float4 vmin = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        for (int t0 = 0; t0 < g.k; t0 += LPP) {
            int cur_idx = my_idx;
            if (t0 > 0) {  // k > lanes-per-point: fetch the next batch of indices
                cur_idx = (t0 + cl < g.k) ? g.idx[(size_t)mm * g.k + t0 + cl] : 0;
            }
            const int tn = min(g.k - t0, LPP);
#pragma unroll 10
            for (int t = 0; t < tn; ++t) {
                const int j = __shfl(cur_idx, sub * LPP + t, 64);
                const float4 p = *reinterpret_cast<const float4*>(g.P + (size_t)(cloud_base + j) * g.ldp + cl * 4);
                vmax.x = fmaxf(vmax.x, p.x); vmax.y = fmaxf(vmax.y, p.y);
                vmax.z = fmaxf(vmax.z, p.z); vmax.w = fmaxf(vmax.w, p.w);
                vmin.x = fminf(vmin.x, p.x); vmin.y = fminf(vmin.y, p.y);
                vmin.z = fminf(vmin.z, p.z); vmin.w = fminf(vmin.w, p.w);
            }
        }
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.Q) q = *reinterpret_cast<const float4*>(g.Q + (size_t)mm * g.ldq + cl * 4);
        float4 o;
        o.x = lpd_act(sc.x * ((sc.x >= 0.f ? vmax.x : vmin.x) + q.x) + sh.x, g.act, g.slope);
        o.y = lpd_act(sc.y * ((sc.y >= 0.f ? vmax.y : vmin.y) + q.y) + sh.y, g.act, g.slope);
        o.z = lpd_act(sc.z * ((sc.z >= 0.f ? vmax.z : vmin.z) + q.z) + sh.z, g.act, g.slope);
        o.w = lpd_act(sc.w * ((sc.w >= 0.f ? vmax.w : vmin.w) + q.w) + sh.w, g.act, g.slope);
        if (ok) *reinterpret_cast<float4*>(g.out + (size_t)m * g.ldo + cl * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------
// fused edge MLP (DG1 activation -> DG2 conv -> max over k)
// ------------------------------------------------------------------------------------------
struct EdgeMlpArgs {
    const float* P;      // [M][ldp] raw neighbour projection of stage 1
    const float* Q;      // [M][ldq] raw centre projection of stage 1, or null
    const int32_t* idx;  // [M][k]
    const float* s1;     // [CM] stage-1 BN scale
    const float* b1;     // [CM] stage-1 BN shift
    const float* W2;     // [CO][CM] stage-2 conv weight (torch [out,in] layout)
    const float* s2;     // [CO] stage-2 BN scale
    const float* b2;     // [CO] stage-2 BN shift
    float* out;          // [M][ldo] act(s2 * sel_t z + b2)
    int M, N, k;
    int ldp, ldq, ldo;
    int act;
    float slope;
};

constexpr int EM_PTS = 64;       // points per block
constexpr int EM_THREADS = 256;

template <int CM, int CO>
struct EdgeMlpCfg {
    static constexpr int LDA = EM_PTS + 1;   // As[c][p]
    static constexpr int LDB = CO + 1;       // Bs[c][o]
    static constexpr int A_FLOATS = CM * LDA;
    static constexpr int B_FLOATS = CM * LDB;
    static constexpr int F4_PER_ROW = CM / 4;              // float4 per gathered row
    static constexpr int ROWS_PER_PASS = EM_THREADS / F4_PER_ROW;
    static constexpr int PASSES = EM_PTS / ROWS_PER_PASS;  // rows each thread gathers per slot
    static constexpr int TN = CO / 64;                     // 32-wide n-tiles per wave
};

template <int CM, int CO>
__global__ __launch_bounds__(EM_THREADS) void edge_mlp_kernel(EdgeMlpArgs g)
{
    using Cfg = EdgeMlpCfg<CM, CO>;
    constexpr int LDA = Cfg::LDA, LDB = Cfg::LDB, TN = Cfg::TN, PASSES = Cfg::PASSES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                              // [CM][LDB]
    float* As = smem + Cfg::B_FLOATS;              // [2][CM][LDA]
    int* idxs = reinterpret_cast<int*>(As + 2 * Cfg::A_FLOATS);  // [EM_PTS][k]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const int wp = wave & 1;   // point tile (32 points)
    const int wo = wave >> 1;  // output-channel half
    // consecutive point blocks on one XCD: their gathered P rows share that XCD's L2
    const int m0 = lpd_xcd_remap(blockIdx.x, gridDim.x) * EM_PTS;

    // stage W2 [CO][CM] -> Bs[c][o]
    for (int f = tid; f < CO * CM / 4; f += EM_THREADS) {
        int o = f / (CM / 4), cq = f % (CM / 4);
        float4 w = *reinterpret_cast<const float4*>(g.W2 + (size_t)o * CM + cq * 4);
        Bs[(cq * 4 + 0) * LDB + o] = w.x;
        Bs[(cq * 4 + 1) * LDB + o] = w.y;
        Bs[(cq * 4 + 2) * LDB + o] = w.z;
        Bs[(cq * 4 + 3) * LDB + o] = w.w;
    }
    for (int f = tid; f < EM_PTS * g.k; f += EM_THREADS) {
        int p = f / g.k, t = f - p * g.k;
        int m = m0 + p;
        m = m < g.M ? m : g.M - 1;
        idxs[f] = (m / g.N) * g.N + g.idx[(size_t)m * g.k + t];  // global row of the neighbour
    }

    // gather role of this thread: float4 column c4 of rows prow + ROWS_PER_PASS * e
    const int c4 = tid % Cfg::F4_PER_ROW;
    const int prow = tid / Cfg::F4_PER_ROW;
    const float4 s1 = *reinterpret_cast<const float4*>(g.s1 + c4 * 4);
    const float4 b1 = *reinterpret_cast<const float4*>(g.b1 + c4 * 4);
    float4 qc[PASSES];  // centre term folded with the BN affine: s1 * Q + b1
#pragma unroll
    for (int e = 0; e < PASSES; ++e) {
        int m = m0 + prow + Cfg::ROWS_PER_PASS * e;
        m = m < g.M ? m : g.M - 1;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.Q) q = *reinterpret_cast<const float4*>(g.Q + (size_t)m * g.ldq + c4 * 4);
        qc[e].x = s1.x * q.x + b1.x;
        qc[e].y = s1.y * q.y + b1.y;
        qc[e].z = s1.z * q.z + b1.z;
        qc[e].w = s1.w * q.w + b1.w;
    }
    __syncthreads();  // idxs + Bs visible

    float4 pg[PASSES];
    auto gather = [&](int t) {
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            int p = prow + Cfg::ROWS_PER_PASS * e;
            int row = idxs[p * g.k + t];
            pg[e] = *reinterpret_cast<const float4*>(g.P + (size_t)row * g.ldp + c4 * 4);
        }
    };
    auto build = [&](int buf) {
        float* as = As + buf * Cfg::A_FLOATS;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            int p = prow + Cfg::ROWS_PER_PASS * e;
            // s (P + Q) + b  ==  s P + (s Q + b)
            as[(c4 * 4 + 0) * LDA + p] = lpd_act(s1.x * pg[e].x + qc[e].x, g.act, g.slope);
            as[(c4 * 4 + 1) * LDA + p] = lpd_act(s1.y * pg[e].y + qc[e].y, g.act, g.slope);
            as[(c4 * 4 + 2) * LDA + p] = lpd_act(s1.z * pg[e].z + qc[e].z, g.act, g.slope);
            as[(c4 * 4 + 3) * LDA + p] = lpd_act(s1.w * pg[e].w + qc[e].w, g.act, g.slope);
        }
    };

    f32x16 zmax[TN], zmin[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            zmax[j][r] = -INFINITY;
            zmin[j][r] = INFINITY;
        }

    gather(0);
    build(0);
    __syncthreads();

    for (int t = 0; t < g.k; ++t) {
        const int buf = t & 1;
        if (t + 1 < g.k) gather(t + 1);  // rows in flight under this slot's MFMAs
        const float* as = As + buf * Cfg::A_FLOATS + wp * 32 + col;
        const float* bs = Bs + wo * (32 * TN) + col;
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll 8
        for (int s = 0; s < CM / 2; ++s) {
            const float a = as[(2 * s + h) * LDA];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float b = bs[(2 * s + h) * LDB + j * 32];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                zmax[j][r] = fmaxf(zmax[j][r], acc[j][r]);
                zmin[j][r] = fminf(zmin[j][r], acc[j][r]);
            }
        if (t + 1 < g.k) build(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int o = wo * (32 * TN) + j * 32 + col;
        const float sc = g.s2[o], sh = g.b2[o];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wp * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= g.M) continue;
            const float z = sc >= 0.f ? zmax[j][r] : zmin[j][r];
            g.out[(size_t)m * g.ldo + o] = lpd_act(sc * z + sh, g.act, g.slope);
        }
    }
}

template <int CM, int CO>
int edge_mlp_launch(const EdgeMlpArgs& g, hipStream_t stream)
{
    using Cfg = EdgeMlpCfg<CM, CO>;
    size_t lds = (size_t)(Cfg::B_FLOATS + 2 * Cfg::A_FLOATS) * sizeof(float) + (size_t)EM_PTS * g.k * sizeof(int);
    auto kern = edge_mlp_kernel<CM, CO>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((g.M + EM_PTS - 1) / EM_PTS);
    hipLaunchKernelGGL(kern, grid, dim3(EM_THREADS), lds, stream, g);
    LPD_CHECK_LAUNCH("lpd_edge_mlp");
    return LPD_OK;
}

}  // namespace

extern "C" int lpd_edge_gather_max(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, float* out,
                                   int ldo, const float* scale, const float* shift, int M, int N, int C, int k,
                                   int act, float slope, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx && out, "lpd_edge_gather_max: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && k > 0 && M % N == 0, "lpd_edge_gather_max: bad dims M=%d N=%d k=%d", M, N, k);
    LPD_CHECK_ARG(C == 64 || C == 128 || C == 256, "lpd_edge_gather_max: C=%d unsupported (64/128/256)", C);
    LPD_CHECK_ARG(ldp % 4 == 0 && ldo % 4 == 0 && (!Q || ldq % 4 == 0), "lpd_edge_gather_max: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)out | (uintptr_t)Q | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
                  "lpd_edge_gather_max: pointers must be 16-byte aligned");
    GatherArgs g{P, Q, idx, out, scale, shift, M, N, C, k, ldp, ldq, ldo, act, slope};
    const int lpp = C / 4;
    const int nwork = (M + (64 / lpp) - 1) / (64 / lpp);
    int blocks = (nwork + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks of 4 waves per CU, grid-stride beyond
    blocks = (blocks + 7) / 8 * 8;           // every XCD gets the same number of slots (the kernel's sweep relies on it)
    if (lpp == 64) hipLaunchKernelGGL(edge_gather_max_kernel<64>, dim3(blocks), dim3(256), 0, stream, g);
    else if (lpp == 32) hipLaunchKernelGGL(edge_gather_max_kernel<32>, dim3(blocks), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(edge_gather_max_kernel<16>, dim3(blocks), dim3(256), 0, stream, g);
    LPD_CHECK_LAUNCH("lpd_edge_gather_max");
    return LPD_OK;
}

extern "C" int lpd_edge_mlp(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                            const float* b1, const float* W2, const float* s2, const float* b2, float* out, int ldo,
                            int M, int N, int CM, int CO, int k, int act, float slope, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx && s1 && b1 && W2 && s2 && b2 && out, "lpd_edge_mlp: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && k > 0 && M % N == 0, "lpd_edge_mlp: bad dims M=%d N=%d k=%d", M, N, k);
    LPD_CHECK_ARG(k <= 128, "lpd_edge_mlp: k=%d > 128 unsupported", k);
    LPD_CHECK_ARG(ldp % 4 == 0 && (!Q || ldq % 4 == 0), "lpd_edge_mlp: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)s1 | (uintptr_t)b1 | (uintptr_t)W2) & 15) == 0,
                  "lpd_edge_mlp: pointers must be 16-byte aligned");
    EdgeMlpArgs g{P, Q, idx, s1, b1, W2, s2, b2, out, M, N, k, ldp, ldq, ldo, act, slope};
    if (CM == 128 && CO == 128) return edge_mlp_launch<128, 128>(g, stream);
    if (CM == 64 && CO == 64) return edge_mlp_launch<64, 64>(g, stream);
    lpd_set_error("lpd_edge_mlp: (CM=%d, CO=%d) unsupported; built for (128,128) and (64,64)", CM, CO);
    return LPD_ERR_UNSUPPORTED;
}
