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
    // cloud-resident form only: float offsets between consecutive 8-channel slices and between consecutive clouds of
    // P / Q / out.  Row-major operands: slice 8 (the next 8 columns of the row), cloud N * ld.  Cloud-panel operands
    // [cloud][C/8][N][8]: slice N * 8 with a row stride of 8, cloud = floats between clouds in that buffer.
    long long p_slice, q_slice, o_slice;
    long long p_cloud, q_cloud, o_cloud;
    // split output (cloud-resident form, cloud-panel out only): `out` is the hi plane of a bf16 pair of planes, the lo plane o_lo
    // elements behind it (o_slice / o_cloud count bf16 elements); 0 = fp32 output
    long long o_lo;
};

// one point's result of the cloud-resident kernels: fp32 float4 into the row, or (SPLIT) the pair-exchanged hi / lo word
template <bool SPLIT>
__device__ __forceinline__ void kagg_store(float* outc, unsigned row_off, const float4& r, int cl, long long o_lo)
{
    if constexpr (!SPLIT) *reinterpret_cast<float4*>(outc + row_off) = r;
    else {
        // outc = hi plane + cloud + slice (bf16 elements, no lane part); row_off = row * 8
        __bf16* p = reinterpret_cast<__bf16*>(outc) + (cl ? o_lo : 0) + row_off;
        *reinterpret_cast<uint4*>(p) = lpd_split8_pair(r, cl);
    }
}

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
// K-agg, cloud-resident form (the product path for N*32 B <= 160 KiB, k = 20).
//
// One workgroup owns (cloud, slice of 8 channels) and keeps that slice of ALL N rows of P in LDS (N * 32 bytes: 128 KiB
// at N = 4096), so each of the k gathers per point is an LDS read and P, Q and out cross the memory system once; the
// direct form above sends k = 20 row reads per point through L2 (2.7 GB per launch at B = 32) and tops out at the L2
// gather rate.  The rows are stored sign-adjusted (sgn(scale) * P) so that one running max serves both selections:
// sel_j P_j = sgn * max_j (sgn * P_j), exact.  2 lanes per point (16 B each), 512 points per block pass.
// The kernel is bound by bytes through L2 per CU: with C/8 slices per cloud every slice re-reads the index rows, so
// they come as uint16 (lpd_pack_idx16; 40 B instead of 80 B per point and slice) and the per-pass operands (indices,
// centre term) rotate through three register sets, two passes of lookahead, no copies.  Straight-line body: the
// activation is the branch-free  max(v,0) + ns * min(v,0)  with ns = 1 / 0 / slope for none / ReLU / LeakyReLU.
// (Tried and dropped in round 1, HISTORY.md 3.3: Z-order windows of 256..1024 rows in LDS with the misses from
// L2 -- 13 % misses cost more instructions than the hits save; panel-major P/Q -- no gain, the 32-B row pieces are
// not over-fetched.)
// ------------------------------------------------------------------------------------------
constexpr unsigned KAGG_IMG1 = 65536 + 128;   // byte offset of the second LDS image (channels 4-7 of the slice)

// Packed indices (lpd_pack_idx16), round 6: per block of 32 points [chunk 0: indices 0-7 of the 32 points, 16 B each | chunk 1: indices
// 8-15 | chunk 2: indices 16-19, 8 B each] = 1280 bytes.  The two lanes of a point (cl = 0 / 1) load ONE 16-byte piece each (chunk cl) plus
// the shared 8-byte tail and hand their piece to the partner by DPP where the gathers consume it: two load instructions per point and
// pass instead of five 8-byte ones, the same ten cache lines per wave.  (The kernel is bound by memory requests per CU -- its Q and out
// traffic, 2 of 8 instructions per pass, cost 26 % of its time -- and the index loads were 5 of those 8.)
struct CloudOps {
    uint4 own;      // indices 8 cl .. 8 cl + 7 (16-bit byte offsets of the row pieces)
    uint2 tail;     // indices 16-19
    float4 q;
    unsigned m;     // global row of the point (32-bit element offsets: the host checks M * ld * 4 < 2^32)
};
__device__ __forceinline__ void kagg_load_idx(CloudOps& o, const unsigned char* idxb, unsigned cl)
{
    const unsigned char* blk = idxb + (size_t)(o.m >> 5) * 1280u + (o.m & 31u) * 8u;
    o.own = *reinterpret_cast<const uint4*>(blk + cl * 512u + (o.m & 31u) * 8u);
    o.tail = *reinterpret_cast<const uint2*>(blk + 1024u);
}
// the five index quads of a point in SLOT order (ix[i].x: slots 4 i, 4 i + 1; .y: 4 i + 2, 4 i + 3)
__device__ __forceinline__ void kagg_quads(const CloudOps& o, unsigned cl, uint2 (&ix)[5])
{
    const uint4 pt = make_uint4(lpd_lane_xor1(o.own.x), lpd_lane_xor1(o.own.y), lpd_lane_xor1(o.own.z), lpd_lane_xor1(o.own.w));
    const uint4 lo = cl ? pt : o.own, hi = cl ? o.own : pt;
    ix[0] = make_uint2(lo.x, lo.y); ix[1] = make_uint2(lo.z, lo.w);
    ix[2] = make_uint2(hi.x, hi.y); ix[3] = make_uint2(hi.z, hi.w);
    ix[4] = o.tail;
}

template <bool HAS_Q, bool SPLIT = false>
__global__ __launch_bounds__(1024) void edge_gather_max_cloud16_kernel(GatherArgs g, const uint16_t* __restrict__ idx16,
                                                                       int nslices, float ns)
{
    extern __shared__ float4 win[];   // two images [N][16 B]: channels 0-3 at byte 0, channels 4-7 at byte KAGG_IMG1
    constexpr int GROUPS = 512, KQ = 5;
    const int tid = threadIdx.x;
    const int cl = tid & 1;
    const int grp = tid >> 1;
    // The packed indices are BYTE offsets of the 16-byte row pieces (16 * j, lpd_pack_idx16): a gather address is one
    // SDWA add of a 16-bit half to this lane's image base (the 32-byte-row form cost and + shift + add per gather, a
    // fifth of the kernel's instructions; with no memory traffic at all the kernel ran 55 of its 107 us: issue-bound).
    // The second image starts half a bank row (128 B) past 64 KiB so the two lanes of a point never share a bank.
    const unsigned lbase = cl * KAGG_IMG1;
    const char* winb = reinterpret_cast<const char*>(win);
    const int w = lpd_xcd_remap(blockIdx.x, gridDim.x);   // the slices of one cloud run next to each other on one XCD
    const int sl = w % nslices;
    const int b = w / nslices;
    const int col = sl * 8 + cl * 4;
    const unsigned row0 = (unsigned)b * g.N;
    const int N = g.N;
    const int passes = (N + GROUPS - 1) / GROUPS;
    const unsigned char* idxb = reinterpret_cast<const unsigned char*>(idx16);
    const float* Qc = g.Q + b * g.q_cloud + sl * g.q_slice + cl * 4;      // row n of the cloud at + n * ld
    float* outc = SPLIT ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(g.out) + b * g.o_cloud + sl * g.o_slice)
                        : g.out + b * g.o_cloud + sl * g.o_slice + cl * 4;
    const unsigned ldq = g.ldq, ldo = g.ldo;

    // A pass past the end of the cloud re-reads (and later re-stores, with identical values) the last point:
    // no divergent tail, so the row reads below are consumed as they arrive instead of being sunk into a branch.
    auto load = [&](CloudOps& o, int ps) {
        o.m = row0 + min(ps * GROUPS + grp, N - 1);
        kagg_load_idx(o, idxb, cl);
        if (HAS_Q) o.q = *reinterpret_cast<const float4*>(Qc + (o.m - row0) * ldq);
    };

    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.scale) sc = *reinterpret_cast<const float4*>(g.scale + col);
    if (g.shift) sh = *reinterpret_cast<const float4*>(g.shift + col);
    const float4 sg = make_float4(sc.x >= 0.f ? 1.f : -1.f, sc.y >= 0.f ? 1.f : -1.f, sc.z >= 0.f ? 1.f : -1.f,
                                  sc.w >= 0.f ? 1.f : -1.f);
    // the cloud's slice of P: every row piece of the workgroup is requested before the first one is stored to LDS
    const float* Pc = g.P + b * g.p_cloud + sl * g.p_slice + cl * 4;
    {
        constexpr int NP = 4096 / GROUPS;
        float4 pr[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) pr[i] = *reinterpret_cast<const float4*>(Pc + (size_t)min(grp + i * GROUPS, N - 1) * g.ldp);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = grp + i * GROUPS;
            float4 p = pr[i];
            p.x *= sg.x; p.y *= sg.y; p.z *= sg.z; p.w *= sg.w;
            if (r < N) *reinterpret_cast<float4*>(const_cast<char*>(winb) + lbase + r * 16) = p;
        }
    }
    CloudOps A, Bo, Co;
    load(A, 0);
    load(Bo, 1);
    __syncthreads();

    auto process = [&](const CloudOps& o) {
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        const uint2 ix[5] = {make_uint2(o.own.x, o.own.y), make_uint2(o.own.z, o.own.w),
                             make_uint2(lpd_lane_xor1(o.own.x), lpd_lane_xor1(o.own.y)), make_uint2(lpd_lane_xor1(o.own.z), lpd_lane_xor1(o.own.w)), o.tail};
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(winb + ((ix[i].x & 0xffffu) + lbase));
            const float4 bq = *reinterpret_cast<const float4*>(winb + ((ix[i].x >> 16) + lbase));
            const float4 c = *reinterpret_cast<const float4*>(winb + ((ix[i].y & 0xffffu) + lbase));
            const float4 d = *reinterpret_cast<const float4*>(winb + ((ix[i].y >> 16) + lbase));
            v.x = fmaxf(fmaxf(v.x, a.x), bq.x); v.y = fmaxf(fmaxf(v.y, a.y), bq.y);
            v.z = fmaxf(fmaxf(v.z, a.z), bq.z); v.w = fmaxf(fmaxf(v.w, a.w), bq.w);
            v.x = fmaxf(fmaxf(v.x, c.x), d.x); v.y = fmaxf(fmaxf(v.y, c.y), d.y);
            v.z = fmaxf(fmaxf(v.z, c.z), d.z); v.w = fmaxf(fmaxf(v.w, c.w), d.w);
            __builtin_amdgcn_sched_barrier(0);   // 4 row reads in flight per wave (measured at C=256: 2 -> 137 us, 4 -> 131, 8 -> 178)
        }
        float4 r;
        r.x = sc.x * (sg.x * v.x + (HAS_Q ? o.q.x : 0.f)) + sh.x;
        r.y = sc.y * (sg.y * v.y + (HAS_Q ? o.q.y : 0.f)) + sh.y;
        r.z = sc.z * (sg.z * v.z + (HAS_Q ? o.q.z : 0.f)) + sh.z;
        r.w = sc.w * (sg.w * v.w + (HAS_Q ? o.q.w : 0.f)) + sh.w;
        r.x = fmaxf(r.x, 0.f) + ns * fminf(r.x, 0.f);
        r.y = fmaxf(r.y, 0.f) + ns * fminf(r.y, 0.f);
        r.z = fmaxf(r.z, 0.f) + ns * fminf(r.z, 0.f);
        r.w = fmaxf(r.w, 0.f) + ns * fminf(r.w, 0.f);
        kagg_store<SPLIT>(outc, (o.m - row0) * ldo, r, cl, g.o_lo);
    };
    for (int ps = 0; ps < passes; ps += 3) {
        load(Co, ps + 2);
        process(A);
        load(A, ps + 3);
        if (ps + 1 < passes) process(Bo);
        load(Bo, ps + 4);
        if (ps + 2 < passes) process(Co);
    }
}

// The TRAINING forward of the split-form stage (lpd_train2.hip (1): S = sum_t P_nbr, the selected raw value + its slot, and the
// closed-form BatchNorm statistics) on the same cloud-resident organisation: what edge_split_fwd_kernel gathers as 1-KiB rows through
// L2 (3.7 GB at B = 44, C = 256: 559 us) comes from the LDS-resident slice here.  Row-major P / Q ([M][ld], slice = 8 columns).
struct SplitFwdArgs {
    const float* P;
    const float* Q;
    const float* gamma;      // [C]: the sign of the BatchNorm scale (selection = max where gamma >= 0, min otherwise)
    float* S;                // [M][C]
    float* usel;             // [M][C]
    uint8_t* arg;            // [M][C]
    double* sum;             // statistics replicas (lpd_common.h): column c of this block's replica at + c
    double* sumsq;
    int N, C;
    int ldp, ldq;
};

template <int THREADS>     // 1024: the sums / arg-max bookkeeping does not fit 128 registers (102 spilled, 844 us); 512
__global__ __launch_bounds__(THREADS) void edge_split_fwd_cloud16_kernel(SplitFwdArgs g, const uint16_t* __restrict__ idx16, int nslices)
{
    extern __shared__ float4 win[];   // two images [N][16 B]: channels 0-3 at byte 0, channels 4-7 at byte KAGG_IMG1
    constexpr int GROUPS = THREADS / 2, KQ = 5;
    const int tid = threadIdx.x;
    const int cl = tid & 1;
    const int grp = tid >> 1;
    const unsigned lbase = cl * KAGG_IMG1;
    const char* winb = reinterpret_cast<const char*>(win);
    const int w = lpd_xcd_remap(blockIdx.x, gridDim.x);
    const int sl = w % nslices;
    const int b = w / nslices;
    const int col = sl * 8 + cl * 4;
    const unsigned row0 = (unsigned)b * g.N;
    const int N = g.N, C = g.C;
    const int passes = (N + GROUPS - 1) / GROUPS;
    const unsigned char* idxb = reinterpret_cast<const unsigned char*>(idx16);
    const float* Qc = g.Q + (size_t)row0 * g.ldq + col;
    const unsigned ldq = g.ldq;
    auto load = [&](CloudOps& o, int ps) {
        o.m = row0 + min(ps * GROUPS + grp, N - 1);
        kagg_load_idx(o, idxb, cl);
        o.q = *reinterpret_cast<const float4*>(Qc + (o.m - row0) * ldq);
    };
    const float4 gm = *reinterpret_cast<const float4*>(g.gamma + col);
    const float4 sg = make_float4(gm.x >= 0.f ? 1.f : -1.f, gm.y >= 0.f ? 1.f : -1.f, gm.z >= 0.f ? 1.f : -1.f, gm.w >= 0.f ? 1.f : -1.f);
    const float* Pc = g.P + (size_t)row0 * g.ldp + col;
    {   // the cloud's slice of P, sign-adjusted (one running max serves both selections): every row piece requested before the first store
        constexpr int NP = 4096 / GROUPS;
        float4 pr[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) pr[i] = *reinterpret_cast<const float4*>(Pc + (size_t)min(grp + i * GROUPS, N - 1) * g.ldp);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = grp + i * GROUPS;
            float4 p = pr[i];
            p.x *= sg.x; p.y *= sg.y; p.z *= sg.z; p.w *= sg.w;
            if (r < N) *reinterpret_cast<float4*>(const_cast<char*>(winb) + lbase + r * 16) = p;
        }
    }
    CloudOps A, Bo, Co;
    load(A, 0);
    load(Bo, 1);
    __syncthreads();
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float kf = 4.0f * KQ;
    auto process = [&](const CloudOps& o, bool valid) {
        float sp[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned ab[4] = {0, 0, 0, 0};
        auto edge = [&](unsigned off, unsigned t) {
            const float4 a = *reinterpret_cast<const float4*>(winb + (off + lbase));
            const float p[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sp[c] += p[c];
                sq[c] = fmaf(p[c], p[c], sq[c]);
                const bool take = p[c] > best[c];          // first extremum of the sign-adjusted values, like torch.max / torch.min
                best[c] = take ? p[c] : best[c];
                ab[c] = take ? t : ab[c];
            }
        };
        uint2 ix[5];
        kagg_quads(o, cl, ix);       // slot order: the first extremum decides the arg-max
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            edge(ix[i].x & 0xffffu, 4 * i);
            edge(ix[i].x >> 16, 4 * i + 1);
            edge(ix[i].y & 0xffffu, 4 * i + 2);
            edge(ix[i].y >> 16, 4 * i + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float sgn[4] = {sg.x, sg.y, sg.z, sg.w}, q[4] = {o.q.x, o.q.y, o.q.z, o.q.w};
        float S4[4], U4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            S4[c] = sgn[c] * sp[c];                       // exact: the sign flips commute with the rounding of the sums
            U4[c] = sgn[c] * best[c] + q[c];
        }
        const size_t off = (size_t)o.m * C + col;
        *reinterpret_cast<float4*>(g.S + off) = make_float4(S4[0], S4[1], S4[2], S4[3]);
        *reinterpret_cast<float4*>(g.usel + off) = make_float4(U4[0], U4[1], U4[2], U4[3]);
        *reinterpret_cast<uint32_t*>(g.arg + off) = ab[0] | (ab[1] << 8) | (ab[2] << 16) | (ab[3] << 24);
        if (valid) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s[c] += (double)S4[c] + (double)kf * q[c];
                ss[c] += (double)sq[c] + 2.0 * (double)q[c] * S4[c] + (double)kf * q[c] * q[c];
            }
        }
    };
    for (int ps = 0; ps < passes; ps += 3) {     // (a pass past the end of the cloud re-stores the last point with identical values)
        load(Co, ps + 2);
        process(A, ps * GROUPS + grp < N);
        load(A, ps + 3);
        if (ps + 1 < passes) process(Bo, (ps + 1) * GROUPS + grp < N);
        load(Bo, ps + 4);
        if (ps + 2 < passes) process(Co, (ps + 2) * GROUPS + grp < N);
    }
    // 8 sums per lane -> per (wave, cl) by shuffles over the 32 lanes of a parity -> per block through LDS -> the block's replica
    __syncthreads();                              // every gather is done: the images are free
    double* red = reinterpret_cast<double*>(win);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int o2 = 32; o2 > 1; o2 >>= 1) { s[c] += __shfl_xor(s[c], o2, 64); ss[c] += __shfl_xor(ss[c], o2, 64); }
    if ((tid & 62) == 0)                          // lanes 0 and 1 of each wave
#pragma unroll
        for (int c = 0; c < 4; ++c) { red[((tid >> 6) * 2 + cl) * 8 + c] = s[c]; red[((tid >> 6) * 2 + cl) * 8 + 4 + c] = ss[c]; }
    __syncthreads();
    if (tid < 16) {                               // tid = which * 8 + channel of the slice
        const int which = tid >> 3, ch = tid & 7;
        double t = 0.0;
        for (int wv = 0; wv < THREADS / 64; ++wv) t += red[(wv * 2 + (ch >> 2)) * 8 + which * 4 + (ch & 3)];
        atomicAdd((which ? g.sumsq : g.sum) + lpd_stat_rofs() + sl * 8 + ch, t);
    }
}

// Persistent form for 8-pass clouds (3584 < N <= 4096, the benchmark shape): a workgroup walks `per` consecutive
// (cloud, slice) items and requests the NEXT item's P rows while it works through the current one -- one float4 per
// thread and pass, parked in registers until the gathers of the current slice are done, then stored to LDS.  With one
// workgroup per CU (128 KiB of LDS) nothing else can hide the fill (ablation at B = 32, C = 256: 107 us; without the
// fill 80; without Q / out traffic 79; with no global memory traffic at all 55).  During its last item a workgroup
// keeps issuing the loads with a row stride of 0 (one 32-byte piece, no traffic): a branch around them would make the
// compiler's in-order vmcnt accounting wait for them early.
template <bool HAS_Q, bool SPLIT = false>
__global__ __launch_bounds__(1024) void edge_gather_max_cloud16p_kernel(GatherArgs g, const uint16_t* __restrict__ idx16,
                                                                        int nslices, float ns, int nwork, int per)
{
    extern __shared__ float4 win[];
    constexpr int GROUPS = 512, KQ = 5, NP = 8;
    const int tid = threadIdx.x;
    const int cl = tid & 1;
    const int grp = tid >> 1;
    const unsigned lbase = cl * KAGG_IMG1;
    char* winb = reinterpret_cast<char*>(win);
    const int vb = lpd_xcd_remap(blockIdx.x, gridDim.x);    // consecutive items (the slices of a cloud) stay on one XCD
    const int w_begin = vb * per;
    const int w_end = min(w_begin + per, nwork);
    if (w_begin >= w_end) return;
    const int N = g.N;
    const unsigned char* idxb = reinterpret_cast<const unsigned char*>(idx16);
    const unsigned ldq = g.ldq, ldo = g.ldo;
    auto p_slice_ptr = [&](int w) { return g.P + (long long)(w / nslices) * g.p_cloud + (long long)(w % nslices) * g.p_slice + cl * 4; };

    float4 pn[NP];
    {
        const float* Pc = p_slice_ptr(w_begin);
#pragma unroll
        for (int i = 0; i < NP; ++i) pn[i] = *reinterpret_cast<const float4*>(Pc + (size_t)min(grp + i * GROUPS, N - 1) * g.ldp);
    }
    for (int w = w_begin; w < w_end; ++w) {
        int grpw = grp;
        asm volatile("" : "+v"(grpw));   // keeps the row offsets from being hoisted out of the item loop (register pressure)
        const int sl = w % nslices;
        const int b = w / nslices;
        const int col = sl * 8 + cl * 4;
        const unsigned row0 = (unsigned)b * N;
        const float* Qc = g.Q + b * g.q_cloud + sl * g.q_slice + cl * 4;
        float* outc = SPLIT ? reinterpret_cast<float*>(reinterpret_cast<__bf16*>(g.out) + b * g.o_cloud + sl * g.o_slice)
                            : g.out + b * g.o_cloud + sl * g.o_slice + cl * 4;
        const float* Pn = p_slice_ptr(min(w + 1, w_end - 1));
        const unsigned ldpn = w + 1 < w_end ? (unsigned)g.ldp : 0u;   // last item: every lane re-reads one row piece (no traffic)
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.scale) sc = *reinterpret_cast<const float4*>(g.scale + col);
        if (g.shift) sh = *reinterpret_cast<const float4*>(g.shift + col);
        const float4 sg = make_float4(sc.x >= 0.f ? 1.f : -1.f, sc.y >= 0.f ? 1.f : -1.f, sc.z >= 0.f ? 1.f : -1.f,
                                      sc.w >= 0.f ? 1.f : -1.f);
        auto load = [&](CloudOps& o, int ps) {
            o.m = row0 + min(ps * GROUPS + grpw, N - 1);
            kagg_load_idx(o, idxb, cl);
            if (HAS_Q) o.q = *reinterpret_cast<const float4*>(Qc + (o.m - row0) * ldq);
        };
        CloudOps ops3[3];
        load(ops3[0], 0);
        load(ops3[1], 1);
        // this item's rows (requested during the previous item) -> LDS, sign-adjusted
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = grpw + i * GROUPS;
            float4 pv = pn[i];
            pv.x *= sg.x; pv.y *= sg.y; pv.z *= sg.z; pv.w *= sg.w;
            if (r < N) *reinterpret_cast<float4*>(winb + lbase + r * 16) = pv;
        }
        __syncthreads();
        auto process = [&](const CloudOps& o) {
            float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            // (the maximum is order-free: own piece, partner's piece, tail -- no selects)
            const uint2 ix[5] = {make_uint2(o.own.x, o.own.y), make_uint2(o.own.z, o.own.w),
                                 make_uint2(lpd_lane_xor1(o.own.x), lpd_lane_xor1(o.own.y)), make_uint2(lpd_lane_xor1(o.own.z), lpd_lane_xor1(o.own.w)), o.tail};
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(winb + ((ix[i].x & 0xffffu) + lbase));
                const float4 bq = *reinterpret_cast<const float4*>(winb + ((ix[i].x >> 16) + lbase));
                const float4 c = *reinterpret_cast<const float4*>(winb + ((ix[i].y & 0xffffu) + lbase));
                const float4 d = *reinterpret_cast<const float4*>(winb + ((ix[i].y >> 16) + lbase));
                v.x = fmaxf(fmaxf(v.x, a.x), bq.x); v.y = fmaxf(fmaxf(v.y, a.y), bq.y);
                v.z = fmaxf(fmaxf(v.z, a.z), bq.z); v.w = fmaxf(fmaxf(v.w, a.w), bq.w);
                v.x = fmaxf(fmaxf(v.x, c.x), d.x); v.y = fmaxf(fmaxf(v.y, c.y), d.y);
                v.z = fmaxf(fmaxf(v.z, c.z), d.z); v.w = fmaxf(fmaxf(v.w, c.w), d.w);
                __builtin_amdgcn_sched_barrier(0);
            }
            float4 r;
            r.x = sc.x * (sg.x * v.x + (HAS_Q ? o.q.x : 0.f)) + sh.x;
            r.y = sc.y * (sg.y * v.y + (HAS_Q ? o.q.y : 0.f)) + sh.y;
            r.z = sc.z * (sg.z * v.z + (HAS_Q ? o.q.z : 0.f)) + sh.z;
            r.w = sc.w * (sg.w * v.w + (HAS_Q ? o.q.w : 0.f)) + sh.w;
            r.x = fmaxf(r.x, 0.f) + ns * fminf(r.x, 0.f);
            r.y = fmaxf(r.y, 0.f) + ns * fminf(r.y, 0.f);
            r.z = fmaxf(r.z, 0.f) + ns * fminf(r.z, 0.f);
            r.w = fmaxf(r.w, 0.f) + ns * fminf(r.w, 0.f);
            kagg_store<SPLIT>(outc, (o.m - row0) * ldo, r, cl, g.o_lo);
        };
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            __builtin_amdgcn_sched_barrier(0);
            if (ps + 2 < NP) load(ops3[(ps + 2) % 3], ps + 2);
            pn[ps] = *reinterpret_cast<const float4*>(Pn + (size_t)((unsigned)min(grpw + ps * GROUPS, N - 1) * ldpn));
            __builtin_amdgcn_sched_barrier(0);
            process(ops3[ps % 3]);
        }
        __syncthreads();   // every gather of this slice is done: LDS may be refilled
    }
}

// int32 [M][20] -> uint16 (16 * index; indices < 4096), blocked for the kernels above (CloudOps): per 32 points 1280 bytes =
// [32 x 16 B: index quads 0, 1 | 32 x 16 B: quads 2, 3 | 32 x 8 B: quad 4].  One thread per (point, quad).
__global__ void pack_idx16_kernel(const int32_t* __restrict__ in, uint2* __restrict__ out, long long M)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = t / 5;
    const int i = (int)(t - m * 5);
    if (m >= M) return;
    const int4 v = *reinterpret_cast<const int4*>(in + m * 20 + i * 4);
    uint2 o;
    o.x = (uint32_t)((v.x << 4) & 0xffff) | ((uint32_t)v.y << 20);     // 16 * index: the byte offset of the row piece in LDS
    o.y = (uint32_t)((v.z << 4) & 0xffff) | ((uint32_t)v.w << 20);
    const long long blk = (m >> 5) * 160;                               // uint2 units: 1280 bytes per block
    const int p = (int)(m & 31);
    out[blk + (i < 4 ? (i >> 1) * 64 + p * 2 + (i & 1) : 128 + p)] = o;
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
    long long out_cloud;  // 0: out row-major [M][ldo]; else cloud-panel [cloud][.][panel_ld][8] with this many floats between clouds
    int panel_ld;         // rows allotted to one panel (>= N)
    long long out_lo;     // != 0 (split-bf16 kernel, cloud-panel out): `out` is the hi plane of a pair of bf16 planes, the lo plane
                          // out_lo elements behind it; out_cloud counts bf16 elements
    void* x1_hi;          // != null (split planes only): ALSO x1 = max over k of the stage-1 activation (util/lpdnet_model.py:249-250), as
                          // a second pair of planes with the layout of `out` (same cloud stride, panel rows and lo offset)
    int tiles_per_block = 1;   // edge_mlp_x3_kernel: consecutive point tiles a block walks with its weight fragments in registers
};

// m = m0 + (row inside the block); a block's 64 points lie in one cloud when the output is in cloud-panel form
// (the host checks N % 64 == 0), so the cloud index comes from m0 alone.
__device__ __forceinline__ float* edge_mlp_out(const EdgeMlpArgs& g, int m0, size_t m, int o)
{
    if (!g.out_cloud) return g.out + m * g.ldo + o;
    const size_t b = (size_t)(m0 / g.N);
    return g.out + b * g.out_cloud + ((size_t)(o >> 3) * g.panel_ld + (m - b * g.N)) * 8 + (o & 7);
}

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
            *edge_mlp_out(g, m0, (size_t)m, o) = lpd_act(sc * z + sh, g.act, g.slope);
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


// ------------------------------------------------------------------------------------------
// fused edge MLP on the bf16 MFMA (split-bf16, three products per term: see lpd_gemm.hip "bf16x3")
//
// Same contraction as edge_mlp_kernel, re-tiled for v_mfma_f32_32x32x16_bf16:
//   * W2 never touches LDS: each wave owns ONE 32-column tile of the output and keeps its W2 fragments (CM/16 k-steps,
//     hi and lo) in registers for the whole kernel, pre-multiplied by sgn(s2) so that a single running max serves the
//     max/min selection (negation is exact in every product and sum);
//   * the per-slot tile Y1_t [64 points][CM] is built in LDS as two bf16 images (hi, lo), rows k-contiguous with an
//     80/272-byte stride (conflict-free ds_read_b128 operand fetches), double-buffered: 70 KiB per block at CM = 128,
//     two blocks per CU;
//   * CM = CO = 128: wave w -> all 64 points x columns 32w..32w+31 (2 m-tiles);  CM = CO = 64: wave (wp, wo) ->
//     32 points x 32 columns.
// ------------------------------------------------------------------------------------------
typedef __bf16 em_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 em_bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void em_split4(float x0, float x1, float x2, float x3, em_bf16x4& hi, em_bf16x4& lo)
{
    hi[0] = (__bf16)x0; hi[1] = (__bf16)x1; hi[2] = (__bf16)x2; hi[3] = (__bf16)x3;
    lo[0] = (__bf16)(x0 - (float)hi[0]); lo[1] = (__bf16)(x1 - (float)hi[1]);
    lo[2] = (__bf16)(x2 - (float)hi[2]); lo[3] = (__bf16)(x3 - (float)hi[3]);
}

// The same split on packed pairs, spelled out: one v_cvt_pk_bf16_f32 per pair for hi, shift / mask back to fp32, one packed
// subtract, one v_cvt_pk for lo -- 2.5 instructions per element.  Written element-wise (above) around computed values the
// compiler converted every element on its own and re-packed with perm / alignbit / mov: 11 per element, which made the
// tile builder of the fused edge MLP issue twice the cycles of its MFMAs (750 VALU instructions per 48 MFMAs).
typedef __bf16 em_bf16x2 __attribute__((ext_vector_type(2)));
typedef float em_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void em_split4_packed(float x0, float x1, float x2, float x3, uint2& hi, uint2& lo)
{
    const em_bf16x2 h01 = __builtin_convertvector((em_f32x2){x0, x1}, em_bf16x2), h23 = __builtin_convertvector((em_f32x2){x2, x3}, em_bf16x2);
    const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
    const float r0 = x0 - __uint_as_float(u01 << 16), r1 = x1 - __uint_as_float(u01 & 0xffff0000u);
    const float r2 = x2 - __uint_as_float(u23 << 16), r3 = x3 - __uint_as_float(u23 & 0xffff0000u);
    const em_bf16x2 l01 = __builtin_convertvector((em_f32x2){r0, r1}, em_bf16x2), l23 = __builtin_convertvector((em_f32x2){r2, r3}, em_bf16x2);
    hi = make_uint2(u01, u23);
    lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}
// max(a, b) as v_med3_f32(a, b, +inf): one instruction, without the canonicalising self-max the compiler puts in front of
// fmaxf on values it cannot prove quiet (MFMA results, loaded data).  (An inline-asm v_max_f32 is NOT an option on an MFMA
// result: the hazard recogniser does not see inside the asm and the read came too early -- wrong maxima.)
// The +inf operand has to come from a register the optimiser cannot see through (em_opaque_inf): with the literal, this
// compiler folds med3(a, b, +inf) back to maxnum and re-adds the two canonicalising self-maxes -- three v_max_f32 per element.
__device__ __forceinline__ float em_opaque_inf()
{
    float v = INFINITY;
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float em_vmax(float a, float b, float pinf) { return __builtin_amdgcn_fmed3f(a, b, pinf); }

template <int CM, int CO, int PTS = EM_PTS>
struct EdgeMlpX3Cfg {
    static constexpr int LDK = CM + 8;                      // bf16 per LDS row
    static constexpr int IMG = PTS * LDK;                   // bf16 per image (hi or lo)
    static constexpr int WM = (CO == 128 ? 2 : 1) * PTS / EM_PTS;   // m-tiles (32 points) per wave
    static constexpr int KS = CM / 16;                      // k-steps
    static constexpr int F4_PER_ROW = CM / 4;
    static constexpr int ROWS_PER_PASS = EM_THREADS / F4_PER_ROW;
    static constexpr int PASSES = PTS / ROWS_PER_PASS;
    static_assert(WM >= 1 && PASSES >= 1, "32-point blocks are built for 128 -> 128");
};

// X1 (round 6; 128 -> 128, split planes, 32 points per block): the DG1-stage K-agg rides along.  x1 = max over k of the stage-1
// activation act(s1 (P_j + Q_i) + b1) is the maximum over the slots of exactly the values the tile builder produces for the MFMA
// (the activation is monotone: act(max ya) = max act(ya)), so every builder thread keeps a running maximum of its (row, 4 channels)
// elements and stores them at the end -- the separate lpd_edge_gather_max16 launch over the same graph (80 us at 32 clouds: 512 items
// on 256 CUs, two rounds of exposed 128-KiB fills; 0.33 of its HBM roofline) and its uint16 index packing disappear.  The 32 extra
// accumulators per thread do not fit beside the 64-point block's state (239 of 256 registers), so the fused form takes 32 points per
// block: half the builder state per thread, one m-tile per wave.
template <int CM, int CO, int PTS = EM_PTS, bool X1 = false>
__global__ __launch_bounds__(EM_THREADS, 2) void edge_mlp_x3_kernel(EdgeMlpArgs g)      // (three 32-point blocks per CU: 168 registers with 9 spilled, 285 us against 288)
{
    using Cfg = EdgeMlpX3Cfg<CM, CO, PTS>;
    constexpr int LDK = Cfg::LDK, IMG = Cfg::IMG, WM = Cfg::WM, KS = Cfg::KS, PASSES = Cfg::PASSES;
    static_assert((CM == 128 && CO == 128) || (CM == 64 && CO == 64), "edge_mlp_x3: built for 128->128 and 64->64");
    static_assert(!X1 || (CM == 128 && CO == 128), "the x1 output rides with the 128 -> 128 stage");
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];   // [2 buffers][hi | lo][64][LDK], then idx
    int* idxs = reinterpret_cast<int*>(smem16 + 4 * IMG);             // [PTS][k]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int col = lane & 31;
    const int ntile = CO == 128 ? wave : (wave >> 1);       // output-column tile of this wave
    const int ptile = CO == 128 ? 0 : (wave & 1);           // first point tile of this wave
    const int tile0 = lpd_xcd_remap(blockIdx.x, gridDim.x) * g.tiles_per_block;      // consecutive tiles: one cloud, one XCD

    // W2 fragments of this wave's column n = ntile*32 + col: k = 16 s + 8 h .. +7, sign-adjusted, split
    const int n = ntile * 32 + col;
    const float sc2 = g.s2[n], sh2 = g.b2[n];
    const float sgn = sc2 >= 0.f ? 1.0f : -1.0f;
    em_bf16x8 b_hi[KS], b_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float4 w0 = *reinterpret_cast<const float4*>(g.W2 + (size_t)n * CM + s * 16 + h * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(g.W2 + (size_t)n * CM + s * 16 + h * 8 + 4);
        em_bf16x4 h0, l0, h1, l1;
        em_split4(sgn * w0.x, sgn * w0.y, sgn * w0.z, sgn * w0.w, h0, l0);
        em_split4(sgn * w1.x, sgn * w1.y, sgn * w1.z, sgn * w1.w, h1, l1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { b_hi[s][e] = h0[e]; b_hi[s][4 + e] = h1[e]; b_lo[s][e] = l0[e]; b_lo[s][4 + e] = l1[e]; }
    }

    // Round 6: a block walks `tiles_per_block` consecutive tiles with its W2 fragments in registers (one tile per block re-read and
    // re-split the 64-KiB weight in front of every 32-point tile: 4096 times per launch at 32 clouds)
    for (int rep = 0; rep < g.tiles_per_block; ++rep) {
    const int m0 = (tile0 + rep) * PTS;
    if (m0 >= g.M) break;      // uniform
    for (int f = tid; f < PTS * g.k; f += EM_THREADS) {
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
    const float ns = lpd_neg_slope(g.act, g.slope);
    const float pinf = em_opaque_inf();
    const em_f32x2 s1a = {s1.x, s1.y}, s1b = {s1.z, s1.w}, ns2 = {ns, ns};
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
    __syncthreads();  // idxs visible

    float4 xm[X1 ? PASSES : 1];
#pragma unroll
    for (int e = 0; e < (X1 ? PASSES : 1); ++e) xm[e] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 pg[PASSES];
    auto gather = [&](int t) {
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const int p = prow + Cfg::ROWS_PER_PASS * e;
            const int row = idxs[p * g.k + t];
            pg[e] = *reinterpret_cast<const float4*>(g.P + (size_t)row * g.ldp + c4 * 4);
        }
    };
    auto build = [&](int buf) {
        __bf16* hi_img = smem16 + buf * 2 * IMG;
        __bf16* lo_img = hi_img + IMG;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const int p = prow + Cfg::ROWS_PER_PASS * e;
            // s (P + Q) + b  ==  s P + (s Q + b) as one fma; LeakyReLU / ReLU / identity as max(v, ns v) (0 <= ns <= 1)
            // (packed pairs: v_pk_fma_f32 / v_pk_mul_f32 round each element exactly as the scalar forms do)
            const em_f32x2 ya = __builtin_elementwise_fma(s1a, (em_f32x2){pg[e].x, pg[e].y}, (em_f32x2){qc[e].x, qc[e].y});
            const em_f32x2 yb = __builtin_elementwise_fma(s1b, (em_f32x2){pg[e].z, pg[e].w}, (em_f32x2){qc[e].z, qc[e].w});
            if constexpr (X1) {      // running maximum of the pre-activation values (the activation is applied once, at the end)
                xm[e].x = em_vmax(xm[e].x, ya[0], pinf); xm[e].y = em_vmax(xm[e].y, ya[1], pinf);
                xm[e].z = em_vmax(xm[e].z, yb[0], pinf); xm[e].w = em_vmax(xm[e].w, yb[1], pinf);
            }
            const em_f32x2 na = ns2 * ya, nb = ns2 * yb;
            uint2 hh, ll;
            em_split4_packed(em_vmax(ya[0], na[0], pinf), em_vmax(ya[1], na[1], pinf), em_vmax(yb[0], nb[0], pinf),
                             em_vmax(yb[1], nb[1], pinf), hh, ll);
            *reinterpret_cast<uint2*>(hi_img + p * LDK + c4 * 4) = hh;
            *reinterpret_cast<uint2*>(lo_img + p * LDK + c4 * 4) = ll;
        }
    };

    f32x16 zmax[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) zmax[i][r] = -INFINITY;

    gather(0);
    build(0);
    __syncthreads();

    for (int t = 0; t < g.k; ++t) {
        const int buf = t & 1;
        if (t + 1 < g.k) gather(t + 1);  // rows in flight under this slot's MFMAs
        const __bf16* ah = smem16 + buf * 2 * IMG + (ptile * 32 + col) * LDK + h * 8;
        const __bf16* al = ah + IMG;
        f32x16 acc[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const em_bf16x8 a_hi = *reinterpret_cast<const em_bf16x8*>(ah + i * 32 * LDK + s * 16);
                const em_bf16x8 a_lo = *reinterpret_cast<const em_bf16x8*>(al + i * 32 * LDK + s * 16);
                // accumulator rows = points, columns = output channels: A = Y tile, B = W2 fragment
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi[s], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo[s], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi[s], acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) zmax[i][r] = em_vmax(zmax[i][r], acc[i][r], pinf);
        if (t + 1 < g.k) build(buf ^ 1);
        __syncthreads();
    }

    if constexpr (X1) {      // x1 rows of this thread: 4 channels of PASSES points, activated, split, 8 bytes into each plane
        const size_t bc = (size_t)(m0 / g.N);
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const int m = m0 + prow + Cfg::ROWS_PER_PASS * e;
            if (m >= g.M) continue;
            uint2 hh, ll;
            em_split4_packed(em_vmax(xm[e].x, ns * xm[e].x, pinf), em_vmax(xm[e].y, ns * xm[e].y, pinf), em_vmax(xm[e].z, ns * xm[e].z, pinf),
                             em_vmax(xm[e].w, ns * xm[e].w, pinf), hh, ll);
            __bf16* dst = reinterpret_cast<__bf16*>(g.x1_hi) + bc * g.out_cloud + ((size_t)(c4 >> 1) * g.panel_ld + ((size_t)m - bc * g.N)) * 8 + (c4 & 1) * 4;
            *reinterpret_cast<uint2*>(dst) = hh;
            *reinterpret_cast<uint2*>(dst + g.out_lo) = ll;
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (ptile + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= g.M) continue;
            const float v = lpd_act_pl(sc2 * (sgn * zmax[i][r]) + sh2, ns);
            if (!g.out_lo) { *edge_mlp_out(g, m0, (size_t)m, n) = v; continue; }      // uniform
            // split planes: this lane holds channel n of point m; the even lane of a pair stores (hi_n, hi_n+1) into the hi plane,
            // the odd lane (lo_n-1, lo_n) into the lo plane -- one 4-byte store per lane, as in the fp32 form
            const __bf16 hb = (__bf16)v;
            const __bf16 lb = (__bf16)(v - (float)hb);
            const unsigned hu = __builtin_bit_cast(unsigned short, hb), lu = __builtin_bit_cast(unsigned short, lb);
            const int odd = n & 1;
            const unsigned got = lpd_lane_xor1(odd ? hu : lu);
            const unsigned word = odd ? (got | (lu << 16)) : (hu | (got << 16));
            const size_t b = (size_t)(m0 / g.N);
            __bf16* dst = reinterpret_cast<__bf16*>(g.out) + (odd ? g.out_lo : 0) + b * g.out_cloud +
                          ((size_t)(n >> 3) * g.panel_ld + ((size_t)m - b * g.N)) * 8 + ((n & 7) & ~1);
            *reinterpret_cast<unsigned*>(dst) = word;
        }
    }      // tiles of this block
}

// ------------------------------------------------------------------------------------------
// TRAIN-mode DG1 -> DG2 stage in one launch (util/lpdnet_model.py:249-252 with batch statistics): the organisation of
// edge_mlp_x3_kernel<128, 128> with the tensors the backward pass needs written on the way --
//   Y1e[(i,t)][c] = act(s1 (P[nbr(i,t)] + Q[i]) + b1)          (s1, b1: BatchNorm1 scale / shift of THIS batch, from the split-form
//                                                               statistics pass lpd_edge_split_fwd: the raw edge tensor U is never written)
//   Z = Y1e W2^T (raw)                                          stored from the accumulators, row (i, t)
//   sum Z, sum Z^2 per channel (fp32 per lane over the block's 64 x k rows, fp64 across blocks through the replicas)
//   zsel[i][c] = sel_t Z[(i,t)][c], arg2[i][c] = first t that attains it (sel = max where gamma2[c] >= 0, min otherwise: the
//   selection needs the SIGN of the BatchNorm2 scale only, which is known before the statistics)
// BF16 = bf16 storage (autograd.set_train_storage("bf16")): Y1e is stored as bf16 and the product takes the ROUNDED Y1e as its
// operand (a_hi only) against the split weight (two MFMA products, as lpd_gemm_bf16s); fp32 storage: three split-bf16 products.
// ZBF16: Z is stored as bf16 -- in BOTH storage modes by default: its statistics and its selection are taken from the fp32
// accumulators here, and the only later reader of the stored Z is the backward's xhat2 m2 term (the dense, mean-sized part of
// BatchNorm2's backward, 1e-3 of the gradient: a 2^-9 rounding of it is 4e-6), so fp32 rows would be 1.85 GB written and read
// back for nothing.
// x1 = max_t Y1e and its arg-max come from the statistics pass (the activation is monotone), not from here: the running maximum
// of 64 points x 128 channels would not fit the register budget of two workgroups per CU.
// ------------------------------------------------------------------------------------------
struct EdgeMlpTrainArgs {
    const float* P; const float* Q; const int32_t* idx;
    const float* s1; const float* b1; const float* W2; const float* gamma2;
    void* Y; void* Z;            // [M k][128] bf16 or fp32
    float* zsel; uint8_t* arg2;  // [M][ldsel], [M][128]
    double* sum; double* sumsq;  // replicas (lpd_common.h)
    int M, N, k, ldp, ldq, ldsel, act;
    float slope;
};

typedef unsigned em_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned em_u32x4 __attribute__((ext_vector_type(4)));

// ZST = false: Z is not stored at all (its conversion, lane exchange and store instructions leave the epilogue: the kernel is bound by them)
template <bool BF16, bool ZBF16, bool ZST = true>
__global__ __launch_bounds__(EM_THREADS, 2) void edge_mlp_train_kernel(EdgeMlpTrainArgs g)
{
    constexpr int CM = 128;
    using Cfg = EdgeMlpX3Cfg<CM, CM>;
    constexpr int LDK = Cfg::LDK, IMG = Cfg::IMG, KS = Cfg::KS, PASSES = Cfg::PASSES, NIMG = BF16 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];   // [2 buffers][hi (| lo)][64][LDK], then idx
    int* idxs = reinterpret_cast<int*>(smem16 + 2 * NIMG * IMG);      // [EM_PTS][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int m0 = lpd_xcd_remap(blockIdx.x, gridDim.x) * EM_PTS;    // M % 64 == 0 (host check)
    const int n = wave * 32 + col;                                   // this lane's output channel
    const float sgn = g.gamma2[n] >= 0.f ? 1.0f : -1.0f;
    em_bf16x8 b_hi[KS], b_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float4 w0 = *reinterpret_cast<const float4*>(g.W2 + (size_t)n * CM + s * 16 + h * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(g.W2 + (size_t)n * CM + s * 16 + h * 8 + 4);
        em_bf16x4 h0, l0, h1, l1;
        em_split4(sgn * w0.x, sgn * w0.y, sgn * w0.z, sgn * w0.w, h0, l0);
        em_split4(sgn * w1.x, sgn * w1.y, sgn * w1.z, sgn * w1.w, h1, l1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { b_hi[s][e] = h0[e]; b_hi[s][4 + e] = h1[e]; b_lo[s][e] = l0[e]; b_lo[s][4 + e] = l1[e]; }
    }
    for (int f = tid; f < EM_PTS * g.k; f += EM_THREADS) {
        const int p = f / g.k, t = f - p * g.k;
        const int m = m0 + p;
        idxs[f] = (m / g.N) * g.N + g.idx[(size_t)m * g.k + t];
    }
    const int c4 = tid % Cfg::F4_PER_ROW, prow = tid / Cfg::F4_PER_ROW;
    const float4 s1 = *reinterpret_cast<const float4*>(g.s1 + c4 * 4);
    const float4 b1 = *reinterpret_cast<const float4*>(g.b1 + c4 * 4);
    const float ns = lpd_neg_slope(g.act, g.slope);
    const float pinf = em_opaque_inf();
    const em_f32x2 s1a = {s1.x, s1.y}, s1b = {s1.z, s1.w}, ns2 = {ns, ns};
    float4 qc[PASSES];
#pragma unroll
    for (int e = 0; e < PASSES; ++e) {
        const int m = m0 + prow + Cfg::ROWS_PER_PASS * e;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.Q) q = *reinterpret_cast<const float4*>(g.Q + (size_t)m * g.ldq + c4 * 4);
        qc[e] = make_float4(s1.x * q.x + b1.x, s1.y * q.y + b1.y, s1.z * q.z + b1.z, s1.w * q.w + b1.w);
    }
    __syncthreads();

    float4 pg[PASSES];
    auto gather = [&](int t) {
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const int row = idxs[(prow + Cfg::ROWS_PER_PASS * e) * g.k + t];
            pg[e] = *reinterpret_cast<const float4*>(g.P + (size_t)row * g.ldp + c4 * 4);
        }
    };
    constexpr int ES = BF16 ? 2 : 4, ESZ = ZBF16 ? 2 : 4;                  // bytes per stored element of Y1e / of Z
    // The block's slabs of Y1e and Z (64 k rows) as buffer resources: a store is (resource, this thread's constant 32-bit offset,
    // scalar offset of the row) -- with flat pointers the compiler kept a 64-bit address register per store of the unrolled
    // epilogue (85-100 spilled registers)
    // (Z == null: a resource of ZERO records over Y -- out-of-range buffer stores are dropped by the hardware, so the epilogue's stores
    //  stay in the instruction stream and move nothing: the backward then works without Z, lpd_edge_mlp_train_bwd)
    const unsigned slab = (unsigned)EM_PTS * (unsigned)g.k * (CM * ES), slabz = g.Z ? (unsigned)EM_PTS * (unsigned)g.k * (CM * ESZ) : 0u;
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char*>(g.Y) + (size_t)m0 * g.k * (CM * ES), 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc(
        g.Z ? reinterpret_cast<unsigned char*>(g.Z) + (size_t)m0 * g.k * (CM * ESZ) : reinterpret_cast<unsigned char*>(g.Y), 0, slabz, 0x00020000);
    const unsigned rowb = (unsigned)g.k * (CM * ES), rowbz = (unsigned)g.k * (CM * ESZ);      // bytes between consecutive POINTS' rows of one slot
    const unsigned yoff = ((unsigned)prow * (unsigned)g.k * CM + c4 * 4) * ES;       // M k 128 ES < 2^32 (host check)
    const unsigned zoff = ((unsigned)(4 * h) * (unsigned)g.k * CM + (ZBF16 ? (n & ~1) : n)) * ESZ;
    // slot t's tile into LDS buffer `buf` AND its rows (i, t) into Y1e: 32 threads (c4) cover one 256- / 512-byte row
    auto build = [&](int buf, int t) {
        __bf16* hi_img = smem16 + buf * NIMG * IMG;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const int p = prow + Cfg::ROWS_PER_PASS * e;
            const em_f32x2 ya = __builtin_elementwise_fma(s1a, (em_f32x2){pg[e].x, pg[e].y}, (em_f32x2){qc[e].x, qc[e].y});
            const em_f32x2 yb = __builtin_elementwise_fma(s1b, (em_f32x2){pg[e].z, pg[e].w}, (em_f32x2){qc[e].z, qc[e].w});
            const em_f32x2 na = ns2 * ya, nb = ns2 * yb;
            const float y0 = em_vmax(ya[0], na[0], pinf), y1 = em_vmax(ya[1], na[1], pinf), y2 = em_vmax(yb[0], nb[0], pinf),
                        y3 = em_vmax(yb[1], nb[1], pinf);
            // uniform base (scalar registers) + this thread's constant 32-bit offset: no 64-bit address register per store
            const unsigned ysoff = (unsigned)(Cfg::ROWS_PER_PASS * e) * rowb + (unsigned)t * (CM * ES);      // uniform
            uint2 hh, ll;
            em_split4_packed(y0, y1, y2, y3, hh, ll);
            *reinterpret_cast<uint2*>(hi_img + p * LDK + c4 * 4) = hh;
            if constexpr (BF16) {
                __builtin_amdgcn_raw_buffer_store_b64((em_u32x2){hh.x, hh.y}, yres, yoff, ysoff, 0);
            } else {
                *reinterpret_cast<uint2*>(hi_img + IMG + p * LDK + c4 * 4) = ll;
                __builtin_amdgcn_raw_buffer_store_b128((em_u32x4){__float_as_uint(y0), __float_as_uint(y1), __float_as_uint(y2), __float_as_uint(y3)},
                                                       yres, yoff, ysoff, 0);
            }
        }
    };

    f32x16 zmax[2];
    uint32_t arg2w[2][4];        // byte (r & 3) of word r >> 2: slot of the running extremum of accumulator element r
    float ssum = 0.0f, ssq = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) zmax[i][r] = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) arg2w[i][w] = 0u;
    }
    gather(0);
    build(0, 0);
    __syncthreads();

    for (int t = 0; t < g.k; ++t) {
        const int buf = t & 1;
        if (t + 1 < g.k) gather(t + 1);
        const __bf16* ah = smem16 + buf * NIMG * IMG + col * LDK + h * 8;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const em_bf16x8 a_hi = *reinterpret_cast<const em_bf16x8*>(ah + i * 32 * LDK + s * 16);
                if constexpr (!BF16) {
                    const em_bf16x8 a_lo = *reinterpret_cast<const em_bf16x8*>(ah + IMG + i * 32 * LDK + s * 16);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi[s], acc[i], 0, 0, 0);
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo[s], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi[s], acc[i], 0, 0, 0);
            }
        }
        // acc = sgn * Z tile: rows = points (i, r -> i * 32 + (r & 3) + 8 (r >> 2) + 4 h), column = this lane's channel n
        const uint32_t tb = (uint32_t)t;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][r];
                const unsigned zsoff = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * rowbz + (unsigned)t * (CM * ESZ);      // uniform
                const float z = sgn * v;
                if constexpr (!ZST) {
                } else if constexpr (ZBF16) {
                    // the stored value is the rounded one; statistics and selection are taken of the fp32 value (as lpd_gemm_bf16s +
                    // lpd_group_sel_stats_bf16 take them of the stored one: the difference is the 2^-9 the storage mode states)
                    const __bf16 zb = (__bf16)z;
                    const unsigned mine = __builtin_bit_cast(unsigned short, zb);
                    const unsigned other = lpd_lane_xor1(mine);
                    if (!(col & 1)) __builtin_amdgcn_raw_buffer_store_b32(mine | (other << 16), zres, zoff, zsoff, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z), zres, zoff, zsoff, 0);
                }
                ssum += v;
                ssq = fmaf(v, v, ssq);
                const bool gt = v > zmax[i][r];
                zmax[i][r] = em_vmax(zmax[i][r], v, pinf);
                const uint32_t sh8 = 8u * (r & 3);
                const uint32_t cand = (arg2w[i][r >> 2] & ~(0xffu << sh8)) | (tb << sh8);
                arg2w[i][r >> 2] = gt ? cand : arg2w[i][r >> 2];
            }
        if (t + 1 < g.k) build(buf ^ 1, t + 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            g.zsel[(size_t)m * g.ldsel + n] = sgn * zmax[i][r];
            g.arg2[(size_t)m * CM + n] = (uint8_t)((arg2w[i][r >> 2] >> (8 * (r & 3))) & 0xffu);
        }
    // column statistics of Z: the two half-waves hold disjoint rows of the same channel
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    if (h == 0) {
        atomicAdd(&g.sum[lpd_stat_rofs() + n], (double)(sgn * ssum));
        atomicAdd(&g.sumsq[lpd_stat_rofs() + n], (double)ssq);
    }
}

// ------------------------------------------------------------------------------------------
// Backward of the train-mode DG1 -> DG2 stage, dense part, in one launch (the counterpart of edge_mlp_train_kernel):
//   dZ[(i,t)][c]  = s2_c (delta_{t,arg2[i][c]} dpre2[i][c] - m1_c - xhat2 m2_c)      generated while the A tile is built, from the stored
//                   Z: A = z a1 + a0 + delta dpre2 (a1 = -invstd2 m2, a0 = mu2 invstd2 m2 - m1), the factor s2_c folded into the weight
//                   fragments (cf. lpd_gemm_bf16s_bnbwd, csrc/lpd_train3.hip);
//   dY1e = dZ W2  on the MFMA (rows = 32 points of one neighbour slot, a wave owns 32 of the 128 columns);
//   G[(i,t)][n]   = (dY1e + delta_{t,arg1[i][n]} dx1[i][n]) act'(pre1)      the gradient in front of BatchNorm1, written INSTEAD of dY1e;
//   sum G, sum G xhat1 per channel (BatchNorm1's dbeta / dgamma) and gsum[i][n] = sum_t G[(i,t)][n] (the centre-term sums) on the way,
//   act' and xhat1 = (pre1 - beta1) / gamma1 from the stored post-activation Y1e (pre1 = y or y / ns).
// The former chain wrote dY1e, read it twice with U1 (reduce, apply), wrote dU1 and gathered it; what follows this kernel is ONE
// gather pass over the transposed graph with closed-form sums (edge_dense_bwd_apply_kernel, lpd_train2.hip).
// ------------------------------------------------------------------------------------------
struct EdgeMlpBwdArgs {
    const void* Z; const uint8_t* arg2; const void* dpre2;      // [E][128] bf16 / fp32; [M][128]; [M][128] bf16 / fp32
    const float* W2; const float* scale2; const float* mean2; const float* invstd2;
    const double* dbeta2; const double* dgamma2;                // sums over the E edges (lpd_bn_sel_bwd_reduce)
    const void* Y; const uint8_t* arg1; const float* dx1; int lddx1;
    const float* beta1; const float* rgamma1;                   // BatchNorm1 bias, 1 / weight (0 where the weight is 0)
    void* G; float* gsum;                                       // [E][128] bf16 / fp32; [M][128]
    double* sum; double* sumx;                                  // replicas
    int M, k, act;
    float slope, inv_ns;
    const uint16_t* Kb; const float* cvec;                      // NOZ: K as bf16 [n][m], constants [128] (edge_mlp_bwd_prep_kernel)
    int tiles_per_block = 1;
};

constexpr int EB_PTS = 32;

// Without Z (NOZ).  dY1e = dZ (S2 W2) with dZ_c = delta dpre_c + a0_c + a1_c Z_c, and Z = Y1e W2^T, so the dense part is a product with
// Y1e itself:   dY1e[e][n] = sum_c delta dpre[c] s2_c W2[c][n]  +  sum_m Y1e[e][m] K[m][n]  +  cvec[n],
//     K[m][n] = sum_c a1_c s2_c W2[c][m] W2[c][n],   cvec[n] = sum_c a0_c s2_c W2[c][n]          (a1 = -invstd2 m2, a0 = mean2 invstd2 m2 - m1).
// The forward need not store Z (0.92 GB at configs[2]) and this kernel reads Y1e once, as whole rows for the operand image, instead
// of Z rows plus sixteen 2- / 4-byte elements of Y1e per lane and slot.  The K term is the mean-sized part of BatchNorm2's backward
// (~1e-3 of the gradient in the trained net): ONE bf16 product (the stored Z it replaces carried a 2^-9 too, but of the SUM: on
// random data the product of rounded operands is ~6x further off, 2e-3 of the gradient -- fine at the bf16 mode's tolerances, not at
// the fp32 mode's: built for bf16 storage only).  K sits in LDS as an [n][m] image (its fragments in registers: 68 spilled registers).
__global__ __launch_bounds__(128) void edge_mlp_bwd_prep_kernel(const float* __restrict__ W2, const float* __restrict__ scale2,
                                                                const float* __restrict__ mean2, const float* __restrict__ invstd2,
                                                                const double* __restrict__ dbeta2, const double* __restrict__ dgamma2,
                                                                double count, uint16_t* __restrict__ Kb, float* __restrict__ cvec)
{
    // Kb [n][m] bf16: row n = the 128 contraction values of output column n (what a lane of the backward kernel reads as 16-byte pieces)
    __shared__ float a1s[128], a0s[128], wn[128];
    const int n = blockIdx.x, t = threadIdx.x;
    {
        const float m1 = (float)(dbeta2[t] / count), m2 = (float)(dgamma2[t] / count);
        a1s[t] = -invstd2[t] * m2;
        a0s[t] = mean2[t] * invstd2[t] * m2 - m1;
        wn[t] = scale2[t] * W2[(size_t)t * 128 + n];
    }
    __syncthreads();
    float acc = 0.0f, cv = 0.0f;
    for (int c = 0; c < 128; ++c) {
        acc = fmaf(a1s[c] * wn[c], W2[(size_t)c * 128 + t], acc);
        cv = fmaf(a0s[c], wn[c], cv);
    }
    Kb[(size_t)n * 128 + t] = __builtin_bit_cast(unsigned short, (__bf16)acc);
    if (t == 0) cvec[n] = cv;
}

template <bool BF16, bool ZBF16, bool NOZ = false>
__global__ __launch_bounds__(EM_THREADS, 2) void edge_mlp_train_bwd_kernel(EdgeMlpBwdArgs g)
{
    constexpr int CM = 128, LDK = CM + 8, IMG = EB_PTS * LDK, KS = CM / 16, NIMG = BF16 ? 1 : 2, ES = BF16 ? 2 : 4, ESZ = ZBF16 ? 2 : 4;
    constexpr int NB = NIMG + (NOZ ? 1 : 0), YI = NIMG * IMG;      // images per buffer: dZ hi (| lo) (| Y1e hi); offset of the Y1e image
    static_assert(!NOZ || BF16, "the form without Z is built for bf16 storage");
    constexpr int KIMG = 2 * NB * IMG;                             // NOZ: the K image [128 n][LDK] behind the two buffers
    constexpr int RPP = EM_THREADS / (CM / 4), PASSES = EB_PTS / RPP;      // 8 rows per pass, 4 passes
    extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];       // [2 buffers][hi (| lo) (| Y1e hi)][32][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int tile0 = lpd_xcd_remap(blockIdx.x, gridDim.x) * g.tiles_per_block;      // consecutive 32-point tiles; M % 32 == 0 (host check)
    const int n = wave * 32 + col;                                        // this lane's column of dY1e / G
    const double count = (double)g.M * (double)g.k;
    // NOZ: K into LDS ([n][m] bf16, 16-byte pieces), this column's constant
    float cvn = 0.0f;
    if constexpr (NOZ) {
        for (int e = tid; e < CM * CM / 8; e += EM_THREADS) {
            const int kn = e >> 4, kp = e & 15;
            *reinterpret_cast<uint4*>(smem16 + KIMG + kn * LDK + kp * 8) = *reinterpret_cast<const uint4*>(g.Kb + (size_t)kn * CM + kp * 8);
        }
        cvn = g.cvec[n];
    }
    const __bf16* kfrag = smem16 + KIMG + n * LDK + h * 8;        // + 16 s

    // weight fragments: B[k = c][n] = s2_c W2[c][n], c = 16 s + 8 h + e
    em_bf16x8 b_hi[KS], b_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = s * 16 + h * 8 + e;
            w[e] = g.scale2[c] * g.W2[(size_t)c * CM + n];
        }
        em_bf16x4 h0, l0, h1, l1;
        em_split4(w[0], w[1], w[2], w[3], h0, l0);
        em_split4(w[4], w[5], w[6], w[7], h1, l1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { b_hi[s][e] = h0[e]; b_hi[s][4 + e] = h1[e]; b_lo[s][e] = l0[e]; b_lo[s][4 + e] = l1[e]; }
    }
    // builder role: channel quad c4 of rows prow + 8 e
    const int c4 = tid % (CM / 4), prow = tid / (CM / 4);
    float a1[4], a0[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = c4 * 4 + c;
        const float m1 = (float)(g.dbeta2[ch] / count), m2 = (float)(g.dgamma2[ch] / count);
        a1[c] = NOZ ? 0.0f : -g.invstd2[ch] * m2;
        a0[c] = NOZ ? 0.0f : g.mean2[ch] * g.invstd2[ch] * m2 - m1;
    }
    const float beta1 = g.beta1[n], rg1 = g.rgamma1[n];
    const float ns = lpd_neg_slope(g.act, g.slope);
    float sg = 0.0f, sgx = 0.0f;
    // Round 6: a block walks `tiles_per_block` consecutive tiles with its weight fragments in registers, the K image in LDS and its two
    // reduction sums in registers: the 34-KiB K image, the 64-KiB weight and the atomics once per block instead of once per 32 points
    for (int rep = 0; rep < g.tiles_per_block; ++rep) {
    const int m0 = (tile0 + rep) * EB_PTS;
    if (m0 >= g.M) break;      // uniform
    float dp2[PASSES][4];
    uint32_t ar2[PASSES];
#pragma unroll
    for (int e = 0; e < PASSES; ++e) {
        const size_t m = (size_t)(m0 + prow + RPP * e);
        ar2[e] = *reinterpret_cast<const uint32_t*>(g.arg2 + m * CM + c4 * 4);
        if constexpr (BF16) {
            const uint2 d = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(g.dpre2) + m * CM + c4 * 4);
            dp2[e][0] = __uint_as_float(d.x << 16); dp2[e][1] = __uint_as_float(d.x & 0xffff0000u);
            dp2[e][2] = __uint_as_float(d.y << 16); dp2[e][3] = __uint_as_float(d.y & 0xffff0000u);
        } else {
            const float4 d = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.dpre2) + m * CM + c4 * 4);
            dp2[e][0] = d.x; dp2[e][1] = d.y; dp2[e][2] = d.z; dp2[e][3] = d.w;
        }
    }
    // the block's slabs of Z, Y1e and G (32 k rows) as buffer resources (see edge_mlp_train_kernel)
    const unsigned slab = (unsigned)EB_PTS * (unsigned)g.k * (CM * ES);
    const size_t slab0 = (size_t)m0 * g.k * (CM * ES);
    const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(NOZ ? g.Y : g.Z)) + (size_t)m0 * g.k * (CM * ESZ), 0,
        NOZ ? 0u : (unsigned)EB_PTS * (unsigned)g.k * (CM * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.Y)) + slab0, 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(g.G) + slab0, 0, slab, 0x00020000);
    const unsigned rowb = (unsigned)g.k * (CM * ES), rowbz = (unsigned)g.k * (CM * ESZ);
    const unsigned zoff = ((unsigned)prow * (unsigned)g.k * CM + c4 * 4) * ESZ;                   // builder: its rows of Z
    const unsigned yqoff = ((unsigned)prow * (unsigned)g.k * CM + c4 * 4) * ES;                  // builder (NOZ): its rows of Y1e
    const unsigned eoff = ((unsigned)(4 * h) * (unsigned)g.k * CM + n) * ES;                      // epilogue: element (row 4 h + ., column n)
    const unsigned goff = ((unsigned)(4 * h) * (unsigned)g.k * CM + (BF16 ? (n & ~1) : n)) * ES;

    // epilogue role: rows p = (r & 3) + 8 (r >> 2) + 4 h, column n
    float dx1v[16];
    uint32_t ar1[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t m = (size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h);
        dx1v[r] = g.dx1[m * g.lddx1 + n];
        ar1[r >> 2] |= (uint32_t)g.arg1[m * CM + n] << (8 * (r & 3));
    }

    typedef unsigned zraw_t __attribute__((ext_vector_type((NOZ ? BF16 : ZBF16) ? 2 : 4)));
    zraw_t zr[PASSES];          // the builder's rows of Z -- NOZ: of Y1e
    auto load_z = [&](int t) {
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            if constexpr (NOZ) {
                const unsigned so = (unsigned)(RPP * e) * rowb + (unsigned)t * (CM * ES);
                if constexpr (BF16) zr[e] = __builtin_amdgcn_raw_buffer_load_b64(yres, yqoff, so, 0);
                else zr[e] = __builtin_amdgcn_raw_buffer_load_b128(yres, yqoff, so, 0);
            } else {
                const unsigned so = (unsigned)(RPP * e) * rowbz + (unsigned)t * (CM * ESZ);
                if constexpr (ZBF16) zr[e] = __builtin_amdgcn_raw_buffer_load_b64(zres, zoff, so, 0);
                else zr[e] = __builtin_amdgcn_raw_buffer_load_b128(zres, zoff, so, 0);
            }
        }
    };
    auto build = [&](int buf, int t) {
        __bf16* hi_img = smem16 + buf * NB * IMG;
        const uint32_t tb = (uint32_t)t;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const int p = prow + RPP * e;
            if constexpr (NOZ) {      // the rows of Y1e: its hi image (bf16 rows: copied); dZ is the selected gradient alone
                uint2 yh;
                if constexpr (BF16) yh = make_uint2(zr[e][0], zr[e][1]);
                else {
                    uint2 ylo;
                    em_split4_packed(__uint_as_float(zr[e][0]), __uint_as_float(zr[e][1]), __uint_as_float(zr[e][2]), __uint_as_float(zr[e][3]), yh, ylo);
                }
                *reinterpret_cast<uint2*>(hi_img + YI + p * LDK + c4 * 4) = yh;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ((ar2[e] >> (8 * c)) & 0xffu) == tb ? dp2[e][c] : 0.0f;
                uint2 hh, ll;
                em_split4_packed(v[0], v[1], v[2], v[3], hh, ll);
                *reinterpret_cast<uint2*>(hi_img + p * LDK + c4 * 4) = hh;
                if constexpr (!BF16) *reinterpret_cast<uint2*>(hi_img + IMG + p * LDK + c4 * 4) = ll;
                continue;
            }
            float z[4];
            if constexpr (ZBF16) {
                z[0] = __uint_as_float(zr[e][0] << 16); z[1] = __uint_as_float(zr[e][0] & 0xffff0000u);
                z[2] = __uint_as_float(zr[e][1] << 16); z[3] = __uint_as_float(zr[e][1] & 0xffff0000u);
            } else {
                z[0] = __uint_as_float(zr[e][0]); z[1] = __uint_as_float(zr[e][1]); z[2] = __uint_as_float(zr[e][2]); z[3] = __uint_as_float(zr[e][3]);
            }
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = fmaf(z[c], a1[c], a0[c]) + (((ar2[e] >> (8 * c)) & 0xffu) == tb ? dp2[e][c] : 0.0f);
            uint2 hh, ll;
            em_split4_packed(v[0], v[1], v[2], v[3], hh, ll);
            *reinterpret_cast<uint2*>(hi_img + p * LDK + c4 * 4) = hh;
            if constexpr (!BF16) *reinterpret_cast<uint2*>(hi_img + IMG + p * LDK + c4 * 4) = ll;
        }
    };
    unsigned yv[16];
    auto load_y = [&](int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned so = (unsigned)((r & 3) + 8 * (r >> 2)) * rowb + (unsigned)t * (CM * ES);
            if constexpr (BF16) yv[r] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(yres, eoff, so, 0) << 16;
            else yv[r] = __builtin_amdgcn_raw_buffer_load_b32(yres, eoff, so, 0);
        }
    };

    float gsum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gsum[r] = 0.0f;

    load_z(0);
    build(0, 0);
    if (g.k > 1) load_z(1);
    __syncthreads();
    for (int t = 0; t < g.k; ++t) {
        const int buf = t & 1;
        if constexpr (!(NOZ && BF16)) load_y(t);     // this slot's Y1e elements: in flight under the MFMAs (NOZ, bf16: read from the image)
        const __bf16* ah = smem16 + buf * NB * IMG + col * LDK + h * 8;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const em_bf16x8 a_hi = *reinterpret_cast<const em_bf16x8*>(ah + s * 16);
            if constexpr (!BF16) {
                const em_bf16x8 a_lo = *reinterpret_cast<const em_bf16x8*>(ah + IMG + s * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi[s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi[s], acc, 0, 0, 0);
            if constexpr (NOZ) {
                const em_bf16x8 a_y = *reinterpret_cast<const em_bf16x8*>(ah + YI + s * 16);
                const em_bf16x8 b_k = *reinterpret_cast<const em_bf16x8*>(kfrag + s * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_y, b_k, acc, 0, 0, 0);
            }
        }
        // (NOZ: this slot's Y1e elements -- rows p(r), column n -- are read out of the image the MFMAs just used, where they are needed)
        const __bf16* yi = smem16 + buf * NB * IMG + YI + (4 * h) * LDK + n;
        if (t + 1 < g.k) {
            build(buf ^ 1, t + 1);                   // (its Z rows were requested a slot ago)
            if (t + 2 < g.k) load_z(t + 2);
        }
        const uint32_t tb = (uint32_t)t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y;
            if constexpr (NOZ) y = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, yi[((r & 3) + 8 * (r >> 2)) * LDK]) << 16);
            else y = __uint_as_float(yv[r]);
            const float gy = acc[r] + cvn + (((ar1[r >> 2] >> (8 * (r & 3))) & 0xffu) == tb ? dx1v[r] : 0.0f);
            const float gg = gy * (y > 0.0f ? 1.0f : ns);
            sg += gg;
            sgx = fmaf(gy, y, sgx);      // sum G pre1: G pre1 = gy y on BOTH sides of the activation (G = gy ns, pre1 = y / ns); xhat1 at the end
            gsum[r] += gg;
            const unsigned so = (unsigned)((r & 3) + 8 * (r >> 2)) * rowb + (unsigned)t * (CM * ES);
            if constexpr (BF16) {
                const unsigned mine = __builtin_bit_cast(unsigned short, (__bf16)gg);
                const unsigned other = lpd_lane_xor1(mine);
                if (!(col & 1)) __builtin_amdgcn_raw_buffer_store_b32(mine | (other << 16), gres, goff, so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gg), gres, goff, so, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) g.gsum[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h) * CM + n] = gsum[r];
    }      // tiles of this block
    sg += __shfl_xor(sg, 32);
    sgx += __shfl_xor(sgx, 32);
    sgx = (sgx - beta1 * sg) * rg1;      // sum G xhat1 = (sum G pre1 - beta1 sum G) / gamma1
    if (h == 0) {
        atomicAdd(&g.sum[lpd_stat_rofs() + n], (double)sg);
        atomicAdd(&g.sumx[lpd_stat_rofs() + n], (double)sgx);
    }
}

// tiles a block of the fused edge-MLP kernels walks: the smallest count that lets ONE resident round of blocks (two per CU = 512) cover
// the launch (a second, part-filled round costs a whole block time), at most 16; small launches keep one tile per block
inline int em_tiles_per_block(int tiles)
{
    static const int tmax = lpd_debug("edge-mlp-tiles", 16);
    const int t = (tiles + 511) / 512;
    return t < 1 ? 1 : (t > tmax ? tmax : t);
}

template <int CM, int CO>
int edge_mlp_x3_launch(const EdgeMlpArgs& g, hipStream_t stream)
{
    if constexpr (CM == 128 && CO == 128) {
        if (g.x1_hi) {      // with the x1 planes: 32 points per block (see edge_mlp_x3_kernel)
            constexpr int PTS = 32;
            using Cfg = EdgeMlpX3Cfg<CM, CO, PTS>;
            const size_t lds = (size_t)4 * Cfg::IMG * sizeof(__bf16) + (size_t)PTS * g.k * sizeof(int);
            auto kern = edge_mlp_x3_kernel<CM, CO, PTS, true>;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            // one resident round of blocks (two per CU) where the launch is large enough: 32 clouds = 4096 tiles = 512 blocks of 8 tiles
            // (1 / 2 / 4 / 8 tiles per block: 304 / 292 / 285 / 281 us); small launches keep one tile per block
            const int tiles = (g.M + PTS - 1) / PTS;
            EdgeMlpArgs g2 = g;
            g2.tiles_per_block = em_tiles_per_block(tiles);
            hipLaunchKernelGGL(kern, dim3((tiles + g2.tiles_per_block - 1) / g2.tiles_per_block), dim3(EM_THREADS), lds, stream, g2);
            LPD_CHECK_LAUNCH("lpd_edge_mlp(bf16x3 + x1)");
            return LPD_OK;
        }
    }
    using Cfg = EdgeMlpX3Cfg<CM, CO>;
    size_t lds = (size_t)4 * Cfg::IMG * sizeof(__bf16) + (size_t)EM_PTS * g.k * sizeof(int);
    auto kern = edge_mlp_x3_kernel<CM, CO>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (g.M + EM_PTS - 1) / EM_PTS;
    EdgeMlpArgs g2 = g;
    g2.tiles_per_block = em_tiles_per_block(tiles);
    dim3 grid((tiles + g2.tiles_per_block - 1) / g2.tiles_per_block);
    hipLaunchKernelGGL(kern, grid, dim3(EM_THREADS), lds, stream, g2);
    LPD_CHECK_LAUNCH("lpd_edge_mlp(bf16x3)");
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
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_gather_max: act=%d unsupported (none/ReLU/LeakyReLU)", act);
    LPD_CHECK_ARG(ldp % 4 == 0 && ldo % 4 == 0 && (!Q || ldq % 4 == 0), "lpd_edge_gather_max: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)out | (uintptr_t)Q | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
                  "lpd_edge_gather_max: pointers must be 16-byte aligned");
    GatherArgs g{P, Q, idx, out, scale, shift, M, N, C, k, ldp, ldq, ldo, act, slope, 0, 0, 0, 0, 0, 0, 0};
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

static int edge_mlp_entry(bool x3, const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                          const float* b1, const float* W2, const float* s2, const float* b2, float* out, int ldo,
                          int M, int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream_,
                          long long out_lo = 0, void* x1_hi = nullptr)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx && s1 && b1 && W2 && s2 && b2 && out, "lpd_edge_mlp: null pointer");
    LPD_CHECK_ARG(!out_lo || (x3 && out_cloud && out_lo % 8 == 0), "lpd_edge_mlp: split output needs the bf16x3 kernel and cloud-panel planes");
    LPD_CHECK_ARG(M > 0 && N > 0 && k > 0 && M % N == 0, "lpd_edge_mlp: bad dims M=%d N=%d k=%d", M, N, k);
    LPD_CHECK_ARG(k <= 128, "lpd_edge_mlp: k=%d > 128 unsupported", k);
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_mlp: act=%d unsupported (none/ReLU/LeakyReLU)", act);
    LPD_CHECK_ARG(act != 2 || (slope >= 0.0f && slope <= 1.0f), "lpd_edge_mlp: LeakyReLU slope %g outside [0, 1]", (double)slope);
    LPD_CHECK_ARG(ldp % 4 == 0 && (!Q || ldq % 4 == 0), "lpd_edge_mlp: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)s1 | (uintptr_t)b1 | (uintptr_t)W2) & 15) == 0,
                  "lpd_edge_mlp: pointers must be 16-byte aligned");
    LPD_CHECK_ARG(!out_cloud || (N % EM_PTS == 0 && panel_ld >= N), "lpd_edge_mlp: cloud-panel out needs N %% 64 == 0 and panel_ld >= N");
    LPD_CHECK_ARG(!x1_hi || (out_lo && CM == 128 && CO == 128 && M % 32 == 0 && ((uintptr_t)x1_hi & 7) == 0),
                  "lpd_edge_mlp: the x1 planes ride with the split 128 -> 128 stage (M %% 32 == 0)");
    EdgeMlpArgs g{P, Q, idx, s1, b1, W2, s2, b2, out, M, N, k, ldp, ldq, ldo, act, slope, out_cloud, panel_ld, out_lo, x1_hi};
    if (CM == 128 && CO == 128) return x3 ? edge_mlp_x3_launch<128, 128>(g, stream) : edge_mlp_launch<128, 128>(g, stream);
    if (CM == 64 && CO == 64) return x3 ? edge_mlp_x3_launch<64, 64>(g, stream) : edge_mlp_launch<64, 64>(g, stream);
    lpd_set_error("lpd_edge_mlp: (CM=%d, CO=%d) unsupported; built for (128,128) and (64,64)", CM, CO);
    return LPD_ERR_UNSUPPORTED;
}

extern "C" int lpd_pack_idx16(const int32_t* idx, uint16_t* idx16, long long M, int k, void* stream_)
{
    LPD_CHECK_ARG(idx && idx16 && M > 0, "lpd_pack_idx16: bad arguments");
    LPD_CHECK_ARG(k == 20, "lpd_pack_idx16: built for k = 20 (got %d)", k);
    LPD_CHECK_ARG((((uintptr_t)idx | (uintptr_t)idx16) & 15) == 0, "lpd_pack_idx16: pointers must be 16-byte aligned");
    const long long threads = M * 5;
    hipLaunchKernelGGL(pack_idx16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, idx,
                       reinterpret_cast<uint2*>(idx16), M);
    LPD_CHECK_LAUNCH("lpd_pack_idx16");
    return LPD_OK;
}

static int edge_gather_max16_impl(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, float* out,
                                  int ldo, const float* scale, const float* shift, int M, int N, int C, int k, int act,
                                  float slope, long long p_cloud, long long q_cloud, long long o_cloud, int panel_ld, long long o_lo,
                                  void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx16 && out, "lpd_edge_gather_max16: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && k > 0 && M % N == 0, "lpd_edge_gather_max16: bad dims M=%d N=%d k=%d", M, N, k);
    LPD_CHECK_ARG(C > 0 && C % 8 == 0, "lpd_edge_gather_max16: C=%d must be a multiple of 8", C);
    LPD_CHECK_ARG((p_cloud || ldp % 4 == 0) && (o_cloud || ldo % 4 == 0) && (!Q || q_cloud || ldq % 4 == 0),
                  "lpd_edge_gather_max16: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)out | (uintptr_t)Q | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0 &&
                  ((uintptr_t)idx16 & 7) == 0, "lpd_edge_gather_max16: pointers must be 16-byte aligned (idx16: 8)");
    LPD_CHECK_ARG(k == 20 && N <= 4096,
                  "lpd_edge_gather_max16: built for k = 20 and N <= 4096 (an 8-channel slice of one cloud in LDS, 16-bit byte offsets); got k=%d N=%d", k, N);
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_gather_max16: act=%d unsupported (none/ReLU/LeakyReLU)", act);
    LPD_CHECK_ARG((unsigned long long)N * (unsigned long long)(o_cloud ? 8 : ldo) * 4ull < (1ull << 32) &&
                  (unsigned long long)N * (unsigned long long)(q_cloud ? 8 : ldq) * 4ull < (1ull << 32) &&
                  (unsigned long long)M * 40ull < (1ull << 32), "lpd_edge_gather_max16: M=%d too large for 32-bit row offsets", M);
    // p_cloud / q_cloud / o_cloud != 0: that operand is in cloud-panel layout [cloud][.][N][8] with this many floats between
    // clouds (its leading dim is then ignored)
    const long long ps = (long long)panel_ld * 8;
    GatherArgs g{P, Q, nullptr, out, scale, shift, M, N, C, k, p_cloud ? 8 : ldp, q_cloud ? 8 : ldq, o_cloud ? 8 : ldo, act, slope,
                 p_cloud ? ps : 8, q_cloud ? ps : 8, o_cloud ? ps : 8,
                 p_cloud ? p_cloud : (long long)N * ldp, q_cloud ? q_cloud : (long long)N * ldq, o_cloud ? o_cloud : (long long)N * ldo, o_lo};
    const int nslices = C / 8;
    const size_t lds = (size_t)KAGG_IMG1 + (size_t)N * 16;
    const float ns = act == 0 ? 1.0f : (act == 1 ? 0.0f : slope);
    // persistent form: one workgroup per CU, (items / 256 CUs) consecutive items each; LPD_DEBUG=kagg-persist=0 disables, n forces n
    static const int persist = lpd_debug("kagg-persist", -1);
    const int nwork = (M / N) * nslices;
    const int per_auto = (nwork + 255) / 256;
    const bool persistent = persist != 0 && N > 3584 && N <= 4096 && (persist > 0 ? persist : per_auto) >= 3;   // eight passes of 512
                                              // points; with two items per CU the dynamic launch is faster (C = 128: 61 vs 72 us)
    const int per = persist > 0 ? persist : per_auto;
    auto launch_p = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((nwork + per - 1) / per), dim3(1024), lds, stream, g, idx16, nslices, ns, nwork, per);
    };
    auto launch_d = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((M / N) * nslices), dim3(1024), lds, stream, g, idx16, nslices, ns);
    };
    if (persistent) {
        if (o_lo) { if (Q) launch_p(edge_gather_max_cloud16p_kernel<true, true>); else launch_p(edge_gather_max_cloud16p_kernel<false, true>); }
        else { if (Q) launch_p(edge_gather_max_cloud16p_kernel<true, false>); else launch_p(edge_gather_max_cloud16p_kernel<false, false>); }
    } else {
        if (o_lo) { if (Q) launch_d(edge_gather_max_cloud16_kernel<true, true>); else launch_d(edge_gather_max_cloud16_kernel<false, true>); }
        else { if (Q) launch_d(edge_gather_max_cloud16_kernel<true, false>); else launch_d(edge_gather_max_cloud16_kernel<false, false>); }
    }
    LPD_CHECK_LAUNCH("lpd_edge_gather_max16");
    return LPD_OK;
}

extern "C" int lpd_edge_gather_max16(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, float* out,
                                     int ldo, const float* scale, const float* shift, int M, int N, int C, int k, int act,
                                     float slope, long long p_cloud, long long q_cloud, long long o_cloud, int panel_ld, void* stream_)
{
    return edge_gather_max16_impl(P, ldp, Q, ldq, idx16, out, ldo, scale, shift, M, N, C, k, act, slope, p_cloud, q_cloud, o_cloud,
                                  panel_ld, 0, stream_);
}

// the same with SPLIT output: out_hi = hi plane of a pair of bf16 cloud-panel planes [cloud][C/8][panel_ld][8] (o_cloud elements
// between clouds), the lo plane o_lo elements behind it (hi = bf16(x), lo = bf16(x - hi): what lpd_gemm_p8 / lpd_gemm_x3t read)
extern "C" int lpd_edge_gather_max16s(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, void* out_hi,
                                      long long o_lo, const float* scale, const float* shift, int M, int N, int C, int k, int act,
                                      float slope, long long p_cloud, long long q_cloud, long long o_cloud, int panel_ld, void* stream_)
{
    LPD_CHECK_ARG(o_cloud != 0 && o_lo != 0 && (o_lo % 8) == 0, "lpd_edge_gather_max16s: split output needs cloud panels and a lo-plane offset");
    return edge_gather_max16_impl(P, ldp, Q, ldq, idx16, reinterpret_cast<float*>(out_hi), 8, scale, shift, M, N, C, k, act, slope,
                                  p_cloud, q_cloud, o_cloud, panel_ld, o_lo, stream_);
}

extern "C" int lpd_edge_mlp(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                            const float* b1, const float* W2, const float* s2, const float* b2, float* out, int ldo,
                            int M, int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream)
{
    return edge_mlp_entry(false, P, ldp, Q, ldq, idx, s1, b1, W2, s2, b2, out, ldo, M, N, CM, CO, k, act, slope, out_cloud, panel_ld, stream);
}

extern "C" int lpd_edge_mlp_bf16x3(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                                   const float* b1, const float* W2, const float* s2, const float* b2, float* out, int ldo,
                                   int M, int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream)
{
    return edge_mlp_entry(true, P, ldp, Q, ldq, idx, s1, b1, W2, s2, b2, out, ldo, M, N, CM, CO, k, act, slope, out_cloud, panel_ld, stream);
}

// lpd_edge_mlp_bf16x3 with SPLIT output: out_hi = hi plane of a pair of bf16 cloud-panel planes (out_cloud elements between clouds),
// the lo plane out_lo elements behind it
extern "C" int lpd_edge_mlp_bf16x3s(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                                    const float* b1, const float* W2, const float* s2, const float* b2, void* out_hi, long long out_lo,
                                    int M, int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream)
{
    LPD_CHECK_ARG(out_lo != 0 && out_cloud != 0, "lpd_edge_mlp_bf16x3s: split output needs cloud panels and a lo-plane offset");
    return edge_mlp_entry(true, P, ldp, Q, ldq, idx, s1, b1, W2, s2, b2, reinterpret_cast<float*>(out_hi), 8, M, N, CM, CO, k, act, slope,
                          out_cloud, panel_ld, stream, out_lo);
}

// ... and with x1 = max over k of the stage-1 activation (the DG1-stage K-agg, util/lpdnet_model.py:249-250) written on the way, as a
// second pair of planes x1_hi (+ out_lo) with the layout of out_hi: replaces lpd_pack_idx16 + lpd_edge_gather_max16s on the same graph.
// 128 -> 128 channels, M % 32 == 0.
extern "C" int lpd_edge_mlp_x1_bf16x3s(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                                       const float* b1, const float* W2, const float* s2, const float* b2, void* out_hi, void* x1_hi,
                                       long long out_lo, int M, int N, int k, int act, float slope, long long out_cloud, int panel_ld,
                                       void* stream)
{
    LPD_CHECK_ARG(out_lo != 0 && out_cloud != 0 && x1_hi, "lpd_edge_mlp_x1_bf16x3s: split outputs need cloud panels, a lo-plane offset and the x1 planes");
    return edge_mlp_entry(true, P, ldp, Q, ldq, idx, s1, b1, W2, s2, b2, reinterpret_cast<float*>(out_hi), 8, M, N, 128, 128, k, act, slope,
                          out_cloud, panel_ld, stream, out_lo, x1_hi);
}

// Train-mode DG1 -> DG2 stage in one launch (edge_mlp_train_kernel): Y1e, Z (raw), the statistics of Z, the selected raw values and
// their slots.  bf16 != 0: Y1e / Z are bf16 tensors.  M % 64 == 0, N % 64 == 0, 128 -> 128 channels, k <= 255.
extern "C" int lpd_edge_mlp_train(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1, const float* b1,
                                  const float* W2, const float* gamma2, void* Y1e, void* Z, int bf16, int z_bf16, float* zsel, int ldsel,
                                  uint8_t* arg2, double* sum, double* sumsq, int M, int N, int k, int act, float slope, double* stat_ws,
                                  void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx && s1 && b1 && W2 && gamma2 && Y1e && zsel && arg2 && sum && sumsq, "lpd_edge_mlp_train: null pointer");      // (Z may be null)
    LPD_CHECK_ARG(M > 0 && N > 0 && M % N == 0 && M % EM_PTS == 0 && N % EM_PTS == 0 && k > 0 && k <= 255,
                  "lpd_edge_mlp_train: M=%d N=%d k=%d unsupported (multiples of 64, k <= 255)", M, N, k);
    LPD_CHECK_ARG(act >= 0 && act <= 2 && (act != 2 || (slope >= 0.0f && slope <= 1.0f)), "lpd_edge_mlp_train: activation unsupported");
    LPD_CHECK_ARG((unsigned long long)64 * k * 128 * 4 < (1ull << 31), "lpd_edge_mlp_train: k too large");
    LPD_CHECK_ARG(ldp % 4 == 0 && (!Q || ldq % 4 == 0), "lpd_edge_mlp_train: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)s1 | (uintptr_t)b1 | (uintptr_t)W2 | (uintptr_t)Y1e | (uintptr_t)Z) & 15) == 0,
                  "lpd_edge_mlp_train: pointers must be 16-byte aligned");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_mlp_train: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_mlp_train", 128);
    EdgeMlpTrainArgs g{P, Q, idx, s1, b1, W2, gamma2, Y1e, Z, zsel, arg2, ws.sum(), ws.sumsq(), M, N, k, ldp, ldq, ldsel, act, slope};
    using Cfg = EdgeMlpX3Cfg<128, 128>;
    const size_t lds = (size_t)(bf16 ? 2 : 4) * Cfg::IMG * sizeof(__bf16) + (size_t)EM_PTS * k * sizeof(int);
    LPD_CHECK_ARG(!bf16 || z_bf16, "lpd_edge_mlp_train: bf16 Y1e goes with bf16 Z");
    // (one 64-point tile per block: the tile loop that pays in the eval kernel and in the backward costs this kernel 25 more spilled
    //  registers -- 444 -> 496 us)
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(M / EM_PTS), dim3(EM_THREADS), lds, stream, g);
    };
    if (bf16 && !Z) launch(edge_mlp_train_kernel<true, true, false>);
    else if (bf16) launch(edge_mlp_train_kernel<true, true>);
    else if (z_bf16) launch(edge_mlp_train_kernel<false, true>);
    else launch(edge_mlp_train_kernel<false, false>);
    LPD_CHECK_LAUNCH("lpd_edge_mlp_train");
    return lpd_stat_finish(ws, sum, sumsq, 128, stream);
}

// Backward of the train-mode DG1 -> DG2 stage, dense part (edge_mlp_train_bwd_kernel): G = the gradient in front of BatchNorm1 [E][128]
// (bf16 or fp32 like Z / Y1e), gsum[i] = sum_t G[(i,t)], dbeta1 = sum G, dgamma1 = sum G xhat1.  dbeta2 / dgamma2: the sums of
// lpd_bn_sel_bwd_reduce; beta1 / rgamma1: BatchNorm1's bias and 1 / weight; inv_ns = 1 / negative slope (1 for act none).
extern "C" int lpd_edge_mlp_train_bwd(const void* Z, const uint8_t* arg2, const void* dpre2, const float* W2, const float* scale2,
                                      const float* mean2, const float* invstd2, const double* dbeta2, const double* dgamma2, const void* Y1e,
                                      const uint8_t* arg1, const float* dx1, int lddx1, const float* beta1, const float* rgamma1, int bf16,
                                      int z_bf16, void* G, float* gsum, double* dbeta1, double* dgamma1, int M, int k, int act, float slope, float inv_ns,
                                      float* kws, double* stat_ws, void* stream_)
{
    // Z == null: the form without Z (edge_mlp_bwd_prep_kernel); kws: 128 * 128 + 128 floats of scratch for it
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG((Z || kws) && arg2 && dpre2 && W2 && scale2 && mean2 && invstd2 && dbeta2 && dgamma2 && Y1e && arg1 && dx1 && beta1 && rgamma1 && G &&
                  gsum && dbeta1 && dgamma1, "lpd_edge_mlp_train_bwd: null pointer");
    LPD_CHECK_ARG(M > 0 && M % EB_PTS == 0 && k > 0 && k <= 255, "lpd_edge_mlp_train_bwd: M=%d k=%d unsupported (M %% 32 == 0, k <= 255)", M, k);
    LPD_CHECK_ARG((act == 0 || act == 2) && inv_ns >= 1.0f, "lpd_edge_mlp_train_bwd: needs an invertible activation (none / LeakyReLU)");
    LPD_CHECK_ARG((((uintptr_t)Z | (uintptr_t)Y1e | (uintptr_t)G | (uintptr_t)dpre2 | (uintptr_t)arg2) & 15) == 0,
                  "lpd_edge_mlp_train_bwd: pointers must be 16-byte aligned");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_mlp_train_bwd: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_mlp_train_bwd", 128);
    EdgeMlpBwdArgs g{Z, arg2, dpre2, W2, scale2, mean2, invstd2, dbeta2, dgamma2, Y1e, arg1, dx1, lddx1, beta1, rgamma1, G, gsum,
                     ws.sum(), ws.sumsq(), M, k, act, slope, inv_ns, reinterpret_cast<const uint16_t*>(kws), kws ? kws + 128 * 128 : nullptr};
    const int tiles = M / EB_PTS;
    g.tiles_per_block = em_tiles_per_block(tiles);
    const dim3 grid((tiles + g.tiles_per_block - 1) / g.tiles_per_block);
    if (!Z) {
        LPD_CHECK_ARG(bf16, "lpd_edge_mlp_train_bwd: the form without Z is built for bf16 storage");
        LPD_CHECK_ARG(((uintptr_t)kws & 15) == 0, "lpd_edge_mlp_train_bwd: kws must be 16-byte aligned");
        hipLaunchKernelGGL(edge_mlp_bwd_prep_kernel, dim3(128), dim3(128), 0, stream, W2, scale2, mean2, invstd2, dbeta2, dgamma2,
                           (double)M * (double)k, reinterpret_cast<uint16_t*>(kws), kws + 128 * 128);
        LPD_CHECK_LAUNCH("lpd_edge_mlp_train_bwd(prep)");
        const size_t lds_n = ((size_t)2 * 2 * EB_PTS + 128) * (128 + 8) * sizeof(__bf16);      // two buffers of (dZ hi | Y1e hi) + the K image
        (void)hipFuncSetAttribute((const void*)edge_mlp_train_bwd_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n);
        hipLaunchKernelGGL((edge_mlp_train_bwd_kernel<true, true, true>), grid, dim3(EM_THREADS), lds_n, stream, g);
        LPD_CHECK_LAUNCH("lpd_edge_mlp_train_bwd(noz)");
        return lpd_stat_finish(ws, dbeta1, dgamma1, 128, stream);
    }
    const size_t lds = (size_t)(bf16 ? 2 : 4) * EB_PTS * (128 + 8) * sizeof(__bf16);
    LPD_CHECK_ARG(!bf16 || z_bf16, "lpd_edge_mlp_train_bwd: bf16 Y1e goes with bf16 Z");
    if (bf16) hipLaunchKernelGGL((edge_mlp_train_bwd_kernel<true, true>), grid, dim3(EM_THREADS), lds, stream, g);
    else if (z_bf16) hipLaunchKernelGGL((edge_mlp_train_bwd_kernel<false, true>), grid, dim3(EM_THREADS), lds, stream, g);
    else hipLaunchKernelGGL((edge_mlp_train_bwd_kernel<false, false>), grid, dim3(EM_THREADS), lds, stream, g);
    LPD_CHECK_LAUNCH("lpd_edge_mlp_train_bwd");
    return lpd_stat_finish(ws, dbeta1, dgamma1, 128, stream);
}

// Training forward of the split-form edge stage on cloud-resident slices (see edge_split_fwd_cloud16_kernel): same results as
// lpd_edge_split_fwd (S, usel, arg bit-identical; the statistics up to the order of the fp64 additions).  idx16 from lpd_pack_idx16;
// k = 20, N <= 4096, C % 8 == 0, row-major P / Q.
extern "C" int lpd_edge_split_fwd16_applies(int N, int C, int k)
{
    static const bool on = lpd_debug("split-lds", 1) != 0;
    return on && k == 20 && N >= 32 && N <= 4096 && C > 0 && C % 8 == 0 && C <= LPD_STAT_CMAX;
}

extern "C" int lpd_edge_split_fwd16(const float* P, long long ldp, const float* Q, long long ldq, const uint16_t* idx16, const float* gamma,
                                    float* S, float* usel, uint8_t* arg, long long M, int N, int C, int k, double* sum, double* sumsq,
                                    double* stat_ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && Q && idx16 && gamma && S && usel && arg && sum && sumsq, "lpd_edge_split_fwd16: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && M % N == 0 && lpd_edge_split_fwd16_applies(N, C, k), "lpd_edge_split_fwd16: N=%d C=%d k=%d unsupported", N, C, k);
    LPD_CHECK_ARG(ldp % 4 == 0 && ldq % 4 == 0 && ldp < (1ll << 31) && ldq < (1ll << 31), "lpd_edge_split_fwd16: leading dims");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)S | (uintptr_t)usel | (uintptr_t)arg | (uintptr_t)gamma | (uintptr_t)idx16) & 15) == 0,
                  "lpd_edge_split_fwd16: pointers must be 16-byte aligned");
    LPD_CHECK_ARG((unsigned long long)M * (unsigned long long)C * 4ull < (1ull << 34), "lpd_edge_split_fwd16: tensor too large");
    const LpdStatWs ws = lpd_stat_arg(stat_ws);
    LPD_CHECK_ARG(ws.rep, "lpd_edge_split_fwd16: stat_ws is null (lpd_stat_ws_bytes() bytes, zero-filled once by the caller)");
    LPD_CHECK_STAT_COLS("lpd_edge_split_fwd16", C);
    SplitFwdArgs g{P, Q, gamma, S, usel, arg, ws.sum(), ws.sumsq(), N, C, (int)ldp, (int)ldq};
    const int nslices = C / 8;
    const size_t lds = (size_t)KAGG_IMG1 + (size_t)N * 16;
    (void)hipFuncSetAttribute((const void*)edge_split_fwd_cloud16_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(edge_split_fwd_cloud16_kernel<512>, dim3((unsigned)((M / N) * nslices)), dim3(512), lds, stream, g, idx16, nslices);
    LPD_CHECK_LAUNCH("lpd_edge_split_fwd16");
    return lpd_stat_finish(ws, sum, sumsq, C, stream);
}
