// lpd_edge_win.hip -- K-agg for LARGE clouds (BASELINE.json configs[4]: N = 16384, k = 64): Z-order WINDOW of the cloud's
// projection rows in LDS, out-of-window neighbours from L2.
//
// Same contract as lpd_edge_gather_max (lpd_edge.hip; replaces util/lpdnet_model.py:331-363 + the split edge conv + BN +
// activation + max over k):   out[m][c] = act(scale[c] * (sel_t P[cloud(m)*N + idx[m][t]][c] + Q[m][c]) + shift[c]).
//
// Why: the cloud-resident kernel keeps an 8-channel slice of ALL N rows of one cloud in LDS, which ends at N = 4096
// (128 KiB); beyond that the direct form sends all k gathers per point through L2 -- at N = 16384, k = 64 that is
// 68.7 GB per launch against 3.49 GB of algorithmic bytes, and the launch runs at the L2 gather ceiling (5.3 ms = 8 % of the
// HBM roofline, profiles/r01g_bench_cfg5_stress.json).  The clouds are Z-ordered (lpd_morton.hip), so the k nearest
// neighbours of a point cluster around it in memory: a work item here is (cloud, WINDOW of 4095 consecutive rows, slice of 8
// channels); the window's rows go to LDS once (two 64-KiB images, sign-adjusted like the cloud-resident kernel) and the
// 4095 points of the window take their neighbours from LDS when they fall inside it (a 16-bit byte offset, one SDWA add per
// gather) and from global memory / L2 otherwise.  Row 4095 of each image holds -inf: an out-of-window index is CLAMPED onto
// it (off = min(j - w0, 4095) * 16, packed 16-bit arithmetic on index pairs), so the LDS phase is branch-free; lanes that saw
// a clamped index then run the miss phase, which re-walks the index quads and loads the missing row pieces (exec-masked,
// the four loads of a quad in flight together).  Bit-identical to lpd_edge_gather_max (max / min are exact).
#include "lpd_common.h"
#include <math.h>

namespace {

constexpr int KW_ROWS = 4095;                  // rows per window; image row 4095 = -inf (clamp target)
constexpr unsigned KW_IMG1 = 65536 + 128;      // byte offset of the second image (channels 4-7), half a bank row past 64 KiB
constexpr size_t KW_LDS = (size_t)KW_IMG1 + 65536;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct WinArgs {
    const float* P;
    const float* Q;
    float* out;
    const float* scale;
    const float* shift;
    int M, N, C;
    int ldp, ldq, ldo;            // row strides in floats (8 for cloud-panel operands)
    long long p_slice, q_slice, o_slice, p_cloud, q_cloud, o_cloud;   // as in lpd_edge.hip GatherArgs
    int nslices, nwin;
    float ns;                     // negative slope of the activation (1 / 0 / slope)
};

// int32 [M][k] -> raw uint16 indices, blocked by 32 points: quad i (4 indices, one uint2) of point m at ((m / 32) * KQ + i) * 32 + m % 32
__global__ void pack_idx16w_kernel(const int32_t* __restrict__ in, uint2* __restrict__ out, long long M, int KQ)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = t / KQ;
    const int i = (int)(t - m * KQ);
    if (m >= M) return;
    const int4 v = *reinterpret_cast<const int4*>(in + (m * KQ + i) * 4);
    out[((m >> 5) * KQ + i) * 32 + (m & 31)] = make_uint2((uint32_t)(v.x & 0xffff) | ((uint32_t)v.y << 16), (uint32_t)(v.z & 0xffff) | ((uint32_t)v.w << 16));
}

// The same packing with every point's list PARTITIONED: the neighbours outside the point's window first, the others behind them, each
// part in its kNN order (the maximum over a list does not depend on its order).  The window kernel's miss phase walks the index quads
// and skips a quad in which no lane of the wave has a miss; with the misses scattered over the 64 slots every quad of a border wave
// holds some (7.8 % of the gathers at N = 16384, k = 64), i.e. 16 dependent memory round trips; partitioned, the wave stops after
// ceil(max misses of a point / 4) quads.  One wavefront per point: lane e holds list entry e, a ballot and two popcounts place it.
// (A permutation against the LDS bank conflicts of the hit phase -- slot s of point p asking for row residue (pos(p) + s) mod 8, so
// that the 8 points of a ds_read_b128 lane group hit 8 different bank groups -- was built and measured first: all-hit synthetic graph
// 395 -> 335 us, the real kNN graph 731 -> 718 us at B = 16: the hit phase is not where the real graph loses its time.)
__global__ __launch_bounds__(256) void pack_idx16w_part_kernel(const int32_t* __restrict__ in, unsigned short* __restrict__ out, long long M, int N,
                                                               int k)
{
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int e = threadIdx.x & 63;
    if (m >= M) return;
    const int n = (int)(m % N);
    const int w0 = (n / KW_ROWS) * KW_ROWS;
    const bool live = e < k;
    const int j = live ? in[m * k + e] : 0;
    const bool miss = live && (((unsigned)(j - w0) & 0xffffu) >= (unsigned)KW_ROWS);
    const unsigned long long mm = __ballot(miss), hm = __ballot(live && !miss);
    const unsigned long long below = (1ull << e) - 1ull;
    const int at = miss ? __popcll(mm & below) : __popcll(mm) + __popcll(hm & below);
    // element (quad i = at / 4, entry at % 4) of point m: uint16 index ((m / 32) * KQ + i) * 32 * 4 + (m % 32) * 4 + at % 4
    if (live) out[(((m >> 5) * (k >> 2) + (at >> 2)) * 32 + (m & 31)) * 4 + (at & 3)] = (unsigned short)j;
}

template <int KQ, bool HAS_Q>
__global__ __launch_bounds__(1024) void edge_gather_max_window_kernel(WinArgs g, const uint2* __restrict__ idx2)
{
    extern __shared__ float4 win[];
    constexpr int GROUPS = 512;
    const int tid = threadIdx.x;
    const int cl = tid & 1;
    const int grp = tid >> 1;
    const unsigned lbase = cl * KW_IMG1;
    char* winb = reinterpret_cast<char*>(win);
    const int item = lpd_xcd_remap(blockIdx.x, gridDim.x);     // the slices of one window run next to each other on one XCD
    const int sl = item % g.nslices;
    const int w = (item / g.nslices) % g.nwin;
    const int b = item / (g.nslices * g.nwin);
    const int N = g.N;
    const int w0 = w * KW_ROWS;
    const int wn = min(KW_ROWS, N - w0);                        // rows in this window
    const int col = sl * 8 + cl * 4;
    const unsigned row0 = (unsigned)b * N;
    const float* Pc = g.P + b * g.p_cloud + sl * g.p_slice + cl * 4;     // row n of the cloud at + n * ldp
    const float* Qc = HAS_Q ? g.Q + b * g.q_cloud + sl * g.q_slice + cl * 4 : nullptr;
    float* outc = g.out + b * g.o_cloud + sl * g.o_slice + cl * 4;
    const unsigned ldp = g.ldp, ldq = g.ldq, ldo = g.ldo;

    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.scale) sc = *reinterpret_cast<const float4*>(g.scale + col);
    if (g.shift) sh = *reinterpret_cast<const float4*>(g.shift + col);
    const float4 sg = make_float4(sc.x >= 0.f ? 1.f : -1.f, sc.y >= 0.f ? 1.f : -1.f, sc.z >= 0.f ? 1.f : -1.f, sc.w >= 0.f ? 1.f : -1.f);
    const float ns = g.ns;

    // ---- the window's slice of P -> LDS (sign-adjusted); image row 4095 = -inf ----
    {
        float4 pr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pr[i] = *reinterpret_cast<const float4*>(Pc + (size_t)(w0 + min(grp + i * GROUPS, wn - 1)) * ldp);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = grp + i * GROUPS;
            float4 p = pr[i];
            p.x *= sg.x; p.y *= sg.y; p.z *= sg.z; p.w *= sg.w;
            if (r >= wn) p = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);     // incl. the clamp row 4095
            *reinterpret_cast<float4*>(winb + lbase + r * 16) = p;                           // r <= 4095: 65536 bytes per image
        }
    }
    __syncthreads();

    const u16x2 w0v = {(unsigned short)w0, (unsigned short)w0};
    const u16x2 capv = {(unsigned short)KW_ROWS, (unsigned short)KW_ROWS};
    const int passes = (wn + GROUPS - 1) / GROUPS;
    for (int ps = 0; ps < passes; ++ps) {
        const int n = w0 + min(ps * GROUPS + grp, wn - 1);     // a pass past the end re-does the last point (identical store)
        const unsigned m = row0 + n;
        const uint2* ip = idx2 + ((size_t)(m >> 5) * (KQ * 32) + (m & 31));
        uint2 ix[KQ];
#pragma unroll
        for (int i = 0; i < KQ; ++i) ix[i] = ip[i * 32];
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HAS_Q) q = *reinterpret_cast<const float4*>(Qc + (size_t)n * ldq);
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        u16x2 far = {0, 0};                                    // running max of the window-relative indices: >= 4095 <=> a miss
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const u16x2 d0 = __builtin_bit_cast(u16x2, ix[i].x) - w0v, d1 = __builtin_bit_cast(u16x2, ix[i].y) - w0v;   // wraps below w0
            far = __builtin_elementwise_max(far, __builtin_elementwise_max(d0, d1));
            const unsigned o0 = __builtin_bit_cast(unsigned, (u16x2)(__builtin_elementwise_min(d0, capv) << 4));
            const unsigned o1 = __builtin_bit_cast(unsigned, (u16x2)(__builtin_elementwise_min(d1, capv) << 4));
            const float4 a = *reinterpret_cast<const float4*>(winb + ((o0 & 0xffffu) + lbase));
            const float4 bq = *reinterpret_cast<const float4*>(winb + ((o0 >> 16) + lbase));
            const float4 c = *reinterpret_cast<const float4*>(winb + ((o1 & 0xffffu) + lbase));
            const float4 d = *reinterpret_cast<const float4*>(winb + ((o1 >> 16) + lbase));
            v.x = fmaxf(fmaxf(v.x, a.x), bq.x); v.y = fmaxf(fmaxf(v.y, a.y), bq.y);
            v.z = fmaxf(fmaxf(v.z, a.z), bq.z); v.w = fmaxf(fmaxf(v.w, a.w), bq.w);
            v.x = fmaxf(fmaxf(v.x, c.x), d.x); v.y = fmaxf(fmaxf(v.y, c.y), d.y);
            v.z = fmaxf(fmaxf(v.z, c.z), d.z); v.w = fmaxf(fmaxf(v.w, c.w), d.w);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool lane_miss = far.x >= KW_ROWS || far.y >= KW_ROWS;
        if (__any(lane_miss)) {
            // ---- miss phase: neighbours outside the window, from global memory (L2), only by the lanes that miss them: the
            // other lanes of the quad keep a neutral element (negs: sg * negs = -inf); the four loads of a quad are in flight
            // together.  Measured at B = 16, N = 16384, k = 64 (tools/kaggw_bench.py; 7.8 % of the gathers miss): hit phase alone
            // 400 us (an all-hit graph: the LDS gather floor, bank conflicts included), + 340 us for the misses = one memory round
            // trip per quad of a wave with a border point.  Tried and dropped: every lane loading at every slot (a member of
            // its own set where it had no miss): same time on panels, slower on row-major operands; per-point miss lists
            // compacted into LDS and walked four or eight at a time: 1.4 ms (points near a window corner miss more than the
            // list holds and fall back, the list walk is as serial as the quads). ----
            const float4 negs = make_float4(-INFINITY * sg.x, -INFINITY * sg.y, -INFINITY * sg.z, -INFINITY * sg.w);
            // (lpd_pack_idx16w with N > 0 puts a point's misses at the front of its list: the walk then ends at the wave's longest miss
            //  list; two quads = eight row pieces in flight per step)
            constexpr int QS = (KQ % 4 == 0) ? 4 : ((KQ % 2 == 0) ? 2 : 1);
#pragma unroll
            for (int i = 0; i < KQ; i += QS) {
                unsigned jj[4 * QS];
                bool ms[4 * QS];
                bool anym = false;
#pragma unroll
                for (int u = 0; u < QS; ++u) {
                    jj[4 * u] = ix[i + u].x & 0xffffu; jj[4 * u + 1] = ix[i + u].x >> 16;
                    jj[4 * u + 2] = ix[i + u].y & 0xffffu; jj[4 * u + 3] = ix[i + u].y >> 16;
                }
#pragma unroll
                for (int e = 0; e < 4 * QS; ++e) { ms[e] = ((jj[e] - (unsigned)w0) & 0xffffu) >= (unsigned)KW_ROWS; anym |= ms[e]; }
                if (!__any(anym)) continue;
                float4 p[4 * QS];
#pragma unroll
                for (int e = 0; e < 4 * QS; ++e) {
                    p[e] = negs;
                    if (ms[e]) p[e] = *reinterpret_cast<const float4*>(Pc + (size_t)jj[e] * ldp);
                }
#pragma unroll
                for (int e = 0; e < 4 * QS; ++e) {
                    v.x = fmaxf(v.x, sg.x * p[e].x); v.y = fmaxf(v.y, sg.y * p[e].y);
                    v.z = fmaxf(v.z, sg.z * p[e].z); v.w = fmaxf(v.w, sg.w * p[e].w);
                }
            }
        }
        float4 r;
        r.x = sc.x * (sg.x * v.x + q.x) + sh.x;
        r.y = sc.y * (sg.y * v.y + q.y) + sh.y;
        r.z = sc.z * (sg.z * v.z + q.z) + sh.z;
        r.w = sc.w * (sg.w * v.w + q.w) + sh.w;
        r.x = fmaxf(r.x, 0.f) + ns * fminf(r.x, 0.f);
        r.y = fmaxf(r.y, 0.f) + ns * fminf(r.y, 0.f);
        r.z = fmaxf(r.z, 0.f) + ns * fminf(r.z, 0.f);
        r.w = fmaxf(r.w, 0.f) + ns * fminf(r.w, 0.f);
        *reinterpret_cast<float4*>(outc + (size_t)n * ldo) = r;
    }
}

template <int KQ>
void launch_window(const WinArgs& g, const uint2* idx2, int items, hipStream_t stream)
{
    if (g.Q) {
        auto kern = edge_gather_max_window_kernel<KQ, true>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KW_LDS);
        hipLaunchKernelGGL(kern, dim3(items), dim3(1024), KW_LDS, stream, g, idx2);
    } else {
        auto kern = edge_gather_max_window_kernel<KQ, false>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KW_LDS);
        hipLaunchKernelGGL(kern, dim3(items), dim3(1024), KW_LDS, stream, g, idx2);
    }
}

}  // namespace

extern "C" int lpd_pack_idx16w(const int32_t* idx, uint16_t* idx16, long long M, int k, int N, void* stream_)
{
    LPD_CHECK_ARG(idx && idx16 && M > 0, "lpd_pack_idx16w: bad arguments");
    LPD_CHECK_ARG(k > 0 && k % 4 == 0 && k <= 64, "lpd_pack_idx16w: k = %d must be a multiple of 4, <= 64", k);
    LPD_CHECK_ARG((((uintptr_t)idx | (uintptr_t)idx16) & 15) == 0, "lpd_pack_idx16w: pointers must be 16-byte aligned");
    LPD_CHECK_ARG(N == 0 || (N > 0 && M % N == 0 && N <= 57344), "lpd_pack_idx16w: N = %d (points per cloud; 0 = keep the list order)", N);
    const int KQ = k / 4;
    if (N > 0) {      // out-of-window neighbours first (same sets: the result does not change; see pack_idx16w_part_kernel)
        hipLaunchKernelGGL(pack_idx16w_part_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, idx, idx16, M, N, k);
        LPD_CHECK_LAUNCH("lpd_pack_idx16w");
        return LPD_OK;
    }
    const long long threads = M * KQ;
    hipLaunchKernelGGL(pack_idx16w_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, idx,
                       reinterpret_cast<uint2*>(idx16), M, KQ);
    LPD_CHECK_LAUNCH("lpd_pack_idx16w");
    return LPD_OK;
}

extern "C" int lpd_edge_gather_maxw(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, float* out, int ldo,
                                    const float* scale, const float* shift, int M, int N, int C, int k, int act, float slope,
                                    long long p_cloud, long long q_cloud, long long o_cloud, int panel_ld, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(P && idx16 && out, "lpd_edge_gather_maxw: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && M % N == 0, "lpd_edge_gather_maxw: bad dims M=%d N=%d", M, N);
    LPD_CHECK_ARG(C > 0 && C % 8 == 0, "lpd_edge_gather_maxw: C=%d must be a multiple of 8", C);
    LPD_CHECK_ARG(k == 20 || k == 32 || k == 64, "lpd_edge_gather_maxw: built for k in {20, 32, 64} (got %d)", k);
    LPD_CHECK_ARG(N <= 57344, "lpd_edge_gather_maxw: N=%d exceeds the 16-bit window-relative index arithmetic", N);
    LPD_CHECK_ARG((p_cloud || ldp % 4 == 0) && (o_cloud || ldo % 4 == 0) && (!Q || q_cloud || ldq % 4 == 0),
                  "lpd_edge_gather_maxw: leading dims must be multiples of 4");
    LPD_CHECK_ARG((((uintptr_t)P | (uintptr_t)out | (uintptr_t)Q | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0 &&
                  ((uintptr_t)idx16 & 7) == 0, "lpd_edge_gather_maxw: pointers must be 16-byte aligned (idx16: 8)");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_edge_gather_maxw: act=%d unsupported (none/ReLU/LeakyReLU)", act);
    const long long ps = (long long)panel_ld * 8;
    WinArgs g;
    g.P = P; g.Q = Q; g.out = out; g.scale = scale; g.shift = shift;
    g.M = M; g.N = N; g.C = C;
    g.ldp = p_cloud ? 8 : ldp; g.ldq = q_cloud ? 8 : ldq; g.ldo = o_cloud ? 8 : ldo;
    g.p_slice = p_cloud ? ps : 8; g.q_slice = q_cloud ? ps : 8; g.o_slice = o_cloud ? ps : 8;
    g.p_cloud = p_cloud ? p_cloud : (long long)N * ldp;
    g.q_cloud = q_cloud ? q_cloud : (long long)N * ldq;
    g.o_cloud = o_cloud ? o_cloud : (long long)N * ldo;
    g.nslices = C / 8;
    g.nwin = (N + KW_ROWS - 1) / KW_ROWS;
    g.ns = act == 0 ? 1.0f : (act == 1 ? 0.0f : slope);
    const long long items = (long long)(M / N) * g.nwin * g.nslices;
    LPD_CHECK_ARG(items < (1ll << 31), "lpd_edge_gather_maxw: too many work items");
    const uint2* idx2 = reinterpret_cast<const uint2*>(idx16);
    if (k == 20) launch_window<5>(g, idx2, (int)items, stream);
    else if (k == 32) launch_window<8>(g, idx2, (int)items, stream);
    else launch_window<16>(g, idx2, (int)items, stream);
    LPD_CHECK_LAUNCH("lpd_edge_gather_maxw");
    return LPD_OK;
}
