// lpd_gemm_p8.hip -- split-bf16 GEMM on PRE-SPLIT operands: conv3_lpd (512 -> 1024 per point, util/lpdnet_model.py:262) of the eval path.
//
//     C[m][n] = act(scale[n] * sum_k (Ah[m][k] Bh[k][n] + Al[m][k] Bh[k][n] + Ah[m][k] Bl[k][n]) + shift[n])
//
// A = the [x1 | x2 | x3] activations of a batch of clouds, already split by their producers into two bf16 planes
// (hi = bf16(x), lo = bf16(x - hi)) in CLOUD-PANEL layout [cloud][K/8][panel_ld][8]: the 8 consecutive k of one row that a lane of
// v_mfma_f32_32x32x16_bf16 holds are 16 contiguous bytes, and 32 consecutive rows of a panel are 512 contiguous bytes.
// B = a weight matrix as prepared by lpd_gemm_prep_b (hi / lo bf16 in MFMA fragment order, 1 KiB per (32 columns, 16 k)).
// Both operands therefore reach LDS by LDS-DMA (global_load_lds_dwordx4) as whole 1-KiB MFMA fragments -- lane-linear, no swizzle,
// every ds_read_b128 conflict-free -- and nothing is converted inside the kernel.
//
// Structure (MI355X guide, "256^2 8-phase template", re-derived for three products per operand pair):
//   * block tile 256 rows x 256 columns, 8 waves; wave w owns rows 32 w .. 32 w + 31 and ALL 256 columns (8 accumulator tiles =
//     128 registers), so a follow-up product over the columns needs no cross-wave reduction.  The product is computed TRANSPOSED
//     (weights as the MFMA's row operand): an accumulator lane holds 4 consecutive output channels of ONE point, i.e. one float4
//     store per lane and tile quarter, 1 KiB contiguous per wave-instruction when C is cloud panels.
//   * K walks in 32-deep tiles; a K-tile is four 16-KiB staging units -- A hi, A lo, B columns 0..127 (hi + lo), B columns
//     128..255 -- in a ring of 8 LDS slots (128 KiB).  One unit is issued per phase, D units ahead of its first use; waits are
//     counted (vmcnt(2(D-2)) / vmcnt(2(D-3))), never drained.
//   * a K-tile is 4 phases of 12 MFMAs (2 column tiles x 2 k-steps x 3 products); the two wave groups (waves 0-3 / 4-7, SIMD
//     partners) run one barrier apart, so that one group's LDS reads and DMA issue sit beside the other group's MFMAs.
//   * persistent: one workgroup per CU walks its tiles (the 4 column blocks of a row tile run on 4 CUs of one XCD at the same
//     time: A comes from HBM once); the ring keeps filling across tile boundaries, so the epilogue's stores overlap the next tile's
//     first loads.
// LDS reads are inline asm (the compiler's LDS-DMA tracking would otherwise wait vmcnt(0) in front of every ds_read).
#include "lpd_common.h"

typedef __bf16 p8_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 p8_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int p8_u32x4 __attribute__((ext_vector_type(4)));

#define P8_THREADS 512
#define P8_UNIT 16384
#define P8_LDS_BYTES (8 * P8_UNIT)
#define P8_STORES 32          // float4 stores per wave in the epilogue (8 column tiles x 4 quarters)

struct P8Args {
    const __bf16* a_hi;
    const __bf16* a_lo;
    const __bf16* fhi;
    const __bf16* flo;
    float* C;
    const float* scale;
    const float* shift;
    int M, N, K, KS;
    long long a_cloud;      // bf16 elements between consecutive clouds of a_hi / a_lo
    long long c_cloud;      // floats between consecutive clouds of C (cloud-panel C)
    int panel_n;            // points per cloud
    int a_panel_ld;         // rows per A panel in memory
    int c_panel_ld;         // rows per C panel in memory (cloud-panel C)
    int ldc;                // row-major C
    int act;
    float slope;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void p8_glds(const void* src, void* lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ p8_bf16x8 p8_lds_read(unsigned addr)
{
    p8_u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_bit_cast(p8_bf16x8, v);
}

// 4 floats at a wave-uniform address through the scalar cache (constant address space: s_load, no vmcnt traffic -- an ordinary
// load would sit in the vector-memory queue between the LDS-DMA units and make the compiler drain it)
__device__ __forceinline__ float4 p8_uniform4(const float* p)
{
    typedef float cf4 __attribute__((ext_vector_type(4)));
    const cf4 v = *reinterpret_cast<const __attribute__((address_space(4))) cf4*>((unsigned long long)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

template <int N>
__device__ __forceinline__ void p8_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// D: units in flight ahead of the phase that issues; CPANELS: C in cloud panels
template <int D, bool CPANELS>
__global__ __launch_bounds__(P8_THREADS, 2) void gemm_p8_kernel(P8Args g)
{
    extern __shared__ __attribute__((aligned(1024))) char p8_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int h = lane >> 5, col = lane & 31;

    // ---- this workgroup's tiles: blocks b and b + 8 share an XCD; inside an XCD, tiles_n consecutive items share a row tile ----
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int rows_xcd = (g.tiles_m + 7 - xcd) >> 3;                 // row tiles xcd, xcd + 8, ...
    const int items = rows_xcd * g.tiles_n;
    const int nmine = items > slot ? (items - slot + nslot - 1) / nslot : 0;
    if (nmine == 0) return;
    const int nkt = g.K >> 5;

    auto tile_of = [&](int seq, int& rt, int& ct) {
        const int j = slot + seq * nslot;
        rt = xcd + 8 * (j / g.tiles_n);
        ct = j % g.tiles_n;
    };

    // ---- issue cursor: one 16-KiB unit per call, in ring order ----
    int iu = 0;                      // global unit counter of this workgroup
    int i_seq = 0, i_kt = 0;         // tile (sequence index) and K-tile of the unit at the cursor
    long long i_aoff = 0;            // element offset of the cursor's tile (cloud + first row) inside a_hi / a_lo
    int i_nt0 = 0;
    auto cursor_tile = [&]() {
        int rt, ct;
        tile_of(i_seq, rt, ct);
        const int m0 = rt * 256;
        const int cloud = m0 / g.panel_n;
        i_aoff = (long long)cloud * g.a_cloud + (long long)(m0 - cloud * g.panel_n) * 8;
        i_nt0 = ct * 8;
    };
    cursor_tile();
    const long long a_lane = ((long long)h * g.a_panel_ld + col) * 8;      // lane part of an A fragment address (elements)
    auto issue = [&](const int ty) {       // ty = iu & 3, passed as a constant of the unrolled phase so that the branches fold
        char* dst = p8_lds + (iu & 7) * P8_UNIT + wave * 1024;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int f = wave + 8 * e;
            const __bf16* src;
            if (ty < 2) {            // A hi / A lo: fragment f = (row tile f >> 1, k-step f & 1)
                const __bf16* plane = ty == 0 ? g.a_hi : g.a_lo;
                src = plane + i_aoff + ((long long)(i_kt * 4 + (f & 1) * 2) * g.a_panel_ld + (f >> 1) * 32) * 8 + a_lane;
            } else {                 // B columns (ty - 2) * 128 ..: fragment f = (column tile f >> 2, k-step (f >> 1) & 1, hi / lo f & 1)
                const __bf16* plane = (f & 1) ? g.flo : g.fhi;
                const int nt = i_nt0 + (ty - 2) * 4 + (f >> 2);
                src = plane + (((long long)nt * g.KS + i_kt * 2 + ((f >> 1) & 1)) * 64 + lane) * 8;
            }
            p8_glds(src, dst + e * 8192);
        }
        ++iu;
        if ((iu & 3) == 0) {         // next K-tile (past the end: keep re-reading the last one into slots nothing reads)
            if (i_kt + 1 < nkt) ++i_kt;
            else if (i_seq + 1 < nmine) { ++i_seq; i_kt = 0; cursor_tile(); }
        }
    };

    // ---- prologue: D units in flight, units 0..2 landed ----
#pragma unroll
    for (int u = 0; u < D; ++u) issue(u & 3);
    p8_wait_vm<2 * (D - 3)>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();     // the second group runs one barrier behind from here on

    const unsigned lds0 = (unsigned)(size_t)p8_lds;
    const unsigned rd_lane = lane * 16;
    const float ns = g.act == 0 ? 1.0f : (g.act == 1 ? 0.0f : g.slope);

    int cu = 0;      // unit counter of the compute side: first unit of the current K-tile
    for (int seq = 0; seq < nmine; ++seq) {
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        for (int kt = 0; kt < nkt; ++kt, cu += 4) {
            const unsigned sbase = lds0 + (cu & 4) * P8_UNIT + rd_lane;
            p8_bf16x8 ah[2], al[2];
            const bool first = (kt == 0) && (seq > 0);        // the epilogue's stores sit in the vmcnt queue of this K-tile's first waits
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // ---- load segment: operand fragments of this phase, one unit of a later K-tile ----
                p8_bf16x8 bh[2][2], bl[2][2];
                if (p == 0) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        ah[s] = p8_lds_read(sbase + (wave * 2 + s) * 1024);
                        al[s] = p8_lds_read(sbase + P8_UNIT + (wave * 2 + s) * 1024);
                    }
                }
                const unsigned sb = sbase + (2 + (p >> 1)) * P8_UNIT;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int ctl = (2 * p + j) & 3;
                        bh[j][s] = p8_lds_read(sb + ((ctl * 2 + s) * 2 + 0) * 1024);
                        bl[j][s] = p8_lds_read(sb + ((ctl * 2 + s) * 2 + 1) * 1024);
                    }
                issue((p + D) & 3);
                if (p == 1) {          // unit 3 of this K-tile (columns 128..255) is read in the next phase
                    if (first) p8_wait_vm<2 * (D - 2) + P8_STORES>();
                    else p8_wait_vm<2 * (D - 2)>();
                }
                if (p == 3) p8_wait_vm<2 * (D - 3)>();      // units 0..2 of the next K-tile
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- matrix segment ----
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16& c = acc[2 * p + j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j][s], al[s], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j][s], ah[s], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j][s], ah[s], c, 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }

        // ---- epilogue: lane (point col, half h) holds channels 32 t + 8 q + 4 h + 0..3 of its point in acc[t][4 q ..] ----
        int rt, ct;
        tile_of(seq, rt, ct);
        const int m = rt * 256 + wave * 32 + col;
        const int n0 = ct * 256;
        float* crow;
        if constexpr (CPANELS) {
            const int cloud = m / g.panel_n;
            crow = g.C + (long long)cloud * g.c_cloud + (long long)(m - cloud * g.panel_n) * 8 + h * 4;
        } else crow = g.C + (long long)m * g.ldc + h * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c8 = n0 + t * 32 + q * 8;               // uniform: scale / shift come through the scalar cache
                float4 v;
                float* vv = reinterpret_cast<float*>(&v);
                float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), sc1 = sc0, sh0 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sh0;
                if (g.scale) {
                    sc0 = p8_uniform4(g.scale + c8); sc1 = p8_uniform4(g.scale + c8 + 4);
                    sh0 = p8_uniform4(g.shift + c8); sh1 = p8_uniform4(g.shift + c8 + 4);
                }
                const float scs[4] = {h ? sc1.x : sc0.x, h ? sc1.y : sc0.y, h ? sc1.z : sc0.z, h ? sc1.w : sc0.w};
                const float shs[4] = {h ? sh1.x : sh0.x, h ? sh1.y : sh0.y, h ? sh1.z : sh0.z, h ? sh1.w : sh0.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = fmaf(acc[t][4 * q + i], scs[i], shs[i]);
                    vv[i] = fmaxf(x, 0.0f) + ns * fminf(x, 0.0f);
                }
                if constexpr (CPANELS) *reinterpret_cast<float4*>(crow + (long long)(c8 >> 3) * g.c_panel_ld * 8) = v;
                else *reinterpret_cast<float4*>(crow + c8) = v;
            }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();     // pairs with the second group's last barrier
    p8_wait_vm<0>();                                // no LDS-DMA may outlive the workgroup
}

extern "C" int lpd_gemm_p8_applies(int M, int N, int K, int panel_n)
{
    return (M > 0 && N > 0 && K > 0 && N % 256 == 0 && K % 32 == 0 && panel_n > 0 && panel_n % 256 == 0 && M % panel_n == 0) ? 1 : 0;
}

extern "C" int lpd_gemm_p8(const void* a_hi, const void* a_lo, long long a_cloud, int a_panel_ld, const void* frags, float* C, int ldc,
                           long long c_cloud, int c_panel_ld, int M, int N, int K, int panel_n, const float* scale, const float* shift,
                           int act, float slope, int impl, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(a_hi && a_lo && frags && C, "lpd_gemm_p8: null pointer");
    LPD_CHECK_ARG(lpd_gemm_p8_applies(M, N, K, panel_n), "lpd_gemm_p8: needs N %% 256 == 0, K %% 32 == 0, points per cloud %% 256 == 0");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_gemm_p8: scale and shift must be given together");
    LPD_CHECK_ARG(act >= 0 && act <= 2, "lpd_gemm_p8: act %d", act);
    LPD_CHECK_ARG(a_panel_ld >= panel_n && (c_cloud == 0 || c_panel_ld >= panel_n), "lpd_gemm_p8: panel stride < points per cloud");
    LPD_CHECK_ARG((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)frags | (uintptr_t)C) & 15) == 0, "lpd_gemm_p8: operands must be 16-byte aligned");
    LPD_CHECK_ARG(c_cloud != 0 || ldc % 4 == 0, "lpd_gemm_p8: ldc %% 4 != 0");
    P8Args g;
    g.a_hi = (const __bf16*)a_hi;
    g.a_lo = (const __bf16*)a_lo;
    g.KS = (K + 15) / 16;
    g.fhi = (const __bf16*)frags;
    g.flo = g.fhi + (long long)((N + 31) / 32) * g.KS * 512;
    g.C = C;
    g.scale = scale;
    g.shift = shift;
    g.M = M; g.N = N; g.K = K;
    g.a_cloud = a_cloud; g.c_cloud = c_cloud;
    g.panel_n = panel_n; g.a_panel_ld = a_panel_ld; g.c_panel_ld = c_panel_ld; g.ldc = ldc;
    g.act = act; g.slope = slope;
    g.tiles_m = M / 256; g.tiles_n = N / 256;
    const long long ntiles = (long long)g.tiles_m * g.tiles_n;
    int grid = 256;                                   // one workgroup per CU (128 KiB of LDS each)
    while (grid > 8 && grid / 2 >= ntiles) grid /= 2;
    const int d = impl & 7;
    if (c_cloud != 0) {
        auto kern = d == 6 ? gemm_p8_kernel<6, true> : gemm_p8_kernel<5, true>;
        static bool attr_set[2] = {false, false};
        if (!attr_set[d == 6]) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES); attr_set[d == 6] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(P8_THREADS), P8_LDS_BYTES, stream, g);
    } else {
        auto kern = d == 6 ? gemm_p8_kernel<6, false> : gemm_p8_kernel<5, false>;
        static bool attr_set[2] = {false, false};
        if (!attr_set[d == 6]) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES); attr_set[d == 6] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(P8_THREADS), P8_LDS_BYTES, stream, g);
    }
    LPD_CHECK_LAUNCH("lpd_gemm_p8");
    return LPD_OK;
}

// fp32 activations -> the two bf16 planes lpd_gemm_p8 reads (hi = bf16(x), lo = bf16(x - hi)), cloud panels in and out.
// Used by tests and by producers that do not write the split planes themselves.
__global__ void split_panels_kernel(const float* __restrict__ src, long long s_cloud, int s_panel_ld, __bf16* __restrict__ hi,
                                    __bf16* __restrict__ lo, long long d_cloud, int d_panel_ld, int panels, int n, long long total)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (cloud, panel, row): 8 channels
    if (t >= total) return;
    const int row = (int)(t % n);
    const long long cp = t / n;
    const int panel = (int)(cp % panels);
    const long long cloud = cp / panels;
    const float4* s = reinterpret_cast<const float4*>(src + cloud * s_cloud + ((long long)panel * s_panel_ld + row) * 8);
    const float4 a = s[0], b = s[1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    p8_bf16x8 hh, ll;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hh[i] = (__bf16)x[i];
        ll[i] = (__bf16)(x[i] - (float)hh[i]);
    }
    const long long o = cloud * d_cloud + ((long long)panel * d_panel_ld + row) * 8;
    *reinterpret_cast<p8_bf16x8*>(hi + o) = hh;
    *reinterpret_cast<p8_bf16x8*>(lo + o) = ll;
}

extern "C" int lpd_split_panels(const float* src, long long s_cloud, int s_panel_ld, void* hi, void* lo, long long d_cloud, int d_panel_ld,
                                int clouds, int panels, int n, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(src && hi && lo && clouds > 0 && panels > 0 && n > 0, "lpd_split_panels: bad arguments");
    LPD_CHECK_ARG((((uintptr_t)src | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0, "lpd_split_panels: operands must be 16-byte aligned");
    const long long total = (long long)clouds * panels * n;
    hipLaunchKernelGGL(split_panels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, s_cloud, s_panel_ld,
                       (__bf16*)hi, (__bf16*)lo, d_cloud, d_panel_ld, panels, n, total);
    LPD_CHECK_LAUNCH("lpd_split_panels");
    return LPD_OK;
}
