// lpd_gemm_p8.hip -- split-bf16 GEMM on PRE-SPLIT operands: conv3_lpd (512 -> 1024 per point, util/lpdnet_model.py:262) of the eval path.
//
//     C[m][n] = act(sum_k (Ah[m][k] Bh[k][n] + Al[m][k] Bh[k][n] + Ah[m][k] Bl[k][n]) + bias[n])      (a BatchNorm scale is folded into B)
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
//     time: A comes from HBM once); the ring keeps filling across tile boundaries.
//   * no separate epilogue: the accumulators START from the bias, and a phase works on ONE pair of column tiles, so after the last
//     K-tile's phase p that pair is final -- it is activated and stored in the NEXT phase's load segment (beside the partner
//     group's MFMAs) and re-initialised for the next tile; the waits count those 8 stores per phase.
// LDS reads are inline asm (the compiler's LDS-DMA tracking would otherwise wait vmcnt(0) in front of every ds_read).
#include "lpd_common.h"

typedef __bf16 p8_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 p8_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int p8_u32x4 __attribute__((ext_vector_type(4)));

#define P8_THREADS 512
#define P8_UNIT 16384
#define P8_LDS_BYTES (8 * P8_UNIT)

struct P8Args {
    const __bf16* a_hi;
    const __bf16* a_lo;
    const __bf16* fhi;
    const __bf16* flo;
    float* C;
    const float* bias;      // [N] or null
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
    // fused second product (NetVLAD assignment, util/PointNetVlad.py:48): parts[column block][m][64] = C[m][block's 256 columns] . W2
    const __bf16* w2hi;     // lpd_gemm_prep_b(W2 [N][64], b_kmajor = 1): fragment (cluster tile jt, k-step ks) at ((jt * N/16 + ks) * 64 + lane) * 8
    const __bf16* w2lo;
    float* parts;           // [tiles_n][M][64]
    long long part_stride;  // floats between the planes of two column blocks
};

template <int AUX = 0>
__device__ __forceinline__ void p8_glds(const void* src, void* lds_dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, AUX);
}

// one MFMA fragment (16 bytes per lane) from LDS; the offset is an immediate, so a phase's reads share one address register
template <int OFF>
__device__ __forceinline__ p8_bf16x8 p8_lds_read(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    p8_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
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
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int V>
struct P8Int {
    static constexpr int value = V;
};

// D: units in flight ahead of the phase that issues (5 or 6); CPANELS: C in cloud panels
// DBG (timing-only builds, tools/p8_bench.py): 1 = no LDS-DMA after the prologue, 2 = no MFMAs, 4 = no barriers' partner (one group idle),
// 32 = no output stores (round 3: 333 us against 429 with them in the tool's per-call timing -- the 537 MB of output cost ~95 us that
// the MFMAs do not hide)
// FUSE: a 17th "K-tile" per tile whose ring units are fragments of a second weight matrix W2 [N][64] and whose other operand are the
//   activated accumulators themselves: phase p activates and stores column tiles 2p, 2p + 1 (no separate epilogue either), turns them
//   into MFMA operands in registers (v_permlane32_swap between the two halves of a point, hi / lo split) and accumulates
//   C[32 points][64 of the block's 256 columns] . W2 into two more accumulator tiles; after phase 3 the wave stores its
//   [32 points][64] partial product.  Wave w owns ALL 256 columns of its 32 rows, so no cross-wave reduction is needed; the four
//   column blocks of a row tile leave four partial planes which the consumer (lpd_softmax_affine_parts) sums.
template <int D, bool CPANELS, int DBG = 0, bool FUSE = false>
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

    // ---- issue cursor: one 16-KiB unit (two 1-KiB fragments per wave) per call, in ring order ----
    // wave w stages fragments f = w and w + 8 of a unit.  A unit (hi or lo plane): fragment f = (row tile f >> 1, k-step f & 1);
    // B unit: fragment f = (column tile f >> 2, k-step (f >> 1) & 1, hi / lo f & 1).  The per-lane source pointers run along
    // the K-tiles (one add per unit type and K-tile) and are rebuilt when the cursor moves to the next tile.
    int iu = 0;                      // unit counter of this workgroup (ring slot = iu & 7)
    int i_seq = 0, i_kt = 0;
    const char* pa[2];               // A hi plane, fragments wave and wave + 8 (the lo plane a_delta bytes further)
    const char* pb[2];               // B columns 0..127 of the tile (columns 128..255: b_delta bytes further)
    const long long a_delta = (long long)((const char*)g.a_lo - (const char*)g.a_hi);
    const long long b_delta = (long long)4 * g.KS * 1024;
    const long long a_step = (long long)4 * g.a_panel_ld * 16, b_step = 2048;
    const char* pw[2] = {nullptr, nullptr};      // FUSE: W2 fragments of the tile's column block, unit 0 (unit u: 4 KiB further)
    const int last_kt = FUSE ? nkt : nkt - 1;    // K-tile index nkt = the W2 units
    auto cursor_tile = [&]() {
        int rt, ct;
        tile_of(i_seq, rt, ct);
        const int m0 = rt * 256;
        const int cloud = m0 / g.panel_n;
        const long long aoff = (long long)cloud * g.a_cloud + (long long)(m0 - cloud * g.panel_n) * 8;       // elements
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int f = wave + 8 * e;
            pa[e] = (const char*)g.a_hi + 2 * (aoff + ((long long)((f & 1) * 2 + h) * g.a_panel_ld + (f >> 1) * 32 + col) * 8);
            const __bf16* plane = (f & 1) ? g.flo : g.fhi;
            pb[e] = (const char*)plane + (((long long)(ct * 8 + (f >> 2)) * g.KS + ((f >> 1) & 1)) * 64 + lane) * 16;
            if constexpr (FUSE) {        // W2 unit: fragment f = (k-step f >> 2 of the unit's four, cluster tile (f >> 1) & 1, hi / lo f & 1)
                const __bf16* wpl = (f & 1) ? g.w2lo : g.w2hi;
                pw[e] = (const char*)wpl + (((long long)((f >> 1) & 1) * (g.N >> 4) + ct * 16 + (f >> 2)) * 64 + lane) * 16;
            }
        }
    };
    cursor_tile();
    auto issue = [&](const int ty) {       // ty = iu & 3, a constant of the unrolled phase
        char* dst = p8_lds + (iu & 7) * P8_UNIT + wave * 1024;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const char* src = ty == 0 ? pa[e] : (ty == 1 ? pa[e] + a_delta : (ty == 2 ? pb[e] : pb[e] + b_delta));
            if (FUSE && i_kt == nkt) src = pw[e] + ty * 4096;
            if ((DBG & 1) == 0 || iu < D) p8_glds<(DBG & 8) ? 2 : 0>(src, dst + e * 8192);
        }
        ++iu;
        if (ty == 3) {               // next K-tile (past the end: keep re-reading the last one into slots nothing reads)
            if (i_kt + 1 <= last_kt) { ++i_kt; pa[0] += a_step; pa[1] += a_step; pb[0] += b_step; pb[1] += b_step; }
            else if (i_seq + 1 < nmine) { ++i_seq; i_kt = 0; cursor_tile(); }
        }
    };

    // ---- accumulators start from the bias of their channels: lane (point col, half h), register 4 q + i of tile t <-> channel
    //      n0 + 32 t + 8 q + 4 h + i ----
    f32x16 acc[8];
    auto init_tile = [&](const int t, int n0) {        // n0 < 0: zeros (no bias / no further tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (n0 >= 0) { b0 = p8_uniform4(g.bias + n0 + t * 32 + q * 8); b1 = p8_uniform4(g.bias + n0 + t * 32 + q * 8 + 4); }
            acc[t][4 * q + 0] = h ? b1.x : b0.x;
            acc[t][4 * q + 1] = h ? b1.y : b0.y;
            acc[t][4 * q + 2] = h ? b1.z : b0.z;
            acc[t][4 * q + 3] = h ? b1.w : b0.w;
        }
    };
    const float ns = g.act == 0 ? 1.0f : (g.act == 1 ? 0.0f : g.slope);      // act(v) = max(v, ns * v) for 0 <= ns <= 1
    float pinf = INFINITY;           // max(a, b) = med3(a, b, +inf): fmaxf on accumulator values costs a canonicalising self-max per
    asm volatile("" : "+v"(pinf));   // operand, and the compiler folds a literal +inf back to that form
    float* crow = nullptr;           // this lane's row of C (tile being finished), + 4 h
    int c_n0 = 0;
    auto store_tile = [&](const int t) {
        if constexpr ((DBG & 32) != 0) { if (ns != 12345.0f) return; }      // timing only: no output stores (the waits then over-count: garbage)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c8 = c_n0 + t * 32 + q * 8;
            float4 v;
            v.x = __builtin_amdgcn_fmed3f(acc[t][4 * q + 0], ns * acc[t][4 * q + 0], pinf);
            v.y = __builtin_amdgcn_fmed3f(acc[t][4 * q + 1], ns * acc[t][4 * q + 1], pinf);
            v.z = __builtin_amdgcn_fmed3f(acc[t][4 * q + 2], ns * acc[t][4 * q + 2], pinf);
            v.w = __builtin_amdgcn_fmed3f(acc[t][4 * q + 3], ns * acc[t][4 * q + 3], pinf);
            if constexpr (CPANELS) *reinterpret_cast<float4*>(crow + (long long)(c8 >> 3) * g.c_panel_ld * 8) = v;
            else *reinterpret_cast<float4*>(crow + c8) = v;      // (non-temporal stores: 430 -> 700 us together with nt loads, 485 with nt loads alone)
        }
    };
    auto set_out_tile = [&](int seq) {
        int rt, ct;
        tile_of(seq, rt, ct);
        const int m = rt * 256 + wave * 32 + col;
        c_n0 = ct * 256;
        if constexpr (CPANELS) {
            const int cloud = m / g.panel_n;
            crow = g.C + (long long)cloud * g.c_cloud + (long long)(m - cloud * g.panel_n) * 8 + h * 4;
        } else crow = g.C + (long long)m * g.ldc + h * 4;
    };
    auto bias_n0 = [&](int seq) -> int {       // first column of tile `seq` if it exists and there is a bias, else -1
        if (seq >= nmine || !g.bias) return -1;
        int rt, ct;
        tile_of(seq, rt, ct);
        return ct * 256;
    };

    // ---- prologue: D units in flight, units 0..2 landed ----
#pragma unroll
    for (int u = 0; u < D; ++u) issue(u & 3);
    {
        const int n0 = bias_n0(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) init_tile(t, n0);
    }
    p8_wait_vm<2 * (D - 3)>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();     // the second group runs one barrier behind from here on

    const unsigned lds0 = (unsigned)(size_t)p8_lds + lane * 16;
    int cu = 0;          // first unit of the current K-tile
    int pending = -1;    // tile (sequence index) whose last column-tile pair (6, 7) still has to be stored
    for (int seq = 0; seq < nmine; ++seq) {
        for (int kt = 0; kt < nkt; ++kt, cu += 4) {
            const unsigned sbase = lds0 + (cu & 4) * P8_UNIT;
            p8_bf16x8 ah0, ah1, al0, al1;
            const bool first = (kt == 0) && (seq > 0);        // the previous tile's stores sit in the vmcnt queue of this K-tile's first wait
            const bool last = !FUSE && kt == nkt - 1;         // this tile's column tiles finish phase by phase: stored in the next phase's load segment
            auto phase = [&](auto Pc) {
                constexpr int P = decltype(Pc)::value;
                constexpr int UA = 0, UL = P8_UNIT, UB = (2 + (P >> 1)) * P8_UNIT;
                constexpr int C0 = ((2 * P) & 3) * 4096, C1 = ((2 * P + 1) & 3) * 4096;      // column tile inside its B unit: [tile][k-step][hi, lo] KiB
                // ---- load segment: operand fragments of this phase, the finished column tiles of the previous phase, one unit ----
                if constexpr (P == 0) {
                    ah0 = p8_lds_read<UA>(sbase + wave * 2048);
                    ah1 = p8_lds_read<UA + 1024>(sbase + wave * 2048);
                    al0 = p8_lds_read<UL>(sbase + wave * 2048);
                    al1 = p8_lds_read<UL + 1024>(sbase + wave * 2048);
                }
                const p8_bf16x8 bh00 = p8_lds_read<UB + C0>(sbase), bl00 = p8_lds_read<UB + C0 + 1024>(sbase);
                const p8_bf16x8 bh01 = p8_lds_read<UB + C0 + 2048>(sbase), bl01 = p8_lds_read<UB + C0 + 3072>(sbase);
                const p8_bf16x8 bh10 = p8_lds_read<UB + C1>(sbase), bl10 = p8_lds_read<UB + C1 + 1024>(sbase);
                const p8_bf16x8 bh11 = p8_lds_read<UB + C1 + 2048>(sbase), bl11 = p8_lds_read<UB + C1 + 3072>(sbase);
                if constexpr (P == 0) {
                    if (pending >= 0) {             // column tiles 6, 7 of the previous tile (finished in its last phase)
                        store_tile(6);
                        store_tile(7);
                        const int n0 = bias_n0(pending + 1);
                        init_tile(6, n0);
                        init_tile(7, n0);
                        pending = -1;
                    }
                } else {
                    if (last) {                     // column tiles 2 P - 2, 2 P - 1 are final since the previous phase
                        if constexpr (P == 1) set_out_tile(seq);
                        store_tile(2 * P - 2);
                        store_tile(2 * P - 1);
                        const int n0 = bias_n0(seq + 1);
                        init_tile(2 * P - 2, n0);
                        init_tile(2 * P - 1, n0);
                    }
                }
                issue((P + D) & 3);
                if constexpr (P == 1) {          // unit 3 of this K-tile (columns 128..255) is read in the next phase
                    if (first) p8_wait_vm<FUSE ? 2 * (D - 2) + 32 : 2 * (D - 2) + 8 * (D - 3)>();
                    else if (last) p8_wait_vm<2 * (D - 2) + 8>();
                    else p8_wait_vm<2 * (D - 2)>();
                }
                if constexpr (P == 3) {          // units 0..2 of the next K-tile
                    if (last) p8_wait_vm<2 * (D - 3) + 8 * (D - 3)>();
                    else p8_wait_vm<2 * (D - 3)>();
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- matrix segment: 2 column tiles x 2 k-steps x 3 products ----
                if constexpr ((DBG & 16) != 0) __builtin_amdgcn_s_setprio(1);      // measured: 423 us without, 440 with
                f32x16& c0 = acc[2 * P];
                f32x16& c1 = acc[2 * P + 1];
                if constexpr ((DBG & 2) != 0) {
                    asm volatile("" ::"v"(bh00), "v"(bl00), "v"(bh01), "v"(bl01), "v"(bh10), "v"(bl10), "v"(bh11), "v"(bl11), "v"(ah0), "v"(ah1), "v"(al0), "v"(al1));
                } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh00, al0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh10, al0, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl00, ah0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl10, ah0, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh00, ah0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh10, ah0, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh01, al1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh11, al1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl01, ah1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl11, ah1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh01, ah1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh11, ah1, c1, 0, 0, 0);
                if constexpr ((DBG & 4) != 0) {       // timing only: the same 12 MFMAs once more (what a 24-MFMA phase would cost per barrier pair)
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh00, al0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh10, al0, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl00, ah0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl10, ah0, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh00, ah0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh10, ah0, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh01, al1, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh11, al1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl01, ah1, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl11, ah1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh01, ah1, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh11, ah1, c1, 0, 0, 0);
                }
                }
                if constexpr ((DBG & 16) != 0) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            };
            phase(P8Int<0>{});
            phase(P8Int<1>{});
            phase(P8Int<2>{});
            phase(P8Int<3>{});
            if (last) pending = seq;
        }
        if constexpr (FUSE) {
            // ---- the W2 "K-tile": unit p (ring slot (cu & 4) + p) = W2 fragments of the block's channels 64 p .. 64 p + 63 ----
            static_assert(!FUSE || D == 6, "the store counts in the waits below are for 6 units in flight");
            const unsigned sbase = lds0 + (cu & 4) * P8_UNIT;
            f32x16 z0, z1;                 // [32 points][clusters 0..31 | 32..63]
            auto sphase = [&](auto Pc) {
                constexpr int P = decltype(Pc)::value;
                constexpr int U = P * P8_UNIT;
                // fragments of k-step 0 of this unit: [k-step][cluster tile][hi, lo] KiB
                p8_bf16x8 wa[4], wb[4];
                wa[0] = p8_lds_read<U>(sbase); wa[1] = p8_lds_read<U + 1024>(sbase);
                wa[2] = p8_lds_read<U + 2048>(sbase); wa[3] = p8_lds_read<U + 3072>(sbase);
                issue((P + D) & 3);
                if constexpr (P == 1) p8_wait_vm<2 * (D - 2) + 8>();          // + the 8 stores of phase 0
                if constexpr (P == 3) p8_wait_vm<2 * (D - 3) + 24>();         // + the stores of phases 0..2
                __builtin_amdgcn_s_barrier();
                if constexpr (P == 0) {
                    set_out_tile(seq);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { z0[r] = 0.0f; z1[r] = 0.0f; }
                }
                auto step = [&](auto Ic, p8_bf16x8 (&w)[4], p8_bf16x8 (&wn)[4]) {
                    constexpr int I = decltype(Ic)::value;
                    constexpr int T = 2 * P + (I >> 1), Q = 4 * (I & 1);       // accumulator tile, first register of the 16-channel group / 2
                    if constexpr (I < 3) {
                        constexpr int O = U + (I + 1) * 4096;
                        wn[0] = p8_lds_read<O>(sbase); wn[1] = p8_lds_read<O + 1024>(sbase);
                        wn[2] = p8_lds_read<O + 2048>(sbase); wn[3] = p8_lds_read<O + 3072>(sbase);
                        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    float x[4], y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[j] = __builtin_amdgcn_fmed3f(acc[T][2 * Q + j], ns * acc[T][2 * Q + j], pinf);
                        y[j] = __builtin_amdgcn_fmed3f(acc[T][2 * Q + 4 + j], ns * acc[T][2 * Q + 4 + j], pinf);
                    }
                    const int c8 = c_n0 + T * 32 + (I & 1) * 16;
                    if constexpr (CPANELS) {
                        *reinterpret_cast<float4*>(crow + (long long)(c8 >> 3) * g.c_panel_ld * 8) = make_float4(x[0], x[1], x[2], x[3]);
                        *reinterpret_cast<float4*>(crow + (long long)((c8 >> 3) + 1) * g.c_panel_ld * 8) = make_float4(y[0], y[1], y[2], y[3]);
                    } else {
                        *reinterpret_cast<float4*>(crow + c8) = make_float4(x[0], x[1], x[2], x[3]);
                        *reinterpret_cast<float4*>(crow + c8 + 8) = make_float4(y[0], y[1], y[2], y[3]);
                    }
                    // the MFMA's column operand of lane (point, half h') is channels 16 s + 8 h' + 0..7: the two halves of a point
                    // exchange one register group each (x of half 1 <-> y of half 0)
                    unsigned kx[4], ky[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[j]), __float_as_uint(y[j]), false, false);
                        kx[j] = r[0];
                        ky[j] = r[1];
                    }
                    unsigned fh[4], fl[4];
                    lpd_split2(__uint_as_float(kx[0]), __uint_as_float(kx[1]), fh[0], fl[0]);
                    lpd_split2(__uint_as_float(kx[2]), __uint_as_float(kx[3]), fh[1], fl[1]);
                    lpd_split2(__uint_as_float(ky[0]), __uint_as_float(ky[1]), fh[2], fl[2]);
                    lpd_split2(__uint_as_float(ky[2]), __uint_as_float(ky[3]), fh[3], fl[3]);
                    const p8_bf16x8 f_hi = __builtin_bit_cast(p8_bf16x8, make_uint4(fh[0], fh[1], fh[2], fh[3]));
                    const p8_bf16x8 f_lo = __builtin_bit_cast(p8_bf16x8, make_uint4(fl[0], fl[1], fl[2], fl[3]));
                    z0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], f_lo, z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], f_lo, z1, 0, 0, 0);
                    z0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], f_hi, z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[3], f_hi, z1, 0, 0, 0);
                    z0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], f_hi, z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], f_hi, z1, 0, 0, 0);
                };
                step(P8Int<0>{}, wa, wb);
                step(P8Int<1>{}, wb, wa);
                step(P8Int<2>{}, wa, wb);
                step(P8Int<3>{}, wb, wa);
                {   // the two column tiles start the next tile from its bias
                    const int n0 = bias_n0(seq + 1);
                    init_tile(2 * P, n0);
                    init_tile(2 * P + 1, n0);
                }
                if constexpr (P == 3) {        // [32 points][64] partial product of this column block: register 4 q + i of z0 / z1 <-> cluster
                    int rt, ct;                // 32 jt + 8 q + 4 h + i of the lane's point
                    tile_of(seq, rt, ct);
                    float* prow = g.parts + (long long)ct * g.part_stride + (long long)(rt * 256 + wave * 32 + col) * 64 + h * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        *reinterpret_cast<float4*>(prow + q * 8) = make_float4(z0[4 * q], z0[4 * q + 1], z0[4 * q + 2], z0[4 * q + 3]);
                        *reinterpret_cast<float4*>(prow + 32 + q * 8) = make_float4(z1[4 * q], z1[4 * q + 1], z1[4 * q + 2], z1[4 * q + 3]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            };
            sphase(P8Int<0>{});
            sphase(P8Int<1>{});
            sphase(P8Int<2>{});
            sphase(P8Int<3>{});
            cu += 4;
        }
    }
    if constexpr (!FUSE) {       // the last tile's column tiles 6, 7
        store_tile(6);
        store_tile(7);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();     // pairs with the second group's last barrier
    p8_wait_vm<0>();                                // no LDS-DMA may outlive the workgroup
}

extern "C" int lpd_gemm_p8_applies(int M, int N, int K, int panel_n)
{
    return (M > 0 && N > 0 && K >= 64 && N % 256 == 0 && K % 32 == 0 && panel_n > 0 && panel_n % 256 == 0 && M % panel_n == 0) ? 1 : 0;
}

// Pricing runs on this kernel (tools/p8_bench.py, B = 32: 131072 x 1024 x 512, one box): full 430 us; without LDS-DMA in the loop 362;
// without MFMAs 315; with neither -- LDS reads, barriers, stores -- 238; every phase's 12 MFMAs issued twice 645 (1.5x for 2x the
// matrix work); without s_setprio around the MFMAs 423 (kept off); nt (aux = 2) loads 487.  A second generation with TWO phases of 24
// MFMAs per K-tile (4 LDS-DMA pieces and 16 - 20 fragment reads per load segment, 10-slot ring in all 160 KiB of LDS, 252 - 256 registers)
// was built on the strength of the 645 us figure and measured 438 us (fused form 520 against 444): the load segment grows with the
// phase -- its LDS-DMA issue (~70 cycles a piece) and fragment reads are what fills the ~270 cycles a barrier interval costs beside
// its MFMAs -- so the ratio does not move; it was removed again.
static int gemm_p8_impl(const void* a_hi, const void* a_lo, long long a_cloud, int a_panel_ld, const void* frags, float* C, int ldc,
                        long long c_cloud, int c_panel_ld, int M, int N, int K, int panel_n, const float* bias, int act, float slope,
                        const void* w2_frags, float* parts, long long part_stride, int impl, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(a_hi && a_lo && frags && C, "lpd_gemm_p8: null pointer");
    LPD_CHECK_ARG(lpd_gemm_p8_applies(M, N, K, panel_n), "lpd_gemm_p8: needs N %% 256 == 0, K %% 32 == 0, K >= 64, points per cloud %% 256 == 0");
    LPD_CHECK_ARG(act >= 0 && act <= 2 && (act != 2 || (slope >= 0.0f && slope <= 1.0f)), "lpd_gemm_p8: act %d slope %g", act, (double)slope);
    LPD_CHECK_ARG(((uintptr_t)bias & 15) == 0, "lpd_gemm_p8: bias must be 16-byte aligned");
    LPD_CHECK_ARG(a_panel_ld >= panel_n && (c_cloud == 0 || c_panel_ld >= panel_n), "lpd_gemm_p8: panel stride < points per cloud");
    LPD_CHECK_ARG((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)frags | (uintptr_t)C) & 15) == 0, "lpd_gemm_p8: operands must be 16-byte aligned");
    LPD_CHECK_ARG(c_cloud != 0 || ldc % 4 == 0, "lpd_gemm_p8: ldc %% 4 != 0");
    LPD_CHECK_ARG(!w2_frags || (parts && part_stride >= (long long)M * 64 && (((uintptr_t)w2_frags | (uintptr_t)parts) & 15) == 0),
                  "lpd_gemm_p8_fused: parts / part_stride");
    P8Args g;
    g.a_hi = (const __bf16*)a_hi;
    g.a_lo = (const __bf16*)a_lo;
    g.KS = (K + 15) / 16;
    g.fhi = (const __bf16*)frags;
    g.flo = g.fhi + (long long)((N + 31) / 32) * g.KS * 512;
    g.C = C;
    g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.a_cloud = a_cloud; g.c_cloud = c_cloud;
    g.panel_n = panel_n; g.a_panel_ld = a_panel_ld; g.c_panel_ld = c_panel_ld; g.ldc = ldc;
    g.act = act; g.slope = slope;
    g.tiles_m = M / 256; g.tiles_n = N / 256;
    g.w2hi = (const __bf16*)w2_frags;
    g.w2lo = g.w2hi ? g.w2hi + (long long)2 * (N / 16) * 512 : nullptr;      // prep_b of [N][64]: 2 column tiles x N / 16 k-steps
    g.parts = parts;
    g.part_stride = part_stride;
    const long long ntiles = (long long)g.tiles_m * g.tiles_n;
    int grid = 256;                                   // one workgroup per CU (128 KiB of LDS each)
    while (grid > 8 && grid / 2 >= ntiles) grid /= 2;
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(P8_THREADS), P8_LDS_BYTES, stream, g);
    };
    const bool d5 = (impl & 7) == 5;
#ifdef LPD_P8_BENCH
    // timing-only variants (tools/p8_bench.py builds its own library with -DLPD_P8_BENCH): their RESULTS ARE GARBAGE, so the product
    // library does not contain them
    if (!w2_frags && (impl & ~7)) {
        if (impl & 128) launch(gemm_p8_kernel<6, false, 32>);
        else if ((impl & 127) == 96) launch(gemm_p8_kernel<6, false, 16>);
        else if (impl & 64) launch(gemm_p8_kernel<6, false, 4>);
        else if ((impl & 24) == 24) launch(gemm_p8_kernel<6, false, 3>);
        else if (impl & 32) launch(gemm_p8_kernel<6, false, 8>);
        else if ((impl & 24) == 8) launch(gemm_p8_kernel<6, false, 1>);
        else if ((impl & 16) && d5) launch(gemm_p8_kernel<5, false, 2>);
        else launch(gemm_p8_kernel<6, false, 2>);
        LPD_CHECK_LAUNCH("lpd_gemm_p8");
        return LPD_OK;
    }
#else
    LPD_CHECK_ARG(impl == 0 || impl == 5 || impl == 6, "lpd_gemm_p8: impl %d is not built (0 / 6 = six units of lookahead, 5 = five)", impl);
#endif
    if (w2_frags) {
        if (c_cloud != 0) launch(gemm_p8_kernel<6, true, 0, true>); else launch(gemm_p8_kernel<6, false, 0, true>);
    }
    else if (c_cloud != 0) { if (d5) launch(gemm_p8_kernel<5, true>); else launch(gemm_p8_kernel<6, true>); }
    else { if (d5) launch(gemm_p8_kernel<5, false>); else launch(gemm_p8_kernel<6, false>); }
    LPD_CHECK_LAUNCH("lpd_gemm_p8");
    return LPD_OK;
}

extern "C" int lpd_gemm_p8(const void* a_hi, const void* a_lo, long long a_cloud, int a_panel_ld, const void* frags, float* C, int ldc,
                           long long c_cloud, int c_panel_ld, int M, int N, int K, int panel_n, const float* bias,
                           int act, float slope, int impl, void* stream_)
{
    return gemm_p8_impl(a_hi, a_lo, a_cloud, a_panel_ld, frags, C, ldc, c_cloud, c_panel_ld, M, N, K, panel_n, bias, act, slope, nullptr,
                        nullptr, 0, impl, stream_);
}

// conv3 + the NetVLAD assignment product in one launch: besides C, parts[j][m][0..64) = C[m][256 j .. 256 j + 255] . W2[256 j ..][0..64)
// for the N / 256 column blocks j (part_stride floats between the planes; their sum over j is C . W2, util/PointNetVlad.py:48).
// w2_frags: lpd_gemm_prep_b(W2 [N][64], ldb, b_kmajor = 1, 64, N).
extern "C" int lpd_gemm_p8_fused(const void* a_hi, const void* a_lo, long long a_cloud, int a_panel_ld, const void* frags, float* C, int ldc,
                                 long long c_cloud, int c_panel_ld, int M, int N, int K, int panel_n, const float* bias, int act,
                                 float slope, const void* w2_frags, float* parts, long long part_stride, void* stream_)
{
    LPD_CHECK_ARG(w2_frags && parts, "lpd_gemm_p8_fused: null pointer");
    return gemm_p8_impl(a_hi, a_lo, a_cloud, a_panel_ld, frags, C, ldc, c_cloud, c_panel_ld, M, N, K, panel_n, bias, act, slope, w2_frags,
                        parts, part_stride, 0, stream_);
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
