// lpd_common.h -- shared helpers for the gfx950 kernels behind the C-ABI in include/lpd_hip.h.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, f32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// The public C-ABI: every extern "C" definition in csrc/ is compiled against its declaration, so a prototype that drifts
// from the definition is a compile error (conflicting types), not a corrupted call in a C caller.
#include "../../include/lpd_hip.h"

// thread-local last-error text (lpd_last_error()); defined in lpd_abi.hip
void lpd_set_error(const char* fmt, ...);

// A/B ablation switches: ONE environment variable, LPD_DEBUG = "name,no-name,name=value,..." (read once; README.md lists the names;
// lpdnet_hip/_debug.py reads the same variable).  Value of `name` (a bare name is 1, no-name is 0), else dflt.  For timing comparisons
// and tests of superseded paths only: the product never needs it set.
int lpd_debug(const char* name, int dflt);

#define LPD_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            lpd_set_error(__VA_ARGS__);   \
            return LPD_ERR_ARG;           \
        }                                 \
    } while (0)

// Kernels only enqueue; a launch error is reported, nothing synchronises.
#define LPD_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            lpd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return LPD_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// activation codes shared by every epilogue (include/lpd_hip.h LPD_ACT_*): 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 sigmoid.
// The piecewise-linear three are ONE branch-free expression,  max(v,0) + ns * min(v,0)  with ns = 1 / 0 / slope (exact:
// one of the two terms is +-0).  The sigmoid costs an exp and a divide: written as a switch the compiler if-converted it
// into every element of every kernel (80 v_exp_f32 in the edge-MLP tile builder), so it is a separate function that
// only the kernels with a sigmoid epilogue call, behind a branch on the (uniform) activation code.
__device__ __forceinline__ float lpd_neg_slope(int act, float slope) { return act == 0 ? 1.0f : (act == 1 ? 0.0f : slope); }
__device__ __forceinline__ float lpd_act_pl(float v, float ns) { return fmaxf(v, 0.0f) + ns * fminf(v, 0.0f); }
__device__ __forceinline__ float lpd_sigmoid(float v)
{
    asm volatile("" ::: "memory");   // not speculatable: keeps the exp + divide inside the branch that needs them
    return 1.0f / (1.0f + __expf(-v));
}

// acts 0..2 only (hosts reject the sigmoid where it is not built)
__device__ __forceinline__ float lpd_act(float v, int act, float slope) { return lpd_act_pl(v, lpd_neg_slope(act, slope)); }

// any activation; the caller keeps `act` uniform so that the branch is scalar
__device__ __forceinline__ float lpd_act_any(float v, int act, float slope)
{
    if (act == 3) return lpd_sigmoid(v);
    return lpd_act_pl(v, lpd_neg_slope(act, slope));
}

// ---- column statistics across blocks -------------------------------------------------------------------------------------
// Every statistics / reduction kernel of the training path ends with each block adding its 2 x C fp64 partial sums to the
// SAME 2 x C doubles.  Device-scope fp64 atomics on one 128-byte line are served one after the other: 4096 blocks x 256 addresses
// in 16 lines = 330 us of drain behind an [M, C] pass that takes 60 us (tools/train_profile.py, bn_sel_bwd_reduce: 375 us).
// The blocks therefore add into one of 32 REPLICAS (blockIdx % 32: thirty-two times the lines, 1/32 of the queue on each), and a
// small second launch (lpd_stat_finish) sums the replicas into the caller's 2 x C doubles and clears them.  (A ticket instead
// of the second launch -- the block that draws the last one sums the replicas -- costs more than it saves: 4096 returning
// atomics on ONE address behind a __threadfence took 620 us.)  The replicas live in a CALLER-OWNED workspace (the `stat_ws` argument of every entry point
// that reduces over blocks: lpd_stat_ws_bytes() bytes, zero-filled once by the caller, left all-zero by every call that returns
// LPD_OK; one workspace per stream -- launches on one stream are ordered, so they share it).  The library allocates nothing.
#define LPD_STAT_REPLICAS 32
#define LPD_STAT_CMAX 1024
struct LpdStatWs {
    double* rep;             // [LPD_STAT_REPLICAS][2][LPD_STAT_CMAX], all zero between (statistics kernel, lpd_stat_finish) pairs
    double* sum() const { return rep; }                      // what the kernels take in place of the caller's sum / sumsq pointers
    double* sumsq() const { return rep + LPD_STAT_CMAX; }
};
inline LpdStatWs lpd_stat_arg(double* stat_ws) { return LpdStatWs{stat_ws}; }      // the caller's workspace (may be null: checked by the entry points)
// every entry point that hands ws.sum() / ws.sumsq() to a kernel checks its column count first: column c of a replica sits at + c, the
// second array LPD_STAT_CMAX doubles behind the first -- a wider layer would add into the other array and past the last replica
#define LPD_CHECK_STAT_COLS(name, ncols) \
    LPD_CHECK_ARG((ncols) > 0 && (ncols) <= LPD_STAT_CMAX, "%s: %d columns exceed the %d of the statistics workspace", name, (int)(ncols), LPD_STAT_CMAX)
// o0[c] = sum over the replicas of column c (c < ncols <= LPD_STAT_CMAX), o1 likewise; clears the replicas
int lpd_stat_finish(LpdStatWs ws, double* o0, double* o1, int ncols, hipStream_t stream);
// Grid of a kernel that ENDS with such a block-level reduction + atomics: every block sends 2 x C fp64 atomics to the replicas, so the
// block count is the atomic count.  At the 4096 blocks these kernels used to launch (16 per CU) the bn3 backward's reduction pass spent a
// quarter of its time there: 4096 / 768 / 384 blocks = 502 / 446 / 440 us for the two passes over the [180224, 1024] bf16 map
// (round 6).  Three blocks per CU keep enough loads in flight for a streaming read (each thread has 4-8 sixteen-byte loads open).
inline int lpd_reduce_grid(long long blocks_wanted)
{
    static const int cap = lpd_debug("reduce-grid", 768);
    if (blocks_wanted > cap) blocks_wanted = cap;
    return blocks_wanted < 1 ? 1 : (int)blocks_wanted;
}
// offset of this block's replica, to be added to the column index of both pointers
__device__ __forceinline__ size_t lpd_stat_rofs() { return (size_t)(blockIdx.x % LPD_STAT_REPLICAS) * 2 * LPD_STAT_CMAX; }

// Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md "Workgroup dispatch").
// Map a linear block id so that each XCD owns a contiguous range of work items; bijective for any n.
__device__ __forceinline__ int lpd_xcd_remap(int bid, int nblocks)
{
    const int nx = 8;
    int q = nblocks / nx, r = nblocks % nx;
    int xcd = bid % nx, slot = bid / nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// Grid-stride walk over `nwork` wave-sized items for the row-gather kernels of the training path, XCD-aware.  Blocks b, b + 8, ... share
// an XCD and its L2.  With the plain walk (item = global wave id, stride = all waves) the blocks resident at one time cover ONE window
// of consecutive rows that is spread over all eight XCDs, so every L2 fills with the same gathered neighbour rows: SN1's backward gather
// moved 2.85 GB through the fabric for ~1 GB of distinct bytes (rocprofv3 FETCH_SIZE, profiles/r05_pmc_train_bf16.txt).  Here each XCD
// owns one contiguous range of items and its resident blocks sweep that range together (the walk lpd_edge_gather_max_kernel has used
// since round 1): the rows a window gathers -- Z-ordered clouds: neighbours are nearby rows -- are fetched into ONE L2.
struct LpdXcdSweep { long long begin, end, step; };
__device__ __forceinline__ LpdXcdSweep lpd_xcd_sweep(long long nwork)
{
    // a launch of fewer than 8 blocks occupies only that many XCDs: the range is cut into as many shares as there are XCDs WITH a
    // block (with 8 fixed shares the shares of the empty XCDs were never visited: rows of dP / dQ left unwritten for M <= 56 at C = 256)
    const int nx = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int wpb = blockDim.x >> 6;
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const long long blocks_here = gridDim.x / nx + (xcd < (int)(gridDim.x % nx) ? 1 : 0);      // blocks of this launch on this XCD
    const long long per_xcd = (nwork + nx - 1) / nx;
    LpdXcdSweep s;
    const long long b = xcd * per_xcd;
    s.end = b + per_xcd < nwork ? b + per_xcd : nwork;
    s.begin = b + (long long)slot * wpb + (threadIdx.x >> 6);
    s.step = blocks_here * wpb;
    return s;
}

// ---- split-bf16 activation planes (what lpd_gemm_p8 / lpd_gemm_x3t read): x = hi + lo, hi = bf16(x), lo = bf16(x - hi) ----
typedef __bf16 lpd_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lpd_f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed hi pair and packed lo pair (element 0 in the low half of the dword)
__device__ __forceinline__ void lpd_split2(float x0, float x1, unsigned& hi, unsigned& lo)
{
    const lpd_bf16x2 h2 = __builtin_convertvector((lpd_f32x2){x0, x1}, lpd_bf16x2);
    hi = __builtin_bit_cast(unsigned, h2);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    const lpd_bf16x2 l2 = __builtin_convertvector((lpd_f32x2){r0, r1}, lpd_bf16x2);
    lo = __builtin_bit_cast(unsigned, l2);
}

// the value of lane (id ^ 1) (DPP quad_perm [1,0,3,2]: no LDS traffic)
__device__ __forceinline__ unsigned lpd_lane_xor1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

// Two adjacent lanes (even: channels c..c+3, odd: c+4..c+7 of one point) hold a float4 each; afterwards the EVEN lane holds the 8
// hi values and the ODD lane the 8 lo values of the point's 8 channels as one 16-byte word: each lane stores 16 bytes into "its"
// plane (even -> hi plane, odd -> lo plane) at the point's row.
__device__ __forceinline__ uint4 lpd_split8_pair(const float4& r, int odd)
{
    unsigned h0, l0, h1, l1;
    lpd_split2(r.x, r.y, h0, l0);
    lpd_split2(r.z, r.w, h1, l1);
    const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;       // what the partner needs
    const unsigned r0 = lpd_lane_xor1(s0), r1 = lpd_lane_xor1(s1);
    return odd ? make_uint4(r0, r1, l0, l1) : make_uint4(h0, h1, r0, r1);
}
