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
