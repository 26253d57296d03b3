// lpd_common.h -- shared helpers for the gfx950 kernels behind the C-ABI in include/lpd_hip.h.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, f32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define LPD_OK 0
#define LPD_ERR_ARG (-1)
#define LPD_ERR_LAUNCH (-2)
#define LPD_ERR_UNSUPPORTED (-3)

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

// activation codes shared by every epilogue (include/lpd_hip.h LPD_ACT_*)
__device__ __forceinline__ float lpd_act(float v, int act, float slope)
{
    switch (act) {
        case 1: return v > 0.0f ? v : 0.0f;             // ReLU
        case 2: return v > 0.0f ? v : v * slope;         // LeakyReLU(slope)
        case 3: return 1.0f / (1.0f + __expf(-v));       // sigmoid
        default: return v;
    }
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
