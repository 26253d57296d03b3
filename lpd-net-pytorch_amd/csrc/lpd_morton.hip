// lpd_morton.hip -- per-cloud Morton (Z-order) reordering of the input points.
//
// No counterpart in the reference: it feeds clouds in file order (random after the Oxford
// down-sampling, loading_pointclouds.py:26-35).  The descriptor is invariant to point order
// (max over neighbours, sum over points, BatchNorm statistics), so the points
// are sorted along a Z-order curve once per forward; afterwards the k neighbours of consecutive
// points live in a few nearby rows and the gathers of the kNN-aggregation kernels
// (lpd_edge.hip) are served by L1/L2 instead of the Infinity Cache.
//
// One 1024-thread block per cloud: bounding box -> 10 bits per axis -> 30-bit key -> bitonic sort of
// (key, index) in LDS -> permuted copy of the cloud.  N <= 16384 (128 KiB of LDS).
#include "lpd_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

__global__ __launch_bounds__(1024) void morton_sort_kernel(const float* __restrict__ xyz, float* __restrict__ out,
                                                            int32_t* __restrict__ perm, int N, int NP)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t* keys = sm;          // [NP]
    uint32_t* vals = sm + NP;     // [NP]
    __shared__ float red[6][16];
    __shared__ float box[6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = p[i * 3 + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
        if ((tid & 63) == 0) { red[c][tid >> 6] = mn[c]; red[3 + c][tid >> 6] = mx[c]; }
    }
    __syncthreads();
    if (tid < 3) {
        float a = red[tid][0], z = red[3 + tid][0];
        for (int w = 1; w < 16; ++w) { a = fminf(a, red[tid][w]); z = fmaxf(z, red[3 + tid][w]); }
        box[tid] = a;
        box[3 + tid] = z > a ? 1023.0f / (z - a) : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < NP; i += 1024) {
        uint32_t key = 0xffffffffu;
        if (i < N) {
            uint32_t q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = (p[i * 3 + c] - box[c]) * box[3 + c];
                t = fminf(fmaxf(t, 0.0f), 1023.0f);
                q[c] = (uint32_t)t;
            }
            key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        }
        keys[i] = key;
        vals[i] = (uint32_t)i;
    }
    __syncthreads();
    // bitonic sort, ascending by (key, index): deterministic for equal keys
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP; i += 1024) {
                int l = i ^ j;
                if (l > i) {
                    uint32_t ki = keys[i], kl = keys[l], vi = vals[i], vl = vals[l];
                    bool up = (i & k) == 0;
                    bool gt = (ki > kl) || (ki == kl && vi > vl);
                    if (gt == up) { keys[i] = kl; keys[l] = ki; vals[i] = vl; vals[l] = vi; }
                }
            }
            __syncthreads();
        }
    }
    float* o = out + (size_t)b * N * 3;
    for (int r = tid; r < N; r += 1024) {
        uint32_t src = vals[r];
        o[r * 3 + 0] = p[src * 3 + 0];
        o[r * 3 + 1] = p[src * 3 + 1];
        o[r * 3 + 2] = p[src * 3 + 2];
        if (perm) perm[(size_t)b * N + r] = (int32_t)src;
    }
}

}  // namespace

extern "C" int lpd_morton_sort(const float* xyz, float* out, int32_t* perm, int B, int N, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(xyz && out && B > 0 && N > 0, "lpd_morton_sort: bad arguments");
    LPD_CHECK_ARG(xyz != out, "lpd_morton_sort: in-place reordering is not supported");
    if (N > 16384) {
        lpd_set_error("lpd_morton_sort: N=%d > 16384 unsupported", N);
        return LPD_ERR_UNSUPPORTED;
    }
    int NP = 1;
    while (NP < N) NP <<= 1;
    size_t lds = (size_t)NP * 2 * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)morton_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(morton_sort_kernel, dim3(B), dim3(1024), lds, stream, xyz, out, perm, N, NP);
    LPD_CHECK_LAUNCH("lpd_morton_sort");
    return LPD_OK;
}
