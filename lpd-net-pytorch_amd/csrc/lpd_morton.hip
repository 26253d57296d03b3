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
// (key, index) in registers / lane shuffles / LDS -> permuted copy of the cloud.  N <= 16384 (128 KiB of LDS).
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

// One compare-exchange partner step of the bitonic network for an element held by this thread: `o` is the partner's
// value, `lower` says whether this element has the smaller index of the pair, `up` the direction of the pair's block.
__device__ __forceinline__ uint64_t bitonic_keep(uint64_t v, uint64_t o, bool lower, bool up)
{
    const uint64_t mn = v < o ? v : o, mx = v < o ? o : v;
    return (lower == up) ? mn : mx;
}

// E elements per thread (NP = 1024 E keys): thread t owns the E consecutive positions E t .. E t + E - 1 of the network.
// Keys are (morton << 32 | index): one 64-bit compare orders by (key, index), so equal keys keep a deterministic order.
// Strides below E exchange inside the thread's registers, strides below 64 E between lanes (two 32-bit shuffles), only
// the strides that cross waves (10 of the 78 steps at N = 4096) go through LDS -- the first version ran every step
// through LDS with a block barrier and half of the threads idle (84 us at B = 32, on 32 of the 256 CUs).
template <int E>
__global__ __launch_bounds__(1024) void morton_sort_kernel(const float* __restrict__ xyz, float* __restrict__ out,
                                                            int32_t* __restrict__ perm, int N)
{
    constexpr int NP = 1024 * E;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];   // [E][1024]
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
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
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
        v[e] = ((uint64_t)key << 32) | (uint32_t)i;
    }
    // bitonic sort, ascending
#pragma unroll
    for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {                                     // both elements in this thread
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((e & j) == 0) {
                        const bool up = ((tid * E + e) & k) == 0;
                        const uint64_t a = v[e], c = v[e | j];
                        const bool sw = (a > c) == up;
                        v[e] = sw ? c : a;
                        v[e | j] = sw ? a : c;
                    }
            } else if (j < 64 * E) {                         // partner in another lane of this wave
                const int lj = j / E;
                const bool lower = (tid & lj) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t lo = __shfl_xor((uint32_t)v[e], lj, 64), hi = __shfl_xor((uint32_t)(v[e] >> 32), lj, 64);
                    v[e] = bitonic_keep(v[e], ((uint64_t)hi << 32) | lo, lower, ((tid * E + e) & k) == 0);
                }
            } else {                                         // partner in another wave: through LDS
                const int tj = j / E;
                const bool lower = (tid & tj) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e) lds[e * 1024 + tid] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e)
                    v[e] = bitonic_keep(v[e], lds[e * 1024 + (tid ^ tj)], lower, ((tid * E + e) & k) == 0);
                __syncthreads();
            }
        }
    }
    float* o = out + (size_t)b * N * 3;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int r = tid * E + e;
        if (r < N) {
            const uint32_t src = (uint32_t)v[e];
            o[r * 3 + 0] = p[src * 3 + 0];
            o[r * 3 + 1] = p[src * 3 + 1];
            o[r * 3 + 2] = p[src * 3 + 2];
            if (perm) perm[(size_t)b * N + r] = (int32_t)src;
        }
    }
}

template <int E>
void morton_launch(const float* xyz, float* out, int32_t* perm, int B, int N, hipStream_t stream)
{
    const size_t lds = (size_t)E * 1024 * sizeof(uint64_t);
    auto kern = morton_sort_kernel<E>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(1024), lds, stream, xyz, out, perm, N);
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
    if (N <= 1024) morton_launch<1>(xyz, out, perm, B, N, stream);
    else if (N <= 2048) morton_launch<2>(xyz, out, perm, B, N, stream);
    else if (N <= 4096) morton_launch<4>(xyz, out, perm, B, N, stream);
    else if (N <= 8192) morton_launch<8>(xyz, out, perm, B, N, stream);
    else morton_launch<16>(xyz, out, perm, B, N, stream);
    LPD_CHECK_LAUNCH("lpd_morton_sort");
    return LPD_OK;
}
