// lpd_morton.hip -- per-cloud Morton (Z-order) reordering of the input points.
//
// No counterpart in the reference: it feeds clouds in file order (random after the Oxford
// down-sampling, loading_pointclouds.py:26-35).  The descriptor is invariant to point order
// (max over neighbours, sum over points, BatchNorm statistics), so the points
// are sorted along a Z-order curve once per forward; afterwards the k neighbours of consecutive
// points live in a few nearby rows and the gathers of the kNN-aggregation kernels
// (lpd_edge.hip) are served by L1/L2 instead of the Infinity Cache.
//
// One 1024-thread block per cloud: bounding box -> 10 bits per axis -> 30-bit Morton code -> bitonic sort of 32-bit words
// (top bits of the code | index) in registers / lane shuffles / LDS -> permuted copy of the cloud.  N <= 16384 (64 KiB of LDS).
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
__device__ __forceinline__ uint32_t bitonic_keep(uint32_t v, uint32_t o, bool lower, bool up)
{
    const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
    return (lower == up) ? mn : mx;
}

// E elements per thread (NP = 1024 E keys): thread t owns the E consecutive positions E t .. E t + E - 1 of the network.
// Sort words are 32 bits (round 5): the top 32 - IB bits of the 30-bit Morton code above the IB = log2(NP) index bits -- one
// shuffle, one compare and one select per exchange instead of two / a 64-bit pair (the kernel is instruction-bound on the ONE CU a
// cloud's block runs on: 33 us at N = 4096 whatever the batch, at the head of every forward).  20 code bits at N = 4096 are a
// 128 x 128 x 64 grid over the bounding box, ~0.004 points per cell: a prefix of the same Z-curve; points that share a cell keep their
// input order (the index breaks the tie), so the order is deterministic.  Strides below E exchange inside the thread's registers,
// strides below 64 E between lanes, only the strides that cross waves (10 of the 78 steps at N = 4096) go through LDS.
template <int E, int T = 1024>
__global__ __launch_bounds__(T) void morton_sort_kernel(const float* __restrict__ xyz, float* __restrict__ out,
                                                         int32_t* __restrict__ perm, int N)
{
    constexpr int NP = T * E;
    constexpr int IB = NP == 1024 ? 10 : NP == 2048 ? 11 : NP == 4096 ? 12 : NP == 8192 ? 13 : 14;      // index bits
    static_assert(NP >= 1024 && NP <= 16384 && (NP & (NP - 1)) == 0, "1024 .. 16384 keys");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [E][T]
    __shared__ float red[6][16];
    __shared__ float box[6];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < N; i += T)
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
        for (int w = 1; w < T / 64; ++w) { a = fminf(a, red[tid][w]); z = fmaxf(z, red[3 + tid][w]); }
        box[tid] = a;
        box[3 + tid] = z > a ? 1023.0f / (z - a) : 0.0f;
    }
    __syncthreads();
    uint32_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        uint32_t word = 0xffffffffu;                         // padding sorts to the end
        if (i < N) {
            uint32_t q[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = (p[i * 3 + c] - box[c]) * box[3 + c];
                t = fminf(fmaxf(t, 0.0f), 1023.0f);
                q[c] = (uint32_t)t;
            }
            const uint32_t key = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);      // 30 bits
            word = ((key >> (IB - 2)) << IB) | (uint32_t)i;                                          // its top 32 - IB bits | index
        }
        v[e] = word;
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
                        const uint32_t a = v[e], c = v[e | j];
                        const bool sw = (a > c) == up;
                        v[e] = sw ? c : a;
                        v[e | j] = sw ? a : c;
                    }
            } else if (j < 64 * E) {                         // partner in another lane of this wave
                const int lj = j / E;
                const bool lower = (tid & lj) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    v[e] = bitonic_keep(v[e], __shfl_xor(v[e], lj, 64), lower, ((tid * E + e) & k) == 0);
            } else {                                         // partner in another wave: through LDS
                const int tj = j / E;
                const bool lower = (tid & tj) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e) lds[e * T + tid] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e)
                    v[e] = bitonic_keep(v[e], lds[e * T + (tid ^ tj)], lower, ((tid * E + e) & k) == 0);
                __syncthreads();
            }
        }
    }
    float* o = out + (size_t)b * N * 3;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int r = tid * E + e;
        if (r < N) {
            const uint32_t src = v[e] & ((1u << IB) - 1u);
            o[r * 3 + 0] = p[src * 3 + 0];
            o[r * 3 + 1] = p[src * 3 + 1];
            o[r * 3 + 2] = p[src * 3 + 2];
            if (perm) perm[(size_t)b * N + r] = (int32_t)src;
        }
    }
}

template <int E, int T = 1024>
void morton_launch(const float* xyz, float* out, int32_t* perm, int B, int N, hipStream_t stream)
{
    const size_t lds = (size_t)E * T * sizeof(uint32_t);
    auto kern = morton_sort_kernel<E, T>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(T), lds, stream, xyz, out, perm, N);
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
    else if (N <= 4096) morton_launch<4>(xyz, out, perm, B, N, stream);      // (8 keys x 512 threads: 24 us against 23; 16 x 256: 34)
    else if (N <= 8192) morton_launch<8>(xyz, out, perm, B, N, stream);
    else morton_launch<16>(xyz, out, perm, B, N, stream);
    LPD_CHECK_LAUNCH("lpd_morton_sort");
    return LPD_OK;
}
