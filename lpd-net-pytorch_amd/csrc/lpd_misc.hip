// lpd_misc.hip -- the small bandwidth-bound pieces around the MFMA kernels.
//
//   lpd_linear_smallk   per-point linear layer with K <= 8 inputs (the 3 -> 64 first layers:
//                       util/lpdnet_model.py:185,231 conv1_lpd; util/PointNetVlad.py:190,213 conv1
//                       Conv2d(1,64,(1,3)); T-Net conv1 lpdnet_model.py:276) + affine + activation
//   lpd_transpose       batched [R][C] -> [C][R] (point-major <-> the reference's channel-major
//                       [B,C,N] layout that `knn` consumes, lpdnet_model.py:212,317)
//   lpd_softmax_affine  NetVLAD soft-assignment: softmax_c(scale_c * a_c + shift_c)
//                       (util/PointNetVlad.py:51-58: bn1 over B*N rows, then softmax over clusters)
//   lpd_vlad_finalize   a = a_sum * cluster_weights2; vlad - a; intra-normalise per cluster over the
//                       feature axis; flatten f*K+c; L2 normalise  (PointNetVlad.py:61-74)
//   lpd_colmax          per-cloud column max over the N points (MaxPool2d((num_points,1)) of STN3d
//                       PointNetVlad.py:137,162 and torch.max(x, 2) of TranformNet lpdnet_model.py:300)
//   lpd_mul             context gating product x * gates (PointNetVlad.py:113)
#include "lpd_common.h"
#include <math.h>

namespace {

// W element (n, c) of weight set b lives at W[b * w_sb + n * w_sn + c * w_sk]; rows m use set m / rows_per_w
__global__ void linear_smallk_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                     int w_sn, int w_sk, long long w_sb, int rows_per_w,
                                     float* __restrict__ Y, int ldy, int M, int N, int K, const float* bias,
                                     const float* scale, const float* shift, int act, float slope)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)M * N) return;
    const int m = (int)(e / N), n = (int)(e - (long long)m * N);
    const float* x = X + (size_t)m * ldx;
    const float* w = W + (rows_per_w > 0 ? (long long)(m / rows_per_w) * w_sb : 0) + (size_t)n * w_sn;
    float v = 0.0f;
    for (int c = 0; c < K; ++c) v = fmaf(x[c], w[(size_t)c * w_sk], v);
    if (bias) v += bias[n];
    if (scale) v = v * scale[n] + shift[n];
    Y[(size_t)m * ldy + n] = lpd_act_any(v, act, slope);
}

// in [batch][R][C] (row stride ldi) -> out [batch][C][R] (row stride ldo)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C, int ldi, int ldo,
                                 long long si, long long so)
{
    __shared__ float tile[32][33];
    const float* src = in + (long long)blockIdx.z * si;
    float* dst = out + (long long)blockIdx.z * so;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * ldi + c] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ldo + r] = tile[tx][i];
    }
}

// one wavefront per row; ncols <= 64
__global__ void softmax_affine_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int ncols,
                                      const float* scale, const float* shift)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v = -INFINITY;
    if (lane < ncols) {
        v = in[(size_t)row * ncols + lane];
        if (scale) v = v * scale[lane] + shift[lane];
    }
    float mx = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = lane < ncols ? expf(v - mx) : 0.0f;
    float s = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane < ncols) out[(size_t)row * ncols + lane] = e / s;
}

// The same with the per-cloud column sums of the result (NetVLAD's a_sum, util/PointNetVlad.py:63) accumulated on the way:
// one wavefront walks SM_RPW consecutive rows of one cloud (lane = column), four rows in flight, and adds its 64 partial
// sums to colsum[cloud][lane] with one atomic instruction -- the separate a_sum pass re-read the 33 MB of assignments
// (28 us at B = 32; this kernel: the softmax's own time).
constexpr int SM_RPW = 16;
__global__ __launch_bounds__(256) void softmax_affine_colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                                    int ncols, const float* scale, const float* shift,
                                                                    int group_rows, float* __restrict__ colsum, int colsum_ld)
{
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * SM_RPW;
    if (row0 >= rows) return;
    const bool on = lane < ncols;
    float sc = 1.0f, sh = 0.0f;
    if (scale && on) { sc = scale[lane]; sh = shift[lane]; }
    float acc = 0.0f;
#pragma unroll 4
    for (int r = 0; r < SM_RPW; ++r) {
        const long long row = row0 + r;       // rows % SM_RPW == 0 (host check)
        float v = on ? in[row * ncols + lane] * sc + sh : -INFINITY;
        float mx = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float e = on ? expf(v - mx) : 0.0f;
        float s = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float p = e / s;
        if (on) out[row * ncols + lane] = p;
        acc += p;
    }
    if (on) atomicAdd(&colsum[(row0 / group_rows) * colsum_ld + lane], acc);
}

// 64 columns: sixteen lanes per row (float4 each), four rows per wave-instruction -- 1 KiB loads / stores and four
// shuffle steps per reduction instead of 256-byte accesses and six (36 -> 2x faster at 131072 rows).
// parts > 1: the input is the SUM of `parts` planes, part_stride floats apart (the per-column-block partial products of
// lpd_gemm_p8_fused)
template <int PARTS = 1>
__global__ __launch_bounds__(256) void softmax_affine_colsum64_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                                      const float* scale, const float* shift, int group_rows,
                                                                      float* __restrict__ colsum, int colsum_ld, long long part_stride = 0)
{
    const int lane = threadIdx.x & 63;
    const int q = lane & 15, sub = lane >> 4;                  // column quad, row inside the group of four
    const long long row0 = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * SM_RPW;   // rows % (4 SM_RPW) == 0
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) { sc = *reinterpret_cast<const float4*>(scale + 4 * q); sh = *reinterpret_cast<const float4*>(shift + 4 * q); }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < SM_RPW; r += 4) {
        const long long row = row0 + r + sub;
        float4 pv[PARTS];
#pragma unroll
        for (int pz = 0; pz < PARTS; ++pz) pv[pz] = *reinterpret_cast<const float4*>(in + pz * part_stride + row * 64 + 4 * q);   // all in flight
        float4 v = pv[0];
#pragma unroll
        for (int pz = 1; pz < PARTS; ++pz) { v.x += pv[pz].x; v.y += pv[pz].y; v.z += pv[pz].z; v.w += pv[pz].w; }
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float4 e = make_float4(expf(v.x - mx), expf(v.y - mx), expf(v.z - mx), expf(v.w - mx));
        float s = (e.x + e.y) + (e.z + e.w);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        e.x /= s; e.y /= s; e.z /= s; e.w /= s;
        *reinterpret_cast<float4*>(out + row * 64 + 4 * q) = e;
        acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    // one 256-byte atomic instruction per workgroup (its 4 x SM_RPW rows lie in one cloud): every wave of a cloud adds
    // into the same 64 floats, and the first version's four 16-lane atomics per WAVE cost more than the softmax itself
    __shared__ float part[4][64];
    if (sub == 0) *reinterpret_cast<float4*>(&part[threadIdx.x >> 6][4 * q]) = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        atomicAdd(colsum + (row0 / group_rows) * colsum_ld + lane, t);
    }
}

// block reduce helper (sum) over 256 threads
__device__ __forceinline__ float block_sum_256(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// NetVLAD finalize in three multi-block passes (grid y = cloud), vraw [B][F][KC] (index f*KC + c), act [B][N][KC]:
//   1. a_sum[b][c] = sum_n act[b][n][c]                         (atomics into ws[b][0..KC))
//   2. r = vraw - a_sum * cw2 -> out ; column sums of squares    (atomics into ws[b][KC..2KC))
//   3. out *= inv_c[c] * inv_g, inv_c = 1/max(|r[:,c]|, eps), inv_g = 1/max(sqrt(sum_c |r[:,c]|^2 inv_c^2), eps)
// (after the intra-normalisation every non-degenerate column has unit norm, so the global norm follows from the
//  column norms alone -- no third reduction over the 65536 values).
template <int KC>
__global__ __launch_bounds__(256) void vlad_asum_kernel(const float* __restrict__ act, float* __restrict__ ws, int N)
{
    __shared__ float part[256];
    constexpr int RPB = 256 / KC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c = tid % KC, rg = tid / KC;
    const int chunk = (N + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * chunk, n1 = min(n0 + chunk, N);
    const float* a = act + (size_t)b * N * KC;
    float s = 0.0f;
    for (int n = n0 + rg; n < n1; n += RPB) s += a[(size_t)n * KC + c];
    part[tid] = s;
    __syncthreads();
    if (tid < KC) {
        float t = 0.0f;
        for (int r = 0; r < RPB; ++r) t += part[r * KC + tid];
        atomicAdd(&ws[(size_t)b * 2 * KC + tid], t);
    }
}

template <int KC>
__global__ __launch_bounds__(256) void vlad_resid_kernel(const float* __restrict__ vraw, const float* __restrict__ cw2,
                                                         float* __restrict__ out, float* __restrict__ ws, int F, int FCH)
{
    __shared__ float part[256];
    constexpr int RPB = 256 / KC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c = tid % KC, rg = tid / KC;
    const int f0 = blockIdx.x * FCH, f1 = min(f0 + FCH, F);
    const float* v = vraw + (size_t)b * F * KC;
    float* o = out + (size_t)b * F * KC;
    const float as = ws[(size_t)b * 2 * KC + c];
    float ss = 0.0f;
    for (int f = f0 + rg; f < f1; f += RPB) {
        float r = v[(size_t)f * KC + c] - as * cw2[(size_t)f * KC + c];
        o[(size_t)f * KC + c] = r;
        ss += r * r;
    }
    part[tid] = ss;
    __syncthreads();
    if (tid < KC) {
        float t = 0.0f;
        for (int r = 0; r < RPB; ++r) t += part[r * KC + tid];
        atomicAdd(&ws[(size_t)b * 2 * KC + KC + tid], t);
    }
}

template <int KC>
__global__ __launch_bounds__(256) void vlad_scale_kernel(float* __restrict__ out, const float* __restrict__ ws,
                                                         float* aux_asum, float* aux_inv_c, float* aux_inv_g, int F, int FCH)
{
    __shared__ float s_inv[KC];
    __shared__ float s_g;
    constexpr int RPB = 256 / KC;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c = tid % KC, rg = tid / KC;
    if (tid < KC) {
        const float inv = 1.0f / fmaxf(sqrtf(ws[(size_t)b * 2 * KC + KC + tid]), 1e-12f);   // F.normalize eps
        s_inv[tid] = inv;
        if (blockIdx.x == 0) {
            if (aux_inv_c) aux_inv_c[b * KC + tid] = inv;
            if (aux_asum) aux_asum[b * KC + tid] = ws[(size_t)b * 2 * KC + tid];
        }
    }
    __syncthreads();
    if (tid == 0) {
        float tot = 0.0f;
        for (int k = 0; k < KC; ++k) tot += ws[(size_t)b * 2 * KC + KC + k] * s_inv[k] * s_inv[k];
        s_g = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
        if (aux_inv_g && blockIdx.x == 0) aux_inv_g[b] = s_g;
    }
    __syncthreads();
    const float sc = s_inv[c] * s_g;
    const int f0 = blockIdx.x * FCH, f1 = min(f0 + FCH, F);
    float* o = out + (size_t)b * F * KC;
    for (int f = f0 + rg; f < f1; f += RPB) o[(size_t)f * KC + c] *= sc;
}

// per-cloud column max: in [B][N][C] -> out [B][C].  grid (ceil(C/64), B), 256 threads = 4 row groups x 64 cols
__global__ __launch_bounds__(256) void colmax_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int N, int C)
{
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const int b = blockIdx.y;
    float m = -INFINITY;
    if (c < C) {
        const float* p = in + (size_t)b * N * ldi + c;
        for (int n = rg; n < N; n += 4) m = fmaxf(m, p[(size_t)n * ldi]);
    }
    part[rg][threadIdx.x & 63] = m;
    __syncthreads();
    if (rg == 0 && c < C) out[(size_t)b * C + c] = fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]),
                                                         fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] * b[i];
}

// ---------------------------------------------------------------------------------------------
// Descriptor retrieval (evaluate.py:162-206 builds a KDTree per database and queries recall_num = 25
// neighbours per query): squared L2 distances from the score matrix S = Q D^T and the row norms, then the k smallest per
// query by k passes of a wave-wide lexicographic (distance, index) arg-min.  nq and ndb are a few hundred to a few
// thousand 256-d descriptors: one wave per query, rows stay in L2.
// ---------------------------------------------------------------------------------------------
__global__ void rownorm_kernel(const float* __restrict__ X, int ld, int n, int dim, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < dim; ++c) s = fmaf(X[(size_t)i * ld + c], X[(size_t)i * ld + c], s);
    out[i] = s;
}

__global__ __launch_bounds__(256) void retrieval_topk_kernel(const float* __restrict__ S, const float* __restrict__ qn,
                                                             const float* __restrict__ dn, int nq, int ndb, int k,
                                                             int32_t* __restrict__ idx, float* __restrict__ dist)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* s = S + (size_t)q * ndb;
    const float nq2 = qn[q];
    float last_v = -INFINITY;
    int last_j = -1;
    for (int r = 0; r < k; ++r) {
        float best_v = INFINITY;
        int best_j = 0x7fffffff;
        for (int j = lane; j < ndb; j += 64) {
            const float v = fmaxf(nq2 + dn[j] - 2.0f * s[j], 0.0f);
            const bool after_last = v > last_v || (v == last_v && j > last_j);          // not yet emitted
            const bool better = v < best_v || (v == best_v && j < best_j);
            if (after_last && better) { best_v = v; best_j = j; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best_v, m, 64);
            const int oj = __shfl_xor(best_j, m, 64);
            if (ov < best_v || (ov == best_v && oj < best_j)) { best_v = ov; best_j = oj; }
        }
        if (lane == 0) {
            idx[(size_t)q * k + r] = best_j == 0x7fffffff ? -1 : best_j;
            dist[(size_t)q * k + r] = best_v;
        }
        last_v = best_v;
        last_j = best_j;
    }
}

// Batched hard-negative selection (util/data.py:103-115 inside every __getitem__ of the second training phase: a KDTree over
// the 4000 sampled negatives of ONE query per call): query b has its OWN candidate list cand[b][0..nc) (row numbers of the
// latent-vector table); the hard_neg_num candidates nearest to its descriptor come out nearest first, as POSITIONS into
// cand[b].  One workgroup per query: squared distances (direct form sum (t - q)^2, one wave per candidate row) into LDS, then
// k rounds of a block-wide lexicographic (distance, position) arg-min.
__global__ __launch_bounds__(256) void hard_negatives_kernel(const float* __restrict__ table, long long ldt, const float* __restrict__ Q,
                                                             long long ldq, const int32_t* __restrict__ cand, int nc, int dim, int k,
                                                             int32_t* __restrict__ pos, float* __restrict__ dist)
{
    extern __shared__ float dsm[];          // [nc] distances, then [8] reduction scratch
    __shared__ float rv[4];
    __shared__ int rj[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* q = Q + (long long)b * ldq;
    const int32_t* cb = cand + (long long)b * nc;
    for (int c = wave; c < nc; c += 4) {
        const float* row = table + (long long)cb[c] * ldt;
        float s = 0.f;
        for (int d = lane; d < dim; d += 64) { const float t = row[d] - q[d]; s = fmaf(t, t, s); }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) dsm[c] = s;
    }
    __syncthreads();
    float last_v = -INFINITY;
    int last_j = -1;
    for (int r = 0; r < k; ++r) {
        float best_v = INFINITY;
        int best_j = 0x7fffffff;
        for (int j = tid; j < nc; j += 256) {
            const float v = dsm[j];
            const bool after_last = v > last_v || (v == last_v && j > last_j);
            const bool better = v < best_v || (v == best_v && j < best_j);
            if (after_last && better) { best_v = v; best_j = j; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best_v, m, 64);
            const int oj = __shfl_xor(best_j, m, 64);
            if (ov < best_v || (ov == best_v && oj < best_j)) { best_v = ov; best_j = oj; }
        }
        if (lane == 0) { rv[wave] = best_v; rj[wave] = best_j; }
        __syncthreads();
        best_v = rv[0]; best_j = rj[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (rv[w] < best_v || (rv[w] == best_v && rj[w] < best_j)) { best_v = rv[w]; best_j = rj[w]; }
        if (tid == 0) {
            pos[(long long)b * k + r] = best_j == 0x7fffffff ? -1 : best_j;
            dist[(long long)b * k + r] = best_v;
        }
        last_v = best_v;
        last_j = best_j;
        __syncthreads();
    }
}

// float64 -> float32 (round to nearest even, what numpy's astype / torch's .float() do): the Oxford submaps are stored as
// 4096 x 3 float64 (loading_pointclouds.py:26-35); the raw bytes go over PCIe and are narrowed here.
__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, long long n)
{
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        const double2 v = *reinterpret_cast<const double2*>(in + i);
        *reinterpret_cast<float2*>(out + i) = make_float2((float)v.x, (float)v.y);
    } else if (i < n) {
        out[i] = (float)in[i];
    }
}

}  // namespace

extern "C" int lpd_linear_smallk(const float* X, int ldx, const float* W, int w_sn, int w_sk, long long w_sb,
                                 int rows_per_w, float* Y, int ldy, int M, int N, int K, const float* bias,
                                 const float* scale, const float* shift, int act, float slope, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(X && W && Y, "lpd_linear_smallk: null pointer");
    LPD_CHECK_ARG(M > 0 && N > 0 && K > 0 && K <= 8, "lpd_linear_smallk: bad dims M=%d N=%d K=%d (K <= 8)", M, N, K);
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_linear_smallk: scale and shift must be given together");
    long long total = (long long)M * N;
    hipLaunchKernelGGL(linear_smallk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, X, ldx, W, w_sn,
                       w_sk, w_sb, rows_per_w, Y, ldy, M, N, K, bias, scale, shift, act, slope);
    LPD_CHECK_LAUNCH("lpd_linear_smallk");
    return LPD_OK;
}

extern "C" int lpd_transpose(const float* in, float* out, int batch, int R, int C, int ldi, int ldo, long long si,
                             long long so, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(in && out, "lpd_transpose: null pointer");
    LPD_CHECK_ARG(batch > 0 && batch <= 65535 && R > 0 && C > 0, "lpd_transpose: bad dims batch=%d R=%d C=%d", batch, R, C);
    dim3 grid((C + 31) / 32, (R + 31) / 32, batch);
    LPD_CHECK_ARG(grid.y <= 65535, "lpd_transpose: R=%d too large", R);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, out, R, C, ldi, ldo, si, so);
    LPD_CHECK_LAUNCH("lpd_transpose");
    return LPD_OK;
}

extern "C" int lpd_softmax_affine(const float* in, float* out, int rows, int ncols, const float* scale, const float* shift,
                                  int group_rows, float* colsum, int colsum_ld, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(in && out, "lpd_softmax_affine: null pointer");
    LPD_CHECK_ARG(rows > 0 && ncols > 0 && ncols <= 64, "lpd_softmax_affine: bad dims rows=%d ncols=%d (<= 64)", rows, ncols);
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_softmax_affine: scale and shift must be given together");
    if (colsum) {
        LPD_CHECK_ARG(group_rows > 0 && group_rows % SM_RPW == 0 && rows % group_rows == 0 && colsum_ld >= ncols,
                      "lpd_softmax_affine: column sums need group_rows %% %d == 0 and rows %% group_rows == 0", SM_RPW);
        const int waves = rows / SM_RPW;
        if (ncols == 64 && group_rows % (4 * SM_RPW) == 0 && ((((uintptr_t)in | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0))
            hipLaunchKernelGGL(softmax_affine_colsum64_kernel<1>, dim3((waves + 3) / 4), dim3(256), 0, stream, in, out, rows, scale, shift,
                               group_rows, colsum, colsum_ld, 0);
        else
            hipLaunchKernelGGL(softmax_affine_colsum_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, in, out, rows, ncols, scale, shift,
                               group_rows, colsum, colsum_ld);
    } else {
        hipLaunchKernelGGL(softmax_affine_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, in, out, rows, ncols, scale, shift);
    }
    LPD_CHECK_LAUNCH("lpd_softmax_affine");
    return LPD_OK;
}

// softmax(scale * (sum of `parts` planes) + shift) over 64 columns + per-group column sums: the consumer of lpd_gemm_p8_fused's
// partial assignment products.  rows % group_rows == 0, group_rows % 64 == 0.
extern "C" int lpd_softmax_affine_parts(const float* in, int parts, long long part_stride, float* out, int rows, const float* scale,
                                        const float* shift, int group_rows, float* colsum, int colsum_ld, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(in && out && colsum, "lpd_softmax_affine_parts: null pointer");
    LPD_CHECK_ARG((parts == 1 || parts == 2 || parts == 4 || parts == 8) && part_stride >= (long long)rows * 64 && part_stride % 4 == 0,
                  "lpd_softmax_affine_parts: parts in {1, 2, 4, 8}");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_softmax_affine_parts: scale and shift must be given together");
    LPD_CHECK_ARG(rows > 0 && group_rows > 0 && group_rows % (4 * SM_RPW) == 0 && rows % group_rows == 0 && colsum_ld >= 64,
                  "lpd_softmax_affine_parts: rows %% group_rows == 0, group_rows %% %d == 0", 4 * SM_RPW);
    LPD_CHECK_ARG((((uintptr_t)in | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0, "lpd_softmax_affine_parts: alignment");
    const int waves = rows / SM_RPW;
    const dim3 grid((waves + 3) / 4), block(256);
    switch (parts) {
        case 1: hipLaunchKernelGGL(softmax_affine_colsum64_kernel<1>, grid, block, 0, stream, in, out, rows, scale, shift, group_rows, colsum, colsum_ld, part_stride); break;
        case 2: hipLaunchKernelGGL(softmax_affine_colsum64_kernel<2>, grid, block, 0, stream, in, out, rows, scale, shift, group_rows, colsum, colsum_ld, part_stride); break;
        case 4: hipLaunchKernelGGL(softmax_affine_colsum64_kernel<4>, grid, block, 0, stream, in, out, rows, scale, shift, group_rows, colsum, colsum_ld, part_stride); break;
        default: hipLaunchKernelGGL(softmax_affine_colsum64_kernel<8>, grid, block, 0, stream, in, out, rows, scale, shift, group_rows, colsum, colsum_ld, part_stride); break;
    }
    LPD_CHECK_LAUNCH("lpd_softmax_affine_parts");
    return LPD_OK;
}

extern "C" int lpd_vlad_finalize(const float* vraw, const float* act, const float* cw2, float* out, float* ws,
                                 float* aux_asum, float* aux_inv_c, float* aux_inv_g, int B, int N, int F, int KC,
                                 int asum_ready, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(vraw && (act || asum_ready) && cw2 && out && ws, "lpd_vlad_finalize: null pointer");
    LPD_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && F > 0, "lpd_vlad_finalize: bad dims");
    LPD_CHECK_ARG(KC == 64, "lpd_vlad_finalize: cluster_size=%d unsupported (64)", KC);
    if (!asum_ready) {   // else ws[b][0..KC) already holds a_sum (lpd_softmax_affine with colsum) and ws[b][KC..2KC) is zero
        (void)hipMemsetAsync(ws, 0, sizeof(float) * (size_t)B * 2 * KC, stream);
        const int nchunks = N >= 1024 ? 16 : (N + 63) / 64;
        hipLaunchKernelGGL(vlad_asum_kernel<64>, dim3(nchunks, B), dim3(256), 0, stream, act, ws, N);
        LPD_CHECK_LAUNCH("lpd_vlad_finalize(asum)");
    }
    const int FCH = 64;
    const int fblocks = (F + FCH - 1) / FCH;
    hipLaunchKernelGGL(vlad_resid_kernel<64>, dim3(fblocks, B), dim3(256), 0, stream, vraw, cw2, out, ws, F, FCH);
    LPD_CHECK_LAUNCH("lpd_vlad_finalize(resid)");
    hipLaunchKernelGGL(vlad_scale_kernel<64>, dim3(fblocks, B), dim3(256), 0, stream, out, (const float*)ws, aux_asum,
                       aux_inv_c, aux_inv_g, F, FCH);
    LPD_CHECK_LAUNCH("lpd_vlad_finalize(scale)");
    return LPD_OK;
}

extern "C" int lpd_colmax(const float* in, int ldi, float* out, int B, int N, int C, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(in && out, "lpd_colmax: null pointer");
    LPD_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && C > 0, "lpd_colmax: bad dims");
    hipLaunchKernelGGL(colmax_kernel, dim3((C + 63) / 64, B), dim3(256), 0, stream, in, ldi, out, N, C);
    LPD_CHECK_LAUNCH("lpd_colmax");
    return LPD_OK;
}

// Context gating in one launch (util/PointNetVlad.py:103-115): out = h * sigmoid((h . Wg + bias) * scale + shift), h [B][D],
// Wg [D][D] k-major.  One workgroup per row, one output column per thread (a wave reads 256 contiguous bytes of Wg per
// k; the 256 KiB of Wg stay in L2), the k range in four quarters (four 256-thread groups, summed through LDS).  The MFMA GEMM spent 30 us on this 4-MFLOP product
// (two workgroups on the whole chip) and the product h * gates was a third launch.
__global__ __launch_bounds__(1024) void gating_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ Wg, int ldw,
                                                      const float* __restrict__ bias, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ out, int ldo, int D)
{
    extern __shared__ float hrow[];     // [D] then [4][256] partial sums
    float* part = hrow + D;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int c = tid & 255, kq = tid >> 8;                 // column inside the 256-wide tile, quarter of the k range
    for (int k = tid; k < D; k += 1024) hrow[k] = h[(long long)b * ldh + k];
    __syncthreads();
    const int kper = (D + 3) / 4;
    const int kb = kq * kper, ke = min(kb + kper, D);
    for (int n0 = 0; n0 < D; n0 += 256) {
        const int n = n0 + c;
        const float* w = Wg + min(n, D - 1);
        float acc = 0.0f;
        int k = kb;
        for (; k + 8 <= ke; k += 8) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = w[(long long)(k + j) * ldw];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(hrow[k + j], wv[j], acc);
        }
        for (; k < ke; ++k) acc = fmaf(hrow[k], w[(long long)k * ldw], acc);
        part[kq * 256 + c] = acc;
        __syncthreads();
        if (kq == 0 && n < D) {
            float v = (part[c] + part[256 + c]) + (part[512 + c] + part[768 + c]);
            if (bias) v += bias[n];
            if (scale) v = v * scale[n] + shift[n];
            out[(long long)b * ldo + n] = hrow[n] * lpd_sigmoid(v);
        }
        __syncthreads();
    }
}

extern "C" int lpd_gating(const float* h, int ldh, const float* Wg, int ldw, const float* bias, const float* scale, const float* shift,
                          float* out, int ldo, int B, int D, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(h && Wg && out && B > 0 && D > 0, "lpd_gating: bad arguments");
    LPD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "lpd_gating: scale and shift must be given together");
    LPD_CHECK_ARG(D <= 8192 && ldh >= D && ldw >= D && ldo >= D, "lpd_gating: D=%d (<= 8192), leading dims >= D", D);
    hipLaunchKernelGGL(gating_kernel, dim3(B), dim3(1024), ((size_t)D + 1024) * sizeof(float), stream, h, ldh, Wg, ldw, bias, scale, shift, out, ldo, D);
    LPD_CHECK_LAUNCH("lpd_gating");
    return LPD_OK;
}

extern "C" int lpd_mul(const float* a, const float* b, float* out, long long n, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(a && b && out && n > 0, "lpd_mul: bad arguments");
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, out, n);
    LPD_CHECK_LAUNCH("lpd_mul");
    return LPD_OK;
}

extern "C" int lpd_retrieval_topk(const float* S, const float* Q, int ldq, const float* D, int ldd, int nq, int ndb, int dim,
                                  int k, int32_t* idx, float* dist, float* ws, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    LPD_CHECK_ARG(S && Q && D && idx && dist && ws, "lpd_retrieval_topk: null pointer");
    LPD_CHECK_ARG(nq > 0 && ndb > 0 && dim > 0 && k > 0 && k <= ndb, "lpd_retrieval_topk: bad dims nq=%d ndb=%d dim=%d k=%d", nq, ndb, dim, k);
    float* qn = ws;            // [nq]
    float* dn = ws + nq;       // [ndb]
    hipLaunchKernelGGL(rownorm_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, Q, ldq, nq, dim, qn);
    hipLaunchKernelGGL(rownorm_kernel, dim3((ndb + 255) / 256), dim3(256), 0, stream, D, ldd, ndb, dim, dn);
    hipLaunchKernelGGL(retrieval_topk_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, S, (const float*)qn, (const float*)dn, nq, ndb, k, idx,
                       dist);
    LPD_CHECK_LAUNCH("lpd_retrieval_topk");
    return LPD_OK;
}

extern "C" int lpd_hard_negatives(const float* table, long long ldt, const float* Q, long long ldq, const int32_t* cand, int bq, int nc,
                                  int dim, int k, int32_t* pos, float* dist, void* stream_)
{
    LPD_CHECK_ARG(table && Q && cand && pos && dist, "lpd_hard_negatives: null pointer");
    LPD_CHECK_ARG(bq > 0 && nc > 0 && dim > 0 && k > 0 && k <= nc, "lpd_hard_negatives: bad dims bq=%d nc=%d dim=%d k=%d", bq, nc, dim, k);
    LPD_CHECK_ARG(nc <= 36864, "lpd_hard_negatives: nc=%d candidates exceed the LDS distance table (36864)", nc);
    const size_t lds = (size_t)nc * sizeof(float);
    (void)hipFuncSetAttribute((const void*)hard_negatives_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(hard_negatives_kernel, dim3(bq), dim3(256), lds, (hipStream_t)stream_, table, ldt, Q, ldq, cand, nc, dim, k, pos, dist);
    LPD_CHECK_LAUNCH("lpd_hard_negatives");
    return LPD_OK;
}

extern "C" int lpd_f64_to_f32(const double* in, float* out, long long n, void* stream)
{
    LPD_CHECK_ARG(in && out && n > 0, "lpd_f64_to_f32: bad arguments");
    LPD_CHECK_ARG((((uintptr_t)in & 15) | ((uintptr_t)out & 7)) == 0, "lpd_f64_to_f32: in must be 16-byte, out 8-byte aligned");
    const long long threads = (n + 1) / 2;
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    LPD_CHECK_LAUNCH("lpd_f64_to_f32");
    return LPD_OK;
}
