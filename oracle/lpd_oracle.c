/*
 * oracle/lpd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the bit-critical pieces of the LPD-Net hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library (through oracle/lpd_oracle.py); the product path (lpd-net-pytorch_amd/) never does.
 *
 * Parity status: PINNED against outputs of the reference itself, generated in the build
 * container by tests/golden/make_golden.py (the reference has no tests or golden vectors
 * of its own) and committed under tests/golden/.
 *
 * Reference behaviour restated here (file:line relative to /root/reference):
 *   util/lpdnet_model.py:317-326  knn(x, k):
 *       inner = -2 * (x^T x)              (:318)  torch.matmul on CPU == sequential fp32
 *                                                 FMA chain over channels c = 0..C-1 from 0
 *       xx    = sum_c x^2                 (:320)  non-fused squares; torch's outer-dim sum ==
 *                                                 sequential inside 16-channel blocks, block
 *                                                 partials then added sequentially
 *       pd    = (-xx_j - inner_ij) - xx_i (:322,:324)
 *       idx   = topk(pd, k) largest, sorted descending (:325)
 *     Tie rule of this restatement (torch's CPU topk leaves ties unspecified): larger pd
 *     first, equal pd -> lower index first.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the arithmetic below distinguishes fused from
 * non-fused operations.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#include "lpd_oracle_knn.h"

/*
 * x: [B][N][C] point-major fp32.  idx: [B][N][k] int32 (descending pd).
 * pd_out (optional, may be NULL): [B][N][k] the selected pd values.
 */
void lpd_oracle_knn(const float *x, int B, int N, int C, int k, int32_t *idx, float *pd_out)
{
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * C;
        float *xx = (float *)malloc(sizeof(float) * (size_t)N);
        for (int i = 0; i < N; ++i) xx[i] = oracle_sumsq(xb + (size_t)i * C, C);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            float *bv = (float *)malloc(sizeof(float) * (size_t)k);
            int *bi = (int *)malloc(sizeof(int) * (size_t)k);
            int cnt = 0;
            const float *xi = xb + (size_t)i * C;
            for (int j = 0; j < N; ++j) {
                float pd = oracle_pd(xi, xb + (size_t)j * C, xx[i], xx[j], C);
                if (cnt == k && !oracle_before(pd, j, bv[k - 1], bi[k - 1])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && oracle_before(pd, j, bv[pos - 1], bi[pos - 1])) {
                    bv[pos] = bv[pos - 1];
                    bi[pos] = bi[pos - 1];
                    --pos;
                }
                bv[pos] = pd;
                bi[pos] = j;
                if (cnt < k) ++cnt;
            }
            for (int t = 0; t < k; ++t) {
                idx[((size_t)b * N + i) * k + t] = t < cnt ? bi[t] : -1;
                if (pd_out) pd_out[((size_t)b * N + i) * k + t] = t < cnt ? bv[t] : 0.0f;
            }
            free(bv);
            free(bi);
        }
        free(xx);
    }
}

/*
 * Full pd matrix of one cloud (for tie analysis in fixtures/tests): x [N][C] -> pd [N][N].
 */
void lpd_oracle_pd_matrix(const float *x, int N, int C, float *pd)
{
    float *xx = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < N; ++i) xx[i] = oracle_sumsq(x + (size_t)i * C, C);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            pd[(size_t)i * N + j] = oracle_pd(x + (size_t)i * C, x + (size_t)j * C, xx[i], xx[j], C);
    free(xx);
}

/*
 * Rows whose top-k result depends on a tie rule: the k-th and (k+1)-th best pd are equal,
 * or two of the selected pd are equal (order among them is then unspecified in torch).
 * tie_mask: [B][N] uint8.
 */
void lpd_oracle_knn_tie_rows(const float *x, int B, int N, int C, int k, uint8_t *tie_mask)
{
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * C;
        float *xx = (float *)malloc(sizeof(float) * (size_t)N);
        for (int i = 0; i < N; ++i) xx[i] = oracle_sumsq(xb + (size_t)i * C, C);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
            int kk = k + 1 < N ? k + 1 : N;
            float *bv = (float *)malloc(sizeof(float) * (size_t)kk);
            int cnt = 0;
            const float *xi = xb + (size_t)i * C;
            for (int j = 0; j < N; ++j) {
                float pd = oracle_pd(xi, xb + (size_t)j * C, xx[i], xx[j], C);
                if (cnt == kk && !(pd > bv[kk - 1])) continue;
                int pos = cnt < kk ? cnt : kk - 1;
                while (pos > 0 && pd > bv[pos - 1]) { bv[pos] = bv[pos - 1]; --pos; }
                bv[pos] = pd;
                if (cnt < kk) ++cnt;
            }
            uint8_t tie = 0;
            for (int t = 1; t < cnt; ++t) if (bv[t] == bv[t - 1]) tie = 1;
            tie_mask[(size_t)b * N + i] = tie;
            free(bv);
        }
        free(xx);
    }
}

int lpd_oracle_num_threads(void)
{
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
