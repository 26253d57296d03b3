/*
 * oracle/lpd_forward.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (gcc + OpenMP) restatement of the WHOLE eval-mode hot path of the reference, in the reference's own formulation
 * (edge tensors gathered and convolved per edge, no algebraic split, no fusion):
 *   util/PointNetVlad.py:261-270  PointNetVlad.forward (featnet = 'lpdnet', no T-Nets)
 *   util/lpdnet_model.py:211-268  LPDNet.forward: conv1/bn1/act, conv2/bn2/act (:231-232), get_graph_feature on the features
 *                                 (:246, 331-363) -> convDG1 -> max (:249-250), convDG2 on the un-maxed tensor -> max
 *                                 (:251-252), knn on the raw xyz (:255), get_graph_feature(x2, idx) -> convSN1 -> max
 *                                 (:256-258), cat -> conv3/bn3/act (:260-262)
 *   util/lpdnet_model.py:317-326  knn: the bit-exact arithmetic of lpd_oracle_knn.h
 *   util/PointNetVlad.py:45-83    NetVLADLoupe.forward;  :103-115 GatingContext.forward
 * Eval-mode BatchNorm: (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias.  LeakyReLU slope 0.01.
 *
 * Two uses: (1) tests/test_oracle_golden.py checks it against the reference's golden descriptors (parity pinned like the
 * Python oracle's); (2) bench.py's cpu_baseline leg times it on the GPU box's host cores: one cloud per OpenMP thread,
 * every cloud processed serially, so the rate scales with the cores actually used ("kind": "port").
 *
 * Weights: an array of float pointers in the order listed at LPD_W_* below (state_dict tensors, row-major as torch stores them).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#include "lpd_oracle_knn.h"

enum {
    LPD_W_CONV1 = 0, LPD_W_BN1 = 1,          /* conv1_lpd.weight [64][3];   bn1_lpd: weight, bias, running_mean, running_var */
    LPD_W_CONV2 = 5, LPD_W_BN2 = 6,          /* conv2_lpd.weight [64][64] */
    LPD_W_DG1 = 10, LPD_W_BNDG1 = 11,        /* convDG1.0.weight [128][128] (input = cat(neighbour, centre)) */
    LPD_W_DG2 = 15, LPD_W_BNDG2 = 16,        /* convDG2.0.weight [128][128] */
    LPD_W_SN1 = 20, LPD_W_BNSN1 = 21,        /* convSN1.0.weight [256][256] */
    LPD_W_CONV3 = 25, LPD_W_BN3 = 26,        /* conv3_lpd.weight [E][512] */
    LPD_W_CW = 30, LPD_W_VBN1 = 31,          /* net_vlad.cluster_weights [E][64]; net_vlad.bn1 */
    LPD_W_CW2 = 35, LPD_W_HID = 36,          /* cluster_weights2 [E][64]; hidden1_weights [E*64][256] */
    LPD_W_VBN2 = 37,                         /* net_vlad.bn2 */
    LPD_W_GATE = 41, LPD_W_GBN = 42,         /* context_gating.gating_weights [256][256]; context_gating.bn1 */
    LPD_W_COUNT = 46
};

#define EPS_BN 1e-5f
#define SLOPE 0.01f

static inline float leaky(float v) { return v > 0.0f ? v : v * SLOPE; }

/* y[o] = sum_i W[o][i] * x[i]  (torch conv / linear weight layout [out][in]) */
static void matvec(const float *restrict W, const float *restrict x, float *restrict y, int O, int I)
{
    for (int o = 0; o < O; ++o) {
        const float *w = W + (size_t)o * I;
        float acc = 0.0f;
#pragma omp simd reduction(+ : acc)
        for (int i = 0; i < I; ++i) acc += w[i] * x[i];
        y[o] = acc;
    }
}

/* Y[p][o] = sum_i W[o][i] * X[p][i] for P rows of X (stride ldx) at once: a weight row is loaded once for a group of 8 rows
 * and the 8 dot products are independent accumulation chains -- a plain matvec per point re-reads W (2 MiB for conv3) for
 * every point and runs one latency-bound chain (first version: 28 s per cloud with all cores busy, bound by L3 traffic). */
static void matmat(const float *restrict W, const float *restrict X, int ldx, float *restrict Y, int ldy, int O, int I, int P)
{
    for (int p0 = 0; p0 < P; p0 += 8) {
        const int np = P - p0 < 8 ? P - p0 : 8;
        const float *x[8];
        for (int r = 0; r < 8; ++r) x[r] = X + (size_t)(p0 + (r < np ? r : 0)) * ldx;
        for (int o = 0; o < O; ++o) {
            const float *w = W + (size_t)o * I;
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
#pragma omp simd reduction(+ : a0, a1, a2, a3, a4, a5, a6, a7)
            for (int i = 0; i < I; ++i) {
                const float wi = w[i];
                a0 += wi * x[0][i]; a1 += wi * x[1][i]; a2 += wi * x[2][i]; a3 += wi * x[3][i];
                a4 += wi * x[4][i]; a5 += wi * x[5][i]; a6 += wi * x[6][i]; a7 += wi * x[7][i];
            }
            const float acc[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
            for (int r = 0; r < np; ++r) Y[(size_t)(p0 + r) * ldy + o] = acc[r];
        }
    }
}

/* eval-mode BatchNorm + optional LeakyReLU, in place, C channels; bn = {weight, bias, mean, var} */
static void bn_act(float *restrict y, const float *const *bn, int C, int act)
{
    for (int c = 0; c < C; ++c) {
        float v = (y[c] - bn[2][c]) / sqrtf(bn[3][c] + EPS_BN) * bn[0][c] + bn[1][c];
        y[c] = act ? leaky(v) : v;
    }
}

/* kNN of one cloud, the reference's arithmetic (lpd_oracle_knn.h: a c-ordered fmaf chain per pair), organised so that the
 * chains of 64 candidates j run side by side (vectorised over j on a channel-major copy; every chain keeps its order). */
static void knn_cloud(const float *x, int N, int C, int k, int32_t *idx)
{
    enum { JB = 64 };
    const int Np = (N + JB - 1) / JB * JB;
    float *xx = (float *)malloc(sizeof(float) * (size_t)Np);
    float *xt = (float *)calloc((size_t)C * Np, sizeof(float));           /* [C][Np] channel-major, zero padded */
    float *bv = (float *)malloc(sizeof(float) * (size_t)k);
    int *bi = (int *)malloc(sizeof(int) * (size_t)k);
    for (int i = 0; i < N; ++i) {
        xx[i] = oracle_sumsq(x + (size_t)i * C, C);
        for (int c = 0; c < C; ++c) xt[(size_t)c * Np + i] = x[(size_t)i * C + c];
    }
    for (int i = 0; i < N; ++i) {
        int cnt = 0;
        const float *xi = x + (size_t)i * C;
        const float xxi = xx[i];
        for (int j0 = 0; j0 < N; j0 += JB) {
            float dot[JB], pdv[JB];
            for (int jj = 0; jj < JB; ++jj) dot[jj] = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float xic = xi[c];
                const float *row = xt + (size_t)c * Np + j0;
#pragma omp simd
                for (int jj = 0; jj < JB; ++jj) dot[jj] = fmaf(xic, row[jj], dot[jj]);
            }
            const int jn = N - j0 < JB ? N - j0 : JB;
            for (int jj = 0; jj < jn; ++jj) {
                const float inner = -2.0f * dot[jj];
                const float t = (-xx[j0 + jj]) - inner;
                pdv[jj] = t - xxi;
            }
            for (int jj = 0; jj < jn; ++jj) {
                const float pd = pdv[jj];
                const int j = j0 + jj;
                if (cnt == k && !oracle_before(pd, j, bv[k - 1], bi[k - 1])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && oracle_before(pd, j, bv[pos - 1], bi[pos - 1])) { bv[pos] = bv[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bv[pos] = pd;
                bi[pos] = j;
                if (cnt < k) ++cnt;
            }
        }
        for (int t = 0; t < k; ++t) idx[(size_t)i * k + t] = bi[t];
    }
    free(xx); free(xt); free(bv); free(bi);
}

/* one cloud: x [N][3] -> desc [256]; returns 0, or -1 when out of memory */
static int forward_cloud(const float *x, int N, int k, int E, const float *const *w, float *desc)
{
    const int K = 64, O = 256;
    float *f1 = (float *)malloc(sizeof(float) * (size_t)N * 64);
    float *f0 = (float *)malloc(sizeof(float) * (size_t)N * 64);
    float *cat = (float *)malloc(sizeof(float) * (size_t)N * 512);        /* [x1 (128) | x2 (128) | x3 (256)] per point */
    float *feat = (float *)malloc(sizeof(float) * (size_t)N * E);
    float *a = (float *)malloc(sizeof(float) * (size_t)N * K);
    float *vlad = (float *)calloc((size_t)E * K, sizeof(float));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * k);
    if (!f1 || !f0 || !cat || !feat || !a || !vlad || !idx) return -1;
    /* per-point layers (:231-232) */
    for (int n = 0; n < N; ++n) {
        matvec(w[LPD_W_CONV1], x + (size_t)n * 3, f1 + (size_t)n * 64, 64, 3);
        bn_act(f1 + (size_t)n * 64, w + LPD_W_BN1, 64, 1);
        matvec(w[LPD_W_CONV2], f1 + (size_t)n * 64, f0 + (size_t)n * 64, 64, 64);
        bn_act(f0 + (size_t)n * 64, w + LPD_W_BN2, 64, 1);
    }
    /* dynamic graph in feature space: edge (neighbour, centre) -> convDG1 -> max; convDG2 on every edge -> max (:246-252) */
    knn_cloud(f0, N, 64, k, idx);
    float *eb = (float *)malloc(sizeof(float) * (size_t)k * 256 * 3);     /* the k edges of one point: input, layer 1, layer 2 */
    if (!eb) return -1;
    for (int n = 0; n < N; ++n) {
        float *e = eb, *y1 = eb + (size_t)k * 256, *z = eb + (size_t)k * 512;
        float *x1 = cat + (size_t)n * 512, *x2 = x1 + 128;
        for (int c = 0; c < 128; ++c) { x1[c] = -INFINITY; x2[c] = -INFINITY; }
        for (int t = 0; t < k; ++t) {                                     /* the [k][128] edge tensor of this point (:350-357) */
            memcpy(e + (size_t)t * 128, f0 + (size_t)idx[(size_t)n * k + t] * 64, sizeof(float) * 64);
            memcpy(e + (size_t)t * 128 + 64, f0 + (size_t)n * 64, sizeof(float) * 64);
        }
        matmat(w[LPD_W_DG1], e, 128, y1, 128, 128, 128, k);
        for (int t = 0; t < k; ++t) bn_act(y1 + (size_t)t * 128, w + LPD_W_BNDG1, 128, 1);
        matmat(w[LPD_W_DG2], y1, 128, z, 128, 128, 128, k);
        for (int t = 0; t < k; ++t) {
            bn_act(z + (size_t)t * 128, w + LPD_W_BNDG2, 128, 1);
            for (int c = 0; c < 128; ++c) { x1[c] = fmaxf(x1[c], y1[(size_t)t * 128 + c]); x2[c] = fmaxf(x2[c], z[(size_t)t * 128 + c]); }
        }
    }
    /* static graph in Cartesian space (raw xyz): edge (x2 neighbour, x2 centre) -> convSN1 -> max (:255-258) */
    knn_cloud(x, N, 3, k, idx);
    for (int n = 0; n < N; ++n) {
        float *e = eb, *y = eb + (size_t)k * 256;
        float *x3 = cat + (size_t)n * 512 + 256;
        for (int c = 0; c < 256; ++c) x3[c] = -INFINITY;
        for (int t = 0; t < k; ++t) {
            memcpy(e + (size_t)t * 256, cat + (size_t)idx[(size_t)n * k + t] * 512 + 128, sizeof(float) * 128);
            memcpy(e + (size_t)t * 256 + 128, cat + (size_t)n * 512 + 128, sizeof(float) * 128);
        }
        matmat(w[LPD_W_SN1], e, 256, y, 256, 256, 256, k);
        for (int t = 0; t < k; ++t) {
            bn_act(y + (size_t)t * 256, w + LPD_W_BNSN1, 256, 1);
            for (int c = 0; c < 256; ++c) x3[c] = fmaxf(x3[c], y[(size_t)t * 256 + c]);
        }
    }
    free(eb);
    /* conv3 / bn3 / act (:260-262) and the NetVLAD soft assignment (PointNetVlad.py:48-58): a[n] = softmax(bn1(feat[n] . Wc)) */
    float *wct = (float *)malloc(sizeof(float) * (size_t)K * E);       /* cluster_weights transposed: [K][E] */
    if (!wct) return -1;
    for (int e2 = 0; e2 < E; ++e2)
        for (int c = 0; c < K; ++c) wct[(size_t)c * E + e2] = w[LPD_W_CW][(size_t)e2 * K + c];
    for (int n0 = 0; n0 < N; n0 += 32) {      /* 32 points per block: the 2 MiB conv3 weight is streamed once per block */
        const int np = N - n0 < 32 ? N - n0 : 32;
        matmat(w[LPD_W_CONV3], cat + (size_t)n0 * 512, 512, feat + (size_t)n0 * E, E, E, 512, np);
        for (int r = 0; r < np; ++r) bn_act(feat + (size_t)(n0 + r) * E, w + LPD_W_BN3, E, 1);
        matmat(wct, feat + (size_t)n0 * E, E, a + (size_t)n0 * K, K, K, E, np);
    }
    for (int n = 0; n < N; ++n) {
        float *an = a + (size_t)n * K;
        bn_act(an, w + LPD_W_VBN1, K, 0);
        float mx = an[0], s = 0.0f;
        for (int c = 1; c < K; ++c) mx = fmaxf(mx, an[c]);
        for (int c = 0; c < K; ++c) { an[c] = expf(an[c] - mx); s += an[c]; }
        for (int c = 0; c < K; ++c) an[c] /= s;
    }
    /* residual pooling (:61-68): vlad[e][c] = sum_n a[n][c] feat[n][e] - a_sum[c] * cw2[e][c] */
    float asum[64];
    memset(asum, 0, sizeof(asum));
    for (int n = 0; n < N; ++n) {
        const float *fn = feat + (size_t)n * E, *an = a + (size_t)n * K;
        for (int c = 0; c < K; ++c) asum[c] += an[c];
        for (int e2 = 0; e2 < E; ++e2) {
            const float f = fn[e2];
            float *v = vlad + (size_t)e2 * K;
#pragma omp simd
            for (int c = 0; c < K; ++c) v[c] += an[c] * f;
        }
    }
    for (int e2 = 0; e2 < E; ++e2)
        for (int c = 0; c < K; ++c) vlad[(size_t)e2 * K + c] -= asum[c] * w[LPD_W_CW2][(size_t)e2 * K + c];
    /* intra-normalisation over E per cluster (:70), flatten e*K + c (:73), L2 normalise (:74) */
    for (int c = 0; c < K; ++c) {
        float s = 0.0f;
        for (int e2 = 0; e2 < E; ++e2) s += vlad[(size_t)e2 * K + c] * vlad[(size_t)e2 * K + c];
        const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        for (int e2 = 0; e2 < E; ++e2) vlad[(size_t)e2 * K + c] *= inv;
    }
    {
        float s = 0.0f;
        for (size_t i = 0; i < (size_t)E * K; ++i) s += vlad[i] * vlad[i];
        const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        for (size_t i = 0; i < (size_t)E * K; ++i) vlad[i] *= inv;
    }
    /* hidden projection (:76), bn2 (:78), context gating (:103-115) */
    float h[256], g[256];
    memset(h, 0, sizeof(h));
    for (size_t i = 0; i < (size_t)E * K; ++i) {
        const float v = vlad[i];
        const float *wr = w[LPD_W_HID] + i * O;
#pragma omp simd
        for (int o = 0; o < O; ++o) h[o] += v * wr[o];
    }
    bn_act(h, w + LPD_W_VBN2, O, 0);
    memset(g, 0, sizeof(g));
    for (int i = 0; i < O; ++i) {
        const float v = h[i];
        const float *wr = w[LPD_W_GATE] + (size_t)i * O;
        for (int o = 0; o < O; ++o) g[o] += v * wr[o];
    }
    bn_act(g, w + LPD_W_GBN, O, 0);
    for (int o = 0; o < O; ++o) desc[o] = h[o] * (1.0f / (1.0f + expf(-g[o])));
    free(f1); free(f0); free(cat); free(feat); free(a); free(vlad); free(idx); free(wct);
    return 0;
}

/*
 * x [B][N][3] -> desc [B][256].  One cloud per OpenMP thread (threads <= 0: the OpenMP default).  Returns the number of
 * threads used, or -1 on an allocation failure.
 */
int lpd_oracle_forward_lpdnet(const float *x, int B, int N, int k, int E, const float *const *w, float *desc, int threads)
{
    int failed = 0, used = 1;
#if defined(_OPENMP)
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
    if (used > B) used = B;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(used)
    for (int b = 0; b < B; ++b)
        if (forward_cloud(x + (size_t)b * N * 3, N, k, E, w, desc + (size_t)b * 256) != 0) failed = 1;
    return failed ? -1 : used;
}
