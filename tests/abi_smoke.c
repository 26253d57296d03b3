/*
 * tests/abi_smoke.c -- a plain C caller of liblpd_hip.so compiled against include/lpd_hip.h.
 *
 * Guards the C-ABI boundary: the prototypes in the public header are what this program is compiled
 * with, so a header that drifts from the definitions corrupts these calls (and fails the checks
 * below) instead of going unnoticed behind the ctypes binding.  Built by __graft_entry__.build()
 * (gcc, links liblpd_hip.so + libamdhip64), run on the GPU box by tests/test_abi_gpu.py.
 *
 *   abi_smoke            -> runs lpd_version, lpd_knn_workspace_floats, lpd_gemm (plain and
 *                           cloud-panel A/C), lpd_gemm_bf16x3, lpd_knn, lpd_lpdnet_front +
 *                           lpd_knn_pm(PREPARED) on the device, checks the
 *                           results against host loops, prints "abi_smoke OK", exit 0
 *   abi_smoke --symbols  -> no GPU call: only lpd_version / workspace sizes (CPU-side check)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lpd_hip.h"

#define CHECK_HIP(x)                                                                  \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

#define CHECK_LPD(x)                                                                  \
    do {                                                                              \
        int rc_ = (x);                                                                \
        if (rc_ != LPD_OK) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, lpd_last_error());             \
            return 3;                                                                 \
        }                                                                             \
    } while (0)

static float frand(unsigned* s)
{
    *s = *s * 1664525u + 1013904223u;
    return (float)((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

static double max_rel(const float* got, const double* ref, int n)
{
    double e = 0, m = 0;
    for (int i = 0; i < n; ++i) {
        double d = fabs((double)got[i] - ref[i]);
        if (d > e) e = d;
        if (fabs(ref[i]) > m) m = fabs(ref[i]);
    }
    return e / (m > 0 ? m : 1);
}

int main(int argc, char** argv)
{
    int ver = lpd_version();
    long long wsf = lpd_knn_workspace_floats(2, 3, 256, 20);
    if (ver < 100 || wsf <= 0) {
        fprintf(stderr, "lpd_version %d / workspace %lld\n", ver, wsf);
        return 1;
    }
    if (argc > 1 && strcmp(argv[1], "--symbols") == 0) {
        printf("abi_smoke symbols OK (version %d, knn workspace %lld floats)\n", ver, wsf);
        return 0;
    }

    /* ---- lpd_gemm: C = relu((A.B^T + bias) * scale + shift), A [M][K] rows, B [N][K] (torch weight layout) ---- */
    enum { M = 256, N = 128, K = 64 };
    unsigned seed = 12345u;
    float *hA = malloc(sizeof(float) * M * K), *hB = malloc(sizeof(float) * N * K), *hC = malloc(sizeof(float) * M * N);
    float hbias[N], hscale[N], hshift[N];
    double* ref = malloc(sizeof(double) * M * N);
    for (int i = 0; i < M * K; ++i) hA[i] = frand(&seed);
    for (int i = 0; i < N * K; ++i) hB[i] = frand(&seed);
    for (int i = 0; i < N; ++i) { hbias[i] = frand(&seed); hscale[i] = 0.5f + 0.5f * frand(&seed); hshift[i] = frand(&seed); }
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[m * K + k] * hB[n * K + k];
            s = (s + hbias[n]) * hscale[n] + hshift[n];
            ref[m * N + n] = s > 0 ? s : 0;
        }
    float *dA, *dB, *dC, *dbias, *dscale, *dshift;
    CHECK_HIP(hipMalloc((void**)&dA, sizeof(float) * M * K));
    CHECK_HIP(hipMalloc((void**)&dB, sizeof(float) * N * K));
    CHECK_HIP(hipMalloc((void**)&dC, sizeof(float) * M * N));
    CHECK_HIP(hipMalloc((void**)&dbias, sizeof(hbias)));
    CHECK_HIP(hipMalloc((void**)&dscale, sizeof(hscale)));
    CHECK_HIP(hipMalloc((void**)&dshift, sizeof(hshift)));
    CHECK_HIP(hipMemcpy(dA, hA, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dB, hB, sizeof(float) * N * K, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dbias, hbias, sizeof(hbias), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dscale, hscale, sizeof(hscale), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dshift, hshift, sizeof(hshift), hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    CHECK_LPD(lpd_gemm(dA, dB, dC, M, N, K, K, K, N, 0, 0, 1, 0, 0, 0, 1, NULL, dbias, dscale, dshift, LPD_ACT_RELU, 0.0f, 0,
                       0, 0, 0, 0, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(hC, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    double e1 = max_rel(hC, ref, M * N);
    printf("lpd_gemm row-major        rel err %.2e\n", e1);
    if (!(e1 < 1e-5)) return 4;

    CHECK_HIP(hipMemset(dC, 0, sizeof(float) * M * N));
    CHECK_LPD(lpd_gemm_bf16x3(dA, dB, dC, M, N, K, K, K, N, 0, 0, 1, 0, 0, 0, 1, NULL, dbias, dscale, dshift, LPD_ACT_RELU, 0.0f,
                              0, 0, 0, 0, 0, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(hC, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    double e2 = max_rel(hC, ref, M * N);
    printf("lpd_gemm_bf16x3           rel err %.2e\n", e2);
    if (!(e2 < 5e-5)) return 5;

    /* ---- the same product with cloud-panel A and C: 2 clouds of 128 points, panels padded to 136 rows: these are the four
     *      trailing arguments (a_cloud, c_cloud, panel_n, panel_ld) whose prototype had drifted in round 1 ---- */
    enum { NC = 2, PN = 128, PLD = 136 };
    long long a_cloud = (long long)(K / 8) * PLD * 8, c_cloud = (long long)(N / 8) * PLD * 8;
    float *hAp = calloc(NC * a_cloud, sizeof(float)), *hCp = malloc(sizeof(float) * NC * c_cloud);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k)
            hAp[(m / PN) * a_cloud + ((long long)(k / 8) * PLD + m % PN) * 8 + k % 8] = hA[m * K + k];
    float *dAp, *dCp;
    CHECK_HIP(hipMalloc((void**)&dAp, sizeof(float) * NC * a_cloud));
    CHECK_HIP(hipMalloc((void**)&dCp, sizeof(float) * NC * c_cloud));
    CHECK_HIP(hipMemcpy(dAp, hAp, sizeof(float) * NC * a_cloud, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(dCp, 0, sizeof(float) * NC * c_cloud));
    CHECK_LPD(lpd_gemm(dAp, dB, dCp, M, N, K, 8, K, 8, 0, 0, 1, 0, 0, 0, 1, NULL, dbias, dscale, dshift, LPD_ACT_RELU, 0.0f, 0,
                       a_cloud, c_cloud, PN, PLD, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(hCp, dCp, sizeof(float) * NC * c_cloud, hipMemcpyDeviceToHost));
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) hC[m * N + n] = hCp[(m / PN) * c_cloud + ((long long)(n / 8) * PLD + m % PN) * 8 + n % 8];
    double e3 = max_rel(hC, ref, M * N);
    printf("lpd_gemm cloud-panel A/C  rel err %.2e\n", e3);
    if (!(e3 < 1e-5)) return 6;

    /* ---- lpd_knn: 1 cloud, 3 channels, 256 points, k = 8; brute-force host check of the index SETS on rows whose k-th and
     *      (k+1)-th distances are not within rounding of each other ---- */
    enum { KN = 256, KK = 8 };
    float* hx = malloc(sizeof(float) * 3 * KN);
    for (int i = 0; i < 3 * KN; ++i) hx[i] = frand(&seed);
    float *dx, *dws;
    int32_t *didx, *hidx = malloc(sizeof(int32_t) * KN * KK);
    long long wsn = lpd_knn_workspace_floats(1, 3, KN, KK);
    CHECK_HIP(hipMalloc((void**)&dx, sizeof(float) * 3 * KN));
    CHECK_HIP(hipMalloc((void**)&dws, sizeof(float) * wsn));
    CHECK_HIP(hipMalloc((void**)&didx, sizeof(int32_t) * KN * KK));
    CHECK_HIP(hipMemcpy(dx, hx, sizeof(float) * 3 * KN, hipMemcpyHostToDevice));
    CHECK_LPD(lpd_knn(dx, 1, 3, KN, KK, didx, dws, 0, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(hidx, didx, sizeof(int32_t) * KN * KK, hipMemcpyDeviceToHost));
    int bad = 0, checked = 0;
    for (int i = 0; i < KN; ++i) {
        double d[KN];
        for (int j = 0; j < KN; ++j) {
            double s = 0;
            for (int c = 0; c < 3; ++c) { double t = (double)hx[c * KN + i] - hx[c * KN + j]; s += t * t; }
            d[j] = s;
        }
        /* k-th smallest and (k+1)-th smallest by selection */
        double sorted[KN];
        memcpy(sorted, d, sizeof(d));
        for (int a = 0; a <= KK; ++a)
            for (int b = a + 1; b < KN; ++b)
                if (sorted[b] < sorted[a]) { double t = sorted[a]; sorted[a] = sorted[b]; sorted[b] = t; }
        if (sorted[KK] - sorted[KK - 1] < 1e-5) continue; /* near-tie at the boundary: order is an fp32 matter */
        ++checked;
        for (int t = 0; t < KK; ++t) {
            int j = hidx[i * KK + t];
            if (j < 0 || j >= KN || d[j] > sorted[KK - 1] + 1e-9) { ++bad; break; }
        }
    }
    printf("lpd_knn C=3 N=%d k=%d: %d rows checked, %d wrong\n", KN, KK, checked, bad);
    if (bad || checked < KN / 2) return 7;

    /* ---- lpd_lpdnet_front + lpd_knn_pm(LPD_KNN_PM_PREPARED): conv1 -> conv2 on 256 points against host loops, and the graph
     *      built from the operands the front kernel left in the workspace against lpd_knn_pm on the same features ---- */
    {
        enum { FN = 256, FK = 8 };
        float *hxyz = malloc(sizeof(float) * FN * 3), *hW1 = malloc(sizeof(float) * 64 * 3), *hW2 = malloc(sizeof(float) * 64 * 64);
        float hs[4][64], *hF = malloc(sizeof(float) * FN * 64);
        double* rF = malloc(sizeof(double) * FN * 64);
        for (int i = 0; i < FN * 3; ++i) hxyz[i] = frand(&seed);
        for (int i = 0; i < 64 * 3; ++i) hW1[i] = frand(&seed);
        for (int i = 0; i < 64 * 64; ++i) hW2[i] = 0.125f * frand(&seed);
        for (int v = 0; v < 4; ++v)
            for (int i = 0; i < 64; ++i) hs[v][i] = (v & 1) ? 0.1f * frand(&seed) : 1.0f + 0.5f * frand(&seed);
        for (int m = 0; m < FN; ++m) {
            double f1[64];
            for (int c = 0; c < 64; ++c) {
                double v = 0;
                for (int k = 0; k < 3; ++k) v += (double)hxyz[m * 3 + k] * hW1[c * 3 + k];
                v = v * hs[0][c] + hs[1][c];
                f1[c] = v > 0 ? v : 0.2 * v;
            }
            for (int n = 0; n < 64; ++n) {
                double v = 0;
                for (int c = 0; c < 64; ++c) v += f1[c] * hW2[n * 64 + c];
                v = v * hs[2][n] + hs[3][n];
                rF[m * 64 + n] = v > 0 ? v : 0.2 * v;
            }
        }
        float *dxyz, *dW1, *dW2, *dsb, *dF, *dkws, *dkws2;
        int32_t *di1, *di2, *hi1 = malloc(sizeof(int32_t) * FN * FK), *hi2 = malloc(sizeof(int32_t) * FN * FK);
        long long kws = lpd_knn_workspace_floats(1, 64, FN, FK);
        CHECK_HIP(hipMalloc((void**)&dxyz, sizeof(float) * FN * 3));
        CHECK_HIP(hipMalloc((void**)&dW1, sizeof(float) * 64 * 3));
        CHECK_HIP(hipMalloc((void**)&dW2, sizeof(float) * 64 * 64));
        CHECK_HIP(hipMalloc((void**)&dsb, sizeof(float) * 4 * 64));
        CHECK_HIP(hipMalloc((void**)&dF, sizeof(float) * FN * 64));
        CHECK_HIP(hipMalloc((void**)&dkws, sizeof(float) * kws));
        CHECK_HIP(hipMalloc((void**)&dkws2, sizeof(float) * kws));
        CHECK_HIP(hipMalloc((void**)&di1, sizeof(int32_t) * FN * FK));
        CHECK_HIP(hipMalloc((void**)&di2, sizeof(int32_t) * FN * FK));
        CHECK_HIP(hipMemcpy(dxyz, hxyz, sizeof(float) * FN * 3, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dW1, hW1, sizeof(float) * 64 * 3, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dW2, hW2, sizeof(float) * 64 * 64, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(dsb, hs, sizeof(float) * 4 * 64, hipMemcpyHostToDevice));
        CHECK_LPD(lpd_lpdnet_front(dxyz, 3, dW1, dsb, dsb + 64, dW2, dsb + 128, dsb + 192, LPD_ACT_LEAKY, 0.2f, dF, 1, FN, FK, dkws, st));
        CHECK_LPD(lpd_knn_pm(NULL, 64, 1, 64, FN, FK, di1, dkws, LPD_KNN_PM_PREPARED, st));
        CHECK_LPD(lpd_knn_pm(dF, 64, 1, 64, FN, FK, di2, dkws2, 0, st));
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(hF, dF, sizeof(float) * FN * 64, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hi1, di1, sizeof(int32_t) * FN * FK, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hi2, di2, sizeof(int32_t) * FN * FK, hipMemcpyDeviceToHost));
        double ef = max_rel(hF, rF, FN * 64);
        int diff = memcmp(hi1, hi2, sizeof(int32_t) * FN * FK) != 0;
        printf("lpd_lpdnet_front rel err %.2e; prepared kNN graph %s lpd_knn_pm's\n", ef, diff ? "DIFFERS from" : "==");
        if (!(ef < 5e-6) || diff) return 9;
    }

    /* ---- error convention: a bad argument returns LPD_ERR_ARG with a message, no exception, no abort ---- */
    int rc = lpd_knn(dx, 1, 3, KN, KN + 1, didx, dws, 0, st);
    if (rc == LPD_OK || strlen(lpd_last_error()) == 0) { fprintf(stderr, "k > N was accepted\n"); return 8; }
    printf("lpd_knn(k > N) -> %d \"%s\"\n", rc, lpd_last_error());

    printf("abi_smoke OK\n");
    return 0;
}
