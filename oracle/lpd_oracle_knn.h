/*
 * oracle/lpd_oracle_knn.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * The bit-critical kNN arithmetic of the reference (util/lpdnet_model.py:317-326), shared by lpd_oracle.c (kNN op oracle) and
 * lpd_forward.c (whole-path restatement).  Must be compiled with -ffp-contract=off.
 */
#ifndef LPD_ORACLE_KNN_H
#define LPD_ORACLE_KNN_H
#include <math.h>

/* sum of squares of one point's C channels, torch CPU order (lpdnet_model.py:320). */
static inline float oracle_sumsq(const float *p, int C)
{
    float total = 0.0f;
    int first_block = 1;
    for (int c0 = 0; c0 < C; c0 += 16) {
        int c1 = c0 + 16 < C ? c0 + 16 : C;
        float acc = p[c0] * p[c0];
        for (int c = c0 + 1; c < c1; ++c) {
            float sq = p[c] * p[c];
            acc = acc + sq;
        }
        if (first_block) { total = acc; first_block = 0; }
        else total = total + acc;
    }
    return total;
}

/* pd[i][j] exactly as the reference computes it (lpdnet_model.py:318-324). */
static inline float oracle_pd(const float *xi, const float *xj, float xxi, float xxj, int C)
{
    float dot = 0.0f;
    for (int c = 0; c < C; ++c) dot = fmaf(xi[c], xj[c], dot);
    float inner = -2.0f * dot;
    float t = (-xxj) - inner;
    return t - xxi;
}

/* better(a,ia, b,ib): does candidate a rank before b? */
static inline int oracle_before(float a, int ia, float b, int ib)
{
    return (a > b) || (a == b && ia < ib);
}


#endif
