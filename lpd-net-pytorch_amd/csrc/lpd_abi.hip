// lpd_abi.hip -- error text + version for the C-ABI (include/lpd_hip.h).
#include "lpd_common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

static thread_local char g_lpd_err[512] = "";

void lpd_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_lpd_err, sizeof(g_lpd_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lpd_last_error(void) { return g_lpd_err; }

// LPD_DEBUG tokens (lpd_common.h): parsed on every call -- the callers keep the result in a function-local static
int lpd_debug(const char* name, int dflt)
{
    const char* e = getenv("LPD_DEBUG");
    if (!e) return dflt;
    const size_t nl = strlen(name);
    for (const char* p = e; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        const char* q = p;
        while (*q && *q != ',') ++q;
        const size_t len = (size_t)(q - p);
        if (len >= nl && strncasecmp(p, name, nl) == 0) {
            if (len == nl) return 1;
            if (p[nl] == '=') return atoi(p + nl + 1);
        }
        if (len == nl + 3 && strncasecmp(p, "no-", 3) == 0 && strncasecmp(p + 3, name, nl) == 0) return 0;
        p = q;
    }
    return dflt;
}
extern "C" int lpd_version(void) { return 100; }  // 0.1.0

// ---- workspace of the cross-block column statistics (lpd_common.h): owned by the caller ----
extern "C" long long lpd_stat_ws_bytes(void) { return (long long)sizeof(double) * LPD_STAT_REPLICAS * 2 * LPD_STAT_CMAX; }

namespace {
__global__ void stat_gather_kernel(double* __restrict__ rep, double* __restrict__ o0, double* __restrict__ o1, int ncols)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * ncols) return;
    const int which = c >= ncols, col = which ? c - ncols : c;
    double t = 0.0;
#pragma unroll 8
    for (int r = 0; r < LPD_STAT_REPLICAS; ++r) {
        double* p = rep + ((size_t)r * 2 + which) * LPD_STAT_CMAX + col;
        t += *p;
        *p = 0.0;
    }
    (which ? o1 : o0)[col] = t;
}
}  // namespace

int lpd_stat_finish(LpdStatWs ws, double* o0, double* o1, int ncols, hipStream_t stream)
{
    hipLaunchKernelGGL(stat_gather_kernel, dim3((2 * ncols + 255) / 256), dim3(256), 0, stream, ws.rep, o0, o1, ncols);
    LPD_CHECK_LAUNCH("lpd_stat_finish");
    return LPD_OK;
}
