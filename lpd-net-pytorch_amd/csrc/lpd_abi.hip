// lpd_abi.hip -- error text + version for the C-ABI (include/lpd_hip.h).
#include "lpd_common.h"
#include <stdarg.h>

static thread_local char g_lpd_err[512] = "";

void lpd_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_lpd_err, sizeof(g_lpd_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lpd_last_error(void) { return g_lpd_err; }
extern "C" int lpd_version(void) { return 100; }  // 0.1.0

// ---- scratch of the cross-block column statistics (lpd_common.h): one per (device, stream), never freed ----
#include <map>
#include <mutex>
#include <utility>

LpdStatWs lpd_stat_ws(hipStream_t stream)
{
    static std::mutex mu;
    static std::map<std::pair<int, void*>, LpdStatWs> table;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, (void*)stream);
    auto it = table.find(key);
    if (it != table.end()) return it->second;
    LpdStatWs ws = {nullptr};
    const size_t bytes = sizeof(double) * LPD_STAT_REPLICAS * 2 * LPD_STAT_CMAX;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) {   // hipMemset: synchronous, once per stream
        (void)hipGetLastError();
        return ws;
    }
    ws.rep = reinterpret_cast<double*>(p);
    table[key] = ws;
    return ws;
}

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
