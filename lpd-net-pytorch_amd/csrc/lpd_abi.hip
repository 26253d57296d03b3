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
