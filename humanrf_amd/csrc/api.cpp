// Error channel and version of the C ABI (include/hrf.h).
#include <cstdarg>
#include <cstdio>
#include "../../include/hrf.h"

static thread_local char g_err[512] = "";

void hrf_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* hrf_last_error(void) { return g_err; }
extern "C" int hrf_abi_version(void) { return HRF_ABI_VERSION; }
