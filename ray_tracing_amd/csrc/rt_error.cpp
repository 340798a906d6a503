#include "rt_error.h"

#include <cstdarg>
#include <cstdio>

#include "../../include/rt_amd.h"

static thread_local char g_rt_error[512] = "";

int rt_fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_rt_error, sizeof g_rt_error, fmt, ap);
    va_end(ap);
    return status;
}

extern "C" const char *rt_last_error(void) { return g_rt_error; }
extern "C" int rt_abi_version(void) { return RT_ABI_VERSION; }
