// capi.hip -- version / error plumbing of the C ABI (include/quip_amd.h)
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

int qa_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int quipamd_version(void) { return QUIPAMD_VERSION; }
extern "C" const char *quipamd_last_error(void) { return g_err; }
