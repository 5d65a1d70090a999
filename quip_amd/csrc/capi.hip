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

#ifdef QA_PROBE
#include <vector>
typedef void (*qa_probe_setter)(unsigned long long *);
static std::vector<qa_probe_setter> &qa_probe_setters()
{
    static std::vector<qa_probe_setter> v;
    return v;
}
void qa_probe_register(qa_probe_setter s) { qa_probe_setters().push_back(s); }
#endif

// phase stamps of the decode launches (csrc/probe.h): buf = device memory for 16 waves x 16 slots of uint64, or NULL to switch them off.
// The shipped library is built without the stamps and says so.
extern "C" int quipamd_probe_set(void *buf)
{
#ifdef QA_PROBE
    for (qa_probe_setter s : qa_probe_setters()) s((unsigned long long *)buf);
    return QUIPAMD_OK;
#else
    (void)buf;
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "probe_set: this library carries no phase stamps (build libquip_amd_probe.so: python __graft_entry__.py --probe)");
#endif
}

extern "C" int quipamd_version(void) { return QUIPAMD_VERSION; }
extern "C" const char *quipamd_last_error(void) { return g_err; }
