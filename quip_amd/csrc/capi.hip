// capi.hip -- version / error plumbing of the C ABI (include/quip_amd.h)
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "prefetch.h"

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

// ---- operand prefetch riding on the next decode launch (csrc/prefetch.h) ------------------------------------------------------------
static thread_local QaPfList g_pf_pending = {};

QaPfList qa_pf_take()
{
    QaPfList l = g_pf_pending;
    g_pf_pending.n = 0;
    return l;
}

extern "C" int quipamd_decode_prefetch_next(const void *const *ptrs, const int64_t *bytes, int n)
{
    QA_REQUIRE(n >= 0 && n <= QA_PF_MAX, QUIPAMD_ERR_ARG, "decode_prefetch_next: 0..%d ranges (n = %d)", QA_PF_MAX, n);
    QA_REQUIRE(n == 0 || (ptrs && bytes), QUIPAMD_ERR_ARG, "decode_prefetch_next: null array");
    QaPfList l = {};
    for (int i = 0; i < n; ++i) {
        QA_REQUIRE(ptrs[i] && bytes[i] >= 0 && bytes[i] < ((int64_t)1 << 30), QUIPAMD_ERR_ARG, "decode_prefetch_next: range %d (null, negative or >= 1 GiB)", i);
        if (bytes[i] == 0) continue;
        l.ptr[l.n] = ptrs[i];
        l.bytes[l.n] = (uint32_t)bytes[i];
        ++l.n;
    }
    g_pf_pending = l;
    return QUIPAMD_OK;
}

extern "C" int quipamd_version(void) { return QUIPAMD_VERSION; }
extern "C" const char *quipamd_last_error(void) { return g_err; }
