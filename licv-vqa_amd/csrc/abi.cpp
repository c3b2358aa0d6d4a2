// Error plumbing + version of the C-ABI (host only).
#include <cstdarg>
#include <cstdio>
#include "licv_hip.h"

static thread_local char g_err[512] = "";

int licv_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int licv_version(void) { return LICV_ABI_VERSION; }
extern "C" const char* licv_last_error(void) { return g_err; }
