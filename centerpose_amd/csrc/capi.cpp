// C-ABI plumbing shared by every entry point of libcenterpose_hip.so: version + last-error string.
#include <cstdarg>
#include <cstdio>
#include "common.h"

static thread_local char g_err[512] = "";

void cp_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cp_last_error(void) { return g_err; }
extern "C" int cp_abi_version(void) { return 2; }      // 2: plan handle (cp_plan_*), decode of any map size
extern "C" const char* cp_target_arch(void) { return "gfx950"; }
