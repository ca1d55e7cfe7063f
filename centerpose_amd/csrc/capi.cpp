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

// name of the device kernel the calling thread launched last (as rocprofv3 --kernel-trace prints it): lets a host-side
// profiler attribute an entry point's time to the template instantiation the dispatch heuristics picked
static thread_local char g_kernel[160] = "";
void cp_note_kernel(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
extern "C" const char* cp_last_kernel(void) { return g_kernel; }
extern "C" int cp_abi_version(void) { return 4; }      // 2: plan handle (cp_plan_*), decode of any map size; 3: cp_dcn_desc.ksplit; 4: cp_dcn_desc.dg
extern "C" const char* cp_target_arch(void) { return "gfx950"; }
