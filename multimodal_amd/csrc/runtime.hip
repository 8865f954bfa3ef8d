// runtime.hip - error reporting shared by every C-ABI entry point.
#include "common.h"
#include <string.h>

namespace mm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local int g_cu_share = 0;
int cu_share() { return g_cu_share; }
}  // namespace mm

extern "C" int mm_set_cu_share(int max_workgroups) {
    const int prev = mm::g_cu_share;
    mm::g_cu_share = max_workgroups > 0 ? max_workgroups : 0;
    return prev;
}
extern "C" const char* mm_last_error(void) { return mm::g_err; }
extern "C" const char* mm_target_arch(void) { return "gfx950"; }
