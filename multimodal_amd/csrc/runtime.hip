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
}  // namespace mm

extern "C" const char* mm_last_error(void) { return mm::g_err; }
extern "C" const char* mm_target_arch(void) { return "gfx950"; }
