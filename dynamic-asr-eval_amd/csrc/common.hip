// Error plumbing + identity queries of the C-ABI (see include/dyneval.h).
#include "common.h"
#include <string.h>

namespace dyn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dyn

extern "C" const char* dyn_last_error(void) { return dyn::g_err; }
extern "C" const char* dyn_version(void) { return "dyneval-hip 0.1"; }
extern "C" const char* dyn_arch(void) { return "gfx950"; }
