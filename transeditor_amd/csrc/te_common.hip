// Error reporting + version entry points of the C ABI (include/te_hip.h).
#include "te_common.h"
#include <stdarg.h>

namespace te {
static thread_local char g_err[512] = "no error";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace te

extern "C" int te_version(void) { return TE_ABI_VERSION; }
extern "C" const char* te_last_error_string(void) { return te::err_buf(); }
extern "C" const char* te_arch(void) { return "gfx950"; }
