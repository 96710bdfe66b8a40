// Thread-local last-error string for the C ABI (no exceptions cross the boundary).
#include <cstdarg>
#include <cstdio>
namespace { thread_local char g_err[512] = ""; }
void cs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cs_last_error() { return g_err; }
