// Error text plumbing shared by every entry point of the C-ABI.
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mars_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* marship_last_error(void) { return g_err; }
extern "C" int marship_version(void) { return 1; }
