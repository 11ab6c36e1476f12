// Thread-local error message and version for libsfhip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sfhip.h"

static thread_local char g_err[512] = "";

void sf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sf_last_error(void) { return g_err; }
extern "C" int sf_version(void) { return 1; }
