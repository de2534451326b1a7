// last-error string + version for libore_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "ore_hip.h"

static thread_local char g_err[512] = "";

void ore_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ore_last_error(void) { return g_err; }
extern "C" int ore_version(void) { return 100; }
