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
extern "C" int32_t ore_det_record_rows(void) { return ORE_DET_RECORD_ROWS; }
extern "C" int ore_version(void) { return 406; }   // round*100 + revision: bumped whenever a kernel on the eval path changes (bench.py keys the PMC traffic file on it)
