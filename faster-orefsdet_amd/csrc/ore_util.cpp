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

// Algorithmic work of the per-op conv entry points (ore_conv2d_fwd / _levels_fwd / ore_conv2d_wgrad*_fwd / ore_stem1*_fwd), summed by the
// host as the calls are made: 2 * rows * Cout * Cin * kh * kw (the DIRECT-convolution count, whatever kernel runs).  bench.py prices
// a training step with it (a data gradient is ore_conv2d_fwd on repacked weights, so it is counted as the conv it is).
#include <atomic>
static std::atomic<double> g_flops{0.0};
static std::atomic<long long> g_flop_calls{0};
void ore_flop_count_add(double flops) {
    double cur = g_flops.load(std::memory_order_relaxed);
    while (!g_flops.compare_exchange_weak(cur, cur + flops, std::memory_order_relaxed)) {}
    g_flop_calls.fetch_add(1, std::memory_order_relaxed);
}
// did the last per-op conv call of this thread run on the Winograd kernel (2.25x fewer multiplies than its algorithmic count)?
static thread_local int g_last_wino = 0;
void ore_note_wino(int v) { g_last_wino = v; }
int ore_last_wino(void) { return g_last_wino; }
extern "C" int ore_flop_counter_read(double* flops, int64_t* calls, int32_t reset) {
    if (flops) *flops = g_flops.load(std::memory_order_relaxed);
    if (calls) *calls = (int64_t)g_flop_calls.load(std::memory_order_relaxed);
    if (reset) { g_flops.store(0.0, std::memory_order_relaxed); g_flop_calls.store(0, std::memory_order_relaxed); }
    return ORE_OK;
}
extern "C" int32_t ore_det_record_rows(void) { return ORE_DET_RECORD_ROWS; }
extern "C" int ore_version(void) { return 504; }   // round*100 + revision: bumped whenever a kernel on the eval path changes (bench.py keys the PMC traffic file on it)
