// Sanitizer check of the product library's host-only translation unit (csrc/ore_util.cpp): the last-error string is bounded.
// TEST INFRASTRUCTURE ONLY (linked into oracle/_build/asan_driver by `make -C oracle asan`).
#include <string.h>
void ore_set_error(const char* fmt, ...);
extern "C" const char* ore_last_error(void);
extern "C" int ore_version(void);
extern "C" int ore_util_check(void) {
    char big[2048];
    memset(big, 'x', sizeof(big) - 1);
    big[sizeof(big) - 1] = 0;
    ore_set_error("%s %d", big, 7);
    return (strlen(ore_last_error()) < 512 && ore_version() >= 300) ? 0 : 1;
}
