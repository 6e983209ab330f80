// Thread-local error text behind fra_last_error().
#include <stdarg.h>

#include "common.h"

namespace fra {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_err; }

}  // namespace fra

extern "C" {
const char* fra_last_error(void) { return fra::last_error(); }
int fra_version(void) { return 1; }
}
