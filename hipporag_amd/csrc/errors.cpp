// Thread-local error text behind hrag_last_error().
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace hrag {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }
}  // namespace hrag

extern "C" const char *hrag_last_error(void) { return hrag::get_error(); }
extern "C" int hrag_version(void) { return HRAG_VERSION_MAJOR * 1000 + HRAG_VERSION_MINOR; }
