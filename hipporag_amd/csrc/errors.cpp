// Thread-local error text behind hrag_last_error().
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

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

// Measurement switches read from the environment (A/B runs of bench.py / tools/): each variable is read ONCE per process
// (a later change of the environment cannot change the numerics of a running server, and ranks that were started with
// the same environment stay bit-identical), and the first time one is found SET the library says so on stderr -- an
// override that changes a plan or a kernel choice must never be silent (round-5 advice).
const char *experiment_env(const char *name) {
    struct Slot { const char *name; const char *value; bool read; };
    static Slot slots[16] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (Slot &s : slots) {
        if (s.read && std::strcmp(s.name, name) == 0) return s.value;
        if (!s.read) {
            const char *v = std::getenv(name);
            s.name = name;                      // callers pass string literals
            s.value = (v && v[0]) ? strdup(v) : nullptr;
            s.read = true;
            if (s.value) std::fprintf(stderr, "libhrag: experiment override %s=%s is active (measurement switch, not a product setting)\n", name, s.value);
            return s.value;
        }
    }
    return std::getenv(name);                   // table full: plain lookup
}
}  // namespace hrag

extern "C" const char *hrag_last_error(void) { return hrag::get_error(); }
extern "C" int hrag_version(void) { return HRAG_VERSION_MAJOR * 1000 + HRAG_VERSION_MINOR; }
