// Error plumbing and version of libd2s_hip.so.
#include "common.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace d2s {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

int hip_fail(hipError_t err, const char* what, const char* file, int line) {
    g_err = std::string("HIP error ") + hipGetErrorName(err) + " (" + hipGetErrorString(err) + ") in " + what +
            " at " + file + ":" + std::to_string(line);
    return D2S_E_HIP;
}

static std::atomic<int> g_env_gen{1};
int env_generation() { return g_env_gen.load(std::memory_order_relaxed); }
// Two host threads may enter a launcher at once (two engines, or the depth and the warp thread of the reference's main loop): the
// cached (generation, value) pair is read and refreshed under one lock -- a few nanoseconds per launch decision.
static std::mutex g_env_mu;
int EnvInt::get() {
    const int g = env_generation();
    std::lock_guard<std::mutex> lk(g_env_mu);
    if (gen != g) { const char* v = getenv(name); val = v ? atoi(v) : dflt; gen = g; }
    return val;
}

}  // namespace d2s

extern "C" int d2s_debug_lds_poison(void) {
#ifdef D2S_LDS_POISON
    return 1;
#else
    return 0;
#endif
}
extern "C" int d2s_debug_reload_env(void) { return d2s::g_env_gen.fetch_add(1) + 1; }
extern "C" const char* d2s_last_error(void) { return d2s::g_err.c_str(); }
extern "C" int d2s_version(void) { return 100; }
