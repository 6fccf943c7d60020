// Error plumbing and version of libd2s_hip.so.
#include "common.h"

#include <atomic>
#include <cstdlib>

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
// cached (generation, value) pair is ONE 64-bit atomic -- the hot read is a relaxed load and a compare; only the first read after
// d2s_debug_reload_env() calls getenv (two racing refreshers store the same pair).
int EnvInt::get() {
    const uint32_t g = (uint32_t)env_generation();
    const uint64_t c = cached.load(std::memory_order_relaxed);
    if ((uint32_t)(c >> 32) == g) return (int)(uint32_t)c;
    const char* v = getenv(name);
    const int val = v ? atoi(v) : dflt;
    cached.store(((uint64_t)g << 32) | (uint32_t)val, std::memory_order_relaxed);
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
// 110 (round 6): d2s_dibr_params carries struct_size (its last 4 bytes; sizeof unchanged = 80) and d2s_dibr_warp rejects a struct
// that does not say 80 -- a caller built against the 72-byte header of version 100 is refused instead of being read past its end.
extern "C" int d2s_version(void) { return 110; }
