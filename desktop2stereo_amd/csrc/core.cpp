// Error plumbing and version of libd2s_hip.so.
#include "common.h"

#include <mutex>

namespace d2s {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

int hip_fail(hipError_t err, const char* what, const char* file, int line) {
    g_err = std::string("HIP error ") + hipGetErrorName(err) + " (" + hipGetErrorString(err) + ") in " + what +
            " at " + file + ":" + std::to_string(line);
    return D2S_E_HIP;
}

}  // namespace d2s

extern "C" const char* d2s_last_error(void) { return d2s::g_err.c_str(); }
extern "C" int d2s_version(void) { return 100; }
