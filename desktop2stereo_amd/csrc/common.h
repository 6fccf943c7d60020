// Shared helpers for libd2s_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/d2s.h"

namespace d2s {

void set_error(const std::string& msg);
int hip_fail(hipError_t err, const char* what, const char* file, int line);

#define D2S_HIP(call)                                                           \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return ::d2s::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define D2S_CHECK_LAUNCH() D2S_HIP(hipGetLastError())

#define D2S_REQUIRE(cond, msg)                                                  \
    do {                                                                        \
        if (!(cond)) { ::d2s::set_error(std::string("invalid argument: ") + msg); return D2S_E_INVALID; } \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef unsigned short bf16_t;   // raw bf16 bits

__host__ __device__ static inline bf16_t f2bf(float f) {
    // round-to-nearest-even, NaN preserved
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ static inline float bf2f(bf16_t h) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)h) << 16; return v.f;
}

// last activation of the DPT head (HF DepthAnythingDepthEstimationHead.forward): ReLU for relative models,
// sigmoid(x) * max_depth for metric ones (depth_estimation_type == "metric")
__device__ static inline float head_activation(float v, float max_depth) {
    return max_depth > 0.f ? (1.0f / (1.0f + expf(-v))) * max_depth : fmaxf(v, 0.f);
}

// ---- bilinear source taps, torch semantics (ATen area_pixel_compute_source_index) -------------
struct Tap { int i0, i1; float w0, w1; };

__host__ __device__ static inline Tap linear_tap(int dst, float scale, int in_size, bool align_corners) {
    float src;
    if (align_corners) src = scale * (float)dst;
    else { src = scale * ((float)dst + 0.5f) - 0.5f; if (src < 0.f) src = 0.f; }
    int i0 = (int)src;                       // src >= 0: trunc == floor
    if (i0 > in_size - 1) i0 = in_size - 1;
    int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    Tap t; t.i0 = i0; t.i1 = i1; t.w1 = src - (float)i0; t.w0 = 1.0f - t.w1;
    return t;
}
static inline float linear_scale(int in_size, int out_size, bool align_corners) {
    if (align_corners) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    return (float)in_size / (float)out_size;
}

}  // namespace d2s
