// Shared helpers for libd2s_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>

#include "../../include/d2s.h"

namespace d2s {

void set_error(const std::string& msg);
int hip_fail(hipError_t err, const char* what, const char* file, int line);

// Debug build (D2S_HIPCC_DEFS=-DD2S_LDS_POISON, `python -m desktop2stereo_amd.build --force`): every kernel that streams operands
// through an LDS ring fills its whole LDS allocation with NaN patterns (bf16 and fp32 alike) before it starts, so a fragment read
// that runs ahead of the LDS-DMA / staging write it depends on -- a slip in a hand-counted vmcnt or a missing barrier -- poisons
// the output instead of quietly reading the previous block's data.  The GPU parity suite is then the detector
// (tests/test_gpu_soak.py::test_lds_poison_build describes the procedure).  Persistent kernels are covered for their first tile.
#ifdef D2S_LDS_POISON
#define D2S_POISON_LDS(PTR, N16)                                                                                  \
    {                                                                                                             \
        for (int i_ = threadIdx.x; i_ < (int)(N16); i_ += blockDim.x)                                             \
            ((uint4*)(PTR))[i_] = make_uint4(0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u);                 \
        __syncthreads();                                                                                          \
    }
#else
#define D2S_POISON_LDS(PTR, N16) {}
#endif

// Kernel-argument warm-up for the latency-bound launches.  The kernarg segment is written by the host for every launch and read
// cold from device memory; the compiler fetches it lazily -- an s_load where a field is first needed, behind the branches of the
// tile map -- so a kernel with ~370 bytes of arguments pays three or four serialised ~0.5 us misses before its first operand
// request (gemm_glds_kernel: first LDS-DMA at instruction 775).  KERNARG_WARM(ka) at the top of the kernel requests one dword of
// each 64-byte line at once (six lines) and waits for them in the same asm statement (one ~0.5 us round trip instead of three or
// four); the compiler's own s_loads then hit the scalar cache.
struct KernargWarm { int d0, d1, d2, d3, d4, d5; };
#ifdef D2S_NO_KERNARG_WARM                       // (A/B builds)
#define KERNARG_WARM(ka) {}
#define KERNARG_WARM_END(ka) {}
#define KERNARG_WARM_BYTES 0                     // (nothing is read: the size asserts beside the kernels hold trivially)
#else
#define KERNARG_WARM(ka)                                                                                              \
    KernargWarm ka;                                                                                                   \
    {                                                                                                                 \
        auto p_ = __builtin_amdgcn_kernarg_segment_ptr();                                                             \
        /* loads AND their wait in ONE asm statement: the six destination SGPRs are written by the hardware when the data lands, */  \
        /* so they must not be visible to the register allocator as "defined" before that (it could copy or reuse them)          */  \
        asm volatile("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\t"          \
                     "s_load_dword %3, %6, 0xc0\n\ts_load_dword %4, %6, 0x100\n\ts_load_dword %5, %6, 0x140\n\t"      \
                     "s_waitcnt lgkmcnt(0)"                                                                           \
                     : "=&s"(ka.d0), "=&s"(ka.d1), "=&s"(ka.d2), "=&s"(ka.d3), "=&s"(ka.d4), "=&s"(ka.d5) : "s"(p_) : "memory");   \
    }
#define KERNARG_WARM_END(ka) {}
// the macro reads one dword at 0x140 of the kernarg segment: a kernel that uses it must have at least that many argument bytes
#define KERNARG_WARM_BYTES 0x144
#endif

// Integer switch from the environment (kernel selection for A/B runs and tests): cached, re-read after
// d2s_debug_reload_env() so that one process can run both sides.  Usage:  static EnvInt f{"D2S_NO_X", 0};  if (f.get()) ...
int env_generation();
struct EnvInt {
    const char* name; int dflt;
    std::atomic<uint64_t> cached{0};          // (generation << 32) | value: one relaxed load on the launch path, no lock
    int get();
};

#define D2S_HIP(call)                                                           \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return ::d2s::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

// status of the launch just made.  (hipGetLastError() is sticky on ROCm 7: it reports the last error ANY earlier runtime call of
// this host thread returned -- e.g. a probing call of the application -- and would blame it on this launch.)
#define D2S_CHECK_LAUNCH() D2S_HIP(hipExtGetLastError())

#define D2S_REQUIRE(cond, msg)                                                  \
    do {                                                                        \
        if (!(cond)) { ::d2s::set_error(std::string("invalid argument: ") + msg); return D2S_E_INVALID; } \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The HIP current device is per host thread; the reference issues predict_depth and make_sbs from different threads.
// Every entry point that takes an engine runs on the ENGINE's device whatever the calling thread's current device is, and
// leaves the thread's device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        (void)hipGetLastError();                  // errors left behind by earlier calls of this thread are not ours
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != dev) { ok = hipSetDevice(dev) == hipSuccess; prev = cur; }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define D2S_ON_DEVICE(dev)                                                                  \
    ::d2s::DeviceGuard _dev_guard(dev);                                                     \
    if (!_dev_guard.ok) { ::d2s::set_error("hipSetDevice failed for the engine's device"); return D2S_E_HIP; }

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

// ---- split precision ("bf16x3", D2S_PREC_BF16X3): an element is 4 bytes like a float, but a row is stored as 32-byte UNITS of 8
// elements, [8 x bf16 hi | 8 x bf16 lo] with hi = bf16(x), lo = bf16(x - hi), so that a 16-byte chunk is one MFMA operand
// (gemm_epi.h).  Pointer arithmetic on bx3_t* is the float layout's; where the halves of an element live follows from its address.
struct bx3_t { uint32_t bits; };
// store 4 consecutive elements (column a multiple of 4, rows 32-byte aligned): p = the float-layout address of the first one
__device__ static inline void bx3_store4(bx3_t* p, const float v[4]) {
    const uintptr_t a = (uintptr_t)p;
    uint2* hi = (uint2*)((a & ~(uintptr_t)31) + ((a >> 4) & 1) * 8);
    const bf16_t h0 = f2bf(v[0]), h1 = f2bf(v[1]), h2 = f2bf(v[2]), h3 = f2bf(v[3]);
    const bf16_t l0 = f2bf(v[0] - bf2f(h0)), l1 = f2bf(v[1] - bf2f(h1)), l2 = f2bf(v[2] - bf2f(h2)), l3 = f2bf(v[3] - bf2f(h3));
    hi[0] = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
    hi[2] = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));      // + 16 bytes: the lo chunk
}

// one element (any column): p = its float-layout address.  load: hi + lo.
__host__ __device__ static inline void bx3_store1(bx3_t* p, float v) {
    const uintptr_t a = (uintptr_t)p;
    bf16_t* u = (bf16_t*)(a & ~(uintptr_t)31) + ((a >> 2) & 7);
    const bf16_t h = f2bf(v);
    u[0] = h; u[8] = f2bf(v - bf2f(h));
}
__host__ __device__ static inline float bx3_load1(const bx3_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const bf16_t* u = (const bf16_t*)(a & ~(uintptr_t)31) + ((a >> 2) & 7);
    return bf2f(u[0]) + bf2f(u[8]);
}

// ---- fp8: OCP e4m3fn (gfx950's format; bias 7, max 448, no infinities, NaN = 0x7f) ----------------------------
typedef unsigned char fp8_t;     // raw e4m3 bits
constexpr float FP8_MAX = 448.0f;

// software encoder (host-side weight packing; also the reference the device path is tested against):
// round-to-nearest-even, saturating at +-448
__host__ __device__ static inline fp8_t f2e4m3(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const uint32_t sign = (v.u >> 24) & 0x80u;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (fp8_t)(sign | 0x7fu);
    v.u &= 0x7fffffffu;
    float a = v.f;
    if (a >= FP8_MAX) return (fp8_t)(sign | 0x7eu);
    if (a < 0.015625f) {                                  // below the smallest normal 2^-6: multiples of 2^-9
        float q = a * 512.0f + 12582912.0f;               // 1.5 * 2^23: the add rounds to nearest-even integer
        q -= 12582912.0f;
        return (fp8_t)(sign | (uint32_t)(int)q);          // 8 is 0x08, the smallest normal
    }
    uint32_t u = v.u;
    u += 0x7ffffu + ((u >> 20) & 1u);                      // RNE to 3 mantissa bits
    const uint32_t e = (u >> 23) - 127u + 7u, m = (u >> 20) & 7u;
    return (fp8_t)(sign | (e << 3) | m);
}
__host__ __device__ static inline float e4m32f(fp8_t b) {
    const uint32_t e = (b >> 3) & 15u, m = b & 7u;
    float a;
    if (e == 0) a = (float)m * 0.001953125f;               // m * 2^-9
    else { union { float f; uint32_t u; } v; v.u = ((e - 7u + 127u) << 23) | (m << 20); a = v.f; }
    return (b & 0x80u) ? -a : a;
}

// last activation of the DPT head (HF DepthAnythingDepthEstimationHead.forward): ReLU for relative models,
// sigmoid(x) * max_depth for metric ones (depth_estimation_type == "metric")
__device__ static inline float head_activation(float v, float max_depth) {
    return max_depth > 0.f ? (1.0f / (1.0f + expf(-v))) * max_depth : fmaxf(v, 0.f);
}

// ---- bilinear source taps, torch semantics (ATen area_pixel_compute_source_index) -------------
struct Tap { int i0, i1; float w0, w1; };

__host__ __device__ static inline Tap linear_tap(int dst, float scale, int in_size, bool align_corners) {
    // no FMA contraction in here: with it, w1 = src - i0 becomes fma(scale, dst, -i0) in one kernel and round(scale * dst) - i0 in
    // another, depending on what else the inlining context does with src (the stand-alone up-sample and the loader that folds it
    // into the head convolution differed in a quarter of the output pixels, by one bf16 ulp of an input); the reference multiplies
    // and subtracts separately
#pragma clang fp contract(off)
    float src;
    if (align_corners) src = scale * (float)dst;
    else { src = scale * ((float)dst + 0.5f) - 0.5f; if (src < 0.f) src = 0.f; }
    int i0 = (int)src;                       // src >= 0: trunc == floor
    if (i0 > in_size - 1) i0 = in_size - 1;
    int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    Tap t; t.i0 = i0; t.i1 = i1; t.w1 = src - (float)i0; t.w0 = 1.0f - t.w1;
    return t;
}
// one bilinear sample: the expression every up-sample in the library evaluates, spelled with explicit fused multiply-adds so that the
// stand-alone kernel (vit_ops.hip) and the halo loader that folds the up-sample into a convolution (conv3.hip, two values at a time
// as v_pk_fma_f32) round identically whatever the compiler would contract on its own
__device__ static inline float bilerp1(const Tap& tx, const Tap& ty, float v00, float v01, float v10, float v11) {
    const float top = fmaf(tx.w1, v01, tx.w0 * v00);
    const float bot = fmaf(tx.w1, v11, tx.w0 * v10);
    return fmaf(ty.w1, bot, ty.w0 * top);
}
static inline float linear_scale(int in_size, int out_size, bool align_corners) {
    if (align_corners) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    return (float)in_size / (float)out_size;
}

}  // namespace d2s
