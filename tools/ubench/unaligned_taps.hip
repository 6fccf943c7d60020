// Round 6: what does an 8-byte tap at an ARBITRARY byte address cost, from global memory (L1/L2-resident row) and from LDS?
// A wave's lanes read 8 bytes at byte 12 * lane + off (off = 0..3: the stereo warp's 3 * x0 with 4 pixels per lane).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/unaligned_taps.hip -o /tmp/unaligned_taps && /tmp/unaligned_taps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint2 gload8(const uint8_t* p) { uint2 d; __builtin_memcpy(&d, p, 8); return d; }

// global: every wave walks `rows` rows of `pitch` bytes; 8 taps per lane per row at 12 * lane + off + 3 * j
__global__ void __launch_bounds__(256) k_global(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, int rows, int pitch, int off, int aligned) {
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint8_t* base = src + (size_t)(w % 64) * rows * pitch;
    uint32_t acc = 0;
    for (int r = 0; r < rows; ++r) {
        const uint8_t* row = base + (size_t)r * pitch;
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int a = 12 * lane + off + 3 * j;
            if (aligned) a &= ~3;
            t[j] = gload8(row + a);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += t[j].x ^ t[j].y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// LDS: a wave stages its 1 KiB row segment (aligned 16-byte loads -> ds_write_b128), then 8 taps per lane as 8-byte LDS reads at any byte
__device__ __forceinline__ uint2 lds_read8(uint32_t addr) {
    uint2 d;
    asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr) : "memory");
    return d;
}
__global__ void __launch_bounds__(256) k_lds(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, int rows, int pitch, int off, int aligned, int check) {
    __shared__ __attribute__((aligned(16))) uint8_t seg[4][1280];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x * 4 + wv;
    const uint8_t* base = src + (size_t)(w % 64) * rows * pitch;
    uint32_t acc = 0;
    const uint32_t sbase = (uint32_t)(size_t)&seg[wv][0];
    for (int r = 0; r < rows; ++r) {
        const uint8_t* row = base + (size_t)r * pitch;
        *(uint4*)&seg[wv][16 * lane] = *(const uint4*)(row + 16 * lane);            // 1024 bytes per wave-row
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int a = 12 * lane + off + 3 * j;
            if (aligned) a &= ~7;
            t[j] = lds_read8(sbase + a);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(t[j]));
        if (check && r == 0) {                       // lane's tap 3 must be the bytes of the row at that address
            uint2 want = gload8(row + 12 * lane + off + 9);
            if (want.x != t[3].x || want.y != t[3].y) acc |= 0x80000000u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (t[j].x ^ t[j].y) & 0x7fffffu;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int rows = 64, pitch = 1280, blocks = 256 * 8;
    std::vector<uint8_t> h((size_t)64 * rows * pitch + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131u + (i >> 8) * 7u);
    uint8_t* d; uint32_t* o;
    CK(hipMalloc(&d, h.size())); CK(hipMalloc(&o, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint32_t> ho((size_t)blocks * 256);
    for (int which = 0; which < 2; ++which)
        for (int aligned = 1; aligned >= 0; --aligned)
            for (int off = 0; off < 4; ++off) {
                if (aligned && off) continue;
                auto launch = [&](int check) {
                    if (which == 0) hipLaunchKernelGGL(k_global, dim3(blocks), dim3(256), 0, 0, d, o, rows, pitch, off, aligned);
                    else hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, d, o, rows, pitch, off, aligned, check);
                };
                launch(1); CK(hipDeviceSynchronize());
                CK(hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost));
                int bad = 0;
                if (which == 1 && !aligned) for (uint32_t v : ho) bad += (v >> 31);
                for (int i = 0; i < 3; ++i) launch(0);
                CK(hipEventRecord(e0));
                for (int i = 0; i < 10; ++i) launch(0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double taps = (double)blocks * 4 * rows * 8;       // wave-level tap instructions per launch
                const double cyc = ms / 10 * 1e-3 * 2.4e9 * 256 / taps;  // CU-cycles per wave-level tap instruction (2.4 GHz, 256 CUs)
                printf("%-6s %-9s off %d: %7.1f us per launch, %5.1f CU-cycles per wave tap instruction%s\n", which ? "LDS" : "global",
                       aligned ? "aligned" : "unaligned", off, ms / 10 * 1e3, cyc, bad ? "  ** WRONG BYTES **" : (which == 1 && !aligned ? "  (bytes verified)" : ""));
            }
    return 0;
}
