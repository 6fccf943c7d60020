// Micro-benchmark: what does HBM sustain for the stereo warp's traffic mix?  Every thread reads 12 bytes of a source row and
// writes 12 bytes into each half of a double-width output row (Full-SBS: 1 byte read : 2 bytes written), no arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 rw_mix.hip -o rw_mix ; ./rw_mix [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
struct U3 { uint32_t x, y, z; };
__global__ void __launch_bounds__(256) rw_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int W, int nt) {
    // one block per (row, 1024-pixel tile) item, grid-strided: same access shape as round 5's stereo_warp_lanes
    const int tiles = (W + 1023) / 1024;
    for (long item = blockIdx.x; item < (long)rows * tiles; item += gridDim.x) {
        const long row = item / tiles; const int tile = (int)(item - row * tiles);
        const int x = tile * 1024 + threadIdx.x * 4;
        if (x >= W) continue;
        U3 v = *(const U3*)(src + (row * W + x) * 3);
        if (nt) {
            __builtin_nontemporal_store(v.x, (uint32_t*)(dst + (row * 2 * W + x) * 3)); __builtin_nontemporal_store(v.y, (uint32_t*)(dst + (row * 2 * W + x) * 3) + 1);
            __builtin_nontemporal_store(v.z, (uint32_t*)(dst + (row * 2 * W + x) * 3) + 2);
            __builtin_nontemporal_store(v.x, (uint32_t*)(dst + (row * 2 * W + W + x) * 3)); __builtin_nontemporal_store(v.y, (uint32_t*)(dst + (row * 2 * W + W + x) * 3) + 1);
            __builtin_nontemporal_store(v.z, (uint32_t*)(dst + (row * 2 * W + W + x) * 3) + 2);
        } else {
            *(U3*)(dst + (row * 2 * W + x) * 3) = v;
            *(U3*)(dst + (row * 2 * W + W + x) * 3) = v;
        }
    }
}
// the same bytes as flat 16-byte accesses (what a memcpy-shaped kernel gets)
__global__ void __launch_bounds__(256) rw_flat(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) { uint4 v = src[i]; dst[2 * i] = v; dst[2 * i + 1] = v; }
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, H = 1080, W = 1920;
    const long rows = (long)B * H, sb = rows * W * 3;
    uint8_t *src, *dst; hipMalloc(&src, sb); hipMalloc(&dst, 2 * sb); hipMemset(src, 1, sb); hipMemset(dst, 0, 2 * sb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt)
        for (int bpc : {2, 4, 8, 16}) {
            const int grid = 256 * bpc;
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rw_rows, dim3(grid), dim3(256), 0, 0, src, dst, (int)rows, W, nt);
            hipEventRecord(a); const int it = 20;
            for (int i = 0; i < it; ++i) hipLaunchKernelGGL(rw_rows, dim3(grid), dim3(256), 0, 0, src, dst, (int)rows, W, nt);
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
            printf("rows  nt=%d blocks/CU %2d: %7.1f us  %6.2f TB/s (%.1f MB)\n", nt, bpc, ms * 1e3, 3.0 * sb / ms / 1e9, 3.0 * sb / 1e6);
        }
    for (int bpc : {4, 8, 16}) {
        const int grid = 256 * bpc;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rw_flat, dim3(grid), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, sb / 16);
        hipEventRecord(a); const int it = 20;
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(rw_flat, dim3(grid), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, sb / 16);
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
        printf("flat        blocks/CU %2d: %7.1f us  %6.2f TB/s\n", bpc, ms * 1e3, 3.0 * sb / ms / 1e9);
    }
    return 0;
}
