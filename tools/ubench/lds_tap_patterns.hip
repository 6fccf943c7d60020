// Round 6: LDS cost of the stereo warp's tap pair (pixels x0, x0 + 1 as RGBX dwords = 8 bytes at a 4-byte aligned address) for the
// candidate pixel->lane maps.  One wave owns a 1.5 KiB window; every lane reads 8 pairs per "row".
//   A  lane owns 4 consecutive pixels (x0 = 4 lane + s): stride-4 dwords across lanes      ds_read_b64 / ds_read2_b32
//   B  same, window stored with 2 pad dwords per 32 pixels (pos = x + 2 (x >> 5))           ds_read2_b32
//   E / F  as B with one pad dword per 32 / per 16 pixels
//   C  lane-strided (x0 = lane + 64 k + s): consecutive lanes, consecutive dwords           ds_read_b64 / ds_read2_b32
//   D  lane owns 2 consecutive pixels, twice (x0 = 2 lane + 128 h + s): stride 2            ds_read2_b32
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_tap_patterns.hip -o /tmp/lds_tap_patterns && /tmp/lds_tap_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int PAT, int INSTR>
__global__ void __launch_bounds__(256) k(uint32_t* __restrict__ out, int rows, int s) {
    __shared__ uint32_t win[4][448];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 448; i += 64) win[wv][i] = i * 2654435761u;
    const uint32_t base = (uint32_t)(size_t)&win[wv][0];
    uint32_t acc = 0;
    for (int r = 0; r < rows; ++r) {
        uint2 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sj = s + (j >> 2) * 3 + ((lane * 7 + r) & 1);            // two eyes' shifts, a little per-lane variation
            int x0;
            if (PAT == 0 || PAT == 1 || PAT == 4 || PAT == 5) x0 = 4 * lane + (j & 3) + sj;
            else if (PAT == 2) x0 = lane + 64 * (j & 3) + sj;
            else x0 = 2 * lane + 128 * ((j >> 1) & 1) + (j & 1) + sj;
            const int pos = PAT == 1 ? x0 + 2 * (x0 >> 5) : (PAT == 4 ? x0 + (x0 >> 5) : (PAT == 5 ? x0 + (x0 >> 4) : x0));
            const uint32_t a = base + 4u * (uint32_t)pos;
            if (INSTR == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(t[j]) : "v"(a) : "memory");
            else asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t[j]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(t[j])); acc += t[j].x ^ t[j].y; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int rows = 256, blocks = 256 * 8;
    uint32_t* o; CK(hipMalloc(&o, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) -> int {
        for (int s = 0; s < 2; ++s) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, rows, s);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, rows, s);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double taps = (double)blocks * 4 * rows * 8;
            printf("%-44s s=%d: %7.1f us, %5.1f CU-cycles per wave tap pair\n", name, s, ms / 10 * 1e3, ms / 10 * 1e-3 * 2.4e9 * 256 / taps);
        }
        return 0;
    };
    run("A stride 4, ds_read_b64", k<0, 0>); run("A stride 4, ds_read2_b32", k<0, 1>);
    run("B stride 4 padded 2/32, ds_read_b64", k<1, 0>); run("B stride 4 padded 2/32, ds_read2_b32", k<1, 1>);
    run("E stride 4 padded 1/32, ds_read2_b32", k<4, 1>); run("F stride 4 padded 1/16, ds_read2_b32", k<5, 1>);
    run("C lane-strided, ds_read_b64", k<2, 0>); run("C lane-strided, ds_read2_b32", k<2, 1>);
    run("D stride 2, ds_read_b64", k<3, 0>); run("D stride 2, ds_read2_b32", k<3, 1>);
    return 0;
}
