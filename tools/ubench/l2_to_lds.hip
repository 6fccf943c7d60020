// Micro-benchmark: sustained global(L2-resident) -> LDS bandwidth with the GEMM's access pattern.
// Each block streams `tiles` stages of ROWS x 128 B (row stride `ld` bytes) via global_load_lds (mode 0)
// or global_load_dwordx4 + ds_write_b128 (mode 1); no MFMA.  Build: hipcc --offload-arch=gfx950 -O3 l2_to_lds.hip -o l2_to_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ROWS, int NS, int MODE>
__global__ void __launch_bounds__(256) stream_kernel(const char* __restrict__ src, long ld, int rows_total, int tiles, unsigned* sink) {
    constexpr int STAGE = ROWS * 8;
    constexpr int LPT = ROWS / 32;
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block b starts at a different row block; rows wrap inside rows_total (L2/MALL resident set)
    long row0 = ((long)blockIdx.x * ROWS) % (rows_total - ROWS);
    const int lrow = wid * 8 + (lane >> 3), chunk = (lane & 7) ^ ((lrow >> 1) & 7);
    unsigned acc = 0;
    if (MODE == 0) {
        for (int t = 0; t < NS - 1 && t < tiles; ++t)
            for (int i = 0; i < LPT; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (row0 + i * 32 + lrow) * ld + t * 128 + chunk * 16),
                                                 (__attribute__((address_space(3))) void*)(lds + (t % NS) * STAGE + (i * 4 + wid) * 64), 16, 0, 0);
        for (int t = 0; t < tiles; ++t) {
            if (t + NS - 2 < tiles) wait_vmcnt<(NS - 2) * LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            int tn = t + NS - 1;
            if (tn < tiles)
                for (int i = 0; i < LPT; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (row0 + i * 32 + lrow) * ld + (long)tn * 128 + chunk * 16),
                                                     (__attribute__((address_space(3))) void*)(lds + (tn % NS) * STAGE + (i * 4 + wid) * 64), 16, 0, 0);
            acc += lds[(t % NS) * STAGE + tid].x;                 // touch the landed stage
        }
    } else {
        u32x4 r[LPT];
        for (int t = 0; t < tiles; ++t) {
            for (int i = 0; i < LPT; ++i) r[i] = *(const u32x4*)(src + (row0 + i * 32 + lrow) * ld + (long)t * 128 + (lane & 7) * 16);
            for (int i = 0; i < LPT; ++i) lds[(t & 1) * STAGE + (i * 32 + lrow) * 8 + chunk] = r[i];
            __syncthreads();
            acc += lds[(t & 1) * STAGE + tid].x;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int ROWS, int NS, int MODE>
void run(const char* name, const char* d, long ld, int rows_total, int tiles, int blocks, unsigned* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<ROWS, NS, MODE>), dim3(blocks), dim3(256), 0, 0, d, ld, rows_total, tiles, sink);
    hipEventRecord(a);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((stream_kernel<ROWS, NS, MODE>), dim3(blocks), dim3(256), 0, 0, d, ld, rows_total, tiles, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    double bytes = (double)blocks * tiles * ROWS * 128;
    printf("%-28s rows/stage %3d NS %d blocks %5d tiles %3d ws %6.1f MB : %8.1f us  %7.2f TB/s  (%.1f B/clk/CU @2.4GHz)\n", name, ROWS, NS, blocks, tiles,
           rows_total * (double)ld / 1e6, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3);
}

int main() {
    const long ld = 8192;                      // bytes per row (K = 4096 bf16): 64 tiles of 128 B
    unsigned* sink; hipMalloc(&sink, 4);
    for (int rows_total : {2048, 16384, 131072}) {       // 16 MB (L2-ish), 128 MB (MALL), 1 GB (HBM)
        char* d; hipMalloc(&d, rows_total * ld); hipMemset(d, 1, rows_total * ld);
        for (int blocks : {256, 512, 1024, 2048}) {
            run<128, 4, 0>("glds 128x128B NS4", d, ld, rows_total, 64, blocks, sink);
            run<256, 3, 0>("glds 256x128B NS3", d, ld, rows_total, 64, blocks, sink);
            run<128, 2, 1>("reg-staged 128x128B", d, ld, rows_total, 64, blocks, sink);
        }
        hipFree(d);
    }
    return 0;
}
