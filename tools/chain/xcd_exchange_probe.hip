// Feasibility probe: partial-sum exchange between the blocks of ONE XCD through that XCD's L2, without device-scope fences.
// S blocks that share blockIdx % 8 (= XCC id, verified at run time) each write a partial tile with plain stores, count
// themselves on a per-group counter with a NON-device-scope atomic (executed in the XCD's own L2), and the last arrival reads
// all S partials back with TCP-bypassing loads and reduces them in slice order.  Variants of the load / atomic scope are
// checked for correctness over many launches with changing data, and timed against (a) no exchange and (b) a second kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_exchange_probe xcd_exchange_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ static inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }

typedef float f4 __attribute__((ext_vector_type(4)));

template <int LD> __device__ __forceinline__ void ld4_issue(f4& v, const f4* p) {
    if (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    else if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=&v"(v) : "v"(p) : "memory");
    else if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
}

// MODE 0: write partials only.  MODE 1: exchange in kernel.   T4: float4s per partial tile, 256 threads
template <int MODE, int LD, int AT>
__global__ void __launch_bounds__(256) part_kernel(f4* ws, unsigned* ctr, f4* out, int S, int T4, float seed, unsigned* err, int epoch) {
    const unsigned xcc = xcc_id();
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (xcc != (unsigned)xcd && threadIdx.x == 0) atomicAdd(err + 1, 1u);
    const int grp_in_x = slot / S, mem = slot % S, ngx = (gridDim.x >> 3) / S;
    const int grp = xcd * ngx + grp_in_x;
    f4* mine = ws + ((long)grp * S + mem) * T4;
    for (int i = threadIdx.x; i < T4; i += 256) {
        const float b = seed + (float)(grp * 131 + mem * 7) + (float)(i & 1023);
        mine[i] = (f4){b, b + 1.f, b + 2.f, b + 3.f};
    }
    if (MODE == 0) return;
    __shared__ unsigned last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned old;
        if (AT == 0) old = __hip_atomic_fetch_add(&ctr[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else old = __hip_atomic_fetch_add(&ctr[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old == (unsigned)(epoch * S + S - 1));
    }
    __syncthreads();
    if (!last) return;
    const f4* base = ws + (long)grp * S * T4;
    for (int i = threadIdx.x; i < T4; i += 256) {
        f4 v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) if (m < S) ld4_issue<LD>(v[m], base + (long)m * T4 + i);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 8; ++m) if (m < S) s += v[m];
        out[(long)grp * T4 + i] = s;
    }
}

__global__ void __launch_bounds__(256) reduce_kernel(const f4* ws, f4* out, int S, int T4) {
    const int grp = blockIdx.x;
    const f4* base = ws + (long)grp * S * T4;
    for (int i = threadIdx.x; i < T4; i += 256) {
        f4 s = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < S; ++m) s += base[(long)m * T4 + i];
        out[(long)grp * T4 + i] = s;
    }
}

int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 4, T4 = argc > 2 ? atoi(argv[2]) : 2048, NB = argc > 3 ? atoi(argv[3]) : 256;
    const int groups = NB / S, iters = 200;
    f4 *ws, *out; unsigned *ctr, *err;
    CK(hipMalloc(&ws, (size_t)NB * T4 * 16)); CK(hipMalloc(&out, (size_t)groups * T4 * 16));
    CK(hipMalloc(&ctr, groups * 4)); CK(hipMalloc(&err, 16));
    CK(hipMemset(err, 0, 16));
    std::vector<f4> h((size_t)groups * T4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto check = [&](float seed, const char* name) {
        CK(hipMemcpy(h.data(), out, h.size() * 16, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int g = 0; g < groups; ++g)
            for (int i = 0; i < T4; i += 97) {
                float s = 0.f;
                for (int m = 0; m < S; ++m) s += seed + (float)(g * 131 + m * 7) + (float)(i & 1023);
                if (h[(size_t)g * T4 + i][0] != s) ++bad;
            }
        if (bad) printf("  %s: %ld WRONG values (seed %.0f)\n", name, bad, seed);
        return bad;
    };
#define RUN(NAME, LAUNCH, CHECKED)                                                                  \
    {                                                                                               \
        CK(hipMemset(ctr, 0, groups * 4));                                                          \
        long bad = 0;                                                                               \
        for (int it = 0; it < 20; ++it) { const float seed = (float)(it * 1000); const int epoch = it; LAUNCH; if (CHECKED) { CK(hipDeviceSynchronize()); bad += check(seed, NAME); } } \
        CK(hipDeviceSynchronize());                                                                 \
        CK(hipEventRecord(e0));                                                                     \
        for (int it = 20; it < 20 + iters; ++it) { const float seed = (float)(it * 1000); const int epoch = it; LAUNCH; } \
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());                                         \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                             \
        if (CHECKED) bad += check((float)((20 + iters - 1) * 1000), NAME);                          \
        printf("%-44s %7.2f us per step   %s\n", NAME, ms * 1000.f / iters, CHECKED ? (bad ? "WRONG" : "ok") : "");  \
    }
    printf("S=%d partial=%d KB blocks=%d groups=%d\n", S, T4 * 16 / 1024, NB, groups);
    RUN("write partials only", (part_kernel<0, 0, 0><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), false)
    RUN("two kernels (partials; reduce)", (part_kernel<0, 0, 0><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch), reduce_kernel<<<groups, 256>>>(ws, out, S, T4)), true)
    RUN("in-kernel, plain loads, wg atomic", (part_kernel<1, 0, 0><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), true)
    RUN("in-kernel, sc0 loads, wg atomic", (part_kernel<1, 1, 0><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), true)
    RUN("in-kernel, sc1 loads, wg atomic", (part_kernel<1, 2, 0><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), true)
    RUN("in-kernel, sc0 sc1 loads, wg atomic", (part_kernel<1, 3, 0><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), true)
    RUN("in-kernel, sc1 loads, agent atomic", (part_kernel<1, 2, 1><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), true)
    RUN("in-kernel, plain loads, agent atomic", (part_kernel<1, 0, 1><<<NB, 256>>>(ws, ctr, out, S, T4, seed, err, epoch)), true)
    unsigned he[4]; CK(hipMemcpy(he, err, 16, hipMemcpyDeviceToHost));
    printf("xcc != blockIdx %% 8 on %u blocks\n", he[1]);
    return 0;
}
