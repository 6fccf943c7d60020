// Feasibility probe: a chain of dependent small kernels, (a) plain stream order, (b) two alternating streams with
// in-kernel flag waits (agent-scope release / acquire), (c) one stream with hipExtAnyOrderLaunch + flags.
// build: hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FLAGS>
__global__ void __launch_bounds__(256) link_kernel(unsigned* ctr, int idx, unsigned expect, const float* src, float* dst, int per_block, unsigned* err, int spin_work) {
    // "prologue" that does not depend on the producer
    float pro = 0.f;
    for (int i = 0; i < spin_work; ++i) pro += __sinf((float)(i + threadIdx.x));
    // FLAGS == 5 (round 5): the fence-free protocol of gemm_pp's tail exchange -- write-through (sc0 sc1) stores, vmcnt(0), a RELAXED
    // agent-scope counter; the consumer polls the counter and reads the producer's data with sc1 loads: no buffer_wbl2 / buffer_inv.
    if (FLAGS == 5 && idx > 0) {
        if (threadIdx.x == 0) {
            long t0 = wall_clock64();
            while (__hip_atomic_load(&ctr[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 20000000L) { atomicAdd(err, 1u); break; }
            }
        }
        __syncthreads();
    }
    if (FLAGS == 1 && idx > 0) {
        if (threadIdx.x == 0) {
            long t0 = wall_clock64();
            while (__hip_atomic_load(&ctr[idx - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 20000000L) { atomicAdd(err, 1u); break; }      // 0.2 s at 100 MHz: give up, never hang
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const int nb = gridDim.x;
    const int sb = (blockIdx.x + 37) % nb;
    for (int i = threadIdx.x; i < per_block; i += 256) {
        float v = 0.f;
        if (idx > 0) {
            if (FLAGS == 5) { asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&src[(long)sb * per_block + i]) : "memory"); }
            else v = src[(long)sb * per_block + i];
        }
        const float o = v + 1.0f + pro * 0.f;
        if (FLAGS == 3 || FLAGS == 5) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(&dst[(long)blockIdx.x * per_block + i]), "v"(o) : "memory");
        else if (FLAGS == 4) asm volatile("global_store_dword %0, %1, off nt" :: "v"(&dst[(long)blockIdx.x * per_block + i]), "v"(o) : "memory");
        else dst[(long)blockIdx.x * per_block + i] = o;
    }
    if (FLAGS == 5) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&ctr[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (FLAGS == 1) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&ctr[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ static inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }

// aggregated protocol: one L2 write-back per XCD (the last block of the XCD to finish), one L2 invalidate per XCD
// (the first block of the XCD to see the producer complete); ctr layout per link: [0..7] blocks done per XCD,
// [8] blocks flushed, [16..23] XCD ready
__global__ void __launch_bounds__(256) link2_kernel(unsigned* ctr, int idx, unsigned expect, const float* src, float* dst, int per_block, unsigned* err, int spin_work) {
    float pro = 0.f;
    for (int i = 0; i < spin_work; ++i) pro += __sinf((float)(i + threadIdx.x));
    const unsigned xcc = xcc_id();
    if (xcc != (blockIdx.x & 7u) && threadIdx.x == 0) atomicAdd(err + 1, 1u);
    if (idx > 0) {
        if (threadIdx.x == 0) {
            unsigned* c = ctr + (idx - 1) * 32;
            if (__hip_atomic_load(&c[16 + xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                long t0 = wall_clock64();
                while (__hip_atomic_load(&c[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > 20000000L) { atomicAdd(err, 1u); break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(&c[16 + xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    const int nb = gridDim.x;
    const int sb = (blockIdx.x + 37) % nb;
    for (int i = threadIdx.x; i < per_block; i += 256) {
        float v = idx > 0 ? src[(long)sb * per_block + i] : 0.f;
        dst[(long)blockIdx.x * per_block + i] = v + 1.0f + pro * 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* c = ctr + idx * 32;
        const unsigned mine = (nb >> 3) + (xcc < (unsigned)(nb & 7) ? 1u : 0u);
        unsigned old = __hip_atomic_fetch_add(&c[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == mine) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&c[8], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 100, grid = argc > 2 ? atoi(argv[2]) : 256, per_block = argc > 3 ? atoi(argv[3]) : 4096;
    const int spin_work = argc > 4 ? atoi(argv[4]) : 0;
    unsigned* ctr; unsigned* err; float *b0, *b1;
    CK(hipMalloc(&ctr, L * 32 * sizeof(unsigned))); CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8));
    CK(hipMalloc(&b0, (size_t)grid * per_block * 4)); CK(hipMalloc(&b1, (size_t)grid * per_block * 4));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    std::vector<float> host((size_t)grid * per_block);
    for (int mode = 0; mode < 9; ++mode) {
        if (mode >= 1 && mode <= 4 && getenv("CHAIN_PLAIN_ONLY")) continue;
        float best = 1e9f; int bad = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctr, 0, L * 32 * sizeof(unsigned), s0));
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s0));
            if (mode == 1 || mode == 4 || mode == 7) { CK(hipEventRecord(ej, s0)); CK(hipStreamWaitEvent(s1, ej, 0)); }
            for (int i = 0; i < L; ++i) {
                const float* src = (i & 1) ? b0 : b1; float* dst = (i & 1) ? b1 : b0;
                if (mode == 0) hipLaunchKernelGGL(link_kernel<0>, dim3(grid), dim3(256), 0, s0, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 7) hipLaunchKernelGGL(link_kernel<5>, dim3(grid), dim3(256), 0, (i & 1) ? s1 : s0, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 8) hipExtLaunchKernelGGL(link_kernel<5>, dim3(grid), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 1) hipLaunchKernelGGL(link_kernel<1>, dim3(grid), dim3(256), 0, (i & 1) ? s1 : s0, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 5) hipLaunchKernelGGL(link_kernel<3>, dim3(grid), dim3(256), 0, s0, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 6) hipLaunchKernelGGL(link_kernel<4>, dim3(grid), dim3(256), 0, s0, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 3) hipExtLaunchKernelGGL(link2_kernel, dim3(grid), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else if (mode == 4) hipLaunchKernelGGL(link2_kernel, dim3(grid), dim3(256), 0, (i & 1) ? s1 : s0, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
                else hipExtLaunchKernelGGL(link_kernel<1>, dim3(grid), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, ctr, i, (unsigned)grid, src, dst, per_block, err, spin_work);
            }
            if (mode == 1 || mode == 4 || mode == 7) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
            CK(hipEventRecord(e1, s0));
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
            CK(hipMemcpy(host.data(), ((L - 1) & 1) ? b1 : b0, host.size() * 4, hipMemcpyDeviceToHost));
            for (size_t j = 0; j < host.size(); ++j) if (host[j] != (float)L) ++bad;
        }
        unsigned herr2[2]; CK(hipMemcpy(herr2, err, 8, hipMemcpyDeviceToHost)); unsigned herr = herr2[0];
        printf("mode %d (%s): %.3f ms for %d links = %.2f us/link, wrong values %d, spin timeouts %u, xcc mismatches %u\n", mode,
               mode == 0 ? "plain stream" : mode == 1 ? "2 streams + flags" : mode == 2 ? "any-order + flags" : mode == 3 ? "any-order + per-XCD flags" : mode == 4 ? "2 streams + per-XCD flags" : mode == 5 ? "plain stream, write-through stores (sc0 sc1)" : mode == 6 ? "plain stream, nt stores" : mode == 7 ? "2 streams + fence-free flags (sc0 sc1 stores, sc1 loads)" : "any-order + fence-free flags", best, L, best * 1000.f / L, bad, herr, herr2[1]);
    }
    return 0;
}
