// Video-Depth-Anything temporal modules (streaming, one frame per call) -- A17 in SURVEY.md section 8a.
//   TemporalTransformer3DModel.forward   reference motion_module/motion_module.py:102-134
//   TemporalTransformerBlock.forward     :164-196
//   TemporalAttention.forward            :242-321   (q = current frame, k/v = 31 cached + current, APE by window index)
//   FeedForward / GEGLU                  motion_module/attention.py:296-384
//   cache update                         vda2_s.py:177-187, 203-218
// Activations are NHWC = [sites, C] token matrices, so every Linear is the engine's MFMA GEMM; this file
// holds the small kernels around them: GroupNorm, the window gather (+positional encoding), the
// 32-key attention per spatial site, GEGLU, and the ring-buffer cache (the reference shift-copies the
// whole cache every frame; here the oldest slot is overwritten in place).
#include "vit_ops.h"

namespace d2s {

template <typename T> __device__ __forceinline__ T tcvt(float v);
template <> __device__ __forceinline__ float tcvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t tcvt<bf16_t>(float v) { return f2bf(v); }
__device__ __forceinline__ float tf(float v) { return v; }
__device__ __forceinline__ float tf(bf16_t v) { return bf2f(v); }

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// nn.GroupNorm(32 groups, eps) on an NHWC map [sites, C]: one block per group, two-pass statistics.
template <typename T>
__global__ void __launch_bounds__(256)
groupnorm_kernel(const T* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, T* __restrict__ out,
                 int sites, int C, int groups, float eps) {
    __shared__ float red[4];
    const int cpg = C / groups, c0 = blockIdx.x * cpg;
    const int n = sites * cpg;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += tf(x[(long)(i / cpg) * C + c0 + i % cpg]);
    const float mu = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { float d = tf(x[(long)(i / cpg) * C + c0 + i % cpg]) - mu; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)n + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        int c = c0 + i % cpg;
        long off = (long)(i / cpg) * C + c;
        out[off] = tcvt<T>((tf(x[off]) - mu) * rstd * g[c] + b[c]);
    }
}

// window rows: kvin[(s*Tw + j), :] = (j < Tw-1 ? cache[(head + j) % slots][s] : cur[s]) + pe[j]
template <typename T>
__global__ void __launch_bounds__(256)
gather_pe_kernel(const T* __restrict__ cache, const T* __restrict__ cur, const float* __restrict__ pe, T* __restrict__ kvin,
                 int sites, int C, int Tw, int slots, int head) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)sites * Tw * C;
    if (idx >= total) return;
    int c = (int)(idx % C);
    long r = idx / C;
    int j = (int)(r % Tw), s = (int)(r / Tw);
    float v = (j < Tw - 1) ? tf(cache[((long)((head + j) % slots) * sites + s) * C + c]) : tf(cur[(long)s * C + c]);
    kvin[idx] = tcvt<T>(v + pe[(long)j * C + c]);
}

// cache[slot][s][c] = cur[s][c] for slot in [slot0, slot0 + nslots)
template <typename T>
__global__ void __launch_bounds__(256)
cache_store_kernel(T* __restrict__ cache, const T* __restrict__ cur, int sites, int C, int slot0, int nslots) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)sites * C;
    if (idx >= per * nslots) return;
    cache[(long)slot0 * per + idx] = cur[idx % per];
}

// One wave per spatial site: 8 heads x Tw (<= 32) keys.  q [sites, C]; kv [sites*Tw, 2C] (k | v); out [sites, C].
template <typename T>
__global__ void __launch_bounds__(256)
temporal_attn_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ out, int sites, int C, int Tw, float scale) {
    __shared__ float prob[4][8 * 32];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s0 = blockIdx.x * 4 + wid;
    const bool live = s0 < sites;                       // (no early return: block-level barriers below)
    const int s = live ? s0 : sites - 1;
    const int dh = C >> 3;
    const T* qs = q + (long)s * C;
    const T* kvs = kv + (long)s * Tw * 2 * C;
    float* p = prob[wid];
    // scores: pair id -> (head, key)
    const int npair = 8 * Tw;
    float sc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pid = lane + 64 * i;
        sc[i] = -1e30f;
        if (pid < npair) {
            int h = pid / Tw, j = pid - h * Tw;
            const T* kr = kvs + (long)j * 2 * C + h * dh;
            const T* qr = qs + h * dh;
            float acc = 0.f;
            for (int d = 0; d < dh; ++d) acc += tf(qr[d]) * tf(kr[d]);
            sc[i] = acc * scale;
            p[pid] = sc[i];
        }
    }
    __syncthreads();
    // softmax per head over Tw keys: lanes 0..7 each own one head's row in LDS
    if (lane < 8) {
        float* row = p + lane * Tw;
        float mx = -1e30f;
        for (int j = 0; j < Tw; ++j) mx = fmaxf(mx, row[j]);
        float sum = 0.f;
        for (int j = 0; j < Tw; ++j) { float e = __expf(row[j] - mx); row[j] = e; sum += e; }
        float inv = 1.0f / sum;
        for (int j = 0; j < Tw; ++j) row[j] *= inv;
    }
    __syncthreads();
    // out[c] = sum_j prob[h(c)][j] * v[j][c]; lanes stride over channels (coalesced V reads)
    for (int c = lane; c < C; c += 64) {
        const float* row = p + (c / dh) * Tw;
        float acc = 0.f;
        for (int j = 0; j < Tw; ++j) acc += row[j] * tf(kvs[(long)j * 2 * C + C + c]);
        if (live) out[(long)s * C + c] = tcvt<T>(acc);
    }
}

// GEGLU: g[r, c] = u[r, c] * gelu_exact(u[r, 4C + c]),  u [rows, 8C] -> g [rows, 4C]
template <typename T>
__global__ void __launch_bounds__(256)
geglu_kernel(const T* __restrict__ u, T* __restrict__ g, long rows, int C4) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C4) return;
    long r = idx / C4;
    int c = (int)(idx % C4);
    float x = tf(u[r * 2 * C4 + c]), gate = tf(u[r * 2 * C4 + C4 + c]);
    g[idx] = tcvt<T>(x * (0.5f * gate * (1.0f + erff(gate * 0.70710678118654752f))));
}

template <typename T>
__global__ void __launch_bounds__(256)
cast_f32_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = tcvt<T>(in[idx]);
}

#define TDISPATCH(prec, KERNEL, GRID, ...)                                                                    \
    do {                                                                                                      \
        if ((prec) == D2S_PREC_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, GRID, dim3(256), 0, st, __VA_ARGS__);  \
        else hipLaunchKernelGGL(KERNEL<float>, GRID, dim3(256), 0, st, __VA_ARGS__);                           \
    } while (0)

int launch_groupnorm(int prec, const void* x, const float* g, const float* b, void* out, int sites, int C, int groups, float eps, hipStream_t st) {
    if (C % groups) { set_error("groupnorm: C must be a multiple of the group count"); return D2S_E_INVALID; }
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(groupnorm_kernel<bf16_t>, dim3(groups), dim3(256), 0, st, (const bf16_t*)x, g, b, (bf16_t*)out, sites, C, groups, eps);
    else hipLaunchKernelGGL(groupnorm_kernel<float>, dim3(groups), dim3(256), 0, st, (const float*)x, g, b, (float*)out, sites, C, groups, eps);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_gather_pe(int prec, const void* cache, const void* cur, const float* pe, void* kvin, int sites, int C, int Tw, int slots, int head, hipStream_t st) {
    long total = (long)sites * Tw * C;
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(gather_pe_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)cache, (const bf16_t*)cur, pe, (bf16_t*)kvin, sites, C, Tw, slots, head);
    else hipLaunchKernelGGL(gather_pe_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)cache, (const float*)cur, pe, (float*)kvin, sites, C, Tw, slots, head);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_cache_store(int prec, void* cache, const void* cur, int sites, int C, int slot0, int nslots, hipStream_t st) {
    long total = (long)sites * C * nslots;
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(cache_store_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (bf16_t*)cache, (const bf16_t*)cur, sites, C, slot0, nslots);
    else hipLaunchKernelGGL(cache_store_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (float*)cache, (const float*)cur, sites, C, slot0, nslots);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_temporal_attn(int prec, const void* q, const void* kv, void* out, int sites, int C, int Tw, hipStream_t st) {
    if (C % 8 || Tw < 1 || Tw > 32) { set_error("temporal_attn: C % 8 == 0 and 1 <= window <= 32 required"); return D2S_E_INVALID; }
    float scale = 1.0f / sqrtf((float)(C / 8));
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(temporal_attn_kernel<bf16_t>, dim3(cdiv(sites, 4)), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)out, sites, C, Tw, scale);
    else hipLaunchKernelGGL(temporal_attn_kernel<float>, dim3(cdiv(sites, 4)), dim3(256), 0, st, (const float*)q, (const float*)kv, (float*)out, sites, C, Tw, scale);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_geglu(int prec, const void* u, void* g, long rows, int C4, hipStream_t st) {
    long total = rows * C4;
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(geglu_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)u, (bf16_t*)g, rows, C4);
    else hipLaunchKernelGGL(geglu_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)u, (float*)g, rows, C4);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_cast_f32(int prec, const float* in, void* out, long n, hipStream_t st) {
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(cast_f32_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, in, (bf16_t*)out, n);
    else hipLaunchKernelGGL(cast_f32_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, in, (float*)out, n);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

}  // namespace d2s
