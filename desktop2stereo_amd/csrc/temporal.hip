// Video-Depth-Anything temporal modules (streaming, one frame per call) -- A17 in SURVEY.md section 8a.
//   TemporalTransformer3DModel.forward   reference motion_module/motion_module.py:102-134
//   TemporalTransformerBlock.forward     :164-196
//   TemporalAttention.forward            :242-321   (q = current frame, k/v = 31 cached + current, APE by window index)
//   FeedForward / GEGLU                  motion_module/attention.py:296-384
//   cache update                         vda2_s.py:177-187, 203-218
// Activations are NHWC = [sites, C] token matrices, so every Linear is the engine's MFMA GEMM; this file
// holds the small kernels around them: GroupNorm, the 32-key attention per (spatial site, head) reading the
// ring-buffer window of projected K'/V' rows, GEGLU, and the ring store (the reference shift-copies the whole
// cache of hidden states every frame and re-projects all 32 window positions; here the oldest slot of projected
// rows is overwritten in place and only the new frame is projected).
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

// The window caches hold the PROJECTED rows K' = W_k x, V' = W_v x of the past 31 frames (x = the normed hidden state,
// without the positional encoding): the reference projects (x_j + pe_j) for all 32 window positions every frame
// (motion_module.py:288-300), but the projection is linear, W (x + pe_j) = W x + W pe_j, and W pe_j is a constant
// 32 x 3C table per attention block.  So a frame projects only its own rows (one [S, C] x [C, 3C] GEMM instead of
// [32 S, C] x [C, 2C]), and the attention adds the table rows on the fly.

// ring[slot][s][0:2C] = cur[s][0:2C] for slot in [slot0, slot0 + nslots); cur rows are 3C wide (k | v | q)
template <typename T>
__global__ void __launch_bounds__(256)
cache_store_kernel(T* __restrict__ cache, const T* __restrict__ cur, int sites, int C, int slot0, int nslots) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int C2 = 2 * C;
    long per = (long)sites * C2;
    if (idx >= per * nslots) return;
    long r = idx % per;
    cache[(long)slot0 * per + idx] = cur[(r / C2) * 3 * C + r % C2];
}

template <int CH> __device__ __forceinline__ void load_ch(const float* p, float v[CH]) {
#pragma unroll
    for (int i = 0; i < CH; i += 4) { float4 t = *(const float4*)(p + i); v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w; }
}
template <int CH> __device__ __forceinline__ void load_ch(const bf16_t* p, float v[CH]) {
#pragma unroll
    for (int i = 0; i < CH; i += 4) {
        uint2 t = *(const uint2*)(p + i);
        v[i] = __uint_as_float(t.x << 16); v[i + 1] = __uint_as_float(t.x & 0xffff0000u);
        v[i + 2] = __uint_as_float(t.y << 16); v[i + 3] = __uint_as_float(t.y & 0xffff0000u);
    }
}

// LPU (2..16, power of two >= head_dim / CH) lanes per (site, head) unit, lane <-> one CH-channel chunk of the head:
// every K' / V' row segment of a head is read once, coalesced (head_dim * sizeof(T) contiguous bytes per key), the
// score partials are xor-reduced over the unit's lanes, the 32 scores / probabilities live in registers.
// cur [S, 3C] (k' | v' | q' of this frame), ring [slots][S][2C] (k' | v' of the past frames, oldest at `head`),
// ptab [32][3C] float = pe @ [W_k | W_v | W_q]^T.  Window position j < Tw-1 is ring slot (head + j) % slots, j = Tw-1 the
// current frame (reference motion_module.py:259-300: q from the last position, k / v from all).  out [S, C].
template <typename T, int CH>
__global__ void __launch_bounds__(256)
temporal_attn_ring_kernel(const T* __restrict__ cur, T* __restrict__ ring, const float* __restrict__ ptab, T* __restrict__ out,
                          int sites, int C, int Tw, int slots, int head, float scale, int lpu, int store_slot) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int unit = gid / lpu, li = gid % lpu;
    const bool live = unit < sites * 8;                            // (no early return: the shuffles below need every lane)
    const int s = live ? unit >> 3 : 0, h = unit & 7, dh = C >> 3, C2 = 2 * C, C3 = 3 * C;
    const bool act = live && li * CH < dh;                         // lanes beyond the head's channels only take part in shuffles
    const int c = h * dh + (act ? li * CH : 0);                    // this lane's first channel
    const T* cur_row = cur + (long)s * C3;
    float q[CH];
    {
        float pq[CH];
        load_ch<CH>(cur_row + C2 + c, q);
        load_ch<CH>(ptab + (long)(Tw - 1) * C3 + C2 + c, pq);
#pragma unroll
        for (int i = 0; i < CH; ++i) q[i] = act ? q[i] + pq[i] : 0.f;
    }
    // keys in groups of 8: all loads of a group are issued before its arithmetic (the kernel is latency-bound: a unit
    // touches 32 x 2 short row segments).  Positions >= Tw (first frame only: Tw = 1) re-read the last valid row and are
    // masked out of the softmax.
    float sc[32];
#pragma unroll
    for (int jb = 0; jb < 32; jb += 8) {
        float k[8][CH], pk[8][CH];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = min(jb + t, Tw - 1);
            const T* krow = j < Tw - 1 ? ring + ((long)((head + j) % slots) * sites + s) * C2 : cur_row;
            load_ch<CH>(krow + c, k[t]);
            load_ch<CH>(ptab + (long)j * C3 + c, pk[t]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) a += q[i] * (k[t][i] + pk[t][i]);
            for (int o = lpu >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o);
            sc[jb + t] = jb + t < Tw ? a * scale : -1e30f;
        }
    }
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 32; ++j) mx = fmaxf(mx, sc[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) { sc[j] = j < Tw ? __expf(sc[j] - mx) : 0.f; sum += sc[j]; }
    const float inv = 1.0f / sum;
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
#pragma unroll
    for (int jb = 0; jb < 32; jb += 8) {
        float v[8][CH], pv[8][CH];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = min(jb + t, Tw - 1);
            const T* vrow = (j < Tw - 1 ? ring + ((long)((head + j) % slots) * sites + s) * C2 : cur_row) + C;
            load_ch<CH>(vrow + c, v[t]);
            load_ch<CH>(ptab + (long)j * C3 + C + c, pv[t]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float p = sc[jb + t] * inv;                      // 0 for masked positions
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] += p * (v[t][i] + pv[t][i]);
        }
    }
    if (act) {
#pragma unroll
        for (int i = 0; i < CH; ++i) out[(long)s * C + c + i] = tcvt<T>(acc[i]);
        // round 5: the frame's k' | v' rows join the window HERE (cache_store_kernel's launch is gone for every frame but the first):
        // this lane alone reads and writes the segment [c, c + CH) of site s in any ring slot, and both of its passes over the oldest
        // slot (keys above, values just now) are behind it -- program order within one thread is all the ordering needed
        if (store_slot >= 0) {
            T* dst = ring + ((long)store_slot * sites + s) * C2 + c;
#pragma unroll
            for (int i = 0; i < CH; ++i) { dst[i] = cur_row[c + i]; dst[C + i] = cur_row[C + c + i]; }
        }
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

int launch_cache_store(int prec, void* cache, const void* cur, int sites, int C, int slot0, int nslots, hipStream_t st) {
    long total = (long)sites * 2 * C * nslots;
    if (prec == D2S_PREC_BF16) hipLaunchKernelGGL(cache_store_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (bf16_t*)cache, (const bf16_t*)cur, sites, C, slot0, nslots);
    else hipLaunchKernelGGL(cache_store_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (float*)cache, (const float*)cur, sites, C, slot0, nslots);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_temporal_attn(int prec, const void* cur, void* ring, const float* ptab, void* out, int sites, int C, int Tw, int slots,
                         int head, hipStream_t st, int store_slot) {
    if (C % 32 || Tw < 1 || Tw > 32) { set_error("temporal_attn: C % 32 == 0 and 1 <= window <= 32 required"); return D2S_E_INVALID; }
    const float scale = 1.0f / sqrtf((float)(C / 8));
    const bool ch8 = (C / 8) % 8 == 0;                 // head dim multiple of 8: 16-byte bf16 chunks
    int lpu = 1;                                       // lanes per (site, head) unit: power of two >= head_dim / chunk
    while (lpu * (ch8 ? 8 : 4) < C / 8) lpu <<= 1;
    if (lpu > 64) { set_error("temporal_attn: head dim too large"); return D2S_E_UNSUPPORTED; }
    const dim3 grid(cdiv((long)sites * 8 * lpu, 256)), block(256);
#define D2S_TATT(TT, CH) hipLaunchKernelGGL((temporal_attn_ring_kernel<TT, CH>), grid, block, 0, st, (const TT*)cur, (TT*)ring, ptab, (TT*)out, \
                                            sites, C, Tw, slots, head, scale, lpu, store_slot)
    if (prec == D2S_PREC_BF16) { if (ch8) D2S_TATT(bf16_t, 8); else D2S_TATT(bf16_t, 4); }
    else { if (ch8) D2S_TATT(float, 8); else D2S_TATT(float, 4); }
#undef D2S_TATT
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
