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
#include <type_traits>

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

// nn.GroupNorm(32 groups, eps) on an NHWC map [sites, C]: one block per group, two-pass statistics (mean, then the sum of squared
// deviations) over values that STAY IN REGISTERS.  Round 4's kernel took 16 us for 2-9 k elements: three passes of scalar bf16 loads in a
// run-time loop, every load waited for before the next was issued (rocprofv3, profiles/r5_04) -- 64 us of a 1.06 ms VDA frame.  Here a
// thread owns whole sites (the group's cpg channels of a site are contiguous: one 8 / 16-byte aligned vector load per chunk), all of a
// thread's loads are issued before the first is used (compile-time unroll), and the normalised values are written from the registers.
// CPG = channels per group (2, 4, 6, 8, 12, 16, 24, 32 for the model zoo: C / 32), RPT = sites per thread.
template <typename T, int CPG, int RPT>
__global__ void __launch_bounds__(1024)
groupnorm_reg_kernel(const T* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, T* __restrict__ out,
                     int sites, int C, float eps) {
    __shared__ float red[16];
    const int c0 = blockIdx.x * CPG, tid = threadIdx.x;
    float v[RPT][CPG];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int s_ = tid + r * 1024;
        const T* p = x + (long)(s_ < sites ? s_ : 0) * C + c0;
#pragma unroll
        for (int c = 0; c < CPG; c += 2) {                            // (CPG is even; 4-byte loads are aligned for any even c0)
            if constexpr (std::is_same<T, float>::value) { const float2 t = *(const float2*)(p + c); v[r][c] = t.x; v[r][c + 1] = t.y; }
            else { const uint32_t t = *(const uint32_t*)(p + c); v[r][c] = __uint_as_float(t << 16); v[r][c + 1] = __uint_as_float(t & 0xffff0000u); }
        }
    }
    auto bsum = [&](float a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = a;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i];
        return t;
    };
    const float n = (float)sites * (float)CPG;
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (tid + r * 1024 < sites) {
#pragma unroll
            for (int c = 0; c < CPG; ++c) s1 += v[r][c];
        }
    const float mu = bsum(s1) / n;
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (tid + r * 1024 < sites) {
#pragma unroll
            for (int c = 0; c < CPG; ++c) { const float d = v[r][c] - mu; s2 += d * d; }
        }
    const float rstd = 1.0f / sqrtf(bsum(s2) / n + eps);
    float gg[CPG], bb[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) { gg[c] = g[c0 + c]; bb[c] = b[c0 + c]; }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int s_ = tid + r * 1024;
        if (s_ < sites) {
            T* q = out + (long)s_ * C + c0;
#pragma unroll
            for (int c = 0; c < CPG; ++c) q[c] = tcvt<T>((v[r][c] - mu) * rstd * gg[c] + bb[c]);
        }
    }
}

// (general shapes: the round-1 kernel, one block per group, three strided passes)
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
// a row segment of CH values kept RAW (as loaded) until it is used: CH * sizeof(T) / 4 registers
template <typename T, int CH> struct RawSeg {
    static constexpr int W = CH * (int)sizeof(T) / 4;
    uint32_t w[W];
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (W == 2) { const uint2 t = *(const uint2*)p; w[0] = t.x; w[1] = t.y; }
        else {
#pragma unroll
            for (int i = 0; i < W; i += 4) { const uint4 t = *(const uint4*)((const uint32_t*)p + i); w[i] = t.x; w[i + 1] = t.y; w[i + 2] = t.z; w[i + 3] = t.w; }
        }
    }
    __device__ __forceinline__ float get(int i) const {
        if constexpr (std::is_same<T, float>::value) return __uint_as_float(w[i]);
        else return (i & 1) ? __uint_as_float(w[i >> 1] & 0xffff0000u) : __uint_as_float(w[i >> 1] << 16);
    }
};

// One (site, head) UNIT = LPU x KG lanes: LPU (2..16, power of two >= head_dim / CH) lanes own one CH-channel chunk of the head each,
// KG = 4 key groups own 8 of the 32 window positions each.  Every K' / V' row segment of a head is read once, coalesced (head_dim *
// sizeof(T) contiguous bytes per key); score partials are xor-reduced over the unit's channel lanes, softmax statistics and the output
// over its key groups (lanes LPU and 2 LPU apart).
// cur [S, 3C] (k' | v' | q' of this frame), ring [slots][S][2C] (k' | v' of the past frames, oldest at `head`),
// ptab [32][3C] float = pe @ [W_k | W_v | W_q]^T.  Window position j < Tw-1 is ring slot (head + j) % slots, j = Tw-1 the
// current frame (reference motion_module.py:259-300: q from the last position, k / v from all).  out [S, C].
// Round 5.  Round 4's kernel gave a lane all 32 keys: ~3 500 instructions per lane (32 x 2 row addresses with an integer modulo each,
// 32 x 3 shuffles, fully unrolled), ONE 4-wave block per CU (194 blocks for a ViT-B stream) -- 19-28 us per launch, 8 launches per
// frame, and the time did not move when its ring requests were batched 8 / 16 / 32 at a time (variant builds, gpurun_out/r5j): it was
// bound by its own instruction stream at one wave per SIMD, not by the rings' latency.  Splitting the keys over lanes gives 4 x the
// threads with ~1/7 of the instructions each: all 16 ring segments of a lane are requested up front, 3+ waves per SIMD cover them.
constexpr int TA_KG = 4, TA_KPG = 8;           // key groups per unit, keys per group (TA_KG * TA_KPG = the 32-frame window)
template <typename T, int CH>
__global__ void __launch_bounds__(256)
temporal_attn_ring_kernel(const T* __restrict__ cur, T* __restrict__ ring, const float* __restrict__ ptab, T* __restrict__ out,
                          int sites, int C, int Tw, int slots, int head, float scale, int lpu, int store_slot) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int upl = lpu * TA_KG;                                   // lanes per unit (<= 64: a unit never straddles a wave)
    const int unit = gid / upl, sub = gid % upl, kg = sub / lpu, li = sub % lpu;
    const bool live = unit < sites * 8;                            // (no early return: the shuffles below need every lane)
    const int s = live ? unit >> 3 : 0, h = unit & 7, dh = C >> 3, C2 = 2 * C, C3 = 3 * C;
    const bool act = live && li * CH < dh;                         // lanes beyond the head's channels only take part in shuffles
    const int c = h * dh + (act ? li * CH : 0);                    // this lane's first channel
    const T* cur_row = cur + (long)s * C3;
    // this lane's 8 window positions and their rows (32-bit element offsets: a ring is < 2^31 elements, checked by the launcher);
    // positions >= Tw (first frame only: Tw = 1) re-read the last valid row and are masked out of the softmax
    const int slot_stride = sites * C2, base_s = s * C2 + c;
    const T* krow[TA_KPG];
    int jpos[TA_KPG];
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) {
        const int j = kg * TA_KPG + t, jj = min(j, Tw - 1);
        int slot = head + jj; if (slot >= slots) slot -= slots;    // (head < slots, jj < 32 <= slots + 1: one conditional subtract, no modulo)
        if (slot >= slots) slot -= slots;
        krow[t] = (act && jj < Tw - 1) ? ring + (slot * slot_stride + base_s) : cur_row + c;   // (padding / dead lanes never touch the ring: see the store below)
        jpos[t] = jj;
    }
    RawSeg<T, CH> rk[TA_KPG], rv[TA_KPG];
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) rk[t].load(krow[t]);
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) rv[t].load(krow[t] + C);
    float q[CH];
    {
        float pq[CH];
        load_ch<CH>(cur_row + C2 + c, q);
        load_ch<CH>(ptab + (long)(Tw - 1) * C3 + C2 + c, pq);
#pragma unroll
        for (int i = 0; i < CH; ++i) q[i] = act ? q[i] + pq[i] : 0.f;
    }
    float sc[TA_KPG];
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) {
        float pk[CH];
        load_ch<CH>(ptab + (long)jpos[t] * C3 + c, pk);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) a += q[i] * (rk[t].get(i) + pk[i]);
        for (int o = lpu >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o);
        sc[t] = kg * TA_KPG + t < Tw ? a * scale : -1e30f;
    }
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) mx = fmaxf(mx, sc[t]);
    mx = fmaxf(mx, __shfl_xor(mx, lpu)); mx = fmaxf(mx, __shfl_xor(mx, 2 * lpu));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) { sc[t] = kg * TA_KPG + t < Tw ? __expf(sc[t] - mx) : 0.f; sum += sc[t]; }
    sum += __shfl_xor(sum, lpu); sum += __shfl_xor(sum, 2 * lpu);
    const float inv = 1.0f / sum;
    float acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < TA_KPG; ++t) {
        float pv[CH];
        load_ch<CH>(ptab + (long)jpos[t] * C3 + C + c, pv);
        const float p = sc[t] * inv;                               // 0 for masked positions
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] += p * (rv[t].get(i) + pv[i]);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) { acc[i] += __shfl_xor(acc[i], lpu); acc[i] += __shfl_xor(acc[i], 2 * lpu); }
    if (act && kg == 0) {
#pragma unroll
        for (int i = 0; i < CH; ++i) out[(long)s * C + c + i] = tcvt<T>(acc[i]);
    }
    // round 5: the frame's k' | v' rows join the window HERE (cache_store_kernel's launch is gone for every frame but the first).  The
    // oldest slot's segment [c, c + CH) of site s is READ by the key group that owns window position 0 -- kg 0's ACTIVE lane -- and by nobody
    // else (the padding lanes of a head, li * CH >= head_dim, and the lanes of dead units read the current frame's row instead of the ring:
    // their q is 0 and they store nothing, but they used to load this segment too -- an unordered read beside the store), so
    // that lane overwrites it once its own two loads of it are behind it: program order within one thread is all the ordering needed.
    if (act && kg == 0 && store_slot >= 0) {
        T* dst = ring + ((long)store_slot * slot_stride + base_s);
#pragma unroll
        for (int i = 0; i < CH; ++i) { dst[i] = cur_row[c + i]; dst[C + i] = cur_row[C + c + i]; }
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

template <typename T>
static bool launch_groupnorm_reg(const void* x, const float* g, const float* b, void* out, int sites, int C, int groups, float eps, hipStream_t st) {
    const int cpg = C / groups, rpt = (sites + 1023) / 1024;
#define D2S_GN(CPG_, RPT_) { hipLaunchKernelGGL((groupnorm_reg_kernel<T, CPG_, RPT_>), dim3(groups), dim3(1024), 0, st, (const T*)x, g, b, (T*)out, sites, C, eps); return true; }
#define D2S_GN_R(CPG_) { if (rpt == 1) D2S_GN(CPG_, 1) if (rpt == 2) D2S_GN(CPG_, 2) if (rpt <= 4 && (CPG_) * 4 <= 96) D2S_GN(CPG_, 4) return false; }
    switch (cpg) {          // C / 32 of the model zoo: fusion 64 / 128 / 256, neck 192 / 384 / 512 / 768 / 1024
        case 2: D2S_GN_R(2) case 4: D2S_GN_R(4) case 6: D2S_GN_R(6) case 8: D2S_GN_R(8) case 12: D2S_GN_R(12)
        case 16: D2S_GN_R(16) case 24: D2S_GN_R(24) case 32: D2S_GN_R(32)
        default: return false;
    }
#undef D2S_GN_R
#undef D2S_GN
}

int launch_groupnorm(int prec, const void* x, const float* g, const float* b, void* out, int sites, int C, int groups, float eps, hipStream_t st) {
    if (C % groups) { set_error("groupnorm: C must be a multiple of the group count"); return D2S_E_INVALID; }
    static EnvInt old{"D2S_GN_OLD", 0};                   // A/B aid: round 4's kernel
    const bool reg_ok = !old.get() && ((C / groups) & 1) == 0;
    if (prec == D2S_PREC_BF16) {
        if (!(reg_ok && launch_groupnorm_reg<bf16_t>(x, g, b, out, sites, C, groups, eps, st)))
            hipLaunchKernelGGL(groupnorm_kernel<bf16_t>, dim3(groups), dim3(256), 0, st, (const bf16_t*)x, g, b, (bf16_t*)out, sites, C, groups, eps);
    } else {
        if (!(reg_ok && launch_groupnorm_reg<float>(x, g, b, out, sites, C, groups, eps, st)))
            hipLaunchKernelGGL(groupnorm_kernel<float>, dim3(groups), dim3(256), 0, st, (const float*)x, g, b, (float*)out, sites, C, groups, eps);
    }
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
    if (lpu * TA_KG > 64) { set_error("temporal_attn: head dim too large"); return D2S_E_UNSUPPORTED; }
    if ((long)slots * sites * 2 * C >= (1L << 31) || slots < 31) { set_error("temporal_attn: ring too large for 32-bit offsets (or fewer than 31 slots)"); return D2S_E_UNSUPPORTED; }
    const dim3 grid(cdiv((long)sites * 8 * lpu * TA_KG, 256)), block(256);
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
