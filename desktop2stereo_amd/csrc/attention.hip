// Fused multi-head self-attention (non-causal, head_dim 64) on MFMA, flash-style online softmax.
// (HF Dinov2SelfAttention.forward: softmax(q k^T * hd^-0.5) v; reference call site depth.py:1778)
//
// Inputs:  qkv  [B*N, 3D] T   (q | k | v), written by the QKV GEMM;
//          vt   [B, heads, 64, Npad] T  = V transposed (written by the same GEMM's epilogue), zero beyond N.
// Output:  out  [B*N, D] T.
//
// One block = NW (4 | 8) waves, one (batch, head, q-tile); each wave owns QF fragments of 16 query rows.
// Per 64-key tile:
//   S^T[key, q] = mfma(K rows, Q rows)         lane: 4 keys x 1 query  -> softmax stats are per lane column
//   P^T feeds the second MFMA straight from the S registers (no LDS round trip): the K rows of a
//   fragment are loaded in a permuted order so that a lane group's values are 8 (bf16) / 4 (f32)
//   CONSECUTIVE keys, i.e. exactly one 16-byte chunk of a V^T row;
//   O^T[d, q] += mfma(V^T rows, P^T)           lane: 4 consecutive d for 1 query -> vector stores.
// K / V^T tiles: global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring; the XOR swizzle is
// applied on the source address (LDS-DMA writes lane-linear), counted vmcnt + one raw barrier per tile.
// Softmax in fp32 (exp2 with the scale folded in).
#include "vit_ops.h"

namespace d2s {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers (HIP's u32x4 struct arrays went to scratch)

template <typename T> struct AT;
template <> struct AT<bf16_t> {
    static constexpr int CE = 8, CPR = 8, NKS = 2;
    __device__ static int swzK(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }
    __device__ static int swzV(int r) { return (r >> 1) & 7; }
    // tile key held by MFMA A-row i of S fragment j
    __device__ static int key_of(int j, int i) { return (j >> 1) * 32 + (i >> 2) * 8 + (i & 3) + 4 * (j & 1); }
};
template <> struct AT<float> {
    static constexpr int CE = 4, CPR = 16, NKS = 4;
    __device__ static int swzK(int r) { return r & 15; }
    __device__ static int swzV(int r) { return r & 15; }
    __device__ static int key_of(int j, int i) { return j * 16 + i; }
};

__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b, float) {
    const float* af = (const float*)&a;
    const float* bf = (const float*)&b;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bf[t], acc, 0, 0, 0);
}

// two floats -> packed bf16 pair, round-to-nearest-even: v_cvt_pk_bf16_f32 (gfx950) instead of ~5 VALU ops per value
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// max over lanes l and l ^ 16 (resp. l ^ 32) with gfx950's row swaps instead of ds_bpermute_b32: no LDS round trip and no
// lgkmcnt wait in the online softmax of every key tile.  v_permlane16_swap exchanges the odd 16-lane rows of its first
// operand with the even rows of the second; with both = v the pair is {r0 r0 r2 r2}, {r1 r1 r3 r3}.
__device__ __forceinline__ float xmax16(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xmax32(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// pack the P values a lane group contributes to one V^T chunk
__device__ __forceinline__ u32x4 pack_p(const f32x4& lo, const f32x4& hi, bf16_t) {
    u32x4 r;
    r.x = pk_bf16(lo[0], lo[1]);
    r.y = pk_bf16(lo[2], lo[3]);
    r.z = pk_bf16(hi[0], hi[1]);
    r.w = pk_bf16(hi[2], hi[3]);
    return r;
}
__device__ __forceinline__ u32x4 pack_p(const f32x4& lo, const f32x4&, float) {
    u32x4 r;
    r.x = __float_as_uint(lo[0]); r.y = __float_as_uint(lo[1]); r.z = __float_as_uint(lo[2]); r.w = __float_as_uint(lo[3]);
    return r;
}

__device__ __forceinline__ void store_o4(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store_o4(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = pk_bf16(v[0], v[1]);
    t.y = pk_bf16(v[2], v[3]);
    *(uint2*)p = t;
}

__device__ __forceinline__ void store_o4(fp8_t* p, const float v[4]) {     // values already scaled by 1 / (activation scale)
    float a = fminf(fmaxf(v[0], -FP8_MAX), FP8_MAX), b = fminf(fmaxf(v[1], -FP8_MAX), FP8_MAX);
    float c = fminf(fmaxf(v[2], -FP8_MAX), FP8_MAX), d = fminf(fmaxf(v[3], -FP8_MAX), FP8_MAX);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    *(uint32_t*)p = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

__device__ u32x4 d2s_attn_zero_page[4];

// OT: output element type (T, or e4m3 when the output projection runs on fp8 operands: out = sat(result * oscale))
// KS > 1 (batch 1: 156 blocks of 4 waves leave the chip empty and each walks 13 key tiles in sequence): KS wave groups
// per block share the q tile and split the KEY range; each group runs the whole pipeline over its keys with its own LDS
// ring, and the groups' (max, sum, O) are merged through LDS at the end (flash-decoding style).
template <typename T, int QF, int NW, int NS = 3, typename OT = T, int KS = 1>
__global__ void __launch_bounds__(64 * NW * KS)
attention_kernel(const T* __restrict__ qkv, const T* __restrict__ vt, OT* __restrict__ out,
                 int N, int Npad, int heads, float scale_log2e, float oscale) {
    using A = AT<T>;
    constexpr int CE = A::CE, CPR = A::CPR, NKS = A::NKS;
    constexpr int TILE_CHUNKS = 64 * CPR;              // chunks in one 64-row tile
    constexpr int BQ = NW * QF * 16;
    constexpr int PD = NS - 1;                         // NS LDS ring stages, prefetch distance PD
    __shared__ __attribute__((aligned(16))) u32x4 lds_all[KS * NS * 2 * TILE_CHUNKS];   // [group][stage][K | V^T]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_all = tid >> 6, grp = KS == 1 ? 0 : wid_all / NW, wid = KS == 1 ? wid_all : wid_all % NW;
    u32x4* const lds = lds_all + grp * (NS * 2 * TILE_CHUNKS);
    const int fr = lane & 15, fg = lane >> 4;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int D = heads * 64;
    const long row3 = 3L * D;
    const T* qbase = qkv + (long)b * N * row3 + h * 64;
    const T* kbase = qbase + D;
    const T* vbase = vt + ((long)b * heads + h) * 64 * Npad;

    // ---- Q fragments (B operand of S^T): Q[q][chunk ks*4+fg]
    u32x4 qf[QF][NKS];
    int qrow[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        qrow[f] = qt * BQ + (wid * QF + f) * 16 + fr;
        bool ok = qrow[f] < N;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            qf[f][ks] = ok ? *(const u32x4*)(qbase + (long)qrow[f] * row3 + (ks * 4 + fg) * CE) : (u32x4){0u, 0u, 0u, 0u};
    }

    f32x4 o[QF][4];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_run[f] = -1e30f; l_run[f] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[f][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // K / V^T tiles travel global -> LDS by LDS-DMA (1 KiB per wave-instruction = 64/CPR rows) into an NS-stage
    // ring, PD tiles ahead, counted vmcnt + one raw barrier per tile (same scheme as gemm_glds_kernel).  The row
    // XOR swizzles move to the source address: the lane owning LDS slot (row r, phys chunk p) fetches chunk p ^ swz(r).
    constexpr int RPI = 64 / CPR;                       // rows per wave-instruction
    constexpr int IPW = 64 / (RPI * NW);                // instructions per wave per operand tile
    const int wu = __builtin_amdgcn_readfirstlane(wid);
    const T* zero = (const T*)d2s_attn_zero_page;
#define ATT_GLDS(SRC, DST) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
#define ATT_ISSUE(TT)                                                                                  \
    {                                                                                                  \
        u32x4* st_ = lds + ((TT) % NS) * 2 * TILE_CHUNKS;                                              \
        _Pragma("unroll") for (int i = 0; i < IPW; ++i) {                                              \
            const int r_ = (i * NW + wu) * RPI + lane / CPR, p_ = lane % CPR;                          \
            const int key_ = ((TT) + t0) * 64 + r_;                                                    \
            const T* ks_ = key_ < N ? kbase + (long)key_ * row3 + (p_ ^ A::swzK(r_)) * CE : zero;      \
            ATT_GLDS(ks_, st_ + (i * NW + wu) * 64);                                                   \
            ATT_GLDS(vbase + (long)r_ * Npad + ((TT) + t0) * 64 + (p_ ^ A::swzV(r_)) * CE, st_ + TILE_CHUNKS + (i * NW + wu) * 64); \
        }                                                                                              \
    }
    constexpr int LPTA = 2 * IPW;                       // LDS-DMA instructions per thread per tile

    const int ntiles_all = (N + 63) / 64;
    const int t0 = KS == 1 ? 0 : (ntiles_all * grp) / KS;                      // this group's key tiles [t0, t0 + ntiles)
    const int ntiles = KS == 1 ? ntiles_all : (ntiles_all * (grp + 1)) / KS - t0;
    const int nit = KS == 1 ? ntiles_all : (ntiles_all + KS - 1) / KS;          // barrier count, the same for every group
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < ntiles) ATT_ISSUE(t)
    for (int t = 0; t < nit; ++t) {
        if (t + PD - 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * LPTA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + PD < ntiles) ATT_ISSUE(t + PD)
        if (KS > 1 && t >= ntiles) continue;
        const u32x4* Kl = lds + (t % NS) * 2 * TILE_CHUNKS;
        const u32x4* Vl = Kl + TILE_CHUNKS;
        // ---- S^T = K Q^T
        f32x4 s[QF][4];
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[f][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int kr = A::key_of(j, fr);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                u32x4 kf = Kl[kr * CPR + ((ks * 4 + fg) ^ A::swzK(kr))];
#pragma unroll
                for (int f = 0; f < QF; ++f) mma16(s[f][j], kf, qf[f][ks], T());
            }
        }
        // ---- mask the ragged last tile: key of register r in fragment j = key_of(j, fg*4 + r)
        if (t + t0 == ntiles_all - 1 && (N & 63)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((t + t0) * 64 + A::key_of(j, fg * 4 + r) >= N) {
#pragma unroll
                        for (int f = 0; f < QF; ++f) s[f][j][r] = -1e30f;
                    }
        }
        // ---- online softmax (per query column = per lane & 15)
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mx = -1e30f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[f][j][r]);
            mx = xmax16(mx);                          // across the 4 lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48)
            mx = xmax32(mx);
            // p = 2^((s - m) * c) as one FMA + one raw v_exp_f32 per value (arguments are <= 0: no range handling needed)
            const float m_new = fmaxf(m_run[f], mx);
            const float mc = m_new * scale_log2e;
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] * scale_log2e - mc);
            const bool grew = m_new > m_run[f];
            m_run[f] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // fp32 parity class: subtract first (exact for nearby values), then scale
                    float arg = sizeof(T) == 4 ? (s[f][j][r] - m_new) * scale_log2e : fmaf(s[f][j][r], scale_log2e, -mc);
                    float p = __builtin_amdgcn_exp2f(arg);
                    s[f][j][r] = p; ps += p;
                }
            l_run[f] = l_run[f] * alpha + ps;          // per-lane partial; groups are summed at the end
            if (__any(grew)) {                          // the running maximum settles after the first tiles: skip the O rescale then
#pragma unroll
                for (int d = 0; d < 4; ++d) { o[f][d][0] *= alpha; o[f][d][1] *= alpha; o[f][d][2] *= alpha; o[f][d][3] *= alpha; }
            }
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int pc = 0; pc < NKS; ++pc) {
            u32x4 pf[QF];
#pragma unroll
            for (int f = 0; f < QF; ++f)
            {
                if constexpr (NKS == 2) pf[f] = pack_p(s[f][2 * pc], s[f][2 * pc + 1], T());
                else pf[f] = pack_p(s[f][pc], s[f][pc], T());
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int vr = d * 16 + fr;
                u32x4 vf = Vl[vr * CPR + ((pc * 4 + fg) ^ A::swzV(vr))];
#pragma unroll
                for (int f = 0; f < QF; ++f) mma16(o[f][d], vf, pf[f], T());
            }
        }
    }
#undef ATT_ISSUE
#undef ATT_GLDS
    if constexpr (KS > 1) {
        // merge the key groups: (m, l, O) of groups 1.. go through LDS (the rings are free after the barrier); group 0
        // rescales to the common maximum and finishes.  m is uniform per query column, l is a per-lane partial, both
        // combine linearly after the exp2 rescale.
        constexpr int PER = QF * 18;                            // floats per lane: QF x (16 O + m + l)
        static_assert((KS - 1) * NW * 64 * PER * 4 <= KS * NS * 2 * TILE_CHUNKS * 16, "merge scratch must fit in the rings");
        float* red = (float*)lds_all;
        __syncthreads();
        if (grp > 0) {
            float* p = red + ((long)((grp - 1) * NW + wid) * PER) * 64 + lane;
#pragma unroll
            for (int f = 0; f < QF; ++f) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[((f * 18) + d * 4 + r) * 64] = o[f][d][r];
                p[(f * 18 + 16) * 64] = m_run[f];
                p[(f * 18 + 17) * 64] = l_run[f];
            }
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 0; g < KS - 1; ++g) {
            const float* p = red + ((long)(g * NW + wid) * PER) * 64 + lane;
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const float m1 = p[(f * 18 + 16) * 64], l1 = p[(f * 18 + 17) * 64];
                const float m = fmaxf(m_run[f], m1);
                const float a0 = __builtin_amdgcn_exp2f((m_run[f] - m) * scale_log2e), a1 = __builtin_amdgcn_exp2f((m1 - m) * scale_log2e);
                m_run[f] = m;
                l_run[f] = l_run[f] * a0 + l1 * a1;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[f][d][r] = o[f][d][r] * a0 + p[((f * 18) + d * 4 + r) * 64] * a1;
            }
        }
    }
    // ---- normalise and store: lane holds d = dfrag*16 + fg*4 + r for query fr
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        float inv = oscale / l;
        if (qrow[f] < N) {
            OT* orow = out + ((long)b * N + qrow[f]) * D + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float v[4] = {o[f][d][0] * inv, o[f][d][1] * inv, o[f][d][2] * inv, o[f][d][3] * inv};
                store_o4(orow + d * 16 + fg * 4, v);
            }
        }
    }
}

int launch_attention(int prec, const void* qkv, const void* vt, void* out, int B, int N, int Npad, int heads, hipStream_t st,
                     float fp8_qscale) {
    const float scale_log2e = 0.125f * 1.4426950408889634f;          // 64^-0.5 * log2(e)
    if (fp8_qscale > 0.f && prec != D2S_PREC_BF16) { set_error("attention: e4m3 output needs bf16 inputs"); return D2S_E_UNSUPPORTED; }
    // q rows per block: 128 as 8 waves x 1 fragment once that fills the chip (batch >= ~8), else 64 as 4 waves.
    // Swept at batch 1 / 16: 4 waves x 2 fragments 80 us, ring depth 2 / 4 within 3 %, 8 x 2 (256 rows) 68 us,
    // 8 x 1 67 us (444 TFLOP/s); 32-row blocks slower (K/V tile loads not amortised).  D2S_ATTN_BQ forces 128 / 64 / 32.
    static const int force = getenv("D2S_ATTN_BQ") ? atoi(getenv("D2S_ATTN_BQ")) : 0;
    long hb = (long)heads * B;
    int bq = force ? force : (cdiv(N, 128) * hb >= 512 ? 128 : 64);
#define D2S_ATT(TT, QF_, NW_) hipLaunchKernelGGL((attention_kernel<TT, QF_, NW_>), dim3(cdiv(N, NW_ * QF_ * 16), heads, B), dim3(64 * NW_), 0, st, \
        (const TT*)qkv, (const TT*)vt, (TT*)out, N, Npad, heads, scale_log2e, 1.0f)
#define D2S_ATT8(QF_, NW_) hipLaunchKernelGGL((attention_kernel<bf16_t, QF_, NW_, 3, fp8_t>), dim3(cdiv(N, NW_ * QF_ * 16), heads, B), dim3(64 * NW_), 0, st, \
        (const bf16_t*)qkv, (const bf16_t*)vt, (fp8_t*)out, N, Npad, heads, scale_log2e, fp8_qscale)
    // too few (q-tile, head) blocks to fill 256 CUs (batch 1: 156): split the keys over 2 / 4 wave groups per block
    // (batch 1, N = 778: 18.9 us -> 13.8 us with 2 groups, 13.2 us with 4)
    static const int force_ks = getenv("D2S_ATTN_KS") ? atoi(getenv("D2S_ATTN_KS")) : 0;
    const int ks = force_ks ? force_ks : (bq == 64 && (long)cdiv(N, 64) * hb < 256 ? (N >= 512 ? 4 : (N >= 256 ? 2 : 1)) : 1);
    if (fp8_qscale > 0.f) {
        if (bq == 128) D2S_ATT8(1, 8);
        else if (ks >= 2)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 2, fp8_t, 4>), dim3(cdiv(N, 64), heads, B), dim3(1024), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (fp8_t*)out, N, Npad, heads, scale_log2e, fp8_qscale);
        else D2S_ATT8(1, 4);
    } else if (prec == D2S_PREC_BF16) {
        if (bq == 64 && ks == 2)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 3, bf16_t, 2>), dim3(cdiv(N, 64), heads, B), dim3(512), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, N, Npad, heads, scale_log2e, 1.0f);
        else if (bq == 64 && ks == 4)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 2, bf16_t, 4>), dim3(cdiv(N, 64), heads, B), dim3(1024), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, N, Npad, heads, scale_log2e, 1.0f);
        else if (bq == 128) D2S_ATT(bf16_t, 1, 8);
        else if (bq == 64) D2S_ATT(bf16_t, 1, 4);
        else D2S_ATT(bf16_t, 1, 2);
    } else {
        if (bq >= 64) D2S_ATT(float, 1, 4);
        else D2S_ATT(float, 1, 2);
    }
#undef D2S_ATT
#undef D2S_ATT8
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

}  // namespace d2s
