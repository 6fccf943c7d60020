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
#include <type_traits>

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

// bf16x3 (split precision, common.h): rows of 64 elements = 8 units of [8 hi | 8 lo] = 16 chunks; a K = 32 step takes, per lane
// group, the hi and the lo chunk of ONE unit; keys are permuted like the bf16 kernel's so that a lane group's P values of two S
// fragments are the 8 consecutive keys of one V^T unit.
template <> struct AT<bx3_t> {
    static constexpr int CE = 4, CPR = 16, NKS = 2;
    __device__ static int swzK(int r) { return r & 15; }
    __device__ static int swzV(int r) { return r & 15; }
    __device__ static int key_of(int j, int i) { return (j >> 1) * 32 + (i >> 2) * 8 + (i & 3) + 4 * (j & 1); }
};
#define ATT_BX3_UNIT(fg) ((0x2130 >> (4 * (fg))) & 3)               /* lane group -> unit of a K = 32 step (gemm_epi.h BX3_UNIT) */
// x = hi + lo per operand: three bf16 MFMAs, small terms first
__device__ __forceinline__ void mma16x3(f32x4& acc, const u32x4& a_hi, const u32x4& a_lo, const u32x4& b_hi, const u32x4& b_lo) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a_lo, *(const bf16x8*)&b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a_hi, *(const bf16x8*)&b_lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a_hi, *(const bf16x8*)&b_hi, acc, 0, 0, 0);
}
// 8 fp32 P values -> the hi chunk and the lo chunk of their unit
__device__ __forceinline__ void pack_p3(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo);

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

__device__ __forceinline__ void pack_p3(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    hi.x = pk_bf16(a[0], a[1]); hi.y = pk_bf16(a[2], a[3]); hi.z = pk_bf16(b[0], b[1]); hi.w = pk_bf16(b[2], b[3]);
    lo.x = pk_bf16(a[0] - __uint_as_float(hi.x << 16), a[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pk_bf16(a[2] - __uint_as_float(hi.y << 16), a[3] - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = pk_bf16(b[0] - __uint_as_float(hi.z << 16), b[1] - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = pk_bf16(b[2] - __uint_as_float(hi.w << 16), b[3] - __uint_as_float(hi.w & 0xffff0000u));
}
__device__ __forceinline__ void store_o4(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store_o4(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = pk_bf16(v[0], v[1]);
    t.y = pk_bf16(v[2], v[3]);
    *(uint2*)p = t;
}

__device__ __forceinline__ void store_o4(bx3_t* p, const float v[4]) { bx3_store4(p, v); }      // pre-split for a bf16x3 projection
__device__ __forceinline__ void store_o4(fp8_t* p, const float v[4]) {     // values already scaled by 1 / (activation scale)
    float a = fminf(fmaxf(v[0], -FP8_MAX), FP8_MAX), b = fminf(fmaxf(v[1], -FP8_MAX), FP8_MAX);
    float c = fminf(fmaxf(v[2], -FP8_MAX), FP8_MAX), d = fminf(fmaxf(v[3], -FP8_MAX), FP8_MAX);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    *(uint32_t*)p = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

__device__ u32x4 d2s_attn_zero_page[4];

// XCD-aware block -> (q tile, head, frame).  Workgroup b lands on XCD b % 8 (observed; speed only), each XCD has its own L2.
// With a (q tile, head, frame) grid the q tiles of one (frame, head) -- which all read the SAME K / V^T -- sit on 7 different
// XCDs, and every L2 fetches that K / V^T again from the fabric: measured at batch 32, both kernels ran at the same
// ~4.5-5 TB/s of K / V fill however their inner loops were built.  Here the grid is linear: XCD x owns a contiguous range of
// (frame, head) pairs and walks it q tile by q tile, so the blocks resident on one XCD at a time share their K / V^T in its L2.
// (fewer than 16 pairs -- batch 1 -- would leave XCDs unevenly loaded: plain order there, every block is resident at once anyway)
__device__ __forceinline__ bool attn_block(int bid, int qtiles, int pairs, int heads, int& qt, int& h, int& b) {
    if (pairs < 16) { const int pr = bid / qtiles; qt = bid - pr * qtiles; b = pr / heads; h = pr - b * heads; return pr < pairs; }
    const int x = bid & 7, seq = bid >> 3;
    const int p0 = (int)(((long)pairs * x) >> 3), p1 = (int)(((long)pairs * (x + 1)) >> 3);
    const int pr = p0 + seq / qtiles;
    if (pr >= p1) return false;
    qt = seq - (seq / qtiles) * qtiles;
    b = pr / heads; h = pr - b * heads;
    return true;
}
static inline unsigned attn_grid(int qtiles, int pairs) { return pairs < 16 ? (unsigned)(pairs * qtiles) : 8u * (unsigned)(cdiv(pairs, 8) * qtiles); }


// OT: output element type (T, or e4m3 when the output projection runs on fp8 operands: out = sat(result * oscale))
// KS > 1 (batch 1: 156 blocks of 4 waves leave the chip empty and each walks 13 key tiles in sequence): KS wave groups
// per block share the q tile and split the KEY range; each group runs the whole pipeline over its keys with its own LDS
// ring, and the groups' (max, sum, O) are merged through LDS at the end (flash-decoding style).
template <typename T, int QF, int NW, int NS = 3, typename OT = T, int KS = 1>
__global__ void __launch_bounds__(64 * NW * KS)
attention_kernel(const T* __restrict__ qkv, const T* __restrict__ vt, OT* __restrict__ out,
                 int N, int Npad, int heads, int pairs, float scale_log2e, float oscale) {
    using A = AT<T>;
    constexpr int CE = A::CE, CPR = A::CPR, NKS = A::NKS;
    constexpr int TILE_CHUNKS = 64 * CPR;              // chunks in one 64-row tile
    constexpr int BQ = NW * QF * 16;
    constexpr int PD = NS - 1;                         // NS LDS ring stages, prefetch distance PD
    __shared__ __attribute__((aligned(16))) u32x4 lds_all[KS * NS * 2 * TILE_CHUNKS];   // [group][stage][K | V^T]
    D2S_POISON_LDS(lds_all, KS * NS * 2 * TILE_CHUNKS)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_all = tid >> 6, grp = KS == 1 ? 0 : wid_all / NW, wid = KS == 1 ? wid_all : wid_all % NW;
    u32x4* const lds = lds_all + grp * (NS * 2 * TILE_CHUNKS);
    const int fr = lane & 15, fg = lane >> 4;
    int qt, h, b;
    if (!attn_block(blockIdx.x, (N + BQ - 1) / BQ, pairs, heads, qt, h, b)) return;
    const int D = heads * 64;
    const long row3 = 3L * D;
    const T* qbase = qkv + (long)b * N * row3 + h * 64;
    const T* kbase = qbase + D;
    const T* vbase = vt + ((long)b * heads + h) * 64 * Npad;

    // ---- Q fragments (B operand of S^T): Q[q][chunk ks*4+fg]   (bf16x3: the hi and the lo chunk of unit ks * 4 + ATT_BX3_UNIT(fg))
    constexpr bool X3 = std::is_same<T, bx3_t>::value;
    u32x4 qf[QF][NKS], qfl[QF][X3 ? NKS : 1];
    int qrow[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        qrow[f] = qt * BQ + (wid * QF + f) * 16 + fr;
        bool ok = qrow[f] < N;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if constexpr (X3) {
                const int c_ = 2 * (ks * 4 + ATT_BX3_UNIT(fg));
                qf[f][ks] = ok ? *(const u32x4*)(qbase + (long)qrow[f] * row3 + c_ * CE) : (u32x4){0u, 0u, 0u, 0u};
                qfl[f][ks] = ok ? *(const u32x4*)(qbase + (long)qrow[f] * row3 + (c_ + 1) * CE) : (u32x4){0u, 0u, 0u, 0u};
            } else
                qf[f][ks] = ok ? *(const u32x4*)(qbase + (long)qrow[f] * row3 + (ks * 4 + fg) * CE) : (u32x4){0u, 0u, 0u, 0u};
        }
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
                if constexpr (X3) {
                    const int c_ = 2 * (ks * 4 + ATT_BX3_UNIT(fg));
                    const u32x4 kh = Kl[kr * CPR + (c_ ^ A::swzK(kr))], kl = Kl[kr * CPR + ((c_ + 1) ^ A::swzK(kr))];
#pragma unroll
                    for (int f = 0; f < QF; ++f) mma16x3(s[f][j], kh, kl, qf[f][ks], qfl[f][ks]);
                } else {
                    u32x4 kf = Kl[kr * CPR + ((ks * 4 + fg) ^ A::swzK(kr))];
#pragma unroll
                    for (int f = 0; f < QF; ++f) mma16(s[f][j], kf, qf[f][ks], T());
                }
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
            if constexpr (X3) {
                // lane group fg holds keys 32 pc + 8 fg .. + 7 of the tile = unit 4 pc + fg of a V^T row (both operands agree)
                u32x4 ph[QF], pl[QF];
#pragma unroll
                for (int f = 0; f < QF; ++f) pack_p3(s[f][2 * pc], s[f][2 * pc + 1], ph[f], pl[f]);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int vr = d * 16 + fr, c_ = 2 * (pc * 4 + fg);
                    const u32x4 vh = Vl[vr * CPR + (c_ ^ A::swzV(vr))], vl = Vl[vr * CPR + ((c_ + 1) ^ A::swzV(vr))];
#pragma unroll
                    for (int f = 0; f < QF; ++f) mma16x3(o[f][d], vh, vl, ph[f], pl[f]);
                }
            } else {
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

// ================================================================================================
// Batched regime (bf16): the same algorithm on v_mfma_f32_32x32x16_bf16 with 32 query rows per wave.
// The 16-row kernel above makes every wave read the whole K and V^T tile (16 KiB of ds_read_b128) for 16 MFMAs of 16 cycles:
// LDS reads, matrix pipe and the softmax VALU all need ~512 cycles per block-tile, so the LDS pipe has no slack and the three
// blocks of a CU cannot overlap (r2: 0.21 of the MFMA peak).  Here a wave reads the same 16 KiB for 16 MFMAs of 32 cycles
// (half the LDS bytes and half the ds_read / address instructions per flop) and the softmax reduction over keys is lane-local
// but for ONE half-swap:
//   S^T[key, q] = mfma32(K rows, Q rows):  lane (q = l & 31, hi = l >> 5) ends with 16 keys of one query per 32-key block.
//   The K row fed to MFMA row i is key (i with bits 2 and 3 swapped), so register r = 8 t + a of a lane is key
//   16 t + 8 hi + a of the block: eight CONSECUTIVE keys = one 16-byte chunk of a V^T row = the lane's B operand of
//   O^T[d, q] += mfma32(V^T rows, P^T) for the 16-key step t, straight from the S registers.
// Block = 4 waves x 32 rows (three blocks per CU: 48 KiB of LDS, <= 168 VGPRs); waves whose 32 rows lie past N (the last q tile
// of N = 778 has 10 live rows) only help with the loads; a last key tile with <= 32 live keys runs one key block.
// ================================================================================================
typedef __attribute__((ext_vector_type(16))) float f32x16;

// 16 bytes per lane, global -> LDS (1 KiB per wave-instruction, lane-linear in LDS) through a buffer descriptor.  (A __device__
// wrapper: called with non-dependent arguments straight from a __global__ template, the device-only builtin makes the HOST pass
// drop the kernel's launch stub without a diagnostic.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, u32x4* lds_dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ void* uniform_ptr(const void* p) {      // a wave-uniform pointer the compiler can SEE is uniform
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long long)hi32 << 32) | lo);
}

__device__ __forceinline__ f32x16 mma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, c, 0, 0, 0);
}

template <typename OT, int NW = 4, int NS = 3>
__global__ void __launch_bounds__(64 * NW, 3)
attention32_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ vt, OT* __restrict__ out,
                   int N, int Npad, int heads, int pairs, float oscale) {
    constexpr int CE = 8, CPR = 8, TILE = 64 * CPR;   // chunks in a 64-row tile (8 KiB)
    constexpr int PD = NS - 1;
    constexpr int BQ = NW * 32;
    constexpr int IPW = 8 / NW;                        // LDS-DMA instructions per wave per operand tile (8 rows each)
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * 2 * TILE];
    D2S_POISON_LDS(lds, NS * 2 * TILE)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 31, hi = lane >> 5;
    int qt, h, b;
    if (!attn_block(blockIdx.x, (N + BQ - 1) / BQ, pairs, heads, qt, h, b)) return;
    const int D = heads * 64;
    const long row3 = 3L * D;
    const bf16_t* qbase = qkv + (long)b * N * row3 + h * 64;
    const int q0 = qt * BQ + wu * 32;                  // this wave's first query row
    const bool dead = q0 >= N;                         // wave-uniform: nothing to compute, loads and barriers only
    const int qrow = q0 + fi;

    // Q fragments (B operand of S^T): Q[q][chunk 2 ks + hi].  The q columns arrive PRE-SCALED by 64^-0.5 log2 e (folded into
    // W_q / b_q when the engine packs its weights), so S is already in the log2 domain.
    u32x4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = qrow < N ? *(const u32x4*)(qbase + (long)qrow * row3 + (ks * 2 + hi) * CE) : (u32x4){0u, 0u, 0u, 0u};

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o0 = zero16, o1 = zero16;                   // O^T: d blocks 0..31 / 32..63
    // Running maximum m of each query (log2 domain), kept NEGATED and splat over a whole accumulator tuple: the first MFMA of
    // every S chain takes it as its C operand, so the matrix pipe delivers S - m and the softmax needs no subtraction (the
    // fused multiply-add in front of each of the 32 v_exp_f32 per tile was a quarter of the kernel's VALU work, and the kernel
    // is VALU-bound).  The tuple is rewritten only when the maximum is raised, which after the first tile is rare (below).
    f32x16 negm = zero16;
    // Row sums of P on the matrix pipe: a 16 x 16 x 32 MFMA whose B operand is the SAME P register as the PV MFMAs'.  Read as a
    // 16-column operand, lane L supplies 8 keys of query (L & 31) in k group L >> 4; the selector A[row][k group] = 1 where
    // ((row >> 2) & 1) == (k group & 1) adds the two k groups of query (L & 15) into rows {0-3, 8-11} and those of query
    // 16 + (L & 15) into rows {4-7, 12-15} -- so lane L's accumulator (rows 4 (L >> 4) ..) is the sum for ITS query, both halves.
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
    const unsigned selw = (((lane & 15) >> 2) & 1) == ((lane >> 4) & 1) ? 0x3F803F80u : 0u;
    const u32x4 sel = {selw, selw, selw, selw};

    // K / V^T tiles: LDS-DMA through buffer descriptors -- the per-lane offsets are computed once, a tile is a scalar offset,
    // and key rows past N fall outside the K descriptor's range (the hardware returns zeros for them: no select, no zero page).
    const bf16_t* kbase = qbase + D;
    const unsigned kbytes = (unsigned)(((long)N * row3 - (h * 64 + D)) * 2);           // from kbase to the end of this frame's rows
    // (descriptor words through readfirstlane: uniform in fact, but not provably so to the compiler, which would otherwise wrap
    //  every load in a waterfall loop)
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(kbase), 0, __builtin_amdgcn_readfirstlane(kbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(vt + ((long)b * heads + h) * 64 * Npad), 0,
                                                                         __builtin_amdgcn_readfirstlane((unsigned)(64 * Npad * 2)), 0x00020000);
    unsigned voK[IPW], voV[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int r_ = (i * NW + wu) * 8 + (lane >> 3), c_ = ((lane & 7) ^ ((r_ >> 1) & 7)) * 16;
        voK[i] = (unsigned)(r_ * row3 * 2) + c_;
        voV[i] = (unsigned)(r_ * Npad * 2) + c_;
    }
    const unsigned ktile = (unsigned)(64 * row3 * 2);  // bytes between key tiles (ViT-L: 393 KiB)
#define ATT_ISSUE(TT)                                                                                                    \
    {                                                                                                                    \
        u32x4* st_ = lds + ((TT) % NS) * 2 * TILE;                                                                       \
        _Pragma("unroll") for (int i = 0; i < IPW; ++i) {                                                                \
            /* (the key-tile offset rides in the VECTOR offset: that is the one the range check is certain to cover) */    \
            lds_dma16(rsK, st_ + (i * NW + wu) * 64, voK[i] + (unsigned)(TT) * ktile, 0);                                \
            lds_dma16(rsV, st_ + TILE + (i * NW + wu) * 64, voV[i], (TT) * 128);                                         \
        }                                                                                                                \
    }
    constexpr int LPTA = 2 * IPW;
    const int ntiles = (N + 63) / 64;
    // fragment rows: MFMA row fi of a key block is key perm(fi) (bits 2 <-> 3); V^T rows are taken in order
    const int krow = (fi & ~12) | ((fi & 4) << 1) | ((fi & 8) >> 1);
    const int ksw = (krow >> 1) & 7, vsw = (fi >> 1) & 7;     // (rows + 32 have the same swizzle)
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < ntiles) ATT_ISSUE(t)
    for (int t = 0; t < ntiles; ++t) {
        if (t + PD - 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * LPTA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + PD < ntiles) ATT_ISSUE(t + PD)
        if (dead) continue;
        const u32x4* Kl = lds + (t % NS) * 2 * TILE;
        const u32x4* Vl = Kl + TILE;
        const int live = N - t * 64;                   // keys of this tile that exist (>= 1)
        const bool two = live > 32;                    // wave-uniform
        // ---- S^T - m = K Q^T - m (the two key blocks interleaved: no MFMA waits for its predecessor's accumulator)
        f32x16 s0, s1;
        if (two) {
            s0 = mma32(Kl[krow * CPR + (hi ^ ksw)], qf[0], negm);
            s1 = mma32(Kl[(32 + krow) * CPR + (hi ^ ksw)], qf[0], negm);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                s0 = mma32(Kl[krow * CPR + ((ks * 2 + hi) ^ ksw)], qf[ks], s0);
                s1 = mma32(Kl[(32 + krow) * CPR + ((ks * 2 + hi) ^ ksw)], qf[ks], s1);
            }
        } else {
            s0 = mma32(Kl[krow * CPR + (hi ^ ksw)], qf[0], negm);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) s0 = mma32(Kl[krow * CPR + ((ks * 2 + hi) ^ ksw)], qf[ks], s0);
            s1 = zero16;                               // (masked below: live <= 32)
        }
        // ---- ragged last tile: register r of block kb is key kb * 32 + 16 (r >> 3) + 8 hi + (r & 7)
        if (live < 64) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 16 * (r >> 3) + 8 * hi + (r & 7);
                if (k >= live) s0[r] = -1e30f;
                if (32 + k >= live) s1[r] = -1e30f;
            }
        }
        // ---- online softmax; all 64 keys of a query sit in lanes l and l ^ 32.  mx = (tile maximum) - m.
        // m is raised only when some row's tile maximum exceeds it by more than 8 (wave-uniform test): P = 2^(s - m) then lies
        // in (0, 256] instead of (0, 1] -- the same relative precision in bf16 and fp32 -- and the rescale of O / l / S and the
        // rewrite of the -m tuple (~90 VALU instructions) run on the first tile and almost never again.  The first tile always
        // takes it (m starts at 0, and its maximum may lie far BELOW 0, where 2^s would underflow).
        float mxa = fmaxf(fmaxf(s0[0], s0[1]), s0[2]), mxb = fmaxf(fmaxf(s1[0], s1[1]), s1[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) { mxa = fmaxf(fmaxf(mxa, s0[r]), s0[r + 1]); mxb = fmaxf(fmaxf(mxb, s1[r]), s1[r + 1]); }
        float mx = fmaxf(fmaxf(mxa, mxb), fmaxf(s0[15], s1[15]));
        mx = xmax32(mx);
        if (t == 0 || !__all(mx <= 8.0f)) {
            const float delta = t == 0 ? mx : fmaxf(mx, 0.f);                  // 0 for the rows that stay as they are
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; s0[r] -= delta; s1[r] -= delta; negm[r] -= delta; }
            lacc *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r]); s1[r] = __builtin_amdgcn_exp2f(s1[r]); }
        // ---- O^T += V^T P^T: 16-key step (kb, ts) is chunk 4 kb + 2 ts + hi of a V^T row; the row sums of P ride on the
        // matrix pipe too (one 16 x 16 x 32 MFMA per step against the 0 / 1 selector `sel`), not on 32 VALU adds per tile
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            u32x4 pf;
            pf.x = pk_bf16(s0[8 * ts + 0], s0[8 * ts + 1]); pf.y = pk_bf16(s0[8 * ts + 2], s0[8 * ts + 3]);
            pf.z = pk_bf16(s0[8 * ts + 4], s0[8 * ts + 5]); pf.w = pk_bf16(s0[8 * ts + 6], s0[8 * ts + 7]);
            o0 = mma32(Vl[fi * CPR + ((2 * ts + hi) ^ vsw)], pf, o0);
            o1 = mma32(Vl[(32 + fi) * CPR + ((2 * ts + hi) ^ vsw)], pf, o1);
            lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&sel, *(const bf16x8*)&pf, lacc, 0, 0, 0);
        }
        if (two) {
#pragma unroll
            for (int ts = 0; ts < 2; ++ts) {
                u32x4 pf;
                pf.x = pk_bf16(s1[8 * ts + 0], s1[8 * ts + 1]); pf.y = pk_bf16(s1[8 * ts + 2], s1[8 * ts + 3]);
                pf.z = pk_bf16(s1[8 * ts + 4], s1[8 * ts + 5]); pf.w = pk_bf16(s1[8 * ts + 6], s1[8 * ts + 7]);
                o0 = mma32(Vl[fi * CPR + ((4 + 2 * ts + hi) ^ vsw)], pf, o0);
                o1 = mma32(Vl[(32 + fi) * CPR + ((4 + 2 * ts + hi) ^ vsw)], pf, o1);
                lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&sel, *(const bf16x8*)&pf, lacc, 0, 0, 0);
            }
        }
    }
#undef ATT_ISSUE
    if (dead) return;
    // ---- normalise and store.  Register r of block db is d = 32 db + 8 (r >> 2) + 4 hi + (r & 3): the two half-waves hold
    // alternating 4-wide pieces of a row.  One v_permlane32_swap per dword pairs them up so that each lane stores 8
    // consecutive d (16 bytes of bf16): lower half d = 8 g .. 8 g + 7, upper half the next eight.
    const float inv = oscale / lacc[0];
    if constexpr (std::is_same<OT, bf16_t>::value) {
        bf16_t* orow = out + ((long)b * N + qrow) * D + h * 64 + hi * 8;
        auto store_block = [&](const f32x16& o, int db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                unsigned a0 = pk_bf16(o[4 * g + 0] * inv, o[4 * g + 1] * inv), a1 = pk_bf16(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                unsigned b0 = pk_bf16(o[4 * g + 4] * inv, o[4 * g + 5] * inv), b1 = pk_bf16(o[4 * g + 6] * inv, o[4 * g + 7] * inv);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                // lower half: (own g | upper's g);  upper half: (lower's g + 1 | own g + 1)
                const uint4 v = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                if (qrow < N) *(uint4*)(orow + db * 32 + g * 8) = v;
            }
        };
        store_block(o0, 0);
        store_block(o1, 1);
    } else {
        OT* orow = out + ((long)b * N + qrow) * D + h * 64 + hi * 4;
        if (qrow < N) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0[4] = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                float v1[4] = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                store_o4(orow + g * 8, v0);
                store_o4(orow + 32 + g * 8, v1);
            }
        }
    }
}

int launch_attention(int prec, const void* qkv, const void* vt, void* out, int B, int N, int Npad, int heads, hipStream_t st,
                     float fp8_qscale, bool prescaled, bool bx3_out) {
    // prescaled: the q columns already carry 64^-0.5 * log2(e) (folded into W_q / b_q, engine.hip): scores are log2-domain as they are
    const float scale_log2e = prescaled ? 1.0f : ATTN_SCALE_LOG2E;
    if (fp8_qscale > 0.f && prec != D2S_PREC_BF16) { set_error("attention: e4m3 output needs bf16 inputs"); return D2S_E_UNSUPPORTED; }
    static EnvInt attn32_min{"D2S_ATTN32_MIN", 168};        // (from batch 2 at N = 778: +0.3 % at 2, +1.5 % at 4, +4.5 % at 6; batch 1 -6 %)
    // batched bf16: the 32 x 32 kernel from ~170 blocks of 128 rows (D2S_ATTN32=0: the 16-row kernel everywhere)
    // (read per call, only for launches in that regime: the parity test flips it inside one process)
    if (prescaled && prec == D2S_PREC_BF16 && (long)cdiv(N, 128) * heads * B >= attn32_min.get() && !(getenv("D2S_ATTN32") && atoi(getenv("D2S_ATTN32")) == 0)) {
        if ((long)N * 3 * heads * 64 * 2 >= (1L << 31)) { set_error("attention: frame too large for 32-bit buffer offsets"); return D2S_E_UNSUPPORTED; }
        const dim3 grid(attn_grid(cdiv(N, 128), heads * B));
        if (fp8_qscale > 0.f)
            hipLaunchKernelGGL((attention32_kernel<fp8_t>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)vt, (fp8_t*)out, N, Npad, heads, heads * B, fp8_qscale);
        else
            hipLaunchKernelGGL((attention32_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, N, Npad, heads, heads * B, 1.0f);
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    // q rows per block: 128 as 8 waves x 1 fragment once that fills the chip (batch >= ~8), else 64 as 4 waves.
    // Swept at batch 1 / 16: 4 waves x 2 fragments 80 us, ring depth 2 / 4 within 3 %, 8 x 2 (256 rows) 68 us,
    // 8 x 1 67 us (444 TFLOP/s); 32-row blocks slower (K/V tile loads not amortised).  D2S_ATTN_BQ forces 128 / 64 / 32.
    static const int force = getenv("D2S_ATTN_BQ") ? atoi(getenv("D2S_ATTN_BQ")) : 0;
    long hb = (long)heads * B;
    int bq = force ? force : (cdiv(N, 128) * hb >= 512 ? 128 : 64);
    const int pairs = heads * B;
#define D2S_ATT(TT, QF_, NW_) hipLaunchKernelGGL((attention_kernel<TT, QF_, NW_>), dim3(attn_grid(cdiv(N, NW_ * QF_ * 16), pairs)), dim3(64 * NW_), 0, st, \
        (const TT*)qkv, (const TT*)vt, (TT*)out, N, Npad, heads, pairs, scale_log2e, 1.0f)
#define D2S_ATT8(QF_, NW_) hipLaunchKernelGGL((attention_kernel<bf16_t, QF_, NW_, 3, fp8_t>), dim3(attn_grid(cdiv(N, NW_ * QF_ * 16), pairs)), dim3(64 * NW_), 0, st, \
        (const bf16_t*)qkv, (const bf16_t*)vt, (fp8_t*)out, N, Npad, heads, pairs, scale_log2e, fp8_qscale)
    // too few (q-tile, head) blocks to fill 256 CUs (batch 1: 156): split the keys over 2 / 4 wave groups per block
    // (batch 1, N = 778: 18.9 us -> 13.8 us with 2 groups, 13.2 us with 4)
    static const int force_ks = getenv("D2S_ATTN_KS") ? atoi(getenv("D2S_ATTN_KS")) : 0;
    const int ks = force_ks ? force_ks : (bq == 64 && (long)cdiv(N, 64) * hb < 256 ? (N >= 512 ? 4 : (N >= 256 ? 2 : 1)) : 1);
    if (fp8_qscale > 0.f) {
        if (bq == 128) D2S_ATT8(1, 8);
        else if (ks >= 2)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 2, fp8_t, 4>), dim3(attn_grid(cdiv(N, 64), pairs)), dim3(1024), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (fp8_t*)out, N, Npad, heads, pairs, scale_log2e, fp8_qscale);
        else D2S_ATT8(1, 4);
    } else if (prec == D2S_PREC_BF16) {
        if (bq == 64 && ks == 2)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 3, bf16_t, 2>), dim3(attn_grid(cdiv(N, 64), pairs)), dim3(512), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, N, Npad, heads, pairs, scale_log2e, 1.0f);
        else if (bq == 64 && ks == 4)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 2, bf16_t, 4>), dim3(attn_grid(cdiv(N, 64), pairs)), dim3(1024), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, N, Npad, heads, pairs, scale_log2e, 1.0f);
        else if (bq == 64 && ks == 3)       // (D2S_ATTN_KS=3: three groups, three ring stages each = two key tiles in flight per group.
                                            //  10.5 us against 10.6 / 10.8 with 4 / 2 groups at N = 778: the batch-1 launch is not waiting
                                            //  for its key tiles -- 156 blocks keep 156 of 256 CUs at four waves per SIMD of softmax VALU work)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 1, 4, 3, bf16_t, 3>), dim3(attn_grid(cdiv(N, 64), pairs)), dim3(768), 0, st,
                               (const bf16_t*)qkv, (const bf16_t*)vt, (bf16_t*)out, N, Npad, heads, pairs, scale_log2e, 1.0f);
        else if (bq == 128) D2S_ATT(bf16_t, 1, 8);
        else if (bq == 64) D2S_ATT(bf16_t, 1, 4);
        else D2S_ATT(bf16_t, 1, 2);
    } else if (prec == D2S_PREC_BF16X3) {
        // split-precision inputs (q | k and V^T pre-split by the QKV epilogue) and output; batch-1-sized launches split the keys.
        // (64-element rows are 256 bytes: two ring stages of K | V^T are 64 KiB -> two blocks per CU)
#define D2S_ATT3(NW_, NS_, KS_) hipLaunchKernelGGL((attention_kernel<bx3_t, 1, NW_, NS_, bx3_t, KS_>), dim3(attn_grid(cdiv(N, NW_ * 16), pairs)), dim3(64 * NW_ * KS_), 0, st, \
        (const bx3_t*)qkv, (const bx3_t*)vt, (bx3_t*)out, N, Npad, heads, pairs, scale_log2e, 1.0f)
        if (bq == 128) D2S_ATT3(8, 2, 1);
        else if (bq == 64 && ks >= 2) D2S_ATT3(4, 2, 2);
        else if (bq == 64) D2S_ATT3(4, 2, 1);
        else D2S_ATT3(2, 3, 1);
#undef D2S_ATT3
    } else if (bx3_out) {
#define D2S_ATTX(NW_) hipLaunchKernelGGL((attention_kernel<float, 1, NW_, 3, bx3_t>), dim3(attn_grid(cdiv(N, NW_ * 16), pairs)), dim3(64 * NW_), 0, st, \
        (const float*)qkv, (const float*)vt, (bx3_t*)out, N, Npad, heads, pairs, scale_log2e, 1.0f)
        if (bq >= 64) D2S_ATTX(4); else D2S_ATTX(2);
#undef D2S_ATTX
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

// ---- test / micro-benchmark probe ----------------------------------------------------------------
namespace d2s {
// q | k | v float32 [B, heads, N, 64] -> the engine's activation layout: qkv [B*N, 3D] (q | k | v, head-major) and
// V^T [B, heads, 64, Npad] in the operand type
template <typename T>
__global__ void attn_probe_pack_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                       T* __restrict__ qkv, T* __restrict__ vt, int B, int heads, int N, int Npad, float qscale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * heads * N * 64;
    if (idx >= total) return;
    const int d = (int)(idx & 63);
    const int t = (int)((idx >> 6) % N);
    const int h = (int)((idx / (64L * N)) % heads);
    const int b = (int)(idx / (64L * N * heads));
    const int D = heads * 64;
    T* row = qkv + ((long)b * N + t) * 3 * D + h * 64 + d;
    T* vrow = vt + (((long)b * heads + h) * 64 + d) * Npad + t;
    if constexpr (std::is_same<T, bx3_t>::value) {
        bx3_store1(row, q[idx]); bx3_store1(row + D, k[idx]); bx3_store1(row + 2 * D, v[idx]); bx3_store1(vrow, v[idx]);
    } else {
        auto cv = [](float x) -> T { if constexpr (sizeof(T) == 2) return f2bf(x); else return x; };
        row[0] = cv(q[idx] * qscale); row[D] = cv(k[idx]); row[2 * D] = cv(v[idx]);       // (qscale: what the engine folds into W_q)
        *vrow = cv(v[idx]);
    }
}
template <typename T>
__global__ void attn_probe_unpack_kernel(const T* __restrict__ o, float* __restrict__ out, long n) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if constexpr (std::is_same<T, bx3_t>::value) out[idx] = bx3_load1(o + idx);
    else if constexpr (sizeof(T) == 2) out[idx] = bf2f(o[idx]); else out[idx] = o[idx];
}
}  // namespace d2s

extern "C" int d2s_attention_probe(const float* q, const float* k, const float* v, float* out, int B, int heads, int N,
                                   int precision, int iters, float* ms_per_iter, void* stream) {
    using namespace d2s;
    D2S_REQUIRE(q && k && v && out && B > 0 && heads > 0 && N > 0 && iters >= 1, "bad argument");
    D2S_REQUIRE(precision == D2S_PREC_BF16 || precision == D2S_PREC_FP32 || precision == D2S_PREC_BF16X3, "bad precision");
    hipStream_t st = (hipStream_t)stream;
    const int Npad = (N + 63) / 64 * 64, D = heads * 64;
    const size_t es = precision == D2S_PREC_BF16 ? 2 : 4;
    void *dqkv = nullptr, *dvt = nullptr, *dout = nullptr;
    D2S_HIP(hipMalloc(&dqkv, (size_t)B * N * 3 * D * es));
    D2S_HIP(hipMalloc(&dvt, (size_t)B * D * Npad * es));
    D2S_HIP(hipMalloc(&dout, (size_t)B * N * D * es));
    D2S_HIP(hipMemsetAsync(dvt, 0, (size_t)B * D * Npad * es, st));
    const long total = (long)B * heads * N * 64;
    if (precision == D2S_PREC_BF16)
        hipLaunchKernelGGL((attn_probe_pack_kernel<bf16_t>), dim3(cdiv(total, 256)), dim3(256), 0, st, q, k, v, (bf16_t*)dqkv, (bf16_t*)dvt, B, heads, N, Npad, ATTN_SCALE_LOG2E);
    else if (precision == D2S_PREC_BF16X3)
        hipLaunchKernelGGL((attn_probe_pack_kernel<bx3_t>), dim3(cdiv(total, 256)), dim3(256), 0, st, q, k, v, (bx3_t*)dqkv, (bx3_t*)dvt, B, heads, N, Npad, 1.0f);
    else
        hipLaunchKernelGGL((attn_probe_pack_kernel<float>), dim3(cdiv(total, 256)), dim3(256), 0, st, q, k, v, (float*)dqkv, (float*)dvt, B, heads, N, Npad, 1.0f);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    D2S_HIP(hipEventCreate(&e0)); D2S_HIP(hipEventCreate(&e1));
    const bool pre = precision == D2S_PREC_BF16;      // as the engines do: bf16 engines fold the scale into W_q, fp32 keeps the reference's order
    int rc = launch_attention(precision, dqkv, dvt, dout, B, N, Npad, heads, st, 0.f, pre);      // warm-up (and the result)
    D2S_HIP(hipEventRecord(e0, st));
    for (int i = 1; i < iters && rc == D2S_OK; ++i) rc = launch_attention(precision, dqkv, dvt, dout, B, N, Npad, heads, st, 0.f, pre);
    D2S_HIP(hipEventRecord(e1, st));
    if (precision == D2S_PREC_BF16)
        hipLaunchKernelGGL((attn_probe_unpack_kernel<bf16_t>), dim3(cdiv((long)B * N * D, 256)), dim3(256), 0, st, (const bf16_t*)dout, out, (long)B * N * D);
    else if (precision == D2S_PREC_BF16X3)
        hipLaunchKernelGGL((attn_probe_unpack_kernel<bx3_t>), dim3(cdiv((long)B * N * D, 256)), dim3(256), 0, st, (const bx3_t*)dout, out, (long)B * N * D);
    else
        hipLaunchKernelGGL((attn_probe_unpack_kernel<float>), dim3(cdiv((long)B * N * D, 256)), dim3(256), 0, st, (const float*)dout, out, (long)B * N * D);
    hipError_t err = hipStreamSynchronize(st);
    float ms = 0.f;
    if (err == hipSuccess && iters > 1) (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms_per_iter) *ms_per_iter = iters > 1 ? ms / (float)(iters - 1) : 0.f;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dqkv); (void)hipFree(dvt); (void)hipFree(dout);
    if (rc != D2S_OK) return rc;
    D2S_HIP(err);
    return D2S_OK;
}
