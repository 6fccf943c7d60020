// Device-side helpers shared by the MFMA GEMM kernels (gemm.hip, gemm_pp.hip): fragment types, packed converts,
// the fused epilogue (bias / GELU / LayerScale / residuals / row re-mapping / QKV split / e4m3 de-quantisation) and the
// XCD-aware block -> tile map.
#pragma once
#include "gemm.h"
#include <type_traits>

namespace d2s {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers (HIP's u32x4 struct arrays went to scratch)

// Split-precision operand ("bf16x3"): the element is 4 bytes like a float, but a row is stored as 32-byte UNITS of 8 elements,
// [8 x bf16 hi | 8 x bf16 lo] with hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to ~16 mantissa bits).  A 16-byte chunk is thus
// either the hi or the lo halves of 8 consecutive elements -- exactly one MFMA operand -- and the byte geometry (128-byte K tile of
// 32 elements, 8 chunks per row, swizzle) is the float kernels'.  Weights are packed in this format at engine build; activations
// stay fp32 in HBM and are split on their way into LDS (register-staged tiles), so nothing else in the engine knows the format.
// 16 bytes per lane, global -> LDS (1 KiB per wave-instruction, lane-linear in LDS) through a buffer descriptor: the per-lane offset
// is fixed for the whole K loop, a K tile is the scalar offset.  (A __device__ wrapper: called with non-dependent arguments straight
// from a __global__ template, the device-only builtin makes the HOST pass drop the kernel's launch stub without a diagnostic.)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_lds_;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, void* lds_dst, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ void* uniform_ptr(const void* p) {      // a wave-uniform pointer the compiler can SEE is uniform (no waterfall loop)
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long long)hi32 << 32) | lo);
}

template <typename T> struct Prec;
template <> struct Prec<bx3_t>  { static constexpr int CE = 4; };   // "elements" per 16-byte chunk for address arithmetic (as float)
template <> struct Prec<bf16_t> { static constexpr int CE = 8; };   // elements per 16-byte chunk
template <> struct Prec<float>  { static constexpr int CE = 4; };
template <> struct Prec<fp8_t>  { static constexpr int CE = 16; };  // e4m3: a 128-byte K tile holds 128 elements

__device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& w, const u32x4& a, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w, *(const bf16x8*)&a, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& w, const u32x4& a, float) {
    const float* wf = (const float*)&w;
    const float* af = (const float*)&a;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t], af[t], acc, 0, 0, 0);
}

// bf16x3: lane group fg of an MFMA takes unit BX3_UNIT(fg) of the K tile (the same for both operands, so the sum is unchanged).
// Not the identity: with the row swizzle (r >> 1) & 7 the chunk pairs {0,1} {6,7} / {2,3} {4,5} of lane groups 0 / 1 and 2 / 3 keep
// the 16-lane ds_read_b128 service groups conflict-free (XOR partners must lie in the subgroup {0, 1, 6, 7}).
#define BX3_UNIT(fg) ((0x2130 >> (4 * (fg))) & 3)                   /* 0 -> 0, 1 -> 3, 2 -> 1, 3 -> 2 */
// x = hi + lo (the dropped lo * lo term is 2^-16 relative): three bf16 MFMAs
__device__ __forceinline__ void mma_bx3(f32x4& acc, const u32x4& w_hi, const u32x4& w_lo, const u32x4& a_hi, const u32x4& a_lo) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w_lo, *(const bf16x8*)&a_hi, acc, 0, 0, 0);     // small terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w_hi, *(const bf16x8*)&a_lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&w_hi, *(const bf16x8*)&a_hi, acc, 0, 0, 0);
}

// e4m3 operands: a 16-byte chunk is 16 K elements = two v_mfma_f32_16x16x32_fp8_fp8 (8 bytes per lane each; the k
// permutation is again the same for both operands).  Same MFMA count per byte as bf16, twice the K per byte moved.
__device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& w, const u32x4& a, fp8_t) {
    const long* wl = (const long*)&w;
    const long* al = (const long*)&a;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[0], al[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[1], al[1], acc, 0, 0, 0);
}

// exact-erf GELU (HF Dinov2MLP: nn.GELU()).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, i.e. at float32 round-off for this
// use) instead of the ~40-instruction libm erff, on two values at once so that the
// multiplies / fmas become v_pk_mul_f32 / v_pk_fma_f32 (one instruction per PAIR), rearranged to
//     gelu(x) = x/2 (1 + erf(x / sqrt 2)) = max(x, 0) - (|x| P(t)/2) exp(-x^2 / 2),    t = 1 / (1 + p |x| / sqrt 2)
// (x/2 + |x|/2 = max(x, 0); the sign of erf cancels against the sign of x): 14 vector + 4 transcendental instructions per
// pair (max |err| 3.3e-7).  FC1's epilogue is 128 x 64 GELUs per wave with the matrix pipe idle in the ping-pong kernel -- about 6 us of a
// 24 us tile at batch 32 -- and 16 per lane between the K loop and the stores in the latency-regime tiles; ONE expression for every
// kernel, so that the engine's output does not depend on which kernel a batch size selects.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v gelu_erf2(f32x2v x) {
    const f32x2v ax = __builtin_elementwise_abs(x);
    const f32x2v den = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
    const f32x2v t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};       // v_rcp_f32 (1 ulp); __frcp_rn is a full division
    const f32x2v hp = t * (t * (t * (t * (t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f)) + (0.5f * 1.421413741f)) + (0.5f * -0.284496736f)) + (0.5f * 0.254829592f));
    const f32x2v arg = (ax * ax) * (-0.5f * 1.4426950408889634f);                           // exp(-x^2/2) = 2^arg
    const f32x2v ex = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
    const f32x2v relu = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return relu - (ax * hp) * ex;
}

__device__ __forceinline__ void load4(const float* p, float v[4]) { float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
    uint2 t = *(const uint2*)p;
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
// two floats -> packed bf16 pair (round-to-nearest-even) in one v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// One 16-byte chunk of an align_corners bilinear sample from its four tap chunks: bilerp1 (common.h) on every element, two at a
// time as v_pk_fma_f32; 8 bf16 (rounded like the stand-alone up-sample kernel's output) or 4 floats.
// (one dword = two bf16 channels of the chunk; kernels that spread a chunk's interpolation over their K loop call this directly)
__device__ __forceinline__ uint32_t lerp_pair_bf16(uint32_t v00, uint32_t v01, uint32_t v10, uint32_t v11, float w0x_, float w1x_, float w0y_, float w1y_) {
    typedef float f2_ __attribute__((ext_vector_type(2)));
    const f2_ w0x = {w0x_, w0x_}, w1x = {w1x_, w1x_}, w0y = {w0y_, w0y_}, w1y = {w1y_, w1y_};
    const f2_ a00 = {__uint_as_float(v00 << 16), __uint_as_float(v00 & 0xffff0000u)}, a01 = {__uint_as_float(v01 << 16), __uint_as_float(v01 & 0xffff0000u)};
    const f2_ a10 = {__uint_as_float(v10 << 16), __uint_as_float(v10 & 0xffff0000u)}, a11 = {__uint_as_float(v11 << 16), __uint_as_float(v11 & 0xffff0000u)};
    const f2_ top = __builtin_elementwise_fma(w1x, a01, w0x * a00);
    const f2_ bot = __builtin_elementwise_fma(w1x, a11, w0x * a10);
    const f2_ o = __builtin_elementwise_fma(w1y, bot, w0y * top);
    return pk_bf16(o[0], o[1]);
}
__device__ __forceinline__ u32x4 lerp_chunk(const u32x4& v00, const u32x4& v01, const u32x4& v10, const u32x4& v11, const Tap& tx, const Tap& ty, bf16_t) {
    u32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = lerp_pair_bf16(v00[q], v01[q], v10[q], v11[q], tx.w0, tx.w1, ty.w0, ty.w1);
    return r;
}
__device__ __forceinline__ u32x4 lerp_chunk(const u32x4& v00, const u32x4& v01, const u32x4& v10, const u32x4& v11, const Tap& tx, const Tap& ty, float) {
    u32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r[q] = __float_as_uint(bilerp1(tx, ty, __uint_as_float(v00[q]), __uint_as_float(v01[q]), __uint_as_float(v10[q]), __uint_as_float(v11[q])));
    return r;
}
// the chunk (channels c .. of pixel (iy, ix) of the UP-SAMPLED map) a convolution's halo loader fetches when GemmA.ups is set
template <typename T>
__device__ __forceinline__ u32x4 ups_chunk(const T* src_img /* [Hs, Ws, C] of this frame */, const GemmA& a, int iy, int ix, int c_elem) {
    const Tap ty = linear_tap(iy, a.usy, a.Hs, true), tx = linear_tap(ix, a.usx, a.Ws, true);
    const T* r0 = src_img + (long)ty.i0 * a.Ws * a.C + c_elem;
    const T* r1 = src_img + (long)ty.i1 * a.Ws * a.C + c_elem;
    const u32x4 v00 = *(const u32x4*)(r0 + tx.i0 * a.C), v01 = *(const u32x4*)(r0 + tx.i1 * a.C);
    const u32x4 v10 = *(const u32x4*)(r1 + tx.i0 * a.C), v11 = *(const u32x4*)(r1 + tx.i1 * a.C);
    return lerp_chunk(v00, v01, v10, v11, tx, ty, T());
}

// Halo fill of the LDS-resident-input convolutions: the (TH + 2) x (TW + 2) x C input window of an output tile, zero outside the image,
// optionally interpolated on the fly (a.ups).  GRP chunks per thread are REQUESTED before the first is used: written as a plain loop
// (load, wait, store; run-time trip count) the ~6 chunks of a thread were six dependent L2 round trips at the head of every block.
// slot(p, c): LDS index of chunk c of halo pixel p;  fin(v): the kernel's ReLU-on-load.
template <typename T, int NT, int GRP, typename SlotF, typename FinF>
__device__ __forceinline__ void conv_halo_fill(const GemmA& a, int b, int ty0, int tx0, int hwd, int npx, int cpp, int tid, u32x4* halo, SlotF slot, FinF fin) {
    constexpr int CE = 16 / (int)sizeof(T);
    const T* img = (const T*)a.ptr + (long)b * (a.ups ? (long)a.Hs * a.Ws : (long)a.Hi * a.Wi) * a.C;
    const int total = npx * cpp;
    for (int base = tid; base < total; base += NT * GRP) {
        int pp[GRP], cc[GRP], iy[GRP], ix[GRP];
        bool in[GRP], st[GRP];
#pragma unroll
        for (int g = 0; g < GRP; ++g) {
            const int idx = base + g * NT;
            st[g] = idx < total;
            const int i2 = st[g] ? idx : 0;
            pp[g] = i2 / cpp; cc[g] = i2 - pp[g] * cpp;
            const int hy = pp[g] / hwd, hx = pp[g] - hy * hwd;
            iy[g] = ty0 + hy - 1; ix[g] = tx0 + hx - 1;
            in[g] = st[g] && iy[g] >= 0 && iy[g] < a.Hi && ix[g] >= 0 && ix[g] < a.Wi;
            if (!in[g]) { iy[g] = 0; ix[g] = 0; }                 // (a valid address: the loads below are unconditional)
        }
        u32x4 r[GRP];
        if (a.ups) {
            Tap ty[GRP], tx[GRP];
            u32x4 v[GRP][4];
#pragma unroll
            for (int g = 0; g < GRP; ++g) {
                ty[g] = linear_tap(iy[g], a.usy, a.Hs, true); tx[g] = linear_tap(ix[g], a.usx, a.Ws, true);
                const T* r0 = img + (long)ty[g].i0 * a.Ws * a.C + cc[g] * CE;
                const T* r1 = img + (long)ty[g].i1 * a.Ws * a.C + cc[g] * CE;
                v[g][0] = *(const u32x4*)(r0 + tx[g].i0 * a.C); v[g][1] = *(const u32x4*)(r0 + tx[g].i1 * a.C);
                v[g][2] = *(const u32x4*)(r1 + tx[g].i0 * a.C); v[g][3] = *(const u32x4*)(r1 + tx[g].i1 * a.C);
            }
#pragma unroll
            for (int g = 0; g < GRP; ++g) r[g] = lerp_chunk(v[g][0], v[g][1], v[g][2], v[g][3], tx[g], ty[g], T());
        } else {
#pragma unroll
            for (int g = 0; g < GRP; ++g) r[g] = *(const u32x4*)(img + ((long)iy[g] * a.Wi + ix[g]) * a.C + cc[g] * CE);
        }
#pragma unroll
        for (int g = 0; g < GRP; ++g)
            if (st[g]) halo[slot(pp[g], cc[g])] = fin(in[g] ? r[g] : (u32x4){0u, 0u, 0u, 0u});
    }
}

__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = pk_bf16(v[0], v[1]);
    t.y = pk_bf16(v[2], v[3]);
    *(uint2*)p = t;
}
// four floats -> the 8-byte hi piece and the 8-byte lo piece of their unit (hi = RNE bf16, lo = RNE bf16 of the exact remainder)
__device__ __forceinline__ void bx3_split4(const f32x4& x, uint2& hi, uint2& lo) {
    hi.x = pk_bf16(x[0], x[1]); hi.y = pk_bf16(x[2], x[3]);
    const float r0 = x[0] - __uint_as_float(hi.x << 16), r1 = x[1] - __uint_as_float(hi.x & 0xffff0000u);
    const float r2 = x[2] - __uint_as_float(hi.y << 16), r3 = x[3] - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pk_bf16(r0, r1); lo.y = pk_bf16(r2, r3);
}

// four floats -> four e4m3 bytes (v_cvt_pk_fp8_f32, round-to-nearest-even), saturating at +-448
__device__ __forceinline__ uint32_t pk_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -FP8_MAX), FP8_MAX); b = fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
    c = fminf(fmaxf(c, -FP8_MAX), FP8_MAX); d = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}
__device__ __forceinline__ void load4(const fp8_t* p, float v[4]) {
    uint32_t t = *(const uint32_t*)p;
    v[0] = e4m32f((fp8_t)(t & 255u)); v[1] = e4m32f((fp8_t)((t >> 8) & 255u));
    v[2] = e4m32f((fp8_t)((t >> 16) & 255u)); v[3] = e4m32f((fp8_t)(t >> 24));
}
__device__ __forceinline__ void store4(fp8_t* p, const float v[4]) { *(uint32_t*)p = pk_fp8x4(v[0], v[1], v[2], v[3]); }

// bf16x3 output (the A operand of a following bf16x3 linear, pre-split so that it can travel by LDS-DMA): hi / lo pieces of the unit
__device__ __forceinline__ void store4(bx3_t* p, const float v[4]) {
    const uintptr_t a = (uintptr_t)p;
    uint2* hi = (uint2*)((a & ~(uintptr_t)31) + ((a >> 4) & 1) * 8);
    uint2 h, l;
    bx3_split4((f32x4){v[0], v[1], v[2], v[3]}, h, l);
    hi[0] = h; hi[2] = l;
}
__device__ __forceinline__ void load4(const bx3_t* p, float v[4]) {
    const uintptr_t a = (uintptr_t)p;
    const uint2* hi = (const uint2*)((a & ~(uintptr_t)31) + ((a >> 4) & 1) * 8);
    const uint2 h = hi[0], l = hi[2];
    v[0] = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16); v[1] = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
    v[2] = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16); v[3] = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
}

template <typename OT> __device__ __forceinline__ OT cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ fp8_t cvt_out<fp8_t>(float v) { return (fp8_t)(pk_fp8x4(v, 0.f, 0.f, 0.f) & 255u); }
template <> __device__ __forceinline__ bx3_t cvt_out<bx3_t>(float v) { return bx3_t{__float_as_uint(v)}; }     // (never stored element-wise: MAP_QKV outputs stay fp32)

// element offset of (row m, column n0) in the output -- and residual -- layout: plain / remapped rows, or the ConvTranspose(k = s)
// pixel shuffle
__device__ __forceinline__ long epi_out_offset(const GemmEpi& e, int m, int n0) {
    if (e.map == MAP_SHUFFLE) {
        int x = m % e.gw, y = (m / e.gw) % e.gh, b = m / (e.gw * e.gh);
        int tap = n0 / e.cout, co = n0 - tap * e.cout;
        int ky = tap / e.ks, kx = tap - ky * e.ks;
        return (((long)b * e.gh * e.ks + (long)y * e.ks + ky) * ((long)e.gw * e.ks) + (long)x * e.ks + kx) * e.cout + co;
    }
    long row = m;
    if (e.rows_per_img) row = (long)(m / e.rows_per_img) * e.img_rows + (m % e.rows_per_img) + e.row_off;
    return row * e.ldc + n0;
}

// The first residual of (m, n0), requested AHEAD of the epilogue.  The output may alias the residual (the ViT residual stream is
// updated in place) and, for all the compiler knows, the bias / scale vectors, so inside epilogue4 every load stays behind the
// previous fragment's store: a wave's FM x FN fragments become as many dependent load -> store round trips (measured in
// conv3_wide_kernel: 17.9 us per tile against 3.4 with all requests first).  Kernels call epi_res1_load / epi_cols_load for all
// their fragments first, then epilogue_dispatch(..., pre, cols).
template <typename OT>
__device__ __forceinline__ void epi_res1_load4(const GemmEpi& e, int m, int n0, float r[4]) {
    const long off = epi_out_offset(e, m, n0);
    const long roff = e.res1_mod ? ((long)(m % e.res1_mod) + e.res1_off) * e.ldc + n0 : off;
    load4((const OT*)e.res1 + roff, r);
}
template <typename T>
__device__ __forceinline__ void epi_res1_load(const GemmEpi& e, int m, int n0, float r[4]) {
    if (e.out_type == OUT_F32) epi_res1_load4<float>(e, m, n0, r);
    else if (e.out_type == OUT_BF16) epi_res1_load4<bf16_t>(e, m, n0, r);
    else if constexpr (std::is_same<T, bx3_t>::value) {
        if (e.out_type == OUT_BX3) epi_res1_load4<bx3_t>(e, m, n0, r);
        else epi_res1_load4<float>(e, m, n0, r);
    }
    else epi_res1_load4<T>(e, m, n0, r);
}
__device__ __forceinline__ bool epi_res1_ahead(const GemmEpi& e) { return e.res1 != nullptr && (e.map == MAP_ROWS || e.map == MAP_SHUFFLE); }

// Column vectors of a lane's n block (bias, LayerScale), loaded once per column fragment ahead of the stores
struct EpiCols { float bias[4], scale[4]; };
__device__ __forceinline__ void epi_cols_load(const GemmEpi& e, int n0, EpiCols& c) {
    if (e.bias) load4(e.bias + n0, c.bias);
    if (e.scale) load4(e.scale + n0, c.scale);
}

// skip_deq: the caller (LN-folded consumer on e4m3 operands) already turned the accumulators into real units;
// pre: the first residual's values if the caller requested them ahead (epi_res1_load), else null; cols: bias / scale likewise
template <typename OT>
__device__ __forceinline__ void epilogue4(const GemmEpi& e, int m, int n0, float v[4], bool skip_deq = false, const float* pre = nullptr,
                                          const EpiCols* cols = nullptr) {
    if (e.deq && !skip_deq) { float q[4]; load4(e.deq + n0, q); v[0] *= q[0]; v[1] *= q[1]; v[2] *= q[2]; v[3] *= q[3]; }   // fp8 operands -> real units
    if (e.bias) {
        if (cols) { v[0] += cols->bias[0]; v[1] += cols->bias[1]; v[2] += cols->bias[2]; v[3] += cols->bias[3]; }
        else { float b[4]; load4(e.bias + n0, b); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
    }
    if (e.act == ACT_GELU) {
        const f32x2v g0 = gelu_erf2((f32x2v){v[0], v[1]}), g1 = gelu_erf2((f32x2v){v[2], v[3]});
        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
    }
    else if (e.act == ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    else if (e.act == ACT_GEGLU) {
        // GEGLU (VDA's FeedForward, motion_module/attention.py:296-384) inside the producing linear: the weight rows are packed
        // interleaved in groups of four -- x0-3 | gate0-3 | x4-7 | gate4-7 ... -- so in gemm_glds_kernel's fragment layout (lane group fg
        // holds columns 16 j + 4 fg .. + 3 of row fr) the gate of this lane's four channels sits in lane ^ 16.  Both lanes share the row
        // and the column guard (N % 8 == 0), so both are here.  The x lanes store x * gelu(gate) at column n0 / 2; the gate lanes are done.
        const float o0 = __shfl_xor(v[0], 16), o1 = __shfl_xor(v[1], 16), o2 = __shfl_xor(v[2], 16), o3 = __shfl_xor(v[3], 16);
        if ((n0 >> 2) & 1) return;
        const f32x2v g0 = gelu_erf2((f32x2v){o0, o1}), g1 = gelu_erf2((f32x2v){o2, o3});
        v[0] *= g0[0]; v[1] *= g0[1]; v[2] *= g1[0]; v[3] *= g1[1];
        n0 = (n0 >> 3) << 2;
    }
    if (e.scale) {
        if (cols) { v[0] *= cols->scale[0]; v[1] *= cols->scale[1]; v[2] *= cols->scale[2]; v[3] *= cols->scale[3]; }
        else { float s[4]; load4(e.scale + n0, s); v[0] *= s[0]; v[1] *= s[1]; v[2] *= s[2]; v[3] *= s[3]; }
    }
    if (e.map == MAP_QKV && n0 >= e.qk_cols) {
        int b = m / e.ntok, t = m - b * e.ntok;
        int c = n0 - e.qk_cols;                       // h*64 + d, 4 consecutive d
        OT* p = (OT*)e.vt + ((long)b * e.heads * 64 + c) * e.npad + t;
        if constexpr (std::is_same<OT, bx3_t>::value) {            // V^T rows in the unit format along the keys
            bx3_store1(p, v[0]); bx3_store1(p + e.npad, v[1]); bx3_store1(p + 2L * e.npad, v[2]); bx3_store1(p + 3L * e.npad, v[3]);
        } else {
            p[0] = cvt_out<OT>(v[0]); p[e.npad] = cvt_out<OT>(v[1]); p[2L * e.npad] = cvt_out<OT>(v[2]); p[3L * e.npad] = cvt_out<OT>(v[3]);
        }
        return;
    }
    // row re-mapping with a negative offset DROPS the rows that would land before their image's block (the folded tap LayerNorm at
    // batch > 1 runs the reassemble projection over ALL token rows and leaves the cls row of every frame out: row_off = -1)
    if (e.rows_per_img && e.row_off < 0 && (m % e.rows_per_img) + e.row_off < 0) return;
    const long off = epi_out_offset(e, m, n0);
    if (e.res1) {
        if (pre) { v[0] += pre[0]; v[1] += pre[1]; v[2] += pre[2]; v[3] += pre[3]; }
        else {
            long roff = e.res1_mod ? ((long)(m % e.res1_mod) + e.res1_off) * e.ldc + n0 : off;
            float r[4]; load4((const OT*)e.res1 + roff, r); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
        }
    }
    if (e.res2) { float r[4]; load4((const OT*)e.res2 + off, r); v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3]; }
    if constexpr (std::is_same<OT, fp8_t>::value) { v[0] *= e.out_qscale; v[1] *= e.out_qscale; v[2] *= e.out_qscale; v[3] *= e.out_qscale; }
    store4((OT*)e.out + off, v);
    if constexpr (std::is_same<OT, float>::value) {            // raw residual copy for the LN-folded consumer: bf16, or e4m3 * scale
        if (e.out2) {
            if (e.out2_qscale > 0.f) {
                float q[4] = {v[0] * e.out2_qscale, v[1] * e.out2_qscale, v[2] * e.out2_qscale, v[3] * e.out2_qscale};
                store4((fp8_t*)e.out2 + off, q);
            } else if (e.out2_bx3) store4((bx3_t*)e.out2 + off, v);
            else store4((bf16_t*)e.out2 + off, v);
        }
    }
}

// out_type -> element type of the output / residuals
template <typename T>
__device__ __forceinline__ void epilogue_dispatch(const GemmEpi& e, int m, int n0, float v[4], bool skip_deq = false, const float* pre = nullptr,
                                                  const EpiCols* cols = nullptr) {
    if (e.out_type == OUT_F32) epilogue4<float>(e, m, n0, v, skip_deq, pre, cols);
    else if (e.out_type == OUT_BF16) epilogue4<bf16_t>(e, m, n0, v, skip_deq, pre, cols);
    else if constexpr (std::is_same<T, bx3_t>::value) {
        if (e.out_type == OUT_BX3) epilogue4<bx3_t>(e, m, n0, v, skip_deq, pre, cols);               // pre-split for the next bf16x3 linear
        else epilogue4<float>(e, m, n0, v, skip_deq, pre, cols);                                     // bf16x3 engines keep fp32 activations
    }
    else epilogue4<T>(e, m, n0, v, skip_deq, pre, cols);
}

// XCD-aware block -> tile map.  Workgroup b is dispatched to XCD b % 8 (observed, used for speed
// only), each XCD has a private 4 MiB L2.  The 8 XCDs form an xn x (8/xn) grid over the tile space;
// XCD (xi, xj) owns a rectangle of tiles and walks it m-slow / n-fast, so the blocks resident on one
// XCD share a few W tiles (L2 hits) and each A tile is fetched from far memory once per XCD.
// xn == 0: plain row-major order.  Returns false for surplus blocks of the padded grid.
__device__ __forceinline__ bool tile_of_block(int bid, int tiles_m, int tiles_n, int xn, int& tm, int& tn) {
    if (xn == 0) { tm = bid / tiles_n; tn = bid - tm * tiles_n; return true; }
    const int xm = 8 / xn;
    const int x = bid & 7, seq = bid >> 3;
    const int xi = x % xn, xj = x / xn;
    const int n0 = (tiles_n * xi) / xn, n1 = (tiles_n * (xi + 1)) / xn;
    const int m0 = (tiles_m * xj) / xm, m1 = (tiles_m * (xj + 1)) / xm;
    const int nl = n1 - n0, ml = m1 - m0;
    if (seq >= nl * ml) return false;
    const int q = seq / nl;
    tm = m0 + q; tn = n0 + (seq - q * nl);
    return true;
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for_impl(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_impl<I + 1, N>(f); }
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }


}  // namespace d2s

namespace d2s {
// pick the XCD grid (xn x 8/xn) for a tiles_m x tiles_n tile space: least padding, W chunk within L2
static inline int pick_xn(int tiles_m, int tiles_n, int BN, int Kpad, size_t es, unsigned& grid) {
    static const int force = getenv("D2S_GEMM_XN") ? atoi(getenv("D2S_GEMM_XN")) : -1;
    long total = (long)tiles_m * tiles_n;
    if (force == 0 || total < 16) { grid = (unsigned)total; return 0; }
    int best = 0; double best_score = 1e30; long best_grid = total;
    for (int xn = 1; xn <= 8; xn *= 2) {
        if (force > 0 && xn != force) continue;
        int xm = 8 / xn;
        long g = 8L * cdiv(tiles_n, xn) * cdiv(tiles_m, xm);
        double score = (double)(g - total) / (double)total;
        double wbytes = (double)cdiv(tiles_n, xn) * BN * Kpad * es;
        if (wbytes > 2.5e6) score += 0.5 * (wbytes / 2.5e6);
        if (score < best_score) { best_score = score; best = xn; best_grid = g; }
    }
    grid = (unsigned)best_grid;
    return best;
}

}  // namespace d2s
