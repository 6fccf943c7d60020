// f1: the reference's GLSL DIBR fragment shader with disocclusion in-painting (viewer.py:386-631; the warp its default
// Viewer / OpenXR modes show) as a HIP kernel: 3-tap depth smoothing along the parallax direction, non-linear depth
// shaping, 5 % edge fall-off, soft disocclusion confidence, 12-tap directional push-pull + opposite sweep + 3-tap
// vertical blur, sub-pixel border alpha, optional edge feathering.  One thread = one output pixel of one eye.
//
// texture() is exact float32 GL_LINEAR filtering with texel centres at (i+0.5)/N and GL_REPEAT wrapping (moderngl's
// defaults; the reference sets neither, viewer.py:2385-2386).  u_resolution is never assigned in the reference
// (viewer.py:395, 413: pixel_size = 1/0), so the resolution is a parameter here (0 -> source size).  Colours are kept
// in 0..255; frag_color.a follows d2s_dibr_params.alpha_mode (the reference draws these quads with blending off: WINDOW = rgb as written).
// Line numbers in the comments below are viewer.py.
#include "common.h"
#include <math.h>
#include <algorithm>

namespace d2s {

struct DibrGeom {
    int H, W;              // source frame == depth size
    int oh, ow;            // per-eye viewport
    int mode;              // D2S_MODE_*: where the two eyes land in the output
    int out_h, out_w;      // packed output
    float c, s;            // cos / sin(u_roll)
    float psx, psy;        // pixel_size
    float half_ipd, strength, conv;
    float tol, blur, feather_w;
    int search, feather;
    int alpha_mode;        // D2S_DIBR_ALPHA_*
    float corner_r, vpx, vpy, vpw, vph;             // u_corner_radius; u_viewport in eye-image pixels (y up)
    float w1[20], w2[20];  // exp(-i*0.15), exp(-i*0.2), i < 16 (the sweeps index in groups of four: up to [16..18], never used)
};

// GL_REPEAT index: one conditional add / subtract covers every coordinate within one period of the texture (all but
// absurd parallax settings); the integer modulo (~25 instructions on this ISA) is the fallback
__device__ __forceinline__ int wrapi(int i, int n) {
    if (i < 0) i += n; else if (i >= n) i -= n;
    if ((unsigned)i >= (unsigned)n) { i %= n; if (i < 0) i += n; }
    return i;
}

struct TexTap { int x0, x1, y0, y1; float fx, fy; };
__device__ __forceinline__ TexTap tex_tap(float u, float v, int H, int W) {
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    TexTap t;
    t.fx = x - x0f; t.fy = y - y0f;
    t.x0 = wrapi((int)x0f, W); t.y0 = wrapi((int)y0f, H);
    t.x1 = t.x0 + 1 == W ? 0 : t.x0 + 1;
    t.y1 = t.y0 + 1 == H ? 0 : t.y0 + 1;
    return t;
}
__device__ __forceinline__ float lerp2(float a, float b, float c, float d, float fx, float fy) {
    float top = a + (b - a) * fx, bot = c + (d - c) * fx;
    return top + (bot - top) * fy;
}
// The two texels of a row are adjacent except across the GL_REPEAT seam: one 8-byte load per row (gfx950 runs with
// unaligned access enabled: a dwordx2 at a 4-byte / a byte address is one instruction) instead of two 4-byte / six
// 1-byte loads -- the kernel is bound by the number of gather instructions, not by bytes.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float tex_depth(const float* __restrict__ dep, int H, int W, float u, float v) {
    TexTap t = tex_tap(u, v, H, W);
    const float* r0 = dep + t.y0 * W;                // (H * W < 2^31 / 3 is checked by the launcher: 32-bit texel indices)
    const float* r1 = dep + t.y1 * W;
    if (t.x1 == t.x0 + 1) {
        f32x2u a = *(const f32x2u*)(r0 + t.x0), b = *(const f32x2u*)(r1 + t.x0);
        return lerp2(a.x, a.y, b.x, b.y, t.fx, t.fy);
    }
    return lerp2(r0[t.x0], r0[t.x1], r1[t.x0], r1[t.x1], t.fx, t.fy);
}
__device__ __forceinline__ void tex_color(const uint8_t* __restrict__ rgb, int H, int W, float u, float v, float o[3]) {
    TexTap t = tex_tap(u, v, H, W);
    const int ia = (t.y0 * W + t.x0) * 3, ic = (t.y1 * W + t.x0) * 3, end = H * W * 3;
    if (t.x1 == t.x0 + 1 && ia + 8 <= end && ic + 8 <= end) {          // (the 8-byte window must stay inside the frame)
        uint2 p, q;
        __builtin_memcpy(&p, rgb + ia, 8);
        __builtin_memcpy(&q, rgb + ic, 8);
        const float a[3] = {(float)(p.x & 255u), (float)((p.x >> 8) & 255u), (float)((p.x >> 16) & 255u)};
        const float b[3] = {(float)(p.x >> 24), (float)(p.y & 255u), (float)((p.y >> 8) & 255u)};
        const float c[3] = {(float)(q.x & 255u), (float)((q.x >> 8) & 255u), (float)((q.x >> 16) & 255u)};
        const float d[3] = {(float)(q.x >> 24), (float)(q.y & 255u), (float)((q.y >> 8) & 255u)};
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = lerp2(a[k], b[k], c[k], d[k], t.fx, t.fy);
        return;
    }
    const uint8_t* a = rgb + ia;
    const uint8_t* b = rgb + (t.y0 * W + t.x1) * 3;
    const uint8_t* c = rgb + ic;
    const uint8_t* d = rgb + (t.y1 * W + t.x1) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = lerp2((float)a[k], (float)b[k], (float)c[k], (float)d[k], t.fx, t.fy);
}
// roll == 0 (the desktop viewer; OpenXR sets a roll): every tap of a pixel except the two vertical-blur taps lies on the pixel's own
// texture row pair, so the y half of tex_tap -- v * H - 0.5, floor, GL_REPEAT wrap, the two row bases -- is formed ONCE per pixel
// (same expressions on the same v: the same bits as a per-tap evaluation) and a tap is its x half alone.
struct RowCtx { const float* d0; const float* d1; int c0, c1; float fy; };
__device__ __forceinline__ RowCtx row_ctx(const float* __restrict__ dep, int H, int W, float v) {
    const float y = v * (float)H - 0.5f, y0f = floorf(y);
    RowCtx r;
    r.fy = y - y0f;
    const int y0 = wrapi((int)y0f, H), y1 = y0 + 1 == H ? 0 : y0 + 1;
    r.d0 = dep + y0 * W; r.d1 = dep + y1 * W;
    r.c0 = y0 * W * 3; r.c1 = y1 * W * 3;
    return r;
}
struct XTap { int x0, x1; float fx; };
__device__ __forceinline__ XTap x_tap(float u, int W) {
    const float x = u * (float)W - 0.5f, x0f = floorf(x);
    XTap t;
    t.fx = x - x0f;
    t.x0 = wrapi((int)x0f, W);
    t.x1 = t.x0 + 1 == W ? 0 : t.x0 + 1;
    return t;
}
__device__ __forceinline__ float tex_depth_row(const RowCtx& r, int W, float u) {
    const XTap t = x_tap(u, W);
    if (t.x1 == t.x0 + 1) {
        f32x2u a = *(const f32x2u*)(r.d0 + t.x0), b = *(const f32x2u*)(r.d1 + t.x0);
        return lerp2(a.x, a.y, b.x, b.y, t.fx, r.fy);
    }
    return lerp2(r.d0[t.x0], r.d0[t.x1], r.d1[t.x0], r.d1[t.x1], t.fx, r.fy);
}
__device__ __forceinline__ void tex_color_row(const uint8_t* __restrict__ rgb, const RowCtx& r, int H, int W, float u, float o[3]) {
    const XTap t = x_tap(u, W);
    const int ia = r.c0 + t.x0 * 3, ic = r.c1 + t.x0 * 3, end = H * W * 3;
    if (t.x1 == t.x0 + 1 && ia + 8 <= end && ic + 8 <= end) {
        uint2 p, q;
        __builtin_memcpy(&p, rgb + ia, 8);
        __builtin_memcpy(&q, rgb + ic, 8);
        const float a[3] = {(float)(p.x & 255u), (float)((p.x >> 8) & 255u), (float)((p.x >> 16) & 255u)};
        const float b[3] = {(float)(p.x >> 24), (float)(p.y & 255u), (float)((p.y >> 8) & 255u)};
        const float c[3] = {(float)(q.x & 255u), (float)((q.x >> 8) & 255u), (float)((q.x >> 16) & 255u)};
        const float d[3] = {(float)(q.x >> 24), (float)(q.y & 255u), (float)((q.y >> 8) & 255u)};
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = lerp2(a[k], b[k], c[k], d[k], t.fx, r.fy);
        return;
    }
    const uint8_t* a = rgb + ia;
    const uint8_t* b = rgb + r.c0 + t.x1 * 3;
    const uint8_t* c = rgb + ic;
    const uint8_t* d = rgb + r.c1 + t.x1 * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = lerp2((float)a[k], (float)b[k], (float)c[k], (float)d[k], t.fx, r.fy);
}
// Where a pixel's taps come from.  own_*: taps on the pixel's own texture row pair when roll == 0 (any (u, v) otherwise);
// any_*: the two vertical-blur taps of the in-painting (other rows), always the general gather.
struct GenSmp {                                    // general: every tap evaluates both coordinates (roll != 0)
    const uint8_t* rgb; const float* dep; int H, W;
    __device__ __forceinline__ float own_depth(float u, float v) const { return tex_depth(dep, H, W, u, v); }
    __device__ __forceinline__ void own_color(float u, float v, float o[3]) const { tex_color(rgb, H, W, u, v, o); }
    __device__ __forceinline__ float any_depth(float u, float v) const { return tex_depth(dep, H, W, u, v); }
    __device__ __forceinline__ void any_color(float u, float v, float o[3]) const { tex_color(rgb, H, W, u, v, o); }
};
struct RowSmp : GenSmp {                           // roll == 0: the row pair is formed once per pixel
    RowCtx rc;
    __device__ __forceinline__ float own_depth(float u, float) const { return tex_depth_row(rc, W, u); }
    __device__ __forceinline__ void own_color(float u, float, float o[3]) const { tex_color_row(rgb, rc, H, W, u, o); }
};
// roll == 0, the block's row pair staged in LDS: a window of WW texels starting at (unwrapped) texel wx0, GL_REPEAT applied by
// the staging loop; planes d0 | d1 | R0 G0 B0 | R1 G1 B1 as floats (the same byte -> float conversions the gather path makes per
// tap).  A tap is index arithmetic + ds_read2_b32 pairs; taps that leave the window (parallax settings beyond the margin the
// launcher sized it for) take the row gather: same values either way.
struct WinSmp : RowSmp {
    const float* dwin;          // [2][WW]: the row pair of the depth texture
    const float* cwin;          // [6][WW]: R0 G0 B0 R1 G1 B1 as floats
    int wx0, WW;
    __device__ __forceinline__ float own_depth(float u, float v) const {
        const float x = u * (float)W - 0.5f, x0f = floorf(x), fx = x - x0f;
        const int j = (int)x0f - wx0;
        if ((unsigned)j < (unsigned)(WW - 1)) {
            const float* p = dwin + j;
            return lerp2(p[0], p[1], p[WW], p[WW + 1], fx, rc.fy);
        }
        return RowSmp::own_depth(u, v);
    }
    __device__ __forceinline__ void own_color(float u, float v, float o[3]) const {
        const float x = u * (float)W - 0.5f, x0f = floorf(x), fx = x - x0f;
        const int j = (int)x0f - wx0;
        if ((unsigned)j < (unsigned)(WW - 1)) {
            const float* p = cwin + j;
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = lerp2(p[k * WW], p[k * WW + 1], p[(3 + k) * WW], p[(3 + k) * WW + 1], fx, rc.fy);
            return;
        }
        RowSmp::own_color(u, v, o);
    }
};
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x) {
    float t = fminf(fmaxf((x - e0) / (e1 - e0), 0.f), 1.f);
    return t * t * (3.f - 2.f * t);
}
// constant edges: 1 / (e1 - e0) is a literal (an IEEE division is ~20 instructions here and the kernel is VALU-bound:
// profiles/r1_09: 610 VALU instructions per pixel before, 7 of these per pixel)
#define SMOOTHSTEP_C(E0, E1, X) smoothstep_inv((E0), (float)(1.0 / ((double)(E1) - (double)(E0))), (X))
__device__ __forceinline__ float smoothstep_inv(float e0, float inv, float x) {
    float t = fminf(fmaxf((x - e0) * inv, 0.f), 1.f);
    return t * t * (3.f - 2.f * t);
}
__device__ __forceinline__ bool oob(float u, float v) { return u < 0.f || v < 0.f || u > 1.f || v > 1.f; }

__device__ __forceinline__ float g_w_phase1(float w1i, float sdi, float cdi) { return w1i * (1.0f + (sdi - cdi) * 10.0f); }   // :459
// phase 3 of push_pull_inpaint (:484-505): normalise + 3-tap vertical blur, or the pixel's own colour when nothing was found
template <class S>
__device__ __forceinline__ void push_pull_finish(const S& smp, const DibrGeom& g, float u, float v, float cdi, const float best[3], float bw, float out[3]) {
    if (bw > 0.01f) {                                                                 // phase 3 (:484-502)
        float va[3] = {best[0] / bw * 0.5f, best[1] / bw * 0.5f, best[2] / bw * 0.5f}, vw = 0.5f;
        // the two vertical taps touch other texture rows: global gathers.  All six of their loads are requested before the first is
        // used (the depth test decides what is ADDED, not what is fetched): one round trip instead of up to four dependent ones at
        // the end of every in-painted pixel; the sums keep the order dy = -1, +1
        float vdi[2], vc[2][3];
        bool ok[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float vv = v + (float)(2 * t - 1) * g.psy * g.blur;
            ok[t] = vv >= 0.f && vv <= 1.f;
            const float vs = ok[t] ? vv : v;                                          // (a valid row for the unconditional loads)
            vdi[t] = 1.0f - smp.any_depth(u, vs);
            smp.any_color(u, vs, vc[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (ok[t] && vdi[t] > cdi + g.tol * 0.5f) {
                va[0] += vc[t][0] * 0.25f; va[1] += vc[t][1] * 0.25f; va[2] += vc[t][2] * 0.25f;
                vw += 0.25f;
            }
        out[0] = va[0] / vw; out[1] = va[1] / vw; out[2] = va[2] / vw;
        return;
    }
    smp.own_color(u, v, out);                                                         // :505
}

// push_pull_inpaint (:437-506)
template <class S>
__device__ void push_pull(const S& smp, const DibrGeom& g,
                          float u, float v, float cdi, float parx, float pary, float sweep_sign, float out[3]) {
    // (roll == 0: sy == 0, every sweep tap lies on the pixel's own row pair -- the samplers' own_* taps)
    auto depth_at = [&](float su, float sv) { return smp.own_depth(su, sv); };
    auto color_at = [&](float su, float sv, float* o) { smp.own_color(su, sv, o); };
    float best[3] = {0.f, 0.f, 0.f}, bw = 0.f, col[3];
    const float sx = parx * g.psx * sweep_sign, sy = pary * g.psx * sweep_sign;      // both use pixel_size.x (:442)
    // (Requesting the sweep's depth taps four at a time before testing them -- same values, same order of the sums -- was built and
    //  measured: no faster in the row kernel, 5-9 % slower in the gather kernels, and 30 more VGPRs took the row kernel from 7 to 4
    //  waves per SIMD.  The loops stay as the shader writes them.)
    for (int i = 1; i <= g.search; ++i) {                                             // phase 1 (:445-466)
        float su = u + sx * (float)i, sv = v + sy * (float)i;
        if (oob(su, sv)) continue;
        float sdi = 1.0f - depth_at(su, sv);
        if (sdi > cdi + g.tol) {
            color_at(su, sv, col);
            float w = g_w_phase1(g.w1[i], sdi, cdi);
            best[0] += col[0] * w; best[1] += col[1] * w; best[2] += col[2] * w;
            bw += w;
            if (bw > 5.0f) break;
        }
    }
    if (bw < 2.0f) {                                                                  // phase 2 (:469-481)
        for (int i = 1; i <= g.search; ++i) {
            float su = u - sx * (float)i, sv = v - sy * (float)i;
            if (oob(su, sv)) continue;
            float sdi = 1.0f - depth_at(su, sv);
            if (sdi > cdi + g.tol) {
                color_at(su, sv, col);
                float w = g.w2[i];
                best[0] += col[0] * w; best[1] += col[1] * w; best[2] += col[2] * w;
                bw += w;
            }
        }
    }
    push_pull_finish(smp, g, u, v, cdi, best, bw, out);
}

// roll == 0: the five depth taps of a pixel that do not depend on its shift -- the centre, the smoothing pair at -+1.5 pixel_size
// and the confidence pair at -+2 pixel_size along the parallax direction -- are the SAME texture positions for the two eyes:
// sg(right) = -sg(left), and a negation is exact through (c * sg) * pixel_size * k, so u - dsx(right) is bit for bit u + dsx(left).
// They are evaluated once per output column with the left eye's offsets and handed to both eyes (dm / dp change places for the right
// eye, |a - b| == |b - a|): 5 taps instead of 10 per column of the row kernel, the same bits.
struct PixTaps { float d0, dA, dB, jA, jB; };      // depth at u, u - dsx(left), u + dsx(left), u - s2x(left), u + s2x(left)
template <class S>
__device__ __forceinline__ PixTaps pix_taps(const S& smp, const DibrGeom& g, int x, int y) {
    const float eye_offset = -g.half_ipd;
    const float sg = eye_offset > 0.f ? 1.f : (eye_offset < 0.f ? -1.f : 0.f);
    const float parx = g.c * sg;
    const float u = ((float)x + 0.5f) / (float)g.ow, v = ((float)y + 0.5f) / (float)g.oh;
    const float dsx = parx * g.psx * 1.5f, s2x = parx * g.psx * 2.0f;
    // (One window test for the five taps instead of a branch pair per tap: measured, no faster, 12 bytes of scratch.)
    PixTaps t;
    t.d0 = smp.own_depth(u, v);
    t.dA = smp.own_depth(u - dsx, v);
    t.dB = smp.own_depth(u + dsx, v);
    t.jA = smp.own_depth(u - s2x, v);
    t.jB = smp.own_depth(u + s2x, v);
    return t;
}

// one output pixel of one eye: FRAGMENT_SHADER.main (:533-631) -> colour * alpha
// DEFER: return true WITHOUT a result when the pixel needs the in-painting (the caller queues it for a second, lane-compacted pass
// that calls this function again with DEFER = false: the same expressions on the same inputs -> the same bits).
// sh: the column's five shift-independent depth taps (roll == 0 only, pix_taps) or nullptr = take them here.
// FX = false: u_feather_enabled == 0 and u_corner_radius == 0 (the desktop viewer's state) as a compile-time fact.  As run-time branches
// the two blocks depend on the column and row only, so the compiler hoists them out of the eye loop and -- free of side effects --
// speculates them: every pixel paid for four IEEE divisions, a powf and a sqrtf it did not use (~250 of ~900 instructions of pass 1).
template <bool DEFER = false, bool SHARED = false, bool FX = true, class S>
__device__ __forceinline__ bool dibr_pixel(const S& smp, const DibrGeom& g, int x, int y, int eye, float outc[4], const PixTaps* sh = nullptr) {
    const float eye_offset = eye ? g.half_ipd : -g.half_ipd;                          // :2701, 2714
    const float sg = eye_offset > 0.f ? 1.f : (eye_offset < 0.f ? -1.f : 0.f);
    const float parx = g.c * sg, pary = g.s * sg;                                     // :540
    const float sweep_sign = eye_offset > 0.f ? -1.f : 1.f;                           // :541
    const float u = ((float)x + 0.5f) / (float)g.ow, v = ((float)y + 0.5f) / (float)g.oh;
    auto depth_at = [&](float su, float sv) { return smp.own_depth(su, sv); };       // (roll == 0: v - 0 * k == v, every tap below shares the row pair)
    // 3-tap depth smoothing along the parallax direction (:545-549)
    const float dsx = parx * g.psx * 1.5f, dsy = pary * g.psy * 1.5f;
    float d0, dm, dp;
    if constexpr (SHARED) { d0 = sh->d0; dm = eye ? sh->dB : sh->dA; dp = eye ? sh->dA : sh->dB; }
    else { d0 = depth_at(u, v); dm = depth_at(u - dsx, v - dsy); dp = depth_at(u + dsx, v + dsy); }
    float d = d0 * 0.7f + dm * 0.15f + dp * 0.15f;
#if defined(DIBR_CUT) && DIBR_CUT == 1      // (tuning aid, timing only: stop after the three smoothing taps)
    outc[0] = outc[1] = outc[2] = d; outc[3] = 1.f; return false;
#endif
    float dinv = -d;
    float shaped = dinv * (1.0f + 0.35f * (1.0f - d));                                // :554
    float shift = shaped + g.conv;
    float fall = 1.f;                                                                 // :560-562 (exactly 1 away from the edges)
    if (u < 0.05f || u > 0.95f) fall = SMOOTHSTEP_C(0.f, 0.05f, u) * SMOOTHSTEP_C(1.f, 0.95f, u);
    float px = eye_offset * shift * g.strength * fall;                                // :563
    float su = u - px * g.c, sv = v - px * g.s;                                       // :564
    float conf;                                                                        // :419-435
#if defined(DIBR_CUT) && DIBR_CUT == 5      // (timing only: no in-painting for pixels whose source lies outside the frame)
    if (su < 0.f || su > 1.f || sv < 0.f || sv > 1.f) conf = 0.f;
#else
    if (su < 0.f || su > 1.f || sv < 0.f || sv > 1.f) conf = 1.f;
#endif
    else {
        const float s2x = parx * g.psx * 2.0f, s2y = pary * g.psy * 2.0f;
        float jump;
        if constexpr (SHARED) jump = eye ? fabsf(sh->jB - sh->jA) : fabsf(sh->jA - sh->jB);
        else jump = fabsf(depth_at(u - s2x, v - s2y) - depth_at(u + s2x, v + s2y));
        conf = SMOOTHSTEP_C(0.04f, 0.10f, jump);
    }
#if defined(DIBR_CUT) && DIBR_CUT == 2      // (timing only: stop after the confidence taps)
    outc[0] = outc[1] = outc[2] = conf + su; outc[3] = 1.f; return false;
#endif
    if (DEFER && conf > 0.001f) return true;
    float col[3];
    smp.own_color(su, sv, col);                                                       // :570
#if defined(DIBR_CUT) && DIBR_CUT == 3      // (timing only: no in-painting)
    outc[0] = col[0] + conf; outc[1] = col[1]; outc[2] = col[2]; outc[3] = 1.f; return false;
#endif
#if defined(DIBR_CUT) && DIBR_CUT == 4      // (timing only: everything but the in-painting call)
    if (conf > 1e30f) {
#else
    if (conf > 0.001f) {
#endif
        float fill[3];
        push_pull(smp, g, u, v, dinv, parx, pary, sweep_sign, fill);
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = col[k] * (1.0f - conf) + fill[k] * conf; // mix (:575)
    }
    float alpha = 1.f;                                                                // :582 (exactly 1 inside the frame)
    if (su < 0.001f || su > 0.999f || sv < 0.001f || sv > 0.999f) {
        float bx = SMOOTHSTEP_C(-0.001f, 0.001f, su) * SMOOTHSTEP_C(1.001f, 0.999f, su);
        float by = SMOOTHSTEP_C(-0.001f, 0.001f, sv) * SMOOTHSTEP_C(1.001f, 0.999f, sv);
        alpha = fminf(bx, by);
    }
    if (FX && g.feather) {                                                             // :587-616
        // (gl_FragCoord.xy - u_viewport.xy) / u_viewport.zw; gl_FragCoord is y-up, pixel centres at +0.5
        float fu = (((float)x + 0.5f) - g.vpx) / g.vpw, fv = (((float)g.oh - ((float)y + 0.5f)) - g.vpy) / g.vph, fw = g.feather_w;
        float fo = smoothstepf(0.f, fw, fu) * smoothstepf(0.f, fw, 1.0f - fu) * smoothstepf(0.f, fw, fv) * smoothstepf(0.f, fw, 1.0f - fv);
        float sh = powf(fo, 0.7f);
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] *= sh;
    }
    if (FX && g.corner_r > 0.f) {
        // rounded-box SDF over the quad's own uv (the shader's inner `uv` of the feather block shadows only that block), :617-624
        const float dx = fabsf(u - 0.5f) - 0.5f + g.corner_r, dy = fabsf(v - 0.5f) - 0.5f + g.corner_r;
        const float mx = fmaxf(dx, 0.f), my = fmaxf(dy, 0.f);
        const float sdf = sqrtf(mx * mx + my * my) + fminf(fmaxf(dx, dy), 0.f) - g.corner_r;
        alpha = fminf(alpha, 1.0f - smoothstepf(0.f, 0.01f, sdf));
    }
    // frag_color = (col, alpha).  The reference's quads are drawn with blending off: its window shows col as written (WINDOW);
    // PREMULTIPLIED = col * alpha (composited over black); RGBA hands out both (d2s.h)
#pragma unroll
    for (int k = 0; k < 3; ++k) outc[k] = g.alpha_mode == D2S_DIBR_ALPHA_PREMULTIPLIED ? col[k] * alpha : col[k];
    outc[3] = alpha;
    return false;
}

template <int OUT_FMT>
__device__ __forceinline__ void dibr_store(void* __restrict__ out_all, long o, int nch, const float c[4]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (OUT_FMT == D2S_FMT_U8_HWC) ((uint8_t*)out_all)[o + k] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(c[k], 0, 0);
        else ((float*)out_all)[o + k] = c[k];
    }
    if (nch == 4) {
        if (OUT_FMT == D2S_FMT_U8_HWC) ((uint8_t*)out_all)[o + 3] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(c[3] * 255.0f, 0, 0);
        else ((float*)out_all)[o + 3] = c[3];
    }
}

// General kernel: one thread = one output pixel of one eye, every tap a global gather.  (4 pixels per thread with packed dword
// stores measured SLOWER -- 113 -> 120 us Full-SBS, 38 -> 97 us Half-SBS at 1080p: this kernel lives on the locality of neighbouring
// threads' gathers, not on its stores.)  ROLL0: the row pair of a pixel formed once (104.5 -> 89.5 us Full-SBS 1080p, same bits).
template <int OUT_FMT, bool ROLL0>
__global__ void __launch_bounds__(256)
dibr_kernel(const uint8_t* __restrict__ rgb_all, const float* __restrict__ dep_all, void* __restrict__ out_all, DibrGeom g) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y % g.oh, eye = blockIdx.y / g.oh, b = blockIdx.z;
    if (x >= g.ow) return;
    const uint8_t* rgb = rgb_all + (long)b * g.H * g.W * 3;
    const float* dep = dep_all + (long)b * g.H * g.W;
    const bool sbs = g.mode == D2S_MODE_HALF_SBS || g.mode == D2S_MODE_FULL_SBS;
    const int ox = sbs ? eye * g.ow + x : x, oy = sbs ? y : eye * g.oh + y;
    const int nch = g.alpha_mode == D2S_DIBR_ALPHA_RGBA ? 4 : 3;
    const long o = (((long)b * g.out_h + oy) * g.out_w + ox) * nch;
    float c[4];
    if constexpr (ROLL0) {
        RowSmp smp;
        smp.rgb = rgb; smp.dep = dep; smp.H = g.H; smp.W = g.W;
        smp.rc = row_ctx(dep, g.H, g.W, ((float)y + 0.5f) / (float)g.oh);     // (the v dibr_pixel forms)
        dibr_pixel(smp, g, x, y, eye, c);
    } else {
        GenSmp smp;
        smp.rgb = rgb; smp.dep = dep; smp.H = g.H; smp.W = g.W;
        dibr_pixel(smp, g, x, y, eye, c);
    }
    dibr_store<OUT_FMT>(out_all, o, nch, c);
}

// roll == 0, the common case (the desktop viewer never rolls): a block = 256 output columns of ONE output row, both eyes.  Every tap
// of those 512 pixels except the in-painting's two vertical-blur taps reads the same two texture rows, within `margin` texels of the
// block's own span: the block stages that window once -- depth rows as they are, colour rows converted to float planes (the same
// conversion a tap makes) -- and the taps become LDS reads at a window index (WinSmp).  Same expressions on the same values as the
// gather kernel: bit-identical (tests/test_gpu_dibr.py).  The gather kernel spent ~600 VALU instructions per pixel, most of them
// address arithmetic of its ~14 eight-byte gathers (64-bit row bases, GL_REPEAT wraps, byte unpacking).
#ifndef DIBR_WAVES
#define DIBR_WAVES 7                  // (<= 72 VGPRs: 5 -> 7 waves per SIMD, 73.8 -> 69.0 us in round 5; 8: no faster.  FX = false: 71 VGPRs, no scratch)
#endif
// COLS (round 6): output columns per block, 256 per thread-pass.  The second pass costs one mostly-empty wave per block that has
// queued pixels whatever their number; a block twice as wide halves those waves (and stages a window 2 x as wide once).
template <int OUT_FMT, bool FX, int COLS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DIBR_WAVES)))
dibr_rows_kernel(const uint8_t* __restrict__ rgb_all, const float* __restrict__ dep_all, void* __restrict__ out_all, DibrGeom g, int margin, int WW) {
    extern __shared__ float dibr_win[];                // [2][WW] the row pair of the depth texture | [6][WW] R0 G0 B0 R1 G1 B1 as floats
    __shared__ int queue[2 * COLS], qn;                // (column - xb) * 2 + eye of the pixels that need the in-painting
    const int tid = threadIdx.x, xb = blockIdx.x * COLS;
    const int y = blockIdx.y, b = blockIdx.z;
    const uint8_t* rgb = rgb_all + (long)b * g.H * g.W * 3;
    const float* dep = dep_all + (long)b * g.H * g.W;
    if (tid == 0) qn = 0;
    WinSmp smp;
    smp.rgb = rgb; smp.dep = dep; smp.H = g.H; smp.W = g.W;
    smp.rc = row_ctx(dep, g.H, g.W, ((float)y + 0.5f) / (float)g.oh);         // block-uniform (the v dibr_pixel forms)
    smp.dwin = dibr_win; smp.cwin = dibr_win + 2 * WW; smp.WW = WW;
    smp.wx0 = (int)floorf((((float)xb + 0.5f) / (float)g.ow) * (float)g.W - 0.5f) - margin;
    // (Four texels per thread -- 16-byte depth loads, 12-byte colour loads, vector LDS writes, a sixth of the load instructions -- was
    //  built and measured: 54.2 us against 51.6 at 1080p Full-SBS; a quarter of the threads then carry the whole round trip.)
#if defined(DIBR_CUT) && (DIBR_CUT == 8 || DIBR_CUT == 10)      // (timing only: no staging, no second pass)
    for (int j = tid; j < 0; j += 256) {
#else
    for (int j = tid; j < WW; j += 256) {
#endif
        const int xs = wrapi(smp.wx0 + j, g.W);        // (one conditional add / subtract unless the window is wider than the texture: then the modulo)
        dibr_win[j] = smp.rc.d0[xs];
        dibr_win[WW + j] = smp.rc.d1[xs];
        const uint8_t* p0 = rgb + smp.rc.c0 + xs * 3;
        const uint8_t* p1 = rgb + smp.rc.c1 + xs * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) { dibr_win[(2 + k) * WW + j] = (float)p0[k]; dibr_win[(5 + k) * WW + j] = (float)p1[k]; }
    }
    __syncthreads();
    const bool sbs = g.mode == D2S_MODE_HALF_SBS || g.mode == D2S_MODE_FULL_SBS;
    const int nch = g.alpha_mode == D2S_DIBR_ALPHA_RGBA ? 4 : 3;
    auto out_index = [&](int px, int eye) {
        const int ox = sbs ? eye * g.ow + px : px, oy = sbs ? y : eye * g.oh + y;
        return (((long)b * g.out_h + oy) * g.out_w + ox) * nch;
    };
#if defined(DIBR_CUT)
    const int x = xb + tid;
#endif
#if defined(DIBR_CUT) && DIBR_CUT == 9      // (timing only: the staging and one dword store per thread)
    if (x < g.ow) ((float*)out_all)[((long)b * g.out_h + y) * g.out_w * nch / 4 + x] = dibr_win[tid] + dibr_win[WW + tid] + dibr_win[2 * WW + tid];
    if (qn >= 0) return;
#endif
#if defined(DIBR_CUT) && (DIBR_CUT == 7 || DIBR_CUT == 10)      // (timing only: the staging and the stores / 10: the stores)
    if (x < g.ow) {
        for (int eye = 0; eye < 2; ++eye) {
            const float c[4] = {dibr_win[tid + eye], dibr_win[WW + tid], dibr_win[2 * WW + tid], 1.f};
            dibr_store<OUT_FMT>(out_all, out_index(x, eye), nch, c);
        }
    }
    if (qn >= 0) return;
#endif
    // pass 1: every pixel up to the in-painting decision.  Disocclusions are thin (0.3-0.5 % of the pixels of a 1080p scene, but a
    // vertical depth edge crosses every row: 5-8 % of the waves): run in place, a wave with three such lanes walks the whole 24-tap
    // sweep at 5 % lane occupancy.  Those pixels are queued instead ...
    for (int cx = tid; cx < COLS; cx += 256) {
        const int x = xb + cx;
        if (x >= g.ow) break;
        const PixTaps taps = pix_taps(smp, g, x, y);
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
            float c[4];
            if (dibr_pixel<true, true, FX>(smp, g, x, y, eye, c, &taps)) queue[atomicAdd(&qn, 1)] = cx * 2 + eye;
            else dibr_store<OUT_FMT>(out_all, out_index(x, eye), nch, c);
        }
    }
    __syncthreads();
#if defined(DIBR_CUT) && (DIBR_CUT == 6 || DIBR_CUT == 8)      // (timing only: no second pass)
    if (qn >= 0) return;
#endif
    // ... and pass 2 gives each queued pixel a lane of its own: the whole pixel function again, in-painting included (same inputs,
    // same expressions: the same bits as the single-pass kernel; the order of the queue does not matter, every entry is independent).
    // Measured at 1080p Full-SBS (tools/dibr_bench.py): in place 85.6 us, queued 73.7; everything but this pass 45 us -- what is left
    // is one mostly-empty wave per block walking ~2 500 instructions.  Round 6 (profiles/r6_06): 50.5 us, 25 of them this pass; a
    // launch-wide queue finished by a second kernel (one lane or sixteen lanes per pixel) was built and lost (31-47 us for that kernel).  Tried on top of it and not kept: queues shared by 2-8 rows with
    // the rows' windows kept in LDS or pass-2 taps by gather (fewer second-pass waves, but 99 VGPRs / 25 KB of LDS per block halve
    // the resident waves: 69-101 us at one frame, 45-98 us Half-SBS), sweep taps requested four at a time (no faster, +30 VGPRs).
    for (int q = tid; q < qn; q += 256) {
        const int e = queue[q], px = xb + (e >> 1), eye = e & 1;
        float c[4];
        dibr_pixel<false, false, FX>(smp, g, px, y, eye, c);
        dibr_store<OUT_FMT>(out_all, out_index(px, eye), nch, c);
    }
}

}  // namespace d2s

using namespace d2s;

extern "C" int d2s_dibr_shape(int H, int W, int display_mode, int* out_h, int* out_w) {
    D2S_REQUIRE(out_h && out_w && H > 0 && W > 0, "bad argument");
    D2S_REQUIRE(display_mode >= D2S_MODE_HALF_SBS && display_mode <= D2S_MODE_FULL_TAB, "bad display_mode");
    *out_h = display_mode == D2S_MODE_FULL_TAB ? 2 * H : (display_mode == D2S_MODE_HALF_TAB ? (H / 2) * 2 : H);
    *out_w = display_mode == D2S_MODE_FULL_SBS ? 2 * W : (display_mode == D2S_MODE_HALF_SBS ? (W / 2) * 2 : W);
    return D2S_OK;
}

extern "C" int d2s_dibr_warp(const uint8_t* rgb, const float* depth, int batch, int H, int W, const d2s_dibr_params* p,
                             void* out, int out_fmt, void* stream) {
    D2S_REQUIRE(rgb && depth && p && out, "null pointer");
    D2S_REQUIRE(p->struct_size == sizeof(d2s_dibr_params),
                "d2s_dibr_params.struct_size must be sizeof(d2s_dibr_params) = 80 (header of d2s_version() >= 110; the 72-byte struct of "
                "version 100 has no alpha_mode)");
    D2S_REQUIRE(batch > 0 && H > 1 && W > 1, "bad shape");
    D2S_REQUIRE((long)H * W * 3 + 8 < (1L << 31), "frame too large (32-bit texel indices)");
    D2S_REQUIRE(out_fmt == D2S_FMT_U8_HWC || out_fmt == D2S_FMT_F32_HWC, "bad out_fmt (U8_HWC or F32_HWC)");
    D2S_REQUIRE(p->search_radius >= 0.f && p->search_radius < 16.f, "search_radius must be in [0,16)");
    DibrGeom g;
    g.H = H; g.W = W; g.mode = p->display_mode;
    int rc = d2s_dibr_shape(H, W, p->display_mode, &g.out_h, &g.out_w);
    if (rc) return rc;
    g.oh = p->display_mode == D2S_MODE_HALF_TAB ? H / 2 : H;
    g.ow = p->display_mode == D2S_MODE_HALF_SBS ? W / 2 : W;
    D2S_REQUIRE(g.oh > 0 && g.ow > 0, "frame too small for a Half mode");
    g.c = cosf(p->roll); g.s = sinf(p->roll);
    g.psx = 1.0f / (p->res_w > 0.f ? p->res_w : (float)W);
    g.psy = 1.0f / (p->res_h > 0.f ? p->res_h : (float)H);
    g.half_ipd = (float)(p->ipd_uv / 2.0);                                            // viewer.py:2701
    g.strength = p->depth_strength; g.conv = p->convergence;
    g.tol = p->depth_tolerance; g.blur = p->blur_radius; g.feather_w = p->feather_width;
    g.search = (int)p->search_radius; g.feather = p->feather_enabled != 0;
    D2S_REQUIRE(p->corner_radius >= 0.f && p->corner_radius <= 0.5f, "corner_radius must be in [0, 0.5]");
    g.corner_r = p->corner_radius;
    D2S_REQUIRE(p->alpha_mode >= D2S_DIBR_ALPHA_WINDOW && p->alpha_mode <= D2S_DIBR_ALPHA_RGBA, "bad alpha_mode");
    g.alpha_mode = p->alpha_mode;
    const bool vp0 = p->viewport[2] == 0.f && p->viewport[3] == 0.f;
    D2S_REQUIRE(vp0 || (p->viewport[2] > 0.f && p->viewport[3] > 0.f), "viewport width / height must be positive (or all zero)");
    g.vpx = vp0 ? 0.f : p->viewport[0]; g.vpy = vp0 ? 0.f : p->viewport[1];
    g.vpw = vp0 ? (float)g.ow : p->viewport[2]; g.vph = vp0 ? (float)g.oh : p->viewport[3];
    for (int i = 0; i < 20; ++i) { g.w1[i] = i < 16 ? expf((float)(-i * 0.15)) : 0.f; g.w2[i] = i < 16 ? expf((float)(-i * 0.2)) : 0.f; }
    D2S_REQUIRE(2 * g.oh <= 65535 && batch <= 65535, "frame / batch too large for one launch");
    dim3 grid(cdiv(g.ow, 256), 2 * g.oh, batch), block(256);
    static EnvInt no_roll0{"D2S_DIBR_NO_ROLL0", 0};        // (A/B aids: the general per-tap evaluation for roll == 0 too;
    static EnvInt no_rows{"D2S_DIBR_NO_ROWS", 0};          //  the gather kernel instead of the LDS-window kernel)
    const bool roll0 = g.s == 0.f && g.c == 1.f && !no_roll0.get();
    // LDS-window kernel: how far from its own texel a pixel's same-row taps can land -- the sweeps (search texels of pixel_size.x),
    // the +-2 pixel_size confidence taps, the parallax shift (|shaped| <= 1 for depth in 0..1) -- in texels of the source
    const double tex_per_px = (double)W * (double)g.psx;
    const double reach = std::max(std::max(2.0, (double)g.search) * tex_per_px,
                                  fabs((double)g.half_ipd) * (1.0 + fabs((double)g.conv)) * fabs((double)g.strength) * (double)W);
    const int margin = (int)ceil(reach) + 2;
    const int WW = (int)ceil(255.0 * (double)W / (double)g.ow) + 2 * margin + 4;
    if (roll0 && !no_rows.get() && WW <= 2048 && g.oh <= 65535) {
        // columns per block: 512 while the window stays <= 640 texels (20 KB of LDS: seven blocks per CU either way).  1080p Full-SBS
        // 50.8 -> 46.8 us (half the second-pass waves); Half-SBS (two source texels per column) keeps 256: 24.4 us against 26.8;
        // 1024 columns: 63.9 us (34 KB per block).  D2S_DIBR_COLS = 256 | 512 | 1024 caps it (A/B aid).
        static EnvInt cols_env{"D2S_DIBR_COLS", 512};
        int cols = cols_env.get() >= 1024 ? 1024 : (cols_env.get() >= 512 ? 512 : 256);
        auto win_words = [&](int c) { return (int)ceil((double)(c - 1) * (double)W / (double)g.ow) + 2 * margin + 4; };
        while (cols > 256 && (win_words(cols) > (cols_env.get() >= 1024 ? 1536 : 640) || g.ow <= cols / 2)) cols >>= 1;   // (1536: 48 KB + the queue stay under 64 KB)
        const int WWc = win_words(cols);
        dim3 rgrid(cdiv(g.ow, cols), g.oh, batch);
        const size_t lds = (size_t)8 * WWc * sizeof(float);
        const bool fx = g.feather || g.corner_r > 0.f;
#define DIBR_ROWS(FMT, FXV, COLS) hipLaunchKernelGGL((dibr_rows_kernel<FMT, FXV, COLS>), rgrid, block, lds, (hipStream_t)stream, rgb, depth, out, g, margin, WWc)
#define DIBR_ROWS_C(FMT, FXV) do { if (cols == 1024) DIBR_ROWS(FMT, FXV, 1024); else if (cols == 512) DIBR_ROWS(FMT, FXV, 512); else DIBR_ROWS(FMT, FXV, 256); } while (0)
        if (out_fmt == D2S_FMT_U8_HWC) { if (fx) DIBR_ROWS_C(D2S_FMT_U8_HWC, true); else DIBR_ROWS_C(D2S_FMT_U8_HWC, false); }
        else { if (fx) DIBR_ROWS_C(D2S_FMT_F32_HWC, true); else DIBR_ROWS_C(D2S_FMT_F32_HWC, false); }
#undef DIBR_ROWS_C
#undef DIBR_ROWS
    } else if (out_fmt == D2S_FMT_U8_HWC) {
        if (roll0) hipLaunchKernelGGL((dibr_kernel<D2S_FMT_U8_HWC, true>), grid, block, 0, (hipStream_t)stream, rgb, depth, out, g);
        else hipLaunchKernelGGL((dibr_kernel<D2S_FMT_U8_HWC, false>), grid, block, 0, (hipStream_t)stream, rgb, depth, out, g);
    } else {
        if (roll0) hipLaunchKernelGGL((dibr_kernel<D2S_FMT_F32_HWC, true>), grid, block, 0, (hipStream_t)stream, rgb, depth, out, g);
        else hipLaunchKernelGGL((dibr_kernel<D2S_FMT_F32_HWC, false>), grid, block, 0, (hipStream_t)stream, rgb, depth, out, g);
    }
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
