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
    float w1[16], w2[16];  // exp(-i*0.15), exp(-i*0.2)
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

// push_pull_inpaint (:437-506)
__device__ void push_pull(const uint8_t* __restrict__ rgb, const float* __restrict__ dep, const DibrGeom& g,
                          float u, float v, float cdi, float parx, float pary, float sweep_sign, float out[3]) {
    float best[3] = {0.f, 0.f, 0.f}, bw = 0.f, col[3];
    const float sx = parx * g.psx * sweep_sign, sy = pary * g.psx * sweep_sign;      // both use pixel_size.x (:442)
    for (int i = 1; i <= g.search; ++i) {                                             // phase 1 (:445-466)
        float su = u + sx * (float)i, sv = v + sy * (float)i;
        if (oob(su, sv)) continue;
        float sdi = 1.0f - tex_depth(dep, g.H, g.W, su, sv);
        if (sdi > cdi + g.tol) {
            tex_color(rgb, g.H, g.W, su, sv, col);
            float w = g.w1[i] * (1.0f + (sdi - cdi) * 10.0f);
            best[0] += col[0] * w; best[1] += col[1] * w; best[2] += col[2] * w;
            bw += w;
            if (bw > 5.0f) break;
        }
    }
    if (bw < 2.0f) {                                                                  // phase 2 (:469-481)
        for (int i = 1; i <= g.search; ++i) {
            float su = u - sx * (float)i, sv = v - sy * (float)i;
            if (oob(su, sv)) continue;
            float sdi = 1.0f - tex_depth(dep, g.H, g.W, su, sv);
            if (sdi > cdi + g.tol) {
                tex_color(rgb, g.H, g.W, su, sv, col);
                float w = g.w2[i];
                best[0] += col[0] * w; best[1] += col[1] * w; best[2] += col[2] * w;
                bw += w;
            }
        }
    }
    if (bw > 0.01f) {                                                                 // phase 3 (:484-502)
        float va[3] = {best[0] / bw * 0.5f, best[1] / bw * 0.5f, best[2] / bw * 0.5f}, vw = 0.5f;
#pragma unroll
        for (int dy = -1; dy <= 1; dy += 2) {
            float vv = v + (float)dy * g.psy * g.blur;
            if (vv >= 0.f && vv <= 1.f) {
                float vdi = 1.0f - tex_depth(dep, g.H, g.W, u, vv);
                if (vdi > cdi + g.tol * 0.5f) {
                    tex_color(rgb, g.H, g.W, u, vv, col);
                    va[0] += col[0] * 0.25f; va[1] += col[1] * 0.25f; va[2] += col[2] * 0.25f;
                    vw += 0.25f;
                }
            }
        }
        out[0] = va[0] / vw; out[1] = va[1] / vw; out[2] = va[2] / vw;
        return;
    }
    tex_color(rgb, g.H, g.W, u, v, out);                                              // :505
}

// one output pixel of one eye: FRAGMENT_SHADER.main (:533-631) -> colour * alpha
__device__ __forceinline__ void dibr_pixel(const uint8_t* __restrict__ rgb, const float* __restrict__ dep, const DibrGeom& g,
                                           int x, int y, int eye, float outc[4]) {
    const float eye_offset = eye ? g.half_ipd : -g.half_ipd;                          // :2701, 2714
    const float sg = eye_offset > 0.f ? 1.f : (eye_offset < 0.f ? -1.f : 0.f);
    const float parx = g.c * sg, pary = g.s * sg;                                     // :540
    const float sweep_sign = eye_offset > 0.f ? -1.f : 1.f;                           // :541
    const float u = ((float)x + 0.5f) / (float)g.ow, v = ((float)y + 0.5f) / (float)g.oh;
    // 3-tap depth smoothing along the parallax direction (:545-549)
    const float dsx = parx * g.psx * 1.5f, dsy = pary * g.psy * 1.5f;
    float d0 = tex_depth(dep, g.H, g.W, u, v);
    float dm = tex_depth(dep, g.H, g.W, u - dsx, v - dsy);
    float dp = tex_depth(dep, g.H, g.W, u + dsx, v + dsy);
    float d = d0 * 0.7f + dm * 0.15f + dp * 0.15f;
    float dinv = -d;
    float shaped = dinv * (1.0f + 0.35f * (1.0f - d));                                // :554
    float shift = shaped + g.conv;
    float fall = 1.f;                                                                 // :560-562 (exactly 1 away from the edges)
    if (u < 0.05f || u > 0.95f) fall = SMOOTHSTEP_C(0.f, 0.05f, u) * SMOOTHSTEP_C(1.f, 0.95f, u);
    float px = eye_offset * shift * g.strength * fall;                                // :563
    float su = u - px * g.c, sv = v - px * g.s;                                       // :564
    float conf;                                                                        // :419-435
    if (su < 0.f || su > 1.f || sv < 0.f || sv > 1.f) conf = 1.f;
    else {
        const float s2x = parx * g.psx * 2.0f, s2y = pary * g.psy * 2.0f;
        float jump = fabsf(tex_depth(dep, g.H, g.W, u - s2x, v - s2y) - tex_depth(dep, g.H, g.W, u + s2x, v + s2y));
        conf = SMOOTHSTEP_C(0.04f, 0.10f, jump);
    }
    float col[3];
    tex_color(rgb, g.H, g.W, su, sv, col);                                            // :570
    if (conf > 0.001f) {
        float fill[3];
        push_pull(rgb, dep, g, u, v, dinv, parx, pary, sweep_sign, fill);
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = col[k] * (1.0f - conf) + fill[k] * conf; // mix (:575)
    }
    float alpha = 1.f;                                                                // :582 (exactly 1 inside the frame)
    if (su < 0.001f || su > 0.999f || sv < 0.001f || sv > 0.999f) {
        float bx = SMOOTHSTEP_C(-0.001f, 0.001f, su) * SMOOTHSTEP_C(1.001f, 0.999f, su);
        float by = SMOOTHSTEP_C(-0.001f, 0.001f, sv) * SMOOTHSTEP_C(1.001f, 0.999f, sv);
        alpha = fminf(bx, by);
    }
    if (g.feather) {                                                                   // :587-616
        // (gl_FragCoord.xy - u_viewport.xy) / u_viewport.zw; gl_FragCoord is y-up, pixel centres at +0.5
        float fu = (((float)x + 0.5f) - g.vpx) / g.vpw, fv = (((float)g.oh - ((float)y + 0.5f)) - g.vpy) / g.vph, fw = g.feather_w;
        float fo = smoothstepf(0.f, fw, fu) * smoothstepf(0.f, fw, 1.0f - fu) * smoothstepf(0.f, fw, fv) * smoothstepf(0.f, fw, 1.0f - fv);
        float sh = powf(fo, 0.7f);
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] *= sh;
    }
    if (g.corner_r > 0.f) {
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
}

// one thread = one output pixel of one eye.  (4 pixels per thread with packed dword stores measured SLOWER -- 113 -> 120 us
// Full-SBS, 38 -> 97 us Half-SBS at 1080p: the kernel lives on the locality of neighbouring threads' gathers, not on
// its stores.)
template <int OUT_FMT>
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
    dibr_pixel(rgb, dep, g, x, y, eye, c);
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
    for (int i = 0; i < 16; ++i) { g.w1[i] = expf((float)(-i * 0.15)); g.w2[i] = expf((float)(-i * 0.2)); }
    D2S_REQUIRE(2 * g.oh <= 65535 && batch <= 65535, "frame / batch too large for one launch");
    dim3 grid(cdiv(g.ow, 256), 2 * g.oh, batch), block(256);
    if (out_fmt == D2S_FMT_U8_HWC)
        hipLaunchKernelGGL(dibr_kernel<D2S_FMT_U8_HWC>, grid, block, 0, (hipStream_t)stream, rgb, depth, out, g);
    else
        hipLaunchKernelGGL(dibr_kernel<D2S_FMT_F32_HWC>, grid, block, 0, (hipStream_t)stream, rgb, depth, out, g);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
