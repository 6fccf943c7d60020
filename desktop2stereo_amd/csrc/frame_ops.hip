// Frame-side kernels of the hot path (HBM-bound, no matrix work):
//   preprocess      A2-A4  (reference depth.py:676-706, 1916-1948)
//   upsample_depth  A13    (reference depth.py:1999-2004)
//   stereo_warp     A14    (reference depth.py:2122-2184), with A13 fused when depth comes at model res
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace d2s {

// ------------------------------------------------------------------------------------------------
// pixel fetch in the boundary formats; returns 0..255 float
// ------------------------------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ void load_px(const void* base, long frame_off_px, int H, int W, int y, int x,
                                        float& r, float& g, float& b) {
    if (FMT == D2S_FMT_U8_HWC) {
        const uint8_t* p = (const uint8_t*)base + (frame_off_px + (long)y * W + x) * 3;
        r = (float)p[0]; g = (float)p[1]; b = (float)p[2];
    } else if (FMT == D2S_FMT_U8_CHW) {
        const uint8_t* p = (const uint8_t*)base + frame_off_px * 3 + (long)y * W + x;
        long pl = (long)H * W;
        r = (float)p[0]; g = (float)p[pl]; b = (float)p[2 * pl];
    } else {  // F32_CHW
        const float* p = (const float*)base + frame_off_px * 3 + (long)y * W + x;
        long pl = (long)H * W;
        r = p[0]; g = p[pl]; b = p[2 * pl];
    }
}

// ------------------------------------------------------------------------------------------------
// A2-A4 preprocess: (optional ::stride decimation) -> bilinear(align_corners=False) -> /255 -> norm
// one thread per output pixel, all three channels; output planes [B,3,h,w] float
// ------------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256)
preprocess_kernel(const void* __restrict__ frames, int B, int H, int W, int stride,
                  float* __restrict__ out, int h, int w, float sy, float sx,
                  float m0, float m1, float m2, float is0, float is1, float is2) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * h * w;
    if (idx >= total) return;
    int ox = (int)(idx % w);
    int oy = (int)((idx / w) % h);
    int b = (int)(idx / ((long)w * h));
    int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
    Tap ty = linear_tap(oy, sy, Hs, false);
    Tap tx = linear_tap(ox, sx, Ws, false);
    long fo = (long)b * H * W;
    float a[3], c[3], d[3], e[3];
    load_px<FMT>(frames, fo, H, W, ty.i0 * stride, tx.i0 * stride, a[0], a[1], a[2]);
    load_px<FMT>(frames, fo, H, W, ty.i0 * stride, tx.i1 * stride, c[0], c[1], c[2]);
    load_px<FMT>(frames, fo, H, W, ty.i1 * stride, tx.i0 * stride, d[0], d[1], d[2]);
    load_px<FMT>(frames, fo, H, W, ty.i1 * stride, tx.i1 * stride, e[0], e[1], e[2]);
    const float mean[3] = {m0, m1, m2};
    const float istd[3] = {is0, is1, is2};
    long plane = (long)h * w;
    float* o = out + (long)b * 3 * plane + (long)oy * w + ox;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        // ATen order: wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d)
        float top = tx.w0 * a[ch] + tx.w1 * c[ch];
        float bot = tx.w0 * d[ch] + tx.w1 * e[ch];
        float v = ty.w0 * top + ty.w1 * bot;
        v = v / 255.0f;
        o[ch * plane] = (v - mean[ch]) / istd[ch];
    }
}

// The same values written straight into the patch-embedding GEMM's A matrix (Conv2d(3 -> D, k = s = p) as im2col rows:
// A[(b, py, px)][c p^2 + i p + j] = x[b][c][py p + i][px p + j], HF Dinov2PatchEmbeddings) in the engine's operand type, plus the
// cls-token rows of the residual stream: d2s_pipeline's pre-process and the engine's patchify in one launch (the [B,3,h,w] float
// planes in between are never materialised).  Columns >= 3 p^2 of A are zero from allocation and never written.
template <int FMT, typename T>
__global__ void __launch_bounds__(256)
preprocess_patch_kernel(const void* __restrict__ frames, int B, int H, int W, int stride, T* __restrict__ A, int h, int w, int p, int Kp,
                        float sy, float sx, float m0, float m1, float m2, float is0, float is1, float is2,
                        const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ resid, int N, int D) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cls && idx < (long)B * D) { int b = (int)(idx / D), d = (int)(idx % D); resid[(long)b * N * D + d] = cls[d] + pos[d]; }
    long total = (long)B * h * w;
    if (idx >= total) return;
    int ox = (int)(idx % w);
    int oy = (int)((idx / w) % h);
    int b = (int)(idx / ((long)w * h));
    int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
    Tap ty = linear_tap(oy, sy, Hs, false);
    Tap tx = linear_tap(ox, sx, Ws, false);
    long fo = (long)b * H * W;
    float a[3], c[3], d[3], e[3];
    load_px<FMT>(frames, fo, H, W, ty.i0 * stride, tx.i0 * stride, a[0], a[1], a[2]);
    load_px<FMT>(frames, fo, H, W, ty.i0 * stride, tx.i1 * stride, c[0], c[1], c[2]);
    load_px<FMT>(frames, fo, H, W, ty.i1 * stride, tx.i0 * stride, d[0], d[1], d[2]);
    load_px<FMT>(frames, fo, H, W, ty.i1 * stride, tx.i1 * stride, e[0], e[1], e[2]);
    const float mean[3] = {m0, m1, m2};
    const float istd[3] = {is0, is1, is2};
    const int gw = w / p, gh = h / p;
    const int py = oy / p, px = ox / p;
    T* o = A + ((long)(b * gh + py) * gw + px) * Kp + (oy - py * p) * p + (ox - px * p);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float top = tx.w0 * a[ch] + tx.w1 * c[ch];
        float bot = tx.w0 * d[ch] + tx.w1 * e[ch];
        float v = ty.w0 * top + ty.w1 * bot;
        v = v / 255.0f;
        v = (v - mean[ch]) / istd[ch];
        if constexpr (sizeof(T) == 2) o[ch * p * p] = f2bf(v); else o[ch * p * p] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// A2-A4, the IS_CUDA branch of _resize_patch_aligned_t (reference depth.py:698-699 -- the branch the reference takes on
// a CUDA *or ROCm* device): ONE F.interpolate(bicubic, align_corners=False, antialias=True) from the full frame, i.e.
// ATen's _upsample_bicubic2d_aa: separable Keys cubic (a = -0.5) stretched by the scale (support = 2 * scale source
// pixels: 15 taps per axis at 1080p -> 294x518, 30 at 4K), weights normalised per output index, horizontal pass over
// every contributing source row, then the vertical pass over those results; no clamp (the lobes overshoot 0..255).
// One block = an AA_TH x AA_TW output tile of one frame: tap tables for its columns / rows are built once in LDS, the
// horizontal pass of the tile's source-row span is staged in LDS ([rows][AA_TW][3] floats), the vertical pass reads it.
// Same operation order per value as the two-pass ATen kernel (taps accumulated first to last, mul then add).
// ------------------------------------------------------------------------------------------------
constexpr int AA_TW = 32, AA_TH = 8;          // output tile (columns x rows); rows shrink for very large frames
constexpr int AA_MAXTAPS = 64;                // taps per axis: ceil(4 * scale) + 2 <= 64  <=>  frames up to ~15x the model input
constexpr int AA_MAXROWS = 128;               // source rows whose horizontal pass one tile stages

struct AaAxis { int xmin, xsize; };
// ATen HelperInterpCubic geometry for output index i (UpSampleKernel.cpp _compute_indices_min_size_weights_aa, float math)
__device__ __forceinline__ AaAxis aa_cubic_taps(int i, float scale, int in_size, float* w /* [AA_MAXTAPS] */) {
    const float support = scale >= 1.f ? 2.0f * scale : 2.0f;
    const float invscale = scale >= 1.f ? 1.0f / scale : 1.f;
    const float center = scale * ((float)i + 0.5f);
    int lo = (int)((double)(center - support) + 0.5);           // ATen adds the double literal 0.5 before truncating
    int hi = (int)((double)(center + support) + 0.5);
    AaAxis t;
    t.xmin = lo > 0 ? lo : 0;
    t.xsize = (hi < in_size ? hi : in_size) - t.xmin;
    if (t.xsize > AA_MAXTAPS) t.xsize = AA_MAXTAPS;             // (the launcher rejects scales that would need more)
    float total = 0.f;
    for (int j = 0; j < t.xsize; ++j) {
        float x = fabsf(((float)(j + t.xmin) - center + 0.5f) * invscale);
        const float a = -0.5f;
        float v = x < 1.f ? ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f : (x < 2.f ? ((a * x - 5.f * a) * x + 8.f * a) * x - 4.f * a : 0.f);
        w[j] = v; total += v;
    }
    if (total != 0.f) for (int j = 0; j < t.xsize; ++j) w[j] = w[j] / total;
    return t;
}

template <int FMT>
__global__ void __launch_bounds__(256)
preprocess_aa_kernel(const void* __restrict__ frames, int B, int H, int W, float* __restrict__ out, int h, int w,
                     float sy, float sx, int th, float m0, float m1, float m2, float s0, float s1, float s2) {
    __shared__ float wx[AA_TW][AA_MAXTAPS + 1], wy[AA_TH][AA_MAXTAPS + 1];
    __shared__ int xmn[AA_TW], xsz[AA_TW], ymn[AA_TH], ysz[AA_TH];
    __shared__ float hp[AA_MAXROWS][AA_TW][3];
    const int tid = threadIdx.x;
    const int tiles_x = (w + AA_TW - 1) / AA_TW, tiles_y = (h + th - 1) / th;
    int bid = blockIdx.x;
    const int tx0 = (bid % tiles_x) * AA_TW, ty0 = ((bid / tiles_x) % tiles_y) * th, b = bid / (tiles_x * tiles_y);
    if (tid < AA_TW) {
        int ox = tx0 + tid < w ? tx0 + tid : w - 1;
        AaAxis t = aa_cubic_taps(ox, sx, W, wx[tid]);
        xmn[tid] = t.xmin; xsz[tid] = t.xsize;
    } else if (tid >= 64 && tid < 64 + th) {
        int r = tid - 64, oy = ty0 + r < h ? ty0 + r : h - 1;
        AaAxis t = aa_cubic_taps(oy, sy, H, wy[r]);
        ymn[r] = t.xmin; ysz[r] = t.xsize;
    }
    __syncthreads();
    const int nr_out = ty0 + th <= h ? th : h - ty0;
    const int y_first = ymn[0], y_last = ymn[nr_out - 1] + ysz[nr_out - 1];     // source rows [y_first, y_last)
    const int nrows = y_last - y_first;
    const long fo = (long)b * H * W;
    // horizontal pass of every contributing source row, for the tile's columns
    for (int idx = tid; idx < nrows * AA_TW; idx += 256) {
        const int r = idx / AA_TW, c = idx - r * AA_TW;
        float ar = 0.f, ag = 0.f, ab = 0.f;
        const int x0 = xmn[c], n = xsz[c];
        for (int k = 0; k < n; ++k) {
            float pr, pg, pb;
            load_px<FMT>(frames, fo, H, W, y_first + r, x0 + k, pr, pg, pb);
            const float wk = wx[c][k];
            ar += wk * pr; ag += wk * pg; ab += wk * pb;
        }
        hp[r][c][0] = ar; hp[r][c][1] = ag; hp[r][c][2] = ab;
    }
    __syncthreads();
    // vertical pass + /255 + (x - mean) / std
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    const long plane = (long)h * w;
    for (int idx = tid; idx < nr_out * AA_TW; idx += 256) {
        const int r = idx / AA_TW, c = idx - r * AA_TW;
        if (tx0 + c >= w) continue;
        const int r0 = ymn[r] - y_first, n = ysz[r];
        float acc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < n; ++k) {
            const float wk = wy[r][k];
            acc[0] += wk * hp[r0 + k][c][0]; acc[1] += wk * hp[r0 + k][c][1]; acc[2] += wk * hp[r0 + k][c][2];
        }
        float* o = out + (long)b * 3 * plane + (long)(ty0 + r) * w + tx0 + c;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = acc[ch] / 255.0f;
            o[ch * plane] = (v - mean[ch]) / stdv[ch];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A13: depth up-sample, bilinear align_corners=False
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
upsample_depth_kernel(const float* __restrict__ in, int B, int h, int w, float* __restrict__ out,
                      int H, int W, float sy, float sx) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * H * W;
    if (idx >= total) return;
    int x = (int)(idx % W);
    int y = (int)((idx / W) % H);
    int b = (int)(idx / ((long)W * H));
    Tap ty = linear_tap(y, sy, h, false);
    Tap tx = linear_tap(x, sx, w, false);
    const float* p = in + (long)b * h * w;
    float top = tx.w0 * p[ty.i0 * w + tx.i0] + tx.w1 * p[ty.i0 * w + tx.i1];
    float bot = tx.w0 * p[ty.i1 * w + tx.i0] + tx.w1 * p[ty.i1 * w + tx.i1];
    out[idx] = ty.w0 * top + ty.w1 * bot;
}

// ------------------------------------------------------------------------------------------------
// A14 stereo warp
// ------------------------------------------------------------------------------------------------
struct WarpGeom {
    int H, W;            // source frame
    int dh, dw;          // depth grid
    float dsy, dsx;      // depth grid scales (in/out)
    int Hp, Wp;          // eye frame after pad_to_aspect
    int pad_top, pad_left;
    int out_h, out_w;    // packed output
    int mode;
    float conv, ratio, max_px;
};

__device__ __forceinline__ float depth_at(const float* __restrict__ dep, const WarpGeom& g, int y, int x) {
    Tap ty = linear_tap(y, g.dsy, g.dh, false);
    Tap tx = linear_tap(x, g.dsx, g.dw, false);
    float top = tx.w0 * dep[ty.i0 * g.dw + tx.i0] + tx.w1 * dep[ty.i0 * g.dw + tx.i1];
    float bot = tx.w0 * dep[ty.i1 * g.dw + tx.i0] + tx.w1 * dep[ty.i1 * g.dw + tx.i1];
    return ty.w0 * top + ty.w1 * bot;
}

// reflect about [0, span] then clip (ATen grid_sampler reflect_coordinates, align_corners=True)
__device__ __noinline__ float reflect_clip_slow(float x, float span) {         // |x| > 2*span: more than two reflections
    float extra = fmodf(x, span);
    int flips = (int)floorf(x / span);
    float r = (flips & 1) ? span - extra : extra;
    return fminf(fmaxf(r, 0.f), span);
}
__device__ __forceinline__ float reflect_clip(float x, float span) {
    if (span <= 0.f) return 0.f;
    x = fabsf(x);                                              // reflection at 0
    if (x <= span) return x;                                   // common case
    if (x <= 2.f * span) return span - (x - span);             // one reflection at span (= span - fmod(x, span), exact)
    return reflect_clip_slow(x, span);
}

// one eye-plane sample: E(eye, y, x) for in-frame (y, x); sign = +1 left eye, -1 right eye
template <int IN_FMT>
__device__ __forceinline__ void eye_sample(const void* __restrict__ rgb, long frame_off_px,
                                           const float* __restrict__ dep, const WarpGeom& g,
                                           int y, int x, float sign, float& r, float& gg, float& b) {
    float d = depth_at(dep, g, y, x) - g.conv;
    float shift = ((-d * g.ratio) * g.max_px) * 0.05f;       // reference depth.py:2144-2147
    float sx = reflect_clip((float)x + sign * shift, (float)(g.W - 1));
    int x0 = (int)sx;
    float w1 = sx - (float)x0;
    float w0 = 1.0f - w1;
    int x1 = x0 + 1 < g.W ? x0 + 1 : x0;                     // out-of-range tap has weight 0 anyway
    float r0, g0, b0, r1, g1, b1;
    load_px<IN_FMT>(rgb, frame_off_px, g.H, g.W, y, x0, r0, g0, b0);
    load_px<IN_FMT>(rgb, frame_off_px, g.H, g.W, y, x1, r1, g1, b1);
    if (IN_FMT == D2S_FMT_F32_CHW) {                          // img.clamp(0,255), depth.py:2142
        r0 = fminf(fmaxf(r0, 0.f), 255.f); g0 = fminf(fmaxf(g0, 0.f), 255.f); b0 = fminf(fmaxf(b0, 0.f), 255.f);
        r1 = fminf(fmaxf(r1, 0.f), 255.f); g1 = fminf(fmaxf(g1, 0.f), 255.f); b1 = fminf(fmaxf(b1, 0.f), 255.f);
    }
    r = w0 * r0 + w1 * r1; gg = w0 * g0 + w1 * g1; b = w0 * b0 + w1 * b1;
}

// value of the concatenated (padded) stereo frame at cat coordinates
template <int IN_FMT>
__device__ __forceinline__ void cat_sample(const void* __restrict__ rgb, long fo, const float* __restrict__ dep,
                                           const WarpGeom& g, int cy, int cx, bool tab,
                                           float& r, float& gg, float& b) {
    int eye, yp, xp;
    if (tab) { eye = cy >= g.Hp; yp = cy - eye * g.Hp; xp = cx; }
    else     { eye = cx >= g.Wp; xp = cx - eye * g.Wp; yp = cy; }
    int y = yp - g.pad_top, x = xp - g.pad_left;
    if (y < 0 || y >= g.H || x < 0 || x >= g.W) { r = gg = b = 0.f; return; }
    eye_sample<IN_FMT>(rgb, fo, dep, g, y, x, eye ? -1.0f : 1.0f, r, gg, b);
}

// v_cvt_pk_u8_f32 = round-half-even + saturate to [0,255] + insert into byte `sel` (checked on gfx950 with
// tools/ubench/cvt_test: 0.5->0, 1.5->2, 2.5->2, 255.5->255, 300->255, -5->0)
__device__ __forceinline__ uint8_t to_u8(float v) { return (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(v, 0, 0); }
__device__ __forceinline__ uint32_t pack4_u8(float a, float b, float c, float d) {
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0);
    r = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(d, 3, r);
}

// Generic kernel: one thread per output pixel, any alignment / padding / mode.
template <int IN_FMT, int OUT_FMT>
__global__ void __launch_bounds__(256)
stereo_warp_generic(const void* __restrict__ rgb, const float* __restrict__ depth, void* __restrict__ out,
                    int B, WarpGeom g) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)g.out_h * g.out_w;
    if (idx >= per * B) return;
    int ox = (int)(idx % g.out_w);
    int oy = (int)((idx / g.out_w) % g.out_h);
    int b = (int)(idx / per);
    long fo = (long)b * g.H * g.W;
    const float* dep = depth + (long)b * g.dh * g.dw;
    bool tab = (g.mode == D2S_MODE_HALF_TAB || g.mode == D2S_MODE_FULL_TAB);
    float r, gg, bl;
    if (g.mode == D2S_MODE_FULL_SBS || g.mode == D2S_MODE_FULL_TAB) {
        cat_sample<IN_FMT>(rgb, fo, dep, g, oy, ox, tab, r, gg, bl);
    } else {
        float r2, g2, b2;
        if (tab) { cat_sample<IN_FMT>(rgb, fo, dep, g, 2 * oy, ox, true, r, gg, bl);
                   cat_sample<IN_FMT>(rgb, fo, dep, g, 2 * oy + 1, ox, true, r2, g2, b2); }
        else     { cat_sample<IN_FMT>(rgb, fo, dep, g, oy, 2 * ox, false, r, gg, bl);
                   cat_sample<IN_FMT>(rgb, fo, dep, g, oy, 2 * ox + 1, false, r2, g2, b2); }
        r = (r + r2) * 0.5f; gg = (gg + g2) * 0.5f; bl = (bl + b2) * 0.5f;   // F.interpolate(mode='area')
    }
    r = fminf(fmaxf(r, 0.f), 255.f); gg = fminf(fmaxf(gg, 0.f), 255.f); bl = fminf(fmaxf(bl, 0.f), 255.f);
    if (OUT_FMT == D2S_FMT_U8_HWC) {
        uint8_t* o = (uint8_t*)out + (b * per + (long)oy * g.out_w + ox) * 3;
        o[0] = to_u8(r); o[1] = to_u8(gg); o[2] = to_u8(bl);
    } else if (OUT_FMT == D2S_FMT_F32_HWC) {
        float* o = (float*)out + (b * per + (long)oy * g.out_w + ox) * 3;
        o[0] = r; o[1] = gg; o[2] = bl;
    } else {  // F32_CHW
        float* o = (float*)out + b * per * 3 + (long)oy * g.out_w + ox;
        o[0] = r; o[per] = gg; o[2 * per] = bl;
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path (round 6; Full-SBS / Full-TAB / Half-SBS / Half-TAB, u8 HWC in and out, no padding, W % 4 == 0; depth at model resolution
// with 3 dw < 2 W, or at the frame's size).  Rounds 2-5 ran LDS-staged float kernels here (stereo_warp_stream / stereo_warp_lanes: fp32
// planes in LDS, six ds_read_b32 and six float operations per pixel and eye, a transpose through LDS for the stores, one block barrier
// per row, 16 waves per CU: 0.41 of the HBM rate, docs/LAB_NOTEBOOK and profiles/r5_09); they were removed when this kernel covered
// their inputs.  It has no barrier, no float blend and no shared state between waves (nothing runs in lock step):
//   * a lane owns FOUR CONSECUTIVE pixels of a row and both eyes: its results are 12 contiguous bytes per eye (one 12-byte store,
//     768 contiguous bytes per wave), no transpose;
//   * a wave stages ITS OWN window of the source row (256 + 2 x 64 pixels) in 2 KiB of LDS as one RGBX dword per pixel -- aligned
//     12-byte global loads, three instructions to unpack four pixels -- TWO rows ahead through two register sets; a wave's LDS
//     operations execute in order, so the window of row r + 1 overwrites row r's behind its reads without a barrier or a second buffer;
//   * the two taps of a pixel and eye are ONE ds_read2_b32 (x0, x0 + 1).  Measured (tools/ubench/lds_tap_patterns.hip): with lanes
//     4 pixels apart that read runs into 4-way bank conflicts (16.5 cycles per wave instruction); two pad dwords per 32 pixels (the
//     first = a copy of the next chunk's first pixel, so x0 + 1 stays adjacent) bring it to 7.2 against 6.8 for consecutive lanes.  An
//     8-byte LDS read at a 4-byte-aligned address (ds_read_b64) takes 65 cycles and an unaligned 8-byte GLOBAL load per tap 29
//     cycles of the CU's texture path (tools/ubench/unaligned_taps.hip: the first form of this kernel, 323 us at batch 32);
//   * coordinates are 16.16 fixed point (the reference's own float32 sum x + shift has 2^-13 .. 2^-14 px of resolution at x ~ 1000):
//     x0 = s >> 16, w1 = s & 0xffff, w0 = 65535 - w1, and a channel is v_dot2_u32_u16([p0, p1], [w0, w1]) + 32768 -> byte 2 of the
//     result is the rounded value: one v_perm_b32 (bytes -> u16 pair) and one dot per channel instead of two conversions, two
//     products and a sum.  Weight error <= 2^-16: the blend is within 0.008 of a level of the exact bilinear value, inside the
//     1-LSB gate like the float kernels' 0.004 (from their float32 coordinate); 0.2-0.4 % of the bytes differ from theirs, by 1;
//   * the depth sample (A13 fused) comes from three depth columns per lane (4 pixels span < 1 column step when 3 dw < W), turned into
//     fixed-point column shifts first; the row's vertical tap is computed by one lane per row and read back with v_readlane;
//   * Half modes chain the second blend of a pair into the first through the dot's accumulator (one shift at the end);
//   * everything about a row is wave-uniform and lives in scalar registers (row pointers, depth row offsets, output row pointers).
// A wave-row whose |shift| reaches the staged halo takes its taps from global memory (8-byte loads, reflection in fixed point, one per
// side); beyond |shift| >= W - 1, and for the last wave of the buffer's last row (whose 8-byte loads could read 2 bytes past the
// allocation), the per-pixel float arithmetic of the generic kernel.
// Batch 32, 1080p Full-SBS: 205 -> 153 us (0.50 of 8 TB/s), 4K 56 -> 33 us (0.56); Half-SBS 213 -> 137, Half-TAB 195 -> 132
// (profiles/r6_01_warp_gather.md: cut-point builds, counters, what the remaining time is).
// ------------------------------------------------------------------------------------------------
typedef unsigned short wg_u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wl_u3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) uint32_t wg_lds_u32;
__device__ __forceinline__ uint2 wg_load8(const uint8_t* p) { uint2 d; __builtin_memcpy(&d, p, 8); return d; }     // global_load_dwordx2, any alignment
__device__ __forceinline__ uint32_t wg_dot(uint32_t a, uint32_t w, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(wg_u16x2, a), __builtin_bit_cast(wg_u16x2, w), c, false);
}
// explicit global-address-space types: pointers that went through scalar arithmetic keep global_load / global_store (not flat_*)
typedef __attribute__((address_space(1))) const uint8_t wg_gc8;
typedef __attribute__((address_space(1))) uint8_t wg_g8;
typedef uint32_t wg_u3 __attribute__((ext_vector_type(3)));         // (a plain vector type: HIP's uint3 is a class on the host pass)
typedef __attribute__((address_space(1), aligned(4))) wg_u3 wg_g_u3w;
// an opaque copy of a per-lane 32-bit offset, made INSIDE the loop: the zero-extension then stays in the loop's block, where instruction
// selection can fold "uniform 64-bit base + zext(lane offset)" into the scalar-base addressing mode (hoisted out of the loop as a
// 64-bit register pair it costs a v_lshl_add_u64 per access)
__device__ __forceinline__ uint32_t wg_v(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// 12 bytes at scalar base + lane offset, issued from inline asm (the compiler neither waits for it nor copies its result: WG_WAIT)
__device__ __forceinline__ void wg_gload3(wl_u3& d, uint32_t off, wg_gc8* base) {
    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
typedef uint32_t wg_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wg_gload(wl_u3& d, uint32_t off, wg_gc8* base) { wg_gload3(d, off, base); }
__device__ __forceinline__ void wg_gload(wg_u4& d, uint32_t off, wg_gc8* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
template <int N> struct wg_vec;
template <> struct wg_vec<3> { typedef wl_u3 type; };
template <> struct wg_vec<4> { typedef wg_u4 type; };
// a wave-uniform value pinned to a scalar register
__device__ __forceinline__ int wg_s(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float wg_sf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
constexpr int WG_MG = 64;                                  // staged halo each side (pixels)
constexpr int WG_WIN_BYTES = 2048;                         // per wave: (256 + 2 * 64) pixels * 34 / 32 dwords = 1 632 bytes, rounded up

// the float path for one lane's 4 pixels of one output row (Half-TAB: of the row pair y, y + 1), both eyes: the generic kernel's
// arithmetic (eye_sample), inlined -- a call in the row loop would make the register allocator keep the loop's values in scratch
__device__ __forceinline__ void wg_eye_sample(const uint8_t* __restrict__ rgb, long fo, const float* __restrict__ dep, const WarpGeom& g,
                                              int y, int x, float sign, float& r, float& gg, float& b) {
    const float d = depth_at(dep, g, y, x) - g.conv;
    const float shift = ((-d * g.ratio) * g.max_px) * 0.05f;
    const float span = (float)(g.W - 1);
    float sx = fabsf((float)x + sign * shift);
    if (sx > span) {                                           // ATen reflect_coordinates + clip
        const float extra = fmodf(sx, span);
        const int flips = (int)floorf(sx / span);
        sx = (flips & 1) ? span - extra : extra;
        sx = fminf(fmaxf(sx, 0.f), span);
    }
    const int x0 = (int)sx;
    const float w1 = sx - (float)x0, w0 = 1.0f - w1;
    const int x1 = x0 + 1 < g.W ? x0 + 1 : x0;
    const uint8_t* p0 = rgb + (fo + (long)y * g.W + x0) * 3;
    const uint8_t* p1 = rgb + (fo + (long)y * g.W + x1) * 3;
    r = w0 * (float)p0[0] + w1 * (float)p1[0]; gg = w0 * (float)p0[1] + w1 * (float)p1[1]; b = w0 * (float)p0[2] + w1 * (float)p1[2];
}
template <int MODE>
__device__ __forceinline__ void wg_slow_row(const uint8_t* __restrict__ rgb, const float* __restrict__ depth, uint8_t* __restrict__ out,
                                            const WarpGeom& g, int b, int y, int x) {
    constexpr bool HSBS = MODE == D2S_MODE_HALF_SBS, HTAB = MODE == D2S_MODE_HALF_TAB;
    const long fo = (long)b * g.H * g.W, per = (long)g.out_h * g.out_w;
    const float* dep = depth + (long)b * g.dh * g.dw;
    for (int eye = 0; eye < 2; ++eye) {
        const float sg = eye ? -1.f : 1.f;
        const long orow = MODE == D2S_MODE_FULL_TAB ? (long)eye * g.H + y : (HTAB ? (long)eye * (g.H / 2) + (y >> 1) : y);
        uint8_t* o = out + (b * per + orow * g.out_w) * 3;
        for (int k = 0; k < 4; k += (HSBS ? 2 : 1)) {
            float r, gg, bl;
            wg_eye_sample(rgb, fo, dep, g, y, x + k, sg, r, gg, bl);
            if (HTAB || HSBS) {
                float r2, g2, b2;
                wg_eye_sample(rgb, fo, dep, g, HTAB ? y + 1 : y, HSBS ? x + k + 1 : x + k, sg, r2, g2, b2);
                r = (r + r2) * 0.5f; gg = (gg + g2) * 0.5f; bl = (bl + b2) * 0.5f;
            }
            const long col = HSBS ? (((long)eye * g.W + x + k) >> 1) : (MODE == D2S_MODE_FULL_SBS ? (long)eye * g.W + x + k : x + k);
            o[col * 3] = to_u8(r); o[col * 3 + 1] = to_u8(gg); o[col * 3 + 2] = to_u8(bl);
        }
    }
}

// NC: depth columns a lane's 4 pixels can touch -- 3 when 3 dw < W (model-resolution depth under a 1080p+ frame), 4 when 3 dw < 2 W
// (720p frames) and for DIRECT; DIRECT: the depth map has the frame's size (the drop-in make_sbs(rgb, depth[H, W]) surface: A13 already
// applied by predict_depth) -- pixel k reads column k of ONE depth row, no interpolation in either direction.
template <int MODE, int NC, bool DIRECT>
__global__ void __launch_bounds__(256, 4)
stereo_warp_gather(const uint8_t* __restrict__ rgb, const float* __restrict__ depth, uint8_t* __restrict__ out,
                   int B, WarpGeom g, int rpw /* source rows per wave (even for Half-TAB) */, int ntx /* 256-pixel column tiles */) {
    constexpr bool HSBS = MODE == D2S_MODE_HALF_SBS, HTAB = MODE == D2S_MODE_HALF_TAB, HALF = HSBS || HTAB;
    constexpr int NR = HTAB ? 2 : 1;                          // source rows per output row
    constexpr int DL = DIRECT ? 1 : 2;                        // depth loads per step
    __shared__ __attribute__((aligned(WG_WIN_BYTES))) uint32_t win[4][WG_WIN_BYTES / 4];
    const int lane = threadIdx.x & 63, wid = wg_s((int)(threadIdx.x >> 6));
    const int gw = wg_s((int)(blockIdx.x * 4)) + wid;
    // (integer division runs on the vector unit: its results are read back into scalar registers, or everything derived from them --
    //  row, frame, every row pointer -- would be computed per lane)
    const int band = wg_s(gw / ntx), tx = gw - band * ntx;
    const int rows = B * g.H;
    const int R0 = band * rpw;
    if (R0 >= rows) return;
    const int nrow = rows - R0 < rpw ? rows - R0 : rpw;
    int b = wg_s(R0 / g.H), y = R0 - b * g.H;               // (one division per wave)
    const int wx0 = tx * 256;
    const int x = wx0 + 4 * lane;
    const bool act = x < g.W;
    const int xc = act ? x : g.W - 4;                       // idle lanes (last tile) repeat the last group, stores masked
    const int span_fx = (g.W - 1) << 16;
    const long W3 = (long)g.W * 3;
    // this wave's window of a source row: pixels [ws, we), 4-pixel groups
    const int ws = wx0 - WG_MG < 0 ? 0 : wx0 - WG_MG;
    const int we = wx0 + 256 + WG_MG > g.W ? g.W : wx0 + 256 + WG_MG;
    const int ngroups = (we - ws) >> 2;
    const int g0 = lane < ngroups ? lane : ngroups - 1, g1 = lane + 64 < ngroups ? lane + 64 : ngroups - 1;   // (clamped: duplicates write the same dwords)
    const uint32_t gofs0 = (uint32_t)(ws + 4 * g0) * 3u, gofs1 = (uint32_t)(ws + 4 * g1) * 3u;
    const uint32_t wbase = (uint32_t)(size_t)&win[wid][0];  // LDS byte address of this wave's window: a multiple of WG_WIN_BYTES
    wg_lds_u32* const wl0 = (wg_lds_u32*)(wbase + 4u * (uint32_t)(4 * g0 + 2 * (g0 >> 3)));     // where this lane's groups go (two pad dwords per 32 pixels)
    wg_lds_u32* const wl1 = (wg_lds_u32*)(wbase + 4u * (uint32_t)(4 * g1 + 2 * (g1 >> 3)));
    const bool dup0 = (g0 & 7) == 0 && g0 > 0, dup1 = (g1 & 7) == 0;       // first pixel of a 32-pixel chunk: also the first pad slot of the chunk before

    // The first row's window is requested BEFORE the per-lane constants are formed (a hundred-odd instructions of tap arithmetic): at one
    // frame a wave lives for 1-3 rows and that arithmetic would otherwise sit in front of its first HBM round trip.
    wg_gc8* srow = (wg_gc8*)rgb + (long)R0 * W3;
    wl_u3 pwA0, pwA1, pwB0, pwB1;
#ifdef WG_CUT_LOAD           // (tuning aid, tools/build_variant.sh: timing only -- no window loads from global memory)
#define WG_WIN_LOAD(P0, P1, ROW_) { P0 = (wl_u3){(uint32_t)(size_t)(ROW_), 0x01020304u, 0x05060708u}; P1 = P0; }
#else
#define WG_WIN_LOAD(P0, P1, ROW_) { wg_gload3(P0, gofs0, (ROW_)); wg_gload3(P1, gofs1, (ROW_)); }
#endif
    WG_WIN_LOAD(pwA0, pwA1, srow)
    // per-lane column constants: NC depth columns c0 .. c0 + NC - 1; pixel k lerps a pair of them by w1[k].  The shift
    // is linear in depth, so the COLUMNS are turned into fixed-point shifts first (shift_fx = depth * K + ck, K = -ratio * max_px *
    // 0.05 * 65536, ck = -conv * K) and a pixel costs NC - 1 FMAs on the column differences (weights 1 below its pair, w1 on it, 0 above).
    const float K = ((-g.ratio * g.max_px) * 0.05f) * 65536.f;
    const float ck = -g.conv * K;
    uint32_t c0b;                                           // byte offset of column c0 inside a depth row
    float wkj[4][NC - 1];                                   // shift(k) = scol[0] + sum_j wkj[k][j] (scol[j + 1] - scol[j])
    {
        const Tap t0 = linear_tap(xc, g.dsx, g.dw, false);
        const int c0 = t0.i0 < g.dw - NC ? t0.i0 : g.dw - NC;
        c0b = 4u * (uint32_t)c0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Tap t = linear_tap(xc + k, g.dsx, g.dw, false);
            const int j0 = t.i0 - c0;                       // first column of the pixel's pair (the clamped last column has no second one)
#pragma unroll
            for (int j = 0; j < NC - 1; ++j) wkj[k][j] = j < j0 ? 1.0f : (j == j0 ? t.w1 : 0.f);
        }
    }
    // a wave whose pixels stay more than MG + 2 away from both frame edges never reflects while |shift| < MG - 1
    const bool interior = wx0 >= WG_MG + 2 && wx0 + 255 <= g.W - 1 - (WG_MG + 2);
    const uint32_t lane_col = (uint32_t)xc * 3u;           // byte column of this lane's group inside one eye's output row
    const int xfx_rel = (xc - ws) << 16, ws_fx = ws << 16;  // fixed-point x of this lane's first pixel relative to the window (the fraction is the same)
    const float small_lim = (float)((WG_MG - 1) << 16);

    // Everything about a ROW is wave-uniform and kept in scalar registers: the source row pointer advances by W * 3 (frames are contiguous:
    // the global row index runs through them), the vertical depth tap is computed on the vector unit once and read back with
    // v_readfirstlane (its row offsets then cost scalar multiplies, not v_mul_lo_u32), lanes add constant 32-bit offsets.
    const uint32_t drow_b = (uint32_t)g.dw * 4u, dplane = (uint32_t)g.dh * drow_b;      // bytes of a depth row / map (the launcher keeps B * dplane < 2^32)
    // Two rows of window loads in flight (register sets A / B; the row loop is unrolled by two so that a set is a fixed group of registers):
    // the set staged at the end of row r was requested at the end of row r - 2.  All loads of the row loop are issued from inline asm and
    // waited for with hand-counted vmcnt: left to the compiler, a load whose result crosses the loop's back edge or
    // a branch gets a register copy -- and an s_waitcnt vmcnt(0) -- right behind its issue (the first builds of this kernel ran 165 us at
    // every prefetch depth and occupancy for that reason).  Every load is unconditional (clamped to the wave's last row): the counts below
    // are the same on every path.
    typename wg_vec<NC>::type dA0, dA1;                     // depth columns (rows i0 / i1 of the depth grid) of the next row to be computed, as bits:
    float dwA0, dwA1;                                       // ONE set -- a step turns them into three column shifts, then requests the next row's
                                                            // (+ their vertical weights, scalar registers)
    // The vertical depth tap of a row (ATen's source index: a dozen float operations) is computed once per row by ONE lane: lane l holds
    // the tap of the band's row 64 q + l (byte offsets of its two depth rows inside the buffer, both weights), refilled every 64 rows; a
    // row reads its lane's four values back with v_readlane (index in a scalar register).
    uint32_t tap_o0 = 0, tap_o1 = 0;
    float tap_w0 = 0.f, tap_w1 = 0.f;
    auto tap_fill = [&](int fb, int fy) {                   // (fb, fy): frame / row of the band's row 64 q
        int yl = fy + lane, bl = fb;
        if (yl >= g.H) { yl -= g.H; ++bl; }                 // (a band is shorter than a frame)
        if (bl >= B) { bl = B - 1; yl = g.H - 1; }
        const Tap t = linear_tap(yl, g.dsy, g.dh, false);
        tap_o0 = (uint32_t)bl * dplane + (uint32_t)t.i0 * drow_b; tap_o1 = (uint32_t)bl * dplane + (uint32_t)t.i1 * drow_b;
        tap_w0 = t.w0; tap_w1 = t.w1;
    };
#define WG_DEPTH_LOAD(RR, T_, B_, W0_, W1_)                                                        \
    {                                                                                             \
        const int l_ = (RR) & 63;                                                                 \
        const uint32_t o0_ = __builtin_amdgcn_readlane(tap_o0, l_), o1_ = __builtin_amdgcn_readlane(tap_o1, l_);      \
        wg_gload(T_, c0b, (wg_gc8*)depth + (unsigned long)o0_);                                   \
        if (!DIRECT) wg_gload(B_, c0b, (wg_gc8*)depth + (unsigned long)o1_);                      \
        W0_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tap_w0), l_));              \
        W1_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tap_w1), l_));              \
    }
    // ONE asm statement per wait (two statements merged by a branch made the compiler copy the in-flight registers ahead of the wait):
    // vmcnt(N) always; when FIRST (wave-uniform: the band's first step, with fewer operations behind it) the stricter vmcnt(N0) first
#ifdef WG_STRICT             // (debugging aid: every wait drains the queue)
#define WG_WAIT(N, N0, FIRST, R0_, R1_) asm volatile("s_waitcnt vmcnt(0)" : "+v"(R0_), "+v"(R1_) : "n"(N), "n"(N0), "s"(wg_s((int)(FIRST))) : "memory")
#define WG_WAIT1(N, N0, FIRST, R0_) asm volatile("s_waitcnt vmcnt(0)" : "+v"(R0_) : "n"(N), "n"(N0), "s"(wg_s((int)(FIRST))) : "memory")
#else
#define WG_WAIT1(N, N0, FIRST, R0_)                                                                \
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%2)\n1:\n\ts_waitcnt vmcnt(%1)"   \
                 : "+v"(R0_) : "n"(N), "n"((N0) < (N) ? (N0) : (N)), "s"(wg_s((int)(FIRST))) : "memory", "scc")
#define WG_WAIT(N, N0, FIRST, R0_, R1_)                                                            \
    asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%3)\n1:\n\ts_waitcnt vmcnt(%2)"   \
                 : "+v"(R0_), "+v"(R1_) : "n"(N), "n"((N0) < (N) ? (N0) : (N)), "s"(wg_s((int)(FIRST))) : "memory", "scc")
#endif
    // 12 bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 -> four RGBX dwords; the first pixel of a 16-pixel chunk is also written into the
    // pad slot of the chunk before it
#define WG_STAGE(PW, WL_, DUP_)                                                                    \
    {                                                                                             \
        WL_[0] = PW[0]; WL_[1] = __builtin_amdgcn_alignbyte(PW[1], PW[0], 3);                     \
        WL_[2] = __builtin_amdgcn_alignbyte(PW[2], PW[1], 2); WL_[3] = PW[2] >> 8;                \
        if (DUP_) WL_[-2] = PW[0];                                                                \
    }
    tap_fill(b, y);
    WG_DEPTH_LOAD(0, dA0, dA1, dwA0, dwA1)
    WG_WAIT(DL, DL, 0, pwA0, pwA1);
    WG_STAGE(pwA0, wl0, dup0) WG_STAGE(pwA1, wl1, dup1)                       // row 0 is staged; rows 1 and 2 are requested
    WG_WIN_LOAD(pwA0, pwA1, srow + (1 < nrow ? 1 : 0) * W3)
    WG_WIN_LOAD(pwB0, pwB1, srow + (2 < nrow ? 2 : nrow - 1) * W3)
    uint32_t hold[2][HTAB ? 12 : 1];                        // Half-TAB: the even row's 16.16 sums, chained into the odd row's dots
    bool pair_slow = false;
    // Vector-memory operations a step issues on the LDS path: 2 depth loads (top), ST stores, 2 window loads (end).  The other paths issue
    // at least as many (taps from global memory: 8 more loads; the float path: byte stores), so the constant wait counts below -- exact on
    // the LDS path -- are never larger than the number of younger operations, which is all a vmcnt wait needs to be safe.  (One count per
    // wait, no run-time choice: two asm statements merged by a branch made the compiler copy the in-flight registers ahead of the wait.)
    constexpr int ST_FULL = HSBS ? 4 : 2;
    // (the first step has fewer operations behind it: WG_WAIT's FIRST_ count, applied when r == 0 inside the same asm statement)
    // Every register set is named by an empty asm statement at the end of every step: a set whose load will never be consumed (the
    // clamped loads of a band's last rows) would otherwise be dead to the compiler, which would hand its registers to other values --
    // and the load lands in them later.
#define WG_KEEP_ALL() { asm volatile("" : "+v"(pwA0), "+v"(pwA1), "+v"(pwB0), "+v"(pwB1), "+v"(dA0)); if (!DIRECT) asm volatile("" : "+v"(dA1)); }
    // one source row; set_c: which register set holds the NEXT row's window and THIS row's depth columns (A on even steps, B on odd ones).
    // Half-TAB: a wave's band starts on an even row (rpw and H are even), so the step's parity is the row's place in its pair
    auto row_step = [&](auto set_c, const int r) {
        constexpr int set = decltype(set_c)::value;
        constexpr int h = HTAB ? set : 0;
        constexpr int ST_CUR = HTAB ? (h ? 2 : 0) : ST_FULL, ST_PREV = HTAB ? (h ? 0 : 2) : ST_FULL;
        {
            int nb = b, ny = y + 1;
            if (ny >= g.H) { ny = 0; ++nb; }
            const bool has_next = r + 1 < nrow;
            nb = wg_s(has_next ? nb : b); ny = wg_s(has_next ? ny : y);      // (readfirstlane: the row state stays in scalar registers; the last row repeats itself)
            // this row's depth columns (requested by the previous step behind its own use of the registers) have landed: younger than them
            // are the previous step's stores (ST_PREV) and window loads (2); first step: the prologue's four window loads
            // (DIRECT has one depth register set: it must not be named twice in one asm statement -- two tied operands on one variable
            //  make the compiler copy the in-flight registers in front of the wait)
            if (DIRECT) WG_WAIT1(2 + ST_PREV, 4, set == 0 && r == 0, dA0); else WG_WAIT(2 + ST_PREV, 4, set == 0 && r == 0, dA0, dA1);
            if (HTAB && h == 0) pair_slow = false;
            // ---- fixed-point shifts: the columns first, then this lane's 4 pixels
            float scol[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j)
                scol[j] = DIRECT ? fmaf(__uint_as_float(dA0[j]), K, ck) : fmaf(fmaf(dwA1, __uint_as_float(dA1[j]), dwA0 * __uint_as_float(dA0[j])), K, ck);
#pragma unroll
            for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(scol[j]));       // (the columns are formed before the registers are reloaded)
            WG_KEEP_ALL()
            {
                const int rn = has_next ? r + 1 : r;
                if ((rn & 63) == 0 && has_next) tap_fill(nb, ny);   // (wave-uniform, every 64 rows)
                WG_DEPTH_LOAD(rn, dA0, dA1, dwA0, dwA1)
            }
            // (a pixel's shift is a convex combination of two columns: bounds on the columns bound it)
            float amax = fabsf(scol[0]);
#pragma unroll
            for (int j = 1; j < NC; ++j) amax = fmaxf(amax, fabsf(scol[j]));
            const bool all_small = __all(amax < small_lim);
            int shq[4];
            float dj[NC - 1];
#pragma unroll
            for (int j = 0; j < NC - 1; ++j) dj[j] = scol[j + 1] - scol[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float sh = scol[0];
#pragma unroll
                for (int j = 0; j < NC - 1; ++j) sh = fmaf(wkj[k][j], dj[j], sh);
                shq[k] = (int)sh;
            }
            bool slow = false;
            if (!all_small) slow = ((R0 + r == rows - 1) && tx == ntx - 1) || !__all(amax < (float)span_fx);
            if (HTAB) { pair_slow = pair_slow || slow; slow = pair_slow; }
            if (slow) {
                if (act && h == NR - 1) wg_slow_row<MODE>(rgb, depth, out, g, b, HTAB ? y - 1 : y, x);
            } else
#ifdef WG_CUT_BLEND          // (timing only: load + stage, no taps / blends / stores)
            if (win[wid][lane] == 0x12345678u && shq[0] == 0x7654321) out[lane] = 1; else if (false)
#endif
            {
                // output row(s) of this source row: wave-uniform pointers
                const long orow = MODE == D2S_MODE_FULL_TAB ? (long)b * 2 * g.H + y : (HTAB ? (long)b * g.H + (y >> 1) : (long)b * g.H + y);
                wg_g8* const o0 = (wg_g8*)out + orow * g.out_w * 3;
                const long eye_ofs = MODE == D2S_MODE_FULL_SBS ? W3 : (HSBS ? W3 / 2 : (MODE == D2S_MODE_FULL_TAB ? (long)g.H * W3 : (long)(g.H / 2) * W3));
                auto blend_row = [&](auto lds_c, auto easy_c) {
                    constexpr bool lds = decltype(lds_c)::value, easy = decltype(easy_c)::value;
#pragma unroll
                    for (int eye = 0; eye < 2; ++eye) {
                        uint2 tp[4];
                        uint32_t wp[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            int s = xfx_rel + (k << 16) + (eye ? -shq[k] : shq[k]);      // window-relative
                            if (!easy) {                      // absolute: one reflection per side, then x0 <= W - 2
                                s += ws_fx;
                                s = s < 0 ? -s : s;
                                const int s2 = 2 * span_fx - s;
                                s = s < s2 ? s : s2;
                                s = s < span_fx - 1 ? s : span_fx - 1;
                                if (lds) s -= ws_fx;
                            }
                            wp[k] = __builtin_amdgcn_perm((uint32_t)s, ~(uint32_t)s, 0x05040100u);      // [65535 - w1 | w1]
                            if (lds) {                        // window pixel xr at dword xr + 2 (xr >> 5)
                                const uint32_t sa = wbase + (((uint32_t)s >> 21) << 3);
                                wg_lds_u32* q = (wg_lds_u32*)(sa + (((uint32_t)s >> 16) << 2));
                                tp[k].x = q[0]; tp[k].y = q[1];
                            } else {
                                wg_gc8* q = srow + (uint32_t)((s >> 16) * 3);
                                __builtin_memcpy(&tp[k], (const uint8_t*)q, 8);
                            }
                        }
                        uint32_t v[12];                       // R0 G0 B0 R1 ... B3 (Half-SBS: two pixels): value in byte 2
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t dx = tp[k].x, dy = tp[k].y, w = wp[k];
                            // [p0, p1] as a u16 pair per channel, from RGBX dwords (LDS) or from the packed bytes R0 G0 B0 R1 | G1 B1 . .
                            const uint32_t aR = __builtin_amdgcn_perm(dy, dx, lds ? 0x0c040c00u : 0x0c030c00u),
                                           aG = __builtin_amdgcn_perm(dy, dx, lds ? 0x0c050c01u : 0x0c040c01u),
                                           aB = __builtin_amdgcn_perm(dy, dx, lds ? 0x0c060c02u : 0x0c050c02u);
                            if (HTAB) {
                                const uint32_t cR = h ? hold[eye][3 * k] : 65536u, cG = h ? hold[eye][3 * k + 1] : 65536u, cB = h ? hold[eye][3 * k + 2] : 65536u;
                                v[3 * k] = wg_dot(aR, w, cR); v[3 * k + 1] = wg_dot(aG, w, cG); v[3 * k + 2] = wg_dot(aB, w, cB);
                            } else if (HSBS) {                // pixels 2 q, 2 q + 1 chain into one sum
                                const int q = k >> 1;
                                const uint32_t cR = (k & 1) ? v[3 * q] : 65536u, cG = (k & 1) ? v[3 * q + 1] : 65536u, cB = (k & 1) ? v[3 * q + 2] : 65536u;
                                v[3 * q] = wg_dot(aR, w, cR); v[3 * q + 1] = wg_dot(aG, w, cG); v[3 * q + 2] = wg_dot(aB, w, cB);
                            } else { v[3 * k] = wg_dot(aR, w, 32768u); v[3 * k + 1] = wg_dot(aG, w, 32768u); v[3 * k + 2] = wg_dot(aB, w, 32768u); }
                        }
                        if (HTAB && h == 0) {
#pragma unroll
                            for (int i = 0; i < 12; ++i) hold[eye][i] = v[i];
                            continue;
                        }
                        if (HALF) {
#pragma unroll
                            for (int i = 0; i < (HSBS ? 6 : 12); ++i) v[i] >>= 1;      // two blends + 65536 -> mean in byte 2
                        }
                        // byte 2 of four sums -> one dword: [a b . .] | [. . c d]
#define WG_PACK(A_, B_, C_, D_) (__builtin_amdgcn_perm(v[B_], v[A_], 0x0c0c0602u) | __builtin_amdgcn_perm(v[D_], v[C_], 0x06020c0cu))
                        if (act) {
                            wg_g8* const oe = eye ? o0 + eye_ofs : o0;
                            if (HSBS) {
                                const uint32_t q0 = WG_PACK(0, 1, 2, 3);
                                const uint16_t q1 = (uint16_t)__builtin_amdgcn_perm(v[5], v[4], 0x0c0c0602u);
                                wg_g8* dst = oe + wg_v(lane_col >> 1);             // 6 bytes at a 2-byte aligned address
                                typedef __attribute__((aligned(2))) uint32_t u32_a2;
                                *(__attribute__((address_space(1))) u32_a2*)dst = q0;
                                *(__attribute__((address_space(1))) uint16_t*)(dst + 4) = q1;
                            } else {
                                wg_u3 w3;
                                w3[0] = WG_PACK(0, 1, 2, 3); w3[1] = WG_PACK(4, 5, 6, 7); w3[2] = WG_PACK(8, 9, 10, 11);
#ifdef WG_CUT_STORE          // (timing only: no global stores; one impossible store keeps the values alive)
                                if (w3[0] == 0x12345678u && w3[1] == 0x9abcdef0u && w3[2] == 0x0fedcba9u)
#endif
                                *(wg_g_u3w*)(oe + wg_v(lane_col)) = w3;
                            }
                        }
#undef WG_PACK
                    }
                };
                // (wave-uniform choices: the reflecting form costs 5 more instructions per pixel and eye; taps beyond the halo come from global memory)
                if (all_small) { if (interior) blend_row(std::true_type(), std::true_type()); else blend_row(std::true_type(), std::false_type()); }
                else blend_row(std::false_type(), std::false_type());
            }
            // the next row's window goes into LDS behind this row's tap reads (one wave's LDS operations execute in order).  Its loads were
            // issued at the end of step r - 2; younger: step r - 1's depth loads (2), stores and window loads (2), this step's depth loads (2)
            // and stores (DL = depth loads per step: 1 when the depth map has the frame's size).  Then the set takes the row after the other set's.
            // (first step: the prologue's second window pair, this step's depth loads and stores)
            if (set == 0) WG_WAIT(2 + 2 * DL + ST_PREV + ST_CUR, 2 + DL + ST_CUR, r == 0, pwA0, pwA1); else WG_WAIT(2 + 2 * DL + ST_PREV + ST_CUR, 2 + 2 * DL + ST_PREV + ST_CUR, 0, pwB0, pwB1);
            if (has_next) { if (set == 0) { WG_STAGE(pwA0, wl0, dup0) WG_STAGE(pwA1, wl1, dup1) } else { WG_STAGE(pwB0, wl0, dup0) WG_STAGE(pwB1, wl1, dup1) } }
            {
                wg_gc8* nrow_p = srow + (long)(r + 3 < nrow ? 3 : nrow - 1 - r) * W3;
                if (set == 0) WG_WIN_LOAD(pwA0, pwA1, nrow_p) else WG_WIN_LOAD(pwB0, pwB1, nrow_p)
            }
            b = nb; y = ny; srow += W3;
            WG_KEEP_ALL();
        }
    };
    for (int r = 0; r < nrow; r += 2) {
        row_step(std::integral_constant<int, 0>(), r);
        if (r + 1 < nrow) row_step(std::integral_constant<int, 1>(), r + 1);
    }
    WG_KEEP_ALL();
#undef WG_KEEP_ALL
#undef WG_DEPTH_LOAD
#undef WG_WAIT
#undef WG_WAIT1
#undef WG_WIN_LOAD
#undef WG_STAGE
}

}  // namespace d2s

using namespace d2s;

// d2s_pipeline's pre-process + patchify in one launch (bilinear branch, uint8 HWC frames, bf16 / fp32 patch rows); D2S_E_UNSUPPORTED =
// not this case, nothing launched (the caller runs d2s_preprocess + launch_patchify)
namespace d2s {
bool preprocess_patches_ok(int prec, int fmt, const d2s_pre_params* pre, int H, int W, int h, int w, int p, int Kp) {
    static EnvInt off{"D2S_NO_PREPATCH", 0};
    const int resample = pre && !pre->square ? pre->resample : D2S_RESAMPLE_BILINEAR;     // (the fixed-square branch is always plain bilinear)
    return !off.get() && fmt == D2S_FMT_U8_HWC && resample == D2S_RESAMPLE_BILINEAR && !(h == H && w == W) && h % p == 0 && w % p == 0 && 3 * p * p <= Kp &&
           (prec == D2S_PREC_BF16 || prec == D2S_PREC_FP32);
}
int launch_preprocess_patches(int prec, const void* frames, int fmt, int batch, int H, int W, int decim_stride, const d2s_pre_params* pre,
                              void* A, int h, int w, int p, int Kp, const float* cls, const float* pos, float* resid, int N, int D, hipStream_t st) {
    static const d2s_pre_params dflt = {{0.485f, 0.456f, 0.406f}, {0.229f, 0.224f, 0.225f}, D2S_RESAMPLE_BILINEAR, 0};
    if (!pre) pre = &dflt;
    if (!preprocess_patches_ok(prec, fmt, pre, H, W, h, w, p, Kp)) return D2S_E_UNSUPPORTED;
    if (pre->square) decim_stride = 1;
    const int Hs = (H + decim_stride - 1) / decim_stride, Ws = (W + decim_stride - 1) / decim_stride;
    const float sy = linear_scale(Hs, h, false), sx = linear_scale(Ws, w, false);
    const long total = std::max((long)batch * h * w, (long)batch * D);
    const dim3 grid(cdiv(total, 256)), block(256);
    if (prec == D2S_PREC_BF16)
        hipLaunchKernelGGL((preprocess_patch_kernel<D2S_FMT_U8_HWC, bf16_t>), grid, block, 0, st, frames, batch, H, W, decim_stride, (bf16_t*)A, h, w, p, Kp,
                           sy, sx, pre->mean[0], pre->mean[1], pre->mean[2], pre->std[0], pre->std[1], pre->std[2], cls, pos, resid, N, D);
    else
        hipLaunchKernelGGL((preprocess_patch_kernel<D2S_FMT_U8_HWC, float>), grid, block, 0, st, frames, batch, H, W, decim_stride, (float*)A, h, w, p, Kp,
                           sy, sx, pre->mean[0], pre->mean[1], pre->mean[2], pre->std[0], pre->std[1], pre->std[2], cls, pos, resid, N, D);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
}  // namespace d2s

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" int d2s_preprocess(const void* frames, int fmt, int batch, int H, int W, float* out, int h, int w,
                              int decim_stride, const d2s_pre_params* pre, void* stream) {
    D2S_REQUIRE(frames && out, "null pointer");
    static const d2s_pre_params dflt = {{0.485f, 0.456f, 0.406f}, {0.229f, 0.224f, 0.225f}, D2S_RESAMPLE_BILINEAR, 0};   // depth.py:1798-1799
    if (!pre) pre = &dflt;
    d2s_pre_params sq;
    if (pre->square) {                                           // fixed-square branch (depth.py:1937-1946): plain bilinear of the full frame
        sq = *pre; sq.resample = D2S_RESAMPLE_BILINEAR; pre = &sq; decim_stride = 1;
    }
    const float* mean = pre->mean;
    const float* stdv = pre->std;
    D2S_REQUIRE(batch > 0 && H > 0 && W > 0 && h > 0 && w > 0 && decim_stride >= 1, "bad shape");
    D2S_REQUIRE(pre->resample == D2S_RESAMPLE_BILINEAR || pre->resample == D2S_RESAMPLE_BICUBIC_AA, "bad resample mode");
    hipStream_t st = (hipStream_t)stream;
    if (pre->resample == D2S_RESAMPLE_BICUBIC_AA && !(h == H && w == W)) {
        // the IS_CUDA branch resamples the FULL frame (no ::stride decimation, depth.py:698-699)
        const float sy = (float)H / (float)h, sx = (float)W / (float)w;
        const float spy = sy >= 1.f ? 2.f * sy : 2.f, spx = sx >= 1.f ? 2.f * sx : 2.f;
        D2S_REQUIRE((int)(2.f * spy) + 2 <= AA_MAXTAPS && (int)(2.f * spx) + 2 <= AA_MAXTAPS,
                    "bicubic + antialias pre-process: frame more than ~15x the model input per axis");
        int th = AA_TH;                                           // output rows per tile whose source-row span fits the LDS stage
        while (th > 1 && (int)((float)th * sy + 2.f * spy) + 3 > AA_MAXROWS) th >>= 1;
        D2S_REQUIRE((int)((float)th * sy + 2.f * spy) + 3 <= AA_MAXROWS, "bicubic + antialias pre-process: vertical scale too large");
        dim3 grid((unsigned)((long)batch * cdiv(h, th) * cdiv(w, AA_TW))), block(256);
#define LAUNCH_AA(F) hipLaunchKernelGGL(preprocess_aa_kernel<F>, grid, block, 0, st, frames, batch, H, W, out, h, w, sy, sx, th, \
        mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2])
        if (fmt == D2S_FMT_U8_HWC) LAUNCH_AA(D2S_FMT_U8_HWC);
        else if (fmt == D2S_FMT_U8_CHW) LAUNCH_AA(D2S_FMT_U8_CHW);
        else if (fmt == D2S_FMT_F32_CHW) LAUNCH_AA(D2S_FMT_F32_CHW);
        else { set_error("d2s_preprocess: unsupported frame format"); return D2S_E_UNSUPPORTED; }
#undef LAUNCH_AA
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    if (pre->resample == D2S_RESAMPLE_BICUBIC_AA) decim_stride = 1;   // (same size: both branches return the frame as is)
    int Hs = (H + decim_stride - 1) / decim_stride, Ws = (W + decim_stride - 1) / decim_stride;
    float sy = linear_scale(Hs, h, false), sx = linear_scale(Ws, w, false);
    long total = (long)batch * h * w;
    dim3 grid(cdiv(total, 256)), block(256);
#define LAUNCH_PRE(F) hipLaunchKernelGGL(preprocess_kernel<F>, grid, block, 0, st, frames, batch, H, W, decim_stride, \
        out, h, w, sy, sx, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2])
    if (fmt == D2S_FMT_U8_HWC) LAUNCH_PRE(D2S_FMT_U8_HWC);
    else if (fmt == D2S_FMT_U8_CHW) LAUNCH_PRE(D2S_FMT_U8_CHW);
    else if (fmt == D2S_FMT_F32_CHW) LAUNCH_PRE(D2S_FMT_F32_CHW);
    else { set_error("d2s_preprocess: unsupported frame format"); return D2S_E_UNSUPPORTED; }
#undef LAUNCH_PRE
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

extern "C" int d2s_upsample_depth(const float* in, int batch, int h, int w, float* out, int H, int W, void* stream) {
    D2S_REQUIRE(in && out && batch > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad argument");
    long total = (long)batch * H * W;
    hipLaunchKernelGGL(upsample_depth_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, batch, h, w, out, H, W, linear_scale(h, H, false), linear_scale(w, W, false));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

static int make_geom(int H, int W, int dh, int dw, const d2s_sbs_params* p, WarpGeom& g) {
    if (!p) return D2S_E_INVALID;
    if (p->display_mode < 0 || p->display_mode > 3) return D2S_E_INVALID;
    g.H = H; g.W = W; g.dh = dh; g.dw = dw;
    g.dsy = linear_scale(dh, H, false); g.dsx = linear_scale(dw, W, false);
    g.Hp = H; g.Wp = W; g.pad_top = 0; g.pad_left = 0;
    if (p->fill_16_9) {                                          // pad_to_aspect_tensor, depth.py:2106-2119
        double r_img = (double)W / (double)H, r_t = 16.0 / 9.0;
        if (!(fabs(r_img - r_t) < 1e-3)) {
            if (r_img > r_t) { int nh = (int)nearbyint((double)W / r_t); g.pad_top = (nh - H) / 2; g.Hp = nh; }
            else             { int nw = (int)nearbyint((double)H * r_t); g.pad_left = (nw - W) / 2; g.Wp = nw; }
        }
    }
    g.mode = p->display_mode;
    bool tab = (g.mode == D2S_MODE_HALF_TAB || g.mode == D2S_MODE_FULL_TAB);
    bool full = (g.mode == D2S_MODE_FULL_SBS || g.mode == D2S_MODE_FULL_TAB);
    g.out_h = full && tab ? 2 * g.Hp : g.Hp;
    g.out_w = full && !tab ? 2 * g.Wp : g.Wp;
    g.conv = p->convergence; g.ratio = p->depth_ratio;
    g.max_px = (float)(p->ipd_uv * (double)W);                   // python double product, cast at the multiply
    return D2S_OK;
}

extern "C" int d2s_sbs_shape(int H, int W, const d2s_sbs_params* p, int* out_h, int* out_w) {
    WarpGeom g;
    D2S_REQUIRE(H > 0 && W > 0 && out_h && out_w, "bad argument");
    if (make_geom(H, W, H, W, p, g) != D2S_OK) { set_error("d2s_sbs_shape: bad params"); return D2S_E_INVALID; }
    *out_h = g.out_h; *out_w = g.out_w;
    return D2S_OK;
}

template <int IN_FMT>
static void launch_generic(const void* rgb, const float* depth, void* out, int out_fmt, int batch,
                           const WarpGeom& g, hipStream_t st) {
    long total = (long)batch * g.out_h * g.out_w;
    dim3 grid(cdiv(total, 256)), block(256);
    if (out_fmt == D2S_FMT_U8_HWC)
        hipLaunchKernelGGL((stereo_warp_generic<IN_FMT, D2S_FMT_U8_HWC>), grid, block, 0, st, rgb, depth, out, batch, g);
    else if (out_fmt == D2S_FMT_F32_HWC)
        hipLaunchKernelGGL((stereo_warp_generic<IN_FMT, D2S_FMT_F32_HWC>), grid, block, 0, st, rgb, depth, out, batch, g);
    else
        hipLaunchKernelGGL((stereo_warp_generic<IN_FMT, D2S_FMT_F32_CHW>), grid, block, 0, st, rgb, depth, out, batch, g);
}

// `force_generic` (env D2S_WARP_GENERIC=1) lets tests compare the two paths.
extern "C" int d2s_make_sbs(const void* rgb, int rgb_fmt, const float* depth, int dh, int dw, int batch, int H, int W,
                            const d2s_sbs_params* p, void* out, int out_fmt, void* stream) {
    D2S_REQUIRE(rgb && depth && out && p, "null pointer");
    D2S_REQUIRE(batch > 0 && H > 0 && W > 0 && dh > 0 && dw > 0, "bad shape");
    D2S_REQUIRE(out_fmt == D2S_FMT_U8_HWC || out_fmt == D2S_FMT_F32_HWC || out_fmt == D2S_FMT_F32_CHW, "bad out_fmt");
    WarpGeom g;
    if (make_geom(H, W, dh, dw, p, g) != D2S_OK) { set_error("d2s_make_sbs: bad display_mode"); return D2S_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    static const bool force_generic = getenv("D2S_WARP_GENERIC") && atoi(getenv("D2S_WARP_GENERIC")) != 0;
    bool nopad = (g.Hp == H && g.Wp == W);
    bool fast_ok = !force_generic && rgb_fmt == D2S_FMT_U8_HWC && out_fmt == D2S_FMT_U8_HWC && nopad && ((long)batch * H * cdiv(W, 256) < (1L << 30)) &&
                   (W % 4 == 0) && dw <= W && dh <= H && ((uintptr_t)rgb % 4 == 0) && ((uintptr_t)out % 4 == 0) &&
                   ((long)H * W * 3 % 4 == 0);
    if (fast_ok && g.mode == D2S_MODE_HALF_TAB && (H % 2 != 0)) fast_ok = false;
    // round 6: the gather kernel (wave-private LDS window, fixed-point blend) serves every display mode when a lane's 4 pixels touch at most
    // 3 (3 dw < W: model-resolution depth under 1080p and larger frames) or 4 (3 dw < 2 W: 720p) depth columns, and when the depth map
    // has the frame's size (the drop-in make_sbs(rgb, depth[H, W]) surface); everything else -- other formats, padding, W % 4 != 0, a depth
    // grid between 2/3 of the frame's width and the frame's width -- takes the generic kernel.  D2S_WARP_GATHER=0: generic too (tests)
    static EnvInt gather_env{"D2S_WARP_GATHER", 1};
    const bool direct = dw == W && dh == H;
    const int nc = direct ? 4 : (3L * dw < W ? 3 : (3L * dw < 2L * W ? 4 : 0));
    if (fast_ok && gather_env.get() && nc && dw >= nc && W >= 8 && W < 16384 && (long)batch * dh * dw * 4 < (1L << 32)) {
        const int ntx = cdiv(W, 256);
        const long rows = (long)H * batch;
        // resident waves per CU the grid is cut for: 32 (tuned at batch 32); a launch of <= 12 288 row tiles (one 1080p frame: 8 640) is cut
        // for 12 -- three rows per wave instead of two amortise the per-wave set-up: 10.5 -> 9.7 us at 1080p batch 1 (profiles/r6_01)
        static EnvInt wpc_env{"D2S_WARP_WPC", 0};
        const long wpc = wpc_env.get() > 0 ? wpc_env.get() : (rows * ntx <= 12288 ? 12 : 32);
        long rpw = cdiv(rows * ntx, 256L * wpc);
        if (g.mode == D2S_MODE_HALF_TAB) rpw += rpw & 1;           // whole row pairs (H is even)
        const long waves = cdiv(rows, rpw) * ntx;
        const dim3 grid((unsigned)cdiv(waves, 4L)), block(256);
        // register budget: 128 VGPRs (4 waves per SIMD): two window sets in flight + the row loop without a spill.  At 96 (5 waves) the Full-TAB /
        // Half modes reload loop constants from scratch every row -- a scratch reload queues behind the window prefetch in the same in-order
        // vmcnt -- and Full-SBS gains nothing (155-159 us against 153-158; profiles/r6_01_warp_gather.md)
#define WG_LAUNCH_N(MODE_, NC_, DIR_) hipLaunchKernelGGL((stereo_warp_gather<MODE_, NC_, DIR_>), grid, block, 0, st, (const uint8_t*)rgb, depth, (uint8_t*)out, batch, g, (int)rpw, ntx)
#define WG_LAUNCH(MODE_) { if (direct) WG_LAUNCH_N(MODE_, 4, true); else if (nc == 3) WG_LAUNCH_N(MODE_, 3, false); else WG_LAUNCH_N(MODE_, 4, false); }
        if (g.mode == D2S_MODE_FULL_SBS) WG_LAUNCH(D2S_MODE_FULL_SBS)
        else if (g.mode == D2S_MODE_FULL_TAB) WG_LAUNCH(D2S_MODE_FULL_TAB)
        else if (g.mode == D2S_MODE_HALF_SBS) WG_LAUNCH(D2S_MODE_HALF_SBS)
        else WG_LAUNCH(D2S_MODE_HALF_TAB)
#undef WG_LAUNCH
#undef WG_LAUNCH_N
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    {
        if (rgb_fmt == D2S_FMT_U8_HWC) launch_generic<D2S_FMT_U8_HWC>(rgb, depth, out, out_fmt, batch, g, st);
        else if (rgb_fmt == D2S_FMT_U8_CHW) launch_generic<D2S_FMT_U8_CHW>(rgb, depth, out, out_fmt, batch, g, st);
        else if (rgb_fmt == D2S_FMT_F32_CHW) launch_generic<D2S_FMT_F32_CHW>(rgb, depth, out, out_fmt, batch, g, st);
        else { set_error("d2s_make_sbs: unsupported rgb_fmt"); return D2S_E_UNSUPPORTED; }
    }
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
