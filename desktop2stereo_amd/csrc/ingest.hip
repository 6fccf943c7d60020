// The two frame-side rows either end of the hot path (latency-bound, tiny):
//   A1   process(): BGR(A) uint8 HWC -> RGB float CHW, optional anti-aliased bilinear down-scale to even dims
//        reference depth.py:540-566 (the torch branch a ROCm device takes; F.interpolate(bilinear,
//        align_corners=False, antialias=True) == ATen _upsample_bilinear2d_aa: separable triangle filter of
//        support `scale`, weights normalised per output index, horizontal pass then vertical pass)
//   A15  overlay_fps(): 5x3 glyph text "FPS: %.1f" painted green on the source frame before the warp
//        reference depth.py:641-658 (glyph table), 2061-2103
#include "common.h"
#include <string.h>
#include <cmath>
#include <cstdlib>

namespace d2s {

// ATen aa-filter geometry for one output index (UpSampleKernel.cpp HelperInterpLinear, float math):
//   center = scale*(i+0.5); support = scale (>= 1, down-scale) or 1; taps [xmin, xmin+xsize)
struct AaTaps { int xmin, xsize; float center, invscale; };
__device__ __forceinline__ AaTaps aa_taps(int i, float scale, int in_size) {
    AaTaps t;
    float support = scale >= 1.f ? scale : 1.f;
    t.invscale = scale >= 1.f ? 1.0f / scale : 1.f;
    t.center = scale * ((float)i + 0.5f);
    int lo = (int)((double)(t.center - support) + 0.5);        // ATen adds the double literal 0.5 before truncating
    int hi = (int)((double)(t.center + support) + 0.5);
    t.xmin = lo > 0 ? lo : 0;
    t.xsize = (hi < in_size ? hi : in_size) - t.xmin;
    return t;
}
__device__ __forceinline__ float aa_weight(const AaTaps& t, int j) {
    float x = fabsf(((float)(j + t.xmin) - t.center + 0.5f) * t.invscale);
    return x < 1.f ? 1.f - x : 0.f;
}

// one thread = one output pixel (all 3 channels).  RESIZE == false: pure swizzle.
template <bool RESIZE>
__global__ void __launch_bounds__(256)
process_kernel(const uint8_t* __restrict__ src, int nch, int H0, int W0, float* __restrict__ out, int h, int w,
               float sy, float sx) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const long plane = (long)h * w;
    float r, g, b;
    if (!RESIZE) {
        const uint8_t* p = src + ((long)y * W0 + x) * nch;
        b = p[0]; g = p[1]; r = p[2];                                   // [..., :3].flip(-1)
    } else {
        AaTaps ty = aa_taps(y, sy, H0), tx = aa_taps(x, sx, W0);
        float wsx = 0.f, wsy = 0.f;
        for (int j = 0; j < tx.xsize; ++j) wsx += aa_weight(tx, j);
        for (int j = 0; j < ty.xsize; ++j) wsy += aa_weight(ty, j);
        r = g = b = 0.f;
        for (int jy = 0; jy < ty.xsize; ++jy) {
            const uint8_t* row = src + ((long)(ty.xmin + jy) * W0 + tx.xmin) * nch;
            float hr = 0.f, hg = 0.f, hb = 0.f;                          // horizontal pass of this source row
            for (int jx = 0; jx < tx.xsize; ++jx) {
                float wx = aa_weight(tx, jx) / wsx;
                hb += wx * (float)row[jx * nch + 0];
                hg += wx * (float)row[jx * nch + 1];
                hr += wx * (float)row[jx * nch + 2];
            }
            float wy = aa_weight(ty, jy) / wsy;                          // vertical pass over the horizontal results
            r += wy * hr; g += wy * hg; b += wy * hb;
        }
    }
    long o = (long)y * w + x;
    out[o] = r; out[plane + o] = g; out[2 * plane + o] = b;
}

// A1, the tensor branch of the NON-CUDA process() (reference depth.py:576-601): the capture tensor "is already RGB" -- first three
// channels, no flip -- and the down-scale is plain bilinear (align_corners=False, antialias=False).  FMT: D2S_FMT_U8_HWC (nch
// interleaved channels), D2S_FMT_U8_CHW / D2S_FMT_F32_CHW (planes).  One thread = one output pixel.  Same product order as ATen
// (and preprocess_kernel): wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d).
template <int FMT>
__global__ void __launch_bounds__(256)
process_rgb_kernel(const void* __restrict__ src, int nch, int H0, int W0, float* __restrict__ out, int h, int w, float sy, float sx) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const Tap ty = linear_tap(y, sy, H0, false), tx = linear_tap(x, sx, W0, false);
    const long plane = (long)h * w, splane = (long)H0 * W0;
    auto px = [&](int yy, int xx, int c) -> float {
        if constexpr (FMT == D2S_FMT_U8_HWC) return (float)((const uint8_t*)src)[((long)yy * W0 + xx) * nch + c];
        else if constexpr (FMT == D2S_FMT_U8_CHW) return (float)((const uint8_t*)src)[c * splane + (long)yy * W0 + xx];
        else return ((const float*)src)[c * splane + (long)yy * W0 + xx];
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = tx.w0 * px(ty.i0, tx.i0, c) + tx.w1 * px(ty.i0, tx.i1, c);
        const float bot = tx.w0 * px(ty.i1, tx.i0, c) + tx.w1 * px(ty.i1, tx.i1, c);
        out[c * plane + (long)y * w + x] = ty.w0 * top + ty.w1 * bot;
    }
}

// A1, the numpy branch of the NON-CUDA process() (reference depth.py:603-629): cv2.cvtColor(BGR(A) -> RGB) +
// cv2.resize(INTER_AREA) of a DOWN-scale, uint8 HWC in and out.  OpenCV's published algorithm (modules/imgproc/src/resize.cpp;
// third party -- opencv-python 4.12.0.88, requirements.txt:4 -- and not installed in this image: parity unpinned, see oracle
// resize_area_u8):  integer scale factors -> ResizeAreaFast (2 x 2: (a+b+c+d+2) >> 2; else saturate_cast<uchar>(int sum * (1.f / area)));
// otherwise ResizeArea_<uchar, float>: per source row the horizontal cells of a destination pixel accumulate in table order
// (buf += S * alpha, float), the rows then accumulate sum (+)= beta * buf, and saturate_cast<uchar>(sum) (round-half-even) ends the
// pixel.  The tables (computeResizeAreaTab: first partial cell, whole cells, last partial cell; weights formed in double, stored as
// float) are evaluated on the fly per thread -- one thread = one output pixel, all three channels.
struct AreaCells { int s0, n; float a_first, a_mid, a_last; bool has_first, has_last; };
__device__ __forceinline__ AreaCells area_cells(int d, double scale, int ssize) {
    AreaCells c;
    const double fsx1 = d * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    c.has_first = (double)sx1 - fsx1 > 1e-3;
    c.a_first = (float)(((double)sx1 - fsx1) / cell);
    c.a_mid = (float)(1.0 / cell);
    c.has_last = fsx2 - (double)sx2 > 1e-3;
    c.a_last = (float)(fmin(fmin(fsx2 - (double)sx2, 1.0), cell) / cell);
    c.s0 = sx1 - (c.has_first ? 1 : 0);
    c.n = (sx2 - sx1) + (c.has_first ? 1 : 0) + (c.has_last ? 1 : 0);
    return c;
}
__device__ __forceinline__ float area_alpha(const AreaCells& c, int k) {
    if (k == 0 && c.has_first) return c.a_first;
    if (k == c.n - 1 && c.has_last) return c.a_last;
    return c.a_mid;
}
__device__ __forceinline__ uint8_t sat_u8_rne(float v) {                  // saturate_cast<uchar>(float): cvRound (half to even), saturate
    float r = rintf(v);
    return (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}
template <int FAST>     // 0: general tables; 1: integer factors; 2: 2 x 2
__global__ void __launch_bounds__(256)
process_area_kernel(const uint8_t* __restrict__ src, int nch, int H0, int W0, uint8_t* __restrict__ out, int h, int w,
                    double sy, double sx, int iy, int ix) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    uint8_t* o = out + ((long)y * w + x) * 3;
    if constexpr (FAST != 0) {
        int sum[3] = {0, 0, 0};
        for (int jy = 0; jy < iy; ++jy) {
            const uint8_t* row = src + ((long)(y * iy + jy) * W0 + (long)x * ix) * nch;
            for (int jx = 0; jx < ix; ++jx) { sum[0] += row[jx * nch + 2]; sum[1] += row[jx * nch + 1]; sum[2] += row[jx * nch + 0]; }   // BGR -> RGB
        }
        if constexpr (FAST == 2) { o[0] = (uint8_t)((sum[0] + 2) >> 2); o[1] = (uint8_t)((sum[1] + 2) >> 2); o[2] = (uint8_t)((sum[2] + 2) >> 2); }
        else { const float sc = 1.f / (float)(ix * iy); o[0] = sat_u8_rne((float)sum[0] * sc); o[1] = sat_u8_rne((float)sum[1] * sc); o[2] = sat_u8_rne((float)sum[2] * sc); }
    } else {
        const AreaCells cy = area_cells(y, sy, H0), cx = area_cells(x, sx, W0);
        float sum[3] = {0.f, 0.f, 0.f};
        for (int jy = 0; jy < cy.n; ++jy) {
            const uint8_t* row = src + ((long)(cy.s0 + jy) * W0 + cx.s0) * nch;
            float buf[3] = {0.f, 0.f, 0.f};
            for (int jx = 0; jx < cx.n; ++jx) {
                const float a = area_alpha(cx, jx);
                buf[0] = buf[0] + (float)row[jx * nch + 2] * a;
                buf[1] = buf[1] + (float)row[jx * nch + 1] * a;
                buf[2] = buf[2] + (float)row[jx * nch + 0] * a;
            }
            const float b = area_alpha(cy, jy);
            if (jy == 0) { sum[0] = b * buf[0]; sum[1] = b * buf[1]; sum[2] = b * buf[2]; }
            else { sum[0] += b * buf[0]; sum[1] += b * buf[1]; sum[2] += b * buf[2]; }
        }
        o[0] = sat_u8_rne(sum[0]); o[1] = sat_u8_rne(sum[1]); o[2] = sat_u8_rne(sum[2]);
    }
}
// no resize: cv2.cvtColor only
__global__ void __launch_bounds__(256)
process_swizzle_u8_kernel(const uint8_t* __restrict__ src, int nch, long npix, uint8_t* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint8_t* p = src + i * nch;
    out[i * 3 + 0] = p[2]; out[i * 3 + 1] = p[1]; out[i * 3 + 2] = p[0];
}

constexpr int MAX_TEXT = 32;
struct GlyphText { int n; uint16_t bits[MAX_TEXT]; };   // 15 bits per glyph: row-major, bit 14 = top-left

// one thread = one pixel of the text box
__global__ void __launch_bounds__(256)
overlay_kernel(void* __restrict__ rgb, int fmt, int H, int W, int scale, GlyphText t) {
    const int char_w = 3 * scale, char_h = 5 * scale, pitch = char_w + scale, margin = 2 * scale;
    int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y;
    if (bx >= t.n * pitch || by >= char_h) return;
    int ci = bx / pitch, cx = bx % pitch;
    int x = margin + bx, y = margin + by;
    if (cx >= char_w || x >= W || y >= H) return;
    int bit = (by / scale) * 3 + cx / scale;
    if (!((t.bits[ci] >> (14 - bit)) & 1)) return;
    const long plane = (long)H * W, o = (long)y * W + x;
    // rgb * (1 - alpha) + color * alpha with alpha in {0,1}, color = (0, 255, 0)       depth.py:2101-2103
    if (fmt == D2S_FMT_U8_HWC) { uint8_t* p = (uint8_t*)rgb + o * 3; p[0] = 0; p[1] = 255; p[2] = 0; }
    else if (fmt == D2S_FMT_U8_CHW) { uint8_t* p = (uint8_t*)rgb; p[o] = 0; p[plane + o] = 255; p[2 * plane + o] = 0; }
    else if (fmt == D2S_FMT_F32_HWC) { float* p = (float*)rgb + o * 3; p[0] = 0.f; p[1] = 255.f; p[2] = 0.f; }
    else { float* p = (float*)rgb; p[o] = 0.f; p[plane + o] = 255.f; p[2 * plane + o] = 0.f; }
}

// the reference's 5x3 font (depth.py:641-658), one 15-bit word per glyph
static uint16_t glyph_bits(char c) {
    switch (c) {
        case '0': return 0b111101101101111; case '1': return 0b010110010010111; case '2': return 0b111001111100111;
        case '3': return 0b111001111001111; case '4': return 0b101101111001001; case '5': return 0b111100111001111;
        case '6': return 0b111100111101111; case '7': return 0b111001010100100; case '8': return 0b111101111101111;
        case '9': return 0b111101111001111; case 'F': return 0b111100110100100; case 'P': return 0b110101110100100;
        case 'S': return 0b111100111001111; case ':': return 0b000010000010000; case '.': return 0b000000000000010;
        default:  return 0;                                             // unknown characters render as ' ' (depth.py:2076)
    }
}

}  // namespace d2s

using namespace d2s;

extern "C" int d2s_process_shape(int H0, int W0, int target_height, int* out_h, int* out_w) {
    D2S_REQUIRE(out_h && out_w && H0 > 0 && W0 > 0 && target_height > 0, "bad argument");
    if (target_height >= H0) { *out_h = H0; *out_w = W0; return D2S_OK; }           // depth.py:551-552
    *out_h = (target_height / 2) * 2;                                                 // depth.py:554
    *out_w = (int)((double)W0 * (double)target_height / (double)H0) / 2 * 2;          // depth.py:555 (Python float division, int())
    D2S_REQUIRE(*out_h > 0 && *out_w > 0, "target height too small");
    return D2S_OK;
}

extern "C" int d2s_process(const uint8_t* bgr, int channels, int H0, int W0, int target_height, float* out, void* stream) {
    D2S_REQUIRE(bgr && out, "null pointer");
    D2S_REQUIRE(channels == 3 || channels == 4, "process(): frame must be HWC with 3 (BGR) or 4 (BGRA) channels");
    int h, w;
    int rc = d2s_process_shape(H0, W0, target_height, &h, &w);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(w, 256), h), block(256);
    if (h == H0 && w == W0 && target_height >= H0)
        hipLaunchKernelGGL(process_kernel<false>, grid, block, 0, st, bgr, channels, H0, W0, out, h, w, 1.f, 1.f);
    else        // area_pixel_compute_scale(align_corners=False, no scale_factor): in / out
        hipLaunchKernelGGL(process_kernel<true>, grid, block, 0, st, bgr, channels, H0, W0, out, h, w,
                           (float)H0 / (float)h, (float)W0 / (float)w);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

extern "C" int d2s_process_rgb(const void* rgb, int fmt, int channels, int H0, int W0, int target_height, float* out, void* stream) {
    D2S_REQUIRE(rgb && out, "null pointer");
    D2S_REQUIRE(channels >= 3 && channels <= 4, "process(): tensor frame must have 3 or 4 channels");
    D2S_REQUIRE(fmt == D2S_FMT_U8_HWC || fmt == D2S_FMT_U8_CHW || fmt == D2S_FMT_F32_CHW, "process(): unsupported tensor layout");
    int h, w;
    int rc = d2s_process_shape(H0, W0, target_height, &h, &w);          // same even-size rule (depth.py:591-593)
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(w, 256), h), block(256);
    const float sy = linear_scale(H0, h, false), sx = linear_scale(W0, w, false);
    if (fmt == D2S_FMT_U8_HWC) hipLaunchKernelGGL(process_rgb_kernel<D2S_FMT_U8_HWC>, grid, block, 0, st, rgb, channels, H0, W0, out, h, w, sy, sx);
    else if (fmt == D2S_FMT_U8_CHW) hipLaunchKernelGGL(process_rgb_kernel<D2S_FMT_U8_CHW>, grid, block, 0, st, rgb, channels, H0, W0, out, h, w, sy, sx);
    else hipLaunchKernelGGL(process_rgb_kernel<D2S_FMT_F32_CHW>, grid, block, 0, st, rgb, channels, H0, W0, out, h, w, sy, sx);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

extern "C" int d2s_process_area_shape(int H0, int W0, int target_height, int* out_h, int* out_w) {
    D2S_REQUIRE(out_h && out_w && H0 > 0 && W0 > 0 && target_height > 0, "bad argument");
    if (target_height >= H0) { *out_h = H0; *out_w = W0; return D2S_OK; }           // depth.py:621-622
    *out_h = target_height;                                                           // no even rounding on this branch
    *out_w = (int)((double)W0 * (double)target_height / (double)H0);                  // depth.py:612
    D2S_REQUIRE(*out_w > 0, "target height too small");
    return D2S_OK;
}

extern "C" int d2s_process_area(const uint8_t* bgr, int channels, int H0, int W0, int target_height, uint8_t* out, void* stream) {
    D2S_REQUIRE(bgr && out, "null pointer");
    D2S_REQUIRE(channels == 3 || channels == 4, "process(): frame must be HWC with 3 (BGR) or 4 (BGRA) channels");
    int h, w;
    int rc = d2s_process_area_shape(H0, W0, target_height, &h, &w);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (target_height >= H0) {
        const long npix = (long)H0 * W0;
        hipLaunchKernelGGL(process_swizzle_u8_kernel, dim3((unsigned)cdiv(npix, 256L)), dim3(256), 0, st, bgr, channels, npix, out);
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    dim3 grid(cdiv(w, 256), h), block(256);
    // cv::resize forms inv_scale = dsize / ssize first and scale = 1. / inv_scale from it (1 ulp from ssize / dsize for some sizes, which
    // can flip the is_area_fast test below or move a cell boundary of the tables)
    const double sx = 1.0 / ((double)w / (double)W0), sy = 1.0 / ((double)h / (double)H0);
    const int ix = (int)nearbyint(sx), iy = (int)nearbyint(sy);                      // saturate_cast<int>(double)
    const bool fast = std::abs(sx - ix) < 2.220446049250313e-16 && std::abs(sy - iy) < 2.220446049250313e-16;
    if (fast && ix == 2 && iy == 2) hipLaunchKernelGGL(process_area_kernel<2>, grid, block, 0, st, bgr, channels, H0, W0, out, h, w, sy, sx, iy, ix);
    else if (fast) hipLaunchKernelGGL(process_area_kernel<1>, grid, block, 0, st, bgr, channels, H0, W0, out, h, w, sy, sx, iy, ix);
    else hipLaunchKernelGGL(process_area_kernel<0>, grid, block, 0, st, bgr, channels, H0, W0, out, h, w, sy, sx, iy, ix);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

extern "C" int d2s_overlay_text(void* rgb, int fmt, int H, int W, const char* text, void* stream) {
    D2S_REQUIRE(rgb && text && H > 0 && W > 0, "bad argument");
    D2S_REQUIRE(fmt >= D2S_FMT_U8_HWC && fmt <= D2S_FMT_U8_CHW, "bad format");
    size_t n = strlen(text);
    D2S_REQUIRE(n <= (size_t)MAX_TEXT, "overlay text too long");
    GlyphText t; t.n = (int)n;
    for (size_t i = 0; i < n; ++i) t.bits[i] = glyph_bits(text[i]);
    int scale = H / 60; scale = scale < 1 ? 1 : (scale > 8 ? 8 : scale);            // depth.py:2080
    if (n == 0) return D2S_OK;
    dim3 grid(cdiv(t.n * 4 * scale, 256), 5 * scale), block(256);
    hipLaunchKernelGGL(overlay_kernel, grid, block, 0, (hipStream_t)stream, rgb, fmt, H, W, scale, t);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
