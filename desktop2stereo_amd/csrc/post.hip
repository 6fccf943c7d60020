// Depth post-process at model resolution (latency-bound, tiny):
//   A10  normalize (subsample + order statistic, no lerp) -> gamma -> foreground scale
//        reference depth.py:816-867, 784-794, 775-776, 709-736
//   A11  anti_alias: separable Gaussian, zero padding, H then V     reference depth.py:740-765
//   A12  DepthStabilizer (EMA)                                      reference depth.py:1865-1887
#include "common.h"
#include <math.h>

namespace d2s {

constexpr int SORT_N = 8192;       // >= subsample_cap (6144), power of two
constexpr int SORT_THREADS = 1024;

__device__ __forceinline__ uint32_t f2key(float f) {          // order-preserving float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// One block per frame: v = depth.flatten()[::step] (m <= cap values, held in registers), then an exact
// radix select (4 passes x 8 bits, LDS histograms) of the two order statistics
//   bounds[b] = { s[tail-1], s[m-tail] }   (the tail-th smallest / largest; no lerp, depth.py:784-794).
//
// METRIC (is_metric(), depth.py:844-847): the values are inv = 1/max(d,1e-12) of the VALID pixels (d > 0) only, in
// row-major order, so the subsample positions depend on the data: a block-wide scan ranks the valid pixels, the
// step / m / tail follow from the valid count, and the r-th valid pixel is sample r/step iff r % step == 0.
__device__ __forceinline__ float metric_inverse(float d) { return d > 0.f ? 1.0f / fmaxf(d, 1e-12f) : d; }

template <bool METRIC>
__global__ void __launch_bounds__(SORT_THREADS)
percentile_bounds_kernel(const float* __restrict__ depth, int n, int step, int m, int tail, int cap, double lo_q,
                         float* __restrict__ bounds) {
    constexpr int PER = SORT_N / SORT_THREADS;
    __shared__ unsigned hist[2][256];
    __shared__ unsigned sel_rank[2];
    const float* d = depth + (long)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t key[PER];
    int nvalid = n;
    if constexpr (METRIC) {
        __shared__ uint32_t samp[SORT_N];
        __shared__ int wave_cnt[SORT_THREADS / 64];
        __shared__ int s_nv;
        const int chunk = (n + SORT_THREADS - 1) / SORT_THREADS;
        const int i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
        int cnt = 0;
        for (int i = i0; i < i1; ++i) cnt += d[i] > 0.f;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wave_cnt[wid] = incl;
        __syncthreads();
        int base = incl - cnt;
        for (int w = 0; w < wid; ++w) base += wave_cnt[w];
        if (tid == SORT_THREADS - 1) s_nv = base + cnt;
        __syncthreads();
        nvalid = s_nv;
        step = nvalid > cap ? (nvalid + cap - 1) / cap : 1;
        m = (nvalid + step - 1) / step;
        tail = (int)nearbyint(lo_q * (double)(m - 1)) + 1;
        tail = max(1, min(tail, m));
        int r = base;
        for (int i = i0; i < i1; ++i) {
            float v = d[i];
            if (v > 0.f) { if (r % step == 0) samp[r / step] = f2key(metric_inverse(v)); ++r; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) { int idx = tid + i * SORT_THREADS; key[i] = idx < m ? samp[idx] : 0u; }
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) { int idx = tid + i * SORT_THREADS; key[i] = idx < m ? f2key(d[(long)idx * step]) : 0u; }
    }
    // Range-adaptive radix select.  A fixed 8-bits-from-the-top radix puts all of a depth map's keys (same sign, one or two
    // exponents) into one or two bins of the first passes -- 6144 LDS atomics on the same address, serialised: 20 us for a
    // 24 KB problem.  Here a pass bins (key - lo) >> shift with [lo, lo + (256 << shift)) the range that still holds the
    // target, starting from the block's [min key, max key]: the first pass spreads over all 256 bins, later passes see only
    // the few keys of the chosen bin.  Exact: the binning is monotone in the key.
    __shared__ unsigned red_min[SORT_THREADS / 64], red_max[SORT_THREADS / 64];
    __shared__ unsigned sel_lo[2], sel_shift;
    {
        unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (tid + i * SORT_THREADS < m) { kmin = min(kmin, key[i]); kmax = max(kmax, key[i]); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o)); kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o)); }
        if (lane == 0) { red_min[wid] = kmin; red_max[wid] = kmax; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < SORT_THREADS / 64; ++w) { kmin = min(kmin, red_min[w]); kmax = max(kmax, red_max[w]); }
            if (m <= 0) { kmin = kmax = 0u; }
            const unsigned span = kmax - kmin;
            const int bits = span ? 32 - __clz((int)span) : 0;              // span < 2^bits
            sel_shift = bits > 8 ? bits - 8 : 0;
            sel_lo[0] = sel_lo[1] = kmin;
            bool all = tail >= m;                                   // depth.py:790-791: (min, max)
            sel_rank[0] = all ? 0 : tail - 1; sel_rank[1] = all ? m - 1 : m - tail;
        }
        __syncthreads();
    }
    while (true) {
        const unsigned shift = sel_shift;
        if (tid < 512) hist[tid >> 8][tid & 255] = 0;
        __syncthreads();
        const unsigned l0 = sel_lo[0], l1 = sel_lo[1];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (tid + i * SORT_THREADS < m) {
                const unsigned d0 = key[i] - l0, d1 = key[i] - l1;           // (unsigned: keys below lo wrap to huge values)
                if (key[i] >= l0 && (d0 >> shift) < 256u) atomicAdd(&hist[0][d0 >> shift], 1u);
                if (key[i] >= l1 && (d1 >> shift) < 256u) atomicAdd(&hist[1][d1 >> shift], 1u);
            }
        }
        __syncthreads();
        if (wid < 2) {                                           // wave r resolves rank r: 4 bins per lane
            unsigned c0 = hist[wid][4 * lane], c1 = hist[wid][4 * lane + 1], c2 = hist[wid][4 * lane + 2], c3 = hist[wid][4 * lane + 3];
            unsigned s = c0 + c1 + c2 + c3, incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            unsigned excl = incl - s, target = sel_rank[wid];
            if (target >= excl && target < incl) {
                unsigned r = target - excl, bin;
                if (r < c0) bin = 0; else if ((r -= c0) < c1) bin = 1; else if ((r -= c1) < c2) bin = 2; else { r -= c2; bin = 3; }
                sel_lo[wid] += (4 * lane + bin) << shift;
                sel_rank[wid] = r;
            }
        }
        __syncthreads();
        if (shift == 0) break;
        if (tid == 0) sel_shift = shift > 8 ? shift - 8 : 0;
        __syncthreads();
    }
    if (tid == 0) {
        float lo = key2f(sel_lo[0]), hi = key2f(sel_lo[1]);
        if (nvalid <= 10) { lo = 0.f; hi = 0.f; }                // depth.py:852-854
        bounds[2 * blockIdx.x] = lo;
        bounds[2 * blockIdx.x + 1] = hi;
    }
}

// x^y for x in [0, 1] on the transcendental pipe: exp2(y log2 x), v_log_f32 / v_exp_f32 (1 ulp each).  On this domain the result is
// <= 1 and |y log2 x| is small wherever the result is not: max |error| 1.0e-7 over [0, 1] for y in {1.45, 1/1.05, 2/3, 2, 1/2} against 5.9e-8
// for a correctly rounded powf (float64 reference, 2 M points) -- two orders below the 5e-6 the post-process is held
// to.  ocml's powf is ~200 instructions of double-float arithmetic; every pixel takes two, and the one-launch kernel re-shapes its
// window borders (2.7 x the pixels): 8 us of a 24 us kernel at batch 1 (rocprofv3, profiles/r5_*).  x = 0 -> 0 (y = 0 -> 1), like powf.
__device__ __forceinline__ float pow01(float x, float y) {
    const float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    return x > 0.f ? r : (y == 0.f ? 1.f : 0.f);
}
__device__ __forceinline__ float shape_depth(float d, float dmin, float dmax, float gamma, float fg_exp, bool fg_on) {
    float denom = fmaxf(dmax - dmin, 1e-6f);                                      // depth.py:865
    float nrm = fminf(fmaxf((d - dmin) / denom, 0.f), 1.f);
    float g = pow01(nrm, gamma);                                                  // depth.py:775-776
    g = fminf(fmaxf(g, 0.f), 1.f);                                                // depth.py:729
    if (!fg_on) return g;
    float dist = g - 0.5f;
    float sgn = dist > 0.f ? 1.f : (dist < 0.f ? -1.f : 0.f);
    float o = 0.5f + sgn * pow01(fabsf(dist), fg_exp);                            // depth.py:733-735
    return fminf(fmaxf(o, 0.f), 1.f);
}

constexpr int MAX_TAPS = 63;
struct GaussTaps { int k; float w[MAX_TAPS]; };

// normalise + gamma + foreground-scale fused with the horizontal Gaussian pass: one block per row,
// shaped row staged in LDS (zero padded), k taps out of LDS.
template <int KT>      // taps known at compile time (13 = the reference's default): unrolled tap loops, weights in SGPRs; 0 = run-time count
__global__ void __launch_bounds__(256)
shape_hblur_kernel(const float* __restrict__ depth, const float* __restrict__ bounds, float* __restrict__ tmp,
                   int h, int w, float gamma, float fg_exp, int fg_on, int metric, GaussTaps taps) {
    extern __shared__ float row[];                       // w + 2r
    int y = blockIdx.x % h, b = blockIdx.x / h;
    const int nt = KT ? KT : taps.k;
    int r = nt / 2;
    float dmin = bounds[2 * b], dmax = bounds[2 * b + 1];
    const float* src = depth + ((long)b * h + y) * w;
    for (int i = threadIdx.x; i < w + 2 * r; i += 256) {
        int x = i - r;
        float v = 0.f;
        if (x >= 0 && x < w) {
            v = src[x];
            if (metric) v = metric_inverse(v);                                     // depth.py:844-846
            v = shape_depth(v, dmin, dmax, gamma, fg_exp, fg_on != 0);
        }
        row[i] = v;
    }
    __syncthreads();
    float* dst = tmp + ((long)b * h + y) * w;
    for (int x = threadIdx.x; x < w; x += 256) {
        float acc = 0.f;
        if (nt >= 3) {
#pragma unroll
            for (int t = 0; t < nt; ++t) acc += taps.w[t] * row[x + t];
        } else acc = row[x + r];
        dst[x] = acc;
    }
}

template <int KT>
__global__ void __launch_bounds__(256)
vblur_kernel(const float* __restrict__ tmp, float* __restrict__ out, int B, int h, int w, GaussTaps taps) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * h * w) return;
    int x = (int)(idx % w);
    int y = (int)((idx / w) % h);
    int b = (int)(idx / ((long)w * h));
    const float* p = tmp + (long)b * h * w + x;
    const int nt = KT ? KT : taps.k;
    int r = nt / 2;
    // (compile-time tap count: all column loads are issued before the first product -- with a run-time count every tap was a scalar
    //  load of its weight plus a dependent global load, 13 round trips per pixel)
    float v[KT ? KT : 1];
    if constexpr (KT != 0) {
#pragma unroll
        for (int t = 0; t < KT; ++t) { const int yy = y + t - r; v[t] = (yy >= 0 && yy < h) ? p[(long)yy * w] : 0.f; }
    }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < nt; ++t) {
        int yy = y + t - r;
        if (yy >= 0 && yy < h) acc += taps.w[t] * (KT ? v[KT ? t : 0] : p[(long)yy * w]);
    }
    out[idx] = acc;
}

// Both passes in one launch (few frames: the two launches above are two launch floors for 0.6 MB): a block owns an 8 x 128 tile, shapes
// the (8 + 2r) x (128 + 2r) window it taps into LDS (zero outside the image, exactly what the two-pass form sees: the horizontal
// pass pads its row with zeros, the vertical pass skips rows outside the image), blurs horizontally into a second LDS plane and
// vertically out of it -- the same products in the same order, so the result is bit-identical to shape_hblur_kernel + vblur_kernel.
constexpr int SB_TR = 8, SB_TC = 128;
__global__ void __launch_bounds__(256)
shape_blur_kernel(const float* __restrict__ depth_in, const float* __restrict__ bounds, float* __restrict__ out,
                  int h, int w, float gamma, float fg_exp, int fg_on, int metric, GaussTaps taps) {
    extern __shared__ float sb_lds[];
    const int r = taps.k / 2, WR = SB_TC + 2 * r, HR = SB_TR + 2 * r;
    float* shp = sb_lds;                 // [HR][WR]
    float* hb = sb_lds + HR * WR;        // [HR][SB_TC]
    const int b = blockIdx.z, y0 = blockIdx.y * SB_TR, x0 = blockIdx.x * SB_TC;
    const float dmin = bounds[2 * b], dmax = bounds[2 * b + 1];
    const float* src = depth_in + (long)b * h * w;
    for (int i = threadIdx.x; i < HR * WR; i += 256) {
        const int ry = i / WR, rx = i - ry * WR;
        const int y = y0 + ry - r, x = x0 + rx - r;
        float v = 0.f;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            v = src[(long)y * w + x];
            if (metric) v = metric_inverse(v);
            v = shape_depth(v, dmin, dmax, gamma, fg_exp, fg_on != 0);
        }
        shp[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HR * SB_TC; i += 256) {
        const int ry = i / SB_TC, cx = i - ry * SB_TC;
        const float* row = shp + ry * WR + cx;
        float acc = 0.f;
        if (taps.k >= 3) { for (int t = 0; t < taps.k; ++t) acc += taps.w[t] * row[t]; }
        else acc = row[r];
        hb[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SB_TR * SB_TC; i += 256) {
        const int ty = i / SB_TC, cx = i - ty * SB_TC;
        const int y = y0 + ty, x = x0 + cx;
        if (y >= h || x >= w) continue;
        float acc = 0.f;
        for (int t = 0; t < taps.k; ++t) {
            const int yy = y + t - r;
            if (yy >= 0 && yy < h) acc += taps.w[t] * hb[(ty + t) * SB_TC + cx];
        }
        out[((long)b * h + y) * w + x] = acc;
    }
}

// ---- round 5: the whole post-process of a few frames in ONE launch (batch <= D2S_POST_FUSE_MAXB, relative models) --------------------
// percentile_bounds_kernel (one 1024-thread block per frame, 18 us) + shape_blur_kernel (23 us) were 45 us at the END of the batch-1
// critical path (3.8 % of the frame) for 0.6 MB of data.  Where the time went: (i) the select's four 8-bit passes are ~20 block barriers of
// 16 waves with single-thread sections between them; (ii) shape_blur's window loop issued one global load per iteration behind the
// previous iteration's two powf() -- eleven dependent memory round trips per thread.  Here every block of the (8 x 128)-tile grid
//   * requests its window pixels FIRST (<= 6 per thread, all in flight at once),
//   * derives the frame's bounds itself while they are in flight -- the same exact order statistics (any exact select returns the same
//     two values): range-adaptive radix select with 2048 bins per target (<= 3 passes), two alternating histogram buffers so a pass
//     costs two barriers, nothing done by one thread alone -- redundantly in every block (the subsample is 24 KB out of L2; 185 blocks
//     on 256 CUs: no block waits for another, no second launch),
//   * shapes the window (16 waves share the powf work a 256-thread block did alone), blurs horizontally and vertically out of LDS with
//     the products in shape_hblur_kernel / vblur_kernel's order: bit-identical to the separate launches.
constexpr int PF_THREADS = 1024, PF_TR = 8, PF_TC = 128, PF_MAXR = 8, PF_BITS = 11, PF_BINS = 1 << PF_BITS;
constexpr int PF_HR = PF_TR + 2 * PF_MAXR, PF_WR = PF_TC + 2 * PF_MAXR;
template <int KT>      // taps known at compile time (13 = the reference's default Anti-aliasing 4: the tap loops unroll and the weights stay
                        // in SGPRs -- with a run-time count every tap is an s_load from the kernarg segment inside the loop); 0 = run-time count
__global__ void __launch_bounds__(PF_THREADS)
post_fused_kernel(const float* __restrict__ depth_in, float* __restrict__ out, float* __restrict__ bounds_out, int h, int w, int step, int m, int tail,
                  float gamma, float fg_exp, int fg_on, GaussTaps taps) {
    constexpr int PER = SORT_N / PF_THREADS;
    __shared__ __attribute__((aligned(16))) unsigned lds_u[2 * 2 * PF_BINS];          // two buffers x two targets; the tile planes alias them afterwards
    __shared__ unsigned red_min[PF_THREADS / 64], red_max[PF_THREADS / 64];
    __shared__ unsigned sel_lo[2], sel_rank[2], sel_n[2], sel_cnt[2], sel_res[2];
    static_assert(sizeof(float) * (PF_HR * PF_WR + PF_HR * PF_TC) <= sizeof(unsigned) * 4 * PF_BINS, "tile planes alias the histograms");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.z, y0 = blockIdx.y * PF_TR, x0 = blockIdx.x * PF_TC;
    const int nt = KT ? KT : taps.k;
    const int r = nt / 2, WR = PF_TC + 2 * r, HR = PF_TR + 2 * r;
    const float* src = depth_in + (long)b * h * w;
    // (1) window pixels: wave `wid` owns window rows wid and wid + 16, lanes the columns lane, lane + 64, lane + 128
    float raw[2][3];
    bool inside[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ry = wid + 16 * i, rx = lane + 64 * j;
            const int y = y0 + ry - r, x = x0 + rx - r;
            inside[i][j] = ry < HR && rx < WR && y >= 0 && y < h && x >= 0 && x < w;
            raw[i][j] = inside[i][j] ? src[(long)y * w + x] : 0.f;
        }
    // (2) the frame's subsample
    uint32_t key[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int idx = tid + i * PF_THREADS; key[i] = idx < m ? f2key(src[(long)idx * step]) : 0u; }
    // both histogram buffers start at zero
    {
        typedef unsigned u4_ __attribute__((ext_vector_type(4)));
        ((u4_*)lds_u)[tid] = (u4_){0u, 0u, 0u, 0u};
        ((u4_*)lds_u)[tid + PF_THREADS] = (u4_){0u, 0u, 0u, 0u};
    }
    unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (tid + i * PF_THREADS < m) { kmin = min(kmin, key[i]); kmax = max(kmax, key[i]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o)); kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o)); }
    if (lane == 0) { red_min[wid] = kmin; red_max[wid] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w_ = 0; w_ < PF_THREADS / 64; ++w_) { kmin = min(kmin, red_min[w_]); kmax = max(kmax, red_max[w_]); }
#if defined(PF_CUT) && PF_CUT == 1              // (tuning aid, tools/build_variant.sh: the kernel ends after phase PF_CUT -- timing only, results wrong)
    if (kmin == 0x12345u) out[tid] = raw[0][0] + raw[1][2]; return;
#endif
    const unsigned span = kmax - kmin;
    const int bits = span ? 32 - __clz((int)span) : 0;                  // span < 2^bits
    int shift = bits > PF_BITS ? bits - PF_BITS : 0;
    const bool all = tail >= m;                                         // depth.py:790-791: (min, max)
    unsigned lo0 = kmin, lo1 = kmin, rk0 = all ? 0u : (unsigned)(tail - 1), rk1 = all ? (unsigned)(m - 1) : (unsigned)(m - tail);
    // ---- fast path (round 5): ONE value-linear histogram pass + ranking of the target bins' few candidates.  The radix passes below bin
    // key BITS: a depth map's keys share their exponent bits, so the first pass piles 6 000 keys on ~50 bins (serialised LDS atomics) and
    // three passes are needed (6.8 us of the 16 us kernel, profiles/r5_01).  Binning floor((v - vmin) * (BINS - 1) / (vmax - vmin)) is monotone
    // non-decreasing in the key order too (rounding is monotone), so the keys of lower bins are strictly smaller than those of the target's
    // bin and the order statistic is the (rank - lower count)-th smallest KEY of that bin: the same element the radix select returns.
    // Both targets share the histogram.  Taken when the value range is finite and both target bins hold <= PF_CAND keys (a map with
    // thousands of identical values -- ReLU zeros -- falls through to the radix passes).
    constexpr int PF_CAND = 512;
    bool done = false;
    {
        const float vmin = key2f(kmin), vmax = key2f(kmax), rng = vmax - vmin;
        if (kmin != kmax && rng > 0.f && rng < 3.0e38f && fabsf(vmin) < 3.0e38f) {                 // (block-uniform; NaN / Inf keys fail it)
            // bins 0 and BINS-1 hold ONLY the values equal to the minimum / maximum (a ReLU head leaves hundreds of exact zeros in the
            // subsample, a sigmoid head saturates at the top): a target that lands there is the minimum / maximum itself, no candidates
            const float bscale = (float)(PF_BINS - 2) / rng;
            auto bin_of = [&](uint32_t k) {
                const float v = key2f(k);
                if (v <= vmin) return 0u;
                if (v >= vmax) return (unsigned)(PF_BINS - 1);
                return 1u + (unsigned)min(max((int)((v - vmin) * bscale), 0), PF_BINS - 3);
            };
            unsigned* hist = lds_u;                                     // buffer 0, target-0 half: zero since the start of the kernel
            unsigned* cand = lds_u + 2 * PF_BINS;                       // buffer 1: 2 x PF_CAND candidate keys
            unsigned mybin[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                mybin[i] = 0xffffffffu;
                if (tid + i * PF_THREADS < m) { mybin[i] = bin_of(key[i]); atomicAdd(&hist[mybin[i]], 1u); }
            }
            if (tid < 2) sel_cnt[tid] = 0u;
            __syncthreads();
            if (wid < 2) {                                              // wave t resolves target t: 32 consecutive bins per lane
                typedef unsigned u4_ __attribute__((ext_vector_type(4)));
                const u4_* hp = (const u4_*)(hist + 32 * lane);
                u4_ c[8];
                unsigned s_ = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { c[q] = hp[q]; s_ += (c[q][0] + c[q][1]) + (c[q][2] + c[q][3]); }
                unsigned incl = s_;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
                const unsigned excl = incl - s_, target = wid == 0 ? rk0 : rk1;
                if (target >= excl && target < incl) {
                    unsigned rr = target - excl, bin = 0, cnt_b = 0;
                    bool found = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned cnt = c[q][k];
                            if (!found) { if (rr < cnt) { found = true; bin = 4 * q + k; cnt_b = cnt; } else rr -= cnt; }
                        }
                    const unsigned tb = 32u * lane + bin;
                    const bool edge = tb == 0u || tb == (unsigned)(PF_BINS - 1);
                    sel_lo[wid] = tb;                                   // (here: the target's BIN)
                    sel_rank[wid] = rr;
                    sel_n[wid] = edge ? 0u : cnt_b;                     // an edge bin's keys all equal the minimum / maximum value
                    sel_res[wid] = tb == 0u ? kmin : kmax;              // (overwritten by the ranking unless the bin is an edge bin)
                }
            }
            __syncthreads();
            const unsigned b0 = sel_lo[0], b1 = sel_lo[1], r0 = sel_rank[0], r1 = sel_rank[1], n0 = sel_n[0], n1 = sel_n[1];
            if (n0 <= (unsigned)PF_CAND && n1 <= (unsigned)PF_CAND) {   // (block-uniform)
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (n0 && mybin[i] == b0) cand[atomicAdd(&sel_cnt[0], 1u)] = key[i];
                    if (n1 && mybin[i] == b1) cand[PF_CAND + atomicAdd(&sel_cnt[1], 1u)] = key[i];
                }
                __syncthreads();
                // threads 0 .. n0-1 rank list 0, threads 512 .. 512+n1-1 rank list 1: the element with  #{u < v} <= r < #{u <= v}
                const int li = tid >> 9, ti = tid & 511;
                const unsigned nn = li ? n1 : n0, rr = li ? r1 : r0;
                if ((unsigned)ti < nn) {
                    const unsigned* L = cand + li * PF_CAND;
                    const unsigned v = L[ti];
                    unsigned less = 0, leq = 0;
                    for (unsigned u = 0; u < nn; ++u) { const unsigned x = L[u]; less += x < v; leq += x <= v; }
                    if (less <= rr && rr < leq) sel_res[li] = v;        // (equal keys write the same value)
                }
                __syncthreads();
                lo0 = sel_res[0]; lo1 = sel_res[1];
                done = true;
            } else {
                // fall through to the radix passes: they expect both histogram buffers at zero
                __syncthreads();
                typedef unsigned u4_ __attribute__((ext_vector_type(4)));
                ((u4_*)lds_u)[tid] = (u4_){0u, 0u, 0u, 0u};
                ((u4_*)lds_u)[tid + PF_THREADS] = (u4_){0u, 0u, 0u, 0u};
                __syncthreads();
            }
        }
    }
    if (!done)
    for (int pass = 0;; ++pass) {
        unsigned* hist = lds_u + (pass & 1) * (2 * PF_BINS);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (tid + i * PF_THREADS < m) {
                const unsigned d0 = key[i] - lo0, d1 = key[i] - lo1;          // (unsigned: keys below lo wrap to huge values)
                if (key[i] >= lo0 && (d0 >> shift) < (unsigned)PF_BINS) atomicAdd(&hist[d0 >> shift], 1u);
                if (key[i] >= lo1 && (d1 >> shift) < (unsigned)PF_BINS) atomicAdd(&hist[PF_BINS + (d1 >> shift)], 1u);
            }
        }
        if (pass >= 1) {                                                // the other buffer (pass - 1's, read for the last time before the barrier behind its scan)
            typedef unsigned u4_ __attribute__((ext_vector_type(4)));
            ((u4_*)(lds_u + ((pass + 1) & 1) * (2 * PF_BINS)))[tid] = (u4_){0u, 0u, 0u, 0u};
        }
        __syncthreads();
        if (wid < 2) {                                                  // wave t resolves target t: 32 consecutive bins per lane
            typedef unsigned u4_ __attribute__((ext_vector_type(4)));
            const u4_* hp = (const u4_*)(hist + wid * PF_BINS + 32 * lane);
            u4_ c[8];
            unsigned s_ = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { c[q] = hp[q]; s_ += (c[q][0] + c[q][1]) + (c[q][2] + c[q][3]); }
            unsigned incl = s_;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            const unsigned excl = incl - s_, target = wid == 0 ? rk0 : rk1;
            if (target >= excl && target < incl) {
                unsigned rr = target - excl, bin = 0;
                bool found = false;
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned cnt = c[q][k];
                        if (!found) { if (rr < cnt) { found = true; bin = 4 * q + k; } else rr -= cnt; }
                    }
                sel_lo[wid] = (wid == 0 ? lo0 : lo1) + ((32u * lane + bin) << shift);
                sel_rank[wid] = rr;
            }
        }
        __syncthreads();
        lo0 = sel_lo[0]; lo1 = sel_lo[1]; rk0 = sel_rank[0]; rk1 = sel_rank[1];
        if (shift == 0) break;
        shift = shift > PF_BITS ? shift - PF_BITS : 0;
    }
#if defined(PF_CUT) && PF_CUT == 2
    if (lo0 == 0x12345u) out[tid] = raw[0][0] + raw[1][2]; return;
#endif
    float dmin = key2f(lo0), dmax = key2f(lo1);
    if (h * w <= 10) { dmin = 0.f; dmax = 0.f; }                     // depth.py:852-854
    if (bounds_out && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { bounds_out[2 * b] = dmin; bounds_out[2 * b + 1] = dmax; }
    // (3) shape the window into LDS (zero outside the image: what the two-pass form's zero padding / skipped rows see)
    float* shp = (float*)lds_u;                  // [HR][WR]
    float* hb = shp + HR * WR;                   // [HR][PF_TC]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ry = wid + 16 * i, rx = lane + 64 * j;
            if (ry < HR && rx < WR) shp[ry * WR + rx] = inside[i][j] ? shape_depth(raw[i][j], dmin, dmax, gamma, fg_exp, fg_on != 0) : 0.f;
        }
    __syncthreads();
#if defined(PF_CUT) && PF_CUT == 3
    if (shp[tid] == 123.f) out[tid] = 1.f; return;
#endif
    // (4) horizontal pass: rows wid, wid + 16; columns lane, lane + 64
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ry = wid + 16 * i, cx = lane + 64 * j;
            if (ry < HR) {
                const float* row = shp + ry * WR + cx;
                float acc = 0.f;
                if (nt >= 3) {
#pragma unroll
                    for (int t = 0; t < nt; ++t) acc += taps.w[t] * row[t];
                } else acc = row[r];
                hb[ry * PF_TC + cx] = acc;
            }
        }
    __syncthreads();
#if defined(PF_CUT) && PF_CUT == 4
    if (hb[tid] == 123.f) out[tid] = 1.f; return;
#endif
    // (5) vertical pass: one output per thread
    {
        const int ty = wid >> 1, cx = (wid & 1) * 64 + lane;
        const int y = y0 + ty, x = x0 + cx;
        if (y < h && x < w) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < nt; ++t) {
                const int yy = y + t - r;
                if (yy >= 0 && yy < h) acc += taps.w[t] * hb[(ty + t) * PF_TC + cx];
            }
            out[((long)b * h + y) * w + x] = acc;
        }
    }
}

// EMA over `nframes` consecutive frames (recurrence in frame order); thread per pixel.
__global__ void __launch_bounds__(256)
ema_kernel(float* __restrict__ depth, float* __restrict__ state, int initialised, int nframes, int hw, float wgt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    float prev = initialised ? state[i] : 0.f;
    for (int f = 0; f < nframes; ++f) {
        float d = depth[(long)f * hw + i];
        if (!initialised && f == 0) prev = d;
        else prev = prev + wgt * (d - prev);                 // torch lerp_, weight < 0.5
        depth[(long)f * hw + i] = prev;
    }
    state[i] = prev;
}

}  // namespace d2s

using namespace d2s;

extern "C" uint64_t d2s_post_process_workspace(int batch, int h, int w) {
    return (uint64_t)batch * h * w * sizeof(float) + (uint64_t)batch * 2 * sizeof(float) + 256;
}

extern "C" int d2s_post_process_to(const float* depth_in, float* depth_out, int batch, int h, int w, const d2s_post_params* p,
                                   void* workspace, uint64_t workspace_bytes, void* stream) {
    const float* depth = depth_in;
    D2S_REQUIRE(depth_in && depth_out && p && workspace, "null pointer");
    D2S_REQUIRE(batch > 0 && h > 0 && w > 0, "bad shape");
    D2S_REQUIRE(workspace_bytes >= d2s_post_process_workspace(batch, h, w), "workspace too small");
    D2S_REQUIRE(p->subsample_cap > 0 && p->subsample_cap <= SORT_N, "subsample_cap must be <= 8192");
    D2S_REQUIRE(p->foreground_scale > -1.0f + 1e-12f, "scale must be greater than -1.0");   // depth.py:726-727
    hipStream_t st = (hipStream_t)stream;
    float* tmp = (float*)workspace;
    float* bounds = (float*)((char*)workspace + (((uint64_t)batch * h * w * sizeof(float) + 255) & ~255ull));
    int n = h * w;
    int step = 1, m = n;
    if (n > p->subsample_cap) { step = (n + p->subsample_cap - 1) / p->subsample_cap; m = (n + step - 1) / step; }
    double lo_q = fmax(0.0, fmin(1.0, (double)p->percentile / 100.0));
    int tail = (int)nearbyint(lo_q * (m - 1)) + 1;
    if (tail < 1) tail = 1;
    if (tail > m) tail = m;
    auto launch_percentile = [&]() {
        if (p->metric)
            hipLaunchKernelGGL(percentile_bounds_kernel<true>, dim3(batch), dim3(SORT_THREADS), 0, st, depth, n, step, m, tail,
                               p->subsample_cap, lo_q, bounds);
        else
            hipLaunchKernelGGL(percentile_bounds_kernel<false>, dim3(batch), dim3(SORT_THREADS), 0, st, depth, n, step, m, tail,
                               p->subsample_cap, lo_q, bounds);
    };
    // Gaussian taps: k = int(3 s) | 1, sigma = 0.5 s, float32 like the reference (depth.py:746-758)
    GaussTaps taps;
    int k = ((int)(3.0f * p->aa_strength)) | 1;
    D2S_REQUIRE(k <= MAX_TAPS, "aa_strength too large");
    taps.k = k >= 3 ? k : 1;
    if (k >= 3) {
        float sigma = 0.5f * p->aa_strength, sum = 0.f;
        for (int i = 0; i < k; ++i) { float c = (float)(i - k / 2); taps.w[i] = expf(-(c * c) / (2.f * sigma * sigma)); sum += taps.w[i]; }
        for (int i = 0; i < k; ++i) taps.w[i] /= sum;
    } else taps.w[0] = 1.f;
    int fg_on = fabsf(p->foreground_scale) >= 1e-6f;
    float fg_exp = 1.0f / (1.0f + p->foreground_scale);
    int r = taps.k / 2;
    // out of place and few frames: one launch for both passes (tiles tap their neighbours' inputs, so never in place)
    static EnvInt fuse_max{"D2S_POST_FUSE_MAXB", 2};        // (it re-shapes the window borders, 1.8 x the pow() work: +0.6 % at batch 1, -0.7 % at 4)
    const size_t sb_bytes = ((size_t)(SB_TR + 2 * r) * (SB_TC + 2 * r) + (size_t)(SB_TR + 2 * r) * SB_TC) * sizeof(float);
    {   // in place (equal pointers) or disjoint: a partially overlapping pair would be corrupted silently by either form below
        const char *i0 = (const char*)depth_in, *o0 = (const char*)depth_out;
        const size_t nb = (size_t)batch * h * w * sizeof(float);
        D2S_REQUIRE(depth_out == depth_in || i0 + nb <= o0 || o0 + nb <= i0, "d2s_post_process_to: depth_in and depth_out overlap without being equal");
    }
    // round 5: bounds + shape + both blurs in ONE launch (relative models, few frames): post_fused_kernel
    static EnvInt one_launch{"D2S_POST_ONE", 1};
    if (one_launch.get() && !p->metric && depth_out != depth_in && batch <= fuse_max.get() && r <= PF_MAXR && cdiv(h, PF_TR) <= 65535 && batch <= 65535) {
        if (taps.k == 13) hipLaunchKernelGGL(post_fused_kernel<13>, dim3(cdiv(w, PF_TC), cdiv(h, PF_TR), batch), dim3(PF_THREADS), 0, st,
                                             depth, depth_out, bounds, h, w, step, m, tail, p->gamma, fg_exp, fg_on, taps);
        else hipLaunchKernelGGL(post_fused_kernel<0>, dim3(cdiv(w, PF_TC), cdiv(h, PF_TR), batch), dim3(PF_THREADS), 0, st,
                                depth, depth_out, bounds, h, w, step, m, tail, p->gamma, fg_exp, fg_on, taps);
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    launch_percentile();
    if (depth_out != depth_in && batch <= fuse_max.get() && sb_bytes <= 64 * 1024 && cdiv(h, SB_TR) <= 65535 && batch <= 65535) {
        hipLaunchKernelGGL(shape_blur_kernel, dim3(cdiv(w, SB_TC), cdiv(h, SB_TR), batch), dim3(256), sb_bytes, st,
                           depth, bounds, depth_out, h, w, p->gamma, fg_exp, fg_on, p->metric != 0, taps);
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    if (taps.k == 13) {
        hipLaunchKernelGGL(shape_hblur_kernel<13>, dim3(batch * h), dim3(256), (w + 2 * r) * sizeof(float), st,
                           depth, bounds, tmp, h, w, p->gamma, fg_exp, fg_on, p->metric != 0, taps);
        hipLaunchKernelGGL(vblur_kernel<13>, dim3(cdiv((long)batch * h * w, 256)), dim3(256), 0, st, tmp, depth_out, batch, h, w, taps);
    } else {
        hipLaunchKernelGGL(shape_hblur_kernel<0>, dim3(batch * h), dim3(256), (w + 2 * r) * sizeof(float), st,
                           depth, bounds, tmp, h, w, p->gamma, fg_exp, fg_on, p->metric != 0, taps);
        hipLaunchKernelGGL(vblur_kernel<0>, dim3(cdiv((long)batch * h * w, 256)), dim3(256), 0, st, tmp, depth_out, batch, h, w, taps);
    }
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

extern "C" int d2s_post_process(float* depth, int batch, int h, int w, const d2s_post_params* p,
                                void* workspace, uint64_t workspace_bytes, void* stream) {
    return d2s_post_process_to(depth, depth, batch, h, w, p, workspace, workspace_bytes, stream);
}

extern "C" int d2s_ema_update(float* depth, float* state, int initialised, int h, int w, float alpha, void* stream) {
    D2S_REQUIRE(depth && state && h > 0 && w > 0, "bad argument");
    hipLaunchKernelGGL(ema_kernel, dim3(cdiv((long)h * w, 256)), dim3(256), 0, (hipStream_t)stream,
                       depth, state, initialised, 1, h * w, 1.0f - alpha);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

namespace d2s {
// engine-internal: EMA over a batch of frames in order
int ema_batch(float* depth, float* state, int initialised, int nframes, int hw, float alpha, hipStream_t st) {
    hipLaunchKernelGGL(ema_kernel, dim3(cdiv(hw, 256)), dim3(256), 0, st, depth, state, initialised, nframes, hw, 1.0f - alpha);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
}  // namespace d2s
