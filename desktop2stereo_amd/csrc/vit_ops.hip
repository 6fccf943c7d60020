// Small model-side kernels around the GEMMs (HBM / L2 bound):
//   patchify        Conv2d(3->D, k=s=14) as im2col rows            (HF Dinov2PatchEmbeddings)
//   cls_rows        cls token + pos[0] rows of the residual stream  (HF Dinov2Embeddings.forward)
//   layernorm       fp32 residual -> T, eps 1e-6, optional cls drop (HF Dinov2Layer norm1/norm2, Dinov2Backbone.layernorm)
//   bilinear_nhwc   align_corners=True up-sample of NHWC maps       (HF DepthAnythingFeatureFusionLayer / head)
//   head_final      conv3 (1x1, C->1) + ReLU | sigmoid*max_depth                      (HF DepthAnythingDepthEstimationHead)
#include "vit_ops.h"

namespace d2s {

template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt<bf16_t>(float v) { return f2bf(v); }
__device__ __forceinline__ float tof(float v) { return v; }
__device__ __forceinline__ float tof(bf16_t v) { return bf2f(v); }

// 4 consecutive outputs of one row; e4m3 rows are scaled by qscale = 1 / (activation scale) and saturate at +-448
__device__ __forceinline__ void store_row4(float* o, const float r[4], float) { *(float4*)o = make_float4(r[0], r[1], r[2], r[3]); }
__device__ __forceinline__ void store_row4(bf16_t* o, const float r[4], float) {
    uint2 t;
    t.x = (uint32_t)f2bf(r[0]) | ((uint32_t)f2bf(r[1]) << 16);
    t.y = (uint32_t)f2bf(r[2]) | ((uint32_t)f2bf(r[3]) << 16);
    *(uint2*)o = t;
}
__device__ __forceinline__ void store_row4(bx3_t* o, const float r[4], float) { bx3_store4(o, r); }      // pre-split A operand of a bf16x3 linear
__device__ __forceinline__ void store_row4(fp8_t* o, const float r[4], float qscale) {
    float a = fminf(fmaxf(r[0] * qscale, -FP8_MAX), FP8_MAX), b = fminf(fmaxf(r[1] * qscale, -FP8_MAX), FP8_MAX);
    float c = fminf(fmaxf(r[2] * qscale, -FP8_MAX), FP8_MAX), d = fminf(fmaxf(r[3] * qscale, -FP8_MAX), FP8_MAX);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    *(uint32_t*)o = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}

template <typename T>
__global__ void __launch_bounds__(256)
patchify_kernel(const float* __restrict__ x, T* __restrict__ A, int B, int h, int w, int p, int Kp,
                const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ resid, int N, int D) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int gh = h / p, gw = w / p, P = gh * gw;
    long total = (long)B * P * Kp;
    // the cls-token row of every frame (cls + pos[0], HF Dinov2Embeddings.forward) rides along here:
    // the patch-embed GEMM only writes rows 1..P of the residual stream
    if (cls && idx < (long)B * D) { int b = (int)(idx / D), d = (int)(idx % D); resid[(long)b * N * D + d] = cls[d] + pos[d]; }
    if (idx >= total) return;
    int k = (int)(idx % Kp);
    int row = (int)(idx / Kp);
    int b = row / P, pi = row % P;
    int py = pi / gw, px = pi % gw;
    float v = 0.f;
    if (k < 3 * p * p) {
        int c = k / (p * p), i = (k % (p * p)) / p, j = k % p;
        v = x[(((long)b * 3 + c) * h + py * p + i) * w + px * p + j];
    }
    A[idx] = cvt<T>(v);
}

__global__ void __launch_bounds__(256)
cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ resid,
                int B, int N, int D) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    int b = idx / D, d = idx % D;
    resid[(long)b * N * D + d] = cls[d] + pos[d];
}

// one wave per row; D <= 1024 (4 float4 per lane)
template <typename T>
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta,
                 T* __restrict__ out, int rows_out, int D, float eps, int rows_per_img, int img_rows, int row_off, float qscale) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows_out) return;
    long in_row = rows_per_img ? (long)(row / rows_per_img) * img_rows + (row % rows_per_img) + row_off : row;
    const float4* xr = (const float4*)(x + in_row * D);
    int nv = D >> 2;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = lane + 64 * i;
        if (c < nv) { v[i] = xr[c]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    float mu = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = lane + 64 * i;
        if (c < nv) { float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu; q += (a * a + b * b) + (cc * cc + d * d); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    float rstd = 1.0f / sqrtf(q / (float)D + eps);
    T* orow = out + (long)row * D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
            float4 gg = ((const float4*)g)[c], bb = ((const float4*)bta)[c];
            float r[4] = {(v[i].x - mu) * rstd * gg.x + bb.x, (v[i].y - mu) * rstd * gg.y + bb.y,
                          (v[i].z - mu) * rstd * gg.z + bb.z, (v[i].w - mu) * rstd * gg.w + bb.w};
            store_row4(orow + 4 * c, r, qscale);
        }
    }
}

// align_corners=True bilinear up-sample, NHWC.  One thread = one output pixel x one 16-byte channel chunk (8 bf16 / 4 f32): 4 x 16-B tap
// loads, one 16-B store.  Grid (x chunks of a row, output row, frame): the row's vertical tap is block-uniform and the column / chunk
// split is 32-bit arithmetic (round 2 decoded a flat 64-bit index per thread -- three emulated 64-bit divisions, ~2.5 TB/s of the
// 5+ this streaming pattern reaches).
// addend (optional, output-shaped): out = upsample(in) + addend -- the next fusion stage's "fused + RCU1(m)" sum
template <typename T>
__global__ void __launch_bounds__(256)
bilinear_nhwc_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                     float sy, float sx, const T* __restrict__ addend) {
    constexpr int CE = 16 / sizeof(T);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const int cc = C / CE;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const int ox = (int)(t / (unsigned)cc);
    if (ox >= Wo) return;
    const int c = (int)(t - (unsigned)ox * (unsigned)cc) * CE;
    const int oy = blockIdx.y, b = blockIdx.z;
    const Tap ty = linear_tap(oy, sy, Hi, true), tx = linear_tap(ox, sx, Wi, true);
    const T* base = in + (long)b * Hi * Wi * C + c;
    const T* r0 = base + (long)ty.i0 * Wi * C;
    const T* r1 = base + (long)ty.i1 * Wi * C;
    u32x4 v00 = *(const u32x4*)(r0 + tx.i0 * C);
    u32x4 v01 = *(const u32x4*)(r0 + tx.i1 * C);
    u32x4 v10 = *(const u32x4*)(r1 + tx.i0 * C);
    u32x4 v11 = *(const u32x4*)(r1 + tx.i1 * C);
    const T *p00 = (const T*)&v00, *p01 = (const T*)&v01, *p10 = (const T*)&v10, *p11 = (const T*)&v11;
    const long po = (((long)b * Ho + oy) * Wo + ox) * C + c;
    u32x4 r, av = {0u, 0u, 0u, 0u};
    if (addend) av = *(const u32x4*)(addend + po);
    const T* pa = (const T*)&av;
    T* o = (T*)&r;
#pragma unroll
    for (int k = 0; k < CE; ++k) {
        o[k] = cvt<T>(bilerp1(tx, ty, tof(p00[k]), tof(p01[k]), tof(p10[k]), tof(p11[k])) + tof(pa[k]));
    }
    *(u32x4*)(out + po) = r;
}

template <typename T>
__global__ void __launch_bounds__(256)
head_final_kernel(const T* __restrict__ x, const float* __restrict__ w3, float b3, float max_depth, float* __restrict__ depth,
                  long npix, int C) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const T* p = x + idx * C;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += tof(p[c]) * w3[c];
    depth[idx] = head_activation(acc + b3, max_depth);
}

template <typename T>
__global__ void __launch_bounds__(256)
to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, long n) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = tof(in[idx]);
}

#define DISPATCH_T(prec, CALL_BF, CALL_F32) do { if ((prec) == D2S_PREC_BF16) { CALL_BF; } else { CALL_F32; } } while (0)

int launch_patchify(int prec, const float* x, void* A, int B, int h, int w, int p, int Kp,
                    const float* cls, const float* pos, float* resid, int N, int D, hipStream_t st) {
    long total = (long)B * (h / p) * (w / p) * Kp;
    if (total < (long)B * D) total = (long)B * D;
    dim3 grid(cdiv(total, 256)), block(256);
    DISPATCH_T(prec, hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, block, 0, st, x, (bf16_t*)A, B, h, w, p, Kp, cls, pos, resid, N, D),
                     hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, st, x, (float*)A, B, h, w, p, Kp, cls, pos, resid, N, D));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_cls_rows(const float* cls, const float* pos, float* resid, int B, int N, int D, hipStream_t st) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3(cdiv((long)B * D, 256)), dim3(256), 0, st, cls, pos, resid, B, N, D);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_layernorm(int prec, const float* x, const float* g, const float* b, void* out, int rows_out, int D, float eps,
                     int rows_per_img, int img_rows, int row_off, hipStream_t st, float fp8_qscale, bool bx3_out) {
    if (D > 1024 || (D & 3)) { set_error("layernorm: D must be a multiple of 4 and <= 1024"); return D2S_E_UNSUPPORTED; }
    dim3 grid(cdiv(rows_out, 4)), block(256);
    if (bx3_out) {              // bf16x3 engines: the unit format, so the consuming linear's A operand can travel by LDS-DMA
        if (prec != D2S_PREC_FP32 || (D & 7)) { set_error("layernorm: bf16x3 output needs the fp32 engine and D % 8 == 0"); return D2S_E_UNSUPPORTED; }
        hipLaunchKernelGGL(layernorm_kernel<bx3_t>, grid, block, 0, st, x, g, b, (bx3_t*)out, rows_out, D, eps, rows_per_img, img_rows, row_off, 0.f);
    } else if (fp8_qscale > 0.f)       // e4m3 output for an fp8 linear: out = sat(LN(x) * qscale)
        hipLaunchKernelGGL(layernorm_kernel<fp8_t>, grid, block, 0, st, x, g, b, (fp8_t*)out, rows_out, D, eps, rows_per_img, img_rows, row_off, fp8_qscale);
    else
        DISPATCH_T(prec, hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, st, x, g, b, (bf16_t*)out, rows_out, D, eps, rows_per_img, img_rows, row_off, 0.f),
                         hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, st, x, g, b, (float*)out, rows_out, D, eps, rows_per_img, img_rows, row_off, 0.f));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

// max |x| over a tensor into *slot (float bits as uint: non-negative floats order like uints); calibration only
template <typename T>
__global__ void __launch_bounds__(256)
amax_kernel(const T* __restrict__ x, long n, unsigned* __restrict__ slot) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(tof(x[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}
int launch_amax(int prec, const void* x, long n, float* slot, hipStream_t st) {
    dim3 grid((unsigned)(n / 2048 > 1024 ? 1024 : (n / 2048 < 1 ? 1 : n / 2048))), block(256);
    DISPATCH_T(prec, hipLaunchKernelGGL(amax_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, n, (unsigned*)slot),
                     hipLaunchKernelGGL(amax_kernel<float>, grid, block, 0, st, (const float*)x, n, (unsigned*)slot));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_bilinear_nhwc(int prec, const void* in, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t st,
                         const void* addend) {
    float sy = linear_scale(Hi, Ho, true), sx = linear_scale(Wi, Wo, true);
    const int ce = prec == D2S_PREC_BF16 ? 8 : 4;
    if (C % ce) { set_error("bilinear_nhwc: channels must be a multiple of the 16-byte chunk"); return D2S_E_INVALID; }
    if (Ho > 65535 || B > 65535) { set_error("bilinear_nhwc: more than 65535 output rows / frames"); return D2S_E_UNSUPPORTED; }
    dim3 grid(cdiv((long)Wo * (C / ce), 256), Ho, B), block(256);
    DISPATCH_T(prec, hipLaunchKernelGGL(bilinear_nhwc_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)in, (bf16_t*)out, B, Hi, Wi, Ho, Wo, C, sy, sx, (const bf16_t*)addend),
                     hipLaunchKernelGGL(bilinear_nhwc_kernel<float>, grid, block, 0, st, (const float*)in, (float*)out, B, Hi, Wi, Ho, Wo, C, sy, sx, (const float*)addend));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_head_final(int prec, const void* x, const float* w3, float b3, float max_depth, float* depth, long npix, int C, hipStream_t st) {
    dim3 grid(cdiv(npix, 256)), block(256);
    DISPATCH_T(prec, hipLaunchKernelGGL(head_final_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, w3, b3, max_depth, depth, npix, C),
                     hipLaunchKernelGGL(head_final_kernel<float>, grid, block, 0, st, (const float*)x, w3, b3, max_depth, depth, npix, C));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

int launch_to_f32(int prec, const void* in, float* out, long n, hipStream_t st) {
    dim3 grid(cdiv(n, 256)), block(256);
    DISPATCH_T(prec, hipLaunchKernelGGL(to_f32_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)in, out, n),
                     hipLaunchKernelGGL(to_f32_kernel<float>, grid, block, 0, st, (const float*)in, out, n));
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

}  // namespace d2s
