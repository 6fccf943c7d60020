// Launchers of the non-GEMM model kernels (vit_ops.hip, attention.hip).
#pragma once
#include "common.h"

namespace d2s {

int launch_patchify(int prec, const float* x, void* A, int B, int h, int w, int p, int Kp,
                    const float* cls, const float* pos, float* resid, int N, int D, hipStream_t st);   // also writes the cls rows
// d2s_pipeline: pre-process + patchify in one launch (frame_ops.hip); D2S_E_UNSUPPORTED = not eligible, nothing launched
bool preprocess_patches_ok(int prec, int fmt, const d2s_pre_params* pre, int H, int W, int h, int w, int p, int Kp);
int launch_preprocess_patches(int prec, const void* frames, int fmt, int batch, int H, int W, int decim_stride, const d2s_pre_params* pre,
                              void* A, int h, int w, int p, int Kp, const float* cls, const float* pos, float* resid, int N, int D, hipStream_t st);
int launch_cls_rows(const float* cls, const float* pos, float* resid, int B, int N, int D, hipStream_t st);
int launch_layernorm(int prec, const float* x, const float* g, const float* b, void* out, int rows_out, int D, float eps,
                     int rows_per_img, int img_rows, int row_off, hipStream_t st, float fp8_qscale = 0.f,   // > 0: e4m3 output
                     bool bx3_out = false);                                                     // fp32 engines: the bf16x3 unit format
int launch_amax(int prec, const void* x, long n, float* slot, hipStream_t st);       // *slot = max(*slot, max |x|); fp8 calibration
int launch_bilinear_nhwc(int prec, const void* in, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t st,
                         const void* addend = nullptr);    // out = upsample(in) [+ addend]
int launch_head_final(int prec, const void* x, const float* w3, float b3, float max_depth, float* depth, long npix, int C, hipStream_t st);
int launch_to_f32(int prec, const void* in, float* out, long n, hipStream_t st);

// Multi-head self-attention over the fused QKV activation [B*N, 3*D] (q | k | v, head-major inside
// each), head_dim 64: out[B*N, D] = softmax(q k^T / 8) v.  (HF Dinov2SelfAttention.forward)
// vt: V transposed [B, heads, 64, Npad] (Npad = N rounded up to 64, zero beyond N), see MAP_QKV in gemm.h.
// fp8_qscale > 0 (bf16 inputs only): out is e4m3 = sat(result * fp8_qscale), the A operand of an fp8 output projection.
// prescaled: the q third of qkv already carries ATTN_SCALE_LOG2E (bf16 / fp8 engines fold it into W_q and b_q: one rounding, and
// the batched kernel gets log2-domain scores straight from the matrix pipe); false: the kernel applies it (fp32 parity class).
constexpr float ATTN_SCALE_LOG2E = 0.125f * 1.4426950408889634f;          // 64^-0.5 * log2(e)
// bx3_out (fp32 inputs only): out in the bf16x3 unit format (common.h), the pre-split A operand of a bf16x3 output projection.
int launch_attention(int prec, const void* qkv, const void* vt, void* out, int B, int N, int Npad, int heads, hipStream_t st,
                     float fp8_qscale = 0.f, bool prescaled = false, bool bx3_out = false);

// Video-Depth-Anything temporal-module kernels (temporal.hip)
int launch_groupnorm(int prec, const void* x, const float* g, const float* b, void* out, int sites, int C, int groups, float eps, hipStream_t st);
// cur [S, 3C] = k' | v' | q' of this frame; ring [slots][S][2C] = k' | v' of the past frames; ptab [32][3C] = pe @ W^T
int launch_cache_store(int prec, void* ring, const void* cur, int sites, int C, int slot0, int nslots, hipStream_t st);
int launch_temporal_attn(int prec, const void* cur, void* ring, const float* ptab, void* out, int sites, int C, int Tw, int slots,
                         int head, hipStream_t st, int store_slot = -1 /* >= 0: also write this frame's k' | v' rows into that ring slot */);
int launch_geglu(int prec, const void* u, void* g, long rows, int C4, hipStream_t st);
int launch_cast_f32(int prec, const float* in, void* out, long n, hipStream_t st);

// engine-internal (post.hip)
int ema_batch(float* depth, float* state, int initialised, int nframes, int hw, float alpha, hipStream_t st);

}  // namespace d2s
