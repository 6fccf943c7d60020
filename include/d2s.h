/*
 * d2s.h -- C-ABI of libd2s_hip.so: the MI355X (gfx950) implementation of the
 * desktop2stereo per-frame hot path  depth inference -> depth post-process -> stereo warp.
 *
 * The reference has no FFI for this path; its boundary is the Python call surface of depth.py
 * plus an internal "engine object" protocol (DepthModelWrapper.__call__, reference
 * depth.py:1763-1781; MIGraphXEngine.__call__, depth.py:1029-1045: raw device pointers,
 * caller-allocated output, launched on torch's current hipStream, no host sync).  This header is
 * what a replacement engine exports under that protocol.  Conventions, mirroring the
 * reference's own ctypes->libamdhip64 usage (viewer.py:283-299: int status, != 0 -> RuntimeError):
 *
 *   - every function returns int: D2S_OK (0) or a D2S_E_* code; d2s_last_error() gives text;
 *   - all image / tensor pointers are DEVICE pointers owned by the caller (PyTorch-ROCm tensors'
 *     data_ptr()); the library allocates only weights, workspaces and per-stream state;
 *   - every call is asynchronous on the hipStream_t passed as `void* stream` (0 = null stream);
 *     there is no host synchronisation inside the library after engine finalisation;
 *   - no torch types, only plain pointers and sizes.
 *
 * Each entry point cites the reference interface it replaces (file:line in /root/reference).
 */
#ifndef D2S_H
#define D2S_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2S_OK              0
#define D2S_E_INVALID       1   /* bad argument / shape / mode */
#define D2S_E_STATE         2   /* call order (e.g. forward before finalize) */
#define D2S_E_HIP           3   /* a HIP runtime call failed */
#define D2S_E_MISSING       4   /* a weight tensor is missing or mis-shaped */
#define D2S_E_UNSUPPORTED   5

/* display modes of make_sbs_core (reference depth.py:2178-2183) */
#define D2S_MODE_HALF_SBS   0
#define D2S_MODE_FULL_SBS   1
#define D2S_MODE_HALF_TAB   2
#define D2S_MODE_FULL_TAB   3

/* pixel formats at the boundary */
#define D2S_FMT_U8_HWC      0   /* uint8  [H,W,3]   (numpy frame, reference depth.py:1921-1922) */
#define D2S_FMT_F32_CHW     1   /* float  [3,H,W]   (make_sbs_core tensor in/out, depth.py:2135-2138) */
#define D2S_FMT_F32_HWC     2   /* float  [H,W,3]   (make_sbs return, depth.py:767-773, 2231) */
#define D2S_FMT_U8_CHW      3   /* uint8  [3,H,W]   (rgb_tensor of predict_depth(return_tuple), depth.py:1922) */

/* arithmetic of the model stage */
#define D2S_PREC_FP32       0   /* f32 MFMA (v_mfma_f32_16x16x4_f32), fp32 activations: parity class */
#define D2S_PREC_BF16       1   /* bf16 MFMA, fp32 accumulate / residual / softmax / LayerNorm */
#define D2S_PREC_FP8        2   /* BASELINE config 3: the encoder's four linears per layer (75 % of the FLOPs) run on e4m3
                                   operands (OCP e4m3fn; weights per-output-channel scales, activations static per-tensor
                                   scales set by d2s_engine_calibrate), fp32 accumulate; everything else as D2S_PREC_BF16 */
#define D2S_PREC_BF16X3     3   /* split precision (parity class on the bf16 pipe): fp32 activations everywhere as D2S_PREC_FP32, but
                                   every GEMM / convolution operand is split x = hi + lo into two bf16 values (the activation on its way
                                   into LDS, the weights once at engine build) and the product is three bf16 MFMAs
                                   hi*hi + hi*lo + lo*hi with fp32 accumulation: ~16 mantissa bits per operand at 3/16 of the
                                   fp32-MFMA cost.  Attention, LayerNorm, softmax, residual stream: exactly the fp32 engine's. */
#define D2S_PREC_FP8_MLP    4   /* round 5: as D2S_PREC_FP8 with e4m3 operands on FC1 / FC2 ONLY (60 % of the encoder FLOPs); QKV and the
                                   attention-out projection stay on bf16 operands.  The error / throughput frontier of the e4m3 schemes
                                   (profiles/r5_fp8_frontier.md): all four linears 0.0202 mean |depth error| on BASELINE configs[2]'s
                                   frame, MLP only 0.0120 */

typedef struct d2s_engine d2s_engine;     /* opaque: weights + workspaces + per-stream state */

/* Depth-Anything-v2 architecture (HF DepthAnythingConfig). */
typedef struct d2s_model_desc {
    int32_t hidden;            /* D */
    int32_t heads;
    int32_t layers;
    int32_t out_indices[4];    /* 1-based layer taps */
    int32_t neck[4];           /* neck_hidden_sizes */
    int32_t fusion;            /* fusion_hidden_size */
    int32_t head_hidden;       /* 32 */
    int32_t mlp;               /* 4*D */
    int32_t patch;             /* 14 */
    int32_t pos_grid;          /* 37 */
    float   ln_eps;            /* 1e-6 */
    int32_t precision;         /* D2S_PREC_* */
    int32_t temporal;          /* 1: Video-Depth-Anything streaming head (4 temporal modules, 32-frame window; reference
                                  models/video_depth_anything/vda2_s.py, dpt_temporal.py); batch must be 1 */
    float   max_depth;         /* 0: relative head (conv3 -> ReLU).  > 0: metric head, sigmoid(conv3) * max_depth
                                  (HF depth_estimation_type="metric": 20 indoor / 80 outdoor; reference model ids
                                  utils.py:761-769) */
} d2s_model_desc;

/* Post-process constants (reference utils.py:858-859, depth.py:775, 816, 1889). */
typedef struct d2s_post_params {
    float   percentile;        /* 2.0 */
    int32_t subsample_cap;     /* 6144 */
    float   gamma;             /* 1.45 */
    float   foreground_scale;  /* FOREGROUND_SCALE = yaml/10 */
    float   aa_strength;       /* AA_STRENGTH = yaml*2 */
    float   ema_alpha;         /* 0.9 */
    int32_t metric;            /* is_metric() (reference depth.py:666-669): normalize() first inverts 1/d on d > 0
                                  and takes the order statistics over the valid values only (depth.py:844-847) */
} d2s_post_params;

/* Pre-processing constants of predict_depth (reference depth.py:676-706, 1794-1799, 1931-1948). */
#define D2S_RESAMPLE_BILINEAR    0   /* _resize_patch_aligned_t's CPU / DirectML branch (depth.py:700-706): [::stride] decimation,
                                        then bilinear align_corners=False -- what the reference's CPU path computes (parity default) */
#define D2S_RESAMPLE_BICUBIC_AA  1   /* its IS_CUDA branch (depth.py:698-699; IS_CUDA is true on a ROCm device): one bicubic +
                                        antialias F.interpolate from the full frame (ATen _upsample_bicubic2d_aa), no decimation */
typedef struct d2s_pre_params {
    float   mean[3];           /* (0.485, 0.456, 0.406); 0.5 for the dpt / zoedepth / depthpro ids (depth.py:1794-1799) */
    float   std[3];            /* (0.229, 0.224, 0.225) */
    int32_t resample;          /* D2S_RESAMPLE_* */
    int32_t square;            /* 0: aspect-preserving patch-aligned input (get_patch_size() = 14, depth.py:531-538).
                                  1: the fixed-square branch predict_depth takes when get_patch_size() is None -- CAPTURE_MODE ==
                                  "Window" (depth.py:532-533, 1937-1946): F.interpolate(bilinear, align_corners=False, no antialias)
                                  of the full frame to depth_resolution x depth_resolution on EVERY device (no decimation, `resample`
                                  is not consulted); the engine is depth_resolution x depth_resolution */
} d2s_pre_params;

/* Stereo parameters of make_sbs_core (reference depth.py:2122-2129). */
typedef struct d2s_sbs_params {
    double  ipd_uv;            /* 0.064; double because the reference forms ipd_uv*W in Python floats */
    float   depth_ratio;       /* 2.0 (function default) / DEPTH_STRENGTH */
    float   convergence;       /* 0.0 */
    int32_t display_mode;      /* D2S_MODE_* */
    int32_t fill_16_9;         /* 0/1 */
} d2s_sbs_params;

/* Uniforms of the reference's GLSL DIBR fragment shader (viewer.py:393-411) for d2s_dibr_warp. */
typedef struct d2s_dibr_params {
    double  ipd_uv;            /* viewer.ipd_uv (0.064): u_eye_offset = -/+ ipd_uv/2 for the left/right eye (viewer.py:2701, 2714) */
    float   depth_strength;    /* u_depth_strength = viewer.depth_strength (0.1) * depth_ratio (viewer.py:1334, 2686) */
    float   convergence;       /* u_convergence (viewer.py:1470) */
    float   roll;              /* u_roll, radians (xr_viewer/effects.py:1113); 0 in the desktop viewer */
    float   search_radius;     /* u_search_radius = 12 */
    float   depth_tolerance;   /* u_depth_tolerance = 0.012 */
    float   blur_radius;       /* u_blur_radius = 2.5 */
    float   res_w, res_h;      /* u_resolution; the reference never assigns it (pixel_size = 1/0 = +inf, viewer.py:395, 413): every
                                  tap at uv +- k * pixel_size then has a non-finite coordinate, which OpenGL leaves undefined
                                  (tests/golden/dibr.npz `as_shipped_*`: what SwiftShader renders for that state -- 6-17 % of the
                                  pixels more than one level from the intended image; recorded, not a target).
                                  0 here -> the source frame size, i.e. pixel_size = one texel: the shader's evident intent */
    int32_t display_mode;      /* D2S_MODE_*: how the two eye viewports are packed */
    int32_t feather_enabled;   /* u_feather_enabled */
    float   feather_width;     /* u_feather_width = 0.02 (viewer.py:1343) */
    float   corner_radius;     /* u_corner_radius (viewer.py:411, 617-624): rounded-box SDF over the quad's uv, alpha =
                                  1 - smoothstep(0, 0.01, sdf); 0 in the desktop viewer, 0.03 in the OpenXR screen
                                  (xr_viewer/implementation.py:293) */
    float   viewport[4];       /* u_viewport = (x, y, w, h) of the eye's viewport in pixels of the eye image, y up like
                                  gl_FragCoord (viewer.py:589: the feathering is relative to it); all 0: the eye image
                                  itself, (0, 0, out_w, out_h) */
    int32_t alpha_mode;        /* what becomes of frag_color.a (screen-edge clip :582, rounded corners :617-624), D2S_DIBR_ALPHA_*:
                                  the reference draws its stereo quads with GL_BLEND OFF (viewer.py:1304-1307 enable it around the
                                  overlay quad only), so its window shows frag_color.rgb as written = WINDOW (0, default);
                                  PREMULTIPLIED (1) = rgb * a, what an alpha-compositing consumer (the OpenXR layer) shows over
                                  black; RGBA (2) = frag_color itself, FOUR channels per pixel (rgb 0..255, a 0..255 / 0..1 for
                                  the u8 / f32 output formats) */
    uint32_t struct_size;      /* MUST be sizeof(d2s_dibr_params) = 80 (d2s_version() >= 110).  The struct grew from 72 to 80 bytes when
                                  alpha_mode was added; this field occupies what was tail padding, so the size is unchanged, and
                                  d2s_dibr_warp refuses anything else -- a caller built against the 72-byte header is rejected
                                  instead of having alpha_mode read from beyond its struct */
} d2s_dibr_params;
enum { D2S_DIBR_ALPHA_WINDOW = 0, D2S_DIBR_ALPHA_PREMULTIPLIED = 1, D2S_DIBR_ALPHA_RGBA = 2 };

const char* d2s_last_error(void);
int d2s_version(void);
/* Kernel-selection switches (D2S_NO_HALO2, D2S_NO_WIDE, ... -- tuning aids) are read from the environment once and
 * cached; this makes the library read them again (tests run both sides of a switch in one process).  Returns the new generation. */
int d2s_debug_reload_env(void);
/* gemm_pp's in-kernel K-split tail reduce never dead-locks: a unit that has waited 50 us for its partners leaves, and the last arrival
 * sums its share (same bits).  That degradation is COUNTED, not silent: *count = units that timed out on the current device since the
 * last clear (synchronise the stream first); tests/test_gpu_soak.py requires 0.   */
int d2s_debug_pp_tail_timeouts(int clear, unsigned* count);
/* 1 if the library was built with -DD2S_LDS_POISON (LDS rings pre-filled with NaN patterns: a debug build for the parity suite). */
int d2s_debug_lds_poison(void);

/* ---------------------------------------------------------------------------------------------
 * Engine life cycle -- replaces DepthModelWrapper construction + lazy engine build
 * (reference depth.py:1539-1631, 1784-1789, 1842-1862).
 * ------------------------------------------------------------------------------------------- */
int d2s_engine_create(const d2s_model_desc* desc, int device_id, d2s_engine** out);
/* Upload one tensor by its HF state-dict name (reference weight contract: SURVEY.md section 8c).  `host`
 * is host float32, C-contiguous, `shape`/`ndim` as in the checkpoint.  Packing (bf16 cast,
 * [N][K] padding, conv tap order) happens here, once. */
int d2s_engine_set_weight(d2s_engine* e, const char* name, const float* host,
                          const int64_t* shape, int ndim);
/* Fix the model-input shape (cf. _ensure_engine_built, depth.py:1842-1862): h, w multiples of
 * patch; allocates workspaces for up to max_batch frames and pre-interpolates the position
 * embedding (HF Dinov2Embeddings.interpolate_pos_encoding, bicubic).  Synchronous. */
int d2s_engine_finalize(d2s_engine* e, int h, int w, int max_batch);
int d2s_engine_destroy(d2s_engine* e);
/* bytes of device memory the engine holds (weights + workspaces) */
int d2s_engine_memory(const d2s_engine* e, uint64_t* bytes);

/* ---------------------------------------------------------------------------------------------
 * Stages.  Each mirrors one reference function; all device pointers, stream-ordered.
 * ------------------------------------------------------------------------------------------- */

/* A1: process(img, height)  (reference depth.py:540-566, the torch branch a ROCm/CUDA device takes):
 * bgr: uint8 HWC [H0,W0,channels], channels 3 (BGR) or 4 (BGRA) -> out: float RGB CHW [3,h,w], 0..255.
 * target_height >= H0: swizzle only ((h,w) = (H0,W0)); else F.interpolate(bilinear, align_corners=False,
 * antialias=True) to (h,w) = ((target//2)*2, (int(W0*target/H0)//2)*2) -- d2s_process_shape gives (h,w). */
int d2s_process_shape(int H0, int W0, int target_height, int* out_h, int* out_w);
int d2s_process(const uint8_t* bgr, int channels, int H0, int W0, int target_height, float* out, void* stream);

/* A1, the branches of the reference's NON-CUDA process() (depth.py:570-629; what a host without a CUDA/ROCm torch device runs):
 *  - tensor input (depth.py:576-601): the capture tensor is already RGB -- first three channels, no flip; target_height < H0:
 *    F.interpolate(bilinear, align_corners=False, antialias=False) to the even sizes of d2s_process_shape; else the frame as is.
 *    rgb: D2S_FMT_U8_HWC [H0,W0,channels] or D2S_FMT_U8_CHW / D2S_FMT_F32_CHW [channels,H0,W0]; out: float RGB CHW [3,h,w];
 *  - numpy input (depth.py:603-629): cv2.cvtColor(BGR(A) -> RGB), then cv2.resize(INTER_AREA) to
 *    (int(W0*target/H0), target) -- no even rounding; d2s_process_area_shape -- uint8 HWC in, uint8 RGB HWC out.  INTER_AREA is
 *    OpenCV's published box filter (resize.cpp: exact 2x2 / integer-factor averages, fractional cell weights otherwise). */
int d2s_process_rgb(const void* rgb, int fmt, int channels, int H0, int W0, int target_height, float* out, void* stream);
int d2s_process_area_shape(int H0, int W0, int target_height, int* out_h, int* out_w);
int d2s_process_area(const uint8_t* bgr, int channels, int H0, int W0, int target_height, uint8_t* out, void* stream);

/* A15: overlay_fps  (reference depth.py:2061-2103, glyphs depth.py:641-658): paints `text` (the reference's
 * "FPS: %.1f"; characters outside its 16-glyph table render as blanks) in green (0,255,0) with the 5x3 font scaled
 * by max(1,min(8,H//60)) at margin 2*scale, IN PLACE on one frame in any D2S_FMT_* layout. */
int d2s_overlay_text(void* rgb, int fmt, int H, int W, const char* text, void* stream);

/* A2-A4: ingest + _resize_patch_aligned_t + /255 + (x-mean)/std  (reference depth.py:676-706, 1916-1948).
 * pre->resample picks the branch of _resize_patch_aligned_t: D2S_RESAMPLE_BILINEAR = the CPU branch (strided decimation
 * by decim_stride = longest // (2*target), then bilinear align_corners=False), D2S_RESAMPLE_BICUBIC_AA = the IS_CUDA
 * branch (bicubic + antialias from the full frame; decim_stride is ignored).  pre->square: the fixed-square branch (depth.py:1937-1946):
 * plain bilinear of the full frame to (h, w), decim_stride and resample ignored.  pre == NULL: ImageNet mean / std, CPU branch.
 * frames: `batch` frames, format D2S_FMT_U8_HWC or D2S_FMT_U8_CHW or D2S_FMT_F32_CHW (0..255),
 * each H x W, contiguous.  out: float [batch,3,h,w] with (h,w) the engine shape. */
int d2s_preprocess(const void* frames, int fmt, int batch, int H, int W,
                   float* out, int h, int w, int decim_stride,
                   const d2s_pre_params* pre, void* stream);

/* A5-A9: model(pixel_values=x).predicted_depth  (reference depth.py:1763-1781 -> HF
 * DepthAnythingForDepthEstimation).  x: float [batch,3,h,w]; depth: float [batch,h,w]. */
int d2s_model_forward(d2s_engine* e, const float* x, float* depth, int batch, void* stream);

/* D2S_PREC_FP8 engines only: post-training calibration of the static per-tensor activation scales.  Runs one bf16
 * forward over x (float [batch,3,h,w], the engine's input layout) recording max |activation| at the four e4m3
 * quantisation sites of every encoder layer (LayerNorm-1 output, attention output, LayerNorm-2 output, GELU output),
 * sets scale = amax / 448 and switches the engine's encoder linears to e4m3 operands.  Synchronous.  Forward calls
 * before a successful calibration fail with D2S_E_STATE.  (No reference counterpart: the reference has FP16 only.) */
int d2s_engine_calibrate(d2s_engine* e, const float* x, int batch, void* stream);

/* A10-A11: post_process_depth = normalize -> gamma -> foreground_scale -> anti_alias
 * (reference depth.py:806-814), in place on float [batch,h,w].  Stateless. */
int d2s_post_process(float* depth, int batch, int h, int w, const d2s_post_params* p,
                     void* workspace, uint64_t workspace_bytes, void* stream);
uint64_t d2s_post_process_workspace(int batch, int h, int w);
/* The same, out of place (depth_out may equal depth_in; a PARTIAL overlap is rejected with D2S_E_INVALID): with distinct buffers and few frames the normalise / gamma / foreground
 * step and both blur passes run as one launch. */
int d2s_post_process_to(const float* depth_in, float* depth_out, int batch, int h, int w, const d2s_post_params* p,
                        void* workspace, uint64_t workspace_bytes, void* stream);

/* A12: DepthStabilizer.__call__ (reference depth.py:1865-1887).  state: float [h,w] owned by the
 * caller; *initialised == 0 -> state = depth (first frame), else state = lerp(state, depth, 1-alpha);
 * depth is overwritten with the returned value.  Frames are processed in order. */
int d2s_ema_update(float* depth, float* state, int initialised, int h, int w, float alpha,
                   void* stream);

/* A13: F.interpolate(depth, (H,W), bilinear, align_corners=False)  (reference depth.py:1999-2004).
 * in: float [batch,h,w] -> out: float [batch,H,W]. */
int d2s_upsample_depth(const float* in, int batch, int h, int w, float* out, int H, int W,
                       void* stream);

/* A14 (+A13 fused): make_sbs_core  (reference depth.py:2122-2184).
 * rgb: `batch` frames H x W in `rgb_fmt`; depth: float [batch,dh,dw] -- either full resolution
 * (dh==H, dw==W: exactly make_sbs_core) or model resolution (the bilinear align_corners=False
 * up-sample of predict_depth, depth.py:1999-2004, is fused into the warp).
 * out: `out_fmt` in {U8_HWC (round-half-even, saturate), F32_HWC, F32_CHW}; shape from d2s_sbs_shape. */
int d2s_make_sbs(const void* rgb, int rgb_fmt, const float* depth, int dh, int dw,
                 int batch, int H, int W, const d2s_sbs_params* p,
                 void* out, int out_fmt, void* stream);
/* output frame size of make_sbs_core for an H x W input (pad_to_aspect_tensor, depth.py:2106-2119) */
int d2s_sbs_shape(int H, int W, const d2s_sbs_params* p, int* out_h, int* out_w);

/* f1: the GLSL DIBR warp with disocclusion in-painting the reference's Viewer / OpenXR modes render
 * (FRAGMENT_SHADER, viewer.py:386-631): rgb uint8 HWC [batch,H,W,3], depth float [batch,H,W] (full resolution, as
 * uploaded to tex_depth, viewer.py:2386, 2456) -> both eyes, each rendered into an H x W viewport (Half modes:
 * W/2 columns resp. H/2 rows per eye) and packed left|right (SBS) or left over right (TAB); three channels per pixel
 * (four with p->alpha_mode == D2S_DIBR_ALPHA_RGBA).  out_fmt: D2S_FMT_U8_HWC (round-half-even) or D2S_FMT_F32_HWC (0..255);
 * shape from d2s_dibr_shape.  Pinned by tests/golden/dibr.npz: the reference's shader run off-screen (make_golden_dibr.py). */
int d2s_dibr_shape(int H, int W, int display_mode, int* out_h, int* out_w);
int d2s_dibr_warp(const uint8_t* rgb, const float* depth, int batch, int H, int W, const d2s_dibr_params* p,
                  void* out, int out_fmt, void* stream);

/* f3: the MJPEG sink of the Streamer modes.  Replaces `cv2.imencode('.jpg', bgr, [IMWRITE_JPEG_QUALITY, q])` on the
 * frame make_sbs returns (reference streamer.py:249-256, 285-291; quality = settings.yaml "Stream Quality",
 * utils.py:821): convertTo(CV_8U) (round-half-even, saturate) + baseline JPEG as libjpeg(-turbo) writes it with
 * jpeg_set_defaults + jpeg_set_quality(q, TRUE) -- YCbCr 4:2:0, slow-integer FDCT, Annex-K tables, no restart
 * markers, JFIF 1.01 -- BYTE-IDENTICAL to libjpeg-turbo's output for the same RGB frame.
 * frames: `batch` RGB frames, D2S_FMT_U8_HWC or D2S_FMT_F32_HWC (0..255), device.  out: device bytes, frame b at
 * out + b*out_stride; sizes[b] (device int32) = JPEG length, or -1 if it would not fit out_stride (nothing usable
 * is written then).  d2s_jpeg_bound gives a stride that can never overflow and the workspace bytes PER FRAME
 * (workspace: device, 256-byte aligned, >= batch * that).  Stream-ordered, no host synchronisation. */
int d2s_jpeg_bound(int H, int W, int64_t* out_bytes, int64_t* workspace_bytes);
int d2s_jpeg_encode(const void* frames, int fmt, int batch, int H, int W, int quality, uint8_t* out,
                    int64_t out_stride, int32_t* sizes, void* workspace, int64_t workspace_bytes, void* stream);

/* f4: device-resident hand-off of a produced frame to the display path -- replaces the viewer's per-frame
 * `stream.synchronize()` + hipGraphicsMapResources + hipMemcpy(D2D into the PBO) + unmap (reference viewer.py:1584-1712,
 * 2399-2428; CUDART_GL viewer.py:232-345).  The consumer lends a ring of its own device buffers (d2s_present_bind: any
 * device pointer, e.g. a mapped GL PBO; d2s_present_bind_gl_buffer: a GL buffer id, registered WRITE_DISCARD like
 * viewer.py:293 -- needs a current GL context); the producer acquires a slot, passes its pointer as `out` of d2s_pipeline /
 * d2s_make_sbs / d2s_upsample_depth, and publishes it; the consumer picks the latest published slot.  All waits are
 * HIP events between the two streams (consumer_stream = (void*)-1: host wait, for a GL consumer): no copy, no host
 * synchronisation on the producer side.  Producer and consumer may be different host threads.  Triple buffering: the
 * producer never takes the slot the consumer holds (between consume and release) and, while another slot is free, not the
 * latest published one either -- use >= 3 slots when both sides run concurrently (with 2, a consume may find nothing
 * published while the producer rewrites the only free slot). */
typedef struct d2s_present d2s_present;
int d2s_present_create(int device_id, int slots, d2s_present** out);
int d2s_present_bind(d2s_present* p, int slot, void* dev_ptr, uint64_t bytes);
int d2s_present_bind_gl_buffer(d2s_present* p, int slot, unsigned gl_buffer);
int d2s_present_acquire(d2s_present* p, void* producer_stream, int* slot, void** dev_ptr, uint64_t* bytes);
int d2s_present_publish(d2s_present* p, int slot, void* producer_stream);
/* producer: hand an acquired slot back unpublished (the frame was not produced); it becomes acquirable again, the consumer never sees it */
int d2s_present_cancel(d2s_present* p, int slot, void* producer_stream);
int d2s_present_consume(d2s_present* p, void* consumer_stream, int* slot, void** dev_ptr, uint64_t* seq);
int d2s_present_release(d2s_present* p, int slot, void* consumer_stream);
int d2s_present_destroy(d2s_present* p);

/* ---------------------------------------------------------------------------------------------
 * Fused frame pipeline: predict_depth + make_sbs for a batch of frames
 * (capture -> depth -> warp of reference main.py:232-262, 1336-1341 in one stream-ordered
 * call).  frames u8 HWC [batch,H,W,3]; out per d2s_sbs_shape in out_fmt; depth_full (optional,
 * may be NULL) receives predict_depth's return value, float [batch,H,W].
 * use_ema: run DepthStabilizer across the batch in frame order using the engine's stream state
 * (d2s_engine_reset_stream clears it). */
int d2s_pipeline(d2s_engine* e, const uint8_t* frames, int batch, int H, int W,
                 int depth_resolution, const d2s_pre_params* pre /* NULL: ImageNet constants, CPU-branch resize */,
                 const d2s_post_params* pp, const d2s_sbs_params* sp,
                 int use_ema, void* out, int out_fmt, float* depth_full, void* stream);
int d2s_engine_reset_stream(d2s_engine* e);

/* Per-kernel-class timing with HIP events on the launch stream (used by bench.py for the
 * roofline figures; not part of the throughput path).  enable=1 clears the records and starts
 * recording around every kernel launch of d2s_model_forward / d2s_pipeline; read() synchronises on
 * the recorded events and returns per class: total milliseconds, algorithmic FLOPs, algorithmic
 * bytes and launch count.  Arrays must hold >= 8 entries; class names via d2s_profile_class_name. */
int d2s_engine_profile(d2s_engine* e, int enable);
int d2s_engine_profile_read(d2s_engine* e, int max_classes, double* ms, double* flops, double* bytes,
                            int64_t* launches, int* n_classes);
const char* d2s_profile_class_name(int cls);

/* Debug / parity taps: copy an internal activation (after the last d2s_model_forward) to a
 * caller device buffer as float32.  name: "embeddings", "layer<N>" (engines created with D2S_TAPS=1), "neck_feat<i>".
 * rows/cols describe [rows, cols] of frame 0 (tokens x D, or pixels x C, NHWC). */
int d2s_engine_tap(d2s_engine* e, const char* name, float* out, uint64_t out_elems,
                   int* rows, int* cols, void* stream);

/* Stand-alone GEMM probe used by tests and the micro-benchmark:
 * C[M,N] = A[M,K] * W[N,K]^T (+bias), A/W given as float32 device arrays, computed with the
 * engine's MFMA kernel in the requested precision. */
int d2s_gemm_probe(const float* A, const float* Wt, const float* bias, float* C,
                   int M, int N, int K, int precision, int tile, int iters, void* stream);

/* Stand-alone attention probe (tests, micro-benchmark): out[B, N, heads*64] = softmax(q k^T / 8) v for float32 device
 * arrays q, k, v [B, heads, N, 64], through the engine's attention kernel for that shape (HF Dinov2SelfAttention.forward;
 * reference call site depth.py:1778).  iters > 1: *ms_per_iter (host) = mean launch time over iters - 1 repeats. */
int d2s_attention_probe(const float* q, const float* k, const float* v, float* out, int B, int heads, int N,
                        int precision, int iters, float* ms_per_iter, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D2S_H */
