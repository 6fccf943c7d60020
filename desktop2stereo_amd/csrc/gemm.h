// MFMA GEMM with implicit-conv A loader and fused epilogues (gemm.hip).
//   C[m, n] = epilogue( sum_k A[m, k] * W[n, k] )          W packed [Npad][Kpad], K contiguous
#pragma once
#include "common.h"
#include <cstring>

namespace d2s {

enum { A_PLAIN = 0, A_CONV3 = 1 };
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2,
       ACT_GEGLU = 3 };   // out[m, c] = x[m, c] * gelu(gate[m, c]) with the weight rows interleaved x0-3 | g0-3 | x4-7 | g4-7 ...: N/2 output columns
                           // (gemm_glds_kernel only: the exchange relies on its lane <-> column map, gemm_epi.h)
enum { MAP_ROWS = 0, MAP_SHUFFLE = 1, MAP_QKV = 2, MAP_HEAD = 3 };
enum { OUT_T = 0, OUT_F32 = 1, OUT_BF16 = 2, OUT_BX3 = 3 };   // OUT_T: operand type (fp8 operands: e4m3 of v * out_qscale; bf16x3: fp32)
                                                            // OUT_BX3 (bf16x3 launches): the unit format, A operand of the next linear

constexpr int D2S_PREC_FP8_OPERANDS = D2S_PREC_FP8;   // as a GEMM precision: e4m3 operands (A and W), fp32 accumulate
// D2S_PREC_BF16X3 as a GEMM precision: A is fp32 in memory (split into bf16 hi + lo on its way into LDS), W is packed in the
// bf16x3 unit format (gemm_epi.h), outputs / residuals of type OUT_T are fp32

struct GemmA {
    const void* ptr;      // T*
    int mode;             // A_PLAIN / A_CONV3
    long lda;             // A_PLAIN: row stride (elements)
    int Hi, Wi, C;        // A_CONV3: input NHWC [B,Hi,Wi,C], 3x3, pad 1
    int Ho, Wo, stride;   //          output grid, conv stride
    int relu;             // max(x,0) on load (pre-activation)
    int buf;              // set by the launcher (plain linears with whole K tiles, < 2 GiB operands): LDS-DMA through buffer descriptors
    int bx3;              // D2S_PREC_BF16X3 launches: A is ALREADY in the split unit format (plain rows): LDS-DMA tiles; 0: fp32, split when staged
    // A_CONV3 with the align_corners bilinear up-sample in front of it folded into the halo loader (the LDS-resident-input kernels; the
    // caller asks conv3_upsample_ok first): ptr = the SOURCE map [B, Hs, Ws, C], (Hi, Wi) = the up-sampled size the conv sees
    int ups, Hs, Ws;
    float usy, usx;       // linear_scale(Hs, Hi, true), linear_scale(Ws, Wi, true)
};

constexpr int GEMM_PART_CTR_WORDS = 256;   // counter words behind GemmEpi::part's part_elems partials (gemm_pp.hip: 4 per tail tile)
struct GemmEpi {
    void* out;
    int out_type;         // OUT_T: same type as the operands; OUT_F32
    long ldc;             // MAP_ROWS: row stride of out (elements)
    const float* bias;    // [N] or null
    const float* scale;   // [N] LayerScale or null: v = scale * (acc + bias)
    int act;
    const void* res1;     // residual(s), type = out type, added after scale/act
    const void* res2;
    // row mapping (MAP_ROWS): out_row = (m / rows_per_img) * img_rows + (m % rows_per_img) + row_off
    int rows_per_img, img_rows, row_off;   // rows_per_img == 0 -> out_row = m
    int res1_mod, res1_off;                // res1 row = res1_mod ? (m % res1_mod) + res1_off : out_row; ld = ldc
    // MAP_SHUFFLE (ConvTranspose k == s): m -> (b, y, x) over [B, gh, gw]; n -> (ky, kx, co)
    int map, gh, gw, ks, cout;
    // MAP_QKV: row-major [M, 3D] for q | k; the v third goes TRANSPOSED to vt[B, heads, 64, npad]
    // (m = b*ntok + t, n - 2D = h*64 + d) so the attention kernel reads V^T rows with 16-byte chunks.
    void* vt; int ntok, npad, qk_cols, heads;
    // split-K workspace (optional): fp32 partials [ksplit][M][N]; ksplit is chosen by the launcher.  The owner allocates
    // part_elems + GEMM_PART_CTR_WORDS words: the words behind the partials are gemm_pp's tail-tile counters (zero between launches)
    float* part; size_t part_elems; int ksplit;
    // MAP_HEAD: out = float depth[M]; bias = conv2 bias, scale = conv3 weights [N], head_b3 = conv3 bias
    float head_b3;
    float head_max_depth;                  // 0: ReLU; > 0: sigmoid * max_depth (metric head)
    // fp8 operands: acc is in units of (activation scale * weight scale[n]); deq[n] = s_act * s_w[n] turns it back
    // into real values before bias.  out_qscale = 1 / s_out for an e4m3 output (OUT_T with fp8 operands).
    const float* deq;
    float out_qscale;
    // LayerNorm folded into the linears either side of it (bf16 encoder; DESIGN.md §3.1 "LayerNorm fusion").
    // Producer (OUT_F32, MAP_ROWS -- the residual update): every stored value also goes to out2 as bf16 (same offsets),
    // and each block writes the (sum v, sum v^2) of its BN columns for every row: stats_out[column block][M][2]; the
    // launcher reports the number of column blocks through stats_slots (host int; the consumer takes at most 16).
    void* out2; float* stats_out; int* stats_slots;
    float out2_qscale;                            // 0: out2 is bf16; > 0: out2 is e4m3 of v * out2_qscale (the e4m3 encoder path)
    int out2_bx3;                                 // bf16x3 engines: out2 is the pre-split unit format (A operand of the LN-folded consumer)
    // Consumer: A rows are the RAW bf16 residual and W is gamma-folded, so LN(x) W = rstd * (x W' - mean * colsum(W'));
    // the kernel applies v = rstd[m] * (acc - mean[m] * ln_csum[n]) before bias (bias already holds b + W beta).
    const float* ln_stats; int ln_slots; const float* ln_csum; float ln_eps; int ln_dim;
    int ln_M;                                     // rows per slot of ln_stats when it is not this GEMM's M (0: M)
};

// precision: D2S_PREC_FP32 (T = float) / D2S_PREC_BF16 (T = bf16) / D2S_PREC_FP8_OPERANDS (T = e4m3).  tile: 0 = auto.
int launch_gemm(int precision, int tile, const GemmA& a, const void* W, int M, int N, int K, int Kpad,
                const GemmEpi& e, hipStream_t st);

// stride-1 3x3 convolutions with the input tile resident in LDS, second generation (conv3.hip): false = not eligible, nothing launched
bool launch_conv3_halo2(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st, bool dry = false);   // dry: eligibility only

// would launch_gemm fold the up-sample a.ups describes into this convolution's halo loader? (engine: skip the bilinear launch)
bool conv3_upsample_ok(int precision, int tile, const GemmA& a, int M, int N, int K, int Kpad, const GemmEpi& e);

// 256 x 256 ping-pong kernel (gemm_pp.hip): batched plain linears
bool pp_supported(int precision, const GemmA& a, int M, int N, int K, int Kpad, const GemmEpi& e);
int launch_gemm_pp(int precision, const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st);

// streaming kernel for the thin linears of the DPT neck (gemm_sk.hip: K <= 256, W tile resident in LDS, A streamed)
bool sk_supported(int precision, const GemmA& a, int M, int N, int K, int Kpad, const GemmEpi& e);
int launch_gemm_sk(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st);

// smallest number of 256 x 256 tiles from which plain linears go to the ping-pong kernel (D2S_GEMM_PP; 0 = never)
int gemm_pp_min_tiles();

// packed-weight geometry
static inline size_t elem_size(int precision) { return precision == D2S_PREC_BF16 ? 2 : (precision == D2S_PREC_FP8_OPERANDS ? 1 : 4); }   // fp32, bf16x3: 4
// host-side packing of one bf16x3 weight row: element k of a row lives in unit k / 8
static inline void bx3_pack_elem(uint8_t* row, int k, float v) {
    const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
    uint8_t* u = row + (size_t)(k >> 3) * 32 + (k & 7) * 2;
    memcpy(u, &hi, 2); memcpy(u + 16, &lo, 2);
}
static inline int gemm_bk(int precision) { return 128 / (int)elem_size(precision); }   // 128-byte K tile
static inline int gemm_kpad(int K, int precision) { int bk = 2 * gemm_bk(precision); return (K + bk - 1) / bk * bk; }   // 256-byte multiple
static inline int gemm_npad(int N) { return (N + 255) / 256 * 256; }

}  // namespace d2s
