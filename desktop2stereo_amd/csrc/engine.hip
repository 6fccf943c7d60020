// Depth-Anything-v2 engine: weights, workspaces, forward orchestration, frame pipeline.
// Replaces DepthModelWrapper + its engine plug-ins (reference depth.py:1539-1781) and the
// per-frame glue of predict_depth / make_sbs (reference depth.py:1897-2025, 2186-2231).
#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>

#include "gemm.h"
#include "vit_ops.h"

using namespace d2s;

namespace {

struct HostT { std::vector<float> data; std::vector<int64_t> shape; };

struct PackedW {           // device: W [Npad][Kpad] T, bias [N] f32 (or null)
    void* w = nullptr;
    float* bias = nullptr;
    int N = 0, K = 0, Kpad = 0;
};

struct Layer {
    float *ln1g, *ln1b, *ln2g, *ln2b, *ls1, *ls2;
    PackedW qkv, proj, fc1, fc2;
    // D2S_PREC_FP8: e4m3 copies of the four linears (index 0 qkv, 1 proj, 2 fc1, 3 fc2) with per-output-channel weight
    // scales; deq[i][n] = s_act(site feeding linear i) * s_w[i][n] is filled in by d2s_engine_calibrate
    PackedW w8[4];
    std::vector<float> sw[4];
    // LayerNorm folded into the linears that consume it (bf16 engines): W' = W diag(gamma) packed like qkv / fc1,
    // bias' = b + W beta in *_ln.bias, csum[n] = sum_k bf16(W'[n][k]) (the values the MFMA actually sums)
    PackedW qkv_ln, fc1_ln;
    float *csum_qkv = nullptr, *csum_fc1 = nullptr;
    float* deq[4] = {nullptr, nullptr, nullptr, nullptr};
    // the same folding on the e4m3 path: [0] QKV, [1] FC1.  A operand = e4m3 of the RAW residual (static scale from the
    // calibration pass, sites 4 / 5), W' quantised per output channel, csum8 over the de-quantised W'
    PackedW w8_ln[2];
    std::vector<float> sw_ln[2];
    float *csum8[2] = {nullptr, nullptr}, *deq_ln[2] = {nullptr, nullptr};
};

constexpr int NSITE = 6;   // calibration sites per layer: LN1 out, attention out, LN2 out, GELU out, residual after proj, residual after FC2

}  // namespace

struct d2s_engine {
    d2s_model_desc d;
    int device = 0;
    int prec = D2S_PREC_BF16;          // activation / kernel type: fp32 or bf16
    int wprec = D2S_PREC_BF16;         // weight packing and GEMM operand precision: = prec, or D2S_PREC_BF16X3 (on fp32 activations)
    std::map<std::string, HostT> host;
    bool finalized = false;
    int h = 0, w = 0, gh = 0, gw = 0, P = 0, N = 0, Npad = 0, maxB = 0;
    uint64_t bytes = 0;
    std::vector<void*> allocs;

    // weights
    PackedW patch;
    float *cls = nullptr, *pos = nullptr;          // pos: [N, D] interpolated (row 0 = cls position)
    std::vector<Layer> L;
    float *lnfg = nullptr, *lnfb = nullptr;
    struct { PackedW proj, resize, conv; PackedW proj_ln; float* csum = nullptr; } re[4];   // proj_ln: final LayerNorm folded in (bf16, batch 1)
    struct { PackedW proj, r1c1, r1c2, r2c1, r2c2; } fu[4];
    PackedW head1, head2;
    float* w3 = nullptr;
    float b3 = 0.f;

    // workspaces
    float* resid = nullptr;                        // [maxB*N, D] fp32 residual stream
    void *lnbuf = nullptr, *qkv = nullptr, *vt = nullptr, *attn = nullptr, *mlp = nullptr, *patchA = nullptr;
    void* tapbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    void* rproj[4] = {nullptr, nullptr, nullptr, nullptr};
    void* rres[4] = {nullptr, nullptr, nullptr, nullptr};
    void* feat[4] = {nullptr, nullptr, nullptr, nullptr};
    int fH[4], fW[4];
    void* scr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* splitk_ws = nullptr;                    // fp32 partials for split-K launches (tiny-M, long-K DPT convs)
    size_t splitk_elems = 0;
    // Neck branches of taps 0..2 (+ the RCU1 of their fusion layer) only depend on their tap, not on later encoder
    // layers: they run on a second stream under the remaining layers (batch 1 leaves most CUs idle per launch).
    hipStream_t side = nullptr;
    hipEvent_t ev_tap[4] = {nullptr, nullptr, nullptr, nullptr}, ev_ln[4] = {nullptr, nullptr, nullptr, nullptr}, ev_side = nullptr;
    int tap_slots = 0;                             // column blocks of the statistics the folded tap projection reads
    bool overlap = true;
    float* splitk_ws_side = nullptr;               // the side stream's own split-K partials
    void* r1[3] = {nullptr, nullptr, nullptr};     // RCU1(feat[i]) = feat[i] + conv2(relu(conv1(relu(feat[i])))), i = 0..2
    void* r1tmp = nullptr;
    // Video-Depth-Anything temporal modules (desc.temporal): layer_3, layer_4, path_4, path_3
    struct TMod {
        int C = 0, sites = 0;
        float *gn_g = nullptr, *gn_b = nullptr;
        float *ln_g[2] = {nullptr, nullptr}, *ln_b[2] = {nullptr, nullptr}, *ffn_g = nullptr, *ffn_b = nullptr;
        PackedW proj_in, proj_out, kvq[2], to_out[2], ff1, ff2;     // kvq: fused to_k | to_v | to_q, [3C][C]
        float* ptab[2] = {nullptr, nullptr};       // [32][3C] = pe @ kvq^T: the positional encoding's share of k | v | q
        void* cache[2] = {nullptr, nullptr};       // ring [31][sites][2C] T per attention block: projected k' | v' rows
        // round 5 (bf16 engine): the three LayerNorms of a module folded into the linears that consume them, like the ViT's (DESIGN.md
        // section 3.1b): W' = W diag(gamma), bias' = b + W beta, colsum over the bf16-rounded W'
        PackedW kvq_ln[2], ff1_ln;                  // (ff1_ln: rows interleaved x | gate in groups of four -- GEGLU happens in its epilogue)
        float *csum_kvq[2] = {nullptr, nullptr}, *csum_ff1 = nullptr;
    } tm[4];
    float* tm_stats = nullptr;                     // (sum, sum of squares) partials per (row, column block) of the folded LayerNorms
    bool tm_fold = false;
    int tm_head = 0, tm_init = 0;                  // oldest ring slot; 0 until the first frame has filled the rings
    float* tm_hs = nullptr;                        // [sites_max, C_max] fp32 residual of the temporal transformer
    void *tm_a = nullptr, *tm_kv = nullptr, *tm_u = nullptr, *tm_g = nullptr, *tm_out = nullptr;
    // pipeline buffers
    float *pre_x = nullptr, *depth_small = nullptr, *depth_post = nullptr;    // model output; post-processed copy (d2s_pipeline)
    void* post_ws = nullptr;
    uint64_t post_ws_bytes = 0;
    float* ema_state = nullptr;
    int ema_init = 0;
    // debug taps (env D2S_TAPS=1): hidden states of frame 0 after embeddings and each layer
    bool taps = false;
    float* tap_hidden = nullptr;                   // [(layers+1), N, D]
    int last_batch = 0;
    // D2S_PREC_FP8 (BASELINE config 3): encoder linears on e4m3 operands once calibrated
    bool fp8 = false, fp8_ready = false, calib = false;
    bool fp8_mlp = false;             // D2S_PREC_FP8_MLP: e4m3 on FC1 / FC2 only (60 % of the encoder FLOPs), QKV / proj stay bf16
    bool lnf = false;                 // LayerNorm folded into the producing / consuming linears (bf16, not fp8)
    bool attn_prescaled = false;      // softmax scale folded into W_q / b_q (bf16 and fp8 engines)
    float* lnstats = nullptr;         // [slots][M][2] partial row sums written by the residual-update GEMMs
    float* amax = nullptr;                         // device [layers][4]: max |.| of LN1 out, attention out, LN2 out, GELU out
    std::vector<float> act_scale;                  // host   [layers][4]: amax / 448
    // per-kernel-class timing with HIP events (d2s_engine_profile): off in the throughput path
    struct ProfRec { int cls; double flops, bytes; hipEvent_t a, b; };
    bool prof_on = false;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
};

namespace {

int dev_alloc(d2s_engine* e, void** p, size_t bytes, bool zero = false) {
    if (bytes == 0) bytes = 16;
    D2S_HIP(hipMalloc(p, bytes));
    e->allocs.push_back(*p);
    e->bytes += bytes;
    if (zero) D2S_HIP(hipMemset(*p, 0, bytes));
    return D2S_OK;
}

enum { PC_GEMM = 0, PC_CONV, PC_ATTN, PC_LN, PC_ELT, PC_PRE, PC_POST, PC_WARP, PC_N };
const char* const PC_NAMES[PC_N] = {"gemm_linear", "gemm_conv3x3", "attention", "layernorm", "elementwise", "preprocess",
                                    "post_process", "stereo_warp"};

hipEvent_t prof_event(d2s_engine* e) {
    if (e->prof_used == e->prof_pool.size()) { hipEvent_t ev; (void)hipEventCreate(&ev); e->prof_pool.push_back(ev); }
    return e->prof_pool[e->prof_used++];
}
void prof_begin(d2s_engine* e, int cls, double flops, double bytes, hipStream_t st) {
    if (!e->prof_on) return;
    d2s_engine::ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes; r.a = prof_event(e); r.b = prof_event(e);
    (void)hipEventRecord(r.a, st);
    e->prof_recs.push_back(r);
}
void prof_end(d2s_engine* e, hipStream_t st) {
    if (!e->prof_on) return;
    (void)hipEventRecord(e->prof_recs.back().b, st);
}
#define PROF(cls, fl, by, call) do { prof_begin(e, cls, fl, by, st); int _rc = (call); prof_end(e, st); if (_rc != D2S_OK) return _rc; } while (0)

const HostT* find(d2s_engine* e, const std::string& name) {
    auto it = e->host.find(name);
    if (it == e->host.end()) { set_error("missing weight tensor: " + name); return nullptr; }
    return &it->second;
}

int upload_f32(d2s_engine* e, const std::string& name, size_t n, float** out) {
    const HostT* t = find(e, name);
    if (!t) return D2S_E_MISSING;
    if (t->data.size() != n) { set_error("weight " + name + ": wrong element count"); return D2S_E_MISSING; }
    int rc = dev_alloc(e, (void**)out, n * sizeof(float));
    if (rc) return rc;
    D2S_HIP(hipMemcpy(*out, t->data.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return D2S_OK;
}

// pack a logical [N][K] float matrix (given by accessor) into device [Npad][Kpad] T
template <typename F>
int pack_matrix(d2s_engine* e, int N, int K, F at, const float* bias_host, PackedW& out) {
    int Kp = gemm_kpad(K, e->wprec), Np = gemm_npad(N);
    size_t es = elem_size(e->wprec);
    std::vector<uint8_t> buf((size_t)Np * Kp * es, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float v = at(n, k);
            if (e->wprec == D2S_PREC_BF16) ((bf16_t*)buf.data())[(size_t)n * Kp + k] = f2bf(v);
            else if (e->wprec == D2S_PREC_BF16X3) bx3_pack_elem(buf.data() + (size_t)n * Kp * 4, k, v);      // [8 hi | 8 lo] units
            else ((float*)buf.data())[(size_t)n * Kp + k] = v;
        }
    int rc = dev_alloc(e, &out.w, buf.size());
    if (rc) return rc;
    D2S_HIP(hipMemcpy(out.w, buf.data(), buf.size(), hipMemcpyHostToDevice));
    out.N = N; out.K = K; out.Kpad = Kp;
    out.bias = nullptr;
    if (bias_host) {
        rc = dev_alloc(e, (void**)&out.bias, (size_t)N * sizeof(float));
        if (rc) return rc;
        D2S_HIP(hipMemcpy(out.bias, bias_host, (size_t)N * sizeof(float), hipMemcpyHostToDevice));
    }
    return D2S_OK;
}

// e4m3 copy of a logical [N][K] matrix: row n is divided by s_w[n] = max|row| / 448 and rounded to e4m3 (RNE)
template <typename F>
int pack_matrix_fp8(d2s_engine* e, int N, int K, F at, const float* bias_dev, PackedW& out, std::vector<float>& sw) {
    int Kp = gemm_kpad(K, D2S_PREC_FP8_OPERANDS), Np = gemm_npad(N);
    std::vector<uint8_t> buf((size_t)Np * Kp, 0);
    sw.assign(N, 1.f);
    for (int n = 0; n < N; ++n) {
        float amax = 0.f;
        for (int k = 0; k < K; ++k) amax = fmaxf(amax, fabsf(at(n, k)));
        float s = amax > 0.f ? amax / FP8_MAX : 1.f;
        sw[n] = s;
        for (int k = 0; k < K; ++k) buf[(size_t)n * Kp + k] = f2e4m3(at(n, k) / s);
    }
    int rc = dev_alloc(e, &out.w, buf.size());
    if (rc) return rc;
    D2S_HIP(hipMemcpy(out.w, buf.data(), buf.size(), hipMemcpyHostToDevice));
    out.N = N; out.K = K; out.Kpad = Kp; out.bias = const_cast<float*>(bias_dev);   // shares the bf16 copy's bias vector
    return D2S_OK;
}

int pack_linear(d2s_engine* e, const std::string& wname, const std::string& bname, int N, int K, PackedW& out) {
    const HostT* w = find(e, wname);
    if (!w) return D2S_E_MISSING;
    if (w->data.size() != (size_t)N * K) { set_error("weight " + wname + ": wrong shape"); return D2S_E_MISSING; }
    const float* b = nullptr;
    if (!bname.empty()) { const HostT* bt = find(e, bname); if (!bt || bt->data.size() != (size_t)N) { set_error("bad bias " + bname); return D2S_E_MISSING; } b = bt->data.data(); }
    const float* p = w->data.data();
    return pack_matrix(e, N, K, [&](int n, int k) { return p[(size_t)n * K + k]; }, b, out);
}

// Conv2d 3x3 weight [Co,Ci,3,3] -> [Co][(ky*3+kx)*Ci + ci]
int pack_conv3(d2s_engine* e, const std::string& wname, const std::string& bname, int Co, int Ci, PackedW& out) {
    const HostT* w = find(e, wname);
    if (!w) return D2S_E_MISSING;
    if (w->data.size() != (size_t)Co * Ci * 9) { set_error("weight " + wname + ": wrong shape"); return D2S_E_MISSING; }
    const float* b = nullptr;
    if (!bname.empty()) { const HostT* bt = find(e, bname); if (!bt || bt->data.size() != (size_t)Co) { set_error("bad bias " + bname); return D2S_E_MISSING; } b = bt->data.data(); }
    const float* p = w->data.data();
    return pack_matrix(e, Co, 9 * Ci, [&](int n, int k) { int tap = k / Ci, ci = k % Ci; return p[((size_t)n * Ci + ci) * 9 + tap]; }, b, out);
}

// ConvTranspose2d k==s weight [Ci,Co,k,k] -> rows n = (ky*k+kx)*Co + co, K = Ci; bias expanded
int pack_convT(d2s_engine* e, const std::string& wname, const std::string& bname, int C, int ks, PackedW& out) {
    const HostT* w = find(e, wname);
    const HostT* bt = find(e, bname);
    if (!w || !bt) return D2S_E_MISSING;
    if (w->data.size() != (size_t)C * C * ks * ks || bt->data.size() != (size_t)C) { set_error("weight " + wname + ": wrong shape"); return D2S_E_MISSING; }
    const float* p = w->data.data();
    int N = ks * ks * C;
    std::vector<float> bias(N);
    for (int n = 0; n < N; ++n) bias[n] = bt->data[n % C];
    return pack_matrix(e, N, C, [&](int n, int k) { int tap = n / C, co = n % C; return p[((size_t)k * C + co) * ks * ks + tap]; }, bias.data(), out);
}

// ---- bicubic (align_corners=False, A=-0.75) resample of the position table, float32 like torch ----
void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    auto c1 = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
    auto c2 = [&](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
    c[0] = c2(t + 1.f); c[1] = c1(t); c[2] = c1(1.f - t); c[3] = c2(2.f - t);
}

void interp_pos(const float* pos, int grid, int D, int gh, int gw, std::vector<float>& out, double offset = 0.0) {
    out.assign((size_t)(1 + gh * gw) * D, 0.f);
    std::memcpy(out.data(), pos, D * sizeof(float));
    if (gh == grid && gw == grid) { std::memcpy(out.data() + D, pos + D, (size_t)grid * grid * D * sizeof(float)); return; }
    float sy = (float)grid / (float)gh, sx = (float)grid / (float)gw;
    if (offset != 0.0) {   // vendored DINOv2 (VDA): scale_factor = (g + 0.1) / grid enters as float(1 / scale_factor), dinov2.py:179-210
        sy = (float)(1.0 / (((double)gh + offset) / (double)grid)); sx = (float)(1.0 / (((double)gw + offset) / (double)grid));
    }
    for (int oy = 0; oy < gh; ++oy) {
        float fy = sy * ((float)oy + 0.5f) - 0.5f;
        float iyf = floorf(fy);
        float cy[4]; cubic_coeffs(fy - iyf, cy);
        int iy = (int)iyf;
        for (int ox = 0; ox < gw; ++ox) {
            float fx = sx * ((float)ox + 0.5f) - 0.5f;
            float ixf = floorf(fx);
            float cx[4]; cubic_coeffs(fx - ixf, cx);
            int ix = (int)ixf;
            float* o = out.data() + (size_t)(1 + oy * gw + ox) * D;
            for (int d = 0; d < D; ++d) {
                float acc = 0.f;
                for (int a = 0; a < 4; ++a) {
                    int yy = std::min(std::max(iy - 1 + a, 0), grid - 1);
                    float row = 0.f;
                    for (int b = 0; b < 4; ++b) {
                        int xx = std::min(std::max(ix - 1 + b, 0), grid - 1);
                        row += pos[(size_t)(1 + yy * grid + xx) * D + d] * cx[b];
                    }
                    acc += row * cy[a];
                }
                o[d] = acc;
            }
        }
    }
}

GemmA plainA(const void* p, long lda) { GemmA a = {}; a.ptr = p; a.mode = A_PLAIN; a.lda = lda; return a; }
GemmA splitA(const void* p, long lda, bool split) { GemmA a = plainA(p, lda); a.bx3 = split ? 1 : 0; return a; }   // bf16x3 engines: pre-split rows
GemmA convA(const void* p, int Hi, int Wi, int C, int Ho, int Wo, int stride, int relu) {
    GemmA a = {}; a.ptr = p; a.mode = A_CONV3; a.Hi = Hi; a.Wi = Wi; a.C = C; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.relu = relu; return a;
}
// tile of the fused head launch: MAP_HEAD needs a tile whose waves own all N columns of their rows (WN == 1)
int head_tile(int bn) {
    static const int t32 = getenv("D2S_HEAD_T32") ? atoi(getenv("D2S_HEAD_T32")) : 0;      // tuning aid
    static const int t64 = getenv("D2S_HEAD_T64") ? atoi(getenv("D2S_HEAD_T64")) : 0;
    return bn == 32 ? (t32 ? t32 : 912832) : (t64 ? t64 : 9256648);
}

GemmEpi rowsE(void* out, int out_type, long ldc, const float* bias) {
    GemmEpi e = {}; e.out = out; e.out_type = out_type; e.ldc = ldc; e.bias = bias; return e;
}

int gemm(d2s_engine* e, const GemmA& a, const PackedW& w, int M, const GemmEpi& ep, hipStream_t st) {
    int Kl = w.K % (e->prec == D2S_PREC_BF16 ? 8 : 4) ? w.Kpad : w.K;   // ragged K (patch embed): A is zero padded to Kpad
    GemmEpi ep2 = ep;
    if (ep.map == MAP_ROWS) {     // launcher decides whether to split K (never the small-tile kernels when LN statistics are due); each stream has its own partials
        ep2.part = (e->side && st == e->side) ? e->splitk_ws_side : e->splitk_ws;
        ep2.part_elems = e->splitk_elems;
    }
    PROF(a.mode == A_CONV3 ? PC_CONV : PC_GEMM, 2.0 * M * w.N * w.K, 0, launch_gemm(e->wprec, 0, a, w.w, M, w.N, Kl, w.Kpad, ep2, st));
    return D2S_OK;
}

// the same linear on e4m3 operands (A rows are e4m3 bytes, lda in elements); the epilogue de-quantises with ep.deq
int gemm8(d2s_engine* e, const GemmA& a, const PackedW& w, int M, const GemmEpi& ep, hipStream_t st) {
    PROF(PC_GEMM, 2.0 * M * w.N * w.K, 0, launch_gemm(D2S_PREC_FP8_OPERANDS, 0, a, w.w, M, w.N, w.K, w.Kpad, ep, st));
    return D2S_OK;
}

#define RC(x) do { int _rc = (x); if (_rc != D2S_OK) return _rc; } while (0)

// 3x3 conv (pad 1) as implicit GEMM over NHWC
int conv3(d2s_engine* e, const void* in, int B, int Hi, int Wi, int C, int stride, int relu_in, const PackedW& w,
          void* out, int act, const void* res1, const void* res2, hipStream_t st) {
    int Ho = (Hi + 2 - 3) / stride + 1, Wo = (Wi + 2 - 3) / stride + 1;
    GemmA a = convA(in, Hi, Wi, C, Ho, Wo, stride, relu_in);
    GemmEpi ep = rowsE(out, OUT_T, w.N, w.bias);
    ep.act = act; ep.res1 = res1; ep.res2 = res2;
    return gemm(e, a, w, B * Ho * Wo, ep, st);
}

// One streaming TemporalModule on an NHWC map x [sites, C] -> out; reads then updates its ring caches.
// (reference motion_module.py:102-134, 164-196, 242-321; cache semantics vda2_s.py:177-218)
int run_temporal(d2s_engine* e, int m, const void* x, void* out, hipStream_t st, const void* add = nullptr) {   // out = module(x) [+ add]
    d2s_engine::TMod& t = e->tm[m];
    const int C = t.C, S = t.sites, prec = e->prec;
    const int Tw = e->tm_init ? 32 : 1;                 // first frame: a window of one (the frame itself at position 0)
    // Round 5, bf16 engine (D2S_VDA_FUSE=0 restores round 4's 18 launches per module): (i) the three LayerNorms live in the linears
    // either side of them -- the residual-update GEMM (proj_in, to_out) also leaves the raw residual as bf16 in tm_a and the row
    // statistics in tm_stats, the consumer (kvq, ff1) runs on gamma-folded weights (section 3.1b's algebra, eps 1e-5); (ii) the ring
    // store of a frame's k' | v' rows happens inside the attention kernel (the lanes that read the oldest slot's segment overwrite it
    // once both of their passes over it are done); (iii) ff2's epilogue leaves the bf16 copy proj_out reads (no cast kernel).
    // 18 -> 11 launches per module.
    const bool fold = e->tm_fold && prec == D2S_PREC_BF16 && e->wprec != D2S_PREC_BF16X3;
    int slots = 0;
    auto producer = [&](GemmEpi& ep) { if (fold) { ep.out2 = e->tm_a; ep.stats_out = e->tm_stats; ep.stats_slots = &slots; } };
    auto consumer = [&](GemmEpi& ep, const float* csum) { ep.ln_stats = e->tm_stats; ep.ln_slots = slots; ep.ln_csum = csum; ep.ln_eps = 1e-5f; ep.ln_dim = C; };
    // (folded: proj_in leaves its bf16 copy in tm_a, so the GroupNorm output it reads goes to tm_out -- free until the attention writes it)
    void* gn_out = fold ? e->tm_out : e->tm_a;
    PROF(PC_ELT, 0, 0, launch_groupnorm(prec, x, t.gn_g, t.gn_b, gn_out, S, C, 32, 1e-6f, st));
    {
        GemmEpi ep = rowsE(e->tm_hs, OUT_F32, C, t.proj_in.bias);
        producer(ep);
        RC(gemm(e, plainA(gn_out, C), t.proj_in, S, ep, st));
    }
    for (int a = 0; a < 2; ++a) {
        const bool folded = fold && slots >= 1 && slots <= 16;
        if (!folded) PROF(PC_LN, 0, 0, launch_layernorm(prec, e->tm_hs, t.ln_g[a], t.ln_b[a], e->tm_a, S, C, 1e-5f, 0, 0, 0, st));
        // project THIS frame only (k' | v' | q'); the window's other 31 positions are already projected in the ring
        {
            GemmEpi ep = rowsE(e->tm_kv, OUT_T, 3 * C, folded ? t.kvq_ln[a].bias : nullptr);
            if (folded) consumer(ep, t.csum_kvq[a]);
            RC(gemm(e, plainA(e->tm_a, C), folded ? t.kvq_ln[a] : t.kvq[a], S, ep, st));
        }
        // the frame's projected rows join the window: the first frame fills all 31 slots (its own launch); later frames replace the
        // oldest slot -- inside the attention kernel when fused
        const int store_slot = (fold && e->tm_init) ? e->tm_head : -1;
        PROF(PC_ATTN, 4.0 * S * Tw * C, 0, launch_temporal_attn(prec, e->tm_kv, t.cache[a], t.ptab[a], e->tm_out, S, C, Tw, 31, e->tm_head, st, store_slot));
        {
            GemmEpi ep = rowsE(e->tm_hs, OUT_F32, C, t.to_out[a].bias);
            ep.res1 = e->tm_hs;
            producer(ep);
            RC(gemm(e, plainA(e->tm_out, C), t.to_out[a], S, ep, st));
        }
        if (store_slot < 0)
            PROF(PC_ELT, 0, 0, launch_cache_store(prec, t.cache[a], e->tm_kv, S, C, e->tm_init ? e->tm_head : 0, e->tm_init ? 1 : 31, st));
    }
    {
        const bool folded = fold && slots >= 1 && slots <= 16;
        if (!folded) PROF(PC_LN, 0, 0, launch_layernorm(prec, e->tm_hs, t.ffn_g, t.ffn_b, e->tm_a, S, C, 1e-5f, 0, 0, 0, st));
        if (folded) {
            // x * gelu(gate) in ff1's own epilogue (interleaved rows): tm_g [S, 4C] directly, no [S, 8C] intermediate, no GEGLU launch
            GemmEpi ep = rowsE(e->tm_g, OUT_T, 4 * C, t.ff1_ln.bias);
            ep.act = ACT_GEGLU;
            consumer(ep, t.csum_ff1);
            RC(gemm(e, plainA(e->tm_a, C), t.ff1_ln, S, ep, st));
        } else {
            GemmEpi ep = rowsE(e->tm_u, OUT_T, 8 * C, t.ff1.bias);
            RC(gemm(e, plainA(e->tm_a, C), t.ff1, S, ep, st));
            PROF(PC_ELT, 0, 0, launch_geglu(prec, e->tm_u, e->tm_g, S, 4 * C, st));
        }
    }
    {
        GemmEpi ep = rowsE(e->tm_hs, OUT_F32, C, t.ff2.bias);
        ep.res1 = e->tm_hs;
        if (fold) ep.out2 = e->tm_a;                    // the bf16 copy proj_out multiplies (no statistics: nothing normalises it)
        RC(gemm(e, plainA(e->tm_g, 4 * C), t.ff2, S, ep, st));
    }
    const void* a_out = e->tm_a;
    if (!fold) {
        if (prec == D2S_PREC_BF16) PROF(PC_ELT, 0, 0, launch_cast_f32(prec, e->tm_hs, e->tm_a, (long)S * C, st));
        else a_out = e->tm_hs;                          // fp32 activations: the residual itself is the operand (round 4 copied it)
    }
    {
        GemmEpi ep = rowsE(out, OUT_T, C, t.proj_out.bias);
        ep.res1 = x; ep.res2 = add;
        RC(gemm(e, plainA(a_out, C), t.proj_out, S, ep, st));
    }
    return D2S_OK;
}

// Tap i of the neck: reassemble (HF DepthAnythingReassembleStage) + 3x3 to fusion width (neck.convs), then -- for the
// three shallower taps -- the first residual unit of their fusion layer, RCU1(m) = m + conv2(relu(conv1(relu(m)))),
// which depends on this map only (HF DepthAnythingFeatureFusionLayer adds it to the deeper stage's output later).
// reassemble projection of tap i.  fold (bf16, batch 1): the shared final LayerNorm happens inside it -- A = the raw bf16
// residual the last FC2 left in lnbuf (patch rows: skip the cls row), statistics from lnstats (same row offset)
int neck_proj(d2s_engine* e, int i, int B, hipStream_t st, bool fold) {
    const d2s_model_desc& d = e->d;
    const int D = d.hidden, Mp = B * e->P, c = d.neck[i];
    if (!fold) return gemm(e, plainA(e->tapbuf[i], D), e->re[i].proj, Mp, rowsE(e->rproj[i], OUT_T, c, e->re[i].proj.bias), st);
    GemmEpi ep = rowsE(e->rproj[i], OUT_T, c, e->re[i].proj_ln.bias);
    ep.ln_slots = e->tap_slots; ep.ln_csum = e->re[i].csum; ep.ln_eps = d.ln_eps; ep.ln_dim = D;
    if (B == 1) {                                      // one frame: skip its cls row by starting one row in
        ep.ln_stats = e->lnstats + 2; ep.ln_M = e->N;
        return gemm(e, plainA((const bf16_t*)e->lnbuf + D, D), e->re[i].proj_ln, Mp, ep, st);
    }
    // several frames: the projection runs over ALL token rows (one cls row per frame: 0.13 % more rows) and the store mapping drops
    // them -- token t of frame b lands on patch row b * P + t - 1 (gemm_epi.h epilogue4: row_off < 0)
    ep.ln_stats = e->lnstats;
    ep.rows_per_img = e->N; ep.img_rows = e->P; ep.row_off = -1;
    return gemm(e, plainA(e->lnbuf, D), e->re[i].proj_ln, B * e->N, ep, st);
}

int neck_rest(d2s_engine* e, int i, int B, hipStream_t st) {
    const d2s_model_desc& d = e->d;
    const int F = d.fusion, gh = e->gh, gw = e->gw, Mp = B * e->P, c = d.neck[i];
    const void* src = e->rproj[i];
    int Hs = gh, Ws = gw;
    if (i < 2) {
        int ks = i == 0 ? 4 : 2;
        GemmEpi ep = rowsE(e->rres[i], OUT_T, c, e->re[i].resize.bias);
        ep.map = MAP_SHUFFLE; ep.gh = gh; ep.gw = gw; ep.ks = ks; ep.cout = c;
        RC(gemm(e, plainA(e->rproj[i], c), e->re[i].resize, Mp, ep, st));
        src = e->rres[i]; Hs = gh * ks; Ws = gw * ks;
    } else if (i == 3) {
        RC(conv3(e, e->rproj[i], B, gh, gw, c, 2, 0, e->re[i].resize, e->rres[i], ACT_NONE, nullptr, nullptr, st));
        src = e->rres[i]; Hs = (gh - 1) / 2 + 1; Ws = (gw - 1) / 2 + 1;
    }
    if (d.temporal && i == 2) { RC(run_temporal(e, 0, src, e->rres[2], st)); src = e->rres[2]; }     // layer_3
    if (d.temporal && i == 3) { RC(run_temporal(e, 1, src, e->scr[0], st)); src = e->scr[0]; }       // layer_4
    RC(conv3(e, src, B, Hs, Ws, c, 1, 0, e->re[i].conv, e->feat[i], ACT_NONE, nullptr, nullptr, st));
    if (i < 3) {
        const int idx = 3 - i;                          // the fusion layer that consumes this map
        RC(conv3(e, e->feat[i], B, Hs, Ws, F, 1, 1, e->fu[idx].r1c1, e->r1tmp, ACT_NONE, nullptr, nullptr, st));
        RC(conv3(e, e->r1tmp, B, Hs, Ws, F, 1, 1, e->fu[idx].r1c2, e->r1[i], ACT_NONE, e->feat[i], nullptr, st));
    }
    return D2S_OK;
}

// x: the pre-processed frames [B,3,h,w], or null when d2s_pipeline has already written the patch rows (and cls rows) itself
int forward(d2s_engine* e, const float* x, float* depth, int B, hipStream_t st) {
    const d2s_model_desc& d = e->d;
    const int D = d.hidden, N = e->N, P = e->P, M = B * N, Mp = B * P, prec = e->prec;
    const int F = d.fusion;
    const bool use_side = e->overlap && e->side != nullptr && !e->prof_on;   // per-kernel timing passes run un-overlapped
    if (e->fp8 && !e->fp8_ready && !e->calib) {
        set_error("D2S_PREC_FP8 engine: activation scales are not set, call d2s_engine_calibrate first");
        return D2S_E_STATE;
    }
    // ---- embeddings (HF Dinov2Embeddings)
    if (x) PROF(PC_ELT, 0, 0, launch_patchify(prec, x, e->patchA, B, e->h, e->w, d.patch, e->patch.Kpad, e->cls, e->pos, e->resid, N, D, st));
    {
        GemmEpi ep = rowsE(e->resid, OUT_F32, D, e->patch.bias);
        ep.rows_per_img = P; ep.img_rows = N; ep.row_off = 1;
        ep.res1 = e->pos; ep.res1_mod = P; ep.res1_off = 1;
        RC(gemm(e, plainA(e->patchA, e->patch.Kpad), e->patch, Mp, ep, st));
    }
    if (e->taps) D2S_HIP(hipMemcpyAsync(e->tap_hidden, e->resid, (size_t)N * D * 4, hipMemcpyDeviceToDevice, st));
    // ---- encoder (HF Dinov2Layer x L)
    int tap_i = 0;
    // D2S_PREC_FP8: the producers of the four linears' A operands write e4m3 (x / s_act, saturated); the linears run on
    // e4m3 operands and de-quantise in their epilogue (deq[n] = s_act * s_w[n]); QKV still emits bf16 for the attention
    const bool f8 = e->fp8 && !e->calib;
    // D2S_PREC_FP8_MLP (round 5): only FC1 / FC2 take e4m3 operands; LN-1 output / attention output stay bf16 and QKV / proj run the bf16
    // kernels.  f8a = "the attention-side linears are e4m3 too" (the all-four scheme)
    const bool f8a = f8 && !e->fp8_mlp;
    // bf16x3 engines: the A operands of the four encoder linears are written PRE-SPLIT (bf16 hi | lo units, common.h) by their
    // producers -- LayerNorm, attention, the GELU epilogue -- so that they travel by LDS-DMA like the bf16 engine's; every other
    // GEMM / conv reads fp32 activations and splits them between its staging registers and LDS
    const bool x3 = e->wprec == D2S_PREC_BF16X3;
    // measured (ViT-B @294x518): folding wins 7 % at 1 frame, 3-5 % at 2-4, 2 % at 8, is even at 16 and loses 1 % at 32
    // (the LN kernels' launch floor is amortised there and the wider epilogues are not); from ~9 frames the encoder linears
    // switch to the 256 x 256 ping-pong kernel (gemm_pp.hip: +12 % frames/s at 16, +13 % at 32), which takes plain
    // LayerNorm-ed operands -> folded up to 8 frames
    // Re-measured at the end of round 3 (the ping-pong kernel now takes launches from 100 tiles, i.e. QKV / FC1 from 4 frames): folding
    // keeps those linears on the small-tile kernels, and from 4 frames that costs more than the LayerNorm launches:
    // 1 670 -> 1 758 frames/s at batch 4, 1 815 -> 2 000 at 5, 2 052 -> 2 275 at 8 with the limit at 3 (same box).
    static const int lnf_maxb = getenv("D2S_LNF_MAXB") ? atoi(getenv("D2S_LNF_MAXB")) : 3;       // tuning aid
    static const bool no_lnf = getenv("D2S_NO_LNFUSE") && atoi(getenv("D2S_NO_LNFUSE")) != 0;
    // Round 4: the ping-pong kernel folds LayerNorm itself (gemm_pp.hip, PP_K_*_LN) once the residual-update linears (N = D) run on
    // it too, i.e. from gemm_pp_min_tiles() tiles of 256 x 256 over [M, D]: the 24 LayerNorm launches of the batched regime (0.66 ms of
    // a 9.8 ms step at batch 32) are gone as well.  In between (QKV / FC1 on the ping-pong kernel, proj / FC2 not yet) nothing folds.
    static EnvInt lnf_pp{"D2S_LNF_PP", 1};
    const bool pp_fold = lnf_pp.get() && e->lnf && !e->fp8 && !x3 && prec == D2S_PREC_BF16 && !e->calib && !e->taps && D % 256 == 0 && D <= 1024 &&
                         gemm_pp_min_tiles() > 0 && (long)cdiv(M, 256) * (D / 256) >= gemm_pp_min_tiles();
    const bool lnf = (((e->lnf && !e->fp8) || (f8 && !no_lnf)) && !e->calib && (prec == D2S_PREC_BF16 || x3) && B <= (x3 ? 8 : lnf_maxb)) || pp_fold;   // (bf16x3: no ping-pong kernel to give way to)
    int ln_slots = 0;
    // batch 1, bf16: the four tap LayerNorms fold into the reassemble projections the same way (the statistics and the raw
    // residual of a tap layer are still in lnbuf / lnstats when its projection runs; the main stream waits for that launch
    // -- ev_ln -- before the next layer's projection GEMM overwrites them)
    // round 4: also in the batched regime where the ping-pong kernel produces the statistics (3-4 partials per row)
    static EnvInt tapf_pp{"D2S_TAPFOLD_PP", 1};
    const bool tap_fold = lnf && !f8 && !x3 && e->lnf && (B == 1 || (pp_fold && tapf_pp.get()));           // (bf16x3: the tap LayerNorms stay kernels)
    bool tap_folded[4] = {false, false, false, false};
    int pending_ln = -1;
    for (int l = 0; l < d.layers; ++l) {
        const Layer& ly = e->L[l];
        const float* sa = f8 ? &e->act_scale[(size_t)l * NSITE] : nullptr; // s_act of LN1 out, attention out, LN2 out, GELU out, residual x 2
        float* am = e->calib ? e->amax + (size_t)l * NSITE : nullptr;
        // lnf: the previous layer's FC2 epilogue left the raw bf16 residual in lnbuf and the row statistics in lnstats;
        // LN1 then happens inside the QKV linear (layer 0 has no such producer and runs the LN kernel)
        const bool ln1_folded = lnf && l > 0 && ln_slots <= 16 && !(f8 && !f8a);     // (MLP-only e4m3: LN-1 stays a kernel, bf16 out)
        if (!ln1_folded) PROF(PC_LN, 0, 0, launch_layernorm(prec, e->resid, ly.ln1g, ly.ln1b, e->lnbuf, M, D, d.ln_eps, 0, 0, 0, st, f8a ? 1.0f / sa[0] : 0.f, x3));
        if (am) RC(launch_amax(prec, e->lnbuf, (long)M * D, am + 0, st));
        {
            GemmEpi ep = rowsE(e->qkv, f8 ? OUT_BF16 : (x3 ? OUT_BX3 : OUT_T), 3 * D, ln1_folded ? (f8a ? ly.w8_ln[0].bias : ly.qkv_ln.bias) : ly.qkv.bias);
            ep.map = MAP_QKV; ep.vt = e->vt; ep.ntok = N; ep.npad = e->Npad; ep.qk_cols = 2 * D; ep.heads = d.heads;
            if (ln1_folded) { ep.ln_stats = e->lnstats; ep.ln_slots = ln_slots; ep.ln_csum = f8a ? ly.csum8[0] : ly.csum_qkv; ep.ln_eps = d.ln_eps; ep.ln_dim = D; }
            if (f8a) { ep.deq = ln1_folded ? ly.deq_ln[0] : ly.deq[0]; RC(gemm8(e, plainA(e->lnbuf, D), ln1_folded ? ly.w8_ln[0] : ly.w8[0], M, ep, st)); }
            else RC(gemm(e, splitA(e->lnbuf, D, x3), ln1_folded ? ly.qkv_ln : ly.qkv, M, ep, st));
        }
        PROF(PC_ATTN, 4.0 * B * d.heads * (double)N * N * 64, 0,
             launch_attention(x3 ? D2S_PREC_BF16X3 : prec, e->qkv, e->vt, e->attn, B, N, e->Npad, d.heads, st, f8a ? 1.0f / sa[1] : 0.f, e->attn_prescaled));
        if (am) RC(launch_amax(prec, e->attn, (long)M * D, am + 1, st));
        {
            if (pending_ln >= 0) { D2S_HIP(hipStreamWaitEvent(st, e->ev_ln[pending_ln], 0)); pending_ln = -1; }
            GemmEpi ep = rowsE(e->resid, OUT_F32, D, ly.proj.bias);
            ep.scale = ly.ls1; ep.res1 = e->resid;
            if (lnf) { ep.out2 = e->lnbuf; ep.out2_bx3 = x3; ep.stats_out = e->lnstats; ep.stats_slots = &ln_slots; ep.out2_qscale = f8 ? 1.0f / sa[4] : 0.f; }
            if (f8a) { ep.deq = ly.deq[1]; RC(gemm8(e, plainA(e->attn, D), ly.w8[1], M, ep, st)); }
            else RC(gemm(e, splitA(e->attn, D, x3), ly.proj, M, ep, st));
        }
        const bool ln2_folded = lnf && ln_slots <= 16;              // (more than 16 column blocks: the LN kernel runs instead)
        if (!ln2_folded) PROF(PC_LN, 0, 0, launch_layernorm(prec, e->resid, ly.ln2g, ly.ln2b, e->lnbuf, M, D, d.ln_eps, 0, 0, 0, st, f8 ? 1.0f / sa[2] : 0.f, x3));
        if (am) RC(launch_amax(D2S_PREC_FP32, e->resid, (long)M * D, am + 4, st));       // (raw residual: the LN-folded FC1's A operand)
        if (am) RC(launch_amax(prec, e->lnbuf, (long)M * D, am + 2, st));
        {
            GemmEpi ep = rowsE(e->mlp, x3 ? OUT_BX3 : OUT_T, d.mlp, ln2_folded ? (f8 ? ly.w8_ln[1].bias : ly.fc1_ln.bias) : ly.fc1.bias);
            ep.act = ACT_GELU;
            if (ln2_folded) { ep.ln_stats = e->lnstats; ep.ln_slots = ln_slots; ep.ln_csum = f8 ? ly.csum8[1] : ly.csum_fc1; ep.ln_eps = d.ln_eps; ep.ln_dim = D; }
            if (f8) { ep.deq = ln2_folded ? ly.deq_ln[1] : ly.deq[2]; ep.out_qscale = 1.0f / sa[3]; RC(gemm8(e, plainA(e->lnbuf, D), ln2_folded ? ly.w8_ln[1] : ly.w8[2], M, ep, st)); }
            else RC(gemm(e, splitA(e->lnbuf, D, x3), ln2_folded ? ly.fc1_ln : ly.fc1, M, ep, st));
        }
        if (am) RC(launch_amax(prec, e->mlp, (long)M * d.mlp, am + 3, st));
        {
            GemmEpi ep = rowsE(e->resid, OUT_F32, D, ly.fc2.bias);
            ep.scale = ly.ls2; ep.res1 = e->resid;
            if (lnf && (l + 1 < d.layers || tap_fold) && !(f8 && !f8a)) { ep.out2 = e->lnbuf; ep.out2_bx3 = x3; ep.stats_out = e->lnstats; ep.stats_slots = &ln_slots; ep.out2_qscale = f8 ? 1.0f / sa[5] : 0.f; }
            if (f8) { ep.deq = ly.deq[3]; RC(gemm8(e, plainA(e->mlp, d.mlp), ly.w8[3], M, ep, st)); }
            else RC(gemm(e, splitA(e->mlp, d.mlp, x3), ly.fc2, M, ep, st));
        }
        if (am) RC(launch_amax(D2S_PREC_FP32, e->resid, (long)M * D, am + 5, st));       // (raw residual: the next layer's LN-folded QKV)
        if (e->taps) D2S_HIP(hipMemcpyAsync(e->tap_hidden + (size_t)(l + 1) * N * D, e->resid, (size_t)N * D * 4, hipMemcpyDeviceToDevice, st));
        if (tap_i < 4 && l + 1 == d.out_indices[tap_i]) {     // HF Dinov2Backbone: shared final LN, drop cls
            const bool fold = tap_fold && ln_slots <= 16;
            tap_folded[tap_i] = fold;
            e->tap_slots = ln_slots;
            if (!fold) PROF(PC_LN, 0, 0, launch_layernorm(prec, e->resid, e->lnfg, e->lnfb, e->tapbuf[tap_i], Mp, D, d.ln_eps, P, N, 1, st));
            // this tap's neck branch runs under the remaining encoder layers.  VDA: the temporal modules of branches 2 and 3
            // share the tm_* workspaces, so branch 3 queues behind branch 2 on the side stream instead of racing it.
            if (use_side && (tap_i < 3 || d.temporal)) {
                D2S_HIP(hipEventRecord(e->ev_tap[tap_i], st));
                D2S_HIP(hipStreamWaitEvent(e->side, e->ev_tap[tap_i], 0));
                RC(neck_proj(e, tap_i, B, e->side, fold));
                if (fold) { D2S_HIP(hipEventRecord(e->ev_ln[tap_i], e->side)); pending_ln = tap_i; }
                RC(neck_rest(e, tap_i, B, e->side));
            } else {
                RC(neck_proj(e, tap_i, B, st, fold));           // (its inputs are overwritten by the next layer; the rest can wait)
            }
            ++tap_i;
        }
    }
    // ---- neck branches that did not run on the side stream, then join it
    for (int i = 0; i < 4; ++i)
        if (!(use_side && (i < 3 || d.temporal))) RC(neck_rest(e, i, B, st));
    if (use_side) {
        D2S_HIP(hipEventRecord(e->ev_side, e->side));
        D2S_HIP(hipStreamWaitEvent(st, e->ev_side, 0));
    }
    // ---- fusion, deep -> shallow (HF DepthAnythingFeatureFusionStage).  hidden(idx) = fused(idx-1) + RCU1(m): the sum is
    // formed where fused(idx-1) is produced (up-sample / temporal-module epilogue), from the r1[] maps of the neck branches.
    void *X = e->scr[0], *Y = e->scr[1], *Z = e->scr[2];
    void* fused = nullptr;
    bool fold1 = false;
    GemmA c1a = {};
    int Hc = 0, Wc = 0;
    for (int idx = 0; idx < 4; ++idx) {
        int mi = 3 - idx;
        void* m = e->feat[mi];
        Hc = e->fH[mi]; Wc = e->fW[mi];
        const void* hcur = idx == 0 ? m : fused;
        RC(conv3(e, hcur, B, Hc, Wc, F, 1, 1, e->fu[idx].r2c1, X, ACT_NONE, nullptr, nullptr, st));
        RC(conv3(e, X, B, Hc, Wc, F, 1, 1, e->fu[idx].r2c2, Z, ACT_NONE, hcur, nullptr, st));
        int Ho, Wo;
        if (idx < 3) { Ho = e->fH[mi - 1]; Wo = e->fW[mi - 1]; } else { Ho = Hc * 2; Wo = Wc * 2; }
        // HF: projection(interpolate(h)).  The 1x1 projection (+bias) commutes with bilinear interpolation
        // (interpolation weights sum to 1), so it runs BEFORE the up-sample on 4x fewer pixels.
        void* pout = e->scr[3 + (idx & 1)];
        const void* next_r1 = idx < 3 ? e->r1[2 - idx] : nullptr;      // RCU1 of the next (shallower) stage's map
        RC(gemm(e, plainA(Z, F), e->fu[idx].proj, B * Hc * Wc, rowsE(X, OUT_T, F, e->fu[idx].proj.bias), st));
        if (d.temporal && idx < 2) {                     // path_4 / path_3 (dpt_temporal.py:98-103)
            PROF(PC_ELT, 0, 0, launch_bilinear_nhwc(prec, X, pout, B, Hc, Wc, Ho, Wo, F, st));
            void* alt = e->scr[3 + ((idx + 1) & 1)];
            RC(run_temporal(e, 2 + idx, pout, alt, st, next_r1));
            pout = alt;
        } else {
            if (idx == 3) {
                // the last stage's up-sample feeds the head's conv1 only: folded into its halo loader where an LDS-resident-input
                // kernel runs it (gemm.h conv3_upsample_ok); conv1 then reads X (the projected map) and writes Y
                c1a = convA(X, Ho, Wo, F, Ho, Wo, 1, 0);
                c1a.ups = 1; c1a.Hs = Hc; c1a.Ws = Wc; c1a.usy = linear_scale(Hc, Ho, true); c1a.usx = linear_scale(Wc, Wo, true);
                GemmEpi ep1 = rowsE(Y, OUT_T, e->head1.N, e->head1.bias);
                fold1 = conv3_upsample_ok(e->wprec, 0, c1a, B * Ho * Wo, e->head1.N, e->head1.K, e->head1.Kpad, ep1);
            }
            if (!fold1) PROF(PC_ELT, 0, 0, launch_bilinear_nhwc(prec, X, pout, B, Hc, Wc, Ho, Wo, F, st, next_r1));
        }
        fused = pout; Hc = Ho; Wc = Wo;
    }
    // ---- head (HF DepthAnythingDepthEstimationHead)
    void* c1out = X;                                     // conv1 output; the (optional) up-sample between conv1 and conv2 goes c1out -> up2
    void* up2 = Y;
    if (fold1) {
        c1out = Y; up2 = X;
        GemmEpi ep1 = rowsE(c1out, OUT_T, e->head1.N, e->head1.bias);
        PROF(PC_CONV, 2.0 * B * Hc * Wc * e->head1.N * e->head1.K, 0, launch_gemm(e->wprec, 0, c1a, e->head1.w, B * Hc * Wc, e->head1.N, e->head1.K, e->head1.Kpad, ep1, st));
    } else {
        RC(conv3(e, fused, B, Hc, Wc, F, 1, 0, e->head1, c1out, ACT_NONE, nullptr, nullptr, st));
    }
    {
        const int Mh = B * e->h * e->w, Nh = d.head_hidden;
        const int bn = Nh <= 32 ? 32 : 64;
        const bool fused_tail = Nh <= 64 && (long)cdiv(Mh, 256) * cdiv(Nh, bn) >= 224;
        GemmA a = convA(up2, e->h, e->w, F / 2, e->h, e->w, 1, 0);
        GemmEpi ep = rowsE(depth, OUT_F32, 1, e->head2.bias);
        ep.map = MAP_HEAD; ep.scale = e->w3; ep.head_b3 = e->b3; ep.head_max_depth = d.max_depth;
        // the interpolate between conv1 and conv2 folded into conv2's halo loader where the persistent head kernel runs (conv3.hip)
        GemmA au = a;
        au.ptr = c1out; au.ups = 1; au.Hs = Hc; au.Ws = Wc; au.usy = linear_scale(Hc, e->h, true); au.usx = linear_scale(Wc, e->w, true);
        const bool ups = fused_tail && conv3_upsample_ok(e->wprec, head_tile(bn), au, Mh, Nh, e->head2.K, e->head2.Kpad, ep);
        if (!ups) PROF(PC_ELT, 0, 0, launch_bilinear_nhwc(prec, c1out, up2, B, Hc, Wc, e->h, e->w, F / 2, st));
        if (fused_tail) {
            // conv2 + ReLU + conv3 (1x1 -> 1 channel) + ReLU | sigmoid in one launch (MAP_HEAD, WN == 1 tiles)
            PROF(PC_CONV, 2.0 * Mh * Nh * e->head2.K, 0, launch_gemm(e->wprec, head_tile(bn), ups ? au : a, e->head2.w, Mh, Nh, e->head2.K, e->head2.Kpad, ep, st));
        } else {
            RC(conv3(e, up2, B, e->h, e->w, F / 2, 1, 0, e->head2, Z, ACT_RELU, nullptr, nullptr, st));
            PROF(PC_ELT, 0, 0, launch_head_final(prec, Z, e->w3, e->b3, d.max_depth, depth, (long)B * e->h * e->w, d.head_hidden, st));
        }
    }
    if (d.temporal) {                                    // one window step per frame (vda2_s.py:177-187, 214-221)
        if (e->tm_init) e->tm_head = (e->tm_head + 1) % 31;
        e->tm_init = 1;
    }
    e->last_batch = B;
    return D2S_OK;
}

}  // namespace

// ================================================================================================
extern "C" int d2s_engine_create(const d2s_model_desc* desc, int device_id, d2s_engine** out) {
    D2S_REQUIRE(desc && out, "null pointer");
    D2S_REQUIRE(desc->hidden > 0 && desc->heads > 0 && desc->hidden == desc->heads * 64, "head_dim must be 64");
    D2S_REQUIRE(desc->layers > 0 && desc->patch > 0 && desc->pos_grid > 0 && desc->fusion % 8 == 0, "bad model desc");
    D2S_REQUIRE(desc->precision == D2S_PREC_FP32 || desc->precision == D2S_PREC_BF16 || desc->precision == D2S_PREC_FP8 ||
                desc->precision == D2S_PREC_BF16X3 || desc->precision == D2S_PREC_FP8_MLP, "bad precision");
    for (int i = 0; i < 4; ++i) D2S_REQUIRE(desc->neck[i] % 8 == 0 && desc->out_indices[i] >= 1 && desc->out_indices[i] <= desc->layers, "bad neck / out_indices");
    D2S_REQUIRE(desc->head_hidden % 4 == 0 && desc->mlp % 8 == 0, "bad head_hidden / mlp");
    D2S_REQUIRE(desc->max_depth >= 0.f && !(desc->temporal && desc->max_depth > 0.f),
                "max_depth must be >= 0, and 0 for a Video-Depth-Anything engine (its head ends in ReLU, dpt_temporal.py:136)");
    D2S_ON_DEVICE(device_id);
    d2s_engine* e = new d2s_engine();
    e->d = *desc; e->device = device_id;
    e->fp8 = desc->precision == D2S_PREC_FP8 || desc->precision == D2S_PREC_FP8_MLP;
    e->fp8_mlp = desc->precision == D2S_PREC_FP8_MLP;
    {   // LayerNorm fusion: bf16 and bf16x3 engines (not the plain fp32 engine; the e4m3 path quantises the LN output itself)
        const char* no = getenv("D2S_NO_LNFUSE");
        e->lnf = (desc->precision == D2S_PREC_BF16 || desc->precision == D2S_PREC_BF16X3) && !(no && atoi(no) != 0);
    }                  // bf16 engine whose encoder linears switch to e4m3 operands
    e->prec = e->fp8 ? D2S_PREC_BF16 : (desc->precision == D2S_PREC_BF16X3 ? D2S_PREC_FP32 : desc->precision);
    e->wprec = desc->precision == D2S_PREC_BF16X3 ? D2S_PREC_BF16X3 : e->prec;      // split-precision GEMM operands on the fp32 engine
    e->attn_prescaled = e->prec == D2S_PREC_BF16;
    const char* t = getenv("D2S_TAPS");
    e->taps = t && atoi(t) != 0;
    *out = e;
    return D2S_OK;
}

extern "C" int d2s_engine_set_weight(d2s_engine* e, const char* name, const float* host, const int64_t* shape, int ndim) {
    D2S_REQUIRE(e && name && host && shape && ndim >= 1 && ndim <= 4, "bad argument");
    if (e->finalized) { set_error("d2s_engine_set_weight after finalize"); return D2S_E_STATE; }
    HostT t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { D2S_REQUIRE(shape[i] > 0, "bad shape"); n *= (size_t)shape[i]; t.shape.push_back(shape[i]); }
    t.data.assign(host, host + n);
    e->host[name] = std::move(t);
    return D2S_OK;
}

extern "C" int d2s_engine_finalize(d2s_engine* e, int h, int w, int max_batch) {
    D2S_REQUIRE(e, "null engine");
    if (e->finalized) { set_error("engine already finalized"); return D2S_E_STATE; }
    const d2s_model_desc& d = e->d;
    D2S_REQUIRE(h > 0 && w > 0 && h % d.patch == 0 && w % d.patch == 0 && max_batch >= 1, "h, w must be patch multiples");
    D2S_ON_DEVICE(e->device);
    const int D = d.hidden, F = d.fusion;
    e->h = h; e->w = w; e->gh = h / d.patch; e->gw = w / d.patch; e->P = e->gh * e->gw; e->N = e->P + 1;
    e->Npad = (e->N + 63) / 64 * 64; e->maxB = max_batch;
    const int N = e->N, P = e->P, B = max_batch;
    const size_t es = elem_size(e->prec);
    std::string pe = "backbone.embeddings.";
    // ---- weights
    RC(pack_linear(e, pe + "patch_embeddings.projection.weight", pe + "patch_embeddings.projection.bias", D, 3 * d.patch * d.patch, e->patch));
    RC(upload_f32(e, pe + "cls_token", D, &e->cls));
    {
        const HostT* pt = find(e, pe + "position_embeddings");
        if (!pt) return D2S_E_MISSING;
        if (pt->data.size() != (size_t)(d.pos_grid * d.pos_grid + 1) * D) { set_error("position_embeddings: wrong shape"); return D2S_E_MISSING; }
        std::vector<float> pos;
        interp_pos(pt->data.data(), d.pos_grid, D, e->gh, e->gw, pos, d.temporal ? 0.1 : 0.0);
        RC(dev_alloc(e, (void**)&e->pos, pos.size() * 4));
        D2S_HIP(hipMemcpy(e->pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
    }
    e->L.resize(d.layers);
    for (int l = 0; l < d.layers; ++l) {
        std::string p = "backbone.encoder.layer." + std::to_string(l) + ".";
        Layer& ly = e->L[l];
        RC(upload_f32(e, p + "norm1.weight", D, &ly.ln1g)); RC(upload_f32(e, p + "norm1.bias", D, &ly.ln1b));
        RC(upload_f32(e, p + "norm2.weight", D, &ly.ln2g)); RC(upload_f32(e, p + "norm2.bias", D, &ly.ln2b));
        RC(upload_f32(e, p + "layer_scale1.lambda1", D, &ly.ls1)); RC(upload_f32(e, p + "layer_scale2.lambda1", D, &ly.ls2));
        // fused QKV: rows q | k | v
        const HostT *wq = find(e, p + "attention.attention.query.weight"), *wk = find(e, p + "attention.attention.key.weight"),
                    *wv = find(e, p + "attention.attention.value.weight"), *bq = find(e, p + "attention.attention.query.bias"),
                    *bk = find(e, p + "attention.attention.key.bias"), *bv = find(e, p + "attention.attention.value.bias");
        if (!wq || !wk || !wv || !bq || !bk || !bv) return D2S_E_MISSING;
        for (const HostT* t : {wq, wk, wv}) if (t->data.size() != (size_t)D * D) { set_error("qkv weight: wrong shape"); return D2S_E_MISSING; }
        std::vector<float> bias(3 * D);
        for (int i = 0; i < D; ++i) { bias[i] = bq->data[i]; bias[D + i] = bk->data[i]; bias[2 * D + i] = bv->data[i]; }
        // bf16 / fp8 engines: the softmax scale 64^-0.5 log2(e) goes into the q rows of the fused QKV weight and bias (fp32
        // product, then the ONE rounding every weight gets): the attention kernels see log2-domain scores, the batched one
        // straight from the matrix pipe (attention.hip).  The fp32 engine (parity class) keeps the reference's order of operations.
        std::vector<float> wq_scaled;
        if (e->attn_prescaled) {
            wq_scaled.resize(wq->data.size());
            for (size_t i = 0; i < wq_scaled.size(); ++i) wq_scaled[i] = wq->data[i] * ATTN_SCALE_LOG2E;
            for (int i = 0; i < D; ++i) bias[i] *= ATTN_SCALE_LOG2E;
        }
        const float* ws[3] = {e->attn_prescaled ? wq_scaled.data() : wq->data.data(), wk->data.data(), wv->data.data()};
        RC(pack_matrix(e, 3 * D, D, [&](int n, int k) { return ws[n / D][(size_t)(n % D) * D + k]; }, bias.data(), ly.qkv));
        RC(pack_linear(e, p + "attention.output.dense.weight", p + "attention.output.dense.bias", D, D, ly.proj));
        RC(pack_linear(e, p + "mlp.fc1.weight", p + "mlp.fc1.bias", d.mlp, D, ly.fc1));
        RC(pack_linear(e, p + "mlp.fc2.weight", p + "mlp.fc2.bias", D, d.mlp, ly.fc2));
        if (e->lnf) {
            // LN(x) W^T + b  =  rstd * (x W'^T - mean * colsum(W')) + (b + W beta),  W' = W diag(gamma)
            auto fold = [&](const HostT* g, const HostT* bt, int Nn, auto at, const float* bias, PackedW& out, float** csum) -> int {
                std::vector<float> b2(Nn), cs(Nn);
                for (int n = 0; n < Nn; ++n) {
                    double sb = bias[n], sc = 0.0;
                    for (int k = 0; k < D; ++k) {
                        sb += (double)bt->data[k] * at(n, k);
                        const float v = g->data[k] * at(n, k);              // colsum over what the MFMAs sum: bf16(W'), or its hi + lo halves
                        const float hi = bf2f(f2bf(v));
                        sc += e->wprec == D2S_PREC_BF16X3 ? (double)hi + (double)bf2f(f2bf(v - hi)) : (double)hi;
                    }
                    b2[n] = (float)sb; cs[n] = (float)sc;
                }
                int rc = pack_matrix(e, Nn, D, [&](int n, int k) { return g->data[k] * at(n, k); }, b2.data(), out);
                if (rc) return rc;
                rc = dev_alloc(e, (void**)csum, (size_t)Nn * sizeof(float));
                if (rc) return rc;
                D2S_HIP(hipMemcpy(*csum, cs.data(), (size_t)Nn * sizeof(float), hipMemcpyHostToDevice));
                return D2S_OK;
            };
            const HostT *g1 = find(e, p + "norm1.weight"), *b1 = find(e, p + "norm1.bias"), *g2 = find(e, p + "norm2.weight"), *b2 = find(e, p + "norm2.bias");
            const HostT *w1t = find(e, p + "mlp.fc1.weight"), *b1t = find(e, p + "mlp.fc1.bias");
            if (!g1 || !b1 || !g2 || !b2 || !w1t || !b1t) return D2S_E_MISSING;
            RC(fold(g1, b1, 3 * D, [&](int n, int k) { return ws[n / D][(size_t)(n % D) * D + k]; }, bias.data(), ly.qkv_ln, &ly.csum_qkv));
            const float* w1p = w1t->data.data();
            RC(fold(g2, b2, d.mlp, [&](int n, int k) { return w1p[(size_t)n * D + k]; }, b1t->data.data(), ly.fc1_ln, &ly.csum_fc1));
        }
        if (e->fp8) {
            const float* wo = find(e, p + "attention.output.dense.weight")->data.data();
            const float* w1 = find(e, p + "mlp.fc1.weight")->data.data();
            const float* w2 = find(e, p + "mlp.fc2.weight")->data.data();
            RC(pack_matrix_fp8(e, 3 * D, D, [&](int n, int k) { return ws[n / D][(size_t)(n % D) * D + k]; }, ly.qkv.bias, ly.w8[0], ly.sw[0]));
            RC(pack_matrix_fp8(e, D, D, [&](int n, int k) { return wo[(size_t)n * D + k]; }, ly.proj.bias, ly.w8[1], ly.sw[1]));
            RC(pack_matrix_fp8(e, d.mlp, D, [&](int n, int k) { return w1[(size_t)n * D + k]; }, ly.fc1.bias, ly.w8[2], ly.sw[2]));
            RC(pack_matrix_fp8(e, D, d.mlp, [&](int n, int k) { return w2[(size_t)n * d.mlp + k]; }, ly.fc2.bias, ly.w8[3], ly.sw[3]));
            for (int i = 0; i < 4; ++i) RC(dev_alloc(e, (void**)&ly.deq[i], (size_t)ly.w8[i].N * sizeof(float), true));
            // LN-folded e4m3 copies of QKV / FC1 (see Layer): W' = W diag(gamma), bias' = b + W beta, csum over the de-quantised W'
            auto fold8 = [&](const HostT* g, const HostT* bt, int Nn, auto at, const float* bias, int slot) -> int {
                RC(pack_matrix_fp8(e, Nn, D, [&](int n, int k) { return g->data[k] * at(n, k); }, nullptr, ly.w8_ln[slot], ly.sw_ln[slot]));
                std::vector<float> b2(Nn), cs(Nn);
                for (int n = 0; n < Nn; ++n) {
                    double sb = bias[n], sc = 0.0;
                    const float sw = ly.sw_ln[slot][n];
                    for (int k = 0; k < D; ++k) { sb += (double)bt->data[k] * at(n, k); sc += e4m32f(f2e4m3(g->data[k] * at(n, k) / sw)); }
                    b2[n] = (float)sb; cs[n] = (float)(sc * sw);
                }
                RC(dev_alloc(e, (void**)&ly.w8_ln[slot].bias, (size_t)Nn * sizeof(float)));
                D2S_HIP(hipMemcpy(ly.w8_ln[slot].bias, b2.data(), (size_t)Nn * sizeof(float), hipMemcpyHostToDevice));
                RC(dev_alloc(e, (void**)&ly.csum8[slot], (size_t)Nn * sizeof(float)));
                D2S_HIP(hipMemcpy(ly.csum8[slot], cs.data(), (size_t)Nn * sizeof(float), hipMemcpyHostToDevice));
                RC(dev_alloc(e, (void**)&ly.deq_ln[slot], (size_t)Nn * sizeof(float), true));
                return D2S_OK;
            };
            const HostT *g1 = find(e, p + "norm1.weight"), *bn1 = find(e, p + "norm1.bias"), *g2 = find(e, p + "norm2.weight"), *bn2 = find(e, p + "norm2.bias");
            const HostT* b1t = find(e, p + "mlp.fc1.bias");
            if (!g1 || !bn1 || !g2 || !bn2 || !b1t) return D2S_E_MISSING;
            RC(fold8(g1, bn1, 3 * D, [&](int n, int k) { return ws[n / D][(size_t)(n % D) * D + k]; }, bias.data(), 0));
            RC(fold8(g2, bn2, d.mlp, [&](int n, int k) { return w1[(size_t)n * D + k]; }, b1t->data.data(), 1));
        }
    }
    if (e->fp8) {
        RC(dev_alloc(e, (void**)&e->amax, (size_t)d.layers * NSITE * sizeof(float), true));
        e->act_scale.assign((size_t)d.layers * NSITE, 0.f);
    }
    RC(upload_f32(e, "backbone.layernorm.weight", D, &e->lnfg));
    RC(upload_f32(e, "backbone.layernorm.bias", D, &e->lnfb));
    for (int i = 0; i < 4; ++i) {
        std::string p = "neck.reassemble_stage.layers." + std::to_string(i) + ".";
        int c = d.neck[i];
        RC(pack_linear(e, p + "projection.weight", p + "projection.bias", c, D, e->re[i].proj));
        if (i == 0) RC(pack_convT(e, p + "resize.weight", p + "resize.bias", c, 4, e->re[i].resize));
        if (i == 1) RC(pack_convT(e, p + "resize.weight", p + "resize.bias", c, 2, e->re[i].resize));
        if (i == 3) RC(pack_conv3(e, p + "resize.weight", p + "resize.bias", c, c, e->re[i].resize));
        RC(pack_conv3(e, "neck.convs." + std::to_string(i) + ".weight", "", F, c, e->re[i].conv));
    }
    if (e->lnf) {                                       // final LayerNorm folded into the four reassemble projections (batch 1)
        const HostT *gf = find(e, "backbone.layernorm.weight"), *bf = find(e, "backbone.layernorm.bias");
        if (!gf || !bf) return D2S_E_MISSING;
        for (int i = 0; i < 4; ++i) {
            std::string p = "neck.reassemble_stage.layers." + std::to_string(i) + ".projection.";
            const HostT *wt = find(e, p + "weight"), *bt = find(e, p + "bias");
            if (!wt || !bt) return D2S_E_MISSING;
            const int c = d.neck[i];
            const float* wp = wt->data.data();
            std::vector<float> b2(c), cs(c);
            for (int n = 0; n < c; ++n) {
                double sb = bt->data[n], sc = 0.0;
                for (int k = 0; k < D; ++k) { sb += (double)bf->data[k] * wp[(size_t)n * D + k]; sc += bf2f(f2bf(gf->data[k] * wp[(size_t)n * D + k])); }
                b2[n] = (float)sb; cs[n] = (float)sc;
            }
            RC(pack_matrix(e, c, D, [&](int n, int k) { return gf->data[k] * wp[(size_t)n * D + k]; }, b2.data(), e->re[i].proj_ln));
            RC(dev_alloc(e, (void**)&e->re[i].csum, (size_t)c * sizeof(float)));
            D2S_HIP(hipMemcpy(e->re[i].csum, cs.data(), (size_t)c * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    for (int i = 0; i < 4; ++i) {
        std::string p = "neck.fusion_stage.layers." + std::to_string(i) + ".";
        RC(pack_linear(e, p + "projection.weight", p + "projection.bias", F, F, e->fu[i].proj));
        RC(pack_conv3(e, p + "residual_layer1.convolution1.weight", p + "residual_layer1.convolution1.bias", F, F, e->fu[i].r1c1));
        RC(pack_conv3(e, p + "residual_layer1.convolution2.weight", p + "residual_layer1.convolution2.bias", F, F, e->fu[i].r1c2));
        RC(pack_conv3(e, p + "residual_layer2.convolution1.weight", p + "residual_layer2.convolution1.bias", F, F, e->fu[i].r2c1));
        RC(pack_conv3(e, p + "residual_layer2.convolution2.weight", p + "residual_layer2.convolution2.bias", F, F, e->fu[i].r2c2));
    }
    RC(pack_conv3(e, "head.conv1.weight", "head.conv1.bias", F / 2, F, e->head1));
    RC(pack_conv3(e, "head.conv2.weight", "head.conv2.bias", d.head_hidden, F / 2, e->head2));
    RC(upload_f32(e, "head.conv3.weight", d.head_hidden, &e->w3));
    { const HostT* b3 = find(e, "head.conv3.bias"); if (!b3) return D2S_E_MISSING; e->b3 = b3->data[0]; }
    // ---- workspaces
    const size_t M = (size_t)B * N, Mp = (size_t)B * P;
    RC(dev_alloc(e, (void**)&e->resid, M * D * 4));
    RC(dev_alloc(e, &e->lnbuf, M * D * es));
    if (e->lnf || e->fp8) RC(dev_alloc(e, (void**)&e->lnstats, (size_t)(D / 16 + 1) * M * 2 * sizeof(float)));
    RC(dev_alloc(e, &e->qkv, M * 3 * D * es));
    RC(dev_alloc(e, &e->vt, (size_t)B * D * e->Npad * es, true));     // zero beyond N, never written there
    RC(dev_alloc(e, &e->attn, M * D * es));
    RC(dev_alloc(e, &e->mlp, M * d.mlp * es));
    RC(dev_alloc(e, &e->patchA, Mp * e->patch.Kpad * es, true));        // (zero: the padding columns are never written again)
    const int gh = e->gh, gw = e->gw;
    e->fH[0] = gh * 4; e->fW[0] = gw * 4; e->fH[1] = gh * 2; e->fW[1] = gw * 2; e->fH[2] = gh; e->fW[2] = gw;
    e->fH[3] = (gh - 1) / 2 + 1; e->fW[3] = (gw - 1) / 2 + 1;
    for (int i = 0; i < 4; ++i) {
        RC(dev_alloc(e, &e->tapbuf[i], Mp * D * es));
        RC(dev_alloc(e, &e->rproj[i], Mp * d.neck[i] * es));
        RC(dev_alloc(e, &e->rres[i], (size_t)B * e->fH[i] * e->fW[i] * d.neck[i] * es));
        RC(dev_alloc(e, &e->feat[i], (size_t)B * e->fH[i] * e->fW[i] * F * es));
    }
    size_t scr_elems = std::max({(size_t)64 * gh * gw * F, (size_t)h * w * (F / 2), (size_t)h * w * d.head_hidden}) * B;
    for (int i = 0; i < 5; ++i) RC(dev_alloc(e, &e->scr[i], scr_elems * es));
    e->splitk_elems = (size_t)B * 16 * e->fH[2] * e->fW[2] * std::max(F, d.neck[3]);
    // (+ GEMM_PART_CTR_WORDS zeroed words BEHIND the partials: the ping-pong kernel's tail counters, which no split-K launch can reach)
    RC(dev_alloc(e, (void**)&e->splitk_ws, (e->splitk_elems + GEMM_PART_CTR_WORDS) * 4, true));
    // side stream of the neck branches (D2S_NO_OVERLAP=1 keeps everything on the caller's stream)
    for (int i = 0; i < 3; ++i) RC(dev_alloc(e, &e->r1[i], (size_t)B * e->fH[i] * e->fW[i] * F * es));
    RC(dev_alloc(e, &e->r1tmp, (size_t)B * e->fH[0] * e->fW[0] * F * es));
    RC(dev_alloc(e, (void**)&e->splitk_ws_side, (e->splitk_elems + GEMM_PART_CTR_WORDS) * 4, true));
    {
        const char* no = getenv("D2S_NO_OVERLAP");
        e->overlap = !(no && atoi(no) != 0);
        if (e->overlap) {
            D2S_HIP(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
            for (int i = 0; i < 4; ++i) D2S_HIP(hipEventCreateWithFlags(&e->ev_tap[i], hipEventDisableTiming));
            for (int i = 0; i < 4; ++i) D2S_HIP(hipEventCreateWithFlags(&e->ev_ln[i], hipEventDisableTiming));
            D2S_HIP(hipEventCreateWithFlags(&e->ev_side, hipEventDisableTiming));
        }
    }
    if (d.temporal) {
        // ---- Video-Depth-Anything temporal modules (reference dpt_temporal.py:50-60): layer_3, layer_4, path_4, path_3
        const int tC[4] = {d.neck[2], d.neck[3], F, F};
        const int tS[4] = {e->fH[2] * e->fW[2], e->fH[3] * e->fW[3], e->fH[2] * e->fW[2], e->fH[1] * e->fW[1]};
        size_t sc_max = 0, max_sites = 0;
        {
            const char* nf = getenv("D2S_VDA_FUSE");
            e->tm_fold = e->prec == D2S_PREC_BF16 && e->wprec != D2S_PREC_BF16X3 && !(nf && atoi(nf) == 0);
        }
        // LN(x) W^T + b  =  rstd * (x W'^T - mean * colsum(W')) + (b + W beta),  W' = W diag(gamma): colsum over the bf16-rounded W'
        auto fold_ln = [&](const float* g, const float* bt, int Nn, int K, auto at, const float* bias, PackedW& out, float** csum) -> int {
            std::vector<float> b2(Nn), cs(Nn);
            for (int n = 0; n < Nn; ++n) {
                double sb = bias ? bias[n] : 0.0, sc = 0.0;
                for (int k = 0; k < K; ++k) { sb += (double)bt[k] * at(n, k); sc += (double)bf2f(f2bf(g[k] * at(n, k))); }
                b2[n] = (float)sb; cs[n] = (float)sc;
            }
            int rc = pack_matrix(e, Nn, K, [&](int n, int k) { return g[k] * at(n, k); }, b2.data(), out);
            if (rc) return rc;
            rc = dev_alloc(e, (void**)csum, (size_t)Nn * sizeof(float));
            if (rc) return rc;
            D2S_HIP(hipMemcpy(*csum, cs.data(), (size_t)Nn * sizeof(float), hipMemcpyHostToDevice));
            return D2S_OK;
        };
        for (int m = 0; m < 4; ++m) {
            d2s_engine::TMod& t = e->tm[m];
            t.C = tC[m]; t.sites = tS[m];
            max_sites = std::max(max_sites, (size_t)t.sites);
            const int C = t.C;
            D2S_REQUIRE(C % 32 == 0 && C <= 1024, "temporal module channels must be a multiple of 32 (GroupNorm) and <= 1024");
            sc_max = std::max(sc_max, (size_t)t.sites * C);
            std::string p = "head.motion_modules." + std::to_string(m) + ".temporal_transformer.";
            std::string b = p + "transformer_blocks.0.";
            RC(upload_f32(e, p + "norm.weight", C, &t.gn_g)); RC(upload_f32(e, p + "norm.bias", C, &t.gn_b));
            RC(pack_linear(e, p + "proj_in.weight", p + "proj_in.bias", C, C, t.proj_in));
            RC(pack_linear(e, p + "proj_out.weight", p + "proj_out.bias", C, C, t.proj_out));
            // sinusoidal APE over the 32-frame window (motion_module.py:214-222), float32 like torch
            std::vector<float> pe((size_t)32 * C);
            for (int i = 0; i < C / 2; ++i) {
                float div = expf((float)(2 * i) * (float)(-std::log(10000.0) / C));
                for (int pos = 0; pos < 32; ++pos) { pe[(size_t)pos * C + 2 * i] = sinf((float)pos * div); pe[(size_t)pos * C + 2 * i + 1] = cosf((float)pos * div); }
            }
            for (int a = 0; a < 2; ++a) {
                std::string q = b + "attention_blocks." + std::to_string(a) + ".";
                RC(upload_f32(e, b + "norms." + std::to_string(a) + ".weight", C, &t.ln_g[a]));
                RC(upload_f32(e, b + "norms." + std::to_string(a) + ".bias", C, &t.ln_b[a]));
                const HostT *wq = find(e, q + "to_q.weight"), *wk = find(e, q + "to_k.weight"), *wv = find(e, q + "to_v.weight");
                if (!wq || !wk || !wv) return D2S_E_MISSING;
                for (const HostT* w3 : {wq, wk, wv}) if (w3->data.size() != (size_t)C * C) { set_error("to_q/to_k/to_v: wrong shape"); return D2S_E_MISSING; }
                const float* kvq[3] = {wk->data.data(), wv->data.data(), wq->data.data()};      // fused rows: k | v | q (no biases)
                RC(pack_matrix(e, 3 * C, C, [&](int n, int k) { return kvq[n / C][(size_t)(n % C) * C + k]; }, nullptr, t.kvq[a]));
                // W (x + pe_j) = W x + W pe_j: the positional share of every window position, float32
                std::vector<float> pt((size_t)32 * 3 * C);
                for (int j = 0; j < 32; ++j)
                    for (int n = 0; n < 3 * C; ++n) {
                        const float* wr = kvq[n / C] + (size_t)(n % C) * C;
                        double acc = 0.0;
                        for (int k = 0; k < C; ++k) acc += (double)pe[(size_t)j * C + k] * (double)wr[k];
                        pt[(size_t)j * 3 * C + n] = (float)acc;
                    }
                RC(dev_alloc(e, (void**)&t.ptab[a], pt.size() * 4));
                D2S_HIP(hipMemcpy(t.ptab[a], pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
                RC(pack_linear(e, q + "to_out.0.weight", q + "to_out.0.bias", C, C, t.to_out[a]));
                RC(dev_alloc(e, &t.cache[a], (size_t)31 * t.sites * 2 * C * es, true));
                if (e->tm_fold) {
                    const HostT *g = find(e, b + "norms." + std::to_string(a) + ".weight"), *bt = find(e, b + "norms." + std::to_string(a) + ".bias");
                    if (!g || !bt) return D2S_E_MISSING;
                    RC(fold_ln(g->data.data(), bt->data.data(), 3 * C, C, [&](int n, int k) { return kvq[n / C][(size_t)(n % C) * C + k]; }, nullptr,
                               t.kvq_ln[a], &t.csum_kvq[a]));
                }
            }
            RC(upload_f32(e, b + "ff_norm.weight", C, &t.ffn_g)); RC(upload_f32(e, b + "ff_norm.bias", C, &t.ffn_b));
            RC(pack_linear(e, b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", 8 * C, C, t.ff1));
            RC(pack_linear(e, b + "ff.net.2.weight", b + "ff.net.2.bias", C, 4 * C, t.ff2));
            if (e->tm_fold) {
                const HostT *g = find(e, b + "ff_norm.weight"), *bt = find(e, b + "ff_norm.bias"), *w1 = find(e, b + "ff.net.0.proj.weight"),
                            *b1 = find(e, b + "ff.net.0.proj.bias");
                if (!g || !bt || !w1 || !b1) return D2S_E_MISSING;
                const float* w1p = w1->data.data();
                // GEGLU in the epilogue (ACT_GEGLU): packed row n' = 8 g + w holds x row 4 g + w (w < 4) or gate row 4C + 4 g + (w - 4)
                auto orig = [C](int n) { const int g = n >> 3, w = n & 7; return w < 4 ? 4 * g + w : 4 * C + 4 * g + (w - 4); };
                std::vector<float> b1p((size_t)8 * C);
                for (int n = 0; n < 8 * C; ++n) b1p[n] = b1->data[orig(n)];
                RC(fold_ln(g->data.data(), bt->data.data(), 8 * C, C, [&](int n, int k) { return w1p[(size_t)orig(n) * C + k]; }, b1p.data(), t.ff1_ln, &t.csum_ff1));
            }
        }
        if (e->tm_fold) RC(dev_alloc(e, (void**)&e->tm_stats, (size_t)17 * max_sites * 2 * sizeof(float)));
        RC(dev_alloc(e, (void**)&e->tm_hs, sc_max * 4));
        RC(dev_alloc(e, &e->tm_a, sc_max * es)); RC(dev_alloc(e, &e->tm_out, sc_max * es));
        RC(dev_alloc(e, &e->tm_kv, sc_max * 3 * es));                  // k' | v' | q' of the current frame
        RC(dev_alloc(e, &e->tm_u, sc_max * 8 * es)); RC(dev_alloc(e, &e->tm_g, sc_max * 4 * es));
    }
    e->host.clear();
    RC(dev_alloc(e, (void**)&e->pre_x, (size_t)B * 3 * h * w * 4));
    RC(dev_alloc(e, (void**)&e->depth_small, (size_t)B * h * w * 4));
    RC(dev_alloc(e, (void**)&e->depth_post, (size_t)B * h * w * 4));
    e->post_ws_bytes = d2s_post_process_workspace(B, h, w);
    RC(dev_alloc(e, &e->post_ws, e->post_ws_bytes));
    RC(dev_alloc(e, (void**)&e->ema_state, (size_t)h * w * 4));
    if (e->taps) RC(dev_alloc(e, (void**)&e->tap_hidden, (size_t)(d.layers + 1) * N * D * 4));
    D2S_HIP(hipDeviceSynchronize());
    e->finalized = true;
    return D2S_OK;
}

extern "C" int d2s_engine_destroy(d2s_engine* e) {
    if (!e) return D2S_OK;
    D2S_ON_DEVICE(e->device);
    if (e->side) (void)hipStreamSynchronize(e->side);
    for (int i = 0; i < 4; ++i) if (e->ev_tap[i]) (void)hipEventDestroy(e->ev_tap[i]);
    for (int i = 0; i < 4; ++i) if (e->ev_ln[i]) (void)hipEventDestroy(e->ev_ln[i]);
    if (e->ev_side) (void)hipEventDestroy(e->ev_side);
    if (e->side) (void)hipStreamDestroy(e->side);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
    return D2S_OK;
}

extern "C" int d2s_engine_memory(const d2s_engine* e, uint64_t* bytes) {
    D2S_REQUIRE(e && bytes, "null pointer");
    *bytes = e->bytes;
    return D2S_OK;
}

extern "C" int d2s_model_forward(d2s_engine* e, const float* x, float* depth, int batch, void* stream) {
    D2S_REQUIRE(e && x && depth, "null pointer");
    if (!e->finalized) { set_error("d2s_model_forward before d2s_engine_finalize"); return D2S_E_STATE; }
    D2S_REQUIRE(batch >= 1 && batch <= e->maxB, "batch exceeds max_batch");
    D2S_REQUIRE(!e->d.temporal || batch == 1, "a Video-Depth-Anything engine is one stream: batch must be 1");
    D2S_ON_DEVICE(e->device);
    return forward(e, x, depth, batch, (hipStream_t)stream);
}

extern "C" int d2s_engine_calibrate(d2s_engine* e, const float* x, int batch, void* stream) {
    D2S_REQUIRE(e && x, "null pointer");
    if (!e->finalized) { set_error("d2s_engine_calibrate before d2s_engine_finalize"); return D2S_E_STATE; }
    D2S_REQUIRE(e->fp8, "d2s_engine_calibrate: not a D2S_PREC_FP8 engine");
    D2S_REQUIRE(batch >= 1 && batch <= e->maxB && !e->d.temporal, "bad batch (or a Video-Depth-Anything engine)");
    D2S_ON_DEVICE(e->device);
    hipStream_t st = (hipStream_t)stream;
    const int L = e->d.layers;
    // one bf16 forward over the calibration frames, recording max |activation| at the four quantisation sites per layer
    D2S_HIP(hipMemsetAsync(e->amax, 0, (size_t)L * NSITE * sizeof(float), st));
    const bool prof = e->prof_on;
    e->prof_on = false;
    e->calib = true;
    int rc = forward(e, x, e->depth_small, batch, st);
    e->calib = false;
    e->prof_on = prof;
    if (rc != D2S_OK) return rc;
    // headroom over the calibration frames' maxima (D2S_FP8_HEADROOM, default 1: later frames with larger activations
    // saturate at +-448 instead of wrapping; 1.25-1.5 trades a fraction of a bit of resolution for that margin)
    static const float headroom = getenv("D2S_FP8_HEADROOM") ? std::max(1.0f, (float)atof(getenv("D2S_FP8_HEADROOM"))) : 1.0f;
    std::vector<float> am((size_t)L * NSITE);
    D2S_HIP(hipMemcpyAsync(am.data(), e->amax, am.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    D2S_HIP(hipStreamSynchronize(st));
    for (int l = 0; l < L; ++l) {
        Layer& ly = e->L[l];
        for (int s = 0; s < NSITE; ++s) {
            float a = am[(size_t)l * NSITE + s];
            if (!(a > 0.f) || !std::isfinite(a)) { set_error("d2s_engine_calibrate: degenerate activation range"); return D2S_E_INVALID; }
            e->act_scale[(size_t)l * NSITE + s] = a * headroom / FP8_MAX;
        }
        for (int i = 0; i < 4; ++i) {                         // linear i reads site i (qkv <- LN1, proj <- attention, fc1 <- LN2, fc2 <- GELU)
            std::vector<float> dq(ly.sw[i].size());
            for (size_t n = 0; n < dq.size(); ++n) dq[n] = e->act_scale[(size_t)l * NSITE + i] * ly.sw[i][n];
            D2S_HIP(hipMemcpy(ly.deq[i], dq.data(), dq.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        // LN-folded linears read the raw residual: FC1 <- site 4 of this layer, QKV <- site 5 of the previous layer
        for (int i = 0; i < 2; ++i) {
            if (i == 0 && l == 0) continue;
            const float sraw = i == 0 ? e->act_scale[(size_t)(l - 1) * NSITE + 5] : e->act_scale[(size_t)l * NSITE + 4];
            std::vector<float> dq(ly.sw_ln[i].size());
            for (size_t n = 0; n < dq.size(); ++n) dq[n] = sraw * ly.sw_ln[i][n];
            D2S_HIP(hipMemcpy(ly.deq_ln[i], dq.data(), dq.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    e->fp8_ready = true;
    return D2S_OK;
}

extern "C" int d2s_engine_reset_stream(d2s_engine* e) {
    D2S_REQUIRE(e, "null engine");
    e->ema_init = 0;
    e->tm_init = 0; e->tm_head = 0;                     // VDA: drop the temporal window (next frame re-seeds it)
    return D2S_OK;
}

extern "C" int d2s_pipeline(d2s_engine* e, const uint8_t* frames, int batch, int H, int W, int depth_resolution,
                            const d2s_pre_params* pre, const d2s_post_params* pp, const d2s_sbs_params* sp, int use_ema,
                            void* out, int out_fmt, float* depth_full, void* stream) {
    D2S_REQUIRE(e && frames && pp && sp && out, "null pointer");
    if (!e->finalized) { set_error("d2s_pipeline before d2s_engine_finalize"); return D2S_E_STATE; }
    D2S_REQUIRE(batch >= 1 && batch <= e->maxB, "batch exceeds max_batch");
    D2S_REQUIRE(!e->d.temporal || batch == 1, "a Video-Depth-Anything engine is one stream: batch must be 1");
    // model-input shape of this frame size must be the engine's (reference: fixed at first frame, depth.py:1951-1953)
    D2S_REQUIRE(H > 0 && W > 0 && depth_resolution > 0, "bad frame shape");
    int longest = H > W ? H : W;
    int stride = 1;
    if (pre && pre->square) {
        // fixed-square branch (get_patch_size() is None: CAPTURE_MODE == "Window", reference depth.py:531-538, 1937-1946)
        if (e->h != depth_resolution || e->w != depth_resolution) {
            set_error("d2s_pipeline: the fixed-square branch needs a depth_resolution x depth_resolution engine"); return D2S_E_INVALID;
        }
    } else {   // _resize_patch_aligned_t integer logic (reference depth.py:677-689); Python round() = half-to-even
        double scale = longest != depth_resolution ? (double)depth_resolution / (double)longest : 1.0;
        int sh = std::max(1, (int)nearbyint(H * scale)), sw = std::max(1, (int)nearbyint(W * scale));
        auto nm = [&](int x) { int p = e->d.patch, down = (x / p) * p, up = down + p; return (std::abs(up - x) <= std::abs(x - down)) ? up : down; };
        if (std::max(1, nm(sh)) != e->h || std::max(1, nm(sw)) != e->w) {
            set_error("d2s_pipeline: frame maps to a model-input shape different from the engine's"); return D2S_E_INVALID;
        }
        stride = longest / (depth_resolution * 2);
        if (stride < 1) stride = 1;
    }
    D2S_ON_DEVICE(e->device);
    hipStream_t st = (hipStream_t)stream;
    {   // pre-process straight into the patch rows where that form exists (bilinear branch), else planes + the engine's patchify
        const bool fused = !e->taps && preprocess_patches_ok(e->prec, D2S_FMT_U8_HWC, pre, H, W, e->h, e->w, e->d.patch, e->patch.Kpad);
        if (fused) PROF(PC_PRE, 0, (double)batch * ((double)H * W * 3 + (double)e->h * e->w * 6),
                        launch_preprocess_patches(e->prec, frames, D2S_FMT_U8_HWC, batch, H, W, stride, pre, e->patchA, e->h, e->w, e->d.patch, e->patch.Kpad,
                                                  e->cls, e->pos, e->resid, e->N, e->d.hidden, st));
        if (!fused) PROF(PC_PRE, 0, (double)batch * ((double)H * W * 3 + (double)e->h * e->w * 12), d2s_preprocess(frames, D2S_FMT_U8_HWC, batch, H, W, e->pre_x, e->h, e->w, stride, pre, stream));
        RC(forward(e, fused ? nullptr : e->pre_x, e->depth_small, batch, st));
    }
    // post-process out of place (raw model output -> depth_post): few frames take the one-launch form (post.hip)
    PROF(PC_POST, 0, 0, d2s_post_process_to(e->depth_small, e->depth_post, batch, e->h, e->w, pp, e->post_ws, e->post_ws_bytes, stream));
    if (use_ema) {
        RC(ema_batch(e->depth_post, e->ema_state, e->ema_init, batch, e->h * e->w, pp->ema_alpha, st));
        e->ema_init = 1;
    }
    if (depth_full) RC(d2s_upsample_depth(e->depth_post, batch, e->h, e->w, depth_full, H, W, stream));
    {
        int oh = 0, ow = 0;
        RC(d2s_sbs_shape(H, W, sp, &oh, &ow));
        double obytes = (double)oh * ow * 3 * (out_fmt == D2S_FMT_U8_HWC ? 1 : 4);
        PROF(PC_WARP, 0, batch * ((double)H * W * 3 + (double)e->h * e->w * 4 + obytes),
             d2s_make_sbs(frames, D2S_FMT_U8_HWC, e->depth_post, e->h, e->w, batch, H, W, sp, out, out_fmt, stream));
    }
    return D2S_OK;
}

extern "C" int d2s_engine_tap(d2s_engine* e, const char* name, float* out, uint64_t out_elems, int* rows, int* cols, void* stream) {
    D2S_REQUIRE(e && name && out && rows && cols, "null pointer");
    if (!e->finalized || e->last_batch == 0) { set_error("d2s_engine_tap: no forward pass yet"); return D2S_E_STATE; }
    D2S_ON_DEVICE(e->device);
    hipStream_t st = (hipStream_t)stream;
    std::string n(name);
    const int D = e->d.hidden, F = e->d.fusion;
    if (n == "embeddings" || n.rfind("layer", 0) == 0) {
        if (!e->taps) { set_error("hidden-state taps need D2S_TAPS=1 at engine creation"); return D2S_E_STATE; }
        int l = n == "embeddings" ? 0 : atoi(n.c_str() + 5);
        D2S_REQUIRE(l >= 0 && l <= e->d.layers, "bad layer index");
        *rows = e->N; *cols = D;
        D2S_REQUIRE(out_elems >= (uint64_t)e->N * D, "tap buffer too small");
        D2S_HIP(hipMemcpyAsync(out, e->tap_hidden + (size_t)l * e->N * D, (size_t)e->N * D * 4, hipMemcpyDeviceToDevice, st));
        return D2S_OK;
    }
    if (n.rfind("neck_feat", 0) == 0) {
        int i = atoi(n.c_str() + 9);
        D2S_REQUIRE(i >= 0 && i < 4, "bad neck index");
        *rows = e->fH[i] * e->fW[i]; *cols = F;
        D2S_REQUIRE(out_elems >= (uint64_t)(*rows) * F, "tap buffer too small");
        return launch_to_f32(e->prec, e->feat[i], out, (long)(*rows) * F, st);
    }
    set_error("unknown tap: " + n);
    return D2S_E_INVALID;
}

extern "C" int d2s_engine_profile(d2s_engine* e, int enable) {
    D2S_REQUIRE(e, "null engine");
    e->prof_on = enable != 0;
    e->prof_recs.clear();
    e->prof_used = 0;
    return D2S_OK;
}

extern "C" int d2s_engine_profile_read(d2s_engine* e, int max_classes, double* ms, double* flops, double* bytes,
                                       int64_t* launches, int* n_classes) {
    D2S_REQUIRE(e && ms && flops && bytes && launches && n_classes && max_classes >= PC_N, "bad argument");
    D2S_ON_DEVICE(e->device);
    for (int c = 0; c < PC_N; ++c) { ms[c] = 0; flops[c] = 0; bytes[c] = 0; launches[c] = 0; }
    const char* dump = getenv("D2S_PROF_DUMP");             // tuning aid: one line per recorded launch on stderr
    int idx = 0;
    for (auto& r : e->prof_recs) {
        D2S_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        D2S_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.cls] += t; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes; launches[r.cls] += 1;
        if (dump && atoi(dump))
            fprintf(stderr, "[d2s-prof] %4d %-12s %8.2f us %9.3f GF %7.1f TF/s\n", idx, PC_NAMES[r.cls], t * 1e3, r.flops * 1e-9,
                    t > 0.f ? r.flops / (t * 1e-3) * 1e-12 : 0.0);
        ++idx;
    }
    *n_classes = PC_N;
    return D2S_OK;
}

extern "C" const char* d2s_profile_class_name(int cls) { return cls >= 0 && cls < PC_N ? PC_NAMES[cls] : ""; }
