// 3x3 convolution (stride 1, pad 1, NHWC, bf16) with the input tile resident in LDS -- second generation of gemm.hip's
// conv3_halo_kernel for the DPT neck / head convolutions of the batched step (HF DepthAnythingFeatureFusionLayer residual units,
// neck.convs, head conv1 / conv2; reference call site depth.py:1763-1781).
//
// What the first kernel spent its time on (profiles/r3_02: 6 VALU instructions per MFMA, 16 % LDS bank-conflict cycles, waves
// parked 54 % of their cycles): every A-fragment read computed its halo address at run time -- pixel index from the tap, an XOR
// swizzle from the pixel index, a multiply by the pixel stride.  Here
//   * the pixel stride in LDS is PADDED (CPP + 1 or + 2 chunks) instead of XOR-swizzled, so the address of tile row i, lane
//     (pixel fr, k group fg) at tap (ky, kx), K step ks is   base[i] + a COMPILE-TIME constant   -- the tap loop is fully
//     unrolled and every halo read is one ds_read_b128 with an immediate offset: no address VALU in the K loop at all;
//   * the weights stream through a descriptor-addressed LDS-DMA ring (per-lane offsets fixed, a K tile = one scalar offset);
//   * N = 32 (the head's conv2 with its fused conv3 + activation tail, MAP_HEAD) and N = 64 run here too: the implicit-GEMM
//     loader they used re-reads every input pixel nine times from L2 (head conv2 at batch 32: 5.6 GB of L2 -> LDS traffic for
//     0.6 GB of input) and spends ~19 VALU instructions per MFMA on per-chunk tap / bounds arithmetic.
// Same MFMA fragments (16 x 16 x 32 bf16, operands swapped: a lane ends with 4 consecutive n of one pixel), accumulators and
// epilogues as gemm_glds_kernel.
#include "gemm_epi.h"
#include <algorithm>

namespace d2s {

template <int N_> __device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory"); }

// CPP: 16-byte chunks per input pixel (C / 8: 8 | 16);  PST: pixel stride in LDS, in chunks;  BN: output channels per block;
// WM x WN waves over the 8 x 16 pixel tile (wave_m owns FM = 8 / WM tile rows) and the BN channels;  NS: weight ring stages.
template <int CPP, int PST, int BN, int WM, int WN, int NS>
__global__ void __launch_bounds__(64 * WM * WN)
conv3_halo2_kernel(GemmA a, const bf16_t* __restrict__ W, int M, int N, int Kpad, GemmEpi e, int xn) {
    constexpr int TW = 16, TH = 8, HWD = TW + 2, HPX = (TH + 2) * HWD;
    constexpr int NW = WM * WN, FM = TH / WM, FN = BN / WN / 16;
    constexpr int KPT = CPP / 8, NKT = 9 * KPT;             // K tiles (64 channels of one tap) per tap / in all
    constexpr int WST = BN * 8;                             // chunks per weight stage: BN rows x 128 bytes
    constexpr int BI = BN / (8 * NW);                       // LDS-DMA instructions per wave per stage (8 rows each)
    constexpr int PD = NS - 1;
    static_assert(BI >= 1 && BN % (8 * NW) == 0 && TH % WM == 0 && BN % (16 * WN) == 0, "bad tile split");
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * WST + HPX * PST];
    u32x4* const halo = lds + NS * WST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WN, wave_n = wid % WN;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int nimg = M / (a.Ho * a.Wo);
    int tm_, tn_;
    if (!tile_of_block(blockIdx.x, nimg * tiles_y * tiles_x, (N + BN - 1) / BN, xn, tm_, tn_)) return;
    const int b = tm_ / (tiles_y * tiles_x), ty0 = ((tm_ / tiles_x) % tiles_y) * TH, tx0 = (tm_ % tiles_x) * TW;
    const int bn0 = tn_ * BN;

    // ---- weights: descriptor-addressed LDS-DMA ring.  Lane -> (row lane / 8 of the instruction's 8, physical chunk lane % 8);
    // the row swizzle of gemm.hip ((row >> 1) & 7) sits on the source chunk and on the fragment read.
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W), 0, (unsigned)((long)((N + 255) / 256 * 256) * Kpad * 2), 0x00020000);
    unsigned voW[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int r = (i * NW + wid) * 8 + (lane >> 3);
        voW[i] = (unsigned)((long)(bn0 + r) * Kpad * 2) + (unsigned)((((lane & 7) ^ ((r >> 1) & 7))) * 16);
    }
#define C3_ISSUE_W(KT)                                                                                               \
    {                                                                                                                 \
        u32x4* st_ = lds + ((KT) % NS) * WST;                                                                         \
        _Pragma("unroll") for (int i = 0; i < BI; ++i) lds_dma16(rsW, st_ + (i * NW + wid) * 64, voW[i], (KT) * 128); \
    }
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < NKT) C3_ISSUE_W(t)

    // ---- the input halo, once: 10 x 18 pixels x C channels, zero outside the image, ReLU-on-load (pre-activation units)
    {
        const bf16_t* img = (const bf16_t*)a.ptr + (long)b * a.Hi * a.Wi * a.C;
        const short floor_ = a.relu ? (short)0 : (short)0x8000;      // max as int16: 0 = ReLU, most negative = identity
        typedef short s16x8_ __attribute__((ext_vector_type(8)));
        for (int idx = tid; idx < HPX * CPP; idx += 64 * NW) {
            const int p = idx / CPP, c = idx - p * CPP;
            const int hy = p / HWD, hx = p - hy * HWD;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) v = *(const u32x4*)(img + ((long)iy * a.Wi + ix) * a.C + c * 8);
            s16x8_ x = __builtin_bit_cast(s16x8_, v);
            x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
            halo[p * PST + c] = __builtin_bit_cast(u32x4, x);
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    // A fragment of tile row i: halo pixel (row i + ky, column fr + kx), chunk 8 sub + 4 ks + fg  =  hb[i] + constant
    const u32x4* hb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) hb[i] = halo + ((wave_m * FM + i) * HWD + fr) * PST + fg;
    // W fragment rows of this wave: row j * 16 + fr of its BN / WN rows, chunk (4 ks + fg) ^ swizzle(row)
    int wro[FN], wsw[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) { const int r = wave_n * (BN / WN) + j * 16 + fr; wro[j] = r * 8; wsw[j] = (r >> 1) & 7; }

    static_for<NKT>([&](auto ktc) {
        constexpr int kt = decltype(ktc)::value;
        constexpr int tap = kt / KPT, sub = kt % KPT, ky = tap / 3, kx = tap % 3;
        // my W loads of tile kt have landed (and, the first time, my halo stores); then everybody's
        if constexpr (kt + PD - 1 < NKT) c3_wait_vm<(PD - 1) * BI>(); else c3_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if constexpr (kt + PD < NKT) C3_ISSUE_W(kt + PD)
        const u32x4* B_l = lds + (kt % NS) * WST;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fb[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = B_l[wro[j] + ((ks * 4 + fg) ^ wsw[j])];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const u32x4 fa = hb[i][(ky * HWD + kx) * PST + sub * 8 + ks * 4];
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk(acc[i][j], fb[j], fa, bf16_t());
            }
        }
    });
#undef C3_ISSUE_W

    // ---- epilogue: tile row -> output pixel (ty0 + row, tx0 + fr)
    const int x = tx0 + fr;
    if constexpr (WN == 1) {
        // MAP_HEAD: depth[m] = act(b3 + sum_n w3[n] * relu(acc[m][n] + bias[n]))  (conv2 -> ReLU -> conv3 1x1 -> ReLU | sigmoid)
        if (e.map == MAP_HEAD) {
            static_for<FM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int y = ty0 + wave_m * FM + i;
                float s = 0.f;
                static_for<FN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int n0 = j * 16 + fg * 4;
                    if (n0 < N) {
                        float bb[4], w[4];
                        load4(e.bias + n0, bb); load4(e.scale + n0, w);
                        s += fmaxf(acc[i][j][0] + bb[0], 0.f) * w[0] + fmaxf(acc[i][j][1] + bb[1], 0.f) * w[1] +
                             fmaxf(acc[i][j][2] + bb[2], 0.f) * w[2] + fmaxf(acc[i][j][3] + bb[3], 0.f) * w[3];
                    }
                });
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (fg == 0 && y < a.Ho && x < a.Wo) ((float*)e.out)[((long)b * a.Ho + y) * a.Wo + x] = head_activation(s + e.head_b3, e.head_max_depth);
            });
            return;
        }
    }
    static_for<FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int y = ty0 + wave_m * FM + i;
        if (y < a.Ho && x < a.Wo) {
            const int m = (b * a.Ho + y) * a.Wo + x;
            static_for<FN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
                if (n0 < N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    epilogue_dispatch<bf16_t>(e, m, n0, v);
                }
            });
        }
    });
}

// ================================================================================================
// The head's last convolution (conv2: 64 -> 32 channels at model resolution, with conv3 1x1 + ReLU | sigmoid fused, MAP_HEAD):
// persistent blocks, weights in REGISTERS, halo double-buffered.
// PMC of the one-shot blocks above on this launch (profiles/r3_03): a block lives 18 000 cycles for 1 150 cycles of MFMA work --
// it loads its halo from HBM, waits, computes nine short K tiles between barriers and leaves; twelve resident waves per CU
// cannot cover that.  Here one 8-wave block per CU walks the 16 x 16-pixel tiles of the whole batch:
//   * all of W (32 x 576 bf16 = 36 KiB) lives in registers as MFMA fragments (9 taps x 2 K steps x 2 n blocks x 4 VGPRs = 144),
//     loaded once per block: the K loop reads ONLY halo fragments from LDS and has no barrier;
//   * the halo of the NEXT tile is fetched into registers before the current tile is computed and stored to the other LDS buffer
//     after it: HBM latency sits under 72 MFMAs per wave; one barrier per tile;
//   * padded pixel stride (10 chunks for 8): compile-time LDS offsets, conflict-free ds_read_b128 (PMC: 0 conflict cycles).
// At this arithmetic intensity (N = 32) the launch is HBM-bound once the latency is hidden: 41.5 KB of halo per 256 pixels.
// ================================================================================================
template <int DUMMY>
__global__ void __launch_bounds__(512)
conv3_head_kernel(GemmA a, const bf16_t* __restrict__ W, int N, int Kpad, GemmEpi e, int ntiles) {
    constexpr int CPP = 8, PST = 10, TH = 16, TW = 16, HWD = TW + 2, HPX = (TH + 2) * HWD, HALO = HPX * PST;
    constexpr int NCH = HPX * CPP, NLD = (NCH + 511) / 512;       // halo chunks, loads per thread (6)
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * HALO];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;

    // ---- W fragments, once: MFMA row j * 16 + fr, K step (tap, ks) -> chunk ks * 4 + fg of the tap's 64 channels
    u32x4 wf[9][2][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = j * 16 + fr;
                wf[tap][ks][j] = n < N ? *(const u32x4*)(W + (long)n * Kpad + tap * 64 + (ks * 4 + fg) * 8) : (u32x4){0u, 0u, 0u, 0u};
            }
    // epilogue constants of this lane's columns (conv2 bias, conv3 weights)
    float cb[2][4], cw[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n0 = j * 16 + fg * 4;
        if (n0 < N) { load4(e.bias + n0, cb[j]); load4(e.scale + n0, cw[j]); }
        else { cb[j][0] = cb[j][1] = cb[j][2] = cb[j][3] = 0.f; cw[j][0] = cw[j][1] = cw[j][2] = cw[j][3] = 0.f; }
    }

    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    const short floor_ = a.relu ? (short)0 : (short)0x8000;
    u32x4 hr[NLD];
    auto tile_org = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / (tiles_y * tiles_x);
        const int r = t - b * (tiles_y * tiles_x);
        ty0 = (r / tiles_x) * TH; tx0 = (r % tiles_x) * TW;
    };
    auto load_halo = [&](int t) {
        int b, ty0, tx0;
        tile_org(t, b, ty0, tx0);
        const bf16_t* img = (const bf16_t*)a.ptr + (long)b * a.Hi * a.Wi * a.C;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 512;
            const int p = idx >> 3, c = idx & 7;
            const int hy = p / HWD, hx = p - hy * HWD;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            hr[k] = (u32x4){0u, 0u, 0u, 0u};
            if (idx < NCH && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) hr[k] = *(const u32x4*)(img + ((long)iy * a.Wi + ix) * a.C + c * 8);
        }
    };
    auto store_halo = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 512;
            if (idx < NCH) {
                s16x8_ x = __builtin_bit_cast(s16x8_, hr[k]);
                x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
                lds[buf * HALO + (idx >> 3) * PST + (idx & 7)] = __builtin_bit_cast(u32x4, x);
            }
        }
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
    load_halo(t);
    store_halo(0);
    __syncthreads();
    // A fragment of this wave's tile row i (rows 2 wid, 2 wid + 1): pixel (row + ky, fr + kx), chunk 4 ks + fg = hb + constant
    const int hb0 = ((wid * 2) * HWD + fr) * PST + fg;
    int buf = 0;
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < ntiles) load_halo(tn);                        // in flight under the 72 MFMAs below
        const u32x4* hp = lds + buf * HALO + hb0;
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        static_for<9>([&](auto tc) {
            constexpr int tap = decltype(tc)::value, ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const u32x4 fa = hp[((i + ky) * HWD + kx) * PST + ks * 4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma_chunk(acc[i][j], wf[tap][ks][j], fa, bf16_t());
                }
        });
        // ---- epilogue: depth = act(b3 + sum_n w3[n] relu(acc + bias[n]))
        int b, ty0, tx0;
        tile_org(t, b, ty0, tx0);
        const int x = tx0 + fr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = ty0 + wid * 2 + i;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                s += fmaxf(acc[i][j][0] + cb[j][0], 0.f) * cw[j][0] + fmaxf(acc[i][j][1] + cb[j][1], 0.f) * cw[j][1] +
                     fmaxf(acc[i][j][2] + cb[j][2], 0.f) * cw[j][2] + fmaxf(acc[i][j][3] + cb[j][3], 0.f) * cw[j][3];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (fg == 0 && y < a.Ho && x < a.Wo) ((float*)e.out)[((long)b * a.Ho + y) * a.Wo + x] = head_activation(s + e.head_b3, e.head_max_depth);
        }
        if (tn < ntiles) store_halo(buf ^ 1);
        __syncthreads();                                       // tile t is read out, tile t + 1's halo is in place
        buf ^= 1;
    }
}

// (A persistent form of the 128-channel kernel -- one 8-wave block per CU, next halo prefetched by four "halo" waves while four
// "weight" waves ran a 3-stage LDS-DMA ring across tile boundaries -- was built and measured at batch 32: 226-246 us per
// 84 x 148 convolution against 168-185 us for the one-shot blocks above, head conv1 446 against 370.  With 16 KiB weight stages the
// ring is two K tiles deep at most beside two halo buffers, i.e. ~1 000 cycles of cover for an LDS-DMA that needs ~2 000 under
// load; two independent blocks per CU hide that better than one deeper-pipelined one.  Removed.)

// Eligible: bf16, stride 1, same-size output, C = 64 | 128, K = 9 C, plain row mapping or the fused head, enough tiles to fill the
// chip.  D2S_NO_HALO2=1 keeps the first-generation kernels (the parity tests run both).
bool launch_conv3_halo2(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st) {
    static const bool off = getenv("D2S_NO_HALO2") && atoi(getenv("D2S_NO_HALO2")) != 0;
    if (off) return false;
    if (a.mode != A_CONV3 || a.stride != 1 || a.Hi != a.Ho || a.Wi != a.Wo || (a.C != 64 && a.C != 128) || K != 9 * a.C) return false;
    if (!(e.map == MAP_ROWS || e.map == MAP_HEAD) || e.rows_per_img || e.ln_stats || e.stats_out || e.deq || e.ksplit > 1) return false;
    if (e.map == MAP_HEAD && N > 32) return false;
    if (N != 32 && N != 64 && (N & 127)) return false;
    const int nimg = M / (a.Ho * a.Wo);
    if ((long)nimg * a.Ho * a.Wo != M) return false;
    static const bool no_persist = getenv("D2S_NO_HEADP") && atoi(getenv("D2S_NO_HEADP")) != 0;
    if (!no_persist && e.map == MAP_HEAD && a.C == 64 && N <= 32 && (long)nimg * cdiv(a.Ho, 16) * cdiv(a.Wo, 16) >= 256) {
        static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
        const int ntiles = nimg * cdiv(a.Ho, 16) * cdiv(a.Wo, 16);
        hipLaunchKernelGGL((conv3_head_kernel<0>), dim3(std::min(ncu, ntiles)), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e, ntiles);
        return true;
    }
    const long tiles_m = (long)nimg * cdiv(a.Ho, 8) * cdiv(a.Wo, 16);
    const int bn = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
    if (tiles_m * cdiv(N, bn) < 512) return false;                // small maps: latency-bound, the small-tile kernels do better
    if ((long)gemm_npad(N) * Kpad * 2 >= (1L << 31)) return false;
    static const int pst16 = getenv("D2S_HALO2_PST") ? atoi(getenv("D2S_HALO2_PST")) : 17;      // tuning aid: 17 (2 blocks / CU) | 18 (conflict-free)
    GemmEpi e1 = e; e1.ksplit = 1;
    unsigned grid = 0;
    const int xn = pick_xn((int)tiles_m, cdiv(N, bn), bn, Kpad, 2, grid);
#define C3_LAUNCH(CPP_, PST_, BN_, WM_, WN_, NS_)                                                                                      \
    hipLaunchKernelGGL((conv3_halo2_kernel<CPP_, PST_, BN_, WM_, WN_, NS_>), dim3(grid), dim3(64 * WM_ * WN_), 0, st, a, (const bf16_t*)W, M, N, Kpad, e1, xn)
    if (a.C == 128) {
        if (bn == 128) { if (pst16 == 18) C3_LAUNCH(16, 18, 128, 2, 4, 2); else C3_LAUNCH(16, 17, 128, 2, 4, 2); }
        else if (bn == 64) C3_LAUNCH(16, 17, 64, 4, 2, 3);
        else C3_LAUNCH(16, 17, 32, 4, 1, 3);
    } else {
        if (bn == 128) C3_LAUNCH(8, 10, 128, 2, 4, 3);
        else if (bn == 64) C3_LAUNCH(8, 10, 64, 4, 2, 3);
        else C3_LAUNCH(8, 10, 32, 4, 1, 3);
    }
#undef C3_LAUNCH
    return true;
}

}  // namespace d2s
