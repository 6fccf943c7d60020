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

// every kernel of this file that warms its argument lines (KERNARG_WARM, common.h) takes at least a GemmA and a GemmEpi by value
static_assert(sizeof(GemmA) + sizeof(GemmEpi) + 8 >= KERNARG_WARM_BYTES, "KERNARG_WARM reads past the kernarg segment");


template <int N_> __device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory"); }

// Bank rule of the halo reads.  ds_read_b128 serves the wave in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table): with lane = 16 fg + fr a group holds fr in S = {0-3, 12-15} of
// one k group and fr in S' = {4-11} of the NEXT k group.  Chunk slot mod 16 of a lane = pixel * PST + chunk.
//   * even PST (10, C = 64): pixel * 10 is even, the neighbouring k group is one chunk on -> evens and odds, conflict-free;
//   * odd PST (17, C = 128; 18 would need 2 x 84.6 KB of LDS -- one block per CU instead of two): slot = pixel + chunk mod 16,
//     and with pixel = fr the two halves {S + k} and {S' + k + 1} collide in one bank quad per group (PMC: 39 % of the LDS
//     cycles were conflict cycles, profiles/r3_02).  Fix: lanes of S take the EVEN tile columns and lanes of S' the odd ones
//     (which lane owns which pixel is free: the epilogue follows), and the chunks of a pixel are stored with bits 0 / 1
//     swapped, so that neighbouring k groups sit TWO slots apart: evens + k and odds + k + 2 never meet.
template <int PST> __device__ __forceinline__ int c3_lane_pixel(int fr) {
    if constexpr ((PST & 1) == 0) return fr;
    else return fr < 4 ? 2 * fr : (fr < 12 ? 2 * (fr - 4) + 1 : 2 * (fr - 8));
}
template <int PST> __device__ __forceinline__ int c3_chunk_slot(int c) {
    if constexpr ((PST & 1) == 0) return c;
    else return (c & ~3) | ((c & 1) << 1) | ((c >> 1) & 1);
}

// CPP: 16-byte chunks per input pixel (C / 8: 8 | 16);  PST: pixel stride in LDS, in chunks;  BN: output channels per block;
// WM x WN waves over the 8 x 16 pixel tile (wave_m owns FM = 8 / WM tile rows) and the BN channels;  NS: weight ring stages.
template <int CPP, int PST, int BN, int WM, int WN, int NS, int HG = 3 /* halo chunks a thread keeps in flight per pass of the fill */>
__global__ void __launch_bounds__(64 * WM * WN)
conv3_halo2_kernel(GemmA a, const bf16_t* __restrict__ W, int M, int N, int Kpad, GemmEpi e, int xn) {
    KERNARG_WARM(kaw_)                                   // all argument lines in one round trip (common.h)
    KERNARG_WARM_END(kaw_)
    constexpr int TW = 16, TH = 8, HWD = TW + 2, HPX = (TH + 2) * HWD;
    constexpr int NW = WM * WN, FM = TH / WM, FN = BN / WN / 16;
    constexpr int KPT = CPP / 8, NKT = 9 * KPT;             // K tiles (64 channels of one tap) per tap / in all
    constexpr int WST = BN * 8;                             // chunks per weight stage: BN rows x 128 bytes
    constexpr int BI = BN / (8 * NW);                       // LDS-DMA instructions per wave per stage (8 rows each)
    constexpr int PD = NS - 1;
    static_assert(BI >= 1 && BN % (8 * NW) == 0 && TH % WM == 0 && BN % (16 * WN) == 0, "bad tile split");
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * WST + HPX * PST];
    D2S_POISON_LDS(lds, NS * WST + HPX * PST)
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

    // ---- the input halo, once: 10 x 18 pixels x C channels, zero outside the image, ReLU-on-load (pre-activation units); a.ups: the
    // align_corners up-sample in front of this convolution happens here (gemm_epi.h conv_halo_fill)
    {
        const short floor_ = a.relu ? (short)0 : (short)0x8000;      // max as int16: 0 = ReLU, most negative = identity
        typedef short s16x8_ __attribute__((ext_vector_type(8)));
        conv_halo_fill<bf16_t, 64 * NW, HG>(a, b, ty0, tx0, HWD, HPX, CPP, tid, halo,
            [&](int p, int c) { return p * PST + c3_chunk_slot<PST>(c); },
            [&](u32x4 v) {
                s16x8_ x = __builtin_bit_cast(s16x8_, v);
                x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
                return __builtin_bit_cast(u32x4, x);
            });
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int px = c3_lane_pixel<PST>(fr);                 // tile column of this lane's pixel (see c3_lane_pixel)
    // A fragment of tile row i: halo pixel (row i + ky, column px + kx), chunk 8 sub + 4 ks + fg  =  hb[i] + constant
    const u32x4* hb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) hb[i] = halo + ((wave_m * FM + i) * HWD + px) * PST + c3_chunk_slot<PST>(fg);
    // W fragment rows of this wave: row j * 16 + fr of its BN / WN rows, chunk (4 ks + fg) ^ swizzle(row)
    int wro[FN], wsw[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) { const int r = wave_n * (BN / WN) + j * 16 + fr; wro[j] = r * 8; wsw[j] = (r >> 1) & 7; }

    static_for<NKT>([&](auto ktc) {
        constexpr int kt = decltype(ktc)::value;
        constexpr int tap = kt / KPT, sub = kt % KPT, ky = tap / 3, kx = tap % 3;
        // my W loads of tile kt have landed (and, the first time, my halo stores); then everybody's
        // (tiles kt .. min(kt + PD, NKT) - 1 are in flight; vmcnt retires in order: all but tile kt may stay out.  PD >= NKT -- the
        //  "deep" instantiations of the small maps -- means every weight tile was requested in the prologue)
        c3_wait_vm<((kt + PD - 1 < NKT) ? PD - 1 : NKT - 1 - kt) * BI>();
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

    // ---- epilogue: tile row -> output pixel (ty0 + row, tx0 + px)
    const int x = tx0 + px;
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
    EpiCols cols[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
        if (n0 < N) epi_cols_load(e, n0, cols[j]);
    }
    float pre[FM][FN][4];                                   // residual values, all requested before the first store (gemm_epi.h)
    const bool pre_on = epi_res1_ahead(e);
    if (pre_on) {
        static_for<FM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int y = ty0 + wave_m * FM + i;
            static_for<FN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
                if (y < a.Ho && x < a.Wo && n0 < N) epi_res1_load<bf16_t>(e, (b * a.Ho + y) * a.Wo + x, n0, pre[i][j]);
            });
        });
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
                    epilogue_dispatch<bf16_t>(e, m, n0, v, false, pre_on ? pre[i][j] : nullptr, &cols[j]);
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
// UPS = 1: the align_corners bilinear up-sample in front of the convolution (HF DepthAnythingDepthEstimationHead: conv1 -> interpolate
// -> conv2) happens in the halo loader.  Stand-alone it writes and the convolution re-reads the largest activation of the model
// (294 x 518 x 64 bf16 = 19.5 MB per frame: 292 + 223 us at batch 32, both HBM-bound); here a block fetches the SOURCE pixels under
// its next tile (at most 13 x 13 for scales <= 0.6: 21 KB instead of 41.5 KB of halo, and from a map a quarter the size that stays
// in L2) into registers during the current tile, parks them in an LDS staging area afterwards and interpolates the 18 x 18 halo
// from there with bilerp1 -- the same expression, the same rounding to bf16 as the stand-alone kernel (bit-identical results).
template <int UPS>
__global__ void __launch_bounds__(512)
conv3_head_kernel(GemmA a, const bf16_t* __restrict__ W, int N, int Kpad, GemmEpi e, int ntiles) {
    KERNARG_WARM(kaw_)                                   // all argument lines in one round trip (common.h)
    KERNARG_WARM_END(kaw_)
    constexpr int CPP = 8, PST = 10, TH = 16, TW = 16, HWD = TW + 2, HPX = (TH + 2) * HWD, HALO = HPX * PST;
    constexpr int NCH = HPX * CPP, NLD = (NCH + 511) / 512;       // halo chunks, loads per thread (6)
    constexpr int SR = 13, SCH = SR * SR * CPP, NSL = (SCH + 511) / 512;   // UPS: source window (pixels per side), its chunks, loads per thread (3)
    constexpr int SPS = 2 * CPP + 1;                         // UPS: staging stride of a source pixel in 16-byte units (odd: the taps of neighbouring pixels spread over the banks)
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * HALO + (UPS ? SR * SR * SPS : 0)];
    D2S_POISON_LDS(lds, 2 * HALO + (UPS ? SR * SR * SPS : 0))
    f32x4* const stg = (f32x4*)(lds + 2 * HALO);            // UPS: the source window as floats (unpacked once, tapped ~8 times): [pixel][64]

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
    // UPS: first source row / column under the halo of the tile at (ty0, tx0); the window is SR x SR pixels from there
    auto src_org = [&](int ty0, int tx0, int& rs0, int& cs0) {
        rs0 = linear_tap(ty0 > 0 ? ty0 - 1 : 0, a.usy, a.Hs, true).i0;
        cs0 = linear_tap(tx0 > 0 ? tx0 - 1 : 0, a.usx, a.Ws, true).i0;
    };
    auto load_halo = [&](int t) {
        int b, ty0, tx0;
        tile_org(t, b, ty0, tx0);
        if constexpr (UPS) {
            int rs0, cs0;
            src_org(ty0, tx0, rs0, cs0);
            const bf16_t* img = (const bf16_t*)a.ptr + (long)b * a.Hs * a.Ws * a.C;
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const int idx = tid + k * 512;
                const int p = idx >> 3, c = idx & 7;
                const int sr = p / SR, sc = p - sr * SR;
                const int row = rs0 + sr < a.Hs ? rs0 + sr : a.Hs - 1, col = cs0 + sc < a.Ws ? cs0 + sc : a.Ws - 1;
                hr[k] = (u32x4){0u, 0u, 0u, 0u};
                if (idx < SCH) hr[k] = *(const u32x4*)(img + ((long)row * a.Ws + col) * a.C + c * 8);
            }
        } else {
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
        }
    };
    // t: the tile whose data sits in hr (UPS needs its origin again)
    auto store_halo = [&](int buf, int t) {
        if constexpr (UPS) {
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const int idx = tid + k * 512;
                if (idx < SCH) {
                    const u32x4 v = hr[k];
                    stg[(idx >> 3) * SPS + 2 * (idx & 7)] = (f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
                    stg[(idx >> 3) * SPS + 2 * (idx & 7) + 1] = (f32x4){__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xffff0000u), __uint_as_float(v[3] << 16), __uint_as_float(v[3] & 0xffff0000u)};
                }
            }
            __syncthreads();
            int b, ty0, tx0, rs0, cs0;
            tile_org(t, b, ty0, tx0);
            src_org(ty0, tx0, rs0, cs0);
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 512;
                if (idx < NCH) {
                    const int p = idx >> 3, c = idx & 7;
                    const int hy = p / HWD, hx = p - hy * HWD;
                    const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
                    u32x4 r = {0u, 0u, 0u, 0u};
                    if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
                        const Tap ty = linear_tap(iy, a.usy, a.Hs, true), tx = linear_tap(ix, a.usx, a.Ws, true);
                        const int r0 = (ty.i0 - rs0) * SR, r1 = (ty.i1 - rs0) * SR, c0 = tx.i0 - cs0, c1 = tx.i1 - cs0;
                        const f32x4* q00 = stg + (r0 + c0) * SPS + 2 * c; const f32x4* q01 = stg + (r0 + c1) * SPS + 2 * c;
                        const f32x4* q10 = stg + (r1 + c0) * SPS + 2 * c; const f32x4* q11 = stg + (r1 + c1) * SPS + 2 * c;
                        typedef float f2_ __attribute__((ext_vector_type(2)));
                        const f2_ w0x = {tx.w0, tx.w0}, w1x = {tx.w1, tx.w1}, w0y = {ty.w0, ty.w0}, w1y = {ty.w1, ty.w1};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {          // bilerp1, two channels per instruction
                            const f32x4 a00 = q00[h], a01 = q01[h], a10 = q10[h], a11 = q11[h];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const f2_ v00 = {a00[2 * u], a00[2 * u + 1]}, v01 = {a01[2 * u], a01[2 * u + 1]};
                                const f2_ v10 = {a10[2 * u], a10[2 * u + 1]}, v11 = {a11[2 * u], a11[2 * u + 1]};
                                const f2_ top = __builtin_elementwise_fma(w1x, v01, w0x * v00);
                                const f2_ bot = __builtin_elementwise_fma(w1x, v11, w0x * v10);
                                const f2_ o = __builtin_elementwise_fma(w1y, bot, w0y * top);
                                r[2 * h + u] = pk_bf16(o[0], o[1]);
                            }
                        }
                    }
                    s16x8_ x = __builtin_bit_cast(s16x8_, r);
                    x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
                    lds[buf * HALO + p * PST + c] = __builtin_bit_cast(u32x4, x);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 512;
                if (idx < NCH) {
                    s16x8_ x = __builtin_bit_cast(s16x8_, hr[k]);
                    x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
                    lds[buf * HALO + (idx >> 3) * PST + (idx & 7)] = __builtin_bit_cast(u32x4, x);
                }
            }
        }
    };

    // tile walk: XCD x (= blockIdx % 8) owns the contiguous run [x per, (x + 1) per) and its CUs take consecutive tiles of it, so the
    // overlapping windows of neighbouring tiles meet in ONE L2 (in launch order -- tile = block + k grid -- neighbours sat on eight
    // XCDs: PMC with the up-sample folded in, profiles/r3_06: L2 hit 0.15, 376 MB fetched per launch for a 204 MB source)
    const bool xcd_walk = (gridDim.x & 7) == 0;
    const int xcd_ = blockIdx.x & 7, slot_ = blockIdx.x >> 3, nslot_ = gridDim.x >> 3, per_ = (ntiles + 7) >> 3;
    auto tile_at = [&](int k) {
        if (!xcd_walk) { const int tt = blockIdx.x + k * gridDim.x; return tt < ntiles ? tt : -1; }
        const int j = slot_ + k * nslot_, tt = xcd_ * per_ + j;
        return (j < per_ && tt < ntiles) ? tt : -1;
    };
    int kk = 0;
    int t = tile_at(0);
    if (t < 0) return;
    load_halo(t);
    store_halo(0, t);
    __syncthreads();
    // A fragment of this wave's tile row i (rows 2 wid, 2 wid + 1): pixel (row + ky, fr + kx), chunk 4 ks + fg = hb + constant
    const int hb0 = ((wid * 2) * HWD + fr) * PST + fg;
    int buf = 0;
    for (; t >= 0;) {
        const int tn = tile_at(++kk);
        if (tn >= 0) load_halo(tn);                            // in flight under the 72 MFMAs below
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
        if (tn >= 0) store_halo(buf ^ 1, tn);
        __syncthreads();                                       // tile t is read out, tile t + 1's halo is in place
        buf ^= 1;
        t = tn;
    }
}

// ================================================================================================
// conv3_c128_ups_kernel (round 5): the head's conv1 (C = 128 -> 64 channels, the fusion stage's x2 align_corners up-sample folded into
// its loader) at batch.  As a one-shot conv3_halo2 launch (8 x 16-pixel tiles x 64 channels, 12 768 blocks at batch 32) every block
// streams its 147 KB of weights through LDS for 128 pixels: 2.5 GB of L2 -> LDS traffic per launch, 334-378 us = 0.25-0.28 of the MFMA
// peak, the largest single convolution of the batched step.  Here ONE persistent 8-wave block per CU, cut like conv3_head_ups_kernel:
//   * waves 0-3, CONSUMERS (one per SIMD): W in REGISTERS -- wave nq owns channels 16 nq .. + 15 of all 8 tile rows: 9 taps x 4 K groups
//     x 4 VGPRs = 144 -- so the K loop reads only halo fragments from LDS; a halo row's fragment is read once (8 ahead of its MFMAs) and
//     feeds the up-to-three output rows that tap it (120 reads per 288 MFMAs).  Per accumulator the MFMA order is conv3_halo2's
//     (tap-major, channel groups ascending), bias add and rounding are epilogue4's: bit-identical results (tests/test_gpu_parity.py).
//     The finished tile goes to an LDS patch (chunk index XOR pixel) and leaves as whole 128-byte rows during the next tile's first phase
//     (stored as 32-byte pieces straight from the accumulators of four waves the launch took 438 us).
//   * waves 4-7, PRODUCERS: the SOURCE window under a tile's halo (<= 7 x 11 pixels for scales <= 0.5) is fetched raw two tiles ahead,
//     parked in one of two LDS staging buffers with per-tile tap tables (offsets pre-multiplied), and the 10 x 18 halo of the NEXT tile is
//     interpolated from there into the other halo buffer with lerp_chunk's own expression on the same four chunks while the consumers
//     compute; a thread's 12 halo chunks are the same in every tile (packed once).
//   * two block barriers per tile (A: halo / staging / patch complete; B: patch read out, half the next halo in place); the roles run
//     their own loops (one loop with a role branch kept the 144 weight registers alive through the producers' code: 119 spills).
// LDS: 2 x 48 960 B halo + 2 x 20 944 B staging + 16 384 B patch + tables = 157 KB.
// MEASURED (batch 32; one-shot blocks 338-378 us on the same boxes): **296-315 us** (one-shot 345-360 on those boxes).  How it got there --
// every cut bit-identical: all eight waves in lock-step (K loop, then interpolation) 341-372 us; the interpolation spread over the K loop
// inside each wave 422 us (32 spilled registers); producer / consumer waves with one chunk pair interpolated at a time 330-351 us
// (cut-point builds: consumers alone 185 us, producers alone 218-240 us, together 342: ONE wave per SIMD and role has nobody to cover its
// LDS round trips, and together the round trips get longer; PMC, profiles/r5_08: MFMA busy 0.28, waves waiting 0.45 of their cycles);
// all table reads and all 24 tap reads of a six-chunk half-tile in flight together: 296-315 us (producers alone 205).  By instruction
// count a tile is 2.8 us of issue slots per SIMD (137 us per launch): what is left is latency in both roles' single waves -- the K loop
// alone (cut 3: no interpolation, no output path) was 196-207 us = 31 cycles per v_mfma_f32_16x16x32_bf16 against ~17 back to back: the
// compiler had sunk the "8 ahead" fragment reads to just behind the previous use of their registers (s_waitcnt lgkmcnt(1) before every
// MFMA group); as inline asm with hand-counted lgkmcnt (6 in flight) the K loop alone takes 170 us; the launch 296-307 (producers: 205 alone).
// ================================================================================================
template <int OFF> __device__ __forceinline__ void c3_lds_read_b128(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N_> __device__ __forceinline__ void c3_lgkm_wait(u32x4& frag) {     // the fragment rides along: its consumers cannot move above the wait
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N_));
}
__global__ void __launch_bounds__(512)
conv3_c128_ups_kernel(GemmA a, const bf16_t* __restrict__ W, int N, int Kpad, GemmEpi e, int ntiles) {
    KERNARG_WARM(kaw_)
    KERNARG_WARM_END(kaw_)
    constexpr int CPP = 16, PST = 17, TH = 8, TW = 16, HWD = TW + 2, HPX = (TH + 2) * HWD, HALO = HPX * PST;
    constexpr int NCH = HPX * CPP;                             // halo chunks (2 880)
    constexpr int SRY = 7, SRX = 11, SPX = SRY * SRX, SCH = SPX * CPP;   // source window (scales <= 0.5), its chunks (1 232)
    constexpr int SPS = CPP + 1, STG = SPX * SPS;              // staging stride of a source pixel (odd: neighbouring pixels' taps spread over the banks)
    constexpr int NSLP = (SCH + 255) / 256, NLDP = (NCH + 255) / 256;      // per PRODUCER thread (256 of them): 5 source chunks, 12 halo chunks
    constexpr int PATCH = TH * TW * 8;                         // output tile as bf16 rows: 128 pixels x 8 chunks
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * HALO + 2 * STG + PATCH];
    // tap tables of the staged tiles [staging buffer][rows | columns]: the two source offsets of a halo row / column, already in staging
    // chunks ((i - origin) * SRX * SPS for rows, * SPS for columns), x < 0 outside the image; and the two weights
    __shared__ int2 tap_i[2][2][HWD + TH + 2];
    __shared__ float2 tap_w[2][2][HWD + TH + 2];
    D2S_POISON_LDS(lds, 2 * HALO + 2 * STG + PATCH)
    u32x4* const stg0 = lds + 2 * HALO;
    u32x4* const patch = lds + 2 * HALO + 2 * STG;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wid < 4;
    const int nq = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int px = c3_lane_pixel<PST>(fr);                     // tile column of this lane's pixel
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int ptid = tid - 256;                                // producer thread number

    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    const short floor_ = a.relu ? (short)0 : (short)0x8000;
    auto tile_org = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / (tiles_y * tiles_x);
        const int r = t - b * (tiles_y * tiles_x);
        ty0 = (r / tiles_x) * TH; tx0 = (r % tiles_x) * TW;
    };
    auto src_org = [&](int ty0, int tx0, int& rs0, int& cs0) {
        rs0 = linear_tap(ty0 > 0 ? ty0 - 1 : 0, a.usy, a.Hs, true).i0;
        cs0 = linear_tap(tx0 > 0 ? tx0 - 1 : 0, a.usx, a.Ws, true).i0;
    };
    // ---- the loader (the 256 producer threads)
    u32x4 hr[NSLP];
    auto load_src = [&](int t) {                               // raw source window of tile t -> registers
        int b, ty0, tx0, rs0, cs0;
        tile_org(t, b, ty0, tx0);
        src_org(ty0, tx0, rs0, cs0);
        const bf16_t* img = (const bf16_t*)a.ptr + (long)b * a.Hs * a.Ws * a.C;
#pragma unroll
        for (int k = 0; k < NSLP; ++k) {
            const int idx = ptid + k * 256;
            const int p = idx >> 4, c = idx & 15;
            const int sr = p / SRX, sc = p - sr * SRX;
            const int row = rs0 + sr < a.Hs ? rs0 + sr : a.Hs - 1, col = cs0 + sc < a.Ws ? cs0 + sc : a.Ws - 1;
            hr[k] = (u32x4){0u, 0u, 0u, 0u};
            if (idx < SCH) hr[k] = *(const u32x4*)(img + ((long)row * a.Ws + col) * a.C + c * 8);
        }
    };
    auto stage = [&](int t, int sb) {                          // registers -> staging buffer sb + the tile's tap tables (no barrier here)
        int b, ty0, tx0, rs0, cs0;
        tile_org(t, b, ty0, tx0);
        src_org(ty0, tx0, rs0, cs0);
#pragma unroll
        for (int k = 0; k < NSLP; ++k) {
            const int idx = ptid + k * 256;
            if (idx < SCH) stg0[sb * STG + (idx >> 4) * SPS + (idx & 15)] = hr[k];
        }
        if (ptid < TH + 2 + HWD) {
            const bool rowtab = ptid < TH + 2;
            const int j = rowtab ? ptid : ptid - (TH + 2);
            const int i = (rowtab ? ty0 : tx0) + j - 1, lim = rowtab ? a.Hi : a.Wi;
            int2 v = make_int2(-1, -1);
            float2 w = make_float2(0.f, 0.f);
            if (i >= 0 && i < lim) {
                const Tap tp = linear_tap(i, rowtab ? a.usy : a.usx, rowtab ? a.Hs : a.Ws, true);
                const int mul = rowtab ? SRX * SPS : SPS, org = rowtab ? rs0 : cs0;
                v = make_int2((tp.i0 - org) * mul, (tp.i1 - org) * mul);
                w = make_float2(tp.w0, tp.w1);
            }
            tap_i[sb][rowtab ? 0 : 1][j] = v; tap_w[sb][rowtab ? 0 : 1][j] = w;
        }
    };
    // this producer thread's halo chunks are the same in every tile: chunk k = ptid + 256 k -> (halo pixel p, channel chunk c); packed
    // once: LDS destination (12 bits) | halo row (4) | halo column (5) | c (4), or -1 past the halo
    int pc[NLDP];
#pragma unroll
    for (int k = 0; k < NLDP; ++k) {
        const int idx = ptid + k * 256;
        const int p = idx >> 4, c = idx & 15, hy = p / HWD, hx = p - hy * HWD;
        pc[k] = (consumer || idx >= NCH) ? -1 : ((p * PST + c3_chunk_slot<PST>(c)) | (hy << 12) | (hx << 16) | (c << 21));
    }
    // halo chunks k0 <= k < k1 of the tile staged in sb -> halo buffer hbuf; two at a time, their eight tap chunks requested before the
    // first is used.  (Per chunk: two table reads, four adds, four tap reads, lerp_chunk's arithmetic, one store -- with the index
    // arithmetic and linear_tap per chunk the four producer waves needed 4.3 us per tile against 1.9 us of MFMA work.)
    auto lerp_chunks = [&](int sb, int hbuf, auto k0c, auto k1c) {
        const u32x4* stg = stg0 + sb * STG;
#if defined(C128_CUT) && (C128_CUT == 1 || C128_CUT == 3)     // (tuning aid, timing only: no interpolation)
        if (a.Hs > 0) return;
#endif
        constexpr int k0 = decltype(k0c)::value, NB = 6;       // chunks per batch: all their table reads, then all 24 tap reads, are in flight together
        static_for<(decltype(k1c)::value - k0) / NB>([&](auto jc) {
            constexpr int k = k0 + NB * decltype(jc)::value;
            int2 vy[NB], vx[NB];
            float2 wx[NB], wy[NB];
            u32x4 v[NB][4];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int pk = pc[k + u] < 0 ? 0 : pc[k + u];  // (chunk 0's entries for the threads past the halo: loads are unconditional)
                const int hy = (pk >> 12) & 15, hx = (pk >> 16) & 31;
                vy[u] = tap_i[sb][0][hy]; vx[u] = tap_i[sb][1][hx];
                wy[u] = tap_w[sb][0][hy]; wx[u] = tap_w[sb][1][hx];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int c = (pc[k + u] < 0 ? 0 : pc[k + u]) >> 21;
                const bool ok = (vy[u].x | vx[u].x) >= 0;
                const int y0 = ok ? vy[u].x : 0, y1 = ok ? vy[u].y : 0, x0 = ok ? vx[u].x : 0, x1 = ok ? vx[u].y : 0;
                v[u][0] = stg[y0 + x0 + c]; v[u][1] = stg[y0 + x1 + c]; v[u][2] = stg[y1 + x0 + c]; v[u][3] = stg[y1 + x1 + c];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int pk = pc[k + u];
                if (pk >= 0) {
                    u32x4 r = {0u, 0u, 0u, 0u};
                    if ((vy[u].x | vx[u].x) >= 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) r[q] = lerp_pair_bf16(v[u][0][q], v[u][1][q], v[u][2][q], v[u][3][q], wx[u].x, wx[u].y, wy[u].x, wy[u].y);   // = lerp_chunk
                    }
                    s16x8_ x = __builtin_bit_cast(s16x8_, r);
                    x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
                    lds[hbuf * HALO + (pk & 4095)] = __builtin_bit_cast(u32x4, x);
                }
            }
        });
    };

    // tile walk: XCD x owns a contiguous run of tiles, its CUs take consecutive tiles of it (conv3_head_kernel)
    const bool xcd_walk = (gridDim.x & 7) == 0;
    const int xcd_ = blockIdx.x & 7, slot_ = blockIdx.x >> 3, nslot_ = gridDim.x >> 3, per_ = (ntiles + 7) >> 3;
    auto tile_at = [&](int k) {
        if (!xcd_walk) { const int tt = blockIdx.x + k * gridDim.x; return tt < ntiles ? tt : -1; }
        const int j = slot_ + k * nslot_, tt = xcd_ * per_ + j;
        return (j < per_ && tt < ntiles) ? tt : -1;
    };
    int kk = 0;
    int t = tile_at(0);
    if (t < 0) return;
    int t1 = tile_at(1);
    constexpr std::integral_constant<int, 0> K0{};
    constexpr std::integral_constant<int, NLDP / 2> KH{};
    constexpr std::integral_constant<int, NLDP> KE{};
    static_assert(NLDP == 12, "two halves of six chunks");
    // ---- prologue (producers; the consumers load their weights meanwhile): the first tile's halo, the second tile's source window staged in buffer 0
    if (!consumer) { load_src(t); stage(t, 1); }
    __syncthreads();
    if (!consumer) {
        lerp_chunks(1, 0, K0, KE);
        if (t1 >= 0) { load_src(t1); stage(t1, 0); }
    }

    // From here on the two roles run their OWN loops (one loop with a role branch keeps the consumers' 144 weight registers alive
    // through the producers' code: 119 spilled registers); both pass the same two block barriers per tile, A and B:
    //   A: halo[buf] = tile t, staging sb = the source of tile t1, patch = the rows of the previous tile -- all complete
    //   B: the patch is read out (consumers), the first half of tile t1's halo is in place (producers)
    int buf = 0, sb = 0;
    if (!consumer) {
        for (; t >= 0;) {
            const int t2 = tile_at(kk + 2);
            ++kk;
            __syncthreads();                                   // A
            if (t2 >= 0) load_src(t2);                         // the source window of tile t2 is requested,
            if (t1 >= 0) lerp_chunks(sb, buf ^ 1, K0, KH);     // the first half of tile t1's halo interpolated
            __syncthreads();                                   // B
            if (t1 >= 0) lerp_chunks(sb, buf ^ 1, KH, KE);     // the second half;
            if (t2 >= 0) stage(t2, sb ^ 1);                    // tile t2's source window into the other staging buffer
            buf ^= 1; sb ^= 1;
            t = t1; t1 = t2;
        }
        __syncthreads();
        return;
    }
    // ---- consumers: W fragments (MFMA row 16 nq + fr; K step (tap, g) -> chunk 4 g + fg of the tap's 128 channels), bias
    // (s_setprio 2 for these waves -- the MFMA stream first when both waves of a SIMD can issue: 308 vs 306 us, no effect)
    u32x4 wf[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nq * 16 + fr;
            wf[tap][g] = n < N ? *(const u32x4*)(W + (long)n * Kpad + tap * 128 + (g * 4 + fg) * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
    EpiCols cols;
    const int n0 = nq * 16 + fg * 4;
    if (n0 < N) epi_cols_load(e, n0, cols);
    // A fragment of halo row ir, tap column kx, K group g: pixel (ir, px + kx), chunk 4 g + fg = hb + constant
    const int hb0 = px * PST + c3_chunk_slot<PST>(fg);
    auto store_rows = [&](int tp) {                            // the patch -> global memory, whole 128-byte lines
        int b, ty0, tx0;
        tile_org(tp, b, ty0, tx0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 256, pp = idx >> 3, c = idx & 7;
            const int y = ty0 + (pp >> 4), x = tx0 + (pp & 15);
            if (y < a.Ho && x < a.Wo)
                *(u32x4*)((bf16_t*)e.out + ((long)(b * a.Ho + y) * a.Wo + x) * e.ldc + c * 8) = patch[pp * 8 + (c ^ (pp & 7))];
        }
    };
    int tprev = -1;                                            // the tile whose output rows sit in the patch
    constexpr int DEPTH = 6, NFR = 10 * 12;                    // fragments in flight; fragments per tile (halo row, kx, g)
    for (; t >= 0;) {
        const int t2 = tile_at(kk + 2);
        ++kk;
        __syncthreads();                                       // A
        const u32x4* hp = lds + buf * HALO + hb0;
        f32x4 acc[8];
        u32x4 fq[DEPTH];
        // Fragment reads as inline asm with hand-counted lgkmcnt: written as plain loads "DEPTH ahead", the compiler sank every ds_read_b128
        // to just behind the previous use of its registers and waited lgkmcnt(1) before each group of MFMAs -- the K loop ran at the LDS
        // round trip, 31 cycles per MFMA.  LDS returns in order: before fragment idx is used, at most DEPTH - 1 younger reads may be out
        // (fewer at the end of the tile).  The wait carries the fragment as an operand so that its MFMAs cannot move above it.
        const unsigned hpa = (unsigned)(unsigned long)((__attribute__((address_space(3))) const char*)hp);
        auto frag_issue = [&](auto ic) {
            constexpr int idx = decltype(ic)::value, ir = idx / 12, kx = (idx / 4) % 3, g = idx % 4;
            c3_lds_read_b128<((ir * HWD + kx) * PST + g * 4) * 16>(fq[idx % DEPTH], hpa);
        };
        auto kloop = [&](auto lo, auto hi) {                   // fragments [lo, hi)
            static_for<decltype(hi)::value - decltype(lo)::value>([&](auto jc) {
                constexpr int idx = decltype(lo)::value + decltype(jc)::value, ir = idx / 12, kx = (idx / 4) % 3, g = idx % 4;
                constexpr int out = NFR - 1 - idx < DEPTH - 1 ? NFR - 1 - idx : DEPTH - 1;     // younger reads in flight
                c3_lgkm_wait<out>(fq[idx % DEPTH]);
                const u32x4 fa = fq[idx % DEPTH];
                // output row i = ir - ky: ky ascends with ir for a fixed i, so each accumulator sees (ky, kx, g) in conv3_halo2's order
                static_for<3>([&](auto kyc) {
                    constexpr int ky = decltype(kyc)::value, i = ir - ky;
                    if constexpr (i >= 0 && i < 8) mma_chunk(acc[i], wf[ky * 3 + kx][g], fa, bf16_t());
                });
                if constexpr (idx + DEPTH < NFR) {
                    asm volatile("" : "+v"(acc[ir < 8 ? ir : 7]));                              // (the refill stays BEHIND this fragment's MFMAs)
                    frag_issue(std::integral_constant<int, idx + DEPTH < NFR ? idx + DEPTH : 0>{});
                }
            });
        };
        // phase 1: the previous tile's rows leave the patch; halo rows 0-4
        static_for<DEPTH>([&](auto ic) { frag_issue(ic); });
#if !(defined(C128_CUT) && C128_CUT == 3)    // (CUT 3, timing only: no interpolation, no output path -- the K loop and the barriers)
        if (tprev >= 0) store_rows(tprev);
#endif
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        kloop(std::integral_constant<int, 0>{}, std::integral_constant<int, NFR / 2>{});
        __syncthreads();                                       // B
        // phase 2: halo rows 5-9, then this tile's rows into the patch (bias add and rounding as epilogue4's: the same bits)
        kloop(std::integral_constant<int, NFR / 2>{}, std::integral_constant<int, NFR>{});
        uint2* pw = (uint2*)patch;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pp = i * 16 + px;
            float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
            if (e.bias) { v[0] += cols.bias[0]; v[1] += cols.bias[1]; v[2] += cols.bias[2]; v[3] += cols.bias[3]; }
            uint2 w2;
            w2.x = pk_bf16(v[0], v[1]); w2.y = pk_bf16(v[2], v[3]);
            pw[(pp * 8 + ((nq * 2 + (fg >> 1)) ^ (pp & 7))) * 2 + (fg & 1)] = w2;
        }
        tprev = t;
        buf ^= 1;
        t = t1; t1 = t2;
    }
    __syncthreads();                                           // the last tile's rows
    if (tprev >= 0) store_rows(tprev);
}

// ================================================================================================
// conv3_head_ups_kernel: the head's conv2 with the up-sample in front of it folded in (what conv3_head_kernel<1> does), re-cut into
// PRODUCER and CONSUMER waves.
// What conv3_head_kernel<1> spends (profiles/r3_06, batch 32: 349 us, MFMA busy 0.23, 36 % of the LDS cycles in bank conflicts): its
// eight waves run in lock-step -- 72 MFMAs each, then ALL of them interpolate the next tile's 18 x 18 halo (four taps x two
// ds_read_b128 of unpacked floats + 24 packed FMAs per 8-channel chunk: ~2 900 VALU cycles and ~5 000 LDS cycles per SIMD per tile
// against 2 304 MFMA cycles), then a barrier: the matrix pipe idles through the interpolation and the vector pipe through the MFMAs.
// Here
//   * waves 0-3 (one per SIMD) are consumers: 4 output rows x 16 pixels x 32 channels each, W in registers as before; an input row's
//     fragment (row r of the wave's 6, kx, ks) is read ONCE and feeds the output rows r - 2 .. r that tap it (36 fragment reads per
//     144 MFMAs instead of 72 per 144); per accumulator the (tap, ks) order is conv3_head_kernel's, so results are bit-identical;
//   * waves 4-7 are producers: they build the NEXT tile's halo in the other buffer while the consumers compute, and the bilinear
//     sample is evaluated SEPARABLY with the stand-alone kernel's own expression (bilerp1: top / bot = the horizontal lerps, then the
//     vertical one; the horizontal results stay fp32, so every value rounds exactly as before):
//       H pass  (13 source rows x 18 halo columns x 64 channels): two 8-byte taps straight from global memory (requested one phase
//               ahead, L2-resident source map), one packed mul + fma per channel pair, fp32 rows into LDS  [r][x][64];
//       V pass  (18 x 18 x 64): two ds_read_b128 of H rows, one packed mul + fma per pair, bf16, into the halo buffer.
//     35 700 lerps per tile instead of 62 200, 2 + 2 LDS reads per 4 channels instead of 8 per 8, and every LDS access is a
//     16-lane group on one aligned 256-byte line (conflict-free by construction);
//   * two block barriers per tile: [consumers: input rows 0-2 (72 MFMAs) | producers: H pass]  [consumers: rows 3-5 + epilogue |
//     producers: V pass + the H-pass loads of the tile after].
// LDS: 2 x 51 840 B of halo + 59 904 B of H rows = 163 584 B of the 163 840.
// ================================================================================================
// tuning aid (-DD2S_C3U_TIMING): lane 0 of the first consumer wave / the first producer wave accumulates 100 MHz wall-clock time per
// phase: [0] consumer phase A, [1] its wait at barrier 1, [2] phase B, [3] wait at barrier 2; [4..7] the same for the producer
#ifdef D2S_C3U_TIMING
__device__ unsigned long long c3u_timing[256 * 9];
#define C3U_T(var) const long var = wall_clock64();
#define C3U_ACC(SLOT, expr) { if ((threadIdx.x & 255) == 0 && blockIdx.x < 256) c3u_timing[blockIdx.x * 9 + (SLOT)] += (unsigned long long)(expr); }
#else
#define C3U_T(var) {}
#define C3U_ACC(SLOT, expr) {}
#endif
__global__ void __launch_bounds__(512)
conv3_head_ups_kernel(GemmA a, const bf16_t* __restrict__ W, int N, int Kpad, GemmEpi e, int ntiles) {
    KERNARG_WARM(kaw_)
    KERNARG_WARM_END(kaw_)
    constexpr int CPP = 8, PST = 10, TH = 16, TW = 16, HWD = TW + 2, HPX = (TH + 2) * HWD, HALO = HPX * PST;
    constexpr int SR = 13;                                   // source rows under 18 halo rows, scales <= 0.6
    constexpr int NPAIR = SR * HWD;                          // 234 (row, column) cells of the H buffer, row-major: cell = r * HWD + x
    constexpr int HPL = NPAIR * CPP;                         // one PLANE of the H buffer in f32x4: [cell][chunk], plane h = channels 4 h .. 4 h + 3 of every chunk
    constexpr int HU = (NPAIR + 31) / 32;                    // H-pass cells per producer thread (8): 32 pixel slots x 8 chunk lanes
    constexpr int VU = (HPX + 31) / 32;                      // V-pass halo pixels per producer thread (11)
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * HALO + 2 * HPL];
    D2S_POISON_LDS(lds, 2 * HALO + 2 * HPL)
    f32x4* const hbuf = (f32x4*)(lds + 2 * HALO);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wid < 4;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    typedef float f2_ __attribute__((ext_vector_type(2)));

    auto tile_org = [&](int t, int& b, int& ty0, int& tx0) {
        b = t / (tiles_y * tiles_x);
        const int r = t - b * (tiles_y * tiles_x);
        ty0 = (r / tiles_x) * TH; tx0 = (r % tiles_x) * TW;
    };
    const bool xcd_walk = (gridDim.x & 7) == 0;
    const int xcd_ = blockIdx.x & 7, slot_ = blockIdx.x >> 3, nslot_ = gridDim.x >> 3, per_ = (ntiles + 7) >> 3;
    auto tile_at = [&](int k) {
        if (!xcd_walk) { const int tt = blockIdx.x + k * gridDim.x; return tt < ntiles ? tt : -1; }
        const int j = slot_ + k * nslot_, tt = xcd_ * per_ + j;
        return (j < per_ && tt < ntiles) ? tt : -1;
    };

    // ---------------- producer side.  Thread = (chunk pc of 8 channels, pixel slot ps of 32): a wave covers 8 consecutive cells /
    // halo pixels x 8 chunks, so that every ds_read_b128 / ds_write_b128 service group of 16 lanes meets 16 distinct bank quads (four
    // 128-byte runs whose cells differ by 1 and 3: MI355X_MICROARCH.md, LDS table).
    const int ptid = tid & 255, pc = ptid & 7, ps = ptid >> 3;
    // source map through a buffer descriptor: taps outside the image are requested past num_records and come back as zeros
    const unsigned src_frame = (unsigned)a.Hs * a.Ws * a.C * 2u;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.ptr), 0, (unsigned)(ntiles / (tiles_y * tiles_x)) * src_frame, 0x00020000);
    u32x4 hv0[HU], hv1[HU];                                  // the H-pass taps of the tile in production, requested one phase ahead
    float hw1[HU];                                           // ... and their horizontal weights
    // Tile-invariant coordinates of this thread's work items (the per-item arithmetic that is left inside the tile loop is what the
    // producers are bound by: one wave per SIMD, 4 cycles per instruction).  H pass: cell ps + 32 k = (row hr, column hxx) -> LDS index
    // (ps * 8 + pc) + 256 k.  V pass: halo pixel ps + 32 k = (row vy, column vx) -> halo index (ps * 10 + pc) + 320 k.
    int hr[HU], hx4[HU], vy4[VU], vcell[VU];
#pragma unroll
    for (int k = 0; k < HU; ++k) { const int cell = ps + 32 * k < NPAIR ? ps + 32 * k : NPAIR - 1; hr[k] = cell / HWD; hx4[k] = (cell - hr[k] * HWD) * 4; }
#pragma unroll
    for (int k = 0; k < VU; ++k) { const int p = ps + 32 * k < HPX ? ps + 32 * k : HPX - 1; const int hy = p / HWD; vy4[k] = hy * 4; vcell[k] = (p - hy * HWD) * CPP + pc; }
    // The bilinear taps of a tile's 18 halo columns / rows are computed ONCE per wave -- lane l holds column / row l -- and every work
    // item fetches its own through ds_bpermute_b32 (the LDS crossbar, no LDS memory): packed word = offset | step << 16 | valid << 31,
    // and the weight w1.
    const int tl = lane < HWD ? lane : HWD - 1;
    // geometry of a tile, worked out once (tile_org's integer divisions are ~40 scalar instructions, and a lone wave per SIMD issues
    // one instruction of ANY kind per 4 cycles: the producers are bound by their instruction count, scalar ones included)
    struct TileGeo { int b, ty0, tx0, rs0; };
    auto tile_geo = [&](int t) {
        TileGeo g;
        tile_org(t, g.b, g.ty0, g.tx0);
        g.rs0 = __builtin_amdgcn_readfirstlane(linear_tap(g.ty0 > 0 ? g.ty0 - 1 : 0, a.usy, a.Hs, true).i0);
        return g;
    };
    // H pass, part 1: request the two horizontal taps of cells ps + 32 k of tile t (source row rs0 + hr, halo column hxx)
    auto h_request = [&](const TileGeo& g) {
        const int ty0 = g.ty0, tx0 = g.tx0, rs0 = g.rs0; (void)ty0;
        const unsigned fbase = (unsigned)g.b * src_frame + (unsigned)pc * 16u;
        const int ix = tx0 + tl - 1;
        const Tap tx = linear_tap(ix < 0 ? 0 : ix, a.usx, a.Ws, true);
        const int pw = (tx.i0 * a.C * 2) | ((tx.i1 - tx.i0) * a.C * 2) << 16 | ((ix >= 0 && ix < a.Wi) ? 0 : (int)0x80000000u);    // bit 31: column outside the image
        const int pf = __float_as_int(tx.w1);
        const int rowb = a.Ws * a.C * 2;
#pragma unroll
        for (int k = 0; k < HU; ++k) {
            const int w = __builtin_amdgcn_ds_bpermute(hx4[k], pw);
            hw1[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(hx4[k], pf));
            const int row = rs0 + hr[k] < a.Hs ? rs0 + hr[k] : a.Hs - 1;
            const unsigned o0 = fbase + (unsigned)(row * rowb) + (unsigned)(w & 0xffff);
            hv0[k] = __builtin_amdgcn_raw_buffer_load_b128(rsS, w < 0 ? 0xfffffff0u : o0, 0, 0);
            hv1[k] = __builtin_amdgcn_raw_buffer_load_b128(rsS, w < 0 ? 0xfffffff0u : o0 + (unsigned)((w >> 16) & 0x7fff), 0, 0);
        }
    };
    // H pass, part 2: top = fma(w1x, v01, w0x * v00) per channel (bilerp1's horizontal lerp), fp32, two planes of hbuf; columns outside
    // the image hold zeros (both taps came back as zeros)
    auto h_compute = [&]() {
#pragma unroll
        for (int k = 0; k < HU; ++k) {
            const float w0 = 1.0f - hw1[k];
            const f2_ w0x = {w0, w0}, w1x = {hw1[k], hw1[k]};
            f32x4 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned p0 = hv0[k][2 * h], p1 = hv0[k][2 * h + 1], q0 = hv1[k][2 * h], q1 = hv1[k][2 * h + 1];
                const f2_ a0 = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)}, b0 = {__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
                const f2_ a1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)}, b1 = {__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
#if !defined(C3U_PACKED_H)
                // (beside the consumers' MFMAs a packed f32 instruction costs more than the two plain ones it replaces: MI355X_MICROARCH.md)
                float t_[4];
                asm volatile("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %4, %6\n\tv_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %4, %8"
                             : "=&v"(t_[0]), "=&v"(t_[1]), "=&v"(t_[2]), "=&v"(t_[3]) : "v"(w0), "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]));
                asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %6\n\tv_fmac_f32 %2, %4, %7\n\tv_fmac_f32 %3, %4, %8"
                             : "+v"(t_[0]), "+v"(t_[1]), "+v"(t_[2]), "+v"(t_[3]) : "v"(hw1[k]), "v"(b0[0]), "v"(b0[1]), "v"(b1[0]), "v"(b1[1]));
                o[h] = (f32x4){t_[0], t_[1], t_[2], t_[3]};
                (void)w0x; (void)w1x;
#else
                const f2_ t0 = __builtin_elementwise_fma(w1x, b0, w0x * a0);
                const f2_ t1 = __builtin_elementwise_fma(w1x, b1, w0x * a1);
                o[h] = (f32x4){t0[0], t0[1], t1[0], t1[1]};
#endif
            }
            if (k < HU - 1 || ps + 32 * k < NPAIR) { hbuf[ps * CPP + pc + 32 * CPP * k] = o[0]; hbuf[HPL + ps * CPP + pc + 32 * CPP * k] = o[1]; }
        }
    };
    // V pass: halo pixels ps + 32 k: o = fma(w1y, bot, w0y * top) (bilerp1's vertical lerp), bf16, into halo buffer `buf`; rows outside the
    // image: zeros
    // Items KA .. KB - 1 of the calling thread: the producers take the first VS items of every pixel slot, the CONSUMER waves the rest
    // after their epilogue -- they are the older waves, win the VALU arbitration against the producers on their SIMD and would
    // otherwise wait ~40 % of a tile at the barriers (measured with -DD2S_C3U_TIMING: 1.5 of 3.5 us)
    auto v_pass = [&](const TileGeo& g, int buf, auto kac, auto kbc) {
        constexpr int KA = decltype(kac)::value, KB = decltype(kbc)::value;
        const int ty0 = g.ty0, rs0 = g.rs0;
        const int iy = ty0 + tl - 1;
        const bool rok = iy >= 0 && iy < a.Hi;
        const Tap ty = linear_tap(rok ? iy : 0, a.usy, a.Hs, true);
        const int pw = ((ty.i0 - rs0) * (HWD * CPP)) | ((ty.i1 - ty.i0) * (HWD * CPP)) << 16 | (rok ? 0 : (int)0x80000000u);          // bit 31: row outside the image
        const int pf = __float_as_int(ty.w1);
        u32x4* const hl = lds + buf * HALO + ps * PST + pc;
        // in batches of VB pixels: all H-row reads of a batch are requested before the first is used (the halo stores of one pixel and
        // the reads of the next are accesses to the same array -- left in program order they serialise on the LDS latency)
        constexpr int VB = 4;
        static_for<(KB - KA + VB - 1) / VB>([&](auto bc) {
            constexpr int k0 = KA + decltype(bc)::value * VB, nb = KB - k0 < VB ? KB - k0 : VB;
            f32x4 top[nb][2], bot[nb][2];
            float w1[nb];
            int w[nb];
#pragma unroll
            for (int u = 0; u < nb; ++u) {
                w[u] = __builtin_amdgcn_ds_bpermute(vy4[k0 + u], pw);
                w1[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(vy4[k0 + u], pf));
            }
#pragma unroll
            for (int u = 0; u < nb; ++u) {
                const int c0 = (w[u] & 0xffff) + vcell[k0 + u], c1 = c0 + ((w[u] >> 16) & 0x7fff);
#pragma unroll
                for (int h = 0; h < 2; ++h) { top[u][h] = hbuf[h * HPL + c0]; bot[u][h] = hbuf[h * HPL + c1]; }
            }
#pragma unroll
            for (int u = 0; u < nb; ++u) {
                const float w0 = 1.0f - w1[u];
                const f2_ w0y = {w0, w0}, w1y = {w1[u], w1[u]};
                u32x4 o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f2_ o0 = __builtin_elementwise_fma(w1y, (f2_){bot[u][h][0], bot[u][h][1]}, w0y * (f2_){top[u][h][0], top[u][h][1]});
                    const f2_ o1 = __builtin_elementwise_fma(w1y, (f2_){bot[u][h][2], bot[u][h][3]}, w0y * (f2_){top[u][h][2], top[u][h][3]});
                    o[2 * h] = pk_bf16(o0[0], o0[1]); o[2 * h + 1] = pk_bf16(o1[0], o1[1]);
                }
                if (w[u] < 0) o = (u32x4){0u, 0u, 0u, 0u};
                if (k0 + u < VU - 1 || ps + 32 * (k0 + u) < HPX) hl[32 * PST * (k0 + u)] = o;
            }
        });
    };
#ifndef C3U_VS
#define C3U_VS 6
#endif
    constexpr int VS = C3U_VS;                               // V-pass items per pixel slot that stay with the producers
    typedef std::integral_constant<int, 0> K0_;
    typedef std::integral_constant<int, VS> KS_;
    typedef std::integral_constant<int, VU> KU_;

    int kk = 0;
    int t = tile_at(0);
    if (t < 0) return;
    // The two roles run SEPARATE loops over the same tile sequence with the same number of barriers (the hardware barrier counts
    // waves, not program counters): in one shared loop the register allocator keeps the consumers' 144 W registers live through
    // the producers' code (522 spilled registers, measured).
    if (consumer) {
        // W fragments + epilogue constants (as conv3_head_kernel)
        u32x4 wf[9][2][2];
        float cb[2][4], cw[2][4];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = j * 16 + fr;
                    wf[tap][ks][j] = n < N ? *(const u32x4*)(W + (long)n * Kpad + tap * 64 + (ks * 4 + fg) * 8) : (u32x4){0u, 0u, 0u, 0u};
                }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n0 = j * 16 + fg * 4;
            if (n0 < N) { load4(e.bias + n0, cb[j]); load4(e.scale + n0, cw[j]); }
            else { cb[j][0] = cb[j][1] = cb[j][2] = cb[j][3] = 0.f; cw[j][0] = cw[j][1] = cw[j][2] = cw[j][3] = 0.f; }
        }
        __syncthreads();                                       // (prologue: H rows of the first tile)
        int tn = tile_at(++kk);
        __syncthreads();                                       // (prologue: its halo)
        const int cw4 = wid * 4;                               // first output row of this wave
        const int hb0 = (cw4 * HWD + fr) * PST + fg;
        int buf = 0;
        for (; t >= 0;) {
            f32x4 acc[4][2];
            const u32x4* hp = lds + buf * HALO + hb0;
            // input rows R0 .. R0 + 2 of the wave's six: fragment (r, kx, ks) feeds output rows i = r - ky, ky = 0 .. 2
            // (round 5: these reads pinned four ahead of their MFMAs as in conv3_c128_ups_kernel -- 282 vs 283 us: here the producers
            //  set the pace; not kept)
            auto mma_rows = [&](auto r0c) {
                constexpr int R0 = decltype(r0c)::value;
                static_for<3>([&](auto rc) {
                    constexpr int r = R0 + decltype(rc)::value;
                    static_for<3>([&](auto kxc) {
                        constexpr int kx = decltype(kxc)::value;
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const u32x4 fa = hp[(r * HWD + kx) * PST + ks * 4];
                            static_for<3>([&](auto kyc) {
                                constexpr int ky = 2 - decltype(kyc)::value;
                                constexpr int i = r - ky;
                                if constexpr (i >= 0 && i < 4) {
#pragma unroll
                                    for (int j = 0; j < 2; ++j) mma_chunk(acc[i][j], wf[ky * 3 + kx][ks][j], fa, bf16_t());
                                }
                            });
                        }
                    });
                });
            };
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            C3U_T(u0)
            mma_rows(std::integral_constant<int, 0>());
            C3U_T(u1)
            __syncthreads();
            C3U_T(u2)
            const int tnn = tn >= 0 ? tile_at(++kk) : -1;
            mma_rows(std::integral_constant<int, 3>());
            int b, ty0, tx0;
            tile_org(t, b, ty0, tx0);
            // depth = act(b3 + sum_n w3[n] relu(acc + bias[n])): the four lane groups of a pixel hold 8 channels each.  Summed as
            // (g0 + g1) + (g2 + g3) like conv3_head_kernel's two xor-shuffles, but through v_permlane16_swap / v_permlane32_swap on two
            // rows at a time: no LDS round trips, and lane group fg ends up with output row fg -- one full-wave store per tile
            float sr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    s += fmaxf(acc[i][j][0] + cb[j][0], 0.f) * cw[j][0] + fmaxf(acc[i][j][1] + cb[j][1], 0.f) * cw[j][1] +
                         fmaxf(acc[i][j][2] + cb[j][2], 0.f) * cw[j][2] + fmaxf(acc[i][j][3] + cb[j][3], 0.f) * cw[j][3];
                sr[i] = s;
            }
            typedef unsigned u2_ __attribute__((ext_vector_type(2)));
            const u2_ p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sr[0]), __float_as_uint(sr[1]), false, false);
            const u2_ p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sr[2]), __float_as_uint(sr[3]), false, false);
            const float c01 = __uint_as_float(p01[0]) + __uint_as_float(p01[1]);       // rows: [r0: g0 + g1 | r1: g0 + g1 | r0: g2 + g3 | r1: g2 + g3]
            const float c23 = __uint_as_float(p23[0]) + __uint_as_float(p23[1]);
            const u2_ q = __builtin_amdgcn_permlane32_swap(__float_as_uint(c01), __float_as_uint(c23), false, false);
            const float sv = __uint_as_float(q[0]) + __uint_as_float(q[1]);             // lane group fg: output row fg, pixel fr
            {
                const int y = ty0 + cw4 + fg, x = tx0 + fr;
                if (y < a.Ho && x < a.Wo) ((float*)e.out)[((long)b * a.Ho + y) * a.Wo + x] = head_activation(sv + e.head_b3, e.head_max_depth);
            }
            if (VS < VU && tn >= 0) v_pass(tile_geo(tn), buf ^ 1, KS_(), KU_());     // this wave's share of the next tile's V pass
            C3U_T(u3)
            __syncthreads();                                   // tile t is read out, tile tn's halo is in place
            C3U_T(u4)
            C3U_ACC(0, u1 - u0) C3U_ACC(1, u2 - u1) C3U_ACC(2, u3 - u2) C3U_ACC(3, u4 - u3) C3U_ACC(8, 1)
            buf ^= 1;
            t = tn; tn = tnn;
        }
    } else {
        TileGeo gn = tile_geo(t), gnn = gn;                    // geometry of tile tn (in production) and of the one after
        h_request(gn);
        h_compute();
        __syncthreads();
        int tn = tile_at(++kk);
        v_pass(gn, 0, K0_(), KU_());
        if (tn >= 0) { gn = tile_geo(tn); h_request(gn); }
        __syncthreads();
        int buf = 0;
        for (; t >= 0;) {
            C3U_T(u0)
            if (tn >= 0) h_compute();
            C3U_T(u1)
            __syncthreads();                                   // H rows of tile tn are in place
            C3U_T(u2)
            const int tnn = tn >= 0 ? tile_at(++kk) : -1;
            if (tn >= 0) {
                if (tnn >= 0) { gnn = tile_geo(tnn); h_request(gnn); }     // in flight under the V pass and the barrier
                v_pass(gn, buf ^ 1, K0_(), KS_());
                gn = gnn;
            }
            C3U_T(u3)
            __syncthreads();
            C3U_T(u4)
            C3U_ACC(4, u1 - u0) C3U_ACC(5, u2 - u1) C3U_ACC(6, u3 - u2) C3U_ACC(7, u4 - u3)
            buf ^= 1;
            t = tn; tn = tnn;
        }
    }
}

// ================================================================================================
// C = 128 -> N = 128 on the large maps in the batched regime (the fusion stages' residual units: 8 of the 21 convolutions at batch
// 32, 60 % of their time): persistent 8-wave blocks, 256-pixel tiles, the W ring running on ACROSS tiles, two wave groups in
// ping-pong.  conv3_wide_kernel<TH, TW>.
// The one-shot blocks above are DMA-latency bound: with the 49 KB halo two blocks per CU leave room for TWO 16 KB weight stages
// each, so the refill of a stage has one K tile of cover (~500 MFMA cycles per SIMD) against ~2 000 cycles of LDS-DMA latency
// under load -- waves wait 60 % of their cycles (profiles/r3_02).  Here
//   * ONE block per CU: tile 8 x 32 or 16 x 16 pixels (whichever pads the map less), halo 10 x 34 | 18 x 18 pixels x 272 B
//     = 92 | 88 KB, and an EIGHT-stage W ring of 8 KB stages (128 rows x 32 channels of one tap = one MFMA K step; 36 K tiles per
//     tile): a stage is requested seven K tiles before its MFMAs and waited for five K tiles after the request;
//   * wave tile 64 pixels x 64 channels (4 x 4 fragments, 64 accumulators): 8 fragment reads per 16 MFMAs instead of 6 per 8;
//   * the ring never drains: the K tiles of W are the same for every tile, so K tiles 29..34 of a tile request the first six of the
//     next (36 = 4 mod 8: the stage of K tile 0 alternates between 0 and 4 -- one scalar per tile);
//   * the NEXT tile's halo is fetched into registers (11 x 16 B per thread) at K tile 28 and stored over the halo buffer after the
//     last K tile;
//   * the epilogue requests every residual value of the wave before it touches the first (the output may alias the residual, so the
//     compiler keeps each load behind the previous store: sixteen dependent round trips, 17.9 us per tile, measured -> 3.4 us).
// Schedule inside a tile: see the K loop.  Measured per tile (tools/c3_timeline.py, batch 32): K loop 12.4 us (7.7 us of MFMA at
// 2.4 GHz), epilogue + halo store 4.5 us; the eight launches of a batch-32 step 1.29 ms -> 0.62 ms of block lifetime, the step's
// convolutions 2.01 -> 1.85 ms (0.25 -> 0.275 of the dense bf16 peak), +2.3 % frames/s.
// vmcnt bookkeeping (in-order retire): in its M slot of K tile kt a wave waits for its part of W(kt + 2); younger than it are
// W(kt + 3 .. kt + 7) = 5 requests, and -- for kt = 28..33, the last wait of the loop -- the halo loads issued at K tile 28 (+ NLD):
// no wait inside the K loop depends on HBM answering them; at the start of a tile everything is drained once (W(0..5), the halo
// registers were consumed, the previous tile's stores).
// ================================================================================================
// tuning aid (D2S_HIPCC_DEFS=-DD2S_C3_TIMING): thread 0 of every block accumulates the 100 MHz wall-clock time it spends per phase
// over all its tiles: [0] K loop, [1] epilogue, [2] barrier + halo store, [3] tiles, [4] K tile 0 (incl. the drain), [5] lifetime
#ifdef D2S_C3_TIMING
__device__ unsigned long long c3_timing[256 * 8];
#define C3_T(var) const long var = wall_clock64();
#define C3_ACC(SLOT, expr) { if (threadIdx.x == 0 && blockIdx.x < 256) c3_timing[blockIdx.x * 8 + (SLOT)] += (unsigned long long)(expr); }
#else
#define C3_T(var) {}
#define C3_ACC(SLOT, expr) {}
#endif

template <int TH, int TW>
__global__ void __launch_bounds__(512)
conv3_wide_kernel(GemmA a, const bf16_t* __restrict__ W, int N, int Kpad, GemmEpi e, int ntiles) {
    KERNARG_WARM(kaw_)                                   // all argument lines in one round trip (common.h)
    KERNARG_WARM_END(kaw_)
    constexpr int CPP = 16, PST = 17, HWD = TW + 2, HPX = (TH + 2) * HWD, HALO = HPX * PST;
    constexpr int NCH = HPX * CPP, NLD = (NCH + 511) / 512;
    constexpr int NW = 8, WN = 2, FM = 4, FN = 4, FPR = TW / 16;
    constexpr int NS = 8, WST = 128 * 4, NKT = 36, PD = NS - 1;  // W ring: 8 stages of 128 rows x 64 B (32 channels of one tap), 36 K tiles
    constexpr int HKT = 28;                                   // K tile whose issue slot also requests the next halo (see below)
    static_assert(TH * TW == 256 && TW % 16 == 0, "256-pixel tiles of 16-pixel fragments");
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * WST + HALO];
    D2S_POISON_LDS(lds, NS * WST + HALO)
    u32x4* const halo = lds + NS * WST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WN, wave_n = wid % WN, grp = wid >> 2;
    const int fr = lane & 15, fg = lane >> 4, px = c3_lane_pixel<PST>(fr);
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;

    // tiles: XCD x (= blockIdx % 8) owns the contiguous run [x per, (x + 1) per) and its CUs walk it together
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3, per = (ntiles + 7) >> 3;
    auto tile_id = [&](int j) { return (j < per && xcd * per + j < ntiles) ? xcd * per + j : -1; };
    int j = slot, t = tile_id(j);
    if (t < 0) return;

    // W stage = 128 rows x 4 chunks; one LDS-DMA instruction per wave per stage: wave w brings rows 16 w .. 16 w + 15, lane l row
    // l / 4, slot l % 4.  Row r keeps chunk c in slot c ^ g(r / 4), g = {0, 2, 3, 1}: the 16-lane groups of a fragment read
    // (c3 bank rule above: rows S of k group c with rows S' of k group c + 1) then cover all 16 slots mod 16 once.
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W), 0, (unsigned)((long)((N + 255) / 256 * 256) * Kpad * 2), 0x00020000);
    auto wswz = [](int r) { const int k = (r >> 2) & 3; return (((k ^ (k >> 1)) & 1) << 1) | (k >> 1); };
    unsigned voW;
    {
        const int r = wid * 16 + (lane >> 2);
        voW = (unsigned)((long)r * Kpad * 2) + (unsigned)(((lane & 3) ^ wswz(r)) * 16);
    }
    auto issue_w = [&](int stage, int kt) { lds_dma16(rsW, lds + stage * WST + wid * 64, voW, kt * 64); };

    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    const short floor_ = a.relu ? (short)0 : (short)0x8000;
    u32x4 hr[NLD];
    auto tile_org = [&](int tt, int& b, int& ty0, int& tx0) {
        b = tt / (tiles_y * tiles_x);
        const int r = tt - b * (tiles_y * tiles_x);
        ty0 = (r / tiles_x) * TH; tx0 = (r % tiles_x) * TW;
    };
    auto load_halo = [&](int tt) {
        int b, ty0, tx0;
        tile_org(tt, b, ty0, tx0);
        const bf16_t* img = (const bf16_t*)a.ptr + (long)b * a.Hi * a.Wi * a.C;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 512;
            const int p = idx >> 4, c = idx & 15;
            const int hy = p / HWD, hx = p - hy * HWD;
            int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            const bool in = idx < NCH && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
            iy = in ? iy : 0; ix = in ? ix : 0;                 // (always ONE load per k: the vmcnt arithmetic counts them)
            const u32x4 v = *(const u32x4*)(img + ((long)iy * a.Wi + ix) * a.C + c * 8);
            hr[k] = in ? v : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 512;
            if (idx < NCH) {
                s16x8_ x = __builtin_bit_cast(s16x8_, hr[k]);
                x = __builtin_elementwise_max(x, (s16x8_){floor_, floor_, floor_, floor_, floor_, floor_, floor_, floor_});
                halo[(idx >> 4) * PST + c3_chunk_slot<PST>(idx & 15)] = __builtin_bit_cast(u32x4, x);
            }
        }
    };

    // A fragment i of this wave = tile fragment 4 wave_m + i: tile row f / FPR, columns 16 (f % FPR) ..; halo pixel (row + ky, col + kx)
    const u32x4* hb[FM];
    int frow[FM], fcol[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int f = wave_m * FM + i;
        frow[i] = f / FPR; fcol[i] = (f % FPR) * 16 + px;
        hb[i] = halo + (frow[i] * HWD + fcol[i]) * PST + c3_chunk_slot<PST>(fg);
    }
    int wro[FN];                                              // W fragment of n block jn: row wave_n 64 + 16 jn + fr, chunk fg
#pragma unroll
    for (int jn = 0; jn < FN; ++jn) { const int r = wave_n * 64 + jn * 16 + fr; wro[jn] = r * 4 + (fg ^ wswz(r)); }

#pragma unroll
    for (int k = 0; k < PD - 1; ++k) issue_w(k, k);
    load_halo(t);
    store_halo();
    int gb = 0;                                               // ring stage of this tile's K tile 0
    C3_T(t_begin)
    while (t >= 0) {
        C3_T(t_a)
        const int tn = tile_id(j + nslot);
        const int tl = tn >= 0 ? tn : t;                      // (no next tile: the loads are still issued -- counted -- and dropped)
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) acc[i][jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // K tile kt = tap kt / 4, channels 32 (kt % 4) ..: one MFMA K step.  Fragments are read ONE K tile ahead of the MFMAs that use
        // them (two register sets), right behind the barrier that makes that stage visible.
        u32x4 fa[2][FM], fb[2][FN];
        auto read_frags = [&](auto ktc, u32x4* fa_, u32x4* fb_) {
            constexpr int kt = decltype(ktc)::value;
            constexpr int tap = kt / 4, q = kt % 4, ky = tap / 3, kx = tap % 3;
            const u32x4* B_l = lds + ((gb + kt) & (NS - 1)) * WST;
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) fb_[jn] = B_l[wro[jn]];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa_[i] = hb[i][(ky * HWD + kx) * PST + q * 4];
        };
        // Schedule: the eight waves are two groups of four (waves w and w + 4 share a SIMD) that run the same K loop ONE BARRIER SLOT
        // apart: slot = X (the 16 MFMAs of K tile kt) or M (request W(kt + 7), read the fragments of kt + 1, wait for my part of
        // W(kt + 2)); while one group is in X its SIMD partner is in M.  With everybody in the same phase a barrier per K tile cost
        // ~340 cycles of idle matrix pipe (barrier release + LDS-DMA issue + fragment reads: K loop 13.9 us against 8.9 without
        // barriers, tools/c3_timeline.py).  Dependences: frags(kt + 1) are read in M(kt); every wave waited for its part of
        // W(kt + 1) in its M(kt - 1), at least one barrier earlier; W(kt + 7) overwrites the stage of K tile kt - 1, whose MFMAs every
        // wave has issued before the barrier in front of anybody's M(kt).
        c3_wait_vm<0>();                                      // W(0 .. 5) of this tile, the previous tile's stores; my halo stores (lgkmcnt)
        __builtin_amdgcn_s_barrier();
        issue_w((gb + PD - 1) & (NS - 1), PD - 1);            // W(6) (stage of the previous tile's K tile 34: free since that barrier)
        read_frags(std::integral_constant<int, 0>{}, fa[0], fb[0]);
        if (grp) __builtin_amdgcn_s_barrier();                // group 1 runs one slot behind
        static_for<NKT>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) mma_chunk(acc[i][jn], fb[kt & 1][jn], fa[kt & 1][i], bf16_t());
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if constexpr (kt < NKT - 1) {
                issue_w((gb + kt + PD) & (NS - 1), (kt + PD) % NKT);    // (among the MFMAs of the X slot instead: K loop 12.4 -> 13.9 us)
                if constexpr (kt == HKT) load_halo(tl);
                read_frags(std::integral_constant<int, kt + 1>{}, fa[(kt + 1) & 1], fb[(kt + 1) & 1]);
                // my part of W(kt + 2) (younger: W(kt + 3 .. kt + 7), and the halo loads where they are younger)
                if constexpr (kt + 2 < NKT) {
                    if constexpr (kt >= HKT && kt <= HKT + PD - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD + PD - 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD - 2) : "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
#ifdef D2S_C3_TIMING
            if constexpr (kt == 0) { C3_ACC(4, wall_clock64() - t_a) }
#endif
        });
        if (!grp) __builtin_amdgcn_s_barrier();               // the groups meet again
        C3_T(t_b)
        C3_ACC(0, t_b - t_a)
        // ---- epilogue, in two phases: every residual load of the wave first (one round trip instead of sixteen dependent ones:
        // the output may alias the residual, so the compiler keeps each load behind the previous store), the barrier and the halo
        // store under their latency, then bias / activation / residual / packed stores.  Same operation order as epilogue4<bf16_t>.
        // (requesting the residual values inside the K loop -- K tile 28 with the halo, or K tile 33 -- made the loop 3-4 us slower
        // per tile than it saved here, measured)
        int b, ty0, tx0;
        tile_org(t, b, ty0, tx0);
        bool ok[FM];
        long mo[FM];
        uint2 rr[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int y = ty0 + frow[i], x = tx0 + fcol[i];
            ok[i] = y < a.Ho && x < a.Wo;
            mo[i] = ok[i] ? ((long)(b * a.Ho + y) * a.Wo + x) * e.ldc + wave_n * 64 + fg * 4 : (long)(wave_n * 64 + fg * 4);
        }
        auto load_res = [&]() {
            const bf16_t* rp = (const bf16_t*)e.res1;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jn = 0; jn < FN; ++jn) rr[i][jn] = *(const uint2*)(rp + mo[i] + jn * 16);
        };
        float cb[FN][4];                                      // bias of this lane's columns
#pragma unroll
        for (int jn = 0; jn < FN; ++jn) {
            if (e.bias) load4(e.bias + wave_n * 64 + jn * 16 + fg * 4, cb[jn]);
            else cb[jn][0] = cb[jn][1] = cb[jn][2] = cb[jn][3] = 0.f;
        }
        if (e.res1) load_res();
        C3_T(t_c)
        C3_ACC(1, t_c - t_b)
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): my halo reads are done
        __builtin_amdgcn_s_barrier();                         // everybody's are: the halo buffer may be overwritten
        if (tn >= 0) store_halo();
        C3_T(t_d)
        C3_ACC(2, t_d - t_c)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int jn = 0; jn < FN; ++jn) {
                float v[4] = {acc[i][jn][0] + cb[jn][0], acc[i][jn][1] + cb[jn][1], acc[i][jn][2] + cb[jn][2], acc[i][jn][3] + cb[jn][3]};
                if (e.act == ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (e.res1) {
                    const uint2 r = rr[i][jn];
                    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                }
                if (ok[i]) *(uint2*)((bf16_t*)e.out + mo[i] + jn * 16) = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
            }
        C3_ACC(6, wall_clock64() - t_d)
        C3_ACC(3, 1)
        gb = (gb + NKT) & (NS - 1); t = tn; j += nslot;
    }
    c3_wait_vm<0>();                                          // the ring's last requests must land before the LDS is released
    C3_ACC(5, wall_clock64() - t_begin)
}

// (A persistent form of the 128-channel kernel -- one 8-wave block per CU, next halo prefetched by four "halo" waves while four
// "weight" waves ran a 3-stage LDS-DMA ring across tile boundaries -- was built and measured at batch 32: 226-246 us per
// 84 x 148 convolution against 168-185 us for the one-shot blocks above, head conv1 446 against 370.  With 16 KiB weight stages the
// ring is two K tiles deep at most beside two halo buffers, i.e. ~1 000 cycles of cover for an LDS-DMA that needs ~2 000 under
// load; two independent blocks per CU hide that better than one deeper-pipelined one.  Removed.)

// Eligible: bf16, stride 1, same-size output, C = 64 | 128, K = 9 C, plain row mapping or the fused head, enough tiles to fill the
// chip.  D2S_NO_HALO2=1 keeps the first-generation kernels (the parity tests run both).
bool launch_conv3_halo2(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st, bool dry) {
    static EnvInt off{"D2S_NO_HALO2", 0};
    if (off.get()) return false;
    if (a.mode != A_CONV3 || a.stride != 1 || a.Hi != a.Ho || a.Wi != a.Wo || (a.C != 64 && a.C != 128) || K != 9 * a.C) return false;
    if (!(e.map == MAP_ROWS || e.map == MAP_HEAD) || e.rows_per_img || e.ln_stats || e.stats_out || e.deq || e.ksplit > 1) return false;
    if (e.map == MAP_HEAD && N > 32) return false;
    if (N != 32 && N != 64 && (N & 127)) return false;
    const int nimg = M / (a.Ho * a.Wo);
    if ((long)nimg * a.Ho * a.Wo != M) return false;
    static EnvInt no_persist{"D2S_NO_HEADP", 0};
    // (from ~8 tiles per CU: at batch 1-2 the 627 / 1 254 tiles are 2.4 / 4.9 rounds of 256 persistent blocks, and the one-shot blocks
    //  below -- 1 221 per frame, many per CU -- finish sooner: 25.9 -> 20.3 us at batch 1, even at batch 4)
    static EnvInt headp_min{"D2S_HEADP_MIN", 2048};
    if (!no_persist.get() && e.map == MAP_HEAD && a.C == 64 && N <= 32 && (long)nimg * cdiv(a.Ho, 16) * cdiv(a.Wo, 16) >= headp_min.get()) {
        static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
        const int ntiles = nimg * cdiv(a.Ho, 16) * cdiv(a.Wo, 16);
        static EnvInt no_headups{"D2S_NO_HEADUPS", 0};
        // (the staged 13 x 13 source window holds scales <= 0.6; ReLU-on-load is not part of the interpolating loader)
        if (a.ups && (no_headups.get() || a.usy > 0.6f || a.usx > 0.6f || a.relu)) return false;
        if (dry) return true;
        static EnvInt ups_v1{"D2S_HEADUPS_V1", 0};             // A/B aid: the lock-step kernel of round 3
        // (the source map is read through one buffer descriptor; a tap's source-column byte offset travels in the low 16 bits of the
        //  bpermute word of h_request: a source row must stay within 64 KiB, i.e. Ws <= 512 at C = 64 -- wider maps take <1>)
        const bool ups_fits = (long)nimg * a.Hs * a.Ws * a.C * 2 < (1L << 31) && (long)a.Ws * a.C * 2 <= 65536;
        if (a.ups && !ups_v1.get() && ups_fits) hipLaunchKernelGGL(conv3_head_ups_kernel, dim3(std::min(ncu, ntiles)), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e, ntiles);
        else if (a.ups) hipLaunchKernelGGL((conv3_head_kernel<1>), dim3(std::min(ncu, ntiles)), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e, ntiles);
        else hipLaunchKernelGGL((conv3_head_kernel<0>), dim3(std::min(ncu, ntiles)), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e, ntiles);
        return true;
    }
    // the head's conv1 at batch: persistent blocks with W in registers and the up-sample in the loader (conv3_c128_ups_kernel)
    static EnvInt c128_min{"D2S_HEAD1P_MIN", 2048};          // fewest tiles of 8 x 16 pixels (2048 = 6 frames of 168 x 296); 0 = off
    if (a.ups && a.C == 128 && N == 64 && e.map == MAP_ROWS && !a.relu && a.usy <= 0.5f && a.usx <= 0.5f && a.Hs >= 2 && a.Ws >= 2 &&
        (e.out_type == OUT_T || e.out_type == OUT_BF16) && !e.scale && !e.res1 && !e.res2 && e.act == ACT_NONE && !e.deq && !e.out2 && !(e.ldc & 7) &&
        c128_min.get() > 0 && (long)nimg * cdiv(a.Ho, 8) * cdiv(a.Wo, 16) >= c128_min.get() &&
        (long)nimg * cdiv(a.Ho, 8) * cdiv(a.Wo, 16) < (1L << 30) && (long)nimg * a.Hs * a.Ws * a.C * 2 < (1L << 31)) {
        if (dry) return true;
        static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
        const int ntl = (int)((long)nimg * cdiv(a.Ho, 8) * cdiv(a.Wo, 16));
        GemmEpi e1 = e; e1.ksplit = 1;
        hipLaunchKernelGGL(conv3_c128_ups_kernel, dim3(std::min(ncu & ~7, ntl)), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e1, ntl);
        return true;
    }
    static EnvInt no_wide{"D2S_NO_WIDE", 0};
    if (!no_wide.get() && !a.ups && e.map == MAP_ROWS && a.C == 128 && N == 128 && (long)gemm_npad(N) * Kpad * 2 < (1L << 31) &&
        (e.out_type == OUT_T || e.out_type == OUT_BF16) && !e.scale && !e.res2 && !e.res1_mod && (e.act == ACT_NONE || e.act == ACT_RELU)) {
        static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
        static EnvInt wide_min{"D2S_WIDE_MIN", 384};                                                       // tiles (1.5 rounds of the CUs)
        const long pad_a = (long)cdiv(a.Ho, 8) * cdiv(a.Wo, 32), pad_b = (long)cdiv(a.Ho, 16) * cdiv(a.Wo, 16);   // 256-pixel tiles per image
        const long ntl = nimg * std::min(pad_a, pad_b);
        const int grid_w = ncu & ~7;
        if (ntl >= wide_min.get() && ntl < (1L << 30) && grid_w >= 8) {
            if (dry) return true;
            GemmEpi e1 = e; e1.ksplit = 1;
            if (pad_a <= pad_b) hipLaunchKernelGGL((conv3_wide_kernel<8, 32>), dim3(grid_w), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e1, (int)ntl);
            else hipLaunchKernelGGL((conv3_wide_kernel<16, 16>), dim3(grid_w), dim3(512), 0, st, a, (const bf16_t*)W, N, Kpad, e1, (int)ntl);
            return true;
        }
    }
    const long tiles_m = (long)nimg * cdiv(a.Ho, 8) * cdiv(a.Wo, 16);
    // Mid-size maps (round 5): the fusion stage on the 84 x 148 map of the batch-1 frame (110 tiles x 2 blocks of 64 channels = 220
    // blocks, every one resident at once) ran as implicit-GEMM tiles + a split-K reduce launch, 19.5 us per convolution of 3.7 GF.  With
    // one block per CU a block can afford the LDS: ten 8 KB weight stages requested ahead (NS = 10: half of its 18 K tiles in the
    // prologue) and the halo fill with all of a thread's chunks in flight (HG = 6 instead of 3 per pass: nobody else covers its round
    // trips here): 19.5 -> 15.5 us, two launches per frame.  The SMALLER maps (11 x 19 ... 42 x 74: 16-120 blocks) were tried the same
    // way with all 72 KB of a 32-channel block's weights up front (NS = 19, HG = 12): 11.6-12.7 us before, 11.4-14.9 after -- those
    // launches are not paced by the K loop's round trips (boundary, cold code and the epilogue are what is left); not kept.
    // D2S_HALO2_DEEP=0: off
    static EnvInt deep_on{"D2S_HALO2_DEEP", 1};
    static const int ncu_d = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
    if (deep_on.get() && a.C == 128 && e.map == MAP_ROWS && N % 64 == 0 && (long)gemm_npad(N) * Kpad * 2 < (1L << 31) &&
        tiles_m * (N / 64) <= ncu_d && tiles_m * (N / 64) * 2 > ncu_d) {
        if (dry) return true;
        GemmEpi e1 = e; e1.ksplit = 1;
        unsigned grid = 0;
        const int xn = pick_xn((int)tiles_m, N / 64, 64, Kpad, 2, grid);
        hipLaunchKernelGGL((conv3_halo2_kernel<16, 17, 64, 4, 2, 10, 6>), dim3(grid), dim3(512), 0, st, a, (const bf16_t*)W, M, N, Kpad, e1, xn);
        return true;
    }
    const int bn = N <= 32 ? 32 : (N <= 64 ? 64 : 128);
    static EnvInt halo2_min{"D2S_HALO2_MIN", 384};        // (head conv1 at batch 1: 399 tiles, 26.5 -> 21.8 us here; 110-tile maps lose)
    if (tiles_m * cdiv(N, bn) < halo2_min.get()) return false;    // small maps: latency-bound, the small-tile kernels do better
    if ((long)gemm_npad(N) * Kpad * 2 >= (1L << 31)) return false;
    if (dry) return true;
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

#ifdef D2S_C3U_TIMING
extern "C" int d2s_c3u_timing(unsigned long long* out, int clear) {       // out != null: read 256 x 9 counters
    if (out) D2S_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(d2s::c3u_timing), sizeof(unsigned long long) * 256 * 9));
    if (clear) { static unsigned long long z[256 * 9]; D2S_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d2s::c3u_timing), z, sizeof(z))); }
    return D2S_OK;
}
#endif
#ifdef D2S_C3_TIMING
extern "C" int d2s_c3_timing(unsigned long long* out, int clear) {        // out != null: read 256 x 8 counters
    if (out) D2S_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(d2s::c3_timing), sizeof(unsigned long long) * 256 * 8));
    if (clear) { static unsigned long long z[256 * 8]; D2S_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d2s::c3_timing), z, sizeof(z))); }
    return D2S_OK;
}
#endif
