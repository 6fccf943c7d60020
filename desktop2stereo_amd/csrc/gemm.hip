// MFMA GEMM for gfx950: C[m,n] = epi(sum_k A[m,k] W[n,k]).
//
//  * operands bf16 (v_mfma_f32_16x16x32_bf16), e4m3 (v_mfma_f32_16x16x32_fp8_fp8) or f32 (v_mfma_f32_16x16x4_f32,
//    exact fp32), fp32 accumulate;
//  * tile BM x BN x 128 bytes of K, WM x WN waves; two data paths for the K tiles (gemm_glds_kernel): LDS-DMA into an
//    NS-stage ring, or register staging into two LDS stages (STG) -- the tile rule in launch_t picks per launch;
//  * both operands are K-contiguous; a 16-byte chunk per lane is the unit everywhere:
//      global -> LDS (16 B/lane, 8 lanes cover one 128-B row = full cache lines), chunk XOR-swizzle
//      phys = chunk ^ ((row >> 1) & 7), applied on the source address for LDS-DMA
//      LDS -> fragments with ds_read_b128 at (row = lane & 15, chunk = kstep*4 + (lane >> 4)),
//    conflict-free for 128-byte rows.  One chunk feeds one bf16 MFMA (K=32 across the 4 lane groups), two e4m3
//    MFMAs or four f32 MFMAs (the k permutation is the same for both operands, so the sum is unchanged);
//  * operands are swapped (first = W rows, second = A rows) so each lane ends up with 4 consecutive
//    n for one m: bias / LayerScale / residual / output move as 8- or 16-byte vectors;
//  * the A loader is either a plain row-major matrix or an implicit 3x3 (pad 1, stride 1|2)
//    convolution over an NHWC activation, with optional ReLU-on-load (pre-activation units); stride-1 convs on
//    the large maps use conv3_halo_kernel (input tile resident in LDS) instead.
#include "gemm_epi.h"

namespace d2s {

// ================================================================================================
// LDS-DMA ring.  The K tiles travel
// global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging VGPRs, no
// ds_write pass) into a ring of NS stages, PD = NS-1 tiles ahead, with counted vmcnt and ONE raw
// s_barrier per tile:
//     wait my loads of tile t (vmcnt <= (PD-1) tiles in flight)  ->  s_barrier  (all loads of t
//     landed AND everybody finished computing t-1)  ->  issue tile t+PD into the stage t-1 used
//     ->  compute t.
// LDS-DMA writes lane-linear (base + lane*16), so the XOR swizzle moves to the SOURCE address:
// the lane that owns LDS slot (row r, phys chunk p) fetches global chunk p ^ ((r>>1)&7)
// (cdna_hip_programming.md rule 21: linear dest + swizzled source + swizzled read).
// Out-of-range rows / K tail / conv padding fetch from a zero page.  ReLU-on-load is applied when
// the A fragments are read (v_pk_max_i16 for bf16).
// ================================================================================================
__device__ u32x4 d2s_zero_page[4];

// tuning aid (build with D2S_HIPCC_DEFS=-DD2S_GLDS_TIMING): thread 0 of every block of gemm_glds_kernel stamps the 100 MHz wall
// clock at entry / ring primed / first K tile landed / K loop done / epilogue done; tools/glds_timeline.py reads them
#ifdef D2S_GLDS_TIMING
__device__ unsigned long long glds_timing[4096 * 8];
#define GL_STAMP(SLOT) { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) glds_timing[blockIdx.x * 8 + (SLOT)] = wall_clock64(); }
#else
#define GL_STAMP(SLOT) {}
#endif

// max(x, floor) on a fragment: floor = 0 gives ReLU, floor = lowest gives identity (no branch in the MFMA stream).
// bf16 as int16: sign bit set <=> negative, and positive bf16 order like positive int16 -> v_pk_max_i16.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4 relu_frag(u32x4 v, int floor_bits, bf16_t) {
    s16x8 x = __builtin_bit_cast(s16x8, v);
    short f = (short)floor_bits;
    x = __builtin_elementwise_max(x, (s16x8){f, f, f, f, f, f, f, f});
    return __builtin_bit_cast(u32x4, x);
}
__device__ __forceinline__ u32x4 relu_frag(u32x4 v, int, fp8_t) { return v; }
__device__ __forceinline__ u32x4 relu_frag(u32x4 v, int, bx3_t) { return v; }   // bf16x3: ReLU-on-load happens on the fp32 values as they are staged   // e4m3 operands: encoder linears only, no ReLU-on-load
__device__ __forceinline__ u32x4 relu_frag(u32x4 v, int floor_bits, float) {
    f32x4 x = __builtin_bit_cast(f32x4, v);
    float f = floor_bits == 0 ? 0.f : -3.0e38f;
    x = __builtin_elementwise_max(x, (f32x4){f, f, f, f});
    return __builtin_bit_cast(u32x4, x);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// chunk XOR for a row of CPR chunks: conflict-free ds_read_b128 for 128-byte (CPR 8) and 256-byte (CPR 16) rows
// (64-byte rows, CPR 4: rows r and r+2 share banks -> XOR by (r >> 1) & 3)
template <int CPR> __device__ __forceinline__ int swz_row(int r) { return CPR == 8 ? ((r >> 1) & 7) : (CPR == 4 ? ((r >> 1) & 3) : (r & 15)); }

// STG = true: register-staged variant of the same tile.  The K tile travels global -> VGPRs (global_load_dwordx4)
// -> LDS (ds_write_b128, same lane-linear slots as the LDS-DMA path, so fragment reads are unchanged) with TWO LDS
// stages: at the top of iteration t (after the barrier) the registers holding tile t+1 -- loaded during compute(t-1)
// -- are written to the stage compute(t-1) just released, the loads of tile t+2 are issued, then tile t is computed.
// Measured on this chip (tools/ubench/l2_to_lds): L2-resident data reaches LDS at ~27 TB/s through VGPRs vs
// ~13-16 TB/s by LDS-DMA, and the freed LDS (2 stages instead of 3) admits the 256 x 256 tile.
// (An intra-block split-K variant -- several wave groups per block, each running the pipeline over every KG-th K tile,
// accumulators summed through LDS -- was built and swept at batch 1: the batch-1 launches are L2->LDS-bandwidth-bound
// (~10 TB/s aggregate), not latency-bound; it won 9 % on FC2 in isolation and lost 2 % in the pipeline.  Removed.)
// both kernels of this file that warm their argument lines (KERNARG_WARM, common.h: one dword of every 64-byte line up to 0x140) carry
// at least a GemmA and a GemmEpi by value behind >= 8 bytes of scalars
static_assert(sizeof(GemmA) + sizeof(GemmEpi) + 8 >= KERNARG_WARM_BYTES, "KERNARG_WARM reads past the kernarg segment");
template <typename T, int BM, int BN, int WM, int WN, int NS, int CPR, int STG = 0>
__global__ void __launch_bounds__(64 * WM * WN)
gemm_glds_kernel(const T* __restrict__ W, const void* a_ptr, long a_lda, int M, int N, int K, int Kpad, int xn, GemmA a, GemmEpi e) {
    // Argument order: what the tile map and the first operand requests need -- 11 dwords -- comes first, so that the hardware's kernarg
    // PRELOAD (gfx950: the first 16 dwords arrive in SGPRs with the wave; -mllvm -amdgpu-kernarg-preload-count=16, build.py) covers
    // it: the lean (latency-regime) instantiations request their first K tiles before any kernarg load has returned.  a_ptr / a_lda
    // repeat a.ptr / a.lda for that purpose.
    // STG: 0 = LDS-DMA ring, every loader; 1 = register-staged, two LDS stages; 2 = "lean" LDS-DMA ring: descriptor-addressed plain
    // linears only (a.buf guaranteed by the launcher) -- the pointer / implicit-conv loaders and their per-row set-up are compiled out
    static_assert(STG == 0 || STG == 2 || (STG == 1 && NS == 2), "the register-staged variant uses two LDS stages");
    constexpr bool LEAN = STG == 2;
    constexpr bool BX3 = std::is_same<T, bx3_t>::value;   // A: fp32 in memory, split hi / lo when the staging registers are stored
    static_assert(!BX3 || CPR == 8, "bf16x3 units are laid out for 128-byte K tiles");
    // (bf16x3 with STG == 0: both operands already in the unit format, LDS-DMA like any other type; STG == 1: A is fp32)
    constexpr int CE = Prec<T>::CE;
    constexpr int BK = CPR * CE;                    // K tile: CPR 16-byte chunks per row (8 -> 128 B, 16 -> 256 B)
    constexpr int RPI = 64 / CPR;                   // rows covered by one 1-KiB LDS-DMA wave-instruction
    constexpr int NW = WM * WN;                     // waves per block
    constexpr int AI = BM / (RPI * NW), BI = BN / (RPI * NW);   // LDS-DMA instructions per thread per tile (A / W)
    constexpr int LPT = AI + BI;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;   // 16x16 fragments per wave
    constexpr int PD = NS - 1;
    constexpr int STAGE = (BM + BN) * CPR;          // chunks per stage
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * STAGE];
    KERNARG_WARM(kaw_)                               // every 64-byte line of the ~390 argument bytes in one round trip (common.h)
    if constexpr (STG != 2) { KERNARG_WARM_END(kaw_) }   // (lean: waited for behind the first operand requests)
    D2S_POISON_LDS(lds, NS * STAGE)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WN, wave_n = wid % WN;
    int tm_, tn_;
    GL_STAMP(0)
    if (!tile_of_block(blockIdx.x, (M + BM - 1) / BM, (N + BN - 1) / BN, xn, tm_, tn_)) return;
    const int bm0 = tm_ * BM, bn0 = tn_ * BN;
    // this lane's slot inside a 64-slot wave-instruction: row lane/CPR of RPI, phys chunk lane%CPR;
    // its source chunk is the same for every instruction i (rows differ by multiples of RPI*NW,
    // which the row swizzle's period divides)
    static_assert((RPI * NW) % 16 == 0, "row swizzle must be invariant across a thread's instructions");
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "tile must split evenly over the waves");
    const int lrow = wid * RPI + lane / CPR;
    const int src_chunk = (lane % CPR) ^ swz_row<CPR>(lrow);
    const T* zero = (const T*)d2s_zero_page;

    const T* arow[AI];
    int aiy[AI], aix[AI];
    bool aok[AI];
    if constexpr (!LEAN)
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int m = bm0 + i * NW * RPI + lrow;
        aok[i] = m < M;
        int mm = aok[i] ? m : 0;
        if (a.mode == A_PLAIN) { arow[i] = (const T*)a.ptr + (long)mm * a.lda; aiy[i] = aix[i] = 0; }
        else {
            int ox = mm % a.Wo, oy = (mm / a.Wo) % a.Ho, b = mm / (a.Wo * a.Ho);
            aiy[i] = oy * a.stride - 1; aix[i] = ox * a.stride - 1;
            arow[i] = (const T*)a.ptr + (long)b * a.Hi * a.Wi * a.C;
        }
    }
    const T* wrow = W + (long)(bn0 + lrow) * Kpad + src_chunk * CE;
    const float inv_c = a.mode == A_CONV3 ? 1.0f / (float)a.C : 0.f;
    // Plain linears (a.buf, LDS-DMA path): both operands through buffer descriptors.  The per-lane byte offsets below are fixed
    // for the whole K loop and a K tile is ONE scalar offset -- no 64-bit address arithmetic, bounds selects or zero-page
    // pointers per load (at batch 1 that was ~20 VALU + ~20 SALU per K tile next to 4 MFMAs).  Rows past M get an offset past
    // every descriptor's range: the hardware returns zeros for them.
    constexpr int ES = (int)sizeof(T);
    // (descriptors are built unconditionally -- a few scalar instructions; the offsets only where the path is taken)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a_ptr), 0, (unsigned)((long)M * a_lda * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(W), 0, (unsigned)((long)((N + 255) / 256 * 256) * Kpad * ES), 0x00020000);
    unsigned voA[AI], voW[BI];
    if constexpr (STG != 1) {
        if (LEAN || a.buf) {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int m = bm0 + i * NW * RPI + lrow;
                voA[i] = m < M ? (unsigned)((long)m * a_lda * ES) + (unsigned)(src_chunk * 16) : 0x80000000u;
            }
#pragma unroll
            for (int i = 0; i < BI; ++i) voW[i] = (unsigned)((long)(bn0 + lrow + RPI * NW * i) * Kpad * ES) + (unsigned)(src_chunk * 16);
        }
    }
#define D2S_ISSUE_BUF(KT)                                                                                        \
    {                                                                                                            \
        u32x4* st_ = lds + ((KT) % NS) * STAGE;                                                                  \
        const int so_ = ((KT) + kt0) * (BK * ES);                                                                \
        _Pragma("unroll") for (int i = 0; i < AI; ++i) lds_dma16(rsA, st_ + (i * NW + wid) * 64, voA[i], so_);    \
        _Pragma("unroll") for (int i = 0; i < BI; ++i) lds_dma16(rsW, st_ + BM * CPR + (i * NW + wid) * 64, voW[i], so_); \
    }

    // lean instantiations (one K range per block: the launcher never splits K for them): the ring is primed HERE, from preloaded
    // arguments only; everything that reads the rest of the kernarg segment (LayerNorm statistics, epilogue set-up) comes after
    if constexpr (LEAN) {
        constexpr int kt0 = 0;
        const int nkt_ = (K + BK - 1) / BK;
#pragma unroll
        for (int t = 0; t < PD; ++t)
            if (t < nkt_) D2S_ISSUE_BUF(t)
        KERNARG_WARM_END(kaw_)
    }

    // D2S_MOVE(slot index, source, LDS destination): LDS-DMA straight into the ring, or a load into staging registers
    u32x4 stg[1][STG == 1 ? LPT : 1];
#define D2S_MOVE(S, IDX, SRC, DST)                                                                                \
    do {                                                                                                         \
        if constexpr (STG == 1) stg[S][IDX] = *(const u32x4*)(SRC);                                                 \
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),             \
                                              (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);         \
    } while (0)
#define D2S_ISSUE_TILE(KT, S)                                                                                    \
    {                                                                                                            \
        u32x4* st_ = lds + ((KT) % NS) * STAGE;                                                                  \
        const int k_ = ((KT) + kt0) * BK + src_chunk * CE;                                                       \
        if (a.mode == A_PLAIN) {                                                                                 \
            _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                     \
                const T* s_ = (aok[i] && k_ < K) ? arow[i] + k_ : zero;                                          \
                D2S_MOVE(S, i, s_, st_ + (i * NW + wid) * 64);                                                    \
            }                                                                                                    \
        } else {                                                                                                 \
            int tap_ = (int)(((float)k_ + 0.5f) * inv_c);                                                        \
            int c0_ = k_ - tap_ * a.C;                                                                           \
            int ky_ = tap_ / 3, kx_ = tap_ - ky_ * 3;                                                            \
            _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                     \
                int iy_ = aiy[i] + ky_, ix_ = aix[i] + kx_;                                                      \
                bool ok_ = aok[i] && k_ < K && iy_ >= 0 && iy_ < a.Hi && ix_ >= 0 && ix_ < a.Wi;                 \
                const T* s_ = ok_ ? arow[i] + ((long)iy_ * a.Wi + ix_) * a.C + c0_ : zero;                       \
                D2S_MOVE(S, i, s_, st_ + (i * NW + wid) * 64);                                                    \
            }                                                                                                    \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < BI; ++i)                                                           \
            D2S_MOVE(S, AI + i, wrow + (long)(RPI * NW * i) * Kpad + ((KT) + kt0) * BK, st_ + BM * CPR + (i * NW + wid) * 64); \
    }
    // staging registers -> LDS, the same lane-linear slots the LDS-DMA path fills.  bf16x3: this lane's A chunk is 4 fp32 values
    // (logical chunk c = src_chunk of the row's K tile): their bf16 hi / lo pieces are the (c & 1) halves of the hi / lo chunk of
    // unit c >> 1, i.e. logical chunks 2 (c >> 1) and 2 (c >> 1) + 1, swizzled like every other chunk of the row.
    const int bx_row = (lane / CPR) * CPR, bx_sw = swz_row<CPR>(lrow);
    const int bx_hi = (bx_row + (((src_chunk >> 1) * 2) ^ bx_sw)) * 2 + (src_chunk & 1);           // in 8-byte units
    const int bx_lo = (bx_row + (((src_chunk >> 1) * 2 + 1) ^ bx_sw)) * 2 + (src_chunk & 1);
#define D2S_STORE_STG(KT, S)                                                                                     \
    {                                                                                                            \
        u32x4* st_ = lds + ((KT) % NS) * STAGE;                                                                  \
        if constexpr (BX3) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                     \
                f32x4 x_ = __builtin_bit_cast(f32x4, stg[S][i]);                                                 \
                if (a.relu) x_ = __builtin_elementwise_max(x_, (f32x4){0.f, 0.f, 0.f, 0.f});                     \
                uint2 h_, l_;                                                                                    \
                bx3_split4(x_, h_, l_);                                                                          \
                uint2* s2_ = (uint2*)(st_ + (i * NW + wid) * 64);                                                \
                s2_[bx_hi] = h_; s2_[bx_lo] = l_;                                                                \
            }                                                                                                    \
        } else {                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < AI; ++i) st_[(i * NW + wid) * 64 + lane] = stg[S][i];          \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < BI; ++i) st_[BM * CPR + (i * NW + wid) * 64 + lane] = stg[S][AI + i]; \
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split-K (e.ksplit > 1): blockIdx.y owns a contiguous range of K tiles and writes raw fp32 partials
    const int nkt_all = (K + BK - 1) / BK;
    const int ksplit = LEAN ? 1 : (e.ksplit > 1 ? e.ksplit : 1);
    const int kt0 = (int)(((long)nkt_all * blockIdx.y) / ksplit);
    const int nkt = (int)(((long)nkt_all * (blockIdx.y + 1)) / ksplit) - kt0;
    const int fr = lane & 15, fg = lane >> 4;
    // LN-folded consumer: the producer left (sum, sum of squares) per row and column block in e.ln_stats[slot][M][2].
    // The 4 lane groups of a row share the slots (slot = fg + 4q); the loads are issued here, before the K loop, and
    // summed after it.  (Registers, not LDS: a second __shared__ object makes the compiler track the LDS-DMA writes
    // and put s_waitcnt vmcnt(0) in front of every fragment read.)
    constexpr bool LN_ON = WN > 1;                  // the WN == 1 (fused-head) tiles never see an LN-folded linear
    constexpr int LNS = LN_ON ? 4 : 1;              // <= 16 column blocks
    float2 lnp[FM][LNS];
    if (LN_ON && e.ln_stats) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = bm0 + wave_m * (BM / WM) + i * 16 + fr;
#pragma unroll
            for (int q = 0; q < LNS; ++q) {
                const int sl = fg + 4 * q;
                lnp[i][q] = (m < M && sl < e.ln_slots) ? ((const float2*)e.ln_stats)[(long)sl * (e.ln_M ? e.ln_M : M) + m] : make_float2(0.f, 0.f);
            }
        }
    }
    // RELU is a literal: the K loop exists twice (with / without ReLU-on-load) so that plain linears do not pay
    // 4 v_pk_max per A fragment (a third of the loop's VALU work) for an identity
#define D2S_COMPUTE(KT, RELU)                                                                                    \
    {                                                                                                            \
        const u32x4* A_l = lds + ((KT) % NS) * STAGE + (wave_m * (BM / WM)) * CPR;                               \
        const u32x4* B_l = lds + ((KT) % NS) * STAGE + BM * CPR + (wave_n * (BN / WN)) * CPR;                    \
        if constexpr (BX3) {                                                                                     \
            /* one K = 32 step per 128-byte tile: lane group fg takes unit BX3_UNIT(fg) = chunks 2 u (hi), 2 u + 1 (lo) */ \
            const int cu_ = 2 * BX3_UNIT(fg);                                                                    \
            u32x4 fbh[FN], fbl[FN];                                                                              \
            _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                                     \
                int r = j * 16 + fr; fbh[j] = B_l[r * CPR + (cu_ ^ swz_row<CPR>(r))]; fbl[j] = B_l[r * CPR + ((cu_ + 1) ^ swz_row<CPR>(r))]; \
            }                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                                     \
                int r = i * 16 + fr;                                                                             \
                const u32x4 fah = A_l[r * CPR + (cu_ ^ swz_row<CPR>(r))], fal = A_l[r * CPR + ((cu_ + 1) ^ swz_row<CPR>(r))]; \
                _Pragma("unroll") for (int j = 0; j < FN; ++j) mma_bx3(acc[i][j], fbh[j], fbl[j], fah, fal);     \
            }                                                                                                    \
        } else {                                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < CPR / 4; ++ks) {                                                 \
            /* W fragments stay live for the k-step; A fragments stream through one at a time */                 \
            /* (keeps the 8-wave 256-row tiles inside the 256-register budget) */                                \
            u32x4 fb[FN];                                                                                        \
            _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                                     \
                int r = j * 16 + fr; fb[j] = B_l[r * CPR + ((ks * 4 + fg) ^ swz_row<CPR>(r))];                   \
            }                                                                                                    \
            _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                                     \
                int r = i * 16 + fr;                                                                             \
                u32x4 fa = A_l[r * CPR + ((ks * 4 + fg) ^ swz_row<CPR>(r))];                                     \
                if (RELU) fa = relu_frag(fa, 0, T());                                                            \
                _Pragma("unroll") for (int j = 0; j < FN; ++j) mma_chunk(acc[i][j], fb[j], fa, T());             \
            }                                                                                                    \
        }                                                                                                        \
        }                                                                                                        \
    }
#define D2S_K_LOOP(RELU)                                                                                         \
    if constexpr (STG == 1) {                                                                                    \
        for (int kt = 0; kt < nkt; ++kt) {                                                                       \
            __syncthreads();       /* tile kt visible, stage (kt+1)&1 released, my loads of kt+1 landed */        \
            if (kt + 1 < nkt) D2S_STORE_STG(kt + 1, 0)                                                           \
            if (kt + 2 < nkt) D2S_ISSUE_TILE(kt + 2, 0)                                                          \
            D2S_COMPUTE(kt, RELU)                                                                                \
        }                                                                                                        \
    } else {                                                                                                     \
        for (int kt = 0; kt < nkt; ++kt) {                                                                       \
            /* tiles kt .. min(kt+PD-1, nkt-1) are in flight; let all but tile kt stay in flight */              \
            if (kt + PD - 1 < nkt) wait_vmcnt<(PD - 1) * LPT>();                                                 \
            else wait_vmcnt<0>();                                                                                \
            __builtin_amdgcn_s_barrier();                                                                        \
            if (kt + PD < nkt) D2S_ISSUE_TILE(kt + PD, 0)                                                        \
            D2S_COMPUTE(kt, RELU)                                                                                \
        }                                                                                                        \
    }
    bool done_buf = false;
    if constexpr (STG != 1) {
        if (LEAN || a.buf) {                         // plain linear, descriptor-addressed ring (same ring, same waits)
            if constexpr (!LEAN) {
#pragma unroll
                for (int t = 0; t < PD; ++t)
                    if (t < nkt) D2S_ISSUE_BUF(t)
            }
            GL_STAMP(1)
            for (int kt = 0; kt < nkt; ++kt) {
                if (kt + PD - 1 < nkt) wait_vmcnt<(PD - 1) * LPT>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (kt == 0) { GL_STAMP(2) }
                if (kt + PD < nkt) D2S_ISSUE_BUF(kt + PD)
                D2S_COMPUTE(kt, 0)
            }
            done_buf = true;
        }
    }
    if constexpr (!LEAN)
    if (!done_buf) {
    if constexpr (STG == 1) {
        if (nkt > 0) { D2S_ISSUE_TILE(0, 0) D2S_STORE_STG(0, 0) }
        if (nkt > 1) D2S_ISSUE_TILE(1, 0)
    } else {
#pragma unroll
        for (int t = 0; t < PD; ++t)
            if (t < nkt) D2S_ISSUE_TILE(t, 0)
    }
    if (a.relu) { D2S_K_LOOP(1) } else { D2S_K_LOOP(0) }
    }
    GL_STAMP(3)
#undef D2S_ISSUE_BUF
#undef D2S_K_LOOP
#undef D2S_COMPUTE
#undef D2S_ISSUE_TILE
#undef D2S_STORE_STG
#undef D2S_MOVE

    // MAP_HEAD: the DPT head's tail fused into conv2 -- depth[m] = relu(b3 + sum_n w3[n] * relu(acc[m][n] + bias[n]))
    // (HF DepthAnythingDepthEstimationHead: conv2 -> ReLU -> conv3 (1x1, C->1) -> ReLU).  One wave owns all N
    // columns of its rows (WN == 1, N <= BN), so the channel sum is registers + two cross-lane-group shuffles.
    if constexpr (WN == 1) {
        if (e.map == MAP_HEAD) {
            static_for<FM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int m = bm0 + wave_m * (BM / WM) + i * 16 + fr;
                float s = 0.f;
                static_for<FN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int n0 = j * 16 + fg * 4;
                    if (n0 < N) {
                        float b[4], w[4];
                        load4(e.bias + n0, b); load4(e.scale + n0, w);
                        s += fmaxf(acc[i][j][0] + b[0], 0.f) * w[0] + fmaxf(acc[i][j][1] + b[1], 0.f) * w[1] +
                             fmaxf(acc[i][j][2] + b[2], 0.f) * w[2] + fmaxf(acc[i][j][3] + b[3], 0.f) * w[3];
                    }
                });
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (fg == 0 && m < M) ((float*)e.out)[m] = head_activation(s + e.head_b3, e.head_max_depth);
            });
            return;
        }
    }

    float2 ln_s[FM];
    float ln_cs[FN][4];                             // colsum(W') of this lane's columns: independent of the row fragment
    if (LN_ON && e.ln_csum) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
            if (n0 < N) load4(e.ln_csum + n0, ln_cs[j]);
            else { ln_cs[j][0] = ln_cs[j][1] = ln_cs[j][2] = ln_cs[j][3] = 0.f; }
        }
    }
    // Latency regime only (the lean instantiations, every block resident): column vectors and residual values of all the wave's
    // fragments are requested before the first store (gemm_epi.h, epi_res1_load).  In the throughput tiles the extra registers
    // cost resident waves (64 x 128: 70 -> 100 VGPRs, 6 -> 4 waves / SIMD; batch 4: -12 % frames/s, measured) and the other
    // waves cover the round trips anyway.
    constexpr bool PRE = STG == 2 && FM * FN <= 8;
    EpiCols cols[PRE ? FN : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
            if (n0 < N) epi_cols_load(e, n0, cols[j]);
        }
    }
    float pre[PRE ? FM : 1][PRE ? FN : 1][4];
    const bool pre_on = PRE && ksplit == 1 && epi_res1_ahead(e);
    if constexpr (PRE) {
        if (pre_on) {
            static_for<FM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int m = bm0 + wave_m * (BM / WM) + i * 16 + fr;
                static_for<FN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
                    if (m < M && n0 < N) epi_res1_load<T>(e, m, n0, pre[i][j]);
                });
            });
        }
    }
    // compile-time indices (a plain `#pragma unroll` over this large body is not honoured for the
    // 32-fragment tiles, and a run-time index would put the accumulators in scratch)
    static_for<FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = bm0 + wave_m * (BM / WM) + i * 16 + fr;
        float s1 = 0.f, s2 = 0.f;                  // LN-folded producer: this lane's share of the row's sum / sum of squares
        float ln_mean = 0.f, ln_rstd = 1.f;
        if (LN_ON && e.ln_csum) {
            float t1 = (lnp[i][0].x + lnp[i][1 % LNS].x) + (lnp[i][2 % LNS].x + lnp[i][3 % LNS].x);
            float t2 = (lnp[i][0].y + lnp[i][1 % LNS].y) + (lnp[i][2 % LNS].y + lnp[i][3 % LNS].y);
            t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
            t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
            ln_mean = t1 / (float)e.ln_dim;
            ln_rstd = rsqrtf(fmaxf(t2 / (float)e.ln_dim - ln_mean * ln_mean, 0.f) + e.ln_eps);
        }
        if (m < M) {
            static_for<FN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
                if (n0 < N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (ksplit > 1) store4(e.part + ((long)blockIdx.y * M + m) * N + n0, v);     // reduced by splitk_reduce_kernel
                    else {
                        const bool ln = LN_ON && e.ln_csum != nullptr;
                        if (ln) {
                            if (e.deq) { float q[4]; load4(e.deq + n0, q); v[0] *= q[0]; v[1] *= q[1]; v[2] *= q[2]; v[3] *= q[3]; }   // e4m3 operands -> real units first
                            v[0] = ln_rstd * (v[0] - ln_mean * ln_cs[j][0]); v[1] = ln_rstd * (v[1] - ln_mean * ln_cs[j][1]);
                            v[2] = ln_rstd * (v[2] - ln_mean * ln_cs[j][2]); v[3] = ln_rstd * (v[3] - ln_mean * ln_cs[j][3]);
                        }
                        epilogue_dispatch<T>(e, m, n0, v, ln, PRE && pre_on ? pre[PRE ? i : 0][PRE ? j : 0] : nullptr, PRE ? &cols[PRE ? j : 0] : nullptr);
                        s1 += (v[0] + v[1]) + (v[2] + v[3]);
                        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
            });
        }
        if (LN_ON && e.stats_out) {                 // the 4 lane groups of a row hold 4 columns each of every fragment
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            ln_s[i] = make_float2(s1, s2);
        }
    });
    if (LN_ON && e.stats_out) {
        // the block's WN waves cover BN columns of the same rows: add them up through the (now idle) stage memory so the
        // consumer reads one partial per column block.  Fixed order -> bit-reproducible.
        float2* red = (float2*)lds;
        __syncthreads();                            // every wave is done reading its last fragments
#pragma unroll
        for (int i = 0; i < FM; ++i)
            if (fg == 0) red[wave_n * BM + wave_m * (BM / WM) + i * 16 + fr] = ln_s[i];
        __syncthreads();
        for (int r = tid; r < BM; r += 64 * NW) {
            float2 t = red[r];
            for (int w = 1; w < WN; ++w) { t.x += red[w * BM + r].x; t.y += red[w * BM + r].y; }
            if (bm0 + r < M) ((float2*)e.stats_out)[(long)(bn0 / BN) * M + bm0 + r] = t;
        }
    }
#ifdef D2S_GLDS_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the epilogue's stores have left
    GL_STAMP(4)
#endif
}

// split-K second pass: sum the fp32 partials of all splits and run the fused epilogue once
template <typename T>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(GemmEpi e, int M, int N) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int n4 = N >> 2;
    if (idx >= (long)M * n4) return;
    int m = (int)(idx / n4), n0 = (int)(idx % n4) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < e.ksplit; ++s) {
        float p[4];
        load4(e.part + ((long)s * M + m) * N + n0, p);
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
    epilogue_dispatch<T>(e, m, n0, v);
}

// ================================================================================================
// 3x3 convolution (stride 1, pad 1, NHWC) with the INPUT resident in LDS.  The implicit-GEMM loader above fetches
// every input pixel nine times from L2 (once per tap); here a block owns an 8 x 16 output tile, loads its 10 x 18 x C
// input halo ONCE (zero outside the image, ReLU-on-load applied at that point), and the K loop (9 taps x C) only
// streams the weights (LDS-DMA, two stages).  A fragment of tile row ty, tap (ky, kx) is halo pixel
// (ty + ky, fr + kx): 16 consecutive pixels per fragment, chunk index XOR-ed with the pixel index so the 16 lanes hit
// 16 different bank groups.  Same MFMA fragments, accumulators and epilogues as gemm_glds_kernel.
// Requirements (checked by the launcher): C * sizeof(T) a multiple of 128 (a K tile lies inside one tap), at most
// 16 chunks per pixel (C <= 128 bf16 / 64 f32), MAP_ROWS without row remapping, N >= 64.  Measured at batch 16:
// 84x148 RCU convs 118 -> 100 us (606 TFLOP/s), 42x74 45 -> 33 us, head conv1 271 -> 241 us.
// ================================================================================================
template <typename T, int BN, int WM, int WN, int NS = 2>
__global__ void __launch_bounds__(64 * WM * WN)
conv3_halo_kernel(GemmA a, const T* __restrict__ W, int M, int N, int K, int Kpad, GemmEpi e, int xn) {
    KERNARG_WARM(kaw_)                                   // all argument lines in one round trip (common.h)
    KERNARG_WARM_END(kaw_)
    constexpr int CE = Prec<T>::CE, CPR = 8, BK = CPR * CE, PD = NS - 1;
    constexpr int TW = 16, TH = 8, BM = TH * TW, HWD = TW + 2, HPX = (TH + 2) * HWD;
    constexpr int HCPP = 16;                            // capacity: 16-byte chunks per input pixel
    constexpr int NW = WM * WN, RPI = 64 / CPR;
    constexpr int BI = BN / (RPI * NW);
    static_assert(BI >= 1 && BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "bad tile split");
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int STAGE = BN * CPR;
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * STAGE + HPX * HCPP];
    D2S_POISON_LDS(lds, NS * STAGE + HPX * HCPP)
    u32x4* const halo = lds + NS * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WN, wave_n = wid % WN;
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int nimg = M / (a.Ho * a.Wo);
    int tm_, tn_;
    if (!tile_of_block(blockIdx.x, nimg * tiles_y * tiles_x, (N + BN - 1) / BN, xn, tm_, tn_)) return;
    const int b = tm_ / (tiles_y * tiles_x), ty0 = ((tm_ / tiles_x) % tiles_y) * TH, tx0 = (tm_ % tiles_x) * TW;
    const int bn0 = tn_ * BN;
    const int cpp = a.C / CE, smask = cpp - 1;          // chunks per pixel: a power of two <= 16

    // ---- weights: LDS-DMA ring, as in gemm_glds_kernel; the first NS - 1 K tiles are requested BEFORE the halo is fetched (they do not
    // depend on it: their latency sits under the halo phase -- with the up-sample folded in, four tap loads and the interpolation per chunk)
    const int lrow = wid * RPI + lane / CPR;
    const int src_chunk = (lane % CPR) ^ swz_row<CPR>(lrow);
    const T* wrow = W + (long)(bn0 + lrow) * Kpad + src_chunk * CE;
#define D2S_ISSUE_W(KT)                                                                                               \
    {                                                                                                                 \
        u32x4* st_ = lds + ((KT) % NS) * STAGE;                                                                       \
        _Pragma("unroll") for (int i = 0; i < BI; ++i)                                                                \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow + (long)(RPI * NW * i) * Kpad + (KT) * BK), \
                                             (__attribute__((address_space(3))) void*)(st_ + (i * NW + wid) * 64), 16, 0, 0);              \
    }
    const int nkt = K / BK;
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < nkt) D2S_ISSUE_W(t)
    // ---- the input halo, once (a.ups: the align_corners up-sample in front of this convolution happens here; gemm_epi.h)
    conv_halo_fill<T, 64 * NW, 3>(a, b, ty0, tx0, HWD, HPX, cpp, tid, halo,
        [&](int p, int c) { return p * cpp + (c ^ (p & smask)); },
        [&](u32x4 v) { return a.relu ? relu_frag(v, 0, T()) : v; });
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kpt = cpp / CPR;                           // K tiles per tap
    const int fr = lane & 15, fg = lane >> 4;
    int tap = 0, sub = 0;                                // kt = tap * kpt + sub
    for (int kt = 0; kt < nkt; ++kt) {
        // my W loads of tile kt (tiles kt + 1 .. kt + PD - 1 may stay in flight) and, first time, my halo stores
        if (kt + PD - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PD - 1) * BI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + PD < nkt) D2S_ISSUE_W(kt + PD)
        const int ky = tap / 3, kx = tap - ky * 3, cb = sub * CPR;
        const u32x4* B_l = lds + (kt % NS) * STAGE + (wave_n * (BN / WN)) * CPR;
#pragma unroll
        for (int ks = 0; ks < CPR / 4; ++ks) {
            u32x4 fb[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) { int r = j * 16 + fr; fb[j] = B_l[r * CPR + ((ks * 4 + fg) ^ swz_row<CPR>(r))]; }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int p = (wave_m * FM + i + ky) * HWD + fr + kx;
                const u32x4 fa = halo[p * cpp + ((cb + ks * 4 + fg) ^ (p & smask))];
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk(acc[i][j], fb[j], fa, T());
            }
        }
        if (++sub == kpt) { sub = 0; ++tap; }
    }
#undef D2S_ISSUE_W
    // ---- epilogue: tile row ty -> output pixel (ty0 + ty, tx0 + fr)
    const int x = tx0 + fr;
    EpiCols cols[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
        if (n0 < N) epi_cols_load(e, n0, cols[j]);
    }
    float pre[FM][FN][4];                               // residual values, all requested before the first store (gemm_epi.h)
    const bool pre_on = epi_res1_ahead(e);
    if (pre_on) {
        static_for<FM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int y = ty0 + wave_m * FM + i;
            static_for<FN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int n0 = bn0 + wave_n * (BN / WN) + j * 16 + fg * 4;
                if (y < a.Ho && x < a.Wo && n0 < N) epi_res1_load<T>(e, (b * a.Ho + y) * a.Wo + x, n0, pre[i][j]);
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
                    epilogue_dispatch<T>(e, m, n0, v, false, pre_on ? pre[i][j] : nullptr, &cols[j]);
                }
            });
        }
    });
}

// descriptor-addressed LDS-DMA: plain A, whole K tiles, 32-bit byte offsets
static inline bool buf_eligible(const GemmA& a, int M, int N, int K, int Kpad, int bk, size_t es) {
    return a.mode == A_PLAIN && !a.relu && K % bk == 0 && (long)M * a.lda * (long)es < (1L << 31) && (long)gemm_npad(N) * Kpad * (long)es < (1L << 31);
}

// would launch_glds split K for this launch (tiny grid, long K loop, caller-provided workspace)?  The lean instantiations take one K
// range per block (their ring is primed from preloaded arguments before e.ksplit could be read): the dispatchers send such
// launches to the general instantiations.
static inline bool splitk_wanted(const GemmEpi& e, long tiles, int K, int bk) {
    static const int sk_grid = getenv("D2S_SPLITK_GRID") ? atoi(getenv("D2S_SPLITK_GRID")) : 128;
    return e.part && e.part_elems > 0 && !e.stats_out && tiles < sk_grid && cdiv(K, bk) >= 24;     // (splitk_reduce_kernel writes no LN statistics)
}

// tile codes: 64 (64x64), 128 (128x128), 256128 / 256256 (8 waves), 25664 / 25632 (256 x 64|32, 4 waves); 0 = auto
template <typename T, int BM, int BN, int WM, int WN, int NS, int CPR = 8, int STG = 0>
static void launch_glds(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st) {
    unsigned grid = 0;
    int xn = pick_xn(cdiv(M, BM), cdiv(N, BN), BN, Kpad, sizeof(T), grid);
    // split-K for the few launches with almost no tiles but a long K loop (DPT 3x3 convs on the 11x19 /
    // 21x37 maps with 768 input channels: 14-112 blocks x 108 K tiles): the caller provides e.part
    int nkt = cdiv(K, CPR * (16 / (int)sizeof(T)));
    int ks = 1;
    static const int sk_grid = getenv("D2S_SPLITK_GRID") ? atoi(getenv("D2S_SPLITK_GRID")) : 128;       // tuning aids
    static const int sk_div = getenv("D2S_SPLITK_DIV") ? atoi(getenv("D2S_SPLITK_DIV")) : 6;
    if (STG != 2 && e.part && e.part_elems > 0 && !e.stats_out && (int)grid < sk_grid && nkt >= 24) {
        ks = nkt / sk_div; if (ks > 16) ks = 16;
        while (ks > 1 && (size_t)ks * M * N > e.part_elems) --ks;
    }
    static EnvInt sk_force{"D2S_SPLITK_FORCE", 0};          // measurement aid (tools/splitk_probe.py): this many K ranges whatever the grid
    if (sk_force.get() > 1 && STG != 2 && e.part && !e.stats_out && nkt >= sk_force.get() && (size_t)sk_force.get() * M * N <= e.part_elems) ks = sk_force.get();
    if (ks > 1) {
        GemmEpi e2 = e; e2.ksplit = ks;
        hipLaunchKernelGGL((gemm_glds_kernel<T, BM, BN, WM, WN, NS, CPR, STG>), dim3(grid, ks), dim3(64 * WM * WN), 0, st, (const T*)W, a.ptr, a.lda, M, N, K, Kpad, xn, a, e2);
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(cdiv((long)M * (N / 4), 256)), dim3(256), 0, st, e2, M, N);
        return;
    }
    GemmEpi e1 = e; e1.ksplit = 1;
    if (e.stats_slots) *e.stats_slots = WN > 1 ? cdiv(N, BN) : 1 << 20;        // WN == 1 tiles write no statistics: the caller falls back
    GemmA a1 = a;
    {   // descriptor-addressed LDS-DMA: plain A, whole K tiles (the W zero padding covers nothing then), 32-bit byte offsets
        static const bool nobuf = getenv("D2S_GEMM_NOBUF") && atoi(getenv("D2S_GEMM_NOBUF")) != 0;
        constexpr int bk = CPR * (16 / (int)sizeof(T));
        a1.buf = !nobuf && STG != 1 && buf_eligible(a, M, N, K, Kpad, bk, sizeof(T));
    }
    hipLaunchKernelGGL((gemm_glds_kernel<T, BM, BN, WM, WN, NS, CPR, STG>), dim3(grid), dim3(64 * WM * WN), 0, st, (const T*)W, a1.ptr, a1.lda, M, N, K, Kpad, xn, a1, e1);
}

// stride-1 3x3 convs on the large maps go to conv3_halo_kernel (input tile resident in LDS); D2S_NO_HALO=1 keeps the
// implicit-GEMM loader (the parity tests run both)
template <typename T>
static bool launch_conv_halo(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st, bool dry = false) {
    static EnvInt off{"D2S_NO_HALO", 0};
    if (off.get()) return false;
    if constexpr (std::is_same<T, fp8_t>::value) return false;
    else {
    const int cpp = a.C * (int)sizeof(T) / 16;
    if (a.mode != A_CONV3 || a.stride != 1 || a.Hi != a.Ho || a.Wi != a.Wo || (a.C * (int)sizeof(T)) % 128 || cpp > 16 || (cpp & (cpp - 1))) return false;
    if (e.map != MAP_ROWS || e.rows_per_img || K != 9 * a.C || N <= 32) return false;   // (the N = 32 head conv measured 20 % slower here)
    const int nimg = M / (a.Ho * a.Wo);
    const long tiles_m = (long)nimg * cdiv(a.Ho, 8) * cdiv(a.Wo, 16);
    if (tiles_m * cdiv(N, 128) < 200 || (long)nimg * a.Ho * a.Wo != M) return false;      // small maps: too few tiles, latency-bound anyway
    if (dry) return true;
    GemmEpi e1 = e; e1.ksplit = 1;
    unsigned grid = 0;
#define D2S_HALO(BN_, WM_, WN_, NS_)                                                                                  \
    { int xn = pick_xn((int)tiles_m, cdiv(N, BN_), BN_, Kpad, sizeof(T), grid);                                       \
      hipLaunchKernelGGL((conv3_halo_kernel<T, BN_, WM_, WN_, NS_>), dim3(grid), dim3(64 * WM_ * WN_), 0, st, a, (const T*)W, M, N, K, Kpad, e1, xn); }
    // (ring depth: what keeps two blocks per CU beside the 46 KB halo -- four 8 KB stages for 64 output channels, two 16 KB stages for 128)
    if (N <= 64) D2S_HALO(64, 4, 2, 4)
    else D2S_HALO(128, 2, 4, 2)
#undef D2S_HALO
    return true;
    }
}

// bf16x3 operands: the register-staged tiles only (the fp32 -> hi / lo split happens between the staging registers and LDS).
// Same rule as the other types -- resident waves first, then tile intensity -- on the staged instantiations.
static int launch_bx3(int tile, const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st) {
    typedef bx3_t T;
    static const int force_tile = getenv("D2S_GEMM_TILE_BX3") ? atoi(getenv("D2S_GEMM_TILE_BX3")) : 0;
    if (a.bx3) {
        // A pre-split by its producer (LayerNorm / attention / GELU epilogue): LDS-DMA rings for both operands, the bf16 tile rule
        if (a.mode != A_PLAIN || a.relu) { set_error("launch_gemm: a pre-split bf16x3 A operand must be a plain matrix"); return D2S_E_UNSUPPORTED; }
        static const int force_dma = getenv("D2S_GEMM_TILE_BX3D") ? atoi(getenv("D2S_GEMM_TILE_BX3D")) : 0;
        int t = force_dma;
        if (t == 0) {
            const long b128 = (long)cdiv(M, 128) * cdiv(N, 128), b64128 = (long)cdiv(M, 64) * cdiv(N, 128), b64 = (long)cdiv(M, 64) * cdiv(N, 64);
            if (b128 >= 400 && (N >= 1536 || b128 >= 900)) t = 1281288;
            else if (b64128 >= 280) t = 641288;
            else if (b64 >= 384) t = 64648;
            else t = 3264;
        }
        // (latency regime: the deep-ring lean instantiations, as in launch_t)
        static const int deep = getenv("D2S_GEMM_DEEP") ? atoi(getenv("D2S_GEMM_DEEP")) : 1;
        const bool dp = deep && buf_eligible(a, M, N, K, Kpad, 32, 4) &&
                        !splitk_wanted(e, (long)cdiv(M, t == 3264 ? 32 : 64) * cdiv(N, t == 641288 ? 128 : 64), K, 32);
        if (t == 3264 && dp && (long)cdiv(M, 32) * cdiv(N, 64) <= 512) launch_glds<T, 32, 64, 2, 2, 6, 8, 2>(a, W, M, N, K, Kpad, e, st);
        else if (t == 64648 && dp && (long)cdiv(M, 64) * cdiv(N, 64) <= 512) launch_glds<T, 64, 64, 4, 2, 4, 8, 2>(a, W, M, N, K, Kpad, e, st);
        else if (t == 641288 && dp && (long)cdiv(M, 64) * cdiv(N, 128) <= 512) launch_glds<T, 64, 128, 2, 4, 3, 8, 2>(a, W, M, N, K, Kpad, e, st);
        else if (t == 3264) launch_glds<T, 32, 64, 2, 2, 4>(a, W, M, N, K, Kpad, e, st);
        else if (t == 64648) launch_glds<T, 64, 64, 4, 2, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);
        else if (t == 641288) launch_glds<T, 64, 128, 2, 4, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);
        else if (t == 1281288) launch_glds<T, 128, 128, 2, 4, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);
        else { set_error("launch_gemm: bad tile code for pre-split bf16x3 operands"); return D2S_E_INVALID; }
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    if (tile == 0 || tile == 256256) tile = force_tile;
    if (tile == 0) {
        const long b128 = (long)cdiv(M, 128) * cdiv(N, 128), b64128 = (long)cdiv(M, 64) * cdiv(N, 128), b64 = (long)cdiv(M, 64) * cdiv(N, 64);
        if (N <= 64) tile = (long)cdiv(M, 256) >= 224 ? (N <= 32 ? 912832 : 9256648) : 93264;
        else if (b128 >= 400 && (N >= 1536 || b128 >= 900)) tile = 91288;
        else if (b64128 >= 280) tile = 964128;
        else if (b64 >= 384) tile = 964;
        else tile = 93264;
    }
    if (tile == 93264 || tile == 3264) launch_glds<T, 32, 64, 2, 2, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 964 || tile == 64) launch_glds<T, 64, 64, 2, 2, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 964128) launch_glds<T, 64, 128, 2, 4, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 91288 || tile == 128) launch_glds<T, 128, 128, 4, 2, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 912832) launch_glds<T, 128, 32, 4, 1, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);      // WN == 1: MAP_HEAD capable
    else if (tile == 9256648) launch_glds<T, 256, 64, 8, 1, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);     // WN == 1, 8 waves
    else { set_error("launch_gemm: bad tile code for bf16x3 operands"); return D2S_E_INVALID; }
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

// the (BM, BN) of the LDS-DMA tile launch_t picks for a plain linear of this shape (tile == 0 rule below, N > 64)
static void small_tile_of(int M, int N, int& bm, int& bn) {
    const long b128 = (long)cdiv(M, 128) * cdiv(N, 128), b64128 = (long)cdiv(M, 64) * cdiv(N, 128), b64 = (long)cdiv(M, 64) * cdiv(N, 64);
    if (b128 >= 400 && (N >= 1536 || b128 >= 900)) { bm = 128; bn = 128; }
    else if (b64128 >= 280) { bm = 64; bn = 128; }
    else if (b64 >= 384) { bm = 64; bn = 64; }
    else { bm = 32; bn = 64; }
}

// (Weight warm-up, measured and removed: a small kernel on its own stream pulling the NEXT linear's weight rows into the L2 of the
//  XCDs that will read them, two launches ahead -- 761-778 frames/s at batch 1 with and without.  Like the K-loop instruction
//  count (descriptor addressing: +2 %), the K-tile size (256-byte tiles: +-0) and split-K for FC2 (-5 %), cold weights are not
//  what paces the batch-1 launches.)
int gemm_pp_min_tiles() {
    static EnvInt v{"D2S_GEMM_PP", 100};                  // (re-read after d2s_debug_reload_env: the tests switch regimes inside one process)
    return v.get();
}

template <typename T>
static int launch_t(int tile, const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st) {
    static const int force_tile = getenv("D2S_GEMM_TILE") ? atoi(getenv("D2S_GEMM_TILE")) : 0;
    if (tile == 0) tile = force_tile;
    if (tile == 0 && launch_conv_halo<T>(a, W, M, N, K, Kpad, e, st)) { D2S_CHECK_LAUNCH(); return D2S_OK; }
    static EnvInt conv_tile{"D2S_CONV_TILE", 0};          // tuning aid: tile code for the implicit 3x3 convolutions
    if (tile == 0 && a.mode == A_CONV3 && N > 64) tile = conv_tile.get();
    if constexpr (std::is_same<T, bf16_t>::value) {
        // thin linears (K <= 256: ConvTranspose(k = s), fusion 1x1 projections): HBM-bound, one prologue per block instead of per tile
        if (tile == 0 && sk_supported(D2S_PREC_BF16, a, M, N, K, Kpad, e)) return launch_gemm_sk(a, W, M, N, K, Kpad, e, st);
    }
    if constexpr (!std::is_same<T, float>::value) {
        // batched plain linears: the 256 x 256 ping-pong kernel (gemm_pp.hip) once the launch has enough tiles to fill the chip
        // measured (tools/pp_check.py, ViT-B shapes): from ~140 tiles of 256 x 256 the ping-pong kernel wins every encoder
        // linear (batch 16: +14..+30 %, batch 32: +9..+48 %); at 75 tiles (batch 8, N = 768) it loses.  D2S_GEMM_PP=0: off
        const int pp_min_tiles = gemm_pp_min_tiles();                                       // (100 vs 140: +5 % at batch 12, proj / FC2 there)
        const int prec = std::is_same<T, bf16_t>::value ? D2S_PREC_BF16 : D2S_PREC_FP8_OPERANDS;
        if (tile == 0 && pp_min_tiles > 0 && (long)cdiv(M, 256) * cdiv(N, 256) >= pp_min_tiles && pp_supported(prec, a, M, N, K, Kpad, e))
            return launch_gemm_pp(prec, a, W, M, N, K, Kpad, e, st);
        // (Tile rounding, batch 32: 294 tiles of N = 768 pay a second round for 38 tiles.  Giving the ping-pong kernel only the
        //  tile rows that fill whole rounds and the remaining 3136 rows to the small-tile kernel -- two launches, disjoint rows --
        //  was built and measured: FC2 156 -> 148 us, proj unchanged; the small-tile kernel needs as long for those rows as the
        //  half-empty round.  Removed.)
    }
    if (tile == 0) {
        // Measured on the ViT-B shapes at batch 1..32 (tools/gemm_bench.py, profiles/r1_05): what matters most is 16-24
        // resident waves per CU in DIFFERENT phases of the K loop (8-wave blocks, 2-5 blocks per CU), then tile intensity;
        // with that in place the LDS-DMA path beats register staging (no VGPR / ds_write pass).  The 4-wave 32 x 64 ring
        // (NS = 4: deepest prefetch, shortest prologue) keeps the launches with the fewest tiles (batch 1: proj, FC2).
        static const int t64 = getenv("D2S_GEMM_T64") ? atoi(getenv("D2S_GEMM_T64")) : 0;    // tuning: N <= 64 tiles
        static const int t32 = getenv("D2S_GEMM_T32") ? atoi(getenv("D2S_GEMM_T32")) : 0;
        const long b128 = (long)cdiv(M, 128) * cdiv(N, 128), b64128 = (long)cdiv(M, 64) * cdiv(N, 128), b64 = (long)cdiv(M, 64) * cdiv(N, 64);
        if (N <= 64) {                              // DPT head: 64 / 32 output channels, M = pixels
            if ((long)cdiv(M, 256) >= 224) tile = N <= 32 ? (t32 ? t32 : 912832) : (t64 ? t64 : 9256648);
            else tile = 3264;
        }
        else {
            int bm = 0, bn = 0;
            small_tile_of(M, N, bm, bn);
            tile = bm == 128 ? 1281288 : (bn == 128 ? 641288 : (bm == 64 ? 64648 : 3264));     // 3264: skinny launches (batch 1, N = 768)
            // long-K implicit convolutions (tap 3's stride-2 768 -> 768: K = 6 912) from ~300 tiles of 128 x 128: the 64 x 128 tile falls off a
            // cliff there (batch 28: 130 us, batch 32: 188 us for 1.14 x the work) where the 128 x 128 tile stays at 128-134 us at every batch from
            // 16 to 32 -- below 300 tiles the smaller tile is 5-15 % faster.  D2S_CONV_B128: the threshold
            static EnvInt conv_b128{"D2S_CONV_B128", 300};
            if (a.mode == A_CONV3 && K >= 4096 && (long)cdiv(M, 128) * cdiv(N, 128) >= conv_b128.get()) tile = 1281288;
        }
    }
    // Latency regime (batch 1-2: every block of the launch is resident at once, 1-2 per CU).  In-kernel stamps (tools/glds_timeline.py)
    // put a K tile at (LDS-DMA latency ~ 1 800 cycles) / (tiles in flight): 615 cycles with the 4-stage ring of the 32 x 64 tile, 1 100-1 300
    // with the 2-stage rings of the 8-wave tiles, next to 64-256 cycles of MFMA work.  The LDS those few blocks leave unused buys
    // ring depth: as many stages as still let ALL blocks be resident.  "Lean" instantiations (descriptor loader only).
    if constexpr (std::is_same<T, bf16_t>::value || std::is_same<T, fp8_t>::value) {
        static const int deep = getenv("D2S_GEMM_DEEP") ? atoi(getenv("D2S_GEMM_DEEP")) : 1;
        static const bool nobuf = getenv("D2S_GEMM_NOBUF") && atoi(getenv("D2S_GEMM_NOBUF")) != 0;
        if (deep && !nobuf && !(e.part && e.ksplit > 1) && buf_eligible(a, M, N, K, Kpad, 128 / (int)sizeof(T), sizeof(T)) && e.map != MAP_HEAD &&
            !splitk_wanted(e, (long)cdiv(M, tile == 3264 ? 32 : 64) * cdiv(N, tile == 641288 ? 128 : 64), K, 128 / (int)sizeof(T))) {
            bool done = true;
            if (tile == 3264 && (long)cdiv(M, 32) * cdiv(N, 64) <= 512) launch_glds<T, 32, 64, 2, 2, 6, 8, 2>(a, W, M, N, K, Kpad, e, st);            // 72 KiB: 2 blocks / CU
            else if (tile == 64648 && (long)cdiv(M, 64) * cdiv(N, 64) <= 512) launch_glds<T, 64, 64, 4, 2, 4, 8, 2>(a, W, M, N, K, Kpad, e, st);       // 64 KiB: 2 blocks / CU
            else if (tile == 641288 && (long)cdiv(M, 64) * cdiv(N, 128) <= 512) launch_glds<T, 64, 128, 2, 4, 3, 8, 2>(a, W, M, N, K, Kpad, e, st);    // 72 KiB: 2 blocks / CU
            // (one block per CU with a 128 KiB ring -- 64 x 192 x 4 stages, 128 x 128 on 64-byte K tiles x 4 stages -- for FC1, whose 312
            //  blocks of 64 x 128 leave 56 CUs with two blocks: 13.5 / 15.6 us against 13.4 back to back; a single block does not reach
            //  the fill rate two reach together.  Measured, removed.)
            else done = false;
            if (done) { D2S_CHECK_LAUNCH(); return D2S_OK; }
        }
    }
    // (The same deeper rings for the implicit 3x3 convolutions of the small DPT maps at batch 1 -- general loaders, 32 x 64 x 6 / 64 x 64 x 4 /
    //  64 x 128 x 3 stages when every block is resident: 866-873 frames/s with and without, ViT-S 1 200 vs 1 227.  Measured, removed.)
    // LDS-DMA ring (NS stages)
    if (tile == 256128) launch_glds<T, 256, 128, 4, 2, 3>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 128) launch_glds<T, 128, 128, 2, 2, 4>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 64) launch_glds<T, 64, 64, 2, 2, 4>(a, W, M, N, K, Kpad, e, st);
    // (256-byte K tiles for this tile -- half the barrier-paced iterations -- measured at batch 1: 768 vs 766 frames/s with three
    //  ring stages, 726 with two: the batch-1 launches are not paced by their K-loop iteration count)
    else if (tile == 3264) launch_glds<T, 32, 64, 2, 2, 4>(a, W, M, N, K, Kpad, e, st);
    // (One-round shapes for M = 778 -- 48 x 64, 96 x 64, 48 / 96 / 80 / 112 x 128, 112 x 96: 108-240 blocks of 2-4 waves, fewer
    //  fill bytes per CU than two rounds of a small tile -- were instantiated and swept at batch 1: 1.2-4 x SLOWER than the
    //  32 x 64 / 64 x 64 / 64 x 128 tiles on every encoder linear.  Resident waves per CU decide the fill rate a CU reaches.)
    // LDS-DMA, 8 waves, two stages: the winners of the second sweep (profiles/r1_05), code <BM><BN>8
    else if (tile == 1281288) launch_glds<T, 128, 128, 2, 4, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);    // wave tile 64 x 32, 2 blocks / CU
    else if (tile == 641288) launch_glds<T, 64, 128, 2, 4, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);      // 3 blocks / CU
    else if (tile == 64648) launch_glds<T, 64, 64, 4, 2, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);        // 5 blocks / CU
    else if (tile == 256648) launch_glds<T, 256, 64, 8, 1, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);      // WN == 1: MAP_HEAD capable
    else if (tile == 128324) launch_glds<T, 128, 32, 4, 1, 2, 8, 0>(a, W, M, N, K, Kpad, e, st);      // WN == 1, 4 waves
    // register-staged variants (STG), code = 9 <BM> <BN> [waves].  Removed after losing the sweeps: 256 x 128 / 256 x 256
    // with 8 waves (1 block / CU, lock-step), 128 x 128 with 4 or 16 waves, deeper rings, 64- and 256-byte K tiles, two
    // register sets, intra-block split-K
    else if (tile == 964) launch_glds<T, 64, 64, 2, 2, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);
    else if (tile == 91288) launch_glds<T, 128, 128, 4, 2, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);       // 8 waves, 2 blocks / CU
    else if (tile == 912832) launch_glds<T, 128, 32, 4, 1, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);       // WN == 1: MAP_HEAD capable
    else if (tile == 9256648) launch_glds<T, 256, 64, 8, 1, 2, 8, 1>(a, W, M, N, K, Kpad, e, st);      // WN == 1, 8 waves
    else { set_error("launch_gemm: bad tile code"); return D2S_E_INVALID; }
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

// Would launch_gemm fold the up-sample described by a.ups / Hs / Ws / usy / usx into this convolution's loader?  (The engine asks
// before it skips the stand-alone up-sample launch.)  Same order of kernels as launch_gemm below.
bool conv3_upsample_ok(int precision, int tile, const GemmA& a, int M, int N, int K, int Kpad, const GemmEpi& e) {
    static EnvInt no_ups{"D2S_NO_UPSFOLD", 0};
    if (no_ups.get() || precision != D2S_PREC_BF16 || a.mode != A_CONV3 || !a.ups || a.Hs < 2 || a.Ws < 2 || a.usy <= 0.f || a.usx <= 0.f) return false;
    if ((tile == 0 || e.map == MAP_HEAD) && launch_conv3_halo2(a, nullptr, M, N, K, Kpad, e, nullptr, true)) return true;
    return tile == 0 && launch_conv_halo<bf16_t>(a, nullptr, M, N, K, Kpad, e, nullptr, true);
}

int launch_gemm(int precision, int tile, const GemmA& a, const void* W, int M, int N, int K, int Kpad,
                const GemmEpi& e, hipStream_t st) {
    const int ce = 16 / (int)elem_size(precision);
    if (M <= 0 || N <= 0 || K <= 0 || (N & 3) || (K % ce) || Kpad % (2 * gemm_bk(precision))) {
        set_error("launch_gemm: bad dims (N % 4, K % chunk, Kpad % BK)"); return D2S_E_INVALID;
    }
    if (a.mode == A_PLAIN && (a.lda % ce)) { set_error("launch_gemm: lda not chunk aligned"); return D2S_E_INVALID; }
    if (a.mode == A_CONV3 && (a.C % ce)) { set_error("launch_gemm: conv channels not chunk aligned"); return D2S_E_INVALID; }
    if (precision == D2S_PREC_BF16X3) {
        if (e.deq || (e.out2 && !e.out2_bx3)) { set_error("launch_gemm: bf16x3 operands: no de-quantisation, raw-residual copies in the unit format only"); return D2S_E_UNSUPPORTED; }
        return launch_bx3(tile, a, W, M, N, K, Kpad, e, st);
    }
    if (tile == 256256) return launch_gemm_pp(precision, a, W, M, N, K, Kpad, e, st);      // the ping-pong kernel, forced (tests / sweeps)
    // batched 3x3 convolutions (and the fused head, whose caller names a tile): input tile resident in LDS, conv3.hip
    if (precision == D2S_PREC_BF16 && a.mode == A_CONV3 && (tile == 0 || e.map == MAP_HEAD) && launch_conv3_halo2(a, W, M, N, K, Kpad, e, st)) {
        D2S_CHECK_LAUNCH();
        return D2S_OK;
    }
    if (a.ups) {        // only the LDS-resident-input kernels fold the up-sample (conv3_upsample_ok says when)
        if (precision == D2S_PREC_BF16 && tile == 0 && launch_conv_halo<bf16_t>(a, W, M, N, K, Kpad, e, st)) { D2S_CHECK_LAUNCH(); return D2S_OK; }
        set_error("launch_gemm: this launch cannot fold the up-sample into its loader (conv3_upsample_ok says when)");
        return D2S_E_UNSUPPORTED;
    }
    if (precision == D2S_PREC_BF16) return launch_t<bf16_t>(tile, a, W, M, N, K, Kpad, e, st);
    if (precision == D2S_PREC_FP8_OPERANDS) {
        if (a.mode != A_PLAIN || a.relu) { set_error("launch_gemm: e4m3 operands are for plain linears"); return D2S_E_UNSUPPORTED; }
        return launch_t<fp8_t>(tile, a, W, M, N, K, Kpad, e, st);
    }
    return launch_t<float>(tile, a, W, M, N, K, Kpad, e, st);
}

// ---- test / micro-benchmark probe -----------------------------------------------------------------
__global__ void cast_pad_kernel(const float* __restrict__ src, void* __restrict__ dst, int rows, int cols,
                                int rows_pad, int cols_pad, int prec) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows_pad * cols_pad) return;
    int c = (int)(idx % cols_pad), r = (int)(idx / cols_pad);
    float v = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    if (prec == -D2S_PREC_BF16X3) {                  // a bf16x3 WEIGHT matrix: unit format (the A operand stays fp32: prec == D2S_PREC_BF16X3 below)
        const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
        bf16_t* u = (bf16_t*)((char*)dst + ((long)r * cols_pad + (c & ~7)) * 4) + (c & 7);
        u[0] = hi; u[8] = lo;
    } else
    if (prec == D2S_PREC_BF16) ((bf16_t*)dst)[idx] = f2bf(v);
    else if (prec == D2S_PREC_FP8_OPERANDS) ((fp8_t*)dst)[idx] = f2e4m3(v);      // unit scales: the caller keeps |v| <= 448
    else ((float*)dst)[idx] = v;
}

}  // namespace d2s

using namespace d2s;

extern "C" int d2s_gemm_probe(const float* A, const float* Wt, const float* bias, float* Cout, int M, int N, int K,
                              int precision, int tile, int iters, void* stream) {
    D2S_REQUIRE(A && Wt && Cout && M > 0 && N > 0 && K > 0 && (N % 4 == 0) && iters >= 1, "bad argument");
    D2S_REQUIRE(precision == D2S_PREC_BF16 || precision == D2S_PREC_FP32 || precision == D2S_PREC_FP8_OPERANDS || precision == D2S_PREC_BF16X3, "bad precision");
    hipStream_t st = (hipStream_t)stream;
    int bf = precision;
    int Kp = gemm_kpad(K, precision), Np = gemm_npad(N);
    size_t es = elem_size(precision);
    void *dA = nullptr, *dW = nullptr;
    D2S_HIP(hipMalloc(&dA, (size_t)M * Kp * es));
    D2S_HIP(hipMalloc(&dW, (size_t)Np * Kp * es));
    hipLaunchKernelGGL(cast_pad_kernel, dim3(cdiv((long)M * Kp, 256)), dim3(256), 0, st, A, dA, M, K, M, Kp, bf);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(cdiv((long)Np * Kp, 256)), dim3(256), 0, st, Wt, dW, N, K, Np, Kp, bf == D2S_PREC_BF16X3 ? -bf : bf);
    GemmA a = {};
    a.ptr = dA; a.mode = A_PLAIN; a.lda = Kp;
    GemmEpi e = {};
    e.out = Cout; e.out_type = OUT_F32; e.ldc = N; e.bias = bias;
    void* dP = nullptr;
    static EnvInt sk_force{"D2S_SPLITK_FORCE", 0};          // measurement aid: give the launch a split-K workspace
    if (sk_force.get() > 1) {
        const size_t pe = (size_t)sk_force.get() * M * N;
        D2S_HIP(hipMalloc(&dP, pe * sizeof(float) + GEMM_PART_CTR_WORDS * 4));
        e.part = (float*)dP; e.part_elems = pe;
    }
    int rc = D2S_OK;
    for (int i = 0; i < iters && rc == D2S_OK; ++i) rc = launch_gemm(precision, tile, a, dW, M, N, Kp, Kp, e, st);
    hipError_t err = hipStreamSynchronize(st);
    (void)hipFree(dA); (void)hipFree(dW); if (dP) (void)hipFree(dP);
    if (rc != D2S_OK) return rc;
    D2S_HIP(err);
    return D2S_OK;
}

#ifdef D2S_GLDS_TIMING
extern "C" int d2s_glds_timing(unsigned long long* out, int clear) {      // out != null: read 4096 x 8 stamps
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d2s::glds_timing), sizeof(unsigned long long) * 4096 * 8) != hipSuccess) return 1;
    if (clear) { static unsigned long long zeros[4096 * 8]; return hipMemcpyToSymbol(HIP_SYMBOL(d2s::glds_timing), zeros, sizeof(zeros)) != hipSuccess; }
    return 0;
}
#endif
