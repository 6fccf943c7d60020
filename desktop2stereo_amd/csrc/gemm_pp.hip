// Ping-pong MFMA GEMM for the batched encoder linears (gfx950):  C[m,n] = epi(sum_k A[m,k] W[n,k]),  256 x 256 block tile,
// one persistent 8-wave block per CU.
//
// Why a second kernel.  gemm_glds_kernel's 128 x 128 tiles pull (128 + 128) x 128 B through L2 -> LDS per 2.1 MFLOP; at the
// ~14.6 TB/s this chip's LDS-DMA path sustains that caps the batched linears near 950 TFLOP/s and they reach 600-880
// (profiles/r1_07).  A 256 x 256 tile halves the fill bytes per flop, but one 8- or 16-wave block per CU in lock-step
// (every wave loading, then every wave computing) lost to the smaller tiles (r1_05).  Here the eight waves of the block are
// two groups of four -- one wave of each group per SIMD -- that run the SAME instruction stream one barrier apart:
//
//     group 0:  MEM(p)   | MFMA(p)  | MEM(p+1) | MFMA(p+1) | ...
//     group 1:  (wait)   | MEM(p)   | MFMA(p)  | MEM(p+1)  | ...          ( | = s_barrier )
//
// so at any moment each SIMD has one wave in a matrix segment (8 x v_mfma_f32_32x32x16, s_setprio 1) and its partner in a
// memory segment (ds_read_b128 fragment reads + LDS-DMA issue for a later K tile): the matrix pipe and the LDS / vector-memory
// pipes overlap by construction instead of by luck.
//
// Tile geometry (bf16; e4m3 is the same bytes with two MFMAs per 16-byte chunk):
//   block 256 (M) x 256 (N), K tile = 128 bytes per row; 8 waves = 2 (M halves = the two groups) x 4 (N quarters);
//   wave tile 128 x 64 = 4 x 2 blocks of 32 x 32 -> 128 accumulator VGPRs; a K tile is 4 phases, one 64 x 32 quadrant each:
//   (m0,n0) (m0,n1) (m1,n1) (m1,n0) -- a quadrant switch replaces only one operand's registers.
//   LDS: 2 stages x (256 A rows + 256 W rows) x 128 B = 128 KiB, rows XOR-swizzled as in gemm.hip (conflict-free b128 reads
//   for 32-row fragments too: the 16-lane read groups see XOR values {0,1,6,7,2,3,4,5} / {2,3,4,5,0,1,6,7}), plus eight
//   wave-private 4 KiB epilogue patches = the CU's 160 KiB.
// Schedule of tile t (parity p = t & 1), per wave.  I = LDS-DMA issue (one "piece" = 128 rows of one operand = 2 x
// buffer_load_dwordx4 ... lds per wave), R = fragment reads, C = the 8 MFMAs:
//     P1:  I W_n1[t+1]   R fw[!p] <- W(n1)[t],     Y <- A(m0)[t].mb1       C (m0,n0): X|Y, fw[p]
//     P2:  I A_m1[t+1]   R Z      <- A(m1)[t].mb0                            C (m0,n1): X|Y, fw[!p]
//     P3:  I A_m0[t+2]   R Y      <- A(m1)[t].mb1                            C (m1,n1): Z|Y, fw[!p]
//     P4:  I W_n0[t+2]   R fw[!p] <- W(n0)[t+1],   X <- A(m0)[t+1].mb0     C (m1,n0): Z|Y, fw[p]
//   (the two W register sets swap roles every tile: the set that held n1 of tile t is free after P3 and takes n0 of tile t+1;
//    A lives in three 4-chunk sets: X / Z hold the first 32-row block of the m0 / m1 quadrant, Y the second block of both)
//   * W fragments and the FIRST block of an A quadrant are read ONE PHASE BEFORE the MFMAs that use them (into registers the
//     running MFMAs do not touch); the second block is read in its own phase and consumed by the LAST four MFMAs of the
//     segment (the first four run on the pre-read block: 128 cycles of cover).  LDS latency therefore sits under matrix
//     work instead of in front of it (first version, all reads in the phase of use: 1.53 us per K tile = 48 % MFMA busy);
//     a full pre-read of both blocks needs 16 more VGPRs than the 256 there are (spills inside the K loop);
//   * every piece is requested exactly 4 phases (8 barrier slots, > 2000 cycles) before the segment that reads it, in the
//     order it is consumed, and overwrites the piece read 4 phases earlier;
//   * RAW: a piece read in MEM(q) is covered by s_waitcnt vmcnt(6) at the END of MEM(q-1) of every wave (6 = the three
//     younger pieces may stay in flight) and the barrier(s) both groups pass before MEM(q) (MI355X_MICROARCH.md "Two waves
//     per SIMD" item 7; cdna_hip_programming.md 8-phase template);  WAR: a piece is overwritten >= 3 phases after its last read.
// The last two K tiles run the same phases without issuing (waits 4 / 2 / 0).  K tiles per block must be even (fw0 parity).
// Persistent loop: the block walks its XCD's run of the tile sequence (pp_run / pp_tile_of: column group by column group,
// m-slow / n-fast), CU c of the XCD taking tiles c, c + cpx, ...  -- at any moment an XCD's CUs work on cpx consecutive tiles, which
// share A row panels and W column panels in that XCD's L2.  The next tile's first twelve pieces are issued INSIDE the
// epilogue of the current one, and the tile starts on a counted vmcnt that lets the epilogue's own stores stay in flight
// (vmcnt retires in order on gfx9).
// Tile rounding: see the unit lists in the kernel (K-split tail units + pp_tail_reduce_kernel for long-K residual launches).
// (Tried and removed, measured on MI355X: (a) stream-K -- equal K-tile ranges per CU, partial tiles exchanged through fp32
// slabs with agent-scope flags: correct and deterministic, but the slab round trip and the 2-3x wider spread of an XCD's
// CUs over the tile walk cost more than the rounding of tiles / CUs saves on these shapes (QKV at batch 32: 848 -> 697
// TFLOP/s; 8192^3 with every tile in one contiguous run per CU: 1417 -> 932); (b) global_atomic_add_f32 for the in-place
// fp32 residual of proj / FC2: 19 M atomics per launch run at 0.3 TB/s, 296 us instead of 89.)
// Epilogue: through the wave-private LDS patch so that global memory sees whole 128-byte lines (the MFMA layout leaves a
// lane with 4 consecutive n of ONE row: 32 rows x 16-32 B per store, measured 1.8 TB/s).
#include "gemm_epi.h"
#include <algorithm>
#include <mutex>
#include <vector>

namespace d2s {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4 (mma64 / PPFrag<true> below): the fragment chunks become 8-register operand
// tuples and a quadrant is 4 MFMAs of 64 cycles instead of 16 of 32 (twice the e4m3 rate of v_mfma_f32_32x32x16_fp8_fp8, which
// issues at the bf16 rate).  History: built in round 3 and left off because "the last two K tiles of every tile spill ~90 dwords
// around the tuples" (batch 32 fp8: 2 820 vs 3 000 frames/s).  Round 5 read the ISA: nothing to do with the tuples -- the results of
// the last two K tiles are only read by the epilogue, and the compiler SANK their 32 MFMAs below the phase barriers into the
// epilogue's block (operands spilled on the way); one empty asm per quadrant pins them in place (PP_MFMA): 0 spills in kinds 0-2,
// QKV 955 -> 1 112, FC1 888 -> 1 006, proj 643 -> 768, FC2 1 042 -> 1 375 TFLOP/s at batch 32 (tools/pp_check.py --prec fp8 --bench),
// same bits.  On by default; -DPP_FP8_K64=0 builds the non-scaled path.
#ifndef PP_FP8_K64
#define PP_FP8_K64 1
#endif
#ifndef PP_CV_DEQ
#define PP_CV_DEQ 1                            // (bisecting aid) e4m3 residual launches keep their column vectors in LDS
#endif
constexpr int PP_TAIL_MAX = GEMM_PART_CTR_WORDS / 4;   // tail tiles an in-kernel tail reduce can count (4 counter words each, BEHIND the part_elems partials: gemm.h)
constexpr int PP_AUX_SC0_SC1 = 0x11;          // buffer-instruction cache policy on gfx950: bit 0 = sc0, bit 4 = sc1 (write-through / system scope)
constexpr int PP_STAGE = 4096;                 // 16-byte chunks per stage: (256 + 256) rows x 8 chunks
constexpr int PP_WOFF = 2048;                  // W rows start after the 256 A rows

__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& w, const u32x4& a, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&w, *(const bf16x8*)&a, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& w, const u32x4& a, fp8_t) {
    const long* wl = (const long*)&w;
    const long* al = (const long*)&a;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wl[0], al[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wl[1], al[1], acc, 0, 0, 0);
}
// e4m3: two 16-byte chunks of each operand (32 K elements per lane, 64 per instruction) on the scaled K=64 MFMA with unit
// block scales (E8M0 127 = 2^0) -- twice the rate of v_mfma_f32_32x32x16_fp8_fp8: 4 instructions x 64 cycles per quadrant
// instead of 16 x 32.  Which K elements a lane holds does not matter as long as both operands hold the same ones.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma64(f32x16& acc, const i32x8& w, const i32x8& a) {
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, a, acc, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
// The four 16-byte fragment chunks of one operand block: four separate registers quads, or (K = 64 MFMA) two 8-register
// operand tuples whose halves the ds_read_b128s write directly.
template <bool K64> struct PPFrag;
template <> struct PPFrag<false> {
    u32x4 c[4];
    __device__ __forceinline__ void set(int ks, const u32x4& x) { c[ks] = x; }
};
template <> struct PPFrag<true> {
    i32x8 t[2];
    __device__ __forceinline__ void set(int ks, const u32x4& x) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 xi = __builtin_bit_cast(i32x4, x);
        const i32x8 w = __builtin_shufflevector(xi, xi, 0, 1, 2, 3, 0, 1, 2, 3);
        if (ks & 1) t[ks >> 1] = __builtin_shufflevector(t[ks >> 1], w, 0, 1, 2, 3, 12, 13, 14, 15);
        else t[ks >> 1] = __builtin_shufflevector(t[ks >> 1], w, 8, 9, 10, 11, 4, 5, 6, 7);
    }
};

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// tuning aid (build with D2S_HIPCC_DEFS=-DD2S_PP_TIMING): wave 0 of every block stamps the 100 MHz wall clock at kernel entry
// and, per tile, at main-loop start / main-loop end / epilogue end / next tile's operands landed; tools/pp_timeline.py reads them
#ifdef D2S_PP_TIMING
__device__ unsigned long long pp_timing[(256 + 8) * 64];              // rows 256..263: the eight waves of block 8
__device__ int pp_timing_kind = -1;
#define PP_STAMP(SLOT) { if (tid == 0 && KIND == pp_timing_kind && (SLOT) < 64) pp_timing[blockIdx.x * 64 + (SLOT)] = wall_clock64(); }
#define PP_WSTAMP(SLOT) { if (lane == 0 && blockIdx.x == 8 && KIND == pp_timing_kind && (SLOT) < 64) pp_timing[(256 + wid) * 64 + (SLOT)] = wall_clock64(); }
#else
#define PP_STAMP(SLOT)
#define PP_WSTAMP(SLOT)
#endif

// (gelu_erf2: gemm_epi.h -- every kernel's GELU epilogue evaluates the same packed expression)

// Epilogue modes.  Every mode issues at least PP_TAIL vector-memory instructions per wave after its hook (rows past M fall
// outside the output buffer's num_records: the store is dropped but still issues), because the caller counts them in vmcnt.
enum { PP_EP_BF16 = 0, PP_EP_F32 = 1, PP_EP_VT = 3 };
// lower bound of the vector-memory instructions every epilogue issues AFTER its hook (the point where the next segment's
// LDS-DMA is issued): the bf16 epilogue's 16 stores (the others issue 32 or more)
constexpr int PP_TAIL = 16;
constexpr int PP_MAXN = 16384;                 // widest N (length of the stand-in column vectors)
constexpr int PP_RING = 1;                      // passes of fp32 residuals in flight (16 VGPRs each; 2: 19 spilled registers, epilogue 25.2 -> 27.4 us, round 5)

// (absent bias / LayerScale / de-quantisation vectors are replaced by constant vectors of zeros / ones on the host: a null
//  check per load would put a branch around every one of them)
__device__ __forceinline__ f32x4 pp_col4(const float* p, int n0) { return *(const f32x4*)(p + n0); }
template <int ACT>
__device__ __forceinline__ f32x4 pp_act4(f32x4 v) {
    if constexpr (ACT == ACT_GELU) {
        const f32x2v a = gelu_erf2((f32x2v){v[0], v[1]}), b = gelu_erf2((f32x2v){v[2], v[3]});
        return (f32x4){a[0], a[1], b[0], b[1]};
    } else if constexpr (ACT == ACT_RELU) {
        return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    } else return v;
}
#define PP_PIN4(X) asm volatile("" : "+v"(X))       // the value is complete (and opaque) from here on

// One wave's 128 x 64 tile -> global memory.  `hook()` is called exactly once, at the point after which at least PP_TAIL
// vector-memory instructions follow.
//   PP_EP_BF16   32 rows x 64 columns per pass (4 passes): ds_write_b64 -> ds_read_b128 -> global_store_dwordx4 (8 rows x 128 B)
//   PP_EP_F32    32 rows x 32 columns per pass (8 passes, column half outer): ds_write_b128 -> ds_read_b128 -> (+ residuals)
//                -> store; the residual loads (the in-place fp32 residual stream of proj / FC2) run four passes ahead
//   PP_EP_VT     the V third of a QKV launch goes transposed to vt[b, head, d, token]; the MFMA layout already has
//                consecutive tokens in consecutive lanes, so it is stored directly (2-byte stores, 64 B per 32 lanes)
// Memory order is what the time goes to here, not arithmetic: vmcnt retires in order on gfx9, so a load issued after a
// store waits for that store's round trip to HBM, and a load issued after the hook waits for the next tile's operands.
// (Measured with the in-kernel stamps, batch 32: with the column vectors fetched inside every pass the epilogue of a tile
// took 9 us for bf16 output, 14-19 us with GELU or the fp32 residual -- as long as the 12-K-tile main loop.)  So every
// column vector (de-quantisation, bias, LayerScale) and the first four passes of residuals are requested BEFORE the hook
// and pinned (one exposed L2 latency per tile); after the hook only stores and the ring's refills are issued.
// LDS patch rows are 128 bytes = 8 chunks, chunk index XOR (row & 7) on both sides.  Wave-private: no barrier, the LDS ops
// of one wave execute in order.
// LN = LayerNorm folded into the linears either side of it, here for the batched regime:
//   producer (PP_EP_F32 with the residual): the pass order becomes row block outer / column half inner, so that a lane meets both
//     halves of its rows back to back; it also stores the value as bf16 (out2: the raw residual the next linear reads) and adds up
//     (sum x, sum x^2) of its 4 columns per row; the 8 lanes of a row combine by ds_swizzle, the block's four N-quarter waves
//     through `lnred` -- LDS the next tile's prologue does not touch until its second phase -- behind ONE extra block barrier,
//     in fixed order (bit-reproducible): one float2 per (row, 256-column tile) in stats_out[tile column][M];
//   consumer (PP_EP_BF16 / PP_EP_VT): A is that raw bf16 residual, W is gamma-folded; v = rstd[m] (acc - mean[m] colsum[n]) + bias
//     with mean / rstd from the <= 4 partials per row (the two K halves of a lane pair load two slots each).
template <int MODE, int ACT, bool DEQ, bool RES, bool IDENT = false, bool LN = false, bool OUT8 = false /* PP_EP_BF16 only: e4m3 output (v * out_qscale), 64-byte row segments */, typename HOOK>
__device__ __forceinline__ void pp_epilogue(f32x16 (&acc)[4][2], const GemmEpi& e, int bm0, int bn0, int M /* rows that exist FOR THIS UNIT: a row-split tail unit clips at the end of its slice */,
                                            int Mfull /* rows of the matrix (leading dimension of the statistics) */, int grp, int wn,
                                            int lane, u32x4* stg, float2* lnred, float* colv, HOOK&& hook,
                                            bool skip_dead = false /* LN producer, LAST unit of a block only (nothing counts its stores afterwards): row blocks past M are skipped */) {
#define PP_QUAD(I, J, Q4) (f32x4){acc[I][J][4 * (Q4) + 0], acc[I][J][4 * (Q4) + 1], acc[I][J][4 * (Q4) + 2], acc[I][J][4 * (Q4) + 3]}
    const int fl = lane & 31, kg = lane >> 5;
    const int rr = lane >> 3, rc = lane & 7;
    const int wm0 = bm0 + grp * 128, wn0 = bn0 + wn * 64;
    // quad (j, q4) of this lane = columns wn0 + j * 32 + 8 * q4 + 4 * kg .. + 3
    if constexpr (MODE == PP_EP_F32) {
        // out = res + scale * (deq * acc + bias) = res + ca * acc + cc  with  ca = scale * deq,  cc = scale * bias
        // CV: the two column vectors live in LDS (`colv`) instead of 64 pinned VGPRs -- the LN producer, and every e4m3 launch
        // (kind 3 with DEQ spilled 86 registers around its three epilogue variants)
        constexpr bool CV = LN || (DEQ && !IDENT && PP_CV_DEQ);
        f32x4 ca[2][4], cc[2][4];                    // per column half j
        f32x4 res[PP_RING][4];                       // residual prefetch ring: pass p = j * 4 + i
        // raw buffers over [M][ldc]: one per-lane byte offset serves every row of every pass (+ a wave-uniform term), and
        // rows past M fall outside num_records -- their loads return 0 and their stores are dropped, yet the instruction
        // still issues, which keeps the vmcnt accounting fixed without clamping rows
        const unsigned ldc = (unsigned)e.ldc, nrec = (unsigned)M * ldc * 4u;
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)e.res1, 0, RES ? nrec : 0u, 0x00020000);
        const unsigned vo = ((unsigned)(wm0 + rr) * ldc + (unsigned)(wn0 + rc * 4)) * 4u;
        const __amdgpu_buffer_rsrc_t rsO2 = __builtin_amdgcn_make_buffer_rsrc(LN ? e.out2 : e.out, 0, LN ? (unsigned)M * ldc * 2u : 0u, 0x00020000);
        float s1[4], s2[4];                          // LN producer: this lane's share of (sum, sum of squares) of its 4 rows of block i
        auto col_load = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = wn0 + j * 32 + 8 * q + 4 * kg;
                const f32x4 sc = pp_col4(e.scale, n0);
                cc[j][q] = pp_col4(e.bias, n0) * sc;
                if constexpr (DEQ) ca[j][q] = pp_col4(e.deq, n0) * sc; else ca[j][q] = sc;
            }
        };
        auto res_load = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            constexpr int j = LN ? (p & 1) : (p >> 2), i = LN ? (p >> 1) : (p & 3);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                res[p % PP_RING][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo + ((i * 32 + r * 8) * ldc + j * 32) * 4u, 0, 0));
        };
        // IDENT: raw accumulators (the partial sums of a K split go to their slab untouched)
        if constexpr (!IDENT && !CV) col_load(std::integral_constant<int, 0>{});
        if constexpr (CV) {                          // ca | cc of the wave's 64 columns -> colv[0..63] | colv[64..127]
            if (lane < 32) {
                const int c = (lane & 15) * 4;
                const f32x4 sc = pp_col4(e.scale, wn0 + c);
                f32x4 x;
                if (lane < 16) { if constexpr (DEQ) x = pp_col4(e.deq, wn0 + c) * sc; else x = sc; }
                else x = pp_col4(e.bias, wn0 + c) * sc;
                *(f32x4*)(colv + (lane >> 4) * 64 + c) = x;
            }
        }
        if constexpr (RES) static_for<PP_RING>([&](auto pc) { res_load(pc); });
        if constexpr (!IDENT && !CV) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { PP_PIN4(ca[0][q]); PP_PIN4(cc[0][q]); }
        }
        if constexpr (RES) {
#pragma unroll
            for (int p = 0; p < PP_RING; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) PP_PIN4(res[p][r]);
        }
        hook();
        static_for<8>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            constexpr int j = LN ? (p & 1) : (p >> 2), i = LN ? (p >> 1) : (p & 3);
            // (LN order: the row block i grows with p, so once a block lies past M every later pass does too -- a row-split tail unit
            //  runs 2-4 of its 8 passes; the tail round is a latency chain of dependent passes, not bandwidth: 38 CUs)
            if (LN && skip_dead && wm0 + i * 32 >= M) return;
            static_for<4>([&](auto qc) {
                constexpr int q4 = decltype(qc)::value;
                f32x4 v = PP_QUAD(i, j, q4);
                if constexpr (CV) v = v * *(const f32x4*)(colv + j * 32 + 8 * q4 + 4 * kg) + *(const f32x4*)(colv + 64 + j * 32 + 8 * q4 + 4 * kg);
                else if constexpr (!IDENT) v = v * ca[j][q4] + cc[j][q4];
                stg[fl * 8 + ((2 * q4 + kg) ^ (fl & 7))] = __builtin_bit_cast(u32x4, v);
            });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r * 8 + rr;
                f32x4 x = __builtin_bit_cast(f32x4, stg[row * 8 + (rc ^ (row & 7))]);
                if constexpr (RES) x += res[p % PP_RING][r];
                // (IDENT = a K-split unit's slab: written through, sc0 sc1, so the partner units that sum it -- pp_tail_reduce_inkernel --
                //  read it coherently wherever they run; the other kinds keep their lines in the XCD's L2 for the next launch)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rsO, vo + ((i * 32 + r * 8) * ldc + j * 32) * 4u, 0, IDENT ? PP_AUX_SC0_SC1 : 0);
                if constexpr (LN) {
                    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                    const u32x2_ t = {pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(t, rsO2, (vo >> 1) + ((i * 32 + r * 8) * ldc + j * 32) * 2u, 0, 0);
                    const float a1 = (x[0] + x[1]) + (x[2] + x[3]), a2 = (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
                    if constexpr (j == 0) { s1[r] = a1; s2[r] = a2; } else { s1[r] += a1; s2[r] += a2; }
                }
            }
            if constexpr (LN && j == 1) {
                // the 8 lanes rc = 0..7 of a row: xor 1, 2, 4 inside the wave's 32-lane halves (ds_swizzle: no LDS memory involved)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // DPP, no LDS round trips: quad_perm [1,0,3,2] (xor 1), [2,3,0,1] (xor 2), row_half_mirror (lane k <-> 7 - k of its 8)
                    float t1 = s1[r], t2 = s2[r];
                    t1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1), 0xB1, 0xf, 0xf, false));
                    t2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t2), 0xB1, 0xf, 0xf, false));
                    t1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1), 0x4E, 0xf, 0xf, false));
                    t2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t2), 0x4E, 0xf, 0xf, false));
                    t1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1), 0x141, 0xf, 0xf, false));
                    t2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t2), 0x141, 0xf, 0xf, false));
                    if (rc == 0) lnred[wn * 128 + i * 32 + r * 8 + rr] = make_float2(t1, t2);
                }
            }
            asm volatile("" ::: "memory");                        // (keeps the ring PP_RING passes deep: no hoisting of later loads)
            if constexpr (RES && p + PP_RING < 8) res_load(std::integral_constant<int, p + PP_RING>{});
            if constexpr (p == 1 && !IDENT && !CV) col_load(std::integral_constant<int, 1>{});     // two more passes until the other column half
        });
        if constexpr (LN) {
            // the four N-quarter waves of this M half -> one partial per row and tile column.  One block barrier (every wave of the
            // block passes it exactly once per tile, so the two groups' barrier counts stay equal), then wave wn == 0 adds in fixed order.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pp_barrier();
            if (wn == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = h * 64 + lane;
                    float2 t = lnred[row];
#pragma unroll
                    for (int w = 1; w < 4; ++w) { const float2 u = lnred[w * 128 + row]; t.x += u.x; t.y += u.y; }
                    const int m = wm0 + row;
                    if (m < M) ((float2*)e.stats_out)[(long)(bn0 >> 8) * Mfull + m] = t;
                }
            }
        }
        return;
    } else {
        f32x4 cb[2][4], cd[2][4];
        float lmean[4], lrstd[4];
        float2 lst[4][2];
        auto ln_request = [&]() {
            // LN consumer: the (sum, sum of squares) partials of this lane's four rows (they come from the fabric: another XCD wrote
            // them) -- K half kg takes slots kg and kg + 2; a slot past ln_slots re-reads slot 0 and is dropped by a select (no
            // branch around a load).  colsum(W') of the wave's 64 columns goes through `colv` (LDS); the bias stays in registers like
            // the plain kinds' where there is room (FC1): with both vectors in LDS every quad waits for two dependent ds_read_b128.
            static_assert(!LN || !DEQ, "LN-folded consumers run on bf16 operands");
            const long lnM = e.ln_M ? e.ln_M : M;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m = wm0 + i * 32 + fl; m = m < M ? m : M - 1;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int sl = kg + 2 * q;
                    lst[i][q] = ((const float2*)e.ln_stats)[(sl < e.ln_slots ? sl : 0) * lnM + m];
                }
            }
        };
        // (the QKV kinds carry two epilogues -- row-major and V transposed -- and have no registers left for the bias: both vectors
        //  through LDS there, measured 8 spilled registers against 88 and 110 vs 150 us per launch at batch 32)
        constexpr bool CBREG = !LN || ACT == ACT_GELU;
        if constexpr (LN && CBREG) ln_request();
        if constexpr (CBREG) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = wn0 + j * 32 + 8 * q + 4 * kg;
                    cb[j][q] = pp_col4(e.bias, n0);
                    if constexpr (DEQ) cd[j][q] = pp_col4(e.deq, n0);
                }
        }
        if constexpr (LN) {
            if constexpr (CBREG) { if (lane < 16) *(f32x4*)(colv + lane * 4) = pp_col4(e.ln_csum, wn0 + lane * 4); }
            else if (lane < 32) *(f32x4*)(colv + (lane >> 4) * 64 + (lane & 15) * 4) = pp_col4(lane < 16 ? e.ln_csum : e.bias, wn0 + (lane & 15) * 4);
        }
        if constexpr (LN && !CBREG) ln_request();
        if constexpr (CBREG) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) { PP_PIN4(cb[j][q]); if constexpr (DEQ) PP_PIN4(cd[j][q]); }
        }
        auto ln_rows = [&]() {                       // mean / rstd of this lane's four rows, as the two factors the passes use
            const float inv_d = 1.0f / (float)e.ln_dim;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t1 = (kg + 2 < e.ln_slots ? lst[i][1].x : 0.f) + (kg < e.ln_slots ? lst[i][0].x : 0.f);
                float t2 = (kg + 2 < e.ln_slots ? lst[i][1].y : 0.f) + (kg < e.ln_slots ? lst[i][0].y : 0.f);
                t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
                lmean[i] = t1 * inv_d;
                lrstd[i] = rsqrtf(fmaxf(t2 * inv_d - lmean[i] * lmean[i], 0.f) + e.ln_eps);
                if constexpr (CBREG) lmean[i] = -lmean[i] * lrstd[i];   // FC1: v = acc * rstd + colsum * (-mean * rstd) + bias
                asm volatile("" : "+v"(lmean[i]), "+v"(lrstd[i]));
            }
        };
        if constexpr (LN && CBREG) ln_rows();        // (FC1: before the hook, 8 live registers instead of 16)
        if constexpr (LN && !CBREG) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) { asm volatile("" : "+v"(lst[i][q].x), "+v"(lst[i][q].y)); }
        }
        const unsigned ldc = (unsigned)e.ldc;
        constexpr unsigned OB = OUT8 ? 1u : 2u;     // bytes per output element
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, (unsigned)M * ldc * OB, 0x00020000);
        const unsigned vo = ((unsigned)(wm0 + rr) * ldc + (unsigned)(wn0 + rc * 8)) * OB;
        const float qs = e.out_qscale;
        hook();
        if constexpr (LN && !CBREG) ln_rows();
        static_for<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (MODE == PP_EP_VT) {
                int m = wm0 + i * 32 + fl; m = m < M ? m : M - 1;
                const int b = m / e.ntok, t = m - b * e.ntok;
                // one per-lane byte offset for the row block (column 0 of this wave) + a wave-uniform offset per column: buffer stores,
                // no 64-bit pointer per quad (the V^T tensor is far below 2 GiB: pp_supported)
                const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(e.vt, 0, 0x7fffffff, 0x00020000);
                const unsigned vtv = (unsigned)((((long)b * e.heads * 64 + (wn0 + 4 * kg - e.qk_cols)) * e.npad + t) * 2);
                const int col_bytes = e.npad * 2;
                static_for<2>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    static_for<4>([&](auto qc) {
                        constexpr int q4 = decltype(qc)::value;
                        const int n0 = wn0 + j * 32 + 8 * q4 + 4 * kg;
                        f32x4 v = PP_QUAD(i, j, q4);
                        if constexpr (DEQ) v *= cd[j][q4];
                        if constexpr (LN && CBREG) v = v * lrstd[i] + (*(const f32x4*)(colv + j * 32 + 8 * q4 + 4 * kg) * lmean[i] + cb[j][q4]);
                        else if constexpr (LN) v = (v - *(const f32x4*)(colv + j * 32 + 8 * q4 + 4 * kg) * lmean[i]) * lrstd[i] + *(const f32x4*)(colv + 64 + j * 32 + 8 * q4 + 4 * kg);
                        else v += cb[j][q4];
                        (void)n0;
                        const unsigned pk01 = pk_bf16(v[0], v[1]), pk23 = pk_bf16(v[2], v[3]);
                        constexpr int c0 = j * 32 + 8 * q4;
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk01 & 0xffffu), rsV, vtv, (c0 + 0) * col_bytes, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk01 >> 16), rsV, vtv, (c0 + 1) * col_bytes, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk23 & 0xffffu), rsV, vtv, (c0 + 2) * col_bytes, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk23 >> 16), rsV, vtv, (c0 + 3) * col_bytes, 0);
                    });
                });
            } else {
                static_for<2>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    static_for<4>([&](auto qc) {
                        constexpr int q4 = decltype(qc)::value;
                        f32x4 v = PP_QUAD(i, j, q4);
                        if constexpr (DEQ) v *= cd[j][q4];
                        if constexpr (LN && CBREG) v = pp_act4<ACT>(v * lrstd[i] + (*(const f32x4*)(colv + j * 32 + 8 * q4 + 4 * kg) * lmean[i] + cb[j][q4]));
                        else if constexpr (LN) v = pp_act4<ACT>((v - *(const f32x4*)(colv + j * 32 + 8 * q4 + 4 * kg) * lmean[i]) * lrstd[i] + *(const f32x4*)(colv + 64 + j * 32 + 8 * q4 + 4 * kg));
                        else v = pp_act4<ACT>(v + cb[j][q4]);
                        if constexpr (OUT8) {
                            // the same patch geometry at half the element size: row pitch 64 bytes, 8-byte pieces (8 columns), piece index
                            // XOR (row & 7); a pass still issues four stores (8 rows x 64 bytes each): PP_TAIL holds
                            v *= qs;
                            ((uint32_t*)stg)[fl * 16 + 2 * ((j * 4 + q4) ^ (fl & 7)) + kg] = pk_fp8x4(v[0], v[1], v[2], v[3]);
                        } else {
                            uint2 t;
                            t.x = pk_bf16(v[0], v[1]); t.y = pk_bf16(v[2], v[3]);
                            ((uint2*)stg)[(fl * 8 + ((j * 4 + q4) ^ (fl & 7))) * 2 + kg] = t;
                        }
                    });
                });
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r * 8 + rr;
                    if constexpr (OUT8) {
                        typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
                        const uint2 t = ((const uint2*)stg)[row * 8 + (rc ^ (row & 7))];
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2s){t.x, t.y}, rsO, vo + (i * 32 + r * 8) * ldc, 0, 0);
                    } else
                    __builtin_amdgcn_raw_buffer_store_b128(stg[row * 8 + (rc ^ (row & 7))], rsO, vo + (i * 32 + r * 8) * ldc * 2u, 0, 0);
                }
            }
        });
    }
#undef PP_QUAD
}

// Tile order: the tile columns are cut into xn groups (so that a group's W panel stays in one XCD's L2); the tiles are
// ordered group by group, m-slow / n-fast inside a group, and that sequence is cut into 8 contiguous runs of equal length
// (+-1 tile), one per XCD.  Runs rather than whole-row rectangles: 249 tiles (batch 27, N = 768) are 31-32 per XCD -- one
// round on its 32 CUs -- where rectangles of whole tile rows gave one XCD 33; a run may straddle two column groups.
// (xn is a power of two, passed as its log2; the one runtime division per tile is ~40 scalar instructions)
__host__ __device__ __forceinline__ void pp_run(int x, int total, int& start, int& len) {
    start = (int)(((long)total * x) >> 3); len = (int)(((long)total * (x + 1)) >> 3) - start;
}
__device__ __forceinline__ void pp_tile_of(int lin, int tiles_m, int tiles_n, int lxn, int& tm, int& tn) {
    int g = 0, n0 = 0, nl = 1;
    for (;;) {
        const int n1 = (tiles_n * (g + 1)) >> lxn;
        nl = n1 - n0;
        const int T = tiles_m * nl;
        if (lin < T) break;
        lin -= T; n0 = n1; ++g;
    }
    tm = lin / nl; tn = n0 + (lin - tm * nl);
}
enum { PP_K_BF16 = 0, PP_K_GELU = 1, PP_K_QKV = 2, PP_K_F32 = 3,       // epilogue of a launch (one kernel instance each)
       PP_K_GELU_LN = 4, PP_K_QKV_LN = 5, PP_K_F32_LN = 6 };             // the same with LayerNorm folded in (consumer, consumer, producer)
constexpr bool pp_kind_f32(int k) { return k == PP_K_F32 || k == PP_K_F32_LN; }

// ---- in-kernel tail reduce (ink).  The ks K-split units of a tail tile run at the same time (every block of the launch is resident:
// one per CU) and -- for speed only -- on CUs of ONE XCD (unit lists above).  The exchange itself is placement-independent (round 5,
// ADVICE r4): a unit writes its raw partial sums to its fp32 slab with WRITE-THROUGH stores (sc0 sc1: the bytes leave the XCD's L2
// for the fabric, pp_epilogue's IDENT path), waits for their acknowledgement (vmcnt 0), and counts itself with an AGENT-scope atomic;
// whoever sums a share reads the slabs with sc1 loads (never served from a CU's L1) -- MI355X_MICROARCH.md's valid form
// "{sc0 sc1 stores, sc1 loads, agent atomics}", no buffer_wbl2 / buffer_inv of a whole L2 (the `buffer_wbl2` of an agent-scope
// release fence is what made the stream-K exchange of round 2 slower than the rounding it removed).  Unit s sums rows
// [s R, (s + 1) R) of all ks slabs in slab order (the order pp_tail_reduce_kernel uses: bit-identical to the two-launch path) and
// runs the fused epilogue on them.  Round 4 did the same with plain stores + workgroup-scope atomics, correct only while all ks
// units really shared an L2 (blockIdx % 8 = XCC id under the default partition mode, tools/chain/xcd_exchange_probe.hip).
// Counters: four words per tail tile in the engine's own counter block behind the split-K workspace (arrivals, claimed shares,
// departures; zero between launches: the last unit to leave clears them; the workspace belongs to one stream).
__device__ __forceinline__ f32x4 pp_load4_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// Nobody depends on a unit that is not running: a unit waits a BOUNDED time for its partners (they are its XCD's other CUs in the same
// round of the same launch: microseconds apart) and then leaves without its share; the LAST arrival never waits, and after its own
// rows it finishes every share nobody has claimed (atomic claim mask: each share is summed exactly once, in the same slab order
// whoever sums it -- the result does not depend on who did).  So two launches that hold part of the chip each (two engines on two
// streams) cannot dead-lock on each other's unscheduled blocks; they only lose the parallel reduce.
// Units that gave up waiting for their partners (then the last arrival sums their shares: the result is the same bits, only the
// parallel reduce is lost).  Counted so that it cannot happen silently: d2s_debug_pp_tail_timeouts() reads / clears it, the soak tests
// require 0 on a chip the launch has to itself.
__device__ unsigned pp_tail_timeouts;
template <typename T>
__device__ __forceinline__ void pp_tail_reduce_inkernel(const GemmEpi& e, float* part, size_t part_elems, int M, int N, int tm, int tn, int slab, int ks, int tid,
                                                        volatile unsigned* sh /* 4 words of LDS, free at this point */, bool no_wait /* test aid */) {
    const int tt = slab / ks, sl = slab - tt * ks;
    unsigned* ctr = (unsigned*)(part + part_elems) + 4 * tt;       // [0] arrivals, [1] claimed shares (bit s), [2] departures
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's slab stores are in L2
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)ks - 1;
        bool all = last;
        if (!last && !no_wait) {
            const long t0 = wall_clock64();
            while (!(all = __hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)ks)) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > 5000L) break;              // 50 us at 100 MHz
            }
        }
        if (!all) __hip_atomic_fetch_add(&pp_tail_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh[0] = all ? 1u : 0u; sh[1] = last ? 1u : 0u;
    }
    __syncthreads();
    // [2] counts DEPARTURES: every unit leaves exactly once (timed out or done), and whoever leaves last clears the three words --
    // clearing on "all shares finished" instead would let a slow owner claim its (already stolen and finished) share a second time
    auto depart = [&]() {
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&ctr[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)ks - 1) {
                __hip_atomic_store(&ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctr[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctr[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (!sh[0]) { depart(); return; }                                // timed out: the last arrival takes this unit's rows
    const bool last = sh[1] != 0;
    const int RW = (256 + ks - 1) / ks;                              // rows per share
    const int w = tid >> 6, c = (tid & 63) * 4;                      // a wave = one row of the tile (256 columns), 4 columns per lane
    const float* base = part + (size_t)tt * ks * 65536 + c;
    for (int k = 0; k < (last ? ks : 1); ++k) {
        const int s_ = (sl + k) % ks;                                // own share first; the last arrival then sweeps the others
        __syncthreads();                                             // (sh[2] of the previous round has been read)
        if (tid == 0) sh[2] = (__hip_atomic_fetch_or(&ctr[1], 1u << s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> s_) & 1u;
        __syncthreads();
        if (sh[2]) continue;                                         // somebody else has it
        const int r_lo = s_ * RW, r_hi = r_lo + RW < 256 ? r_lo + RW : 256;
        constexpr int RB = 3;                                        // rows per wave in flight
        for (int r0 = r_lo + w; r0 < r_hi; r0 += 8 * RB) {
            f32x4 x[RB][8];
            float pre[RB][4];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = r0 + 8 * i;
                const bool ok = r < r_hi && tm * 256 + r < M;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q < ks) x[i][q] = pp_load4_sc1(base + (size_t)q * 65536 + r * 256);
                    if (e.res1) epi_res1_load<T>(e, tm * 256 + r, tn * 256 + c, pre[i]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = r0 + 8 * i, m = tm * 256 + r;
                // (the asm loads above are invisible to the compiler's own wait counting: tie every value to the wait)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(x[i][q]));
                if (r < r_hi && m < M) {
                    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q < ks) { v[0] += x[i][q][0]; v[1] += x[i][q][1]; v[2] += x[i][q][2]; v[3] += x[i][q][3]; }
                    epilogue_dispatch<T>(e, m, tn * 256 + c, v, false, e.res1 ? pre[i] : nullptr);
                    if (e.stats_out) {                               // LN producer: the wave holds the 256 columns of this row
                        float t1 = (v[0] + v[1]) + (v[2] + v[3]), t2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
                        if ((tid & 63) == 0) ((float2*)e.stats_out)[(long)tn * M + m] = make_float2(t1, t2);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (this unit's stores are out before it counts as gone)
    __syncthreads();
    depart();
}

template <typename T, int KIND>
__global__ void __launch_bounds__(512)
gemm_pp_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, int M, int N, int K, int Kpad, GemmEpi e, int xn /* log2 */,
               int skew_us, int tw /* whole tiles */, int ks /* K splits of the remaining tiles */, int kps /* K tiles per split */,
               int ink /* the K splits of a tail tile share an XCD and reduce in this kernel */) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BK = 128 / ES;                                    // K elements per tile
    constexpr bool DEQ = ES == 1;                                   // e4m3 operands: the accumulator is de-quantised per column
    // 128 KiB of stages + 8 wave-private 4 KiB epilogue patches = the CU's 160 KiB: the ONLY __shared__ object (see gemm.hip)
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * PP_STAGE + 8 * 256];
    D2S_POISON_LDS(lds, 2 * PP_STAGE + 8 * 256)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wn = wid & 3;
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int nkt = K / BK;                                          // K tiles per output tile: even, >= 2
    const int cpx = gridDim.x >> 3, xcd = blockIdx.x & 7;
    // This XCD's units: its run of the `tw` whole tiles, then (fp32-residual launches only) its run of the K-split units of the
    // remaining tiles.  Tile rounding: one tile per CU per round, so 294 tiles (batch 32, N = 768) would pay a whole second
    // round for 38 tiles; instead those 38 are cut into ks K ranges each (228 units, one short round) whose raw partial sums
    // go to fp32 slabs [unit][256][256] and are summed + finished by pp_tail_reduce_kernel.
    // ink: the TILES of the tail are dealt to the XCDs (not their K-split units), so that all ks units of a tile run on CUs of one
    // XCD and can exchange their slabs through that XCD's L2 without device-scope fences (below, "in-kernel tail reduce").
    int runA = 0, nA = 0, runB = 0, nB = 0;
    pp_run(xcd, tw, runA, nA);
    if constexpr (pp_kind_f32(KIND)) {
        if (ink & 1) { pp_run(xcd, tiles_m * tiles_n - tw, runB, nB); runB *= ks; nB *= ks; }
        else pp_run(xcd, (tiles_m * tiles_n - tw) * ks, runB, nB);
    }
    const int ntl = nA + nB;
    int tl = blockIdx.x >> 3;                                        // this block's position in its XCD's unit list
    if (tl >= ntl) return;
    PP_STAMP(0)
    int stamp_ = 1; (void)stamp_;
    // Blocks that walk one tile fewer than the busiest of their XCD start late by about half a tile time: it costs nothing
    // (they finish before the others anyway) and it takes them out of the store bursts -- a lock-step grid dumps 128 KiB
    // per CU at the same instant, that drains in ~8 us, and every CU sits on vmcnt behind its own stores meanwhile.
    // (A further phase shift of x us per XCD, to spread the store bursts of the epilogues: x = 1 -> -2 us per launch, x = 2..3 -> worse.)
    if (skew_us > 0 && (ntl - 1 - tl) / cpx < (ntl - 1) / cpx)
        for (int i = 0; i < skew_us; ++i) __builtin_amdgcn_s_sleep(32);        // ~1 us each (64 x 32 cycles)

    // ---- loader: this wave's two LDS-DMA instructions of each piece.  lane -> (row lr of 8, physical chunk lc)
    // per-lane byte offsets (K tile 0).  A: [quadrant][instruction] (rows are clamped per lane).  W: ONE register -- row
    // rw0 + lr of piece (q, j) has swizzle ((rw0 + lr) >> 1) & 7 = (lr >> 1) + 4 j, so the lane part is laneW ^ (64 j) and
    // everything else ((bn0 + rw0) * Kpad) is wave-uniform and rides in the scalar offset
    unsigned offA[2][2], laneW;
    int dstA[2][2], dstW[2][2];                  // wave-uniform LDS chunk index of the instruction's first slot
    int sofW[2][2];                              // wave-uniform byte offset of W piece (q, j) of the current tile
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = wid * 2 + j;
            dstA[q][j] = ((i >> 3) * 128 + q * 64 + (i & 7) * 8) * 8;                  // A piece q: rows {0..63 | 64..127} of both M halves
            dstW[q][j] = PP_WOFF + ((i >> 2) * 64 + q * 32 + (i & 3) * 8) * 8;         // W piece q: rows {0..31 | 32..63} of the four N quarters
        }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    int kt0 = 0;                                  // first K tile of the unit whose operands are being requested
    // per-lane source offsets of a tile: the lane that owns LDS slot (row r, physical chunk lc) fetches chunk lc ^ ((r >> 1) & 7)
#define PP_SET_TILE(BM0, BN0, LANE)                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                      \
            const int lr = (LANE) >> 3, lc = (LANE) & 7;                                                                     \
            const int ra = (dstA[q][j] >> 3) + lr, rw = ((dstW[q][j] - PP_WOFF) >> 3) + lr;                                  \
            int m_ = (BM0) + ra; m_ = m_ < M ? m_ : M - 1;         /* rows past M duplicate the last row */                  \
            offA[q][j] = (unsigned)((long)m_ * lda * ES) + (unsigned)((lc ^ ((ra >> 1) & 7)) * 16);                          \
            sofW[q][j] = ((BN0) + rw - lr) * Kpad * ES;                                                                      \
            if (q == 0 && j == 0) laneW = (unsigned)(lr * Kpad * ES) + (unsigned)((lc ^ (lr >> 1)) * 16);                    \
        }

#define PP_ISSUE_A(Q, KT)                                                                                                   \
    {                                                                                                                        \
        const int so_ = ((KT) + kt0) * 128;                                                                                  \
        u32x4* st_ = lds + ((KT) & 1) * PP_STAGE;                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(st_ + dstA[Q][0]), 16, offA[Q][0], so_, 0, 0);           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(st_ + dstA[Q][1]), 16, offA[Q][1], so_, 0, 0);           \
    }
#define PP_ISSUE_W(Q, KT)                                                                                                   \
    {                                                                                                                        \
        const int so_ = ((KT) + kt0) * 128;                                                                                  \
        u32x4* st_ = lds + ((KT) & 1) * PP_STAGE;                                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(st_ + dstW[Q][0]), 16, laneW, so_ + sofW[Q][0], 0, 0);     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(st_ + dstW[Q][1]), 16, laneW ^ 64u, so_ + sofW[Q][1], 0, 0); \
    }
    // what the steady state would have issued before (0, P1): tile 0 complete + A_m0 / W_n0 of tile 1, in consumption
    // order (12 instructions per wave)
#define PP_PROLOGUE() { PP_ISSUE_A(0, 0) PP_ISSUE_W(0, 0) PP_ISSUE_W(1, 0) PP_ISSUE_A(1, 0) PP_ISSUE_A(0, 1) PP_ISSUE_W(0, 1) }

    // ---- fragment reads: lane -> (row fl of a 32-row block, K half kg); chunk (2 ks + kg) ^ swizzle(row).
    // The four per-lane A addresses are the only address VGPRs kept across the K loop; a W read uses the same register plus
    // a wave-uniform delta that is re-materialised (opaque to the compiler) at each use -- four more live VGPRs are what
    // pushed the loop over 256 registers.
    const int fl = lane & 31, kg = lane >> 5, sw = (fl >> 1) & 7;
    const u32x4* rdA[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rdA[ks] = lds + (grp * 128 + fl) * 8 + ((2 * ks + kg) ^ sw);
    int wdelta = PP_WOFF + (wn * 64 - grp * 128) * 8;            // chunks from this wave's A rows to its W rows

    constexpr bool K64 = ES == 1 && PP_FP8_K64 != 0;
    PPFrag<K64> fX, fY, fZ, fw[2];                // A: [ks] (see the table on top); W: [set][ks], n0 in set (tile parity), n1 in the other
    f32x16 acc[4][2];                             // [M quadrant * 2 + mb][N quadrant]

#define PP_READ_A(DST, Q, MB, KT)                                                                                           \
    {                                                                                                                        \
        const int o_ = ((KT) & 1) * PP_STAGE + (Q) * 64 * 8 + (MB) * 32 * 8;                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) DST.set(ks, rdA[ks][o_]);                                           \
    }
#define PP_READ_W(DST, Q, KT)                                                                                               \
    {                                                                                                                        \
        asm volatile("" : "+s"(wdelta));                                                                                     \
        const int o_ = wdelta + ((KT) & 1) * PP_STAGE + (Q) * 32 * 8;                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) DST.set(ks, rdA[ks][o_]);                                           \
    }
    // 8 MFMAs of one quadrant: the four of the first 32-row block (pre-read operands) run first
#define PP_MFMA(QM, QN, FA0, FA1, FW)                                                                                       \
    {                                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        if constexpr (K64) {                                                                                                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) mma64(acc[(QM) * 2 + 0][QN], FW.t[s_], FA0.t[s_]);              \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) mma64(acc[(QM) * 2 + 1][QN], FW.t[s_], FA1.t[s_]);              \
            /* (pinned in place: the results of the LAST two K tiles are only read by the epilogue, and machine sinking moved  \
                their 32 MFMAs -- operands spilled, ~90 dwords -- below the barriers into the epilogue's block) */              \
            asm volatile("" : "+v"(acc[(QM) * 2 + 0][QN]), "+v"(acc[(QM) * 2 + 1][QN]));                                       \
        } else {                                                                                                             \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) mma32(acc[(QM) * 2 + 0][QN], FW.c[ks], FA0.c[ks], T());         \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) mma32(acc[(QM) * 2 + 1][QN], FW.c[ks], FA1.c[ks], T());         \
        }                                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
    }
    // one K tile = 4 phases (see the table on top).  P = tile parity (literal).  ISSUE: 2 = steady state, 1 = second-to-last
    // tile (pieces of the last tile only), 0 = last tile (nothing to issue, nothing to pre-read).  X (first K tile of a
    // segment that follows an epilogue): that many younger global stores of the epilogue may stay in flight behind the
    // prologue pieces the first three waits are about.
#define PP_TILE(KT, ISSUE, P, X)                                                                                            \
    {                                                                                                                        \
        if (ISSUE >= 1) PP_ISSUE_W(1, (KT) + 1)                                                                              \
        PP_READ_W(fw[(P) ^ 1], 1, KT) PP_READ_A(fY, 0, 1, KT)                                                                \
        if (ISSUE >= 1) pp_wait_vm<6 + (X)>(); else pp_wait_vm<0>();                                                         \
        pp_barrier();                                                                                                        \
        PP_MFMA(0, 0, fX, fY, fw[P]) pp_barrier();                                                                           \
        if (ISSUE >= 1) PP_ISSUE_A(1, (KT) + 1)                                                                              \
        PP_READ_A(fZ, 1, 0, KT)                                                                                              \
        if (ISSUE >= 1) pp_wait_vm<6 + (X)>();                                                                               \
        pp_barrier();                                                                                                        \
        PP_MFMA(0, 1, fX, fY, fw[(P) ^ 1]) pp_barrier();                                                                     \
        if (ISSUE >= 2) PP_ISSUE_A(0, (KT) + 2)                                                                              \
        PP_READ_A(fY, 1, 1, KT)                                                                                              \
        if (ISSUE >= 2) pp_wait_vm<6 + (X)>(); else if (ISSUE == 1) pp_wait_vm<4 + (X)>();                                   \
        pp_barrier();                                                                                                        \
        PP_MFMA(1, 1, fZ, fY, fw[(P) ^ 1]) pp_barrier();                                                                     \
        if (ISSUE >= 2) PP_ISSUE_W(0, (KT) + 2)                                                                              \
        if (ISSUE >= 1) { PP_READ_W(fw[(P) ^ 1], 0, (KT) + 1) PP_READ_A(fX, 0, 0, (KT) + 1) }                                \
        if (ISSUE >= 2) pp_wait_vm<6>(); else if (ISSUE == 1) pp_wait_vm<2>();                                               \
        pp_barrier();                                                                                                        \
        PP_MFMA(1, 0, fZ, fY, fw[P]) pp_barrier();                                                                           \
    }

    // ---- this block's units.  unit j of the list -> (tile, first K tile, K tiles, slab index or -1)
    // (ink & 4: ROW-split tail for the short-K residual launches -- proj: a tail tile is cut into ks row slices of kps rows, each unit
    //  runs the whole K loop on the 256-row window that STARTS at its slice and stores only its slice; slab = -2 - slice)
    auto unit_of = [&](int j, int& tm, int& tn, int& k0, int& nk, int& slab) {
        if (!pp_kind_f32(KIND) || j < nA) { pp_tile_of(runA + j, tiles_m, tiles_n, xn, tm, tn); k0 = 0; nk = nkt; slab = -1; }
        else if (ink & 4) {
            const int u = runB + (j - nA), tt = u / ks;
            pp_tile_of(tw + tt, tiles_m, tiles_n, xn, tm, tn);
            k0 = 0; nk = nkt; slab = -2 - (u - tt * ks);
        }
        else {
            const int u = runB + (j - nA), tt = u / ks;
            pp_tile_of(tw + tt, tiles_m, tiles_n, xn, tm, tn);
            k0 = (u - tt * ks) * kps; nk = kps; slab = u;
        }
    };
    int tm_ = 0, tn_ = 0, nkt_u = nkt, slab_u = -1;
    unit_of(tl, tm_, tn_, kt0, nkt_u, slab_u);
    auto row0_of = [&](int tm, int slab) { return tm * 256 + (slab <= -2 ? (-2 - slab) * kps : 0); };
    PP_SET_TILE(row0_of(tm_, slab_u), tn_ * 256, lane)
    PP_PROLOGUE()
    pp_wait_vm<6>();                                // first tile: A_m0[0], W_n0[0], W_n1[0] of this wave have landed
    bool after_epi = false;
    while (true) {
        const int bm0 = row0_of(tm_, slab_u), bn0 = tn_ * 256;
        int Mu = M;                                  // a row-split unit stores its slice only (clipped at the end of its tile)
        if (slab_u <= -2) { Mu = bm0 + kps < tm_ * 256 + 256 ? bm0 + kps : tm_ * 256 + 256; Mu = Mu < M ? Mu : M; }
        float zero_ = 0.f;                           // opaque: a loop-invariant zero TUPLE gets hoisted out of the persistent loop and spilled
        asm volatile("" : "+v"(zero_));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = zero_;
        pp_barrier();                               // every wave's share of A_m0[0], W_n0[0], W_n1[0] is in LDS
        PP_READ_A(fX, 0, 0, 0) PP_READ_W(fw[0], 0, 0)
        if (grp == 1) pp_barrier();                 // group 1 runs one barrier behind group 0 from here on
        PP_STAMP(stamp_)
        int kt = 0;
        // behind an epilogue the first three waits leave its (>= PP_TAIL) stores in flight: they sit between the prologue's
        // pieces and the ones issued here, and a plain vmcnt(6) would wait for their round trip to HBM
        asm volatile("" : "+s"(kt));                // (a literal K tile index would put sixteen hoisted LDS addresses in VGPRs)
        if (after_epi && nkt_u >= 4) { PP_TILE(kt, 2, 0, PP_TAIL) PP_TILE(kt + 1, 2, 1, 0) kt += 2; }
        for (; kt + 2 < nkt_u; kt += 2) { PP_TILE(kt, 2, 0, 0) PP_TILE(kt + 1, 2, 1, 0) }
        PP_TILE(kt, 1, 0, 0)
        PP_TILE(kt + 1, 0, 1, 0)
        if (grp == 0) pp_barrier();                 // same barrier count for both groups; every fragment read has returned
        PP_STAMP(stamp_ + 1)
        PP_WSTAMP(stamp_ + 1)
        // next tile of this block
        const int ntl_next = tl + cpx;
        const bool more = ntl_next < ntl;
        int ntm = 0, ntn = 0, nk0 = 0, nnk = nkt, nslab = -1;
        if (more) unit_of(ntl_next, ntm, ntn, nk0, nnk, nslab);
        kt0 = nk0;                                  // (the K loop of this unit is over: from here on kt0 serves the hook's prologue)
        // ---- epilogue.  Its operands pass through an empty asm so that nothing of it is loop-invariant to the compiler:
        // hoisted out of the persistent loop, the epilogue's addresses and column vectors would live (and spill) across the K loop
        GemmEpi el = e;
        int lane_e = lane;
        {
            long z = 0;                              // an opaque zero: pointers stay kernel-argument (global) pointers
            asm volatile("" : "+s"(z));
            asm volatile("" : "+v"(lane_e));
            el.out = (char*)e.out + z; el.bias = e.bias + z; el.scale = e.scale + z; el.deq = e.deq + z;
            el.res1 = e.res1 ? (const char*)e.res1 + z : nullptr; el.vt = e.vt ? (char*)e.vt + z : nullptr;
            if constexpr (KIND >= PP_K_GELU_LN) {
                el.out2 = e.out2 ? (char*)e.out2 + z : nullptr; el.stats_out = e.stats_out ? e.stats_out + z : nullptr;
                el.ln_stats = e.ln_stats ? e.ln_stats + z : nullptr; el.ln_csum = e.ln_csum ? e.ln_csum + z : nullptr;
            }
        }
        u32x4* stg = lds + 2 * PP_STAGE + wid * 256;
        // LN producer: 2 x 4 x 128 float2 partials in the rows 64..127 of stage 1's A part -- the last K tile's fragment reads have
        // returned (barrier above) and the next tile's prologue does not write there before its first K tile's second phase
        float2* lnred = (float2*)(lds + PP_STAGE + 64 * 8) + grp * 512;
        // LN variants: the two column vectors of this wave's 64 columns (128 floats) live in LDS during the epilogue instead of 64
        // VGPRs -- rows 32.. of this wave's N quarter of stage 1's W part, which the next tile requests in its first phase, i.e.
        // behind the tile-start barrier every wave reaches with its epilogue (and its LDS reads) complete
        float* colv = (float*)(lds + PP_STAGE + PP_WOFF + (wn * 64 + 32) * 8 + grp * 64);
        // the next segment's first twelve pieces leave from inside the epilogue (both stages are free: the last fragment
        // read returned before the barrier above), followed by >= PP_TAIL stores of this wave
        auto hook = [&]() { if (more) { PP_SET_TILE(row0_of(ntm, nslab), ntn * 256, lane_e) PP_PROLOGUE() } };
        {
            if constexpr (pp_kind_f32(KIND)) {
                if (slab_u >= 0) {                  // a K split of a tail tile: raw partial sums to the unit's slab
                    GemmEpi es = el;
                    es.out = (char*)e.part + (size_t)slab_u * (65536 * 4);
                    es.ldc = 256;
                    pp_epilogue<PP_EP_F32, ACT_NONE, DEQ, false, true>(acc, es, 0, 0, 256, 256, grp, wn, lane_e, stg, lnred, colv, hook);
                } else if constexpr (KIND == PP_K_F32_LN) pp_epilogue<PP_EP_F32, ACT_NONE, DEQ, true, false, true>(acc, el, bm0, bn0, Mu, M, grp, wn, lane_e, stg, lnred, colv, hook, !more);
                else if (e.res1) pp_epilogue<PP_EP_F32, ACT_NONE, DEQ, true>(acc, el, bm0, bn0, Mu, M, grp, wn, lane_e, stg, lnred, colv, hook);
                else pp_epilogue<PP_EP_F32, ACT_NONE, DEQ, false>(acc, el, bm0, bn0, Mu, M, grp, wn, lane_e, stg, lnred, colv, hook);
            }
            // (e4m3 operands: the GELU kind writes e4m3 only -- FC1 hands e4m3 to FC2, OUT_T; one epilogue per kernel instance: with a
            //  bf16 variant beside it the two sets of pinned column vectors spill 58 registers)
            else if constexpr (KIND == PP_K_GELU) pp_epilogue<PP_EP_BF16, ACT_GELU, DEQ, false, false, false, ES == 1>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
            else if constexpr (KIND == PP_K_GELU_LN) pp_epilogue<PP_EP_BF16, ACT_GELU, DEQ, false, false, true>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
            else if constexpr (KIND == PP_K_QKV) {
                if (bn0 >= e.qk_cols) pp_epilogue<PP_EP_VT, ACT_NONE, DEQ, false>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
                else pp_epilogue<PP_EP_BF16, ACT_NONE, DEQ, false>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
            } else if constexpr (KIND == PP_K_QKV_LN) {
                if (bn0 >= e.qk_cols) pp_epilogue<PP_EP_VT, ACT_NONE, DEQ, false, false, true>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
                else pp_epilogue<PP_EP_BF16, ACT_NONE, DEQ, false, false, true>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
            } else pp_epilogue<PP_EP_BF16, ACT_NONE, DEQ, false>(acc, el, bm0, bn0, M, M, grp, wn, lane_e, stg, lnred, colv, hook);
        }
        PP_STAMP(stamp_ + 2)
        PP_WSTAMP(stamp_ + 2)
        if (!more) {
            // in-kernel tail reduce: a K-split unit is always the LAST unit of its block (the launcher keeps an XCD's tail units <= its
            // CUs), so the exchange runs here, outside the persistent loop -- nothing of the loop is live any more
            if constexpr (pp_kind_f32(KIND)) { if ((ink & 1) && slab_u >= 0) pp_tail_reduce_inkernel<T>(e, e.part, e.part_elems, M, N, tm_, tn_, slab_u, ks, tid, (volatile unsigned*)lds, (ink & 2) != 0); }
            break;
        }
        pp_wait_vm<6 + PP_TAIL>();                  // the first six pieces of the next segment have landed
        PP_STAMP(stamp_ + 3)
        stamp_ += 4;
        tl = ntl_next; tm_ = ntm; tn_ = ntn; nkt_u = nnk; slab_u = nslab;
        after_epi = true;
    }
#undef PP_TILE
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_W
#undef PP_ISSUE_A
#undef PP_ISSUE_W
#undef PP_PROLOGUE
#undef PP_SET_TILE
}

// second pass of a launch with K-split tail tiles: sum the ks slabs of each tail tile and run the fused epilogue once
template <typename T>
__global__ void __launch_bounds__(256)
pp_tail_reduce_kernel(GemmEpi e, int M, int N, int tw, int ks, int lxn) {
    const int tt = blockIdx.x >> 6;                                 // 64 blocks of 256 threads per 256 x 256 tile, 4 columns per thread
    const int q = (blockIdx.x & 63) * 256 + threadIdx.x;
    const int r = q >> 6, c = (q & 63) * 4;
    int tm = 0, tn = 0;
    pp_tile_of(tw + tt, (M + 255) / 256, (N + 255) / 256, lxn, tm, tn);
    const int m = tm * 256 + r, n0 = tn * 256 + c;
    if (m >= M) return;
    const float* p = e.part + (size_t)tt * ks * 65536 + r * 256 + c;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ks; ++s) { float t[4]; load4(p + (size_t)s * 65536, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
    epilogue_dispatch<T>(e, m, n0, v);                      // (writes out2 as well; v = the stored values)
    if (e.stats_out) {                                       // LN producer: a wave = the 256 columns of one row of this tile
        float t1 = (v[0] + v[1]) + (v[2] + v[3]), t2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
        if ((threadIdx.x & 63) == 0) ((float2*)e.stats_out)[(long)tn * M + m] = make_float2(t1, t2);
    }
}

// tile code 256256: the ping-pong kernel.  Requirements: plain row-major A, no ReLU-on-load, an even number (>= 2) of whole K
// tiles, MAP_ROWS / MAP_QKV, bf16 or f32 output with plain row mapping (the epilogue is staged through LDS), GELU only with
// bf16 output, no LayerNorm folding (the engine keeps the LN kernels at the batch sizes that use this tile).
bool pp_supported(int precision, const GemmA& a, int M, int N, int K, int Kpad, const GemmEpi& e) {
    if (precision != D2S_PREC_BF16 && precision != D2S_PREC_FP8_OPERANDS) return false;
    const int bk = 128 / (int)elem_size(precision);
    if (M <= 0 || N <= 0 || K < 2 * bk) return false;                   // (K = 0 would pass the parity test below and run two unloaded tiles)
    if (a.mode != A_PLAIN || a.relu || K % (2 * bk) || (N & 255)) return false;
    if (e.map != MAP_ROWS && e.map != MAP_QKV) return false;
    const bool ln_cons = e.ln_stats || e.ln_csum, ln_prod = e.stats_out || e.out2;
    if (ln_cons || ln_prod) {
        // LayerNorm folded in (bf16 only): the consumer reads <= 4 partials per row (N of the producer <= 1024), the producer is the
        // fp32 in-place residual update writing the bf16 copy + one partial per 256-column tile
        static EnvInt off{"D2S_PP_NO_LN", 0};
        if (off.get() || precision != D2S_PREC_BF16 || (ln_cons && ln_prod)) return false;
        if (ln_cons && !(e.ln_stats && e.ln_csum && e.ln_slots >= 1 && e.ln_slots <= 4 && e.out_type != OUT_F32 && (e.map == MAP_QKV || e.act == ACT_GELU))) return false;
        if (ln_prod && !(e.stats_out && e.out2 && e.stats_slots && e.out_type == OUT_F32 && e.res1 && e.res1 == e.out && !e.out2_bx3 && e.out2_qscale == 0.f && N <= 1024)) return false;
    }
    if (e.rows_per_img || e.res1_mod || (e.ldc & 7) || e.res2 || N > PP_MAXN) return false;
    bool out_bf16 = e.out_type == OUT_BF16 || (e.out_type == OUT_T && precision == D2S_PREC_BF16);
    // e4m3 operands: the GELU kind writes e4m3 and nothing else (FC1 -> FC2 of the e4m3 schemes), 8-byte aligned rows
    static EnvInt no_out8{"D2S_PP_NO_OUT8", 0};                         // (bisecting aid: FC1 of the e4m3 schemes back on the 128 x 128 tiles)
    if (precision == D2S_PREC_FP8_OPERANDS && e.act == ACT_GELU) out_bf16 = e.out_type == OUT_T && e.map == MAP_ROWS && e.out_qscale > 0.f && !no_out8.get();
    if (e.out_type == OUT_F32) { if (e.act != ACT_NONE || e.map != MAP_ROWS) return false; }
    else if (!(out_bf16 && !e.res1 && !e.res2 && !e.scale && (e.act == ACT_NONE || (e.act == ACT_GELU && e.map == MAP_ROWS)))) return false;
    if (e.map == MAP_QKV && (e.qk_cols & 255)) return false;            // a block tile is entirely q|k or entirely v
    if ((long)M * e.ldc * 4 >= (1L << 31)) return false;               // the epilogue's 32-bit buffer offsets
    if (e.map == MAP_QKV && (long)cdiv(M, e.ntok > 0 ? e.ntok : 1) * e.heads * 64 * e.npad * 2 >= (1L << 31)) return false;   // ... and the V^T store's
    if ((long)M * a.lda * (long)elem_size(precision) >= (1L << 31) || (long)gemm_npad(N) * Kpad * (long)elem_size(precision) >= (1L << 31)) return false;
    return true;
}

// constant column vectors that stand in for an absent bias (zeros) / LayerScale / de-quantisation (ones), one set per device
static const float* pp_const_vec(bool ones) {
    static std::mutex mu;
    static float* buf[64] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> g(mu);
    if (d < 0 || d >= 64) return nullptr;
    if (!buf[d]) {
        std::vector<float> h(2 * PP_MAXN, 0.f);
        std::fill(h.begin() + PP_MAXN, h.end(), 1.f);
        float* p = nullptr;
        if (hipMalloc(&p, h.size() * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return nullptr; }
        buf[d] = p;
    }
    return buf[d] + (ones ? PP_MAXN : 0);
}

int launch_gemm_pp(int precision, const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st) {
    if (!pp_supported(precision, a, M, N, K, Kpad, e)) { set_error("launch_gemm_pp: unsupported problem"); return D2S_E_UNSUPPORTED; }
    const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256);
    unsigned vgrid = 0;
    int xn = pick_xn(tiles_m, tiles_n, 256, Kpad, elem_size(precision), vgrid);
    if (xn == 0) xn = 1;
    // Residual-update launches (N = D: 3-4 tile columns): ONE column group.  With a column group per XCD set every set reads the whole
    // A panel -- FC2 at batch 32 fetched 629 MB for 234 MB of operands + residual (PMC, profiles/r4_04) -- while a single group walks
    // m-slow / n-fast, so the 3 column tiles of a row block run on neighbouring CUs of one XCD at the same time and the A panel comes in
    // once; the W panel (<= 4.7 MB) is then read by all eight XCDs, a tenth of what is saved.  Round 4 measured the launch time unchanged
    // (FC2 159 -> 157 us, proj 74.5 -> 76.4: the K loop runs at the LDS-fill ceiling either way) and dropped it; it is the default now
    // because it halves the launch's fabric traffic for the same time.  D2S_PP_F32_XN1=0 restores the column groups.
    static EnvInt f32_xn1{"D2S_PP_F32_XN1", 1};
    if (e.out_type == OUT_F32 && f32_xn1.get()) xn = 1;
    // one 160-KiB block per CU, cpx blocks per XCD (blocks beyond an XCD's list exit at once)
    static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
    const int lxn = xn <= 1 ? 0 : (xn == 2 ? 1 : (xn == 4 ? 2 : 3));
    const bool ln = e.ln_csum != nullptr || e.stats_out != nullptr;
    if (ln && e.out_type != OUT_F32 && e.map != MAP_QKV && e.act != ACT_GELU) { set_error("launch_gemm_pp: LN-folded consumer: QKV or FC1 only"); return D2S_E_UNSUPPORTED; }
    const int kind = e.out_type == OUT_F32 ? (ln ? PP_K_F32_LN : PP_K_F32) : (e.map == MAP_QKV ? (ln ? PP_K_QKV_LN : PP_K_QKV) : (e.act == ACT_GELU ? (ln ? PP_K_GELU_LN : PP_K_GELU) : PP_K_BF16));
    if (e.stats_slots) *e.stats_slots = e.stats_out ? tiles_n : 1 << 20;                 // partials per row the consumer will find
    // K-split tail (see the kernel): when the last round would be less than 45 % full and the launch is a residual update
    const int tiles = tiles_m * tiles_n, nkt = K / (128 / (int)elem_size(precision));
    int tw = tiles, ks = 1, kps = nkt;
    static const int split_pct = getenv("D2S_PP_SPLIT") ? atoi(getenv("D2S_PP_SPLIT")) : 45;
    // (only for long K loops: the slab round trip + the second launch cost ~30 us, a 12-K-tile round of proj costs 25 --
    //  measured at batch 32: FC2 164 -> 144 us, proj 65 -> 71)
    // Round 4: the K splits of a tail tile share an XCD and are reduced inside the kernel (pp_tail_reduce_inkernel): no second launch,
    // the exchange stays in one L2 -- which makes the split pay for the 12-K-tile proj as well (D2S_PP_INK=0: the two-launch path)
    // (Round 5, measured and removed: cutting EVERY tile of a launch of less than half a round -- ViT-L at batch 8: 25 x 4 = 100 tiles on
    //  256 CUs -- into two K ranges reduced in-kernel: e4m3 FC2 51 -> 66 us, config 3 batch 8 1 056 -> 1 005 frames/s; the slab round trip
    //  costs more than the idle CUs.)
    static EnvInt ink_on{"D2S_PP_INK", 1};
    int ink = 0;
    if (pp_kind_f32(kind) && e.part && split_pct > 0) {
        const int rounds = tiles / ncu, rem = tiles - rounds * ncu;
        if (rounds >= 1 && rem > 0 && rem * 100 <= split_pct * ncu) {
            static EnvInt ink_mink{"D2S_PP_INK_MINK", 24};            // fewest K tiles per output tile for which the tail is split (proj, 12 K tiles, measured at batch 32: 82 -> 86-94 us split, the slab traffic costs more than its second round)
            if (ink_on.get() && nkt >= ink_mink.get() && rem <= PP_TAIL_MAX && (ncu & 7) == 0) {
                const int per_xcd = cdiv(rem, 8);                     // tail tiles of the busiest XCD
                static EnvInt ink_ks{"D2S_PP_INK_KS", 8};             // tuning aid: most K ranges per tail tile
                for (int s = std::min(8, ink_ks.get()); s >= 2; --s)
                    if (nkt % (2 * s) == 0 && per_xcd * s <= ncu / 8 && (size_t)rem * s * 65536 <= e.part_elems) { ks = s; kps = nkt / s; tw = tiles - rem; ink = ink_on.get() == 2 ? 3 : 1; break; }     // D2S_PP_INK=2 (test aid): nobody waits, the last arrival of a tile sums all of it
            }
            // short K loops (proj: 12 K tiles): a K split costs more in slab traffic than the round it removes (section 3.1f); cut the
            // tail tiles by ROWS instead -- every unit repeats the (short) K loop on a window that starts at its slice and runs 1 / rs of
            // the bandwidth-bound residual epilogue.  No exchange, same MFMA sequence per output: bit-identical.  D2S_PP_RSPLIT=0: off
            // (Round 5, measured and removed: walking a block's units backwards -- the short row-split unit first, the whole tile second,
            //  so that the 104 CUs without a tail unit and the 152 with one reach their residual epilogues at different times: proj
            //  125 -> 131 us, batch 32 3 053 -> 3 038 frames/s on the same box.  The epilogue burst is not shortened by halving the
            //  CUs in it.)
            static EnvInt rsplit_on{"D2S_PP_RSPLIT", 1};
            if (!ink && nkt < 24 && rsplit_on.get() && rem > 0) {
                // (measured at batch 32, 38 tail tiles: proj 82.4 us unsplit, 78.4 / 78.1 / 76.6 / 78.8 with 2 / 3 / 4 / 6 slices -- every slice
                //  unit re-reads its whole A window and W panel, so more slices buy shorter epilogues with more fill traffic)
                int rs = ncu / rem; if (rs > 4) rs = 4;
                if (rsplit_on.get() > 1) rs = std::min(ncu / rem, std::min(8, rsplit_on.get()));     // (tuning aid: D2S_PP_RSPLIT=n: n slices per tile)
                if (rs >= 2) { ks = rs; kps = cdiv(256, rs); tw = tiles - rem; ink = 4; }
            }
            if (!ink && nkt >= 24)
                for (int s = 8; s >= 2; --s)
                    if (nkt % (2 * s) == 0 && nkt / s >= 4 && rem * s <= ncu && (size_t)rem * s * 65536 <= e.part_elems) { ks = s; kps = nkt / s; tw = tiles - rem; break; }
        }
    }
    const int list_max = cdiv(tw, 8) + (ks > 1 ? ((ink & 1) ? cdiv(tiles - tw, 8) * ks : cdiv((tiles - tw) * ks, 8)) : 0);      // longest XCD unit list
    const unsigned grid = 8u * (unsigned)std::max(1, std::min(ncu / 8, list_max));
    GemmEpi e1 = e;
    e1.ksplit = 1;
    if (!e1.bias) e1.bias = pp_const_vec(false);
    if (!e1.scale) e1.scale = pp_const_vec(true);
    if (!e1.deq) e1.deq = pp_const_vec(true);
    if (!e1.bias || !e1.scale || !e1.deq) { set_error("launch_gemm_pp: constant vectors"); return D2S_E_HIP; }
    // half a tile time: K tiles x ~1.5 us + ~8 us of prologue / epilogue (D2S_PP_SKEW: percent of that; 0 = off)
    static const int skew_pct = getenv("D2S_PP_SKEW") ? atoi(getenv("D2S_PP_SKEW")) : 50;
    // (with a K-split tail the lists end in short units: a block without one is not half a tile "lighter")
    const int skew_us = ks > 1 ? 0 : (int)((K / (128 / (int)elem_size(precision)) * 1.5 + 8.0) * skew_pct / 100.0);
#define PP_LAUNCH(T_, KIND_) hipLaunchKernelGGL((gemm_pp_kernel<T_, KIND_>), dim3(grid), dim3(512), 0, st, (const T_*)a.ptr, a.lda, (const T_*)W, M, N, K, Kpad, e1, lxn, skew_us, tw, ks, kps, ink)
    if (precision == D2S_PREC_BF16) {
        if (kind == PP_K_F32) PP_LAUNCH(bf16_t, PP_K_F32); else if (kind == PP_K_QKV) PP_LAUNCH(bf16_t, PP_K_QKV);
        else if (kind == PP_K_GELU) PP_LAUNCH(bf16_t, PP_K_GELU);
        else if (kind == PP_K_F32_LN) PP_LAUNCH(bf16_t, PP_K_F32_LN); else if (kind == PP_K_QKV_LN) PP_LAUNCH(bf16_t, PP_K_QKV_LN);
        else if (kind == PP_K_GELU_LN) PP_LAUNCH(bf16_t, PP_K_GELU_LN); else PP_LAUNCH(bf16_t, PP_K_BF16);
    } else {
        if (kind == PP_K_F32) PP_LAUNCH(fp8_t, PP_K_F32); else if (kind == PP_K_QKV) PP_LAUNCH(fp8_t, PP_K_QKV);
        else if (kind == PP_K_GELU) PP_LAUNCH(fp8_t, PP_K_GELU); else PP_LAUNCH(fp8_t, PP_K_BF16);
    }
#undef PP_LAUNCH
    D2S_CHECK_LAUNCH();
    if (ks > 1 && !ink) {                          // (two-launch K-split path: D2S_PP_INK=0)
        GemmEpi e2 = e;
        e2.ksplit = 1;
        const unsigned rgrid = (unsigned)(tiles - tw) * 64u;
        if (precision == D2S_PREC_BF16) hipLaunchKernelGGL((pp_tail_reduce_kernel<bf16_t>), dim3(rgrid), dim3(256), 0, st, e2, M, N, tw, ks, lxn);
        else hipLaunchKernelGGL((pp_tail_reduce_kernel<fp8_t>), dim3(rgrid), dim3(256), 0, st, e2, M, N, tw, ks, lxn);
        D2S_CHECK_LAUNCH();
    }
    return D2S_OK;
}

}  // namespace d2s

// Units of the in-kernel K-split tail reduce that timed out waiting for their partners since the last clear, on the CURRENT device
// (synchronous: call it after the stream has been synchronised).  0 is the healthy value on a chip the launch has to itself.
extern "C" int d2s_debug_pp_tail_timeouts(int clear, unsigned* count) {
    unsigned v = 0;
    if (count) { D2S_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(d2s::pp_tail_timeouts), sizeof(v))); *count = v; }
    if (clear) { v = 0; D2S_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d2s::pp_tail_timeouts), &v, sizeof(v))); }
    return D2S_OK;
}

#ifdef D2S_PP_TIMING
extern "C" int d2s_pp_timing(int kind, unsigned long long* out) {      // kind >= 0: select + clear; out != null: read 264 x 64 stamps
    if (kind >= -1) { if (hipMemcpyToSymbol(HIP_SYMBOL(d2s::pp_timing_kind), &kind, sizeof(int)) != hipSuccess) return 1; }
    if (out) return hipMemcpyFromSymbol(out, HIP_SYMBOL(d2s::pp_timing), sizeof(unsigned long long) * 264 * 64) != hipSuccess;
    static unsigned long long zeros[264 * 64];
    return hipMemcpyToSymbol(HIP_SYMBOL(d2s::pp_timing), zeros, sizeof(zeros)) != hipSuccess;
}
#endif
