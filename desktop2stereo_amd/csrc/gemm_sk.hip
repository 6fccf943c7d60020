// Streaming linear for SMALL K (gfx950):  C[m, n] = act(sum_k A[m, k] W[n, k] + bias[n]),  K <= 256, bf16 in / out.
//
// Where it is used: the DPT neck's thin linears -- ConvTranspose(k = s) of the reassemble stage (K = C_in = 96 / 192: [M, K] x
// [K, k^2 C] with a pixel-shuffle store) and the fusion stages' 1x1 projections (K = N = F) -- reference call sites HF
// DepthAnythingReassembleLayer.resize / DepthAnythingFeatureFusionLayer.projection behind depth.py:1763-1781.  These launches carry
// 0.2-13 GFLOP for 100-400 MB of traffic: they are HBM-bound (bytes / 8 TB/s = 13-50 us at batch 32), yet on the general tile
// kernel they took 54-92 us: tools/glds_timeline.py --neck32 shows a 128 x 128 block living 12 us for 1.9 us of K loop -- 1.3 us to
// prime its ring, 1.5 us for the first K tile, 4-7 us of epilogue -- six block generations one after the other.
//
// Here a block keeps its W tile (128 output columns x all of K: <= 66 KiB, rows padded by 16 bytes -> conflict-free ds_read_b128) in
// LDS for its whole life and STREAMS row tiles of A through: 4 waves x 16 rows, A fragments straight from global memory in MFMA
// layout (lane = (row, 16-byte K chunk): nothing of A passes through LDS -- every byte of A is used by exactly one wave), the next
// row tile's fragments requested before this tile's MFMAs, the output transposed through a wave-private LDS patch so that each
// lane stores 16 bytes and a wave-instruction covers whole 256-byte row segments (a MAP_SHUFFLE row segment never straddles a tap
// because C_out % 8 == 0).  One prologue per block instead of one per tile; the epilogue of tile t overlaps the A requests of t + 1.
#include "gemm_epi.h"

namespace d2s {

constexpr int SK_BN = 128;                 // output columns per block
constexpr int SK_ROWS = 64;                // rows per block iteration: 4 waves x 16

template <int KS /* K steps of 32 elements: K = 32 KS <= 256 */>
__global__ void __launch_bounds__(256)
gemm_sk_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, int Kpad, int M, int N, GemmEpi e) {
    constexpr int WROW = KS * 4 + 1;                                   // 16-byte chunks per W row in LDS (padded)
    __shared__ __attribute__((aligned(16))) u32x4 lds[SK_BN * WROW + 4 * 16 * 17];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bn0 = blockIdx.x * SK_BN;
    // ---- the block's W tile -> LDS, once
    for (int c = tid; c < SK_BN * KS * 4; c += 256) {
        const int r = c / (KS * 4), k = c - r * (KS * 4);
        lds[r * WROW + k] = *(const u32x4*)(W + (long)(bn0 + r) * Kpad + k * 8);
    }
    u32x4* patch = lds + SK_BN * WROW + wid * (16 * 17);               // 16 rows x (16 + 1) chunks
    // per-lane constants of the store side: lane -> (row sub-index l >> 4 of a 4-row pass, columns (l & 15) * 8 .. + 7)
    const int sc = (lane & 15) * 8, sr = lane >> 4;
    long col_part;                                                      // output offset = row part (m) + column part (n): gemm_epi.h epi_out_offset
    {
        const int n0 = bn0 + sc;
        if (e.map == MAP_SHUFFLE) { const int tap = n0 / e.cout, co = n0 - tap * e.cout, ky = tap / e.ks, kx = tap - ky * e.ks; col_part = ((long)ky * ((long)e.gw * e.ks) + kx) * e.cout + co; }
        else col_part = n0;
    }
    f32x4 bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = e.bias ? *(const f32x4*)(e.bias + bn0 + j * 16 + fg * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int tiles_m = (M + SK_ROWS - 1) / SK_ROWS;
    int mt = blockIdx.y;
    u32x4 a_nxt[KS];
    auto load_a = [&](int t) {
        int m = t * SK_ROWS + wid * 16 + fr; m = m < M ? m : M - 1;
        const bf16_t* p = A + (long)m * lda + fg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_nxt[ks] = *(const u32x4*)(p + ks * 32);
    };
    if (mt < tiles_m) load_a(mt);
    for (; mt < tiles_m; mt += gridDim.y) {
        u32x4 a_cur[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
        if (mt + (int)gridDim.y < tiles_m) load_a(mt + gridDim.y);      // in flight under this tile's MFMAs and stores
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) mma_chunk(acc[j], lds[(j * 16 + fr) * WROW + ks * 4 + fg], a_cur[ks], bf16_t());
        // lane (fr, fg) holds row fr, columns j * 16 + fg * 4 .. + 3 of every j: -> patch[row][chunk = 2 j + (fg >> 1)], half (fg & 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc[j] + bias[j];
            if (e.act == ACT_RELU) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            uint2 t;
            t.x = pk_bf16(v[0], v[1]); t.y = pk_bf16(v[2], v[3]);
            ((uint2*)(patch + fr * 17 + 2 * j + (fg >> 1)))[fg & 1] = t;
        }
        // wave-private patch: the LDS operations of one wave execute in order, no barrier
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 4 + sr;
            const int m = mt * SK_ROWS + wid * 16 + row;
            const u32x4 v = patch[row * 17 + (lane & 15)];
            if (m < M && bn0 + sc < N) {
                long row_part;
                if (e.map == MAP_SHUFFLE) {
                    const int x = m % e.gw, q = m / e.gw, y = q % e.gh, b = q / e.gh;
                    row_part = (((long)b * e.gh * e.ks + (long)y * e.ks) * ((long)e.gw * e.ks) + (long)x * e.ks) * e.cout;
                } else row_part = (long)m * e.ldc;
                *(u32x4*)((bf16_t*)e.out + row_part + col_part) = v;
            }
        }
    }
}

// what the streaming kernel takes: plain bf16 rows in and out, whole K in registers, no residuals / LayerNorm folding / split K
bool sk_supported(int precision, const GemmA& a, int M, int N, int K, int Kpad, const GemmEpi& e) {
    static EnvInt off{"D2S_NO_SK", 0};
    if (off.get() || precision != D2S_PREC_BF16 || a.mode != A_PLAIN || a.relu || a.bx3) return false;
    if (K < 32 || K > 256 || (K & 31) || (N & (SK_BN - 1)) || (a.lda & 7) || M < SK_ROWS) return false;
    if (e.out_type != OUT_T && e.out_type != OUT_BF16) return false;
    if (e.map == MAP_SHUFFLE) { if ((e.cout & 7) || e.gw <= 0 || e.gh <= 0 || e.ks <= 0) return false; }
    else if (e.map != MAP_ROWS || e.rows_per_img || (e.ldc & 7)) return false;
    if (e.res1 || e.res2 || e.scale || e.deq || e.ln_stats || e.ln_csum || e.stats_out || e.out2 || (e.act != ACT_NONE && e.act != ACT_RELU)) return false;
    return true;
}

int launch_gemm_sk(const GemmA& a, const void* W, int M, int N, int K, int Kpad, const GemmEpi& e, hipStream_t st) {
    if (!sk_supported(D2S_PREC_BF16, a, M, N, K, Kpad, e)) { set_error("launch_gemm_sk: unsupported problem"); return D2S_E_UNSUPPORTED; }
    static const int ncu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
    const int cols = N / SK_BN, tiles_m = cdiv(M, SK_ROWS);
    // two blocks per CU (<= 83 KiB of LDS each); every block streams >= 2 row tiles where there are that many
    int per_col = std::max(1, std::min(tiles_m, (2 * ncu) / cols));
    GemmEpi e1 = e; e1.ksplit = 1;
    const dim3 grid(cols, per_col), block(256);
#define SK_LAUNCH(KS_) hipLaunchKernelGGL((gemm_sk_kernel<KS_>), grid, block, 0, st, (const bf16_t*)a.ptr, a.lda, (const bf16_t*)W, Kpad, M, N, e1)
    switch (K / 32) {
        case 1: SK_LAUNCH(1); break; case 2: SK_LAUNCH(2); break; case 3: SK_LAUNCH(3); break; case 4: SK_LAUNCH(4); break;
        case 5: SK_LAUNCH(5); break; case 6: SK_LAUNCH(6); break; case 7: SK_LAUNCH(7); break; default: SK_LAUNCH(8); break;
    }
#undef SK_LAUNCH
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}

}  // namespace d2s
