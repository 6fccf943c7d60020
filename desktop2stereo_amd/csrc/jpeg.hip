// MJPEG sink: baseline JPEG encode of the packed stereo frame, on the device.
//
// Replaces `cv2.imencode('.jpg', bgr, [IMWRITE_JPEG_QUALITY, q])` on the float32 frame make_sbs returns
// (reference streamer.py:249-256, 285-291): convertTo(CV_8U) (round-half-even, saturate) followed by
// libjpeg(-turbo) with jpeg_set_defaults + jpeg_set_quality(q, TRUE) — YCbCr 4:2:0, slow-integer FDCT,
// Annex-K Huffman tables, no restart markers, JFIF 1.01 header.  Every stage below is the integer
// arithmetic libjpeg publishes (jccolor.c, jcsample.c, jfdctint.c, jcdctmgr.c, jccoefct.c, jchuff.c,
// jcmarker.c), so the stream is BYTE-IDENTICAL to libjpeg-turbo's (oracle/jpeg_oracle.py pins that).
//
// Stages (all HBM-/latency-bound byte work; one grid dimension = frame; no single-block pass, no host sync):
//   1 jpeg_dct_kernel            8 MCUs (128x16 px) per block: RGB -> YCbCr (+h2v2), FDCT, quantise, zigzag, dummy
//                                blocks; writes int16 coefficients [mcu][6][64] and the DCs [mcu][8]
//   2 jpeg_count_kernel          (round 5) the AC bit count of a block is taken in stage 1 while the block is in LDS; this pass adds the
//                                DC term (needs the predecessor's DC) -> bits per block, bits per 256-block tile, and (round 6) the first
//                                bit of every block within its tile
//   3 jpeg_zero_kernel           zero the words of the (unstuffed) bit stream that will be used; publish the total and (round 6) the
//                                first bit of every tile
//   4 jpeg_emit8_kernel          (round 6, up to a few frames per launch) eight lanes per 8x8 block, runs from the block's non-zero mask,
//                                the thread block's span of the stream assembled in LDS and written out as whole words;
//     jpeg_entropy_kernel<true>  (larger batches) one lane per block walks it: offset = tile prefix + in-block scan; whole words are
//                                plain stores, the two edge words of a block are atomic ORs
//   5 jpeg_ffcount_kernel        0xFF bytes per 64-byte chunk and per 256-chunk tile
//   6 jpeg_stuff_kernel          byte-stuffed copy behind the header (+ header, EOI, size from block 0), staged through LDS so that
//                                global memory sees aligned 16-byte stores (round 5)
#include "common.h"
#include <string.h>
#include <initializer_list>

namespace d2s {

// ---- Annex K (ITU-T T.81) ---------------------------------------------------------------------
static const uint8_t STD_Q[2][64] = {
    {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
     18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
    {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};
static const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const uint8_t DC_BITS[2][16] = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}};
static const uint8_t AC_BITS[2][16] = {{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
static const uint8_t AC_VALS[2][162] = {
    {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
     0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
     0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
     0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
     0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
     0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
     0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa},
    {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
     0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
     0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
     0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
     0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
     0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
     0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa}};

constexpr int HDR_LEN = 623;            // SOI + APP0 + 2 DQT + SOF0 + 4 DHT + SOS
constexpr int BLK_WORDS = 54;           // worst case of one 8x8 block: 22 + 63*26 = 1660 bits < 54 words

// Everything the kernels need that depends on `quality` or is a table: passed BY VALUE (kernarg), staged to LDS.
struct JpegTables {
    uint16_t q8[2][64];        // quantiser << 3, indexed by ZIGZAG position k (jcdctmgr.c: islow divisors are qtbl << 3)
    uint32_t magic[2][64];     // floor(2^32 / q8) + 1: n / q8 == (n * magic) >> 32 for n < 2^20
    uint8_t nat2zig[64];       // natural index -> zigzag position
    uint32_t ac[2][256];       // (length << 16) | code of the AC symbol (run << 4 | size); 0 = unused
    uint32_t dc[2][12];        // same for the DC size categories
    uint8_t header[HDR_LEN + 1];
};

static void derive(const uint8_t* bits, const uint8_t* vals, uint32_t* table) {       // jchuff.c jpeg_make_c_derived_tbl
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len - 1]; ++i) table[vals[k++]] = ((uint32_t)len << 16) | code++;
        code <<= 1;
    }
}

static void make_tables(int H, int W, int quality, JpegTables& t) {
    memset(&t, 0, sizeof(t));
    int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    int scale = q < 50 ? 5000 / q : 200 - 2 * q;                    // jcparam.c jpeg_quality_scaling
    uint8_t qt[2][64];
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < 64; ++i) {
            long v = ((long)STD_Q[c][i] * scale + 50) / 100;        // jpeg_add_quant_table, force_baseline
            qt[c][i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
        }
    for (int k = 0; k < 64; ++k) t.nat2zig[ZIGZAG[k]] = (uint8_t)k;
    for (int c = 0; c < 2; ++c)
        for (int k = 0; k < 64; ++k) {
            uint32_t d = (uint32_t)qt[c][ZIGZAG[k]] << 3;
            t.q8[c][k] = (uint16_t)d;
            t.magic[c][k] = (uint32_t)((1ull << 32) / d) + 1u;
        }
    static const uint8_t dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    for (int c = 0; c < 2; ++c) {
        derive(DC_BITS[c], dc_vals, t.dc[c]);
        derive(AC_BITS[c], AC_VALS[c], t.ac[c]);
    }
    uint8_t* p = t.header;                                           // jcmarker.c write_file_header / frame / scan
    auto put = [&](std::initializer_list<int> b) { for (int v : b) *p++ = (uint8_t)v; };
    put({0xff, 0xd8, 0xff, 0xe0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    for (int c = 0; c < 2; ++c) {
        put({0xff, 0xdb, 0x00, 0x43, c});
        for (int k = 0; k < 64; ++k) *p++ = qt[c][ZIGZAG[k]];
    }
    put({0xff, 0xc0, 0x00, 0x11, 0x08, H >> 8, H & 255, W >> 8, W & 255, 0x03, 0x01, 0x22, 0x00, 0x02, 0x11, 0x01, 0x03, 0x11, 0x01});
    for (int c = 0; c < 2; ++c) {
        put({0xff, 0xc4, 0x00, 0x1f, c});
        for (int i = 0; i < 16; ++i) *p++ = DC_BITS[c][i];
        for (int i = 0; i < 12; ++i) *p++ = dc_vals[i];
        put({0xff, 0xc4, 0x00, 0xb5, 0x10 | c});
        for (int i = 0; i < 16; ++i) *p++ = AC_BITS[c][i];
        for (int i = 0; i < 162; ++i) *p++ = AC_VALS[c][i];
    }
    put({0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00});
}

__host__ __device__ static inline int cdiv_dev(long a, long b) { return (int)((a + b - 1) / b); }

struct JpegGeom {
    int H, W, mr, mc, nmcu;          // MCU rows / cols (16x16 px)
    int ybw, ybh;                    // real luma blocks across / down
    int He;                          // H rounded up to even (the rows the chroma planes are derived from)
    long ws_frame;                   // workspace bytes per frame
    long off_blkbits, off_dcs, off_tiles, off_total, off_stream, off_ffcnt, off_fftiles, off_blkoff, off_tpre;
    long cap_words;                  // words of the unstuffed stream buffer
    int n_chunks;                    // 64-byte chunks of it
};

static JpegGeom make_geom(int H, int W) {
    JpegGeom g;
    g.H = H; g.W = W;
    g.mr = (H + 15) / 16; g.mc = (W + 15) / 16; g.nmcu = g.mr * g.mc;
    g.ybw = (W + 7) / 8; g.ybh = (H + 7) / 8;
    g.He = H + (H & 1);
    auto al = [](long v) { return (v + 255) / 256 * 256; };
    long o = al((long)g.nmcu * 6 * 64 * 2);
    g.off_blkbits = o; o = al(o + (long)g.nmcu * 6 * 2 + 16);        // bits of every block (uint16: <= 1660)
    g.off_dcs = o; o = al(o + (long)g.nmcu * 8 * 2);                 // quantised DC of the 6 blocks of every MCU ([mcu][8] int16)
    g.off_tiles = o; o = al(o + ((long)g.nmcu * 6 / 256 + 2) * 4);    // bits per 256-block tile
    g.off_total = o; o = al(o + 16);                                  // byte-padded total bits of the frame
    g.cap_words = ((long)g.nmcu * 6 * BLK_WORDS + 16 + 15) / 16 * 16;
    g.n_chunks = (int)(g.cap_words / 16);
    g.off_stream = o; o = al(o + g.cap_words * 4);
    g.off_ffcnt = o; o = al(o + (long)(g.n_chunks + 1) * 4);
    g.off_fftiles = o; o = al(o + ((long)g.n_chunks / 256 + 2) * 4);
    g.off_blkoff = o; o = al(o + (long)g.nmcu * 6 * 4 + 16);         // first bit of every block within its 256-block tile (round 6)
    g.off_tpre = o; o = al(o + ((long)g.nmcu * 6 / 256 + 2) * 4);     // first bit of every tile (round 6)
    g.ws_frame = o;
    return g;
}

// ---- stage 1 -----------------------------------------------------------------------------------
#ifndef D2S_DCT_MCUS
#define D2S_DCT_MCUS 8             // (round 6, 3840x1080 q90: 16 -> 28.1 us, 8 -> 27.1, 4 -> 25.8 at one frame; 16 frames: 32.1 / 29.9 / 32.5 us per frame)
#endif
constexpr int DCT_MCUS = D2S_DCT_MCUS;   // MCUs per thread block (128 px x 16 rows)
constexpr int DCT_BLOCKS = DCT_MCUS * 6;

template <int FMT>
__device__ __forceinline__ void load_rgb(const void* frame, long idx, int& r, int& g, int& b) {
    if (FMT == D2S_FMT_U8_HWC) {
        const uint8_t* p = (const uint8_t*)frame + idx * 3;
        r = p[0]; g = p[1]; b = p[2];
    } else {                                                          // float 0..255: cv2 convertTo(CV_8U) = rint + saturate
        const float* p = (const float*)frame + idx * 3;
        r = (int)fminf(fmaxf(rintf(p[0]), 0.f), 255.f);
        g = (int)fminf(fmaxf(rintf(p[1]), 0.f), 255.f);
        b = (int)fminf(fmaxf(rintf(p[2]), 0.f), 255.f);
    }
}
// 4 consecutive pixels of row y starting at column xs (columns clamped to W-1).  `wide`: uint8 frame with W % 4 == 0
// and a 4-byte aligned base, so the 12 bytes are three aligned dwords.
template <int FMT>
__device__ __forceinline__ void load_row4(const void* frame, int W, int y, int xs, bool wide, int* R, int* G, int* B) {
    if (FMT == D2S_FMT_U8_HWC && wide && xs + 3 < W) {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)frame + ((long)y * W + xs) * 3);
        uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        R[0] = w0 & 255; G[0] = (w0 >> 8) & 255; B[0] = (w0 >> 16) & 255;
        R[1] = w0 >> 24; G[1] = w1 & 255; B[1] = (w1 >> 8) & 255;
        R[2] = (w1 >> 16) & 255; G[2] = w1 >> 24; B[2] = w2 & 255;
        R[3] = (w2 >> 8) & 255; G[3] = (w2 >> 16) & 255; B[3] = w2 >> 24;
    } else {
        for (int i = 0; i < 4; ++i) load_rgb<FMT>(frame, (long)y * W + min(xs + i, W - 1), R[i], G[i], B[i]);
    }
}
// jccolor.c rgb_ycc_convert (SCALEBITS 16)
__device__ __forceinline__ int ycc_y(int r, int g, int b) { return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16; }
__device__ __forceinline__ int ycc_cb(int r, int g, int b) { return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16; }
__device__ __forceinline__ int ycc_cr(int r, int g, int b) { return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16; }

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 8-point pass.  FIRST: row pass (outputs scaled up by PASS1_BITS); else column pass.
template <bool FIRST>
__device__ __forceinline__ void fdct8(int* d) {
    constexpr int CB = 13, PB = 2;
    int t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6];
    int t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
    int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int SH = FIRST ? CB - PB : CB + PB;
    if (FIRST) { d[0] = (t10 + t11) << PB; d[4] = (t10 - t11) << PB; }
    else       { d[0] = descale(t10 + t11, PB); d[4] = descale(t10 - t11, PB); }
    int z1 = (t12 + t13) * 4433;
    d[2] = descale(z1 + t13 * 6270, SH);
    d[6] = descale(z1 - t12 * 15137, SH);
    z1 = t4 + t7; int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7, z5 = (z3 + z4) * 9633;
    t4 *= 2446; t5 *= 16819; t6 *= 25172; t7 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * -16069 + z5; z4 = z4 * -3196 + z5;
    d[7] = descale(t4 + z1 + z3, SH);
    d[5] = descale(t5 + z2 + z4, SH);
    d[3] = descale(t6 + z2 + z3, SH);
    d[1] = descale(t7 + z1 + z4, SH);
}

template <int FMT>
__global__ void __launch_bounds__(256)
jpeg_dct_kernel(const void* __restrict__ frames, uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ __attribute__((aligned(16))) uint8_t sY[16][DCT_MCUS * 16];
    __shared__ __attribute__((aligned(16))) uint8_t sC[2][8][DCT_MCUS * 8];
    // (round 6: rows of 72 shorts = 36 dwords.  With 64-short rows the eight 8x8 blocks of a wave sit 32 dwords apart -- two bank
    //  positions for eight blocks -- and the column reads of pass 2 / the zigzag writes ran 4-way conflicted: PMC, LDS bank conflict
    //  cycles 0.48 of the active ones.  36 b mod 64 puts the eight blocks on eight disjoint groups of four banks.)
    __shared__ __attribute__((aligned(16))) short sW[DCT_BLOCKS][72];       // row-pass output: |x| <= 1024 << PASS1_BITS, fits int16
    __shared__ __attribute__((aligned(16))) short sZ[DCT_BLOCKS][72];
    __shared__ uint16_t sQ8[2][64];
    __shared__ uint32_t sMagic[2][64];
    __shared__ uint8_t sN2Z[64];
    __shared__ uint8_t sAcLen[2][256];         // code length of every AC symbol (round 5: the AC bit count happens here)

    const int tid = threadIdx.x;
    const int mrow = blockIdx.y, mcol0 = blockIdx.x * DCT_MCUS, f = blockIdx.z;
    const int nm = min(DCT_MCUS, g.mc - mcol0);
    const char* frame = (const char*)frames + (long)f * g.H * g.W * 3 * (FMT == D2S_FMT_U8_HWC ? 1 : 4);
    uint8_t* wsf = ws + (long)f * g.ws_frame;

    if (tid < 128) { sQ8[tid >> 6][tid & 63] = tb.q8[tid >> 6][tid & 63]; sMagic[tid >> 6][tid & 63] = tb.magic[tid >> 6][tid & 63]; }
    if (tid < 64) sN2Z[tid] = tb.nat2zig[tid];
    for (int i = tid; i < 512; i += 256) sAcLen[i >> 8][i & 255] = (uint8_t)(tb.ac[i >> 8][i & 255] >> 16);

    // colour conversion + h2v2: 4 pixels x 2 rows (two 2x2 quads) per step.  Columns are edge-replicated on the INPUT
    // (expand_right_edge); rows: luma replicates the last row, chroma replicates its last DOWNSAMPLED row (jcprepct.c),
    // whose sources are rows He-2, min(He-1, H-1).
    const int x0 = mcol0 * 16, y0 = mrow * 16;
    const bool wide = FMT == D2S_FMT_U8_HWC && (g.W & 3) == 0 && (((uintptr_t)frames) & 3) == 0;
    for (int it = tid; it < 8 * nm * 4; it += 256) {
        const int r = it / (nm * 4), cg = it % (nm * 4), xs = x0 + 4 * cg;
        const int ya = min(y0 + 2 * r, g.H - 1), yb = min(y0 + 2 * r + 1, g.H - 1);
        int R[2][4], G[2][4], B[2][4];
        load_row4<FMT>(frame, g.W, ya, xs, wide, R[0], G[0], B[0]);
        load_row4<FMT>(frame, g.W, yb, xs, wide, R[1], G[1], B[1]);
        for (int k = 0; k < 2; ++k) {
            uint32_t y4 = 0;
            for (int i = 0; i < 4; ++i) y4 |= (uint32_t)ycc_y(R[k][i], G[k][i], B[k][i]) << (8 * i);
            *(uint32_t*)&sY[2 * r + k][4 * cg] = y4;
        }
        const int cy = min(mrow * 8 + r, g.He / 2 - 1);               // chroma row this sample replicates
        const int ca = 2 * cy, cb2 = min(2 * cy + 1, g.H - 1);
        if (ca != ya || cb2 != yb) {
            load_row4<FMT>(frame, g.W, ca, xs, wide, R[0], G[0], B[0]);
            load_row4<FMT>(frame, g.W, cb2, xs, wide, R[1], G[1], B[1]);
        }
        uint32_t cb16 = 0, cr16 = 0;
        for (int h = 0; h < 2; ++h) {                                 // jcsample.c h2v2_downsample: bias 1,2,1,2,... along the row
            int sb = 1 + h, sr = 1 + h;
            for (int k = 0; k < 2; ++k)
                for (int i = 2 * h; i < 2 * h + 2; ++i) { sb += ycc_cb(R[k][i], G[k][i], B[k][i]); sr += ycc_cr(R[k][i], G[k][i], B[k][i]); }
            cb16 |= (uint32_t)(sb >> 2) << (8 * h);
            cr16 |= (uint32_t)(sr >> 2) << (8 * h);
        }
        *(uint16_t*)&sC[0][r][2 * cg] = (uint16_t)cb16;
        *(uint16_t*)&sC[1][r][2 * cg] = (uint16_t)cr16;
    }
    __syncthreads();

    // FDCT pass 1 (rows): task = (block, row).  Block order inside an MCU: Y00 Y01 Y10 Y11 Cb Cr.
    for (int task = tid; task < nm * 6 * 8; task += 256) {
        int blk = task >> 3, r = task & 7, m = blk / 6, b = blk % 6;
        int d[8];
        const uint2 px = b < 4 ? *(const uint2*)&sY[(b >> 1) * 8 + r][m * 16 + (b & 1) * 8] : *(const uint2*)&sC[b - 4][r][m * 8];
        for (int i = 0; i < 4; ++i) { d[i] = (int)((px.x >> (8 * i)) & 255) - 128; d[4 + i] = (int)((px.y >> (8 * i)) & 255) - 128; }
        fdct8<true>(d);
        uint4 pk;                                                     // the row's eight int16 as one 16-byte store
        pk.x = ((uint32_t)d[0] & 0xffffu) | ((uint32_t)d[1] << 16); pk.y = ((uint32_t)d[2] & 0xffffu) | ((uint32_t)d[3] << 16);
        pk.z = ((uint32_t)d[4] & 0xffffu) | ((uint32_t)d[5] << 16); pk.w = ((uint32_t)d[6] & 0xffffu) | ((uint32_t)d[7] << 16);
        *(uint4*)&sW[blk][r * 8] = pk;
    }
    __syncthreads();
    // pass 2 (columns) + quantise (jcdctmgr.c: sign * ((|x| + q8/2) / q8)) + zigzag
    for (int task = tid; task < nm * 6 * 8; task += 256) {
        int blk = task >> 3, c = task & 7, tbl = (blk % 6) >= 4;
        int d[8];
        for (int i = 0; i < 8; ++i) d[i] = sW[blk][i * 8 + c];
        fdct8<false>(d);
        for (int i = 0; i < 8; ++i) {
            int k = sN2Z[i * 8 + c];
            uint32_t q8 = sQ8[tbl][k];
            uint32_t a = (uint32_t)(d[i] < 0 ? -d[i] : d[i]) + (q8 >> 1);
            uint32_t qv = (uint32_t)(((unsigned long long)a * sMagic[tbl][k]) >> 32);
            sZ[blk][k] = (short)(d[i] < 0 ? -(int)qv : (int)qv);
        }
    }
    __syncthreads();

    // Round 5: the AC half of jchuff.c's bit count, here, while the quantised block is still in LDS -- the COUNT pass used to re-read
    // all 12.5 MB of coefficients (one lane per block, eight 16-byte loads at a 128-byte lane stride) just to add code lengths.  Eight
    // lanes per block, eight zigzag positions each: a non-zero coefficient at position k after a run r of zeros costs
    // (r >> 4) len(ZRL) + len(ac[(r & 15) << 4 | nbits]) + nbits, r = k - 1 - (last non-zero position before k, or 0 = the DC slot);
    // the run crosses lanes through a max-scan of "last non-zero position" over the block's eight lanes; + len(EOB) unless
    // position 63 is non-zero.  The DC term needs the previous block's DC (another thread block's): jpeg_count_kernel adds it.
    uint16_t* acbits = (uint16_t*)(wsf + g.off_blkbits);
    const bool row1 = (2 * mrow + 1) < g.ybh;
    for (int task = tid; task < ((nm * 6 * 8 + 63) & ~63); task += 256) {          // (whole waves: the shuffles need every lane)
        const bool live = task < nm * 6 * 8;
        const int blk = live ? task >> 3 : 0, seg = task & 7, m = blk / 6, b = blk % 6, tbl = b >= 4;
        const uint4 c4 = *(const uint4*)&sZ[blk][8 * seg];
        const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
        int v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = (int)(short)(cw[i] & 0xffffu); v[2 * i + 1] = (int)(short)(cw[i] >> 16); }
        if (seg == 0) v[0] = 0;                                                     // position 0 is the DC
        int last = 0;                                                               // last non-zero position of this segment (0: none)
#pragma unroll
        for (int i = 0; i < 8; ++i) if (v[i] != 0) last = 8 * seg + i;
        int prev = last;                                                            // inclusive max-scan over the 8 lanes of the block
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { const int t = __shfl_up(prev, o, 8); if (seg >= o) prev = max(prev, t); }
        int before = __shfl_up(prev, 1, 8);                                         // exclusive: last non-zero position before this segment
        if (seg == 0) before = 0;
        uint32_t bits = 0;
        const uint32_t zrl = sAcLen[tbl][0xF0];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (v[i] != 0) {
                const int k = 8 * seg + i, run = k - 1 - before;
                const int a = v[i] < 0 ? -v[i] : v[i], nb = 32 - __builtin_clz(a);
                bits += (uint32_t)(run >> 4) * zrl + sAcLen[tbl][((run & 15) << 4) | nb] + (uint32_t)nb;
                before = k;
            }
        }
        if (seg == 7 && before != 63) bits += sAcLen[tbl][0];                       // end of block
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) bits += __shfl_xor(bits, o, 8);
        if (live && seg == 0) {
            // dummy blocks (below) carry the DC of their source and no AC: EOB only
            const bool col1 = (2 * (mcol0 + m) + 1) < g.ybw;
            const int s1 = col1 ? 1 : 0;
            int src = b;
            if (b == 1) src = s1;
            else if (b == 2) src = row1 ? 2 : s1;
            else if (b == 3) src = row1 ? (col1 ? 3 : 2) : s1;
            const long mcu = (long)mrow * g.mc + mcol0 + m;
            acbits[mcu * 6 + b] = (uint16_t)(src == b ? bits : (uint32_t)sAcLen[tbl][0]);
        }
    }

    // dummy blocks (jccoefct.c compress_data) + coefficient store: 32 threads per block, two zigzag positions each
    uint32_t* coefs = (uint32_t*)wsf;
    for (int task = tid; task < nm * 6 * 32; task += 256) {
        const int blk = task >> 5, k2 = task & 31, m = blk / 6, b = blk % 6;
        const bool col1 = (2 * (mcol0 + m) + 1) < g.ybw;
        const int s1 = col1 ? 1 : 0;
        int src = b;
        if (b == 1) src = s1;
        else if (b == 2) src = row1 ? 2 : s1;
        else if (b == 3) src = row1 ? (col1 ? 3 : 2) : s1;
        uint32_t v = (src == b) ? *(const uint32_t*)&sZ[blk][2 * k2] : (k2 == 0 ? (uint32_t)(uint16_t)sZ[m * 6 + src][0] : 0u);
        const long mcu = (long)mrow * g.mc + mcol0 + m;
        coefs[(mcu * 6 + b) * 32 + k2] = v;
        if (k2 == 0) ((short*)(wsf + g.off_dcs))[mcu * 8 + b] = (short)(v & 0xffffu);
    }
}

// ---- stages 2-5: entropy coding, one LANE per 8x8 block -------------------------------------------
// Bit order everywhere: stream bit p lives in word p >> 5 at bit 31 - (p & 31) (MSB first).
// A lane keeps its block's 64 coefficients in registers (8 x 16-byte loads) and walks them in zigzag order exactly
// like jchuff.c encode_one_block; COUNT instantiation adds code lengths, EMIT instantiation streams the codes to the
// block's bit offset.  Words a block covers completely are plain stores; its first and last (possibly shared with the
// neighbouring blocks) are atomic ORs into the zeroed stream.
struct BitSink {
    uint32_t* stream;      // EMIT only
    long w;                // current word
    uint32_t acc;          // bits already placed in the current word (from the MSB)
    int fill;              // how many
    bool first;            // the current word is the block's first (shared) word
    uint32_t count;        // COUNT only
};

template <bool EMIT>
__device__ __forceinline__ void put(BitSink& s, uint32_t val, int len) {          // len <= 27
    if (!EMIT) { s.count += (uint32_t)len; return; }
    unsigned long long win = ((unsigned long long)s.acc << 32) | ((unsigned long long)val << (64 - s.fill - len));
    s.fill += len;
    if (s.fill >= 32) {
        uint32_t word = (uint32_t)(win >> 32);
        if (s.first) { if (word) atomicOr(&s.stream[s.w], word); s.first = false; }
        else s.stream[s.w] = word;
        ++s.w;
        s.acc = (uint32_t)win;
        s.fill -= 32;
    } else {
        s.acc = (uint32_t)(win >> 32);
    }
}

template <bool EMIT>
__device__ __forceinline__ void coef_step(BitSink& s, int v, int& run, const uint32_t* ac) {
    if (v == 0) { ++run; return; }
    while (run > 15) { put<EMIT>(s, ac[0xF0] & 0xffffu, (int)(ac[0xF0] >> 16)); run -= 16; }
    const int a = v < 0 ? -v : v, t2 = v < 0 ? v - 1 : v;
    const int nb = 32 - __builtin_clz(a);
    const uint32_t e = ac[(run << 4) | nb];
    put<EMIT>(s, ((e & 0xffffu) << nb) | ((uint32_t)t2 & ((1u << nb) - 1u)), (int)(e >> 16) + nb);
    run = 0;
}

// one block: q = its 64 int16 coefficients in zigzag order; dcdiff = DC - predictor
template <bool EMIT>
__device__ __forceinline__ void encode_block(BitSink& s, const uint4 (&q)[8], int dcdiff, const uint32_t* dc, const uint32_t* ac) {
    {
        const int a = dcdiff < 0 ? -dcdiff : dcdiff, t2 = dcdiff < 0 ? dcdiff - 1 : dcdiff;
        const int nb = a ? 32 - __builtin_clz(a) : 0;
        const uint32_t e = dc[nb];
        put<EMIT>(s, ((e & 0xffffu) << nb) | ((uint32_t)t2 & ((1u << nb) - 1u)), (int)(e >> 16) + nb);
    }
    int run = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j > 0) coef_step<EMIT>(s, (int)(short)(w[j] & 0xffffu), run, ac);           // k = 8i + 2j (k = 0 is the DC)
            coef_step<EMIT>(s, (int)(short)(w[j] >> 16), run, ac);
        }
    }
    if (run > 0) put<EMIT>(s, ac[0] & 0xffffu, (int)(ac[0] >> 16));                            // end of block
}

// DC predictor of block b of MCU m: the previous block of the same component in scan order
__device__ __forceinline__ int dc_pred(const short* dcs, long m, int b) {
    if (b >= 1 && b <= 3) return dcs[m * 8 + b - 1];
    return m > 0 ? dcs[(m - 1) * 8 + (b == 0 ? 3 : b)] : 0;
}

// 256-thread block helpers (deterministic: no atomics).  s_w: >= 8 words of LDS.
__device__ __forceinline__ uint32_t block256_sum(uint32_t v, uint32_t* s_w) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__device__ __forceinline__ uint32_t block256_exscan(uint32_t v, uint32_t* s_w) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    __syncthreads();
    if (lane == 63) s_w[4 + wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < wv; ++i) base += s_w[4 + i];
    return base + inc - v;
}
// prefix = sum of tiles[0..mine), total = sum of tiles[0..n): every block derives its own offset from the per-tile sums
__device__ __forceinline__ void tile_prefix(const uint32_t* tiles, int n, int mine, uint32_t* s_w, uint32_t& prefix, uint32_t& total) {
    uint32_t p = 0, t = 0;
    for (int i = threadIdx.x; i < n; i += 256) { uint32_t v = tiles[i]; t += v; if (i < mine) p += v; }
    prefix = block256_sum(p, s_w);
    total = block256_sum(t, s_w);
}

template <bool EMIT>
__global__ void __launch_bounds__(256)
jpeg_entropy_kernel(uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ uint32_t sAc[2][256];
    __shared__ uint32_t sDc[2][12];
    __shared__ uint32_t s_w[8];
    const int tid = threadIdx.x;
    for (int i = tid; i < 512; i += 256) sAc[i >> 8][i & 255] = tb.ac[i >> 8][i & 255];
    if (tid < 24) sDc[tid / 12][tid % 12] = tb.dc[tid / 12][tid % 12];
    __syncthreads();
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const long nblk = (long)g.nmcu * 6;
    const long gb = (long)blockIdx.x * 256 + tid;                     // block index in scan order: 6 * mcu + b
    const bool active = gb < nblk;
    const long gbc = active ? gb : nblk - 1;
    const long m = gbc / 6;
    const int b = (int)(gbc % 6), tbl = b >= 4;
    const uint4* src = (const uint4*)wsf + gbc * 8;
    uint4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = src[i];
    const short* dcs = (const short*)(wsf + g.off_dcs);
    const int dcdiff = (int)(short)(q[0].x & 0xffffu) - dc_pred(dcs, m, b);
    uint16_t* blkbits = (uint16_t*)(wsf + g.off_blkbits);
    uint32_t* tiles = (uint32_t*)(wsf + g.off_tiles);
    BitSink s;
    s.count = 0;
    if (!EMIT) {
        encode_block<false>(s, q, dcdiff, sDc[tbl], sAc[tbl]);
        if (active) blkbits[gb] = (uint16_t)s.count;
        const uint32_t sum = block256_sum(active ? s.count : 0u, s_w);
        if (tid == 0) tiles[blockIdx.x] = sum;
        return;
    }
    uint32_t prefix, total;
    tile_prefix(tiles, (int)gridDim.x, (int)blockIdx.x, s_w, prefix, total);
    const uint32_t G = prefix + block256_exscan(active ? (uint32_t)blkbits[gb] : 0u, s_w);
    if (!active) return;
    s.stream = (uint32_t*)(wsf + g.off_stream);
    s.w = (long)(G >> 5);
    s.acc = 0;
    s.fill = (int)(G & 31u);
    s.first = true;
    encode_block<true>(s, q, dcdiff, sDc[tbl], sAc[tbl]);
    if (gb == nblk - 1) {                                             // jchuff.c flush_bits: fill the last byte with ones
        const int pad = (int)((8u - ((uint32_t)s.fill & 7u)) & 7u);
        if (pad) put<true>(s, (1u << pad) - 1u, pad);
    }
    if (s.fill > 0 && s.acc) atomicOr(&s.stream[s.w], s.acc);
}

// stage 4, round 6: EIGHT lanes per 8x8 block.  The one-lane-per-block walk above is a serial chain of 63 table look-ups and
// data-dependent word stores per lane, on 97 200 threads for a 3840x1080 frame (0.37 of the chip's thread slots): latency, 24.6 us.
// Here lane j of a group of eight owns the zigzag coefficients 8j .. 8j+7 (one 16-byte load):
//   * the group ORs its lanes' non-zero masks into one 64-bit mask; the zero run in front of coefficient k is then the distance to the
//     next lower set bit (bit 0 = the DC stands in for "no run pending"), so no lane waits for its predecessor -- ZRL codes (run > 15)
//     and the end-of-block code (coefficient 63 is zero) fall out of the same mask, exactly as jchuff.c encode_one_block emits them;
//   * a lane works out its codes once (registers), an exclusive scan of the lanes' bit totals places it, and it ORs every code into an
//     LDS image of the thread block's span of the stream (32 blocks, <= 32 * BLK_WORDS words), at most two words per code;
//   * the image goes out as whole words: plain stores, except the first and the last word, which the neighbouring thread blocks share
//     (atomic OR into the zeroed stream: two per thread block instead of two per 8x8 block).
// The same bits as jpeg_entropy_kernel<true> (D2S_JPEG_EMIT8=0 selects it; tests/test_gpu_jpeg.py compares the two and Pillow).
constexpr int EMIT8_BLOCKS = 32;                          // 8x8 blocks per thread block
constexpr int EMIT8_WORDS = EMIT8_BLOCKS * BLK_WORDS;     // the LDS image (worst case: every block at its 1 660-bit maximum)

// one code (len <= 27 bits, MSB first) at bit p of the LDS image: at most two words
__device__ __forceinline__ void put8(uint32_t* img, uint32_t& p, uint32_t val, int len) {
    const unsigned long long x = (unsigned long long)val << (64 - (int)(p & 31u) - len);
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    if (hi) atomicOr(&img[p >> 5], hi);
    if (lo) atomicOr(&img[(p >> 5) + 1], lo);
    p += (uint32_t)len;
}
// the codes of a lane's eight coefficients c[0..7] = zigzag 8j .. 8j+7 (lane 0: c[0] is the DC, coded from dcdiff), worked out ONCE:
// code / length per coefficient (length 0: a zero), the ZRL codes in front of it, the lane's total -- then emitted from these registers
struct Lane8 { uint32_t code[8]; uint8_t len[8], zrl[8]; uint32_t dccode, eob; int dclen, eoblen; uint32_t total; };
__device__ __forceinline__ void analyse_lane8(Lane8& L, int j, const int (&c)[8], unsigned long long mask, int dcdiff,
                                              const uint32_t* dc, const uint32_t* ac) {
    L.total = 0; L.dclen = 0; L.dccode = 0; L.eoblen = 0; L.eob = 0;
    if (j == 0) {
        const int a = dcdiff < 0 ? -dcdiff : dcdiff, t2 = dcdiff < 0 ? dcdiff - 1 : dcdiff;
        const int nb = a ? 32 - __builtin_clz(a) : 0;
        const uint32_t e = dc[nb];
        L.dccode = ((e & 0xffffu) << nb) | ((uint32_t)t2 & ((1u << nb) - 1u));
        L.dclen = (int)(e >> 16) + nb;
        L.total = (uint32_t)L.dclen;
    }
    const unsigned long long mx = mask | 1ull;                                       // bit 0: the DC ends every run
    const uint32_t zrl_len = ac[0xF0] >> 16;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int v = c[t], k = 8 * j + t;
        L.len[t] = 0; L.zrl[t] = 0; L.code[t] = 0;
        if (v == 0 || k == 0) continue;
        const unsigned long long below = mx & ((1ull << k) - 1ull);
        const int run = k - (63 - __builtin_clzll(below)) - 1;
        const int a = v < 0 ? -v : v, t2 = v < 0 ? v - 1 : v;
        const int nb = 32 - __builtin_clz(a);
        const uint32_t e = ac[((run & 15) << 4) | nb];
        L.code[t] = ((e & 0xffffu) << nb) | ((uint32_t)t2 & ((1u << nb) - 1u));
        L.len[t] = (uint8_t)((e >> 16) + (uint32_t)nb);
        L.zrl[t] = (uint8_t)(run >> 4);                                              // jchuff.c: while (r > 15) emit 0xF0
        L.total += (uint32_t)L.len[t] + (uint32_t)L.zrl[t] * zrl_len;
    }
    if (j == 7 && !(mask >> 63)) { L.eob = ac[0] & 0xffffu; L.eoblen = (int)(ac[0] >> 16); L.total += (uint32_t)L.eoblen; }   // end of block
}
__device__ __forceinline__ void emit_lane8(const Lane8& L, uint32_t* img, uint32_t& p, const uint32_t* ac) {
    if (L.dclen) put8(img, p, L.dccode, L.dclen);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (!L.len[t]) continue;
        for (int z = 0; z < (int)L.zrl[t]; ++z) put8(img, p, ac[0xF0] & 0xffffu, (int)(ac[0xF0] >> 16));
        put8(img, p, L.code[t], (int)L.len[t]);
    }
    if (L.eoblen) put8(img, p, L.eob, L.eoblen);
}

__global__ void __launch_bounds__(256)
jpeg_emit8_kernel(uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ uint32_t sAc[2][256];
    __shared__ uint32_t sDc[2][12];
    __shared__ uint32_t img[EMIT8_WORDS + 1];
    const int tid = threadIdx.x, grp = tid >> 3, j = tid & 7;
    for (int i = tid; i < 512; i += 256) sAc[i >> 8][i & 255] = tb.ac[i >> 8][i & 255];
    if (tid < 24) sDc[tid / 12][tid % 12] = tb.dc[tid / 12][tid % 12];
    for (int i = tid; i < EMIT8_WORDS + 1; i += 256) img[i] = 0;
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const long nblk = (long)g.nmcu * 6;
    const long gb0 = (long)blockIdx.x * EMIT8_BLOCKS, gb = gb0 + grp;
    const bool active = gb < nblk;
    const long gbc = active ? gb : nblk - 1;
    const uint16_t* blkbits = (const uint16_t*)(wsf + g.off_blkbits);
    // where the blocks start: the tile's first bit (jpeg_zero_kernel) + the block's first bit within its tile (jpeg_count_kernel)
    const long m = gbc / 6;
    const int b = (int)(gbc % 6), tbl = b >= 4;
    const uint4 q = ((const uint4*)wsf)[gbc * 8 + j];                  // (requested before the barrier: one round trip for everything)
    const uint32_t* blkoff = (const uint32_t*)(wsf + g.off_blkoff);
    const uint32_t* tpre = (const uint32_t*)(wsf + g.off_tpre);
    const long gbl = min(gb0 + EMIT8_BLOCKS, nblk) - 1;                // the span's last block (same tile: 256 % 32 == 0)
    const uint32_t t0 = tpre[gb0 >> 8];
    const uint32_t G0 = t0 + blkoff[gb0];                              // first bit of the span; the image's word 0 is stream word G0 >> 5
    const uint32_t Gb = t0 + blkoff[gbc];                              // first bit of this group's block
    uint32_t span = t0 + blkoff[gbl] + (uint32_t)blkbits[gbl] - G0;
    const int pred = j == 0 ? dc_pred((const short*)(wsf + g.off_dcs), m, b) : 0;
    __syncthreads();
    const bool last_tb = gb0 + EMIT8_BLOCKS >= nblk;
    const uint32_t pad = last_tb ? (8u - ((G0 + span) & 7u)) & 7u : 0u;                               // jchuff.c flush_bits
    // this lane's coefficients, the block's non-zero mask
    const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
    int c[8];
    uint32_t m8 = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        c[t] = (int)(short)((t & 1) ? (qw[t >> 1] >> 16) : (qw[t >> 1] & 0xffffu));
        m8 |= (uint32_t)(c[t] != 0) << t;
    }
    if (j == 0) m8 &= ~1u;
    uint32_t mlo = j < 4 ? m8 << (8 * j) : 0u, mhi = j >= 4 ? m8 << (8 * (j - 4)) : 0u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { mlo |= __shfl_xor(mlo, o); mhi |= __shfl_xor(mhi, o); }
    const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
    const int dcdiff = c[0] - pred;                                     // (lane 0 only)
    Lane8 L;
    analyse_lane8(L, j, c, mask, dcdiff, sDc[tbl], sAc[tbl]);
    uint32_t inc = L.total;                                            // exclusive scan over the group's eight lanes
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (j >= o) inc += u; }
    if (active) {
        uint32_t pos = Gb - (G0 & ~31u) + inc - L.total;               // bit position in the image
        emit_lane8(L, img, pos, sAc[tbl]);
        if (pad && gb == nblk - 1 && j == 7) put8(img, pos, (1u << pad) - 1u, (int)pad);
    }
    __syncthreads();
    span += pad;
    uint32_t* stream = (uint32_t*)(wsf + g.off_stream) + (G0 >> 5);
    const uint32_t endbit = (G0 & 31u) + span;
    const int nwords = (int)((endbit + 31u) >> 5);
    for (int i = tid; i < nwords; i += 256) {
        const uint32_t v = img[i];
        const bool shared = (i == 0 && (G0 & 31u)) || (i == nwords - 1 && (endbit & 31u));
        if (shared) { if (v) atomicOr(&stream[i], v); }
        else stream[i] = v;
    }
}

// stage 2 (round 5): bits per block = the AC bits jpeg_dct_kernel left in blkbits[] + the DC term (needs the predecessor's DC, which
// may belong to another thread block of stage 1); bits per 256-block tile.  Reads 2 + 2 bytes per block instead of 128.
__global__ void __launch_bounds__(256)
jpeg_count_kernel(uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ uint32_t s_w[8];
    const int tid = threadIdx.x;
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const long nblk = (long)g.nmcu * 6;
    const long gb = (long)blockIdx.x * 256 + tid;
    const bool active = gb < nblk;
    uint32_t bits = 0;
    uint16_t* blkbits = (uint16_t*)(wsf + g.off_blkbits);
    if (active) {
        const long m = gb / 6;
        const int b = (int)(gb % 6), tbl = b >= 4;
        const short* dcs = (const short*)(wsf + g.off_dcs);
        const int dcdiff = (int)dcs[m * 8 + b] - dc_pred(dcs, m, b);
        const int a = dcdiff < 0 ? -dcdiff : dcdiff;
        const int nb = a ? 32 - __builtin_clz(a) : 0;
        bits = (uint32_t)blkbits[gb] + (tb.dc[tbl][nb] >> 16) + (uint32_t)nb;
        blkbits[gb] = (uint16_t)bits;
    }
    const uint32_t off = block256_exscan(bits, s_w);                   // (round 6) where the block starts within its tile: the emit reads it
    if (active) ((uint32_t*)(wsf + g.off_blkoff))[gb] = off;
    const uint32_t sum = block256_sum(bits, s_w);
    if (tid == 0) ((uint32_t*)(wsf + g.off_tiles))[blockIdx.x] = sum;
}

// stage 3: zero the words of the stream that will be used; block 0 publishes the (byte-padded) total bit count
__global__ void __launch_bounds__(256)
jpeg_zero_kernel(uint8_t* __restrict__ ws, JpegGeom g) {
    __shared__ uint32_t s_w[8];
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    uint32_t prefix, total;
    tile_prefix((const uint32_t*)(wsf + g.off_tiles), cdiv_dev((long)g.nmcu * 6, 256), 0, s_w, prefix, total);
    total = (total + 7u) & ~7u;                                       // flush_bits pads the last byte with 1-bits
    if (blockIdx.x == 0 && threadIdx.x == 0) *(uint32_t*)(wsf + g.off_total) = total;
    if (blockIdx.x == 0) {                                            // (round 6) first bit of every tile, for the emit
        const uint32_t* tiles = (const uint32_t*)(wsf + g.off_tiles);
        uint32_t* tpre = (uint32_t*)(wsf + g.off_tpre);
        const int n = cdiv_dev((long)g.nmcu * 6, 256);
        uint32_t carry = 0;
        for (int base = 0; base < n; base += 256) {                   // (block-uniform trip count)
            const int i = base + (int)threadIdx.x;
            const uint32_t v = i < n ? tiles[i] : 0u;
            const uint32_t ex = block256_exscan(v, s_w);
            if (i < n) tpre[i] = carry + ex;
            carry += block256_sum(v, s_w);
        }
    }
    long used = ((long)total + 31) / 32 + 4;                          // + slack: the last chunk is read whole
    used = min((used + 15) / 16 * 16, g.cap_words);
    const long base = (long)blockIdx.x * 4096;                        // 4096 words per block, 4 x 16 bytes per thread
    for (int r = 0; r < 4; ++r) {
        long i = base + (long)(r * 256 + threadIdx.x) * 4;
        if (i < used) *(uint4*)(wsf + g.off_stream + i * 4) = make_uint4(0, 0, 0, 0);
    }
}

// ---- stages 5-7: byte stuffing -------------------------------------------------------------------
__device__ __forceinline__ uint32_t count_ff(uint32_t w) {
    uint32_t n = 0;
    n += (w >> 24) == 0xffu; n += ((w >> 16) & 0xffu) == 0xffu; n += ((w >> 8) & 0xffu) == 0xffu; n += (w & 0xffu) == 0xffu;
    return n;
}

__global__ void __launch_bounds__(256)
jpeg_ffcount_kernel(uint8_t* __restrict__ ws, JpegGeom g) {
    __shared__ uint32_t s_w[8];
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const uint32_t nbytes = *(const uint32_t*)(wsf + g.off_total) >> 3;
    if ((long)blockIdx.x * 256 * 64 >= nbytes) return;                // whole thread block beyond the stream
    const long chunk = (long)blockIdx.x * 256 + threadIdx.x;
    uint32_t n = 0;
    if (chunk * 64 < nbytes) {
        const uint4* p = (const uint4*)(wsf + g.off_stream + chunk * 64);
        for (int i = 0; i < 4; ++i) {
            uint4 v = p[i];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 4; ++k) {
                long byte0 = chunk * 64 + i * 16 + k * 4;
                if (byte0 + 4 <= nbytes) n += count_ff(w[k]);
                else for (int q = 0; q < 4; ++q) if (byte0 + q < nbytes) n += ((w[k] >> (24 - 8 * q)) & 0xffu) == 0xffu;
            }
        }
        ((uint32_t*)(wsf + g.off_ffcnt))[chunk] = n;
    }
    const uint32_t sum = block256_sum(n, s_w);
    if (threadIdx.x == 0) ((uint32_t*)(wsf + g.off_fftiles))[blockIdx.x] = sum;
}

// byte-stuffed copy behind the header; block 0 also writes the header, the EOI marker and the size.
// Round 5: through LDS.  A thread's 64 stream bytes land at an arbitrary byte offset of the output (header + chunk + the 0xFF count
// before it), so the per-byte global stores of rounds 1-4 were the whole cost of this pass (21 us of an 88 us encode).  A block's 256
// chunks form ONE contiguous output span (<= 32 KiB: every byte a 0xFF); the threads stuff into an LDS image of that span laid out with
// the span's own 16-byte phase, and the block then copies the image out as aligned 16-byte stores (the ragged first / last 16-byte
// group, which the neighbouring blocks share, byte by byte).
constexpr int STUFF_LDS = 256 * 128 + 32;
__global__ void __launch_bounds__(256)
jpeg_stuff_kernel(const uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb, uint8_t* __restrict__ out, long out_stride,
                  int* __restrict__ sizes) {
    __shared__ uint32_t s_w[8];
    __shared__ __attribute__((aligned(16))) uint8_t img[STUFF_LDS];
    const uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const uint32_t nbytes = *(const uint32_t*)(wsf + g.off_total) >> 3;
    if ((long)blockIdx.x * 256 * 64 >= nbytes) return;
    const int ntiles = (int)((((long)nbytes + 63) / 64 + 255) / 256);
    uint32_t prefix, total_ff;
    tile_prefix((const uint32_t*)(wsf + g.off_fftiles), ntiles, (int)blockIdx.x, s_w, prefix, total_ff);
    const long size = (long)HDR_LEN + nbytes + total_ff + 2;
    const bool fits = size <= out_stride;
    uint8_t* o = out + (long)blockIdx.y * out_stride;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) sizes[blockIdx.y] = fits ? (int)size : -1;
        if (fits) {
            for (int i = threadIdx.x; i < HDR_LEN; i += 256) o[i] = tb.header[i];
            if (threadIdx.x == 0) { o[size - 2] = 0xff; o[size - 1] = 0xd9; }
        }
    }
    const long chunk = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = chunk * 64 < nbytes;
    const uint32_t mine = live ? ((const uint32_t*)(wsf + g.off_ffcnt))[chunk] : 0u;
    const uint32_t before = prefix + block256_exscan(mine, s_w);
    const uint32_t block_ff = block256_sum(mine, s_w);                // (every thread takes part in the block sums)
    if (!fits) return;                                                // output too small: nothing but the size (-1) is written
    // this block's output span [lo, hi) in bytes of the frame's output buffer
    const long c0 = (long)blockIdx.x * 256;
    const long in_bytes = min((long)256 * 64, (long)nbytes - c0 * 64);
    const long lo = (long)HDR_LEN + c0 * 64 + prefix, hi = lo + in_bytes + block_ff;
    const uint8_t* dst0 = o + lo;
    const int phase = (int)((uintptr_t)dst0 & 15);                    // img[phase + i] <-> dst0[i]
    if (live) {
        int p = phase + (int)((chunk - c0) * 64 + (before - prefix));
        const uint4* src = (const uint4*)(wsf + g.off_stream + chunk * 64);
        const int n = (int)min((long)64, (long)nbytes - chunk * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = src[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i * 16 + k * 4 + q < n) {
                        const uint8_t byte = (uint8_t)(w[k] >> (24 - 8 * q));
                        img[p++] = byte;
                        if (byte == 0xff) img[p++] = 0;               // jchuff.c emit_byte: stuff a zero after 0xFF
                    }
                }
        }
    }
    __syncthreads();
    const int span = (int)(hi - lo);
    uint8_t* base = o + lo - phase;                                   // 16-byte aligned; img[i] <-> base[i] for i in [phase, phase + span)
    const int groups = (phase + span + 15) >> 4;
    for (int gi = threadIdx.x; gi < groups; gi += 256) {
        const int b0 = gi * 16;
        if (b0 >= phase && b0 + 16 <= phase + span) *(uint4*)(base + b0) = *(const uint4*)(img + b0);
        else for (int q = max(b0, phase); q < min(b0 + 16, phase + span); ++q) base[q] = img[q];
    }
}

}  // namespace d2s

using namespace d2s;

extern "C" int d2s_jpeg_bound(int H, int W, int64_t* out_bytes, int64_t* workspace_bytes) {
    D2S_REQUIRE(H > 0 && W > 0 && H < 65536 && W < 65536, "d2s_jpeg_bound: bad shape");
    JpegGeom g = make_geom(H, W);
    if (out_bytes) *out_bytes = HDR_LEN + 2 + 2 * g.cap_words * 4;      // every stream byte 0xFF: cannot be exceeded
    if (workspace_bytes) *workspace_bytes = g.ws_frame;
    return D2S_OK;
}

extern "C" int d2s_jpeg_encode(const void* frames, int fmt, int batch, int H, int W, int quality, uint8_t* out,
                               int64_t out_stride, int32_t* sizes, void* workspace, int64_t workspace_bytes, void* stream) {
    D2S_REQUIRE(frames && out && sizes && workspace, "d2s_jpeg_encode: null pointer");
    D2S_REQUIRE(batch > 0 && batch <= 65535 && H > 0 && W > 0 && H < 65536 && W < 65536, "d2s_jpeg_encode: bad shape");
    D2S_REQUIRE(fmt == D2S_FMT_U8_HWC || fmt == D2S_FMT_F32_HWC, "d2s_jpeg_encode: frames must be U8_HWC or F32_HWC");
    D2S_REQUIRE(quality >= 1 && quality <= 100, "d2s_jpeg_encode: quality must be 1..100");
    D2S_REQUIRE(out_stride >= HDR_LEN + 2, "d2s_jpeg_encode: out_stride too small");
    D2S_REQUIRE(((uintptr_t)workspace & 255) == 0, "d2s_jpeg_encode: workspace must be 256-byte aligned");
    JpegGeom g = make_geom(H, W);
    D2S_REQUIRE(workspace_bytes >= g.ws_frame * batch, "d2s_jpeg_encode: workspace too small (see d2s_jpeg_bound)");
    D2S_REQUIRE((long)g.nmcu * 6 * BLK_WORDS * 32 < (1L << 32), "d2s_jpeg_encode: frame too large");
    JpegTables tb;
    make_tables(H, W, quality, tb);
    hipStream_t st = (hipStream_t)stream;
    uint8_t* ws = (uint8_t*)workspace;
    dim3 g1(cdiv(g.mc, DCT_MCUS), g.mr, batch);
    if (fmt == D2S_FMT_U8_HWC) hipLaunchKernelGGL(jpeg_dct_kernel<D2S_FMT_U8_HWC>, g1, dim3(256), 0, st, frames, ws, g, tb);
    else hipLaunchKernelGGL(jpeg_dct_kernel<D2S_FMT_F32_HWC>, g1, dim3(256), 0, st, frames, ws, g, tb);
    const dim3 ge(cdiv((long)g.nmcu * 6, 256), batch), gc(cdiv(g.n_chunks, 256), batch);
    hipLaunchKernelGGL(jpeg_count_kernel, ge, dim3(256), 0, st, ws, g, tb);
    hipLaunchKernelGGL(jpeg_zero_kernel, dim3(cdiv(g.cap_words, 4096), batch), dim3(256), 0, st, ws, g);
    // Eight lanes per block when the launch would otherwise leave the chip short of threads (one lane per block: 97 200 threads for a
    // 3840x1080 frame; 23.2 -> 15.8 us); from a few frames per launch on, the one-lane walk fills the chip by itself and issues fewer
    // instructions on sparse blocks (16 frames of a smooth scene: 28.5 against 35.3 us per frame).  D2S_JPEG_EMIT8 = 0 / 1 forces one.
    static EnvInt emit8_env{"D2S_JPEG_EMIT8", -1};
    const bool emit8 = emit8_env.get() < 0 ? (long)batch * g.nmcu * 6 < 300000 : emit8_env.get() != 0;
    if (emit8) hipLaunchKernelGGL(jpeg_emit8_kernel, dim3(cdiv((long)g.nmcu * 6, EMIT8_BLOCKS), batch), dim3(256), 0, st, ws, g, tb);
    else hipLaunchKernelGGL(jpeg_entropy_kernel<true>, ge, dim3(256), 0, st, ws, g, tb);
    hipLaunchKernelGGL(jpeg_ffcount_kernel, gc, dim3(256), 0, st, ws, g);
    hipLaunchKernelGGL(jpeg_stuff_kernel, gc, dim3(256), 0, st, (const uint8_t*)ws, g, tb, out, (long)out_stride, sizes);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
