// MJPEG sink: baseline JPEG encode of the packed stereo frame, on the device.
//
// Replaces `cv2.imencode('.jpg', bgr, [IMWRITE_JPEG_QUALITY, q])` on the float32 frame make_sbs returns
// (reference streamer.py:249-256, 285-291): convertTo(CV_8U) (round-half-even, saturate) followed by
// libjpeg(-turbo) with jpeg_set_defaults + jpeg_set_quality(q, TRUE) — YCbCr 4:2:0, slow-integer FDCT,
// Annex-K Huffman tables, no restart markers, JFIF 1.01 header.  Every stage below is the integer
// arithmetic libjpeg publishes (jccolor.c, jcsample.c, jfdctint.c, jcdctmgr.c, jccoefct.c, jchuff.c,
// jcmarker.c), so the stream is BYTE-IDENTICAL to libjpeg-turbo's (oracle/jpeg_oracle.py pins that).
//
// Stages (all HBM-/latency-bound byte work; grid.z = frame):
//   1 jpeg_dct_kernel     16 MCUs (256x16 px) per block: RGB -> YCbCr (+h2v2), FDCT, quantise, zigzag;
//                         writes int16 coefficients [mcu][6][64] and the AC bit count of every MCU
//   2 jpeg_scan_kernel    one block per frame: adds the DC code lengths (needs the neighbour MCU's DC),
//                         exclusive scan -> bit offset of every MCU, total bits
//   3 jpeg_zero_kernel    zero the words of the (unstuffed) bit stream that will be used
//   4 jpeg_huff_kernel    one wave per MCU, one lane per coefficient: codes assembled in LDS, shifted to the
//                         MCU's bit offset and merged into the stream (atomic OR only on the two edge words)
//   5 jpeg_ffcount_kernel 0xFF bytes per 64-byte chunk;  6 jpeg_ffscan_kernel: scan + header + EOI + size
//   7 jpeg_stuff_kernel   byte-stuffed copy behind the header
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
constexpr int MCU_LDS_WORDS = 320;      // 6 * 1660 bits = 312 words, + slack read by the shifted flush

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

struct JpegGeom {
    int H, W, mr, mc, nmcu;          // MCU rows / cols (16x16 px)
    int ybw, ybh;                    // real luma blocks across / down
    int He;                          // H rounded up to even (the rows the chroma planes are derived from)
    long ws_frame;                   // workspace bytes per frame
    long off_acbits, off_dcs, off_mcuoff, off_stream, off_ffcnt;
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
    g.off_acbits = o; o = al(o + (long)g.nmcu * 4);
    g.off_dcs = o; o = al(o + (long)g.nmcu * 8 * 2);                 // quantised DC of the 6 blocks of every MCU ([mcu][8] int16)
    g.off_mcuoff = o; o = al(o + (long)(g.nmcu + 1) * 4);
    g.cap_words = ((long)g.nmcu * 6 * BLK_WORDS + 16 + 15) / 16 * 16;
    g.n_chunks = (int)(g.cap_words / 16);
    g.off_stream = o; o = al(o + g.cap_words * 4);
    g.off_ffcnt = o; o = al(o + (long)(g.n_chunks + 1) * 4);
    g.ws_frame = o;
    return g;
}

// ---- stage 1 -----------------------------------------------------------------------------------
constexpr int DCT_MCUS = 16;             // MCUs per thread block (256 px x 16 rows)
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

// Bits the AC coefficients of one block take (jchuff.c encode_one_block), one lane per zigzag position.
// v: this lane's coefficient (lane 0 = DC, ignored).  Returns the wave-uniform total.
__device__ __forceinline__ int ac_symbol(int v, int lane, unsigned long long nzmask, int& nb, int& run) {
    unsigned long long below = nzmask & ((1ull << lane) - 1ull);
    int prev = below ? 63 - __builtin_clzll(below) : 0;              // previous non-zero AC position (0 = the DC slot)
    run = lane - prev - 1;
    int a = v < 0 ? -v : v;
    nb = a ? 32 - __builtin_clz(a) : 0;
    return ((run & 15) << 4) | nb;
}

template <int FMT>
__global__ void __launch_bounds__(256)
jpeg_dct_kernel(const void* __restrict__ frames, uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ __attribute__((aligned(16))) uint8_t sY[16][DCT_MCUS * 16];
    __shared__ __attribute__((aligned(16))) uint8_t sC[2][8][DCT_MCUS * 8];
    __shared__ short sW[DCT_BLOCKS][64];       // row-pass output: |x| <= 1024 << PASS1_BITS, fits int16
    __shared__ short sZ[DCT_BLOCKS][64];
    __shared__ uint16_t sQ8[2][64];
    __shared__ uint32_t sMagic[2][64];
    __shared__ uint8_t sN2Z[64];
    __shared__ uint32_t sAcLen[2][256];

    const int tid = threadIdx.x;
    const int mrow = blockIdx.y, mcol0 = blockIdx.x * DCT_MCUS, f = blockIdx.z;
    const int nm = min(DCT_MCUS, g.mc - mcol0);
    const char* frame = (const char*)frames + (long)f * g.H * g.W * 3 * (FMT == D2S_FMT_U8_HWC ? 1 : 4);
    uint8_t* wsf = ws + (long)f * g.ws_frame;

    if (tid < 128) { sQ8[tid >> 6][tid & 63] = tb.q8[tid >> 6][tid & 63]; sMagic[tid >> 6][tid & 63] = tb.magic[tid >> 6][tid & 63]; }
    if (tid < 64) sN2Z[tid] = tb.nat2zig[tid];
    for (int i = tid; i < 512; i += 256) sAcLen[i >> 8][i & 255] = tb.ac[i >> 8][i & 255] >> 16;

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
        for (int i = 0; i < 8; ++i) sW[blk][r * 8 + i] = (short)d[i];
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

    // dummy blocks (jccoefct.c compress_data), coefficient store, AC bit count: one wave per MCU, lane = zigzag k
    const int lane = tid & 63, wv = tid >> 6;
    short* coefs = (short*)wsf;
    const bool row1 = (2 * mrow + 1) < g.ybh;
    for (int m = wv; m < nm; m += 4) {
        const bool col1 = (2 * (mcol0 + m) + 1) < g.ybw;
        const int s1 = col1 ? 1 : 0;
        const int srcs[6] = {0, s1, row1 ? 2 : s1, row1 ? (col1 ? 3 : 2) : s1, 4, 5};
        const long mcu = (long)mrow * g.mc + mcol0 + m;
        int bits = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int src = srcs[b];
            int v = (src == b) ? sZ[m * 6 + b][lane] : (lane == 0 ? sZ[m * 6 + src][0] : 0);
            coefs[(mcu * 6 + b) * 64 + lane] = (short)v;
            if (lane == 0) ((short*)(wsf + g.off_dcs))[mcu * 8 + b] = (short)v;
            unsigned long long nz = __ballot(v != 0 && lane > 0);
            if (v != 0 && lane > 0) {
                int nb, run;
                int sym = ac_symbol(v, lane, nz, nb, run);
                bits += (run >> 4) * (int)sAcLen[b >= 4][0xF0] + (int)sAcLen[b >= 4][sym] + nb;
            }
            if (lane == 0 && (nz >> 63) == 0) bits += (int)sAcLen[b >= 4][0];          // EOB unless position 63 is non-zero
        }
        for (int o = 32; o > 0; o >>= 1) bits += __shfl_xor(bits, o);
        if (lane == 0) ((uint32_t*)(wsf + g.off_acbits))[mcu] = (uint32_t)bits;
    }
}

// ---- stage 2: bit offsets ----------------------------------------------------------------------
__device__ __forceinline__ int dc_bits(int diff, const uint32_t* dctab) {
    int a = diff < 0 ? -diff : diff;
    int nb = a ? 32 - __builtin_clz(a) : 0;
    return (int)(dctab[nb] >> 16) + nb;
}

// Block-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, *total = sum.
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        uint32_t w = lane < 16 ? s_wave[lane] : 0, wi = w;
        for (int o = 1; o < 16; o <<= 1) { uint32_t u = __shfl_up(wi, o); if (lane >= o) wi += u; }
        if (lane < 16) s_wave[lane] = wi - w;
        if (lane == 15) s_wave[16] = wi;
    }
    __syncthreads();
    uint32_t r = s_wave[wv] + inc - v;
    *total = s_wave[16];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(1024)
jpeg_scan_kernel(uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ uint32_t s_wave[17];
    __shared__ uint32_t s_dc[2][12];
    uint8_t* wsf = ws + (long)blockIdx.x * g.ws_frame;
    const short* dcs = (const short*)(wsf + g.off_dcs);
    uint32_t* acbits = (uint32_t*)(wsf + g.off_acbits);
    uint32_t* off = (uint32_t*)(wsf + g.off_mcuoff);
    if (threadIdx.x < 24) s_dc[threadIdx.x / 12][threadIdx.x % 12] = tb.dc[threadIdx.x / 12][threadIdx.x % 12];
    __syncthreads();
    // walk 1 (coalesced): total bits of every MCU = AC bits + the six DC codes (DC prediction runs across MCUs)
    for (int m = threadIdx.x; m < g.nmcu; m += 1024) {
        const short* c = dcs + (long)m * 8;
        int py = 0, pb = 0, pr = 0;
        if (m > 0) { py = c[-8 + 3]; pb = c[-8 + 4]; pr = c[-8 + 5]; }
        uint32_t bits = acbits[m];
        for (int b = 0; b < 4; ++b) { int d = c[b]; bits += dc_bits(d - py, s_dc[0]); py = d; }
        bits += dc_bits(c[4] - pb, s_dc[1]) + dc_bits(c[5] - pr, s_dc[1]);
        acbits[m] = bits;
    }
    __threadfence_block();
    __syncthreads();
    // walk 2: each thread owns a contiguous run of MCUs
    const int per = (g.nmcu + 1023) / 1024;
    const int m0 = threadIdx.x * per, m1 = min(m0 + per, g.nmcu);
    uint32_t sum = 0;
    for (int m = m0; m < m1; ++m) sum += acbits[m];
    uint32_t total;
    uint32_t base = block_exscan(sum, s_wave, &total);
    for (int m = m0; m < m1; ++m) { off[m] = base; base += acbits[m]; }
    if (threadIdx.x == 0) off[g.nmcu] = (total + 7u) & ~7u;           // flush_bits pads the last byte with 1-bits
}

// ---- stage 3 -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
jpeg_zero_kernel(uint8_t* __restrict__ ws, JpegGeom g) {
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const uint32_t total = ((const uint32_t*)(wsf + g.off_mcuoff))[g.nmcu];
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;             // word index, 16 bytes per thread
    long used = ((long)total + 31) / 32 + 4;                          // + slack: the last chunk is read whole
    used = (used + 15) / 16 * 16;
    if (i < used && i < g.cap_words) *(uint4*)(wsf + g.off_stream + i * 4) = make_uint4(0, 0, 0, 0);
}

// ---- stage 4: Huffman coding ----------------------------------------------------------------------
// Bit order everywhere: stream bit p lives in word p >> 5 at bit 31 - (p & 31) (MSB first).
__device__ __forceinline__ void lds_put(uint32_t* buf, int pos, unsigned long long bits, int len) {
    if (len == 0) return;
    unsigned long long F = bits << (64 - len);                        // left-aligned field
    int w = pos >> 5, o = pos & 31;
    unsigned long long hi = F >> o;
    uint32_t w0 = (uint32_t)(hi >> 32), w1 = (uint32_t)hi, w2 = o ? (uint32_t)((F << (64 - o)) >> 32) : 0u;
    if (w0) atomicOr(&buf[w], w0);
    if (w1) atomicOr(&buf[w + 1], w1);
    if (w2) atomicOr(&buf[w + 2], w2);
}

__global__ void __launch_bounds__(256)
jpeg_huff_kernel(uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb) {
    __shared__ uint32_t sAc[2][256];
    __shared__ uint32_t sDc[2][12];
    __shared__ uint32_t sBuf[4][MCU_LDS_WORDS + 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 512; i += 256) sAc[i >> 8][i & 255] = tb.ac[i >> 8][i & 255];
    if (tid < 24) sDc[tid / 12][tid % 12] = tb.dc[tid / 12][tid % 12];
    uint32_t* buf = sBuf[wv];
    for (int i = lane; i < MCU_LDS_WORDS + 4; i += 64) buf[i] = 0;
    __syncthreads();
    const int m = blockIdx.x * 4 + wv;
    if (m >= g.nmcu) return;
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const short* c = (const short*)wsf + (long)m * 384;
    const uint32_t* off = (const uint32_t*)(wsf + g.off_mcuoff);
    uint32_t* stream = (uint32_t*)(wsf + g.off_stream);

    int prev[3] = {0, 0, 0};
    if (m > 0) { const short* d = (const short*)(wsf + g.off_dcs) + (long)(m - 1) * 8; prev[0] = d[3]; prev[1] = d[4]; prev[2] = d[5]; }
    unsigned long long bitsv[6];
    int lens[6], inc[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const int tbl = b >= 4, comp = b < 4 ? 0 : b - 3;
        int v = c[b * 64 + lane];
        int dcv = __builtin_amdgcn_readfirstlane(v);
        unsigned long long nz = __ballot(v != 0 && lane > 0);
        unsigned long long bits = 0;
        int len = 0;
        if (lane == 0) {                                              // DC: category code + the low bits of (diff, or diff-1 if negative)
            int diff = v - prev[comp];
            int a = diff < 0 ? -diff : diff, t2 = diff < 0 ? diff - 1 : diff;
            int nb = a ? 32 - __builtin_clz(a) : 0;
            uint32_t e = sDc[tbl][nb];
            bits = ((unsigned long long)(e & 0xffff) << nb) | (unsigned)(t2 & ((1 << nb) - 1));
            len = (int)(e >> 16) + nb;
        } else if (v != 0) {
            int nb, run;
            int sym = ac_symbol(v, lane, nz, nb, run);
            uint32_t zrl = sAc[tbl][0xF0], e = sAc[tbl][sym];
            for (int z = run >> 4; z > 0; --z) { bits = (bits << (zrl >> 16)) | (zrl & 0xffff); len += (int)(zrl >> 16); }
            int t2 = v < 0 ? v - 1 : v;
            bits = (((bits << (e >> 16)) | (e & 0xffff)) << nb) | (unsigned)(t2 & ((1 << nb) - 1));
            len += (int)(e >> 16) + nb;
        }
        const int last = nz ? 63 - __builtin_clzll(nz) : 0;          // the lane that owns the end-of-block code
        if (lane == last && last != 63) { uint32_t e = sAc[tbl][0]; bits = (bits << (e >> 16)) | (e & 0xffff); len += (int)(e >> 16); }
        prev[comp] = dcv;
        bitsv[b] = bits; lens[b] = len; inc[b] = len;
    }
    for (int o = 1; o < 64; o <<= 1) {                                // six inclusive scans in lockstep (independent shuffles in flight)
        int u[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) u[b] = __shfl_up(inc[b], o);
#pragma unroll
        for (int b = 0; b < 6; ++b) if (lane >= o) inc[b] += u[b];
    }
    int pos = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        lds_put(buf, pos + inc[b] - lens[b], bitsv[b], lens[b]);
        pos += __builtin_amdgcn_readlane(inc[b], 63);
    }
    const uint32_t G = off[m];
    if (m == g.nmcu - 1) {                                            // jchuff.c flush_bits: fill the last byte with ones
        int pad = (int)(off[g.nmcu] - (G + (uint32_t)pos));
        if (lane == 0 && pad) lds_put(buf, pos, (1ull << pad) - 1, pad);
        pos += pad;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int s = (int)(G & 31u);
    const long gw = (long)(G >> 5);
    const int nw = (s + pos + 31) >> 5;
    for (int j = lane; j < nw; j += 64) {
        uint32_t cur = buf[j];
        uint32_t val = s ? (((j > 0 ? buf[j - 1] : 0u) << (32 - s)) | (cur >> s)) : cur;
        if (j == 0 || j == nw - 1) { if (val) atomicOr(&stream[gw + j], val); }
        else stream[gw + j] = val;
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
    uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const uint32_t nbytes = ((const uint32_t*)(wsf + g.off_mcuoff))[g.nmcu] >> 3;
    const long chunk = (long)blockIdx.x * 256 + threadIdx.x;
    if (chunk * 64 >= nbytes) return;
    const uint4* p = (const uint4*)(wsf + g.off_stream + chunk * 64);
    uint32_t n = 0;
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

__global__ void __launch_bounds__(1024)
jpeg_ffscan_kernel(uint8_t* __restrict__ ws, JpegGeom g, JpegTables tb, uint8_t* __restrict__ out, long out_stride,
                   int* __restrict__ sizes) {
    __shared__ uint32_t s_wave[17];
    uint8_t* wsf = ws + (long)blockIdx.x * g.ws_frame;
    const uint32_t nbytes = ((const uint32_t*)(wsf + g.off_mcuoff))[g.nmcu] >> 3;
    uint32_t* cnt = (uint32_t*)(wsf + g.off_ffcnt);
    const int nch = (int)((nbytes + 63) / 64);
    const int per = (nch + 1023) / 1024;
    const int c0 = threadIdx.x * per, c1 = min(c0 + per, nch);
    uint32_t sum = 0;
    for (int i = c0; i < c1; ++i) sum += cnt[i];
    uint32_t total;
    uint32_t base = block_exscan(sum, s_wave, &total);
    for (int i = c0; i < c1; ++i) { uint32_t n = cnt[i]; cnt[i] = base; base += n; }
    const long size = (long)HDR_LEN + nbytes + total + 2;
    uint8_t* o = out + (long)blockIdx.x * out_stride;
    const bool fits = size <= out_stride;
    if (threadIdx.x == 0) { sizes[blockIdx.x] = fits ? (int)size : -1; cnt[nch] = fits ? 1u : 0u; }
    if (!fits) return;
    for (int i = threadIdx.x; i < HDR_LEN; i += 1024) o[i] = tb.header[i];
    if (threadIdx.x == 0) { o[size - 2] = 0xff; o[size - 1] = 0xd9; }
}

__global__ void __launch_bounds__(256)
jpeg_stuff_kernel(const uint8_t* __restrict__ ws, JpegGeom g, uint8_t* __restrict__ out, long out_stride) {
    const uint8_t* wsf = ws + (long)blockIdx.y * g.ws_frame;
    const uint32_t nbytes = ((const uint32_t*)(wsf + g.off_mcuoff))[g.nmcu] >> 3;
    const long chunk = (long)blockIdx.x * 256 + threadIdx.x;
    if (chunk * 64 >= nbytes) return;
    const uint32_t* cnt = (const uint32_t*)(wsf + g.off_ffcnt);
    const int nch = (int)((nbytes + 63) / 64);
    if (cnt[nch] == 0u) return;                                       // output too small: nothing is written
    uint8_t* o = out + (long)blockIdx.y * out_stride + HDR_LEN + chunk * 64 + cnt[chunk];
    const uint4* p = (const uint4*)(wsf + g.off_stream + chunk * 64);
    const int n = (int)min((long)64, (long)nbytes - chunk * 64);
    for (int i = 0; i < 4; ++i) {
        uint4 v = p[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; ++k)
            for (int q = 0; q < 4; ++q) {
                if (i * 16 + k * 4 + q >= n) return;
                uint8_t byte = (uint8_t)(w[k] >> (24 - 8 * q));
                *o++ = byte;
                if (byte == 0xff) *o++ = 0;                           // jchuff.c emit_byte: stuff a zero after 0xFF
            }
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
    D2S_REQUIRE(batch > 0 && H > 0 && W > 0 && H < 65536 && W < 65536, "d2s_jpeg_encode: bad shape");
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
    hipLaunchKernelGGL(jpeg_scan_kernel, dim3(batch), dim3(1024), 0, st, ws, g, tb);
    hipLaunchKernelGGL(jpeg_zero_kernel, dim3(cdiv(g.cap_words, 1024), batch), dim3(256), 0, st, ws, g);
    hipLaunchKernelGGL(jpeg_huff_kernel, dim3(cdiv(g.nmcu, 4), batch), dim3(256), 0, st, ws, g, tb);
    hipLaunchKernelGGL(jpeg_ffcount_kernel, dim3(cdiv(g.n_chunks, 256), batch), dim3(256), 0, st, ws, g);
    hipLaunchKernelGGL(jpeg_ffscan_kernel, dim3(batch), dim3(1024), 0, st, ws, g, tb, out, (long)out_stride, sizes);
    hipLaunchKernelGGL(jpeg_stuff_kernel, dim3(cdiv(g.n_chunks, 256), batch), dim3(256), 0, st, (const uint8_t*)ws, g, out, (long)out_stride);
    D2S_CHECK_LAUNCH();
    return D2S_OK;
}
