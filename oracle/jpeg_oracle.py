"""CPU oracle for the MJPEG sink — TEST INFRASTRUCTURE, never shipped.

The reference's sink is ``cv2.imencode('.jpg', bgr, [IMWRITE_JPEG_QUALITY, q])`` on the float32
HWC frame ``make_sbs`` returns (reference streamer.py:249-256, 285-291; quality from settings.yaml
``Stream Quality``, utils.py:821).  The arithmetic lives in a third-party dependency that is NOT under
/root/reference: opencv-python==4.12.0.88 (requirements.txt:4), whose JPEG writer drives the bundled
libjpeg-turbo with ``jpeg_set_defaults`` + ``jpeg_set_quality(q, force_baseline=TRUE)``: baseline
sequential DCT, YCbCr 4:2:0 (h2v2 chroma), the slow-integer forward DCT, Annex-K Huffman tables, no
restart markers, JFIF 1.01 header with 1:1 density.  cv2 is not installed in this image, so the pin is
the same library reached another way: Pillow 12.2 links libjpeg-turbo too and with
``quality=q, subsampling='4:2:0', optimize=False`` makes the same libjpeg calls.  This file restates
the published libjpeg algorithm (jccolor.c rgb_ycc, jcsample.c h2v2_downsample, jfdctint.c,
jcdctmgr.c quantisation, jccoefct.c dummy blocks, jchuff.c, jcmarker.c) and is pinned BYTE-FOR-BYTE
against Pillow's output (tests/golden/jpeg_*.npz, made by tests/golden/make_jpeg_golden.py).
The float32 -> uint8 step in front of it is cv2's ``convertTo(CV_8U)`` = round-half-even + saturate.
"""
import numpy as np

# ---- Annex K tables (ITU-T T.81), natural order -------------------------------------------------
STD_LUMA_Q = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64)
STD_CHROMA_Q = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.int64)

ZIGZAG = np.array([                     # zigzag index k -> natural index
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])

DC_LUMA_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_CHROMA_BITS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0]
DC_VALS = list(range(12))
AC_LUMA_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d]
AC_LUMA_VALS = [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa]
AC_CHROMA_BITS = [0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77]
AC_CHROMA_VALS = [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa]


def quant_tables(quality):
    """jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline=TRUE) -> (luma, chroma), natural order."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for basic in (STD_LUMA_Q, STD_CHROMA_Q):
        t = (basic * scale + 50) // 100
        out.append(np.clip(t, 1, 255))
    return out


def huff_codes(bits, vals):
    """jchuff.c jpeg_make_c_derived_tbl: symbol -> (code, length)."""
    code, k, table = 0, 0, {}
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


def header(H, W, quality):
    """jcmarker.c: SOI, JFIF APP0, two DQT, SOF0, four DHT, SOS (the bytes before the entropy-coded segment)."""
    ql, qc = quant_tables(quality)
    b = bytearray(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for idx, t in enumerate((ql, qc)):
        b += b"\xff\xdb\x00\x43" + bytes([idx]) + bytes(int(t[ZIGZAG[k]]) for k in range(64))
    b += b"\xff\xc0\x00\x11\x08" + bytes([H >> 8, H & 255, W >> 8, W & 255]) + b"\x03\x01\x22\x00\x02\x11\x01\x03\x11\x01"
    for tc_th, bits, vals in ((0x00, DC_LUMA_BITS, DC_VALS), (0x10, AC_LUMA_BITS, AC_LUMA_VALS),
                              (0x01, DC_CHROMA_BITS, DC_VALS), (0x11, AC_CHROMA_BITS, AC_CHROMA_VALS)):
        n = 2 + 1 + 16 + len(vals)
        b += b"\xff\xc4" + bytes([n >> 8, n & 255, tc_th]) + bytes(bits) + bytes(vals)
    b += b"\xff\xda\x00\x0c\x03\x01\x00\x02\x11\x03\x11\x00\x3f\x00"
    return bytes(b)


def to_u8(frame):
    """cv2 imencode's ``image.convertTo(CV_8U)`` for a float frame: round-half-even, saturate."""
    a = np.asarray(frame)
    if a.dtype == np.uint8:
        return a
    return np.clip(np.rint(a.astype(np.float64)), 0, 255).astype(np.uint8)


def rgb_to_ycc(rgb):
    """jccolor.c rgb_ycc_convert, 16-bit fixed point."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half = 1 << 15
    y = (19595 * r + 38470 * g + 7471 * b + half) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + half - 1) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + half - 1) >> 16
    return y, cb, cr


def _pad_edge(p, h, w):
    return np.pad(p, ((0, h - p.shape[0]), (0, w - p.shape[1])), mode="edge")


def h2v2_downsample(p):
    """jcsample.c h2v2_downsample: 2x2 box, bias alternating 1,2 along the row."""
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)
    return (s + bias) >> 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_1d(d, first):
    """jfdctint.c one pass over the LAST axis (8 samples); `first` selects the row-pass scaling."""
    CB, PB = 13, 2
    t0, t7 = d[..., 0] + d[..., 7], d[..., 0] - d[..., 7]
    t1, t6 = d[..., 1] + d[..., 6], d[..., 1] - d[..., 6]
    t2, t5 = d[..., 2] + d[..., 5], d[..., 2] - d[..., 5]
    t3, t4 = d[..., 3] + d[..., 4], d[..., 3] - d[..., 4]
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o = [None] * 8
    if first:
        o[0], o[4] = (t10 + t11) << PB, (t10 - t11) << PB
        sh = CB - PB
    else:
        o[0], o[4] = _descale(t10 + t11, PB), _descale(t10 - t11, PB)
        sh = CB + PB
    z1 = (t12 + t13) * 4433
    o[2] = _descale(z1 + t13 * 6270, sh)
    o[6] = _descale(z1 - t12 * 15137, sh)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * 9633
    t4, t5, t6, t7 = t4 * 2446, t5 * 16819, t6 * 25172, t7 * 12299
    z1, z2, z3, z4 = z1 * -7373, z2 * -20995, z3 * -16069 + z5, z4 * -3196 + z5
    o[7] = _descale(t4 + z1 + z3, sh)
    o[5] = _descale(t5 + z2 + z4, sh)
    o[3] = _descale(t6 + z2 + z3, sh)
    o[1] = _descale(t7 + z1 + z4, sh)
    return np.stack(o, axis=-1)


def fdct_quant(blocks, qtab):
    """blocks [...,8,8] samples 0..255 -> quantised coefficients [...,64] natural order (jfdctint.c + jcdctmgr.c)."""
    d = blocks.astype(np.int64) - 128
    d = _fdct_1d(d, True)                                   # rows
    d = np.swapaxes(_fdct_1d(np.swapaxes(d, -1, -2), False), -1, -2)   # columns
    c = d.reshape(d.shape[:-2] + (64,))
    q8 = qtab.astype(np.int64) << 3
    mag = (np.abs(c) + (q8 >> 1)) // q8
    return np.where(c < 0, -mag, mag)


def _blocks(plane):
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).swapaxes(1, 2)          # [by, bx, 8, 8]


def mcu_coefficients(rgb_u8, quality):
    """-> int array [mcu_rows, mcu_cols, 6, 64] (Y00 Y01 Y10 Y11 Cb Cr; natural order), dummy blocks as jccoefct.c makes them."""
    H, W, _ = rgb_u8.shape
    ql, qc = quant_tables(quality)
    y, cb, cr = rgb_to_ycc(rgb_u8)
    mr, mc = -(-H // 16), -(-W // 16)
    ybw, ybh = -(-W // 8), -(-H // 8)                      # real luma blocks across / down
    # Sample padding.  Columns: the INPUT of the downsampler is edge-replicated to the padded width
    # (jcsample.c expand_right_edge).  Rows: the colour rows are replicated only up to an even count
    # (jcprepct.c pre_process_data, one row group), the planes are downsampled, and then the DOWNSAMPLED
    # planes are replicated down to the iMCU height (expand_bottom_edge on output_buf) -- for an even H
    # that is not a multiple of 16 (1080!) the last chroma row is repeated, not re-derived.
    He = H + (H & 1)
    yq = fdct_quant(_blocks(_pad_edge(y, mr * 16, mc * 16)), ql)       # [2mr, 2mc, 64]
    cbq = fdct_quant(_blocks(_pad_edge(h2v2_downsample(_pad_edge(cb, He, mc * 16)), mr * 8, mc * 8)), qc)
    crq = fdct_quant(_blocks(_pad_edge(h2v2_downsample(_pad_edge(cr, He, mc * 16)), mr * 8, mc * 8)), qc)
    out = np.zeros((mr, mc, 6, 64), dtype=np.int64)
    for i in range(mr):
        for j in range(mc):
            blk = out[i, j]
            for yi in range(2):
                by = 2 * i + yi
                if by < ybh:
                    for xi in range(2):
                        bx = 2 * j + xi
                        if bx < ybw:
                            blk[2 * yi + xi] = yq[by, bx]
                        else:                                  # right-edge dummy: zero AC, DC of the block before it
                            blk[2 * yi + xi, 0] = blk[2 * yi + xi - 1, 0]
                else:                                          # bottom dummy row: DC of the last block of the row above
                    blk[2 * yi, 0] = blk[2 * yi - 1, 0]
                    blk[2 * yi + 1, 0] = blk[2 * yi - 1, 0]
            blk[4] = cbq[i, j]
            blk[5] = crq[i, j]
    return out


class _BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 255
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put(0x7F, 8 - self.n)                        # jchuff.c flush_bits: pad with 1-bits


def entropy_encode(coefs):
    """jchuff.c encode_one_block over all MCUs in raster order -> stuffed entropy-coded bytes."""
    dc = (huff_codes(DC_LUMA_BITS, DC_VALS), huff_codes(DC_CHROMA_BITS, DC_VALS))
    ac = (huff_codes(AC_LUMA_BITS, AC_LUMA_VALS), huff_codes(AC_CHROMA_BITS, AC_CHROMA_VALS))
    bw = _BitWriter()
    last = [0, 0, 0]
    mr, mc = coefs.shape[:2]
    for i in range(mr):
        for j in range(mc):
            for b in range(6):
                comp = 0 if b < 4 else b - 3
                tbl = 0 if b < 4 else 1
                blk = coefs[i, j, b]
                v = int(blk[0])
                diff = v - last[comp]
                last[comp] = v
                t, t2 = (diff, diff) if diff >= 0 else (-diff, diff - 1)
                nb = t.bit_length()
                bw.put(*dc[tbl][nb])
                if nb:
                    bw.put(t2, nb)
                run = 0
                for k in range(1, 64):
                    v = int(blk[ZIGZAG[k]])
                    if v == 0:
                        run += 1
                        continue
                    while run > 15:
                        bw.put(*ac[tbl][0xF0])
                        run -= 16
                    t, t2 = (v, v) if v >= 0 else (-v, v - 1)
                    nb = t.bit_length()
                    bw.put(*ac[tbl][(run << 4) + nb])
                    bw.put(t2, nb)
                    run = 0
                if run:
                    bw.put(*ac[tbl][0])
    bw.flush()
    return bytes(bw.out)


def encode_jpeg(frame, quality=90):
    """HWC RGB frame (uint8, or float 0..255 as make_sbs returns it) -> JPEG bytes (reference streamer.py:285-291)."""
    rgb = to_u8(frame)
    H, W, _ = rgb.shape
    return header(H, W, quality) + entropy_encode(mcu_coefficients(rgb, quality)) + b"\xff\xd9"
