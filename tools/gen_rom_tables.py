#!/usr/bin/env python3
"""Generate rom_tables.h: the constant tables of an RFC 6716 (Opus) decoder.

Two kinds of table live here:
  * DERIVED tables are computed from their defining formulas (PVQ codebook-size recurrence,
    MDCT/FFT twiddles, the power-complementary window, the mixed-radix digit-reversal
    permutations).  `tools/check_rom_tables.py` compares them value-for-value with the
    reference's literals when /root/reference is present (container only).
  * NORMATIVE tables are constants of the Opus standard itself (band layout, allocation
    matrix, pulse cache, Laplace model parameters, SILK codebooks and iCDFs).  They cannot be
    derived; they are listed below as plain data in this project's own arrangement.
    Reference locations: src/celt.cpp:185-587 (CELT), src/silk.cpp:43-412 (SILK).

The output is written twice, byte-identical: oracle/rom_tables.h (CPU oracle) and
esp32-opus-player_amd/csrc/rom_tables.h (HIP product).  Each array is declared as
`OPUS_ROM <type> name[]`; the including file defines OPUS_ROM (`static const` on the host,
`static __device__ const` in HIP code).
"""
import math, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# --------------------------------------------------------------------------------------------
# fixed-point helpers needed by the derivations (same arithmetic as the decoder's cos approx)
def s16(x):
    x &= 0xFFFF
    return x - 65536 if x >= 32768 else x

def mul_p15(a, b):
    return (16384 + s16(a) * s16(b)) >> 15

def cos_quarter(x):              # cos(pi/2 * x/32768), Q15
    x2 = s16(mul_p15(x, x))
    v = (32767 - x2) + mul_p15(x2, (-7651 + mul_p15(x2, (8277 + mul_p15(-626, x2)))))
    return s16(1 + min(32766, v))

def cos_norm(x):                 # cos(pi * x/65536), Q15, x is a 17-bit phase
    x &= 0x1FFFF
    if x > (1 << 16):
        x = (1 << 17) - x
    if x & 0x7FFF:
        if x < (1 << 15):
            return cos_quarter(s16(x))
        return s16(-cos_quarter(s16(65536 - x)))
    if x & 0xFFFF:
        return 0
    if x & 0x1FFFF:
        return -32767
    return 32767

def trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q

# --------------------------------------------------------------------------------------------
# DERIVED tables
def pvq_u_table(rows=15, cols=177):
    """U(n,k): number of PVQ codewords of dimension n with k pulses whose first entry is >= 0
    (RFC 6716 sec. 4.3.4.2).  U(n,k)=U(n-1,k)+U(n,k-1)+U(n-1,k-1); dense [rows][cols], mod 2^32
    (entries that overflow 32 bits are never addressed by a legal (N,K))."""
    U = [[0] * cols for _ in range(cols)]
    U[0][0] = 1
    for n in range(cols):
        for k in range(cols):
            if n == 0 or k == 0:
                continue
            U[n][k] = U[n - 1][k] + U[n][k - 1] + U[n - 1][k - 1]
    return [U[n][k] & 0xFFFFFFFF for n in range(rows) for k in range(cols)]

def pvq_u_rows():
    """U(lo, hi) for lo = 4 .. 14 stored by ROW lo, every column hi = 0 .. the last one whose entry fits 32 bits, rows back to back:
    entry (lo, hi) is at RB[lo] + hi.  What the leaf walk of the split path searches at a fixed number of pulses k < n -- the next
    pulse's dimension -- lies along rows k and k + 1, so a probe is two independent reads off two bases that only change when k
    does; the candidates of a pulse's size (rows 4 .. 7 at one column) need no base look-up at all; and a leaf with n <= k
    dimensions left searches its pulse's size along row n alone, which is why the rows are complete (U is symmetric: the columns
    below lo repeat entries of earlier rows, 99 words).  Rows 0..3 have closed forms (0 / 1 / 2h-1 / 2h(h-1)+1).
    Returns (table, RB[0..15])."""
    cols = 177
    U = [[0] * cols for _ in range(cols)]
    U[0][0] = 1
    for n in range(1, cols):
        for k in range(1, cols):
            U[n][k] = U[n - 1][k] + U[n][k - 1] + U[n - 1][k - 1]
    tab, rb = [], [0] * 16
    for lo in range(4, 15):
        rb[lo] = len(tab)
        row = [U[lo][hi] for hi in range(cols)]
        assert all(a <= b for a, b in zip(row, row[1:]))
        tab += [v for v in row if v < 2 ** 32]
    for lo in range(4):
        rb[lo] = rb[4]
    rb[15] = rb[14]
    assert min(rb) >= 0 and max(rb) < 65536
    return tab, rb

def pulse_v_table():
    """V(N, K) = U(N, K) + U(N, K + 1), the size of the PVQ codebook a leaf's index is decoded against (celt.cpp:2622,
    ec_dec_uint's ft), laid out like the pulse cache: entry rom_pulse_idx[(LM + 1) * 21 + band] + q holds V for the band's
    N = width << LM (width >> 1 for LM = -1) and K = get_pulses(q).  The entropy half of the split path finds it with the
    index it already has for the cache instead of two lookups in the big U table."""
    cols = 178
    U = [[0] * cols for _ in range(cols)]
    U[0][0] = 1
    for n in range(1, cols):
        for k in range(1, cols):
            U[n][k] = U[n - 1][k] + U[n][k - 1] + U[n - 1][k - 1]
    out = [0] * len(PULSE_BITS)
    for lmp1 in range(5):
        for band in range(21):
            base = PULSE_IDX[lmp1 * 21 + band]
            if base < 0:
                continue
            width = EBAND[band + 1] - EBAND[band]
            n = width << (lmp1 - 1) if lmp1 >= 1 else width >> 1
            for q in range(1, PULSE_BITS[base] + 1):
                k = q if q < 8 else (8 + (q & 7)) << ((q >> 3) - 1)
                lo, hi = min(n, k), max(n, k)
                lo1, hi1 = min(n, k + 1), max(n, k + 1)
                v = U[lo][hi] + U[lo1][hi1]
                assert 0 < v < 2 ** 32, (lmp1, band, q, n, k, v)
                assert out[base + q] in (0, v), "two caches share an entry with different N"
                out[base + q] = v
    return out

def mdct_trig():
    """cos(2*pi*(i+1/8)/N) in Q15 for N = 1920, 960, 480, 240 (N/2 entries each), concatenated."""
    out = []
    N = 1920
    for _ in range(4):
        for i in range(N >> 1):
            v = math.floor(0.5 + 32768.0 * math.cos(2.0 * math.pi * (i + 0.125) / N))
            out.append(max(-32767, min(32767, v)))
        N >>= 1
    return out

def fft_twiddles(nfft=480):
    """e^{-2*pi*i*k/480} in Q15 through the decoder's own integer cosine (r,i interleaved)."""
    out = []
    for i in range(nfft):
        ph = trunc_div((-i) << 17, nfft)
        out += [cos_norm(ph), cos_norm(ph - 32768)]
    return out

def window120(n=120):
    """Vorbis power-complementary window, Q15, capped at 32767."""
    return [min(32767, math.floor(0.5 + 32768.0 * math.sin(0.5 * math.pi * math.sin(0.5 * math.pi * (i + 0.5) / n) ** 2)))
            for i in range(n)]

def digit_reversal(factors, n):
    """Output position of input sample f for a decimation-in-time mixed-radix FFT with the
    given (radix, remaining-length) factor pairs."""
    out = [0] * n
    def rec(base, f, stride, fac):
        p, m = fac[0], fac[1]
        if m == 1:
            for j in range(p):
                out[f] = base + j
                f += stride
        else:
            for j in range(p):
                rec(base, f, stride * p, fac[2:])
                f += stride
                base += m
    rec(0, 0, 1, factors)
    return out

FFT_FACTORS = {
    480: [5, 96, 3, 32, 4, 8, 2, 4, 4, 1],
    240: [5, 48, 3, 16, 4, 4, 4, 1],
    120: [5, 24, 3, 8, 2, 4, 4, 1],
    60:  [5, 12, 3, 4, 4, 1],
}

# --------------------------------------------------------------------------------------------
# NORMATIVE CELT data (48 kHz / 960 standard mode)
BAND_ALLOC = [
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    90, 80, 75, 69, 63, 56, 49, 40, 34, 29, 20, 18, 10, 0, 0, 0, 0, 0, 0, 0, 0,
    110, 100, 90, 84, 78, 71, 65, 58, 51, 45, 39, 32, 26, 20, 12, 0, 0, 0, 0, 0, 0,
    118, 110, 103, 93, 86, 80, 75, 70, 65, 59, 53, 47, 40, 31, 23, 15, 4, 0, 0, 0, 0,
    126, 119, 112, 104, 95, 89, 83, 78, 72, 66, 60, 54, 47, 39, 32, 25, 17, 12, 1, 0, 0,
    134, 127, 120, 114, 103, 97, 91, 85, 78, 72, 66, 60, 54, 47, 41, 35, 29, 23, 16, 10, 1,
    144, 137, 130, 124, 113, 107, 101, 95, 88, 82, 76, 70, 64, 57, 51, 45, 39, 33, 26, 15, 1,
    152, 145, 138, 132, 123, 117, 111, 105, 98, 92, 86, 80, 74, 67, 61, 55, 49, 43, 36, 20, 1,
    162, 155, 148, 142, 133, 127, 121, 115, 108, 102, 96, 90, 84, 77, 71, 65, 59, 53, 46, 30, 1,
    172, 165, 158, 152, 143, 137, 131, 125, 118, 112, 106, 100, 94, 87, 81, 75, 69, 63, 56, 45, 20,
    200, 200, 200, 200, 200, 200, 200, 200, 198, 193, 188, 183, 178, 173, 168, 163, 158, 153, 148, 129, 104,
]

EBAND = [
    0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100,
]

LOGN = [
    0, 0, 0, 0, 0, 0, 0, 0, 8, 8, 8, 8, 16, 16, 16, 21, 21, 24, 29, 34, 36,
]

PULSE_IDX = [
    -1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 41, 41, 41, 82, 82, 123, 164, 200, 222,
    0, 0, 0, 0, 0, 0, 0, 0, 41, 41, 41, 41, 123, 123, 123, 164, 164, 240, 266, 283, 295,
    41, 41, 41, 41, 41, 41, 41, 41, 123, 123, 123, 123, 240, 240, 240, 266, 266, 305, 318, 328, 336,
    123, 123, 123, 123, 123, 123, 123, 123, 240, 240, 240, 240, 305, 305, 305, 318, 318, 343, 351, 358, 364,
    240, 240, 240, 240, 240, 240, 240, 240, 305, 305, 305, 305, 343, 343, 343, 351, 351, 370, 376, 382, 387,
]

PULSE_BITS = [
    40, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7,
    7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 40, 15, 23, 28, 31, 34, 36, 38, 39, 41, 42, 43, 44, 45, 46,
    47, 47, 49, 50, 51, 52, 53, 54, 55, 55, 57, 58, 59, 60, 61, 62, 63, 63, 65, 66, 67, 68, 69, 70, 71, 71, 40, 20,
    33, 41, 48, 53, 57, 61, 64, 66, 69, 71, 73, 75, 76, 78, 80, 82, 85, 87, 89, 91, 92, 94, 96, 98, 101, 103, 105, 107,
    108, 110, 112, 114, 117, 119, 121, 123, 124, 126, 128, 40, 23, 39, 51, 60, 67, 73, 79, 83, 87, 91, 94, 97, 100, 102, 105, 107,
    111, 115, 118, 121, 124, 126, 129, 131, 135, 139, 142, 145, 148, 150, 153, 155, 159, 163, 166, 169, 172, 174, 177, 179, 35, 28, 49, 65,
    78, 89, 99, 107, 114, 120, 126, 132, 136, 141, 145, 149, 153, 159, 165, 171, 176, 180, 185, 189, 192, 199, 205, 211, 216, 220, 225, 229,
    232, 239, 245, 251, 21, 33, 58, 79, 97, 112, 125, 137, 148, 157, 166, 174, 182, 189, 195, 201, 207, 217, 227, 235, 243, 251, 17, 35,
    63, 86, 106, 123, 139, 152, 165, 177, 187, 197, 206, 214, 222, 230, 237, 250, 25, 31, 55, 75, 91, 105, 117, 128, 138, 146, 154, 161,
    168, 174, 180, 185, 190, 200, 208, 215, 222, 229, 235, 240, 245, 255, 16, 36, 65, 89, 110, 128, 144, 159, 173, 185, 196, 207, 217, 226,
    234, 242, 250, 11, 41, 74, 103, 128, 151, 172, 191, 209, 225, 241, 255, 9, 43, 79, 110, 138, 163, 186, 207, 227, 246, 12, 39, 71,
    99, 123, 144, 164, 182, 198, 214, 228, 241, 253, 9, 44, 81, 113, 142, 168, 192, 214, 235, 255, 7, 49, 90, 127, 160, 191, 220, 247,
    6, 51, 95, 134, 170, 203, 234, 7, 47, 87, 123, 155, 184, 212, 237, 6, 52, 97, 137, 174, 208, 240, 5, 57, 106, 151, 192, 231,
    5, 59, 111, 158, 202, 243, 5, 55, 103, 147, 187, 224, 5, 60, 113, 161, 206, 248, 4, 65, 122, 175, 224, 4, 67, 127, 182, 234,
]

PULSE_CAPS = [
    224, 224, 224, 224, 224, 224, 224, 224, 160, 160, 160, 160, 185, 185, 185, 178, 178, 168, 134, 61, 37,
    224, 224, 224, 224, 224, 224, 224, 224, 240, 240, 240, 240, 207, 207, 207, 198, 198, 183, 144, 66, 40,
    160, 160, 160, 160, 160, 160, 160, 160, 185, 185, 185, 185, 193, 193, 193, 183, 183, 172, 138, 64, 38,
    240, 240, 240, 240, 240, 240, 240, 240, 207, 207, 207, 207, 204, 204, 204, 193, 193, 180, 143, 66, 40,
    185, 185, 185, 185, 185, 185, 185, 185, 193, 193, 193, 193, 193, 193, 193, 183, 183, 172, 138, 65, 39,
    207, 207, 207, 207, 207, 207, 207, 207, 204, 204, 204, 204, 201, 201, 201, 188, 188, 176, 141, 66, 40,
    193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 193, 194, 194, 194, 184, 184, 173, 139, 65, 39,
    204, 204, 204, 204, 204, 204, 204, 204, 201, 201, 201, 201, 198, 198, 198, 187, 187, 175, 140, 66, 40,
]

LOG2_FRAC = [
    0, 8, 13, 16, 19, 21, 23, 24, 26, 27, 28, 29, 30, 31, 32, 32, 33, 34, 34, 35, 36, 36, 37, 37,
]

EMEANS = [
    103, 100, 92, 85, 81, 77, 72, 70, 78, 75, 73, 71, 78, 74, 69, 72, 70, 74, 76, 71, 60, 60, 60, 60, 60,
]

EPROB = [
    72, 127, 65, 129, 66, 128, 65, 128, 64, 128, 62, 128, 64, 128, 64, 128, 92, 78, 92, 79, 92, 78, 90, 79, 116, 41, 115, 40, 114, 40, 132, 26, 132, 26, 145, 17, 161, 12, 176, 10, 177, 11,
    24, 179, 48, 138, 54, 135, 54, 132, 53, 134, 56, 133, 55, 132, 55, 132, 61, 114, 70, 96, 74, 88, 75, 88, 87, 74, 89, 66, 91, 67, 100, 59, 108, 50, 120, 40, 122, 37, 97, 43, 78, 50,
    83, 78, 84, 81, 88, 75, 86, 74, 87, 71, 90, 73, 93, 74, 93, 74, 109, 40, 114, 36, 117, 34, 117, 34, 143, 17, 145, 18, 146, 19, 162, 12, 165, 10, 178, 7, 189, 6, 190, 8, 177, 9,
    23, 178, 54, 115, 63, 102, 66, 98, 69, 99, 74, 89, 71, 91, 73, 91, 78, 89, 86, 80, 92, 66, 93, 64, 102, 59, 103, 60, 104, 60, 117, 52, 123, 44, 138, 35, 133, 31, 97, 38, 77, 45,
    61, 90, 93, 60, 105, 42, 107, 41, 110, 45, 116, 38, 113, 38, 112, 38, 124, 26, 132, 27, 136, 19, 140, 20, 155, 14, 159, 16, 158, 18, 170, 13, 177, 10, 187, 8, 192, 6, 175, 9, 159, 10,
    21, 178, 59, 110, 71, 86, 75, 85, 84, 83, 91, 66, 88, 73, 87, 72, 92, 75, 98, 72, 105, 58, 107, 54, 115, 52, 114, 55, 112, 56, 129, 51, 132, 40, 150, 33, 140, 29, 98, 35, 77, 42,
    42, 121, 96, 66, 108, 43, 111, 40, 117, 44, 123, 32, 120, 36, 119, 33, 127, 33, 134, 34, 139, 21, 147, 23, 152, 20, 158, 25, 154, 26, 166, 21, 173, 16, 184, 13, 184, 10, 150, 13, 139, 15,
    22, 178, 63, 114, 74, 82, 84, 83, 92, 82, 103, 62, 96, 72, 96, 67, 101, 73, 107, 72, 113, 55, 118, 52, 125, 52, 118, 52, 117, 55, 135, 49, 137, 39, 157, 32, 145, 29, 97, 33, 77, 40,
]

# --------------------------------------------------------------------------------------------
# NORMATIVE SILK data is appended by the SILK section below (SILK_TABLES: name -> (ctype, values))
SILK_TABLES = {}
sys.path.insert(0, HERE)
try:
    from silk_rom_data import SILK_TABLES as _S   # tools/silk_rom_data.py
    SILK_TABLES = _S
except ImportError:
    pass

# int16 tables that the kernels index with wave-uniform indices are stored as int32: on gfx950 the compiler
# turns adjacent sub-dword scalar loads into one s_load_dword, and a pair that starts on an odd 16-bit index is
# split into a misaligned base + immediate that the scalar memory path does not honour.  (Lane-indexed tables
# such as twiddles and windows stay 16-bit: they are fetched with vector loads.)
WIDEN = {"rom_eband", "rom_logn", "rom_pulse_idx", "rom_silk_cos_q12", "rom_silk_stereo_pred_q13",
         "rom_silk_nb_cb1_wght_q9", "rom_silk_wb_cb1_wght_q9", "rom_silk_nb_delta_min_q15",
         "rom_silk_wb_delta_min_q15", "rom_silk_ltp_scales_q14", "rom_silk_quant_offsets_q10",
         "rom_silk_up2_hq0", "rom_silk_up2_hq1"}


# SILK's inverse-CDF tables are searched by a whole wave at once (lane l looks at entry l, og_range.hpp: rc_icdf), from any
# starting offset inside a table: 64 bytes of zero padding behind each keep those reads inside the array.
ICDF_PAD = 64


def emit(name, ctype, vals, per=16):
    if name in WIDEN:
        ctype = "int32_t"
    size = len(vals) + (ICDF_PAD if name.startswith("rom_silk") and ctype == "uint8_t" else 0)
    s = f"OPUS_ROM {ctype} {name}[{size}] = {{\n"
    for i in range(0, len(vals), per):
        s += "    " + ", ".join(str(v) for v in vals[i:i + per]) + ",\n"
    return s + "};\n\n"

def build_text():
    t = "/* GENERATED by tools/gen_rom_tables.py -- do not edit. */\n"
    t += "#ifndef OPUS_ROM_TABLES_H\n#define OPUS_ROM_TABLES_H\n#include <stdint.h>\n"
    t += "#ifndef OPUS_ROM\n#define OPUS_ROM static const\n#endif\n\n"
    t += "#define ROM_PVQ_COLS 177\n"
    t += emit("rom_pvq_u", "uint32_t", pvq_u_table(), 8)
    # the same table padded to 16 rows x 192 columns: lane l of register (row, seg) holds U(row, 64*seg + l)
    u = pvq_u_table()
    u192 = []
    for r in range(16):
        for c in range(192):
            u192.append(u[r * 177 + c] if (r < 15 and c < 177) else 0)
    t += emit("rom_pvq_u192", "uint32_t", u192, 8)
    # rows 4..14 by column for an LDS copy (the split path's leaf pass)
    ur, rb = pvq_u_rows()
    t += "#define ROM_PVQ_RR_LEN %d\n" % len(ur)
    t += "".join("#define ROM_PVQ_RB%d %d\n" % (r, rb[r]) for r in range(4, 15))
    t += emit("rom_pvq_rr", "uint32_t", ur, 8)
    t += emit("rom_pvq_rb", "uint16_t", rb, 16)
    t += emit("rom_band_alloc", "uint8_t", BAND_ALLOC, 21)
    t += emit("rom_eband", "int16_t", EBAND, 22)
    t += emit("rom_logn", "int16_t", LOGN, 21)
    # 5 ms bin (= 8 coefficients of a 20 ms frame) -> band that holds it, and the bin's index within that band
    b2b = [max(b for b in range(21) if EBAND[b] <= x) for x in range(100)]
    t += emit("rom_bin2band", "uint8_t", b2b, 25)
    t += emit("rom_binoff", "uint8_t", [x - EBAND[b2b[x]] for x in range(100)], 25)
    t += emit("rom_pulse_idx", "int16_t", PULSE_IDX, 21)
    t += emit("rom_pulse_bits", "uint8_t", PULSE_BITS, 28)
    t += emit("rom_pulse_v", "uint32_t", pulse_v_table(), 8)
    t += emit("rom_pulse_caps", "uint8_t", PULSE_CAPS, 21)
    t += emit("rom_log2_frac", "uint8_t", LOG2_FRAC, 24)
    t += emit("rom_emeans", "int8_t", EMEANS, 25)
    t += emit("rom_eprob", "uint8_t", EPROB, 42)
    t += emit("rom_mdct_trig", "int16_t", mdct_trig(), 12)
    t += emit("rom_fft_tw", "int16_t", fft_twiddles(), 12)
    t += emit("rom_win120", "int16_t", window120(), 12)
    # pairs a lane fetches together, packed into one word (low half | high half << 16): the long block's pre-rotation
    # twiddles (trig[i], trig[480 + i]) and the FFT twiddles (real, imaginary)
    trig, tw = mdct_trig(), fft_twiddles()
    t += emit("rom_prerot480", "uint32_t", [(trig[i] & 0xFFFF) | (trig[480 + i] & 0xFFFF) << 16 for i in range(480)], 8)
    t += emit("rom_fft_tw32", "uint32_t", [(tw[2 * i] & 0xFFFF) | (tw[2 * i + 1] & 0xFFFF) << 16 for i in range(480)], 8)
    # the noise generator jumped ahead by n = 1 .. 192 steps (celt_lcg_rand celt.cpp:921 composed with itself: s -> a s + c mod 2^32)
    jump, a, c = [], 1, 0
    for _ in range(192):
        a, c = (1664525 * a) & 0xFFFFFFFF, (1664525 * c + 1013904223) & 0xFFFFFFFF
        jump += [a, c]
    t += emit("rom_lcg_jump", "uint32_t", jump, 8)
    for n in (480, 240, 120, 60):
        t += emit(f"rom_bitrev{n}", "int16_t", digit_reversal(FFT_FACTORS[n], n), 20)
    for name, (ctype, vals) in SILK_TABLES.items():
        t += emit(name, ctype, vals, 16)
    if "rom_silk_frac_fir12" in SILK_TABLES:
        # the 12 phases of the 8-tap interpolation filter (silk_resampler_private_IIR_FIR_INTERPOL, src/silk.cpp:3451-3472) as the
        # kernel multiplies them: per phase t the taps that meet input samples b[0..7] in order -- row t of the table forwards, row
        # 11 - t backwards -- packed two to a word (low half first) for v_dot2_i32_i16
        f = SILK_TABLES["rom_silk_frac_fir12"][1]
        packed = []
        for ph in range(12):
            taps = list(f[4 * ph:4 * ph + 4]) + list(reversed(f[4 * (11 - ph):4 * (11 - ph) + 4]))
            packed += [(taps[2 * i] & 0xFFFF) | (taps[2 * i + 1] & 0xFFFF) << 16 for i in range(4)]
        t += emit("rom_silk_fir12_taps8", "uint32_t", packed, 4)
    # Every 8-bit SILK table once more as ONE blob (+ offsets): the lane-per-frame SILK parse kernel copies it to LDS in
    # one go and addresses tables as SILK_BLOB_<name> + index.
    blob, offs = [], []
    for name, (ctype, vals) in SILK_TABLES.items():
        if ctype == "uint8_t":
            offs.append((name[len("rom_silk_"):], len(blob)))
            blob += list(vals)
    if blob:
        blob += [0] * ((-len(blob)) % 4)
        t += "enum {\n" + "".join(f"    SILK_BLOB_{n} = {o},\n" for n, o in offs) + f"    SILK_BLOB_SIZE = {len(blob)}\n}};\n"
        t += emit("rom_silk_u8_blob", "uint8_t", blob, 24).replace(f"[{len(blob) + ICDF_PAD}]", f"[{len(blob)}]")
    t += "#endif\n"
    return t

def main():
    text = build_text()
    for rel in ("oracle/rom_tables.h", "esp32-opus-player_amd/csrc/rom_tables.h"):
        p = os.path.join(ROOT, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        if not os.path.exists(p) or open(p).read() != text:
            open(p, "w").write(text)
    return 0

if __name__ == "__main__":
    sys.exit(main())
