// og_celt_bands.hpp -- CELT band decoding for one frame per wavefront: PVQ shape decode,
// spreading rotation ("pulse de-spread"), folding / noise fill, time-frequency Haar and Hadamard
// reordering, stereo split and merge, anti-collapse.
//
// Vectors live in the LDS arena S.v and are addressed by index (X, the folding history `norm`, the
// pulse vector and a scratch row).  The bit-budget bookkeeping and every range-decoder read are
// wave-uniform scalar code; the per-coefficient loops are split over the 64 lanes.
// Reference behaviour: src/celt.cpp:684-815 (rotation, residual normalisation, collapse mask,
// renormalise), :1010-1082 (anti-collapse), :1113-1213 (stereo merge, Hadamard, Haar),
// :1215-1355 (theta), :1357-1741 (band / partition decode), :1754-1924 (band loop).
#pragma once
#include "og_celt_math.hpp"

namespace og {

constexpr int BITRES = 3;

struct BandCtx {
    int band, intensity, spread, tf_change, disable_inv;
    i32 remaining_bits;
    u32 seed;
};

OG_DEV const u8 *pulse_cache(int band, int LM) { return rom_pulse_bits + rom_pulse_idx[(LM + 1) * NBANDS + band]; }

OG_DEV int bits2pulses(int band, int LM, int bits) { // celt.h:537
    const u8 *cache = pulse_cache(band, LM);
    int lo = 0, hi = cache[0];
    bits--;
    for (int i = 0; i < 6; i++) {
        int mid = (lo + hi + 1) >> 1;
        if ((int)cache[mid] >= bits) hi = mid; else lo = mid;
    }
    return bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits ? lo : hi;
}
OG_DEV int pulses2bits(int band, int LM, int pulses) { return pulses == 0 ? 0 : pulse_cache(band, LM)[pulses] + 1; }
OG_DEV int get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); } // celt.h:533

// ---- lane-parallel vector helpers ----------------------------------------------------------------
// scale X[0..N) so that its norm becomes `gain` (renormalise_vector celt.cpp:797)
OG_DEV void renormalise(int x, int N, i32 gain) {
    OG_SYNC();
    i32 part = 0;
    OG_FOR_LANES(j, N) part += mul16(S.v[x + j], S.v[x + j]);
    i32 E = 1 + wave_sum(part);
    int k = ilog2(E) >> 1;
    i32 t = vshr32(E, 2 * (k - 7));
    i32 g = tr16(mul16_p15(rsqrt_norm(t), gain));
    OG_FOR_LANES(j, N) S.v[x + j] = (i16)pshr32(mul16(g, S.v[x + j]), k + 1);
    OG_SYNC();
}

// One pass of the 2-tap lattice along chains x[r], x[r+stride], ... (exp_rotation1 celt.cpp:684).
// Inside one block of `len` samples the chains r = 0..stride-1 are independent of each other, and
// the `nblk` blocks are independent too: one lane per (block, chain); along a chain it is serial.
OG_DEV void rotate_pass(int x, int len, int nblk, int stride, i32 c, i32 s) {
    OG_SYNC();
    i32 ms = tr16(-s);
    int nchain = nblk * stride;
    OG_FOR_LANES(id, nchain) {
        int blk = id / stride, r = id - blk * stride;
        int base = x + blk * len;
        // forward: i = r, r+stride, ... while i < len - stride
        int i = r;
        for (; i < len - stride; i += stride) {
            i32 x1 = S.v[base + i], x2 = S.v[base + i + stride];
            S.v[base + i + stride] = (i16)pshr32(mul16(c, x2) + mul16(s, x1), 15);
            S.v[base + i] = (i16)pshr32(mul16(c, x1) + mul16(ms, x2), 15);
        }
        // backward: i = len-2*stride-1 down to 0, restricted to this chain's residue
        int last = len - 2 * stride - 1;
        if (last >= 0) {
            int top = last - ((last - r) % stride + stride) % stride; // largest i <= last with i % stride == r
            for (i = top; i >= 0; i -= stride) {
                i32 x1 = S.v[base + i], x2 = S.v[base + i + stride];
                S.v[base + i + stride] = (i16)pshr32(mul16(c, x2) + mul16(s, x1), 15);
                S.v[base + i] = (i16)pshr32(mul16(c, x1) + mul16(ms, x2), 15);
            }
        }
    }
    OG_SYNC();
}

// inverse spreading rotation (exp_rotation celt.cpp:707 with dir = -1)
OG_DEV void unspread(int x, int len, int stride, int K, int spread) {
    if (2 * K >= len || spread == 0) return;
    int factor = spread == 1 ? 15 : (spread == 2 ? 10 : 5);
    i32 gain = tr16(mul32_q31(mul16(32767, len), celt_rcp(len + factor * K))); // celt_div celt.h:367
    i32 theta = tr16(mul16_q15(gain, gain) >> 1);
    i32 c = cos_norm(theta);
    i32 s = cos_norm(sub16(32767, theta));
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    int blen = (int)udiv((u32)len, (u32)stride);
    if (stride2) rotate_pass(x, blen, stride, stride2, s, c);
    rotate_pass(x, blen, stride, 1, c, s);
}

// PVQ leaf: decode K pulses into X[0..N) with norm `gain` (alg_unquant celt.cpp:782)
OG_DEVN u32 pvq_unquant(Rc &rc, int x, int N, int K, int spread, int B, i32 gain) {
    i32 Ryy = pvq_decode_index(N, K, rc_uint(rc, pvq_v(N, K)));
    int k = ilog2(Ryy) >> 1;
    i32 t = vshr32(Ryy, 2 * (k - 7));
    i32 g = tr16(mul16_p15(rsqrt_norm(t), gain));
    OG_SYNC();
    OG_FOR_LANES(j, N) S.v[x + j] = (i16)pshr32(mul16(g, S.v[V_IY + j]), k + 1); // normalise_residual :745
    unspread(x, N, B, K, spread);
    if (B <= 1) return 1;
    u32 N0 = udiv((u32)N, (u32)B), m = 0; // extract_collapse_mask :760
    OG_FOR_LANES(j, N) m |= (u32)(S.v[V_IY + j] != 0) << udiv((u32)j, N0);
    m = wave_or(m);
    OG_SYNC();
    return m;
}

OG_DEV void haar1(int x, int N0, int stride) { // celt.cpp:1202
    N0 >>= 1;
    OG_SYNC();
    OG_FOR_LANES(id, N0 * stride) {
        int j = id / stride, i = id - j * stride;
        int a = x + stride * 2 * j + i, b = x + stride * (2 * j + 1) + i;
        i32 t1 = mul16(23170, S.v[a]), t2 = mul16(23170, S.v[b]);
        S.v[a] = (i16)pshr32(t1 + t2, 15);
        S.v[b] = (i16)pshr32(t1 - t2, 15);
    }
    OG_SYNC();
}

// natural <-> "ordery" Hadamard index tables for stride 2, 4, 8, 16 (celt.cpp:1160)
OG_DEV int ordery(int stride, int i) {
    const u64 t2 = 0x01ULL, t4 = 0x1203ULL /* 3,0,2,1 */, t8 = 0x25163407ULL /* 7,0,4,3,6,1,5,2 */;
    const u64 t16 = 0x5A2D619E4B3C780FULL; /* 15,0,8,7,12,3,11,4,14,1,9,6,13,2,10,5 */
    u64 t = stride == 2 ? t2 : (stride == 4 ? t4 : (stride == 8 ? t8 : t16));
    return (int)((t >> (4 * i)) & 15);
}

OG_DEV void deinterleave_hadamard(int x, int N0, int stride, int hadamard) { // celt.cpp:1162
    int N = N0 * stride;
    OG_SYNC();
    OG_FOR_LANES(id, N) {
        int i = id / N0, j = id - i * N0;
        int dst = (hadamard ? ordery(stride, i) : i) * N0 + j;
        S.v[V_TMP + dst] = S.v[x + j * stride + i];
    }
    OG_SYNC();
    OG_FOR_LANES(id, N) S.v[x + id] = S.v[V_TMP + id];
    OG_SYNC();
}

OG_DEV void interleave_hadamard(int x, int N0, int stride, int hadamard) { // celt.cpp:1183
    int N = N0 * stride;
    OG_SYNC();
    OG_FOR_LANES(id, N) {
        int i = id / N0, j = id - i * N0;
        int src = (hadamard ? ordery(stride, i) : i) * N0 + j;
        S.v[V_TMP + j * stride + i] = S.v[x + src];
    }
    OG_SYNC();
    OG_FOR_LANES(id, N) S.v[x + id] = S.v[V_TMP + id];
    OG_SYNC();
}

OG_DEV void stereo_merge(int x, int y, i32 mid, int N) { // celt.cpp:1113
    OG_SYNC();
    i32 pxp = 0, pside = 0;
    OG_FOR_LANES(j, N) {
        pxp += mul16(S.v[y + j], S.v[x + j]);
        pside += mul16(S.v[y + j], S.v[y + j]);
    }
    i32 xp = wave_sum(pxp), side = wave_sum(pside);
    xp = mul16x32_q15(mid, xp);
    i32 mid2 = tr16(mid >> 1);
    i32 El = mul16(mid2, mid2) + side - 2 * xp;
    i32 Er = mul16(mid2, mid2) + side + 2 * xp;
    if (Er < 161061 || El < 161061) { // QCONST32(6e-4f, 28)
        OG_FOR_LANES(j, N) S.v[y + j] = S.v[x + j];
        OG_SYNC();
        return;
    }
    int kl = ilog2(El) >> 1, kr = ilog2(Er) >> 1;
    i32 lgain = rsqrt_norm(vshr32(El, (kl - 7) << 1));
    i32 rgain = rsqrt_norm(vshr32(Er, (kr - 7) << 1));
    if (kl < 7) kl = 7;
    if (kr < 7) kr = 7;
    OG_FOR_LANES(j, N) {
        i32 l = tr16(mul16_p15(mid, S.v[x + j])), r = S.v[y + j];
        S.v[x + j] = (i16)pshr32(mul16(lgain, sub16(l, r)), kl + 1);
        S.v[y + j] = (i16)pshr32(mul16(rgain, add16(l, r)), kr + 1);
    }
    OG_SYNC();
}

// ---- split angle ----------------------------------------------------------------------------------
OG_DEV int compute_qn(int N, int b, int offset, int pulse_cap, int stereo) { // celt.cpp:1215
    int N2 = 2 * N - 1;
    if (stereo && N == 2) N2--;
    int qb = (b + N2 * offset) / N2;
    qb = OG_MIN(b - pulse_cap - (4 << BITRES), qb);
    qb = OG_MIN(8 << BITRES, qb);
    if (qb < (1 << BITRES >> 1)) return 1;
    int f = qb & 7; // exp2_table8
    int e = f == 0 ? 16384 : f == 1 ? 17866 : f == 2 ? 19483 : f == 3 ? 21247 : f == 4 ? 23170 : f == 5 ? 25267
          : f == 6 ? 27554 : 30048;
    int qn = e >> (14 - (qb >> BITRES));
    return (qn + 1) >> 1 << 1;
}

struct Split { int inv, imid, iside, delta, itheta, qalloc; };

OG_DEVN void compute_theta(Rc &rc, BandCtx &cx, Split &sc, int N, i32 &b, int B, int B0, int LM, int stereo, i32 &fill) {
    int itheta = 0, inv = 0, i = cx.band;
    int pulse_cap = rom_logn[i] + LM * (1 << BITRES);
    int offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
    int qn = compute_qn(N, b, offset, pulse_cap, stereo);
    if (stereo && i >= cx.intensity) qn = 1;
    u32 tell = rc_tell_frac(rc);
    if (qn != 1) {
        if (stereo && N > 2) {
            int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0, x;
            int fs = (int)rc_decode(rc, ft);
            if (fs < (x0 + 1) * p0) x = fs / p0; else x = x0 + 1 + (fs - (x0 + 1) * p0);
            rc_update(rc, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0,
                      x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
            itheta = x;
        } else if (B0 > 1 || stereo) {
            itheta = (int)rc_uint(rc, qn + 1);
        } else {
            int fs, fl, ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fm = (int)rc_decode(rc, ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
                itheta = (int)((isqrt32(8 * (u32)fm + 1) - 1) >> 1);
                fs = itheta + 1;
                fl = itheta * (itheta + 1) >> 1;
            } else {
                itheta = (int)((2u * (u32)(qn + 1) - isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1);
                fs = qn + 1 - itheta;
                fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            }
            rc_update(rc, fl, fl + fs, ft);
        }
        itheta = (int)udiv((u32)(itheta * 16384), (u32)qn);
    } else if (stereo) {
        if (b > 2 << BITRES && cx.remaining_bits > 2 << BITRES) inv = rc_bit_logp(rc, 2);
        if (cx.disable_inv) inv = 0;
    }
    int qalloc = (int)(rc_tell_frac(rc) - tell);
    b -= qalloc;
    int imid, iside, delta;
    if (itheta == 0) {
        imid = 32767;
        iside = 0;
        fill &= (1 << B) - 1;
        delta = -16384;
    } else if (itheta == 16384) {
        imid = 0;
        iside = 32767;
        fill &= ((1 << B) - 1) << B;
        delta = 16384;
    } else {
        imid = bitexact_cos(itheta);
        iside = bitexact_cos(16384 - itheta);
        delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
    }
    sc.inv = inv;
    sc.imid = imid;
    sc.iside = iside;
    sc.delta = delta;
    sc.itheta = itheta;
    sc.qalloc = qalloc;
}

// ---- partitions -----------------------------------------------------------------------------------
// Leaf of the partition tree: PVQ pulses, or (no pulses) zero / noise / folded-spectrum fill.
// `low` < 0 means "no folding source".
OG_DEVN u32 partition_leaf(Rc &rc, BandCtx &cx, int x, int N, i32 b, int B, int low, int LM, i32 gain, i32 fill) {
    int i = cx.band;
    int q = bits2pulses(i, LM, b), curr_bits = pulses2bits(i, LM, q);
    cx.remaining_bits -= curr_bits;
    while (cx.remaining_bits < 0 && q > 0) {
        cx.remaining_bits += curr_bits;
        q--;
        curr_bits = pulses2bits(i, LM, q);
        cx.remaining_bits -= curr_bits;
    }
    if (q != 0) return pvq_unquant(rc, x, N, get_pulses(q), cx.spread, B, gain);
    u32 cm_mask = (u32)((1ull << B) - 1), cm = 0;
    fill &= (i32)cm_mask;
    OG_SYNC();
    if (!fill) {
        OG_FOR_LANES(j, N) S.v[x + j] = 0;
        OG_SYNC();
        return 0;
    }
    u32 seed = cx.seed;
    if (low < 0) { // noise
        OG_FOR_LANES(j, N) S.v[x + j] = (i16)((i32)lcg_skip(seed, (u32)j + 1) >> 20);
        cm = cm_mask;
    } else { // folded spectrum, +-1/256 dither
        OG_FOR_LANES(j, N) {
            u32 sj = lcg_skip(seed, (u32)j + 1);
            S.v[x + j] = (i16)(S.v[low + j] + ((sj & 0x8000) ? 4 : -4));
        }
        cm = (u32)fill;
    }
    cx.seed = lcg_skip(seed, (u32)N);
    renormalise(x, N, gain);
    return cm;
}

// quant_partition celt.cpp:1382.  The reference recurses (depth <= 4 splits); here the depth is a
// template parameter so the call graph is static and needs no device stack.
template <int LVL>
OG_DEVN u32 partition(Rc &rc, BandCtx &cx, int x, int N, i32 b, int B, int low, int LM, i32 gain, i32 fill) {
    if constexpr (LVL < 4) {
        const u8 *cache = pulse_cache(cx.band, LM);
        if (LM != -1 && b > cache[cache[0]] + 12 && N > 2) {
            int B0 = B;
            Split sc;
            N >>= 1;
            int y = x + N;
            LM -= 1;
            if (B == 1) fill = (fill & 1) | (fill << 1);
            B = (B + 1) >> 1;
            compute_theta(rc, cx, sc, N, b, B, B0, LM, 0, fill);
            i32 mid = sc.imid, side = sc.iside, delta = sc.delta;
            int itheta = sc.itheta;
            if (B0 > 1 && (itheta & 0x3fff)) {
                if (itheta > 8192)
                    delta -= delta >> (4 - LM);
                else
                    delta = OG_MIN(0, delta + (N << BITRES >> (5 - LM)));
            }
            i32 mbits = OG_MAX(0, OG_MIN(b, (b - delta) / 2));
            i32 sbits = b - mbits;
            cx.remaining_bits -= sc.qalloc;
            int low2 = low >= 0 ? low + N : -1;
            i32 rebalance = cx.remaining_bits;
            u32 cm;
            if (mbits >= sbits) {
                cm = partition<LVL + 1>(rc, cx, x, N, mbits, B, low, LM, tr16(mul16_p15(gain, mid)), fill);
                rebalance = mbits - (rebalance - cx.remaining_bits);
                if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
                cm |= partition<LVL + 1>(rc, cx, y, N, sbits, B, low2, LM, tr16(mul16_p15(gain, side)), fill >> B)
                      << (B0 >> 1);
            } else {
                cm = partition<LVL + 1>(rc, cx, y, N, sbits, B, low2, LM, tr16(mul16_p15(gain, side)), fill >> B)
                     << (B0 >> 1);
                rebalance = sbits - (rebalance - cx.remaining_bits);
                if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
                cm |= partition<LVL + 1>(rc, cx, x, N, mbits, B, low, LM, tr16(mul16_p15(gain, mid)), fill);
            }
            return cm;
        }
    }
    return partition_leaf(rc, cx, x, N, b, B, low, LM, gain, fill);
}

// one-coefficient bands: a sign bit each (quant_band_n1 celt.cpp:1357); y < 0 means mono
OG_DEV u32 band_n1(Rc &rc, BandCtx &cx, int x, int y, int low_out) {
    for (int c = 0; c < (y >= 0 ? 2 : 1); c++) {
        int sign = 0;
        if (cx.remaining_bits >= 1 << BITRES) {
            sign = (int)rc_bits(rc, 1);
            cx.remaining_bits -= 1 << BITRES;
        }
        S.v[c ? y : x] = (i16)(sign ? -16384 : 16384);
    }
    if (low_out >= 0) S.v[low_out] = (i16)(S.v[x] >> 4);
    return 1;
}

// quant_band celt.cpp:1526 (mono band or one side of a stereo band)
OG_DEVN u32 band_mono(Rc &rc, BandCtx &cx, int x, int N, i32 b, int B, int low, int LM, int low_out, i32 gain,
                      int low_scratch, i32 fill) {
    int N0 = N, N_B, B0 = B, time_divide = 0, recombine = 0, tf_change = cx.tf_change;
    int longBlocks = B0 == 1;
    N_B = (int)udiv((u32)N, (u32)B);
    if (N == 1) return band_n1(rc, cx, x, -1, low_out);
    if (tf_change > 0) recombine = tf_change;
    if (low_scratch >= 0 && low >= 0 && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
        OG_SYNC();
        OG_FOR_LANES(j, N) S.v[low_scratch + j] = S.v[low + j];
        OG_SYNC();
        low = low_scratch;
    }
    for (int k = 0; k < recombine; k++) {
        if (low >= 0) haar1(low, N >> k, 1 << k);
        // bit_interleave_table celt.cpp:1560 applied to both nibbles: pairs of bits are OR-ed
        int lo = fill & 0xF, hi = fill >> 4;
        int tl = (lo & 3 ? 1 : 0) | (lo & 12 ? 2 : 0), th = (hi & 3 ? 1 : 0) | (hi & 12 ? 2 : 0);
        fill = tl | th << 2;
    }
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        if (low >= 0) haar1(low, N_B, B);
        fill |= fill << B;
        B <<= 1;
        N_B >>= 1;
        time_divide++;
        tf_change++;
    }
    B0 = B;
    int N_B0 = N_B;
    if (B0 > 1 && low >= 0) deinterleave_hadamard(low, N_B >> recombine, B0 << recombine, longBlocks);
    u32 cm = partition<0>(rc, cx, x, N, b, B, low, LM, gain, fill);
    if (B0 > 1) interleave_hadamard(x, N_B >> recombine, B0 << recombine, longBlocks);
    N_B = N_B0;
    B = B0;
    for (int k = 0; k < time_divide; k++) {
        B >>= 1;
        N_B <<= 1;
        cm |= cm >> B;
        haar1(x, N_B, B);
    }
    for (int k = 0; k < recombine; k++) {
        // bit_deinterleave_table celt.cpp:1606: duplicate every bit of the low nibble
        u32 c4 = cm & 0xF;
        cm = ((c4 & 1) * 0x03) | ((c4 >> 1 & 1) * 0x0C) | ((c4 >> 2 & 1) * 0x30) | ((c4 >> 3 & 1) * 0xC0);
        haar1(x, N0 >> k, 1 << k);
    }
    B <<= recombine;
    if (low_out >= 0) {
        i32 n = tr16(celt_sqrt(shl32(N0, 22)));
        OG_SYNC();
        OG_FOR_LANES(j, N0) S.v[low_out + j] = (i16)mul16_q15(n, S.v[x + j]);
        OG_SYNC();
    }
    return cm & ((1u << B) - 1);
}

// quant_band_stereo celt.cpp:1628
OG_DEVN u32 band_stereo(Rc &rc, BandCtx &cx, int x, int y, int N, i32 b, int B, int low, int LM, int low_out,
                        int low_scratch, i32 fill) {
    if (N == 1) return band_n1(rc, cx, x, y, low_out);
    i32 orig_fill = fill;
    Split sc;
    compute_theta(rc, cx, sc, N, b, B, B, LM, 1, fill);
    i32 mid = sc.imid, side = sc.iside, delta = sc.delta;
    int itheta = sc.itheta, qalloc = sc.qalloc, inv = sc.inv;
    u32 cm;
    if (N == 2) {
        i32 mbits = b, sbits = 0;
        if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
        mbits -= sbits;
        int c = itheta > 8192;
        cx.remaining_bits -= qalloc + sbits;
        int x2 = c ? y : x, y2 = c ? x : y, sign = 0;
        if (sbits) sign = (int)rc_bits(rc, 1);
        sign = 1 - 2 * sign;
        cm = band_mono(rc, cx, x2, N, mbits, B, low, LM, low_out, 32767, low_scratch, orig_fill);
        OG_SYNC();
        // all lanes compute the same four values (uniform)
        i32 a0 = S.v[x2], a1 = S.v[x2 + 1];
        i32 b0 = tr16(-sign * a1), b1 = tr16(sign * a0);
        i32 X0 = c ? b0 : a0, X1 = c ? b1 : a1, Y0 = c ? a0 : b0, Y1 = c ? a1 : b1;
        X0 = tr16(mul16_q15(mid, X0));
        X1 = tr16(mul16_q15(mid, X1));
        Y0 = tr16(mul16_q15(side, Y0));
        Y1 = tr16(mul16_q15(side, Y1));
        OG_SYNC();
        S.v[x] = (i16)sub16(X0, Y0);
        S.v[y] = (i16)add16(X0, Y0);
        S.v[x + 1] = (i16)sub16(X1, Y1);
        S.v[y + 1] = (i16)add16(X1, Y1);
        OG_SYNC();
    } else {
        i32 mbits = OG_MAX(0, OG_MIN(b, (b - delta) / 2));
        i32 sbits = b - mbits;
        cx.remaining_bits -= qalloc;
        i32 rebalance = cx.remaining_bits;
        if (mbits >= sbits) {
            cm = band_mono(rc, cx, x, N, mbits, B, low, LM, low_out, 32767, low_scratch, fill);
            rebalance = mbits - (rebalance - cx.remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
            cm |= band_mono(rc, cx, y, N, sbits, B, -1, LM, -1, side, -1, fill >> B);
        } else {
            cm = band_mono(rc, cx, y, N, sbits, B, -1, LM, -1, side, -1, fill >> B);
            rebalance = sbits - (rebalance - cx.remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
            cm |= band_mono(rc, cx, x, N, mbits, B, low, LM, low_out, 32767, low_scratch, fill);
        }
        stereo_merge(x, y, mid, N);
    }
    if (inv) {
        OG_SYNC();
        OG_FOR_LANES(j, N) S.v[y + j] = (i16)(-S.v[y + j]);
        OG_SYNC();
    }
    return cm;
}

// quant_all_bands celt.cpp:1754.  N_ch = 120 << LM is the per-channel stride of X.
OG_DEVN void decode_all_bands(Rc &rc, int start, int end, int C, int N_ch, int shortBlocks, int spread, int dual_stereo,
                              int intensity, i32 total_bits, i32 balance, int LM, int codedBands, u32 &seed,
                              int disable_inv) {
    const int M = 1 << LM, B = shortBlocks ? M : 1;
    const int norm_offset = M * rom_eband[start];
    const int norm = V_NORM, norm2 = V_NORM + M * rom_eband[NBANDS - 1] - norm_offset;
    int low_scratch = V_X + M * rom_eband[NBANDS - 1];
    int lowband_offset = 0, update_lowband = 1;
    BandCtx cx;
    cx.intensity = intensity;
    cx.seed = seed;
    cx.spread = spread;
    cx.disable_inv = disable_inv;
    for (int i = start; i < end; i++) {
        const int last = i == end - 1;
        const int eb0 = M * rom_eband[i], N = M * rom_eband[i + 1] - eb0;
        const int x = V_X + eb0, y = C == 2 ? V_X + N_ch + eb0 : -1;
        cx.band = i;
        i32 tell = (i32)rc_tell_frac(rc);
        if (i != start) balance -= tell;
        i32 remaining_bits = total_bits - tell - 1, b;
        cx.remaining_bits = remaining_bits;
        if (i <= codedBands - 1) {
            i32 curr_balance = balance / OG_MIN(3, codedBands - i);
            b = OG_MAX(0, OG_MIN(16383, OG_MIN(remaining_bits + 1, S.pulses[i] + curr_balance)));
        } else
            b = 0;
        if ((eb0 - N >= M * rom_eband[start] || i == start + 1) && (update_lowband || lowband_offset == 0))
            lowband_offset = i;
        if (i == start + 1) { // special_hybrid_folding celt.cpp:1743
            int n1 = M * (rom_eband[start + 1] - rom_eband[start]), n2 = M * (rom_eband[start + 2] - rom_eband[start + 1]);
            if (n2 > n1) {
                OG_SYNC();
                OG_FOR_LANES(j, n2 - n1) {
                    S.v[norm + n1 + j] = S.v[norm + 2 * n1 - n2 + j];
                    if (dual_stereo) S.v[norm2 + n1 + j] = S.v[norm2 + 2 * n1 - n2 + j];
                }
                OG_SYNC();
            }
        }
        int tf_change = S.tf_res[i];
        cx.tf_change = tf_change;
        if (last) low_scratch = -1;
        int effective_lowband = -1;
        u32 x_cm, y_cm;
        if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
            effective_lowband = OG_MAX(0, M * rom_eband[lowband_offset] - norm_offset - N);
            int fold_start = lowband_offset;
            while (M * rom_eband[--fold_start] > effective_lowband + norm_offset) {}
            int fold_end = lowband_offset - 1;
            while (++fold_end < i && M * rom_eband[fold_end] < effective_lowband + norm_offset + N) {}
            x_cm = y_cm = 0;
            int fold_i = fold_start;
            do {
                x_cm |= S.cmask[fold_i * C + 0];
                y_cm |= S.cmask[fold_i * C + C - 1];
            } while (++fold_i < fold_end);
        } else
            x_cm = y_cm = (1u << B) - 1;
        if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            OG_SYNC();
            OG_FOR_LANES(j, eb0 - norm_offset) S.v[norm + j] = (i16)((S.v[norm + j] + S.v[norm2 + j]) >> 1);
            OG_SYNC();
        }
        const int low1 = effective_lowband != -1 ? norm + effective_lowband : -1;
        const int out1 = last ? -1 : norm + eb0 - norm_offset;
        if (dual_stereo) {
            const int low2 = effective_lowband != -1 ? norm2 + effective_lowband : -1;
            const int out2 = last ? -1 : norm2 + eb0 - norm_offset;
            x_cm = band_mono(rc, cx, x, N, b / 2, B, low1, LM, out1, 32767, low_scratch, (i32)x_cm);
            y_cm = band_mono(rc, cx, y, N, b / 2, B, low2, LM, out2, 32767, low_scratch, (i32)y_cm);
        } else {
            if (y >= 0)
                x_cm = band_stereo(rc, cx, x, y, N, b, B, low1, LM, out1, low_scratch, (i32)(x_cm | y_cm));
            else
                x_cm = band_mono(rc, cx, x, N, b, B, low1, LM, out1, 32767, low_scratch, (i32)(x_cm | y_cm));
            y_cm = x_cm;
        }
        S.cmask[i * C + 0] = (u8)x_cm;
        S.cmask[i * C + C - 1] = (u8)y_cm;
        balance += S.pulses[i] + tell;
        update_lowband = b > (N << BITRES);
    }
    seed = cx.seed;
}

// anti_collapse celt.cpp:1010 (transient frames only).  The noise for a collapsed short block k of a
// band goes to X[(j<<LM)+k]; the LCG is stepped once per written sample, in (band, channel, k, j) order.
OG_DEVN void anti_collapse(int LM, int C, int size, int start, int end, u32 seed) {
    for (int i = start; i < end; i++) {
        int N0 = rom_eband[i + 1] - rom_eband[i];
        int depth = (int)(udiv((u32)(1 + S.pulses[i]), (u32)N0) >> LM);
        i32 thresh32 = celt_exp2(-shl16(depth, 10 - BITRES)) >> 1;
        i32 thresh = tr16(mul16x32_q15(16384, OG_MIN(32767, thresh32)));
        i32 t = N0 << LM;
        int shift = ilog2(t) >> 1;
        t = shl32(t, (7 - shift) << 1);
        i32 sqrt_1 = rsqrt_norm(t);
        for (int c = 0; c < C; c++) {
            i32 prev1 = S.logE1[c * NBANDS + i], prev2 = S.logE2[c * NBANDS + i];
            if (C == 1) {
                prev1 = OG_MAX(prev1, (i32)S.logE1[NBANDS + i]);
                prev2 = OG_MAX(prev2, (i32)S.logE2[NBANDS + i]);
            }
            i32 Ediff = (i32)S.bandE[c * NBANDS + i] - OG_MIN(prev1, prev2);
            Ediff = OG_MAX(0, Ediff);
            i32 r;
            if (Ediff < 16384) {
                i32 r32 = celt_exp2(-tr16(Ediff)) >> 1;
                r = tr16(2 * OG_MIN(16383, r32));
            } else
                r = 0;
            if (LM == 3) r = tr16(mul16_q14(23170, OG_MIN(23169, r)));
            r = tr16(OG_MIN(thresh, r) >> 1);
            r = tr16(mul16_q15(sqrt_1, r) >> shift);
            int x = V_X + c * size + (rom_eband[i] << LM);
            int renorm = 0;
            u32 mask = S.cmask[i * C + c];
            for (int k = 0; k < 1 << LM; k++) {
                if (!(mask & (1u << k))) {
                    OG_SYNC();
                    OG_FOR_LANES(j, N0) {
                        u32 sj = lcg_skip(seed, (u32)j + 1);
                        S.v[x + (j << LM) + k] = (i16)((sj & 0x8000) ? r : -r);
                    }
                    seed = lcg_skip(seed, (u32)N0);
                    renorm = 1;
                }
            }
            if (renorm) renormalise(x, N0 << LM, 32767);
        }
    }
}

} // namespace og
