// og_celt.hpp -- CELT frame decoder, one frame per wavefront.
//
// Flow per frame (reference driver: celt_decode_with_ec, src/celt.cpp:2162-2446):
//   [scalar, wave-uniform]  header flags, coarse energy (Laplace), tf flags, dynalloc, allocation,
//                           fine energy; band loop (og_celt_bands.hpp) with lane-parallel vector ops
//   [lane-parallel]         denormalise + IMDCT pre-rotation straight from X (no `freq` array),
//                           mixed-radix FFT in LDS (all 8 short blocks side by side), post-rotation,
//                           TDAC windowing, pitch comb filter (chunked by the filter lag),
//   [one lane per channel]  de-emphasis IIR (non-linear rounding => serial), PCM staged in LDS and
//                           written out coalesced.
// HBM traffic per frame: packet in, PCM out, 60-sample IMDCT tail + <= 1024+2 samples of comb history
// read, 960 new history samples + tail + ~300 B of scalars written, per channel.
#pragma once
#include "og_celt_bands.hpp"

namespace og {

// Working arrays of one frame.  The energy / allocation code below is written once against these accessors and
// used by both execution shapes: wave-uniform (arrays in the wave's LDS record S; frame-per-wave path) and
// lane-private (one frame per LANE in the parse kernel, og_celt_parse.hpp).
struct WaveArr {
    typedef RomGlobal Rom;
    OG_MEMBER i32 &pulses(int i) const { return S.pulses_row()[i]; }
    // (the bits-per-band array WHILE compute_allocation works on it, and where a provider puts it afterwards: the same array here)
    OG_MEMBER i32 &alloc_bits(int i) const { return S.pulses_row()[i]; }
    OG_MEMBER void pulses_rest(int, int) const {}
    OG_MEMBER i32 &fine_quant(int i) const { return S.fine_quant[i]; }
    OG_MEMBER i32 &fine_prio(int i) const { return S.fine_prio[i]; }
    OG_MEMBER i32 &tf_res(int i) const { return S.tf_res[i]; }
    OG_MEMBER i32 &offsets(int i) const { return S.offsets[i]; }
    OG_MEMBER i32 &bits1(int i) const { return S.bits1[i]; }
    OG_MEMBER i32 &bits2(int i) const { return S.bits2[i]; }
    OG_MEMBER i16 &bandE(int i) const { return S.bandE_row()[i]; }
    // (where a provider's band energies and its allocation scratch share memory: celt_parse_header says when the energies rest)
    OG_MEMBER void energies_rest() const {}
    OG_MEMBER void energies_back() const {}
};

// ---- energies (src/celt.cpp:3613-3700) ------------------------------------------------------------
template <class A, class R>
OG_DEV void coarse_energy(A a, R &rc, int start, int end, int intra, int C, int LM) {
    typedef typename A::Rom T;
    const int pm = (LM * 2 + intra) * 42; // base into the e_prob_model table
    i32 prev0 = 0, prev1 = 0; // inter-band prediction per channel
    i32 coef, beta;
    if (intra) {
        coef = 0;
        beta = 4915;
    } else {
        beta = LM == 0 ? 30147 : LM == 1 ? 22282 : LM == 2 ? 12124 : 6554;
        coef = LM == 0 ? 29440 : LM == 1 ? 26112 : LM == 2 ? 21248 : 16384;
    }
    const i32 budget = (i32)rc.storage * 8;
    for (int i = start; i < end; i++) {
        for (int c = 0; c < C; c++) {
            int qi;
            i32 tell = rc_tell(rc);
            if (budget - tell >= 15) {
                int pi = 2 * OG_MIN(i, 20);
                qi = rc_laplace(rc, (u32)T::eprob(pm + pi) << 7, (int)T::eprob(pm + pi + 1) << 6);
            } else if (budget - tell >= 2) {
                // small_energy_icdf {2,1,0}, ftb 2
                u32 s = rc.rng, d = rc.val, r = s >> 2, t;
                int ret = -1;
                do {
                    t = s;
                    ++ret;
                    s = r * (u32)(2 - ret);
                } while (d < s);
                rc.val = d - s;
                rc.rng = t - s;
                rc_renorm(rc);
                qi = (ret >> 1) ^ -(ret & 1);
            } else if (budget - tell >= 1) {
                qi = -rc_bit_logp(rc, 1);
            } else
                qi = -1;
            i32 q = shl32(qi, 10);
            i32 e = OG_MAX(-9 * 1024, (i32)a.bandE(i + c * NBANDS));
            const i32 pv = c ? prev1 : prev0;
            i32 tmp = pshr32(mul16(coef, e), 8) + pv + shl32(q, 7);
            tmp = OG_MAX(-(28 << 17), tmp);
            a.bandE(i + c * NBANDS) = (i16)pshr32(tmp, 7);
            const i32 nv = pv + shl32(q, 7) - mul16(beta, pshr32(q, 8));
            if (c) prev1 = nv; else prev0 = nv;
        }
    }
}

template <class A, class R>
OG_DEV void fine_energy(A a, R &rc, int start, int end, int C) {
    for (int i = start; i < end; i++) {
        int fq = a.fine_quant(i);
        if (fq <= 0) continue;
        for (int c = 0; c < C; c++) {
            i32 q2 = (i32)rc_bits(rc, fq);
            i32 offset = tr16(sub16((shl32(q2, 10) + 512) >> fq, 512));
            a.bandE(i + c * NBANDS) = (i16)(a.bandE(i + c * NBANDS) + offset);
        }
    }
}

template <class A, class R>
OG_DEV void energy_finalise(A a, R &rc, int start, int end, int bits_left, int C) {
    for (int prio = 0; prio < 2; prio++) {
        for (int i = start; i < end && bits_left >= C; i++) {
            if (a.fine_quant(i) >= 8 || a.fine_prio(i) != prio) continue;
            for (int c = 0; c < C; c++) {
                i32 q2 = (i32)rc_bits(rc, 1);
                i32 offset = tr16((shl16(q2, 10) - 512) >> (a.fine_quant(i) + 1));
                a.bandE(i + c * NBANDS) = (i16)(a.bandE(i + c * NBANDS) + offset);
                bits_left--;
            }
        }
    }
}

// tf_select_table (celt.cpp:903) for LM, index 4*transient + 2*tf_select + flag
OG_DEV int tf_select(int LM, int idx) {
    // rows packed as 8 signed nibbles, entry k at bits 4k
    const u32 row0 = 0xF0F0F0F0u, row1 = 0xF101E0F0u, row2 = 0xF102D0E0u, row3 = 0xF103D0E0u;
    u32 row = LM == 0 ? row0 : LM == 1 ? row1 : LM == 2 ? row2 : row3;
    int v = (int)((row >> (4 * idx)) & 15);
    return v >= 8 ? v - 16 : v;
}

template <class A, class R>
OG_DEV void tf_decode(A a, R &rc, int start, int end, int transient, int LM) { // celt.cpp:2128
    int curr = 0, tf_sel = 0, tf_changed = 0;
    int logp = transient ? 2 : 4;
    u32 budget = rc.storage * 8, tell = (u32)rc_tell(rc);
    int rsv = LM > 0 && tell + logp + 1 <= budget;
    budget -= rsv;
    for (int i = start; i < end; i++) {
        if (tell + logp <= budget) {
            curr ^= rc_bit_logp(rc, logp);
            tell = (u32)rc_tell(rc);
            tf_changed |= curr;
        }
        a.tf_res(i) = curr;
        logp = transient ? 4 : 5;
    }
    if (rsv && tf_select(LM, 4 * transient + 0 + tf_changed) != tf_select(LM, 4 * transient + 2 + tf_changed))
        tf_sel = rc_bit_logp(rc, 1);
    for (int i = start; i < end; i++) a.tf_res(i) = tf_select(LM, 4 * transient + 2 * tf_sel + a.tf_res(i));
}

// init_caps celt.cpp:911: the most bits a band can use, in 1/8 bit
template <class T>
OG_DEV i32 celt_band_cap(int j, int LM, int C) {
    const int Nb = (T::eband(j + 1) - T::eband(j)) << LM;
    return (T::pulse_caps(NBANDS * (2 * LM + C - 1) + j) + 64) * C * Nb >> 2;
}

// ---- bit allocation (clt_compute_allocation celt.cpp:3523, interp_bits2pulses :3298) ----------------
template <class A, class R>
OG_DEV int compute_allocation(A a, R &rc, int start, int end, int alloc_trim, i32 &intensity, i32 &dual_stereo, i32 total,
                               i32 &balance_out, int C, int LM) {
    typedef typename A::Rom T;
    int skip_start = start, intensity_rsv = 0, dual_stereo_rsv = 0;
    total = OG_MAX(total, 0);
    int skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
    total -= skip_rsv;
    if (C == 2) {
        intensity_rsv = T::log2_frac(end - start);
        if (intensity_rsv > total)
            intensity_rsv = 0;
        else {
            total -= intensity_rsv;
            dual_stereo_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
            total -= dual_stereo_rsv;
        }
    }
    // thresh[j] and trim_offset[j] of the reference (celt.cpp:3545-3554) are a few multiplies of per-band constants:
    // computed where they are used instead of being kept in two per-band arrays (2.7 KB of the lane-per-frame parse
    // kernel's LDS per wave, which decides how many of its waves fit a CU)
    auto thresh_of = [&](int j) -> i32 {
        const int w = T::eband(j + 1) - T::eband(j);
        return OG_MAX(C << BITRES, (3 * w << LM << BITRES) >> 4);
    };
    auto trim_off_of = [&](int j) -> i32 {
        const int w = T::eband(j + 1) - T::eband(j);
        i32 to = C * w * (alloc_trim - 5 - LM) * (end - j - 1) * (1 << (LM + BITRES)) >> 6;
        if (w << LM == 1) to -= C << BITRES;
        return to;
    };
    int lo = 1, hi = 10;
    do {
        int done = 0, mid = (lo + hi) >> 1;
        i32 psum = 0;
        for (int j = end; j-- > start;) {
            int w = T::eband(j + 1) - T::eband(j);
            i32 bitsj = C * w * T::band_alloc(mid * NBANDS + j) << LM >> 2;
            if (bitsj > 0) bitsj = OG_MAX(0, bitsj + trim_off_of(j));
            bitsj += a.offsets(j);
            if (bitsj >= thresh_of(j) || done) {
                done = 1;
                psum += OG_MIN(bitsj, celt_band_cap<T>(j, LM, C));
            } else if (bitsj >= C << BITRES)
                psum += C << BITRES;
        }
        if (psum > total) hi = mid - 1; else lo = mid + 1;
    } while (lo <= hi);
    hi = lo--;
    for (int j = start; j < end; j++) {
        int w = T::eband(j + 1) - T::eband(j);
        i32 b1 = C * w * T::band_alloc(lo * NBANDS + j) << LM >> 2;
        i32 b2 = hi >= 11 ? celt_band_cap<T>(j, LM, C) : C * w * T::band_alloc(hi * NBANDS + j) << LM >> 2;
        if (b1 > 0) b1 = OG_MAX(0, b1 + trim_off_of(j));
        if (b2 > 0) b2 = OG_MAX(0, b2 + trim_off_of(j));
        if (lo > 0) b1 += a.offsets(j);
        b2 += a.offsets(j);
        if (a.offsets(j) > 0) skip_start = j;
        b2 = OG_MAX(0, b2 - b1);
        a.bits1(j) = b1;
        a.bits2(j) = b2;
    }
    // ---- interpolation between the two allocation vectors
    const int alloc_floor = C << BITRES, stereo = C > 1, logM = LM << BITRES;
    i32 psum;
    lo = 0;
    hi = 1 << 6;
    for (int i = 0; i < 6; i++) {
        int mid = (lo + hi) >> 1, done = 0;
        psum = 0;
        for (int j = end; j-- > start;) {
            i32 tmp = a.bits1(j) + (mid * a.bits2(j) >> 6);
            if (tmp >= thresh_of(j) || done) {
                done = 1;
                psum += OG_MIN(tmp, celt_band_cap<T>(j, LM, C));
            } else if (tmp >= alloc_floor)
                psum += alloc_floor;
        }
        if (psum > total) hi = mid; else lo = mid;
    }
    psum = 0;
    {
        int done = 0;
        for (int j = end; j-- > start;) {
            i32 tmp = a.bits1(j) + (lo * a.bits2(j) >> 6);
            if (tmp < thresh_of(j) && !done)
                tmp = tmp >= alloc_floor ? alloc_floor : 0;
            else
                done = 1;
            tmp = OG_MIN(tmp, celt_band_cap<T>(j, LM, C));
            a.alloc_bits(j) = tmp;
            psum += tmp;
        }
    }
    int codedBands;
    for (codedBands = end;; codedBands--) {
        int j = codedBands - 1;
        if (j <= skip_start) {
            total += skip_rsv;
            break;
        }
        i32 left = total - psum;
        i32 percoeff = (i32)udiv((u32)left, (u32)(T::eband(codedBands) - T::eband(start)));
        left -= (T::eband(codedBands) - T::eband(start)) * percoeff;
        i32 rem = OG_MAX(left - (T::eband(j) - T::eband(start)), 0);
        i32 band_width = T::eband(codedBands) - T::eband(j);
        i32 band_bits = a.alloc_bits(j) + percoeff * band_width + rem;
        if (band_bits >= OG_MAX(thresh_of(j), alloc_floor + (1 << BITRES))) {
            if (rc_bit_logp(rc, 1)) break;
            psum += 1 << BITRES;
            band_bits -= 1 << BITRES;
        }
        psum -= a.alloc_bits(j) + intensity_rsv;
        if (intensity_rsv > 0) intensity_rsv = T::log2_frac(j - start);
        psum += intensity_rsv;
        if (band_bits >= alloc_floor) {
            psum += alloc_floor;
            a.alloc_bits(j) = alloc_floor;
        } else
            a.alloc_bits(j) = 0;
    }
    if (intensity_rsv > 0)
        intensity = start + (i32)rc_uint(rc, codedBands + 1 - start);
    else
        intensity = 0;
    if (intensity <= start) {
        total += dual_stereo_rsv;
        dual_stereo_rsv = 0;
    }
    dual_stereo = dual_stereo_rsv > 0 ? rc_bit_logp(rc, 1) : 0;

    i32 left = total - psum;
    i32 percoeff = (i32)udiv((u32)left, (u32)(T::eband(codedBands) - T::eband(start)));
    left -= (T::eband(codedBands) - T::eband(start)) * percoeff;
    for (int j = start; j < codedBands; j++) a.alloc_bits(j) += percoeff * (T::eband(j + 1) - T::eband(j));
    for (int j = start; j < codedBands; j++) {
        i32 tmp = OG_MIN(left, (i32)(T::eband(j + 1) - T::eband(j)));
        a.alloc_bits(j) += tmp;
        left -= tmp;
    }
    i32 balance = 0;
    int j;
    for (j = start; j < codedBands; j++) {
        i32 N0 = T::eband(j + 1) - T::eband(j), N = N0 << LM, excess;
        i32 bit = a.alloc_bits(j) + balance, bj, ej, fp;
        if (N > 1) {
            excess = OG_MAX(bit - celt_band_cap<T>(j, LM, C), 0);
            bj = bit - excess;
            i32 den = C * N + ((C == 2 && N > 2 && !dual_stereo && j < intensity) ? 1 : 0);
            i32 NClogN = den * (T::logn(j) + logM);
            i32 offset = (NClogN >> 1) - den * 21;
            if (N == 2) offset += den << BITRES >> 2;
            if (bj + offset < den * 2 << BITRES)
                offset += NClogN >> 2;
            else if (bj + offset < den * 3 << BITRES)
                offset += NClogN >> 3;
            ej = OG_MAX(0, bj + offset + (den << (BITRES - 1)));
            ej = (i32)udiv((u32)ej, (u32)den) >> BITRES;
            if (C * ej > (bj >> BITRES)) ej = bj >> stereo >> BITRES;
            ej = OG_MIN(ej, 8);
            fp = ej * (den << BITRES) >= bj + offset;
            bj -= C * ej << BITRES;
        } else {
            excess = OG_MAX(0, bit - (C << BITRES));
            bj = bit - excess;
            ej = 0;
            fp = 1;
        }
        if (excess > 0) {
            i32 extra_fine = OG_MIN(excess >> (stereo + BITRES), 8 - ej);
            ej += extra_fine;
            i32 extra_bits = extra_fine * C << BITRES;
            fp = extra_bits >= excess - balance;
            excess -= extra_bits;
        }
        balance = excess;
        a.alloc_bits(j) = bj;
        a.fine_quant(j) = ej;
        a.fine_prio(j) = fp;
    }
    balance_out = balance;
    for (; j < end; j++) {
        i32 ej = a.alloc_bits(j) >> stereo >> BITRES;
        a.fine_quant(j) = ej;
        a.alloc_bits(j) = 0;
        a.fine_prio(j) = ej < 1;
    }
    return codedBands;
}

// ---- synthesis ------------------------------------------------------------------------------------
// Per-band linear gain for denormalise_bands (celt.cpp:948): g = 2^frac(lg), applied with a shift.
OG_DEV void denorm_gains(int start, int end, int C, int silence) {
    OG_SYNC();
    OG_FOR_LANES(id, C * NBANDS) {
        int c = id / NBANDS, i = id - c * NBANDS;
        i32 g = 0, shift = 0;
        if (!silence && i >= start && i < end) {
            i32 lg = sat16((i32)S.bandE_row()[c * NBANDS + i] + shl32((i32)rom_emeans[i], 6));
            shift = 16 - (lg >> 10);
            if (shift > 31) {
                shift = 0;
                g = 0;
            } else
                g = tr16(exp2_frac(lg & 1023));
            if (shift < 0 && shift <= -2) { // extreme gains are capped (celt.cpp:991)
                g = 16384;
                shift = -2;
            }
        }
        S.dn_g_row()[id] = (i16)g;
        S.dn_shift_row()[id] = (i16)shift;
    }
    OG_FOR_LANES(bin, 100) S.bin2band_row()[bin] = rom_bin2band[bin]; // 5 ms bin -> band (a search here was 21 dependent loads)
    OG_SYNC();
}

// denormalised MDCT coefficient j (0..N) of coded channel c
OG_DEV i32 freq_coded(int c, int j, int N, int LM) {
    int bin = j >> LM;
    if (bin >= 100) return 0;
    int band = S.bin2band_row()[bin];
    i32 g = S.dn_g_row()[c * NBANDS + band], sh = S.dn_shift_row()[c * NBANDS + band];
    i32 p = mul16(S.v[V_X + c * N + j], g);
    return sh < 0 ? shl32(p, -sh) : p >> sh;
}
// ... of OUTPUT channel co given the stream/decoder channel combination (celt_synthesis celt.cpp:2085-2119)
OG_DEV i32 freq_out(int co, int j, int N, int LM, int C, int CC) {
    if (CC == 2 && C == 1) return freq_coded(0, j, N, LM);
    if (CC == 1 && C == 2) return addw(freq_coded(0, j, N, LM) >> 1, freq_coded(1, j, N, LM) >> 1);
    return freq_coded(co, j, N, LM);
}

#define OG_SMUL(a, b) mul16x32_q15((b), (a)) /* S_MUL celt.h:192 */

struct Cpx { i32 r, i; };
OG_DEV Cpx cld(const i32 *p, int k) { Cpx v = {p[2 * k], p[2 * k + 1]}; return v; }
OG_DEV void cst(i32 *p, int k, Cpx v) { p[2 * k] = v.r; p[2 * k + 1] = v.i; }
OG_DEV Cpx cadd(Cpx a, Cpx b) { Cpx r = {addw(a.r, b.r), addw(a.i, b.i)}; return r; }
OG_DEV Cpx csub(Cpx a, Cpx b) { Cpx r = {subw(a.r, b.r), subw(a.i, b.i)}; return r; }
OG_DEV Cpx ctw(Cpx a, int tw) { // C_MUL celt.h:193 by twiddle index on the 480-point circle
    i32 wr = rom_fft_tw[2 * tw], wi = rom_fft_tw[2 * tw + 1];
    Cpx m = {subw(OG_SMUL(a.r, wr), OG_SMUL(a.i, wi)), addw(OG_SMUL(a.r, wi), OG_SMUL(a.i, wr))};
    return m;
}

// One radix-p stage over `nfft_blocks` independent transforms of `nfft` points laid out `blk_stride`
// i32 apart.  Butterfly (i, j): i in [0, Nrep), j in [0, m); element base i*mm + j.  Lanes split the
// (block, i, j) space.  Arithmetic per butterfly = kf_bfly2/3/4/5 (celt.cpp:2794-2995).
OG_DEV void fft_stage(i32 *base, int nblk, int blk_stride, int p, int m, int Nrep, int mm, int fstride) {
    const int per_blk = (p == 2) ? Nrep * 4 : Nrep * m; // radix-2 stage always has m == 4
    OG_SYNC();
    OG_FOR_LANES(id, nblk * per_blk) {
        int blk = id / per_blk, r = id - blk * per_blk;
        i32 *F = base + blk * blk_stride;
        if (p == 4) {
            int i = r / m, j = r - i * m, o = i * mm + j;
            if (m == 1) {
                Cpx f0 = cld(F, o), f1 = cld(F, o + 1), f2 = cld(F, o + 2), f3 = cld(F, o + 3);
                Cpx s0 = csub(f0, f2);
                f0 = cadd(f0, f2);
                Cpx s1 = cadd(f1, f3);
                f2 = csub(f0, s1);
                f0 = cadd(f0, s1);
                s1 = csub(f1, f3);
                f1.r = addw(s0.r, s1.i);
                f1.i = subw(s0.i, s1.r);
                f3.r = subw(s0.r, s1.i);
                f3.i = addw(s0.i, s1.r);
                cst(F, o, f0); cst(F, o + 1, f1); cst(F, o + 2, f2); cst(F, o + 3, f3);
            } else {
                Cpx f0 = cld(F, o);
                Cpx a = ctw(cld(F, o + m), j * fstride), b = ctw(cld(F, o + 2 * m), 2 * j * fstride),
                    c = ctw(cld(F, o + 3 * m), 3 * j * fstride);
                Cpx s5 = csub(f0, b);
                f0 = cadd(f0, b);
                Cpx s3 = cadd(a, c), s4 = csub(a, c);
                Cpx f2 = csub(f0, s3);
                f0 = cadd(f0, s3);
                Cpx f1 = {addw(s5.r, s4.i), subw(s5.i, s4.r)}, f3 = {subw(s5.r, s4.i), addw(s5.i, s4.r)};
                cst(F, o, f0); cst(F, o + m, f1); cst(F, o + 2 * m, f2); cst(F, o + 3 * m, f3);
            }
        } else if (p == 2) {
            // pairs (q, q+4) inside groups of 8; q selects the fixed twiddle 1, e^{-i pi/4}, -i, e^{-3i pi/4}
            int i = r >> 2, q = r & 3, o = i * 8 + q;
            Cpx a = cld(F, o), b = cld(F, o + 4), t;
            const i32 tw = 23170;
            if (q == 0)
                t = b;
            else if (q == 1) {
                t.r = OG_SMUL(addw(b.r, b.i), tw);
                t.i = OG_SMUL(subw(b.i, b.r), tw);
            } else if (q == 2) {
                t.r = b.i;
                t.i = negw(b.r);
            } else {
                t.r = OG_SMUL(subw(b.i, b.r), tw);
                t.i = OG_SMUL(negw(addw(b.i, b.r)), tw);
            }
            cst(F, o + 4, csub(a, t));
            cst(F, o, cadd(a, t));
        } else if (p == 3) {
            int i = r / m, j = r - i * m, o = i * mm + j;
            Cpx f0 = cld(F, o);
            Cpx s1 = ctw(cld(F, o + m), j * fstride), s2 = ctw(cld(F, o + 2 * m), 2 * j * fstride);
            Cpx s3 = cadd(s1, s2), s0 = csub(s1, s2);
            Cpx f1 = {subw(f0.r, s3.r >> 1), subw(f0.i, s3.i >> 1)};
            s0.r = OG_SMUL(s0.r, -28378);
            s0.i = OG_SMUL(s0.i, -28378);
            f0 = cadd(f0, s3);
            Cpx f2 = {addw(f1.r, s0.i), subw(f1.i, s0.r)};
            f1.r = subw(f1.r, s0.i);
            f1.i = addw(f1.i, s0.r);
            cst(F, o, f0); cst(F, o + m, f1); cst(F, o + 2 * m, f2);
        } else { // p == 5
            int i = r / m, u = r - i * m, o = i * mm + u;
            const i32 ya_r = 10126, ya_i = -31164, yb_r = -26510, yb_i = -19261;
            Cpx s0 = cld(F, o);
            Cpx s1 = ctw(cld(F, o + m), u * fstride), s2 = ctw(cld(F, o + 2 * m), 2 * u * fstride);
            Cpx s3 = ctw(cld(F, o + 3 * m), 3 * u * fstride), s4 = ctw(cld(F, o + 4 * m), 4 * u * fstride);
            Cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
            Cpx f0 = {addw(s0.r, addw(s7.r, s8.r)), addw(s0.i, addw(s7.i, s8.i))};
            Cpx s5 = {addw(s0.r, addw(OG_SMUL(s7.r, ya_r), OG_SMUL(s8.r, yb_r))),
                      addw(s0.i, addw(OG_SMUL(s7.i, ya_r), OG_SMUL(s8.i, yb_r)))};
            Cpx s6 = {addw(OG_SMUL(s10.i, ya_i), OG_SMUL(s9.i, yb_i)),
                      negw(addw(OG_SMUL(s10.r, ya_i), OG_SMUL(s9.r, yb_i)))};
            Cpx s11 = {addw(s0.r, addw(OG_SMUL(s7.r, yb_r), OG_SMUL(s8.r, ya_r))),
                       addw(s0.i, addw(OG_SMUL(s7.i, yb_r), OG_SMUL(s8.i, ya_r)))};
            Cpx s12 = {subw(OG_SMUL(s9.i, ya_i), OG_SMUL(s10.i, yb_i)), subw(OG_SMUL(s10.r, yb_i), OG_SMUL(s9.r, ya_i))};
            cst(F, o, f0);
            cst(F, o + m, csub(s5, s6));
            cst(F, o + 4 * m, cadd(s5, s6));
            cst(F, o + 2 * m, cadd(s11, s12));
            cst(F, o + 3 * m, csub(s11, s12));
        }
    }
    OG_SYNC();
}

// all stages of the 480- or 60-point transform (opus_fft_impl celt.cpp:2997; factor schedules :589-626)
OG_DEV void fft_blocks(i32 *base, int nblk, int blk_stride, int shift) {
    if (shift == 0) { // 480 = 5*3*4*2*4: innermost radix first
        fft_stage(base, nblk, blk_stride, 4, 1, 120, 4, 120);
        fft_stage(base, nblk, blk_stride, 2, 4, 60, 8, 60);
        fft_stage(base, nblk, blk_stride, 4, 8, 15, 32, 15);
        fft_stage(base, nblk, blk_stride, 3, 32, 5, 96, 5);
        fft_stage(base, nblk, blk_stride, 5, 96, 1, 1, 1);
    } else if (shift == 3) { // 60 = 5*3*4; twiddle stride scaled by 8
        fft_stage(base, nblk, blk_stride, 4, 1, 15, 4, 15 << 3);
        fft_stage(base, nblk, blk_stride, 3, 4, 5, 12, 5 << 3);
        fft_stage(base, nblk, blk_stride, 5, 12, 1, 1, 1 << 3);
    } else if (shift == 1) { // 240 = 5*3*4*4
        fft_stage(base, nblk, blk_stride, 4, 1, 60, 4, 60 << 1);
        fft_stage(base, nblk, blk_stride, 4, 4, 15, 16, 15 << 1);
        fft_stage(base, nblk, blk_stride, 3, 16, 5, 48, 5 << 1);
        fft_stage(base, nblk, blk_stride, 5, 48, 1, 1, 1 << 1);
    } else { // 120 = 5*3*2*4
        fft_stage(base, nblk, blk_stride, 4, 1, 30, 4, 30 << 2);
        fft_stage(base, nblk, blk_stride, 2, 4, 15, 8, 15 << 2);
        fft_stage(base, nblk, blk_stride, 3, 8, 5, 24, 5 << 2);
        fft_stage(base, nblk, blk_stride, 5, 24, 1, 1, 1 << 2);
    }
}

OG_DEV const i16 *bitrev_for(int shift) {
    return shift == 0 ? rom_bitrev480 : shift == 1 ? rom_bitrev240 : shift == 2 ? rom_bitrev120 : rom_bitrev60;
}

// The i32 synthesis buffer of the channel being synthesised (overlays norm | iy | tmp | pkt -- in the reconstruction
// kernel's 8 KB layout the second channel's spectrum and what follows X; see og_state.hpp).
OG_DEV i32 *syn_buf() { return reinterpret_cast<i32 *>(&S.v[V_SYN]); }
// Where channel co's PCM plane (960 x i16) goes inside the dead X region: the half this and later channels no longer read.
OG_DEV int pcm_plane(int co, int C, int CC) { return V_X + 960 * ((C == 1 && CC == 2) ? 1 - co : co); }

// PCM planes in LDS -> interleaved int16 PCM in HBM, two samples per lane-store, coalesced.
OG_DEV void pcm_store(i16 *pcm, int n, int C, int CC) {
    u32 *dst = reinterpret_cast<u32 *>(pcm);
    if (CC == 2) {
        const int p0 = pcm_plane(0, C, CC), p1 = pcm_plane(1, C, CC);
        OG_FOR_LANES(j, n) dst[j] = (u32)(u16)S.v[p0 + j] | (u32)(u16)S.v[p1 + j] << 16;
    } else {
        const u32 *src = reinterpret_cast<const u32 *>(&S.v[pcm_plane(0, C, CC)]);
        OG_FOR_LANES(i, n / 2) dst[i] = src[i];
    }
}

#if defined(OG_RECON_TIGHT) && !defined(OG_HOST_EMUL)
// ---- the long block (one 1920-point transform per channel: seven frames of eight on the bench payloads), restructured ----------
// clt_mdct_backward celt.cpp:3204 + opus_fft_impl :2997 for nfft = 480 = 4 x 2 x 4 x 3 x 5 (innermost radix first), every
// butterfly's arithmetic as kf_bfly4 / kf_bfly2 (:2794-2995); what changes is who computes what, and where it waits:
//   A  The digit reversal of the pre-rotation's output (rom_bitrev480) puts input i = d0 + 5 d1 + 15 d2 + 60 d3 + 120 d4 at
//      position 8 (12 d0 + 4 d1 + d2) + (4 d3 + d4): the eight inputs i0 + 60 m of one lane g = 12 d0 + 4 d1 + d2 ARE the eight
//      consecutive points its radix-4 (m = 1) and radix-2 butterflies work on.  So lane g de-normalises and pre-rotates those
//      eight (coefficients 2 i and 959 - 2 i: constant offsets from two per-lane bases, for the spectrum and for the per-bin
//      gain table alike, since 120 m is a multiple of the 8 coefficients of a bin) and runs both stages on them in registers:
//      no scatter through the permutation table, no LDS round trip between the first two stages, 60 lanes busy in ONE pass
//      (the generic code: eight passes of pre-rotation with two band look-ups per coefficient, two of radix-4, four of radix-2
//      with a four-way branch per lane).  Results go to LDS transposed ([point within the lane][lane]: conflict-free stores).
//   B  radix-4, m = 8: butterfly (i2, j) takes point j of lanes 4 i2 .. 4 i2 + 3 -- 32 contiguous bytes of the transposed
//      layout -- with twiddles that depend on j = lane & 7 only (fetched once); both passes' inputs are read before the
//      first output is stored (the stage is not in place across the two layouts), outputs in natural order.
//   C, D  radix-3 and radix-5 in place as before (fft_stage), then post-rotation, TDAC, saturation as before.
// The spectrum this reads may lie where the buffer it writes starts (og_state.hpp): every lane's reads are instructions
// before the first store in program order, and the wave executes them in order.
typedef i32 og_v2i __attribute__((ext_vector_type(2)));
typedef i32 og_v4i __attribute__((ext_vector_type(4)));
// 120 words, g (Q15) | right shift << 16 | left shift << 24 per 5 ms bin -- INSIDE the synthesis buffer, in its last 120 words (the
// overlap tail, SY[960 .. 1080)): the table is written (denorm_bins) when the buffer holds nothing -- before the first channel's
// transform, and after the second-to-last channel's output has gone to the history ring -- and a transform has read every gain,
// like every coefficient of a spectrum the buffer lies over, before it stores its first word (imdct_long_front; the short blocks'
// pre-rotation held in registers, imdct_channel).  og_state.hpp: what that saves the kernel's LDS.
OG_DEV u32 *binpar_row() { return reinterpret_cast<u32 *>(&S.v[V_SYN]) + 960; }
static_assert(960 + 120 <= SYN_LEN, "the per-bin gains lie inside the synthesis buffer");
// denormalise_bands' per-band gain (celt.cpp:948-1007) of coded channel c -- as denorm_gains left it: the band energies it read
// lie inside the synthesis buffer and are gone once the first channel's transform has run -- laid out per 5 ms bin = per group
// of 8 coefficients of a 20 ms frame; bins 100 .. 119 (coefficients the mode does not code) scale to zero
OG_DEV void denorm_bins(int c) {
    OG_SYNC();
    if (OG_LANE <= NBANDS) {
        const int i = OG_LANE;
        u32 w = 0;
        int b0 = 100, b1 = 120;
        if (i < NBANDS) {
            const i32 g = S.dn_g_row()[c * NBANDS + i], shift = S.dn_shift_row()[c * NBANDS + i];
            w = (u32)(g & 0xffff) | (u32)(shift > 0 ? shift : 0) << 16 | (u32)(shift < 0 ? -shift : 0) << 24;
            b0 = rom_eband[i];
            b1 = rom_eband[i + 1];
        }
        for (int b = b0; b < b1; b++) binpar_row()[b] = w;
    }
    OG_SYNC();
}
OG_DEV i32 denorm_coef(i32 x, u32 par) { // freq_coded with the band's parameters in hand
    const i32 p = mul16(x, (i32)(i16)(par & 0xffff));
    return shl32(p >> ((par >> 16) & 31), (int)(par >> 24));
}
struct CpxT { i32 r, i; };
OG_DEV CpxT ctw32(CpxT a, u32 w) { // C_MUL by a packed twiddle (rom_fft_tw32)
    const i32 wr = (i32)(i16)(w & 0xffff), wi = (i32)w >> 16;
    CpxT m = {subw(mul16x32_q15(wr, a.r), mul16x32_q15(wi, a.i)), addw(mul16x32_q15(wi, a.r), mul16x32_q15(wr, a.i))};
    return m;
}
// `SYF`: the transform's 480 complex points (&SY[60]); `xs`: the coded channel's spectrum; binpar_row() holds its gains
OG_DEV void imdct_long_front(i32 *SYF, const i16 *xs) {
    const int g = OG_LANE;
    CpxT v[8];
    if (g < 60) {
        const int d0 = g / 12, r12 = g - 12 * d0, i0 = d0 + 5 * (r12 >> 2) + 15 * (r12 & 3);
        const i16 *x_lo = xs + 2 * i0, *x_hi = xs + 959 - 2 * i0;
        const u32 *p_lo = binpar_row() + (i0 >> 2), *p_hi = binpar_row() + ((959 - 2 * i0) >> 3);
        const u32 *tp = rom_prerot480 + i0;
#pragma unroll
        for (int m = 0; m < 8; m++) { // input i0 + 60 m -> point 4 (m & 1) + (m >> 1) of the lane's eight
            const i32 x1 = denorm_coef(x_lo[120 * m], p_lo[15 * m]), x2 = denorm_coef(x_hi[-120 * m], p_hi[-15 * m]);
            const u32 tt = tp[60 * m];
            const i32 t0 = (i32)(i16)(tt & 0xffff), t1 = (i32)tt >> 16;
            const i32 yr = addw(mul16x32_q15(t0, x2), mul16x32_q15(t1, x1));
            const i32 yi = subw(mul16x32_q15(t0, x1), mul16x32_q15(t1, x2));
            const int k = 4 * (m & 1) + (m >> 1);
            v[k].r = yi;
            v[k].i = yr;
        }
        // radix-4, m = 1 (kf_bfly4 celt.cpp:2841 with unit twiddles), on points 0..3 and 4..7
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            CpxT f0 = v[h], f1 = v[h + 1], f2 = v[h + 2], f3 = v[h + 3];
            CpxT s0 = {subw(f0.r, f2.r), subw(f0.i, f2.i)};
            f0.r = addw(f0.r, f2.r); f0.i = addw(f0.i, f2.i);
            CpxT s1 = {addw(f1.r, f3.r), addw(f1.i, f3.i)};
            f2.r = subw(f0.r, s1.r); f2.i = subw(f0.i, s1.i);
            f0.r = addw(f0.r, s1.r); f0.i = addw(f0.i, s1.i);
            s1.r = subw(f1.r, f3.r); s1.i = subw(f1.i, f3.i);
            f1.r = addw(s0.r, s1.i); f1.i = subw(s0.i, s1.r);
            f3.r = subw(s0.r, s1.i); f3.i = addw(s0.i, s1.r);
            v[h] = f0; v[h + 1] = f1; v[h + 2] = f2; v[h + 3] = f3;
        }
        // radix-2, m = 4 (kf_bfly2 celt.cpp:2794): pairs (q, q + 4), twiddles 1, e^{-i pi/4}, -i, e^{-3i pi/4}
        const i32 tw = 23170;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const CpxT a = v[q], b = v[q + 4];
            CpxT t;
            if (q == 0)
                t = b;
            else if (q == 1) {
                t.r = mul16x32_q15(tw, addw(b.r, b.i));
                t.i = mul16x32_q15(tw, subw(b.i, b.r));
            } else if (q == 2) {
                t.r = b.i;
                t.i = negw(b.r);
            } else {
                t.r = mul16x32_q15(tw, subw(b.i, b.r));
                t.i = mul16x32_q15(tw, negw(addw(b.i, b.r)));
            }
            v[q + 4].r = subw(a.r, t.r); v[q + 4].i = subw(a.i, t.i);
            v[q].r = addw(a.r, t.r); v[q].i = addw(a.i, t.i);
        }
    }
    OG_SYNC(); // (every read of the spectrum above, every store below)
    if (g < 60) {
#pragma unroll
        for (int k = 0; k < 8; k++) *reinterpret_cast<og_v2i *>(&SYF[2 * (k * 60 + g)]) = og_v2i{v[k].r, v[k].i};
    }
    OG_SYNC();
    // B: radix-4, m = 8 (kf_bfly4 celt.cpp:2841): butterfly (i2, j), j = lane & 7, i2 = lane >> 3 and 8 + (lane >> 3)
    {
        const int j = OG_LANE & 7, ia = OG_LANE >> 3, ib = 8 + ia;
        const bool has_b = ib < 15;
        const u32 w1 = rom_fft_tw32[15 * j], w2 = rom_fft_tw32[30 * j], w3 = rom_fft_tw32[45 * j];
        CpxT in[2][4];
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int i2 = it ? (has_b ? ib : ia) : ia;
            const og_v4i lo = *reinterpret_cast<const og_v4i *>(&SYF[2 * (j * 60 + 4 * i2)]);
            const og_v4i hi = *reinterpret_cast<const og_v4i *>(&SYF[2 * (j * 60 + 4 * i2) + 4]);
            in[it][0] = CpxT{lo.x, lo.y}; in[it][1] = CpxT{lo.z, lo.w}; in[it][2] = CpxT{hi.x, hi.y}; in[it][3] = CpxT{hi.z, hi.w};
        }
        OG_SYNC();
#pragma unroll
        for (int it = 0; it < 2; it++) {
            if (it == 1 && !has_b) break;
            const int i2 = it ? ib : ia, o = 32 * i2 + j;
            CpxT f0 = in[it][0];
            const CpxT a = ctw32(in[it][1], w1), b = ctw32(in[it][2], w2), c = ctw32(in[it][3], w3);
            const CpxT s5 = {subw(f0.r, b.r), subw(f0.i, b.i)};
            f0.r = addw(f0.r, b.r); f0.i = addw(f0.i, b.i);
            const CpxT s3 = {addw(a.r, c.r), addw(a.i, c.i)}, s4 = {subw(a.r, c.r), subw(a.i, c.i)};
            const CpxT f2 = {subw(f0.r, s3.r), subw(f0.i, s3.i)};
            f0.r = addw(f0.r, s3.r); f0.i = addw(f0.i, s3.i);
            const CpxT f1 = {addw(s5.r, s4.i), subw(s5.i, s4.r)}, f3 = {subw(s5.r, s4.i), addw(s5.i, s4.r)};
            *reinterpret_cast<og_v2i *>(&SYF[2 * o]) = og_v2i{f0.r, f0.i};
            *reinterpret_cast<og_v2i *>(&SYF[2 * (o + 8)]) = og_v2i{f1.r, f1.i};
            *reinterpret_cast<og_v2i *>(&SYF[2 * (o + 16)]) = og_v2i{f2.r, f2.i};
            *reinterpret_cast<og_v2i *>(&SYF[2 * (o + 24)]) = og_v2i{f3.r, f3.i};
        }
    }
}
// The long block's post-rotation (pairs (i, 479 - i), celt.cpp:3252-3284), TDAC mirror (:3286-3296) and the saturation of the
// 960 output samples (celt_synthesis celt.cpp:2121) in two passes instead of three: a value is clamped where it is last written.
// Post-rotation pair i writes words 60 + 2 i, 61 + 2 i, 1018 - 2 i, 1019 - 2 i of SY: for i >= 30 all four are output samples
// (120 .. 959) and final; for i < 30 the first two (60 .. 119) are still the TDAC's input and the last two (960 .. 1019) are
// the overlap tail the next frame starts from -- neither is saturated (the generic code saturates SY[0 .. 960) in a pass of its
// own).  The TDAC writes words 0 .. 119: final.  Twiddles: trig[i] | trig[480 + i] packed (rom_prerot480).
OG_DEV void imdct_long_back(i32 *SY) {
    i32 *const F = &SY[OVERLAP >> 1];
    OG_SYNC();
    for (int i = OG_LANE; i < 240; i += OG_NLANES) {
        i32 *yp0 = &F[2 * i], *yp1 = &F[958 - 2 * i];
        const og_v2i a = *reinterpret_cast<const og_v2i *>(yp0), b = *reinterpret_cast<const og_v2i *>(yp1);
        const u32 ta = rom_prerot480[i], tb = rom_prerot480[479 - i];
        const i32 t0 = (i32)(i16)(ta & 0xffff), t1 = (i32)ta >> 16, u0 = (i32)(i16)(tb & 0xffff), u1 = (i32)tb >> 16;
        // (re, im) = (word 1, word 0) of a point
        i32 yr = addw(mul16x32_q15(t0, a.y), mul16x32_q15(t1, a.x)), yi = subw(mul16x32_q15(t1, a.y), mul16x32_q15(t0, a.x));
        i32 zr = addw(mul16x32_q15(u0, b.y), mul16x32_q15(u1, b.x)), zi = subw(mul16x32_q15(u1, b.y), mul16x32_q15(u0, b.x));
        if (i >= 30) {
            yr = clampsym(yr, SIG_SAT); yi = clampsym(yi, SIG_SAT); zr = clampsym(zr, SIG_SAT); zi = clampsym(zi, SIG_SAT);
        }
        *reinterpret_cast<og_v2i *>(yp0) = og_v2i{yr, zi}; // yp0[0] = yr, yp0[1] = second rotation's yi
        *reinterpret_cast<og_v2i *>(yp1) = og_v2i{zr, yi}; // yp1[0] = second rotation's yr, yp1[1] = yi
    }
    OG_SYNC();
    if (OG_LANE < OVERLAP / 2) { // TDAC mirror
        const int i = OG_LANE;
        const i32 x1 = SY[OVERLAP - 1 - i], x2 = SY[i];
        const i32 w1 = rom_win120[i], w2 = rom_win120[OVERLAP - 1 - i];
        SY[i] = clampsym(subw(mul16x32_q15(w2, x2), mul16x32_q15(w1, x1)), SIG_SAT);
        SY[OVERLAP - 1 - i] = clampsym(addw(mul16x32_q15(w1, x2), mul16x32_q15(w2, x1)), SIG_SAT);
    }
    OG_SYNC();
}
#endif

// Inverse MDCT of every block of one output channel (clt_mdct_backward celt.cpp:3204), reading
// the denormalised coefficients on the fly.  B blocks of NBk = N/B outputs, transform size 2*NBk.
OG_DEVN void imdct_channel(const i32 *tail, int co, int N, int LM, int B, int shift, int C, int CC) {
    // (the wave's values, in vector registers because the function is not inlined: scalar again, see comb_filter)
    co = OG_UNI(co); N = OG_UNI(N); LM = OG_UNI(LM); B = OG_UNI(B); shift = OG_UNI(shift); C = OG_UNI(C); CC = OG_UNI(CC);
#if defined(OG_RECON_TIGHT) && !defined(OG_HOST_EMUL)
    // this layout only sees 20 ms frames: one 1920-point transform or eight 240-point ones (the code of the other two sizes --
    // a third of the kernel's instruction bytes -- is not generated)
    if (!((N == 960 && LM == 3) && ((B == 1 && shift == 0) || (B == 8 && shift == 3)))) __builtin_unreachable();
#endif
    const int NBk = N / B, N2 = NBk, N4 = N2 >> 1;
    const i16 *trig = rom_mdct_trig + (shift == 0 ? 0 : shift == 1 ? 960 : shift == 2 ? 1440 : 1680);
    const i16 *br = bitrev_for(shift);
    i32 *const SY = syn_buf();
    {
        OG_SYNC();
#if defined(OG_RECON_TIGHT) && !defined(OG_HOST_EMUL)
        const bool long_fast = B == 1 && !(CC == 1 && C == 2); // (a down-mix reads two spectra per coefficient: generic code)
        if (long_fast) {
            const int c_src = (CC == 2 && C == 1) ? 0 : co;
            const i32 tail_l = OG_LANE < OVERLAP / 2 ? tail[OG_LANE] : 0; // (requested before the stages, stored behind them)
            imdct_long_front(&SY[OVERLAP >> 1], &S.v[V_X + c_src * N]);
            if (OG_LANE < OVERLAP / 2) SY[OG_LANE] = tail_l;
            fft_stage(&SY[OVERLAP >> 1], 1, NBk, 3, 32, 5, 96, 5);
            fft_stage(&SY[OVERLAP >> 1], 1, NBk, 5, 96, 1, 1, 1);
            imdct_long_back(SY);
            return;
        } else {
#endif
#ifdef OG_RECON_TIGHT
#if !defined(OG_HOST_EMUL)
        // (short blocks: the coefficient's gain from the per-bin table of denorm_bins -- one look-up instead of three -- unless
        // this is a down-mix, which reads two spectra with two sets of gains)
        const bool bins = !(CC == 1 && C == 2);
        const i16 *const xsrc = &S.v[V_X + ((CC == 2 && C == 1) ? 0 : co) * N];
#define OG_FREQ(j) (bins ? denorm_coef(xsrc[(j)], binpar_row()[(j) >> 3]) : freq_out(co, (j), N, LM, C, CC))
#else
#define OG_FREQ(j) freq_out(co, (j), N, LM, C, CC)
#endif
        // The buffer starts inside X, over the second channel's spectrum, and its last words hold the per-bin gains (og_state.hpp,
        // binpar_row).  A channel that reads either has every coefficient read, and rotated, before the first word of the buffer
        // is written: 480 rotations, 8 per lane, held in registers across the barrier.  (In host emulation the other channel,
        // synthesised second, reads what the buffer does not touch -- the gains come from denorm_gains' rows there.)
#if !defined(OG_HOST_EMUL)
        const bool reads_buffer = bins || (C == 2 && (co == 1 || CC == 1));
#else
        const bool reads_buffer = (C == 2 && (co == 1 || CC == 1));
#endif
        if (!reads_buffer) {
            for (int b = 0; b < B; b++) // (a loop per block: splitting one index by N4 costs a software division per element)
            OG_FOR_LANES(i, N4) {
                i32 x1 = OG_FREQ(b + B * (2 * i));
                i32 x2 = OG_FREQ(b + B * (N2 - 1 - 2 * i));
                i32 t0 = trig[i], t1 = trig[N4 + i];
                i32 yr = addw(OG_SMUL(x2, t0), OG_SMUL(x1, t1));
                i32 yi = subw(OG_SMUL(x1, t0), OG_SMUL(x2, t1));
                i32 *yp = &SY[NBk * b + (OVERLAP >> 1)];
                int rev = br[i];
                yp[2 * rev + 1] = yr;
                yp[2 * rev] = yi;
            }
            OG_FOR_LANES(i, OVERLAP / 2) SY[i] = tail[i];
        } else {
        constexpr int NR = (480 + OG_NLANES - 1) / OG_NLANES;
        i32 hr[NR], hi[NR];
        // pass `it`: a block takes ppb passes of the wave (8 for the one long block, 1 for each of the eight short blocks of a
        // transient frame: this layout only sees 20 ms frames), so block and position follow from wave-uniform values and no
        // lane has to divide its index
        const int ppb = (N4 + OG_NLANES - 1) / OG_NLANES;
#pragma unroll
        for (int it = 0; it < NR; it++) {
            const int b = ppb == 1 ? it : it / ppb, i = OG_LANE + (it - b * ppb) * OG_NLANES;
            hr[it] = hi[it] = 0;
            if (i < N4 && b < B) {
                i32 x1 = OG_FREQ(b + B * (2 * i));
                i32 x2 = OG_FREQ(b + B * (N2 - 1 - 2 * i));
                i32 t0 = trig[i], t1 = trig[N4 + i];
                hr[it] = addw(OG_SMUL(x2, t0), OG_SMUL(x1, t1));
                hi[it] = subw(OG_SMUL(x1, t0), OG_SMUL(x2, t1));
            }
        }
        OG_SYNC();
#pragma unroll
        for (int it = 0; it < NR; it++) {
            const int b = ppb == 1 ? it : it / ppb, i = OG_LANE + (it - b * ppb) * OG_NLANES;
            if (i < N4 && b < B) {
                i32 *yp = &SY[NBk * b + (OVERLAP >> 1)];
                int rev = br[i];
                yp[2 * rev + 1] = hr[it];
                yp[2 * rev] = hi[it];
            }
        }
        OG_FOR_LANES(i, OVERLAP / 2) SY[i] = tail[i];
        }
#undef OG_FREQ
#else
        OG_FOR_LANES(i, OVERLAP / 2) SY[i] = tail[i];
        for (int b = 0; b < B; b++) // pre-rotation into digit-reversed order (a loop per block: no index to divide)
        OG_FOR_LANES(i, N4) {
            i32 x1 = freq_out(co, b + B * (2 * i), N, LM, C, CC);
            i32 x2 = freq_out(co, b + B * (N2 - 1 - 2 * i), N, LM, C, CC);
            i32 t0 = trig[i], t1 = trig[N4 + i];
            i32 yr = addw(OG_SMUL(x2, t0), OG_SMUL(x1, t1));
            i32 yi = subw(OG_SMUL(x1, t0), OG_SMUL(x2, t1));
            i32 *yp = &SY[NBk * b + (OVERLAP >> 1)];
            int rev = br[i];
            yp[2 * rev + 1] = yr;
            yp[2 * rev] = yi;
        }
#endif
        fft_blocks(&SY[OVERLAP >> 1], B, NBk, shift);
#if defined(OG_RECON_TIGHT) && !defined(OG_HOST_EMUL)
        }
#endif
        for (int b = 0; b < B; b++) // post-rotation, pairs (i, N4-1-i)
        OG_FOR_LANES(i, N4 >> 1) {
            i32 *yp0 = &SY[NBk * b + (OVERLAP >> 1) + 2 * i];
            i32 *yp1 = &SY[NBk * b + (OVERLAP >> 1) + N2 - 2 - 2 * i];
            i32 re = yp0[1], im = yp0[0];
            i32 t0 = trig[i], t1 = trig[N4 + i];
            i32 yr = addw(OG_SMUL(re, t0), OG_SMUL(im, t1));
            i32 yi = subw(OG_SMUL(re, t1), OG_SMUL(im, t0));
            re = yp1[1];
            im = yp1[0];
            yp0[0] = yr;
            yp1[1] = yi;
            t0 = trig[N4 - i - 1];
            t1 = trig[N2 - i - 1];
            yr = addw(OG_SMUL(re, t0), OG_SMUL(im, t1));
            yi = subw(OG_SMUL(re, t1), OG_SMUL(im, t0));
            yp1[0] = yr;
            yp0[1] = yi;
        }
        OG_SYNC();
        OG_FOR_LANES(id, B * (OVERLAP / 2)) { // TDAC mirror
            int b = id / (OVERLAP / 2), i = id - b * (OVERLAP / 2);
            i32 *o = &SY[NBk * b];
            i32 x1 = o[OVERLAP - 1 - i], x2 = o[i];
            i32 w1 = rom_win120[i], w2 = rom_win120[OVERLAP - 1 - i];
            o[i] = subw(mul16x32_q15(w2, x2), mul16x32_q15(w1, x1));
            o[OVERLAP - 1 - i] = addw(mul16x32_q15(w1, x2), mul16x32_q15(w2, x1));
        }
        OG_SYNC();
        OG_FOR_LANES(i, N) SY[i] = clampsym(SY[i], SIG_SAT);
        OG_SYNC();
    }
}

// sample `idx` of channel c's synthesis signal: >= 0 in LDS (this frame), < 0 in the HBM ring
OG_DEV i32 syn_at(const CeltState *st, int c, int idx) {
    return idx >= 0 ? syn_buf()[idx] : st->ring[c][(st->ring_pos + idx) & RING_MASK];
}

// In-place pitch comb filter on the synthesis buffer [off .. off+N) of channel c (comb_filter celt.cpp:848).  In place the filter
// is recursive with delay >= min(T0,T1)-2 >= 13 samples, so samples are produced in chunks of that
// many, spread over the lanes; all taps of a chunk are already final.
OG_DEVN void comb_filter(const CeltState *st, int c, int off, int T0, int T1, int N, i32 g0, i32 g1, int tap0, int tap1, int ring_pos_now) {
    // (every argument is the wave's -- but a function that is not inlined receives them in vector registers, and everything derived
    // from them would be vector work: 86 vector instructions before the first sample, four calls per frame)
    c = OG_UNI(c); off = OG_UNI(off); T0 = OG_UNI(T0); T1 = OG_UNI(T1); N = OG_UNI(N); g0 = OG_UNI(g0); g1 = OG_UNI(g1);
    tap0 = OG_UNI(tap0); tap1 = OG_UNI(tap1); ring_pos_now = OG_UNI(ring_pos_now);
    if (g0 == 0 && g1 == 0) return;
    // gains[tapset][0..2] Q15 (celt.cpp:854)
    T0 = OG_MAX(T0, 15);
    T1 = OG_MAX(T1, 15);
    OG_STAT(50, 1);                                   // comb filter calls that filter
    OG_STAT(51, T1 == 15 || T0 == 15);                // ... at the shortest lag
    OG_STAT(52, T1 >= 1020 || T0 >= 1020);            // ... at (almost) the longest: 1020 .. 1022
    OG_STAT(53, T1 == 1022 || T0 == 1022);
    i32 ga0 = tap0 == 0 ? 10048 : tap0 == 1 ? 15200 : 26208, ga1 = tap0 == 0 ? 7112 : tap0 == 1 ? 8784 : 3280,
        ga2 = tap0 == 0 ? 4248 : 0;
    i32 gb0 = tap1 == 0 ? 10048 : tap1 == 1 ? 15200 : 26208, gb1 = tap1 == 0 ? 7112 : tap1 == 1 ? 8784 : 3280,
        gb2 = tap1 == 0 ? 4248 : 0;
    i32 g00 = tr16(mul16_p15(g0, ga0)), g01 = tr16(mul16_p15(g0, ga1)), g02 = tr16(mul16_p15(g0, ga2));
    i32 g10 = tr16(mul16_p15(g1, gb0)), g11 = tr16(mul16_p15(g1, gb1)), g12 = tr16(mul16_p15(g1, gb2));
    int overlap = (g0 == g1 && T0 == T1 && tap0 == tap1) ? 0 : OVERLAP;
    // Every tap that COUNTS lies >= T - 2 samples back: that many samples can be filtered without looking at each
    // other's results (no barrier in between, their history loads overlap).  A filter whose gain is zero contributes
    // exactly zero (its tap gains g?0..g?2 are zero, so is every product), so its lag does not limit the chunk and its
    // taps are not fetched -- a frame that switches the post-filter on cross-fades from "off" at the minimum lag 15, which
    // used to cut all 840 samples of the call into chunks of 13.  Past the cross-fade only T1 counts.
    const bool use0 = g0 != 0, use1 = g1 != 0;
    const int chunk_fade = OG_MIN(use0 ? T0 : 1 << 20, use1 ? T1 : 1 << 20) - 2, chunk_rest = T1 - 2;
    int end = g1 == 0 ? overlap : N; // with g1 == 0 only the cross-fade part changes the signal
#ifdef OG_HOST_EMUL
    for (int base = 0, chunk; base < end; base += chunk) {
        chunk = base < overlap ? chunk_fade : chunk_rest;
        OG_SYNC();
        OG_FOR_LANES(l, chunk) {
            const int i = base + l;
            if (i >= end) continue;
            int p = off + i;
            i32 y = syn_buf()[p];
            i32 a2 = syn_at(st, c, p - T1), a1 = syn_at(st, c, p - T1 + 1), a3 = syn_at(st, c, p - T1 - 1),
                a0 = syn_at(st, c, p - T1 + 2), a4 = syn_at(st, c, p - T1 - 2);
            if (i < overlap) {
                i32 f = tr16(mul16_q15(rom_win120[i], rom_win120[i]));
                y = y + mul16x32_q15(mul16_q15(32767 - f, g00), syn_at(st, c, p - T0)) +
                    mul16x32_q15(mul16_q15(32767 - f, g01), syn_at(st, c, p - T0 + 1) + syn_at(st, c, p - T0 - 1)) +
                    mul16x32_q15(mul16_q15(32767 - f, g02), syn_at(st, c, p - T0 + 2) + syn_at(st, c, p - T0 - 2)) +
                    mul16x32_q15(mul16_q15(f, g10), a2) + mul16x32_q15(mul16_q15(f, g11), a1 + a3) +
                    mul16x32_q15(mul16_q15(f, g12), a0 + a4);
            } else {
                y = y + mul16x32_q15(g10, a2) + mul16x32_q15(g11, a1 + a3) + mul16x32_q15(g12, a0 + a4);
            }
            syn_buf()[p] = clampsym(y, SIG_SAT);
        }
    }
#else
    // 64 consecutive samples per step: the five taps of lane l are elements e[l] .. e[l+4] of one 68-element window of the
    // signal (e[j] = x[p0 - T - 2 + j]).  Lane l fetches e[l+4] only; the other four arrive by shifting that register one
    // lane at a time (DPP wave_shr:1), with e[3] .. e[0] (fetched by lanes 0..3) fed in at lane 0.
    // The ring head does not move during the filter: read it once (a reload per tap would put an HBM round trip in front of
    // every history load).  LDS taps are plain LDS reads (index clamped), only the history taps branch to a global load --
    // never one generic pointer for both.
    const int ring_pos = ring_pos_now; // (the ring head does not move during the frame: the caller has it)
    // (explicit address spaces: two generic pointers would be folded back into one flat load)
    const __attribute__((address_space(1))) i32 *ring = (const __attribute__((address_space(1))) i32 *)st->ring[c];
    const __attribute__((address_space(3))) i32 *lds = (const __attribute__((address_space(3))) i32 *)syn_buf();
    auto tap_at = [&](int idx) -> i32 {
        const i32 in_lds = lds[OG_MAX(idx, 0)];
        i32 in_ring = 0;
        if (idx < 0) in_ring = ring[(ring_pos + idx) & RING_MASK];
        return idx < 0 ? in_ring : in_lds;
    };
    // (a step: up to 64 consecutive samples, never more than the lag allows -- ANY run of at most T - 2 samples is independent, so
    // the steps need not respect the boundaries of chunks of T - 2: 64, 64, 64 ... instead of 64, 34, 64, 34 ... at T = 100)
    for (int base = 0, lim; base < end; base += lim) {
        lim = OG_MIN(OG_MIN(base < overlap ? chunk_fade : chunk_rest, 64), end - base);
        OG_SYNC();
        {
            const int it = 0;
            const int i = base + it + OG_LANE, p = off + i;
            const bool live = it + OG_LANE < lim;
            const i32 y0 = live ? syn_buf()[p] : 0;
            auto window = [&](int T, i32 &t4, i32 &t3, i32 &t2, i32 &t1, i32 &t0) {
                // t0 = x[p-T+2] (e[l+4]) ... t4 = x[p-T-2] (e[l]).
                // Where the step's whole 68-element window lies is the same for every lane (off, base, it and T are the wave's): all of
                // it in the synthesis buffer -- five LDS reads at constant offsets from one address -- or all of it in the history
                // ring without crossing the ring's end -- five loads at constant offsets -- are the common cases and cost no vector
                // ALU work beyond one address (the kernel is bound by vector-instruction issue: the shifting scheme below spends
                // 8 instructions moving four taps across lanes and 10 more telling the two memories apart).  A dead lane's taps
                // (past the step's last live sample) may be anything: nothing is stored for it and no lane reads another's.
                const int first = off + base + it - T - 2; // index of e[0] of lane 0
                if (first >= 0) {
                    const __attribute__((address_space(3))) i32 *e = lds + (p - T - 2);
                    t4 = e[0]; t3 = e[1]; t2 = e[2]; t1 = e[3]; t0 = e[4];
                    return;
                }
                const int head = (ring_pos + first) & RING_MASK; // where e[0] of lane 0 lies in the ring
                if (first + 67 < 0 && head + 67 <= RING_MASK) {
                    const __attribute__((address_space(1))) i32 *e = ring + head + OG_LANE;
                    t4 = e[0]; t3 = e[1]; t2 = e[2]; t1 = e[3]; t0 = e[4];
                    return;
                }
                // (a window that straddles the two memories or the ring's end; indices past this step's last live sample are clamped)
                const int q = OG_MIN(p, off + base + lim - 1) - T + 2;
                t0 = tap_at(q);
                const i32 w = OG_LANE < 4 ? tap_at(off + base + it - T - 2 + OG_LANE) : 0;
                t1 = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(w, 3), t0, 0x138, 0xf, 0xf, false);
                t2 = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(w, 2), t1, 0x138, 0xf, 0xf, false);
                t3 = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(w, 1), t2, 0x138, 0xf, 0xf, false);
                t4 = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(w, 0), t3, 0x138, 0xf, 0xf, false);
            };
            i32 a4 = 0, a3 = 0, a2 = 0, a1 = 0, a0 = 0;
            if (use1) window(T1, a4, a3, a2, a1, a0);
            i32 y;
            if (base + it < overlap) { // (the cross-fade region is a whole number of steps only up to its last step)
                i32 b4 = 0, b3 = 0, b2 = 0, b1 = 0, b0 = 0;
                if (use0) window(T0, b4, b3, b2, b1, b0);
                if (i < overlap) {
                    const i32 f = tr16(mul16_q15(rom_win120[i], rom_win120[i]));
                    y = y0 + mul16x32_q15(mul16_q15(32767 - f, g00), b2) + mul16x32_q15(mul16_q15(32767 - f, g01), b1 + b3) +
                        mul16x32_q15(mul16_q15(32767 - f, g02), b0 + b4) + mul16x32_q15(mul16_q15(f, g10), a2) +
                        mul16x32_q15(mul16_q15(f, g11), a1 + a3) + mul16x32_q15(mul16_q15(f, g12), a0 + a4);
                } else
                    y = y0 + mul16x32_q15(g10, a2) + mul16x32_q15(g11, a1 + a3) + mul16x32_q15(g12, a0 + a4);
            } else
                y = y0 + mul16x32_q15(g10, a2) + mul16x32_q15(g11, a1 + a3) + mul16x32_q15(g12, a0 + a4);
            if (live) syn_buf()[p] = clampsym(y, SIG_SAT);
        }
    }
#endif
    OG_SYNC();
}

// Everything the range decoder delivers before the band loop (celt_decode_with_ec celt.cpp:2239-2330): silence flag,
// post-filter parameters, transient / intra flags, coarse energy, tf flags, spread, dynalloc boosts, allocation trim,
// bit allocation, fine energy.  The band energies in a.bandE() are updated in place.
struct CeltHeader {
    int silence, transient, intra, spread, pf_pitch, pf_tapset, anti_collapse_rsv, codedBands;
    i32 pf_gain, intensity, dual_stereo, balance;
};

template <class A, class R>
OG_DEV void celt_parse_header(A a, R &rc, int start, int end, int C, int LM, CeltHeader &h) {
    typedef typename A::Rom T;
    i32 total_bits = (i32)rc.storage * 8;
    i32 tell = rc_tell(rc);
    int silence;
    if (tell >= total_bits)
        silence = 1;
    else if (tell == 1)
        silence = rc_bit_logp(rc, 15);
    else
        silence = 0;
    if (silence) {
        tell = (i32)rc.storage * 8;
        rc.nbits_total += tell - rc_tell(rc);
    }
    int pf_pitch = 0, pf_tapset = 0;
    i32 pf_gain = 0;
    if (start == 0 && tell + 16 <= total_bits) {
        if (rc_bit_logp(rc, 1)) {
            int octave = (int)rc_uint(rc, 6);
            pf_pitch = (16 << octave) + (int)rc_bits(rc, 4 + octave) - 1;
            int qg = (int)rc_bits(rc, 3);
            if (rc_tell(rc) + 2 <= total_bits) { // tapset_icdf {2,1,0}, ftb 2
                u32 s = rc.rng, d = rc.val, r = s >> 2, t;
                int ret = -1;
                do {
                    t = s;
                    ++ret;
                    s = r * (u32)(2 - ret);
                } while (d < s);
                rc.val = d - s;
                rc.rng = t - s;
                rc_renorm(rc);
                pf_tapset = ret;
            }
            pf_gain = 3072 * (qg + 1);
        }
        tell = rc_tell(rc);
    }
    int transient = 0;
    if (LM > 0 && tell + 3 <= total_bits) {
        transient = rc_bit_logp(rc, 3);
        tell = rc_tell(rc);
    }
    const int intra = tell + 3 <= total_bits ? rc_bit_logp(rc, 3) : 0;
    OG_MARK(21);
    coarse_energy(a, rc, start, end, intra, C, LM);
    a.energies_rest(); // (nothing reads or writes a band energy between here and fine_energy)
    OG_MARK(22);
    tf_decode(a, rc, start, end, transient, LM);
    tell = rc_tell(rc);
    int spread = 2;
    if (tell + 4 <= total_bits) { // spread_icdf {25,23,2,0}, ftb 5
        u32 s = rc.rng, d = rc.val, r = s >> 5, t;
        int ret = -1;
        do {
            t = s;
            ++ret;
            s = r * (u32)(ret == 0 ? 25 : ret == 1 ? 23 : ret == 2 ? 2 : 0);
        } while (d < s);
        rc.val = d - s;
        rc.rng = t - s;
        rc_renorm(rc);
        spread = ret;
    }
    // init_caps celt.cpp:911: a band's cap is a product of per-band constants -- computed where it is used instead of being kept
    // in a per-band array (1.3 KB of the lane-per-frame parse kernel's LDS per wave: one allocation granule, see celt_band_cap)
    int dynalloc_logp = 6;
    total_bits <<= BITRES;
    tell = (i32)rc_tell_frac(rc);
    for (int i = start; i < end; i++) {
        int width = C * (T::eband(i + 1) - T::eband(i)) << LM;
        int quanta = OG_MIN(width << BITRES, OG_MAX(6 << BITRES, width));
        int loop_logp = dynalloc_logp, boost = 0;
        while (tell + (loop_logp << BITRES) < total_bits && boost < celt_band_cap<T>(i, LM, C)) {
            int flag = rc_bit_logp(rc, loop_logp);
            tell = (i32)rc_tell_frac(rc);
            if (!flag) break;
            boost += quanta;
            total_bits -= quanta;
            loop_logp = 1;
        }
        a.offsets(i) = boost;
        if (boost > 0) dynalloc_logp = OG_MAX(2, dynalloc_logp - 1);
    }
    int alloc_trim = 5;
    if (tell + (6 << BITRES) <= total_bits) { // trim_icdf, ftb 7
        u32 s = rc.rng, d = rc.val, r = s >> 7, t;
        int ret = -1;
        do {
            t = s;
            ++ret;
            // {126,124,119,109,87,41,19,9,4,2,0}
            u32 ic = ret == 0 ? 126 : ret == 1 ? 124 : ret == 2 ? 119 : ret == 3 ? 109 : ret == 4 ? 87 : ret == 5 ? 41
                   : ret == 6 ? 19 : ret == 7 ? 9 : ret == 8 ? 4 : ret == 9 ? 2 : 0;
            s = r * ic;
        } while (d < s);
        rc.val = d - s;
        rc.rng = t - s;
        rc_renorm(rc);
        alloc_trim = ret;
    }
    i32 bits = (((i32)rc.storage * 8) << BITRES) - (i32)rc_tell_frac(rc) - 1;
    const int anti_collapse_rsv = transient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
    bits -= anti_collapse_rsv;
    i32 intensity = 0, dual_stereo = 0, balance = 0;
    OG_MARK(23);
    h.codedBands = compute_allocation(a, rc, start, end, alloc_trim, intensity, dual_stereo, bits, balance, C, LM);
    a.pulses_rest(start, end); // (a provider whose allocation scratch holds the bits per band moves them out before the energies come back)
    OG_MARK(24);
    a.energies_back();
    fine_energy(a, rc, start, end, C);
    OG_MARK(25);

    h.silence = silence;
    h.transient = transient;
    h.intra = intra;
    h.spread = spread;
    h.pf_pitch = pf_pitch;
    h.pf_tapset = pf_tapset;
    h.pf_gain = pf_gain;
    h.anti_collapse_rsv = anti_collapse_rsv;
    h.intensity = intensity;
    h.dual_stereo = dual_stereo;
    h.balance = balance;
}

// ---- state helpers ----------------------------------------------------------------------------------
OG_DEV void celt_reset_state(CeltState *st) { // OPUS_RESET_STATE celt.cpp:2479 (partial on purpose, Q5)
    if (OG_LANE == 0) {
        st->rng = 0;
        st->error = 0;
        st->pf_period = st->pf_period_old = 0;
        st->pf_gain = st->pf_gain_old = 0;
        st->pf_tapset = st->pf_tapset_old = 0;
    }
    OG_FOR_LANES(i, 2 * NBANDS) st->logE1[i] = st->logE2[i] = (i16)(-28 * 1024);
}

// Synthesis half of a CELT frame (celt_synthesis celt.cpp:2057 onwards, then :2372-2440): from the normalised bands in
// S.v[V_X..] and the final band energies in S.bandE to PCM staged in S.v[V_X..] (interleaved, CC channels), plus the
// write-back of the stream's history.  Everything it needs from the entropy-decoding half is in CeltSynth, so the
// same code serves the single-kernel path and the reconstruction kernel of the split path.
struct CeltSynth {
    int N, LM, C, CC, start, end, silence, transient, pf_pitch, pf_tapset;
    i32 pf_gain;
    u32 rng_final;
    int rc_error;
    int inline_deemph; // 1: de-emphasis + PCM planes here (single-kernel path); 0: left to celt_post_lane (split path)
    LossState *loss;   // RFC mode: the noise floor follows the decoded energies and the loss counter restarts (celt.cpp:2411-2440)
    int lost;          // a concealed frame (celt_decode_lost): the energy histories, the post-filter and its state stay as they are
    int energies_kept_by_parse = 0; // split path: CeltState::bandE is written by the parse kernel (celt_parse_lane), not here
    // the stream's post-filter state and ring head, when the caller has them in registers already (the reconstruction kernel
    // fetches every scalar it needs of the stream in one batch at its start, og_celt_split.hpp ReconHdr); else read here
    int have_state = 0;
    int st_pf_period, st_pf_period_old, st_pf_tapset, st_pf_tapset_old, st_ring_pos;
    i32 st_pf_gain, st_pf_gain_old;
    // the energy histories the frame starts from are in S.logE1_row() / S.logE2_row() (all callers stage them there)
};

// De-emphasis and float-to-int16 of one channel of one frame, lane-private (celt.cpp:1965-2055, sig2word16 celt.h:413):
// the split path's third kernel runs this with one (frame, channel) per lane, reading the comb-filtered samples the
// reconstruction kernel appended to the stream's history ring.  `pcm` may be null (state update only).
#ifndef OG_HOST_EMUL
typedef i32 og_v4i __attribute__((ext_vector_type(4)));
typedef u32 og_v2u __attribute__((ext_vector_type(2)));
#endif
// four consecutive samples of one channel: the recurrence, then int16 PCM
// `silk`: optional second signal (interleaved like the PCM) added with saturation to the entries below `silk_n`.
OG_DEV i32 post_mix(i32 v, const i16 *silk, int silk_n, int at) {
    return (silk && at < silk_n) ? sat16(v + (i32)silk[at]) : v;
}
#ifndef OG_HOST_EMUL
// the four addends of samples j..j+3 of channel c with one vector load (all four entries lie on the same side of
// silk_n: it is a multiple of 4 * CC); zeros when there is nothing to add
struct PostAdd { i32 a0, a1, a2, a3; };
OG_DEV PostAdd post_addends(const i16 *silk, int silk_n, int j, int c, int CC) {
    PostAdd r = {0, 0, 0, 0};
    if (silk && (j + 3) * CC + c < silk_n) {
        if (CC == 2) {
            const og_v4i v = *reinterpret_cast<const og_v4i *>(silk + 2 * j); // L0 R0 L1 R1 | L2 R2 L3 R3, 16-byte aligned
            const int sh = 16 * c;
            r.a0 = (i32)(i16)((u32)v.x >> sh); r.a1 = (i32)(i16)((u32)v.y >> sh);
            r.a2 = (i32)(i16)((u32)v.z >> sh); r.a3 = (i32)(i16)((u32)v.w >> sh);
        } else {
            const og_v2u v = *reinterpret_cast<const og_v2u *>(silk + j);
            r.a0 = (i32)(i16)v.x; r.a1 = (i32)(i16)(v.x >> 16); r.a2 = (i32)(i16)v.y; r.a3 = (i32)(i16)(v.y >> 16);
        }
    }
    return r;
}
#endif
OG_DEV void celt_post4(i32 &m, i32 s0, i32 s1, i32 s2, i32 s3, int j, int c, int CC, i16 *__restrict__ pcm,
                      const i16 *__restrict__ silk, int silk_n) {
    const i32 t0 = s0 + m;
    m = mul16x32_q15(27853, t0);
    const i32 t1 = s1 + m;
    m = mul16x32_q15(27853, t1);
    const i32 t2 = s2 + m;
    m = mul16x32_q15(27853, t2);
    const i32 t3 = s3 + m;
    m = mul16x32_q15(27853, t3);
#ifdef OG_HOST_EMUL
    const i32 o0 = post_mix(sat16(pshr32(t0, 12)), silk, silk_n, (j + 0) * CC + c), o1 = post_mix(sat16(pshr32(t1, 12)), silk, silk_n, (j + 1) * CC + c);
    const i32 o2 = post_mix(sat16(pshr32(t2, 12)), silk, silk_n, (j + 2) * CC + c), o3 = post_mix(sat16(pshr32(t3, 12)), silk, silk_n, (j + 3) * CC + c);
#else
    const PostAdd ad = post_addends(silk, silk_n, j, c, CC);
    const i32 o0 = sat16(sat16(pshr32(t0, 12)) + ad.a0), o1 = sat16(sat16(pshr32(t1, 12)) + ad.a1);
    const i32 o2 = sat16(sat16(pshr32(t2, 12)) + ad.a2), o3 = sat16(sat16(pshr32(t3, 12)) + ad.a3);
#endif
#ifdef OG_HOST_EMUL
    if (pcm) {
        pcm[(j + 0) * CC + c] = (i16)o0;
        pcm[(j + 1) * CC + c] = (i16)o1;
        pcm[(j + 2) * CC + c] = (i16)o2;
        pcm[(j + 3) * CC + c] = (i16)o3;
    }
#else
    // two packed words; with two channels the lanes of a (left, right) pair swap halves so that each of them writes 8
    // contiguous bytes of the interleaved PCM
    const u32 w01 = (u32)(u16)o0 | (u32)(u16)o1 << 16;
    const u32 w23 = (u32)(u16)o2 | (u32)(u16)o3 << 16;
    og_v2u out;
    int at;
    if (CC == 2) {
        const u32 p01 = (u32)__shfl_xor((int)w01, 1, 64), p23 = (u32)__shfl_xor((int)w23, 1, 64);
        if (c == 0) { // samples j, j+1: L0 R0 L1 R1
            out.x = (w01 & 0xffffu) | p01 << 16;
            out.y = w01 >> 16 | (p01 & 0xffff0000u);
        } else { // samples j+2, j+3: L2 R2 L3 R3
            out.x = (p23 & 0xffffu) | w23 << 16;
            out.y = p23 >> 16 | (w23 & 0xffff0000u);
        }
        at = (j + 2 * c) * 2;
    } else {
        out.x = w01;
        out.y = w23;
        at = j;
    }
    if (pcm) *reinterpret_cast<og_v2u *>(pcm + at) = out;
#endif
}

// `pos`: ring index of the frame's first sample (a multiple of 8, as N is)
OG_DEV void celt_post_lane(CeltState *st, int c, int CC, int N, int pos, i16 *__restrict__ pcm, const i16 *__restrict__ silk, int silk_n) {
    const i32 *ring = st->ring[c];
    i32 m = st->deemph[c];
#ifdef OG_HOST_EMUL
    for (int j = 0; j < N; j += 4) {
        const i32 *src = &ring[(pos + j) & RING_MASK];
        celt_post4(m, src[0], src[1], src[2], src[3], j, c, CC, pcm, silk, silk_n);
    }
#else
    // 16 samples per iteration; the next 16 are requested before the current ones are consumed (the loads do not
    // depend on the recurrence, and nothing else hides their latency).  Groups of 4 are 16-byte aligned, never wrap.
#define OG_LD4(j) (*reinterpret_cast<const og_v4i *>(&ring[(pos + (j)) & RING_MASK]))
    og_v4i a0 = OG_LD4(0), a1 = OG_LD4(4), a2 = OG_LD4(8), a3 = OG_LD4(12);
    for (int j = 0; j < N; j += 16) {
        const int jn = j + 16 < N ? j + 16 : j;
        const og_v4i b0 = OG_LD4(jn), b1 = OG_LD4(jn + 4), b2 = OG_LD4(jn + 8), b3 = OG_LD4(jn + 12);
        celt_post4(m, a0.x, a0.y, a0.z, a0.w, j, c, CC, pcm, silk, silk_n);
        celt_post4(m, a1.x, a1.y, a1.z, a1.w, j + 4, c, CC, pcm, silk, silk_n);
        celt_post4(m, a2.x, a2.y, a2.z, a2.w, j + 8, c, CC, pcm, silk, silk_n);
        celt_post4(m, a3.x, a3.y, a3.z, a3.w, j + 12, c, CC, pcm, silk, silk_n);
        a0 = b0;
        a1 = b1;
        a2 = b2;
        a3 = b3;
    }
#undef OG_LD4
#endif
    st->deemph[c] = m;
}


OG_DEV void celt_synthesis(CeltState *st, const CeltSynth &p) {
    const int N = p.N, LM = p.LM, C = p.C, CC = p.CC, start = p.start, end = p.end, silence = p.silence, transient = p.transient;
    const int pf_pitch = p.pf_pitch, pf_tapset = p.pf_tapset, M = 1 << LM;
    const i32 pf_gain = p.pf_gain;
    denorm_gains(start, end, C, silence);
    // ---- energy history (celt.cpp:2404-2436)
    OG_SYNC();
    if (C == 1) {
        OG_FOR_LANES(i, NBANDS) S.bandE_row()[NBANDS + i] = S.bandE_row()[i];
        OG_SYNC();
    }
    if (!p.lost) OG_FOR_LANES(i, 2 * NBANDS) {
        int band = i >= NBANDS ? i - NBANDS : i;
        i32 e = S.bandE_row()[i], l1 = S.logE1_row()[i], l2 = S.logE2_row()[i];
        if (!transient) {
            l2 = l1;
            l1 = e;
        } else
            l1 = OG_MIN(l1, e);
        if (p.loss && !transient) { // the noise floor rises by at most 2.4 dB/s; 1 dB per update after a long loss
            const i32 inc = p.loss->celt_loss_count < 10 ? M : 1024; // M * QCONST16(0.001f, DB_SHIFT); QCONST16(1.f, DB_SHIFT)
            p.loss->backgroundLogE[i] = (i16)OG_MIN((i32)(i16)(p.loss->backgroundLogE[i] + inc), e);
        }
        if (band < start || band >= end) {
            e = 0;
            l1 = l2 = -28 * 1024;
        }
        if (!p.energies_kept_by_parse) st->bandE[i] = (i16)e;
        st->logE1[i] = (i16)l1;
        st->logE2[i] = (i16)l2;
    }
    const int B = transient ? M : 1, shift = transient ? 3 : 3 - LM;
    const int pp = OG_MAX(p.have_state ? p.st_pf_period : st->pf_period, 15), ppo = OG_MAX(p.have_state ? p.st_pf_period_old : st->pf_period_old, 15);
    const i32 pg = p.have_state ? p.st_pf_gain : st->pf_gain, pgo = p.have_state ? p.st_pf_gain_old : st->pf_gain_old;
    const int pt = p.have_state ? p.st_pf_tapset : st->pf_tapset, pto = p.have_state ? p.st_pf_tapset_old : st->pf_tapset_old;
    const int pos = p.have_state ? p.st_ring_pos : st->ring_pos;
    i32 *const SY = syn_buf();
    // ---- one output channel at a time through the single synthesis buffer
    // (the 8 KB layout synthesises the second channel first: its spectrum lies where the buffer starts)
#ifdef OG_RECON_TIGHT
    for (int c = CC - 1; c >= 0; c--) {
#else
    for (int c = 0; c < CC; c++) {
#endif
        OG_SYNC();
        OG_MARK(14);
#if defined(OG_RECON_TIGHT) && !defined(OG_HOST_EMUL)
        if (!(CC == 1 && C == 2)) denorm_bins((CC == 2 && C == 1) ? 0 : c); // (from denorm_gains' rows)
#endif
        imdct_channel(st->tail[c], c, N, LM, B, shift, C, CC);
        OG_MARK(15);
        OG_TAP(2 + 16 * c); // IMDCT output
        if (!p.lost) {
            comb_filter(st, c, 0, ppo, pp, 120, pgo, pg, pto, pt, pos);
            if (LM != 0) comb_filter(st, c, 120, pp, pf_pitch, N - 120, pg, pf_gain, pt, pf_tapset, pos);
        }
        OG_TAP(3 + 16 * c); // comb filter output
        // de-emphasis (celt.cpp:1965-2055): a rounding IIR, serial; the PCM plane replaces a dead half of X
        OG_SYNC();
        if (p.inline_deemph && OG_LANE == 0) {
            const int plane = pcm_plane(c, C, CC);
            i32 m = st->deemph[c];
            for (int j = 0; j < N; j++) {
                i32 tmp = SY[j] + m;
                m = mul16x32_q15(27853, tmp);
                S.v[plane + j] = (i16)sat16(pshr32(tmp, 12)); // sig2word16 celt.h:413
            }
            st->deemph[c] = m;
        }
        OG_SYNC();
        // history ring and overlap tail
        OG_MARK(16);
        OG_FOR_LANES(i, N) st->ring[c][(pos + i) & RING_MASK] = SY[i];
        OG_FOR_LANES(i, OVERLAP / 2) st->tail[c][i] = SY[N + i];
    }
    OG_SYNC();
    if (OG_LANE == 0 && p.lost) {
        st->ring_pos = (pos + N) & RING_MASK;
        st->rng = p.rng_final;
        p.loss->celt_loss_count += 1;
    } else if (OG_LANE == 0) {
        if (p.loss) p.loss->celt_loss_count = 0;
        st->ring_pos = (pos + N) & RING_MASK;
        st->rng = p.rng_final;
        int new_old_period = pp, new_old_tapset = pt;
        i32 new_old_gain = pg;
        if (LM != 0) {
            new_old_period = pf_pitch;
            new_old_gain = pf_gain;
            new_old_tapset = pf_tapset;
        }
        st->pf_period_old = new_old_period;
        st->pf_gain_old = new_old_gain;
        st->pf_tapset_old = new_old_tapset;
        st->pf_period = pf_pitch;
        st->pf_gain = pf_gain;
        st->pf_tapset = pf_tapset;
        if (p.rc_error) st->error = 1;
    }
}

// Decode one CELT frame of `frame_size` samples (120 << LM) from the live range decoder.
// pcm_out: LDS i16 buffer (interleaved, CC channels) -- S.v[V_X..] is reused for it after synthesis.
// Returns frame_size or a negative code (wave-uniform).
// `end`: one past the last band decoded -- 21 always in reference mode (Q1), by bandwidth in RFC mode
OG_DEV int celt_decode_frame(CeltState *st, Rc &rc, int frame_size, int C, int CC, int start, int disable_inv, int end = NBANDS,
                             LossState *loss = nullptr) {
    const i32 *eb = rom_eband;
    int LM;
    for (LM = 0; LM <= 3; LM++)
        if (120 << LM == frame_size) break;
    if (LM > 3) return CELT_BAD_ARG; // celt.cpp:2211
    const int M = 1 << LM, N = M * 120;
    if (rc.storage > 1275 || rc.storage <= 1) return CELT_BAD_ARG; // :2216, :2225

    // ---- stage persistent scalars in LDS
    OG_SYNC();
    OG_FOR_LANES(i, 2 * NBANDS) {
        S.bandE_row()[i] = st->bandE[i];
        S.logE1_row()[i] = st->logE1[i];
        S.logE2_row()[i] = st->logE2[i];
    }
    OG_FOR_LANES(i, 2 * NBANDS) S.cmask_row()[i] = 0;
    OG_FOR_LANES(i, NBANDS) {
        S.pulses_row()[i] = 0;
        S.fine_quant[i] = 0;
        S.fine_prio[i] = 0;
        S.offsets[i] = 0;
    }
    OG_FOR_LANES(i, 2 * N) S.v[V_X + i] = 0;
    OG_FOR_LANES(i, 1248) S.v[V_NORM + i] = 0;
    OG_SYNC();
    if (C == 1) {
        OG_FOR_LANES(i, NBANDS) S.bandE_row()[i] = OG_MAX(S.bandE_row()[i], S.bandE_row()[NBANDS + i]);
        OG_SYNC();
    }

    // ---- header, energies, allocation (all range-decoder work before the band loop)
    CeltHeader h;
    celt_parse_header(WaveArr(), rc, start, end, C, LM, h);
    const int silence = h.silence, transient = h.transient, spread = h.spread, pf_pitch = h.pf_pitch, pf_tapset = h.pf_tapset;
    const i32 pf_gain = h.pf_gain, intensity = h.intensity, dual_stereo = h.dual_stereo, balance = h.balance;
    OG_MARK(26); // (single-kernel path: the band loop, quant_all_bands)
    const int shortBlocks = transient ? M : 0, anti_collapse_rsv = h.anti_collapse_rsv, codedBands = h.codedBands;
    u32 seed = st->rng;
    decode_all_bands(rc, start, end, C, N, shortBlocks, spread, dual_stereo, intensity,
                     (i32)rc.storage * (8 << BITRES) - anti_collapse_rsv, balance, LM, codedBands, seed, disable_inv);
    OG_MARK(27);
    int anti_collapse_on = 0;
    if (anti_collapse_rsv > 0) anti_collapse_on = (int)rc_bits(rc, 1);
    energy_finalise(WaveArr(), rc, start, end, (i32)rc.storage * 8 - rc_tell(rc), C);
    if (anti_collapse_on) anti_collapse(LM, C, N, start, end, seed);
    if (silence)
        for (int i = 0; i < C * NBANDS; i++) S.bandE_row()[i] = (i16)(-28 * 1024);

    OG_TAP(1); // X and bandE final
    CeltSynth sp;
    sp.N = N; sp.LM = LM; sp.C = C; sp.CC = CC; sp.start = start; sp.end = end; sp.silence = silence; sp.transient = transient;
    sp.pf_pitch = pf_pitch; sp.pf_tapset = pf_tapset; sp.pf_gain = pf_gain; sp.rng_final = rc.rng; sp.rc_error = rc.error; sp.inline_deemph = 1;
    sp.loss = loss; sp.lost = 0;
    celt_synthesis(st, sp);
    if (rc_tell(rc) > 8 * (i32)rc.storage) return INTERNAL_ERROR;
    return frame_size;
}

// Concealment of a lost CELT frame (RFC mode only; the reference has none, Q8): the noise-based branch of RFC 6716's
// celt_decode_lost, as the oracle restates it (oc_celt_decode_lost).  The band energies decay towards the noise floor
// (1.5 dB for the first lost frame, 0.5 dB after), every band from `start` to `end` of every DECODER channel is filled with LCG
// noise and renormalised, then one long MDCT, no post-filter, de-emphasis.  (The pitch-based branch, taken for the first lost frames
// of a CELT-only stream, is og_plc.hpp; celt_conceal there chooses.)  PCM planes as after celt_decode_frame with C == CC.
OG_DEV int celt_decode_lost(CeltState *st, LossState *loss, int frame_size, int CC, int start, int end) {
    const i32 *eb = rom_eband;
    int LM;
    for (LM = 0; LM <= 3; LM++)
        if (120 << LM == frame_size) break;
    if (LM > 3) return BAD_ARG;
    const int N = 120 << LM, effEnd = OG_MAX(start, OG_MIN(end, NBANDS));
    const i32 decay = loss->celt_loss_count == 0 ? 1536 : 512; // QCONST16(1.5f, DB_SHIFT) : QCONST16(.5f, DB_SHIFT)
    OG_SYNC();
    OG_FOR_LANES(i, 2 * NBANDS) {
        const int c = i >= NBANDS, band = i - c * NBANDS;
        i32 e = st->bandE[i];
        if (c < CC && band >= start && band < end) {
            e = OG_MAX((i32)loss->backgroundLogE[i], (i32)(i16)(e - decay));
            st->bandE[i] = (i16)e;
        }
        S.bandE_row()[i] = (i16)e;
    }
    OG_FOR_LANES(i, 2 * N) S.v[V_X + i] = 0;
    OG_SYNC();
    // noise: the LCG runs once per coefficient in (channel, band, bin) order; the bins of a channel's bands are contiguous
    u32 seed = st->rng;
    const int lo = eb[start] << LM, width = (eb[effEnd] << LM) - lo;
    for (int c = 0; c < CC; c++) {
        OG_FOR_LANES(j, width) S.v[V_X + c * N + lo + j] = (i16)((i32)lcg_skip(seed, (u32)j + 1) >> 20);
        seed = lcg_skip(seed, (u32)width);
        for (int i = start; i < effEnd; i++) renormalise(V_X + c * N + (eb[i] << LM), (eb[i + 1] - eb[i]) << LM, 32767);
    }
    CeltSynth sp;
    sp.N = N; sp.LM = LM; sp.C = CC; sp.CC = CC; sp.start = start; sp.end = effEnd; sp.silence = 0; sp.transient = 0;
    sp.pf_pitch = 0; sp.pf_tapset = 0; sp.pf_gain = 0; sp.rng_final = seed; sp.rc_error = 0; sp.inline_deemph = 1;
    sp.loss = loss; sp.lost = 1;
    celt_synthesis(st, sp);
    return frame_size;
}

} // namespace og

#undef OG_SYNC
#define OG_SYNC() OG_FULL_SYNC()
