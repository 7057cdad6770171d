/*
 * oc_celt.c -- CPU ORACLE (test infrastructure): the fixed-point CELT frame decoder.
 * Restates the reference's src/celt.cpp: energy decoding (:3613-3700), time/frequency flags
 * (:2128), bit allocation (:3298-3611), band decoding (:745-815, :1113-1924), anti-collapse
 * (:1010), synthesis (:948, :2057), de-emphasis (:1965-2055) and the frame driver (:2162-2446),
 * with all state in oc_celt / oc_rc instead of globals.
 */
#include "oc_celt_priv.h"

#define NB OC_NBANDS
static const i16 pred_coef[4] = {29440, 26112, 21248, 16384};  /* celt.cpp:541 */
static const i16 beta_coef[4] = {30147, 22282, 12124, 6554};   /* celt.cpp:542 */
static const i16 beta_intra = 4915;                            /* celt.cpp:543 */
static const u8 small_energy_icdf[3] = {2, 1, 0};
static const u8 trim_icdf[11] = {126, 124, 119, 109, 87, 41, 19, 9, 4, 2, 0};
static const u8 spread_icdf[4] = {25, 23, 2, 0};
static const u8 tapset_icdf[3] = {2, 1, 0};
static const signed char tf_select_table[4][8] = { /* celt.cpp:903 */
    {0, -1, 0, -1, 0, -1, 0, -1},
    {0, -1, 0, -2, 1, 0, 1, -1},
    {0, -2, 0, -3, 2, 0, 1, -1},
    {0, -2, 0, -3, 3, 0, 1, -1},
};

/* ---- state ------------------------------------------------------------------------------- */
void oc_celt_reset(oc_celt *st) { /* celt.cpp:2479: partial reset (Q5) */
    int i;
    st->rng = 0;
    st->error = 0;
    st->pf_period = st->pf_period_old = 0;
    st->pf_gain = st->pf_gain_old = 0;
    st->pf_tapset = st->pf_tapset_old = 0;
    for (i = 0; i < 2 * NB; i++) st->logE1[i] = st->logE2[i] = -28 * 1024;
}

void oc_celt_init(oc_celt *st, int channels) { /* celt.cpp:1933: clear everything, then reset */
    memset(st, 0, sizeof(*st));
    st->channels = channels;
    st->stream_channels = channels;
    st->disable_inv = channels == 1;
    st->end_band = NB;
    oc_celt_reset(st);
}

/* ---- band energies ------------------------------------------------------------------------ */
/* celt.cpp:3613 */
static void coarse_energy(oc_rc *rc, int start, int end, i16 *E, int intra, int C, int LM) {
    const u8 *pm = rom_eprob + (LM * 2 + intra) * 42;
    i32 prev[2] = {0, 0};
    i16 coef, beta;
    i32 budget = rc->storage * 8;
    int i, c;
    if (intra) {
        coef = 0;
        beta = beta_intra;
    } else {
        beta = beta_coef[LM];
        coef = pred_coef[LM];
    }
    for (i = start; i < end; i++) {
        c = 0;
        do {
            int qi;
            i32 q, tmp, tell = oc_rc_tell(rc);
            if (budget - tell >= 15) {
                int pi = 2 * OC_MIN(i, 20);
                qi = oc_rc_laplace(rc, pm[pi] << 7, pm[pi + 1] << 6);
            } else if (budget - tell >= 2) {
                qi = oc_rc_icdf(rc, small_energy_icdf, 2);
                qi = (qi >> 1) ^ -(qi & 1);
            } else if (budget - tell >= 1) {
                qi = -oc_rc_bit_logp(rc, 1);
            } else
                qi = -1;
            q = shl32(qi, 10);
            E[i + c * NB] = (i16)OC_MAX(-9 * 1024, (i32)E[i + c * NB]);
            tmp = pshr32(m16(coef, E[i + c * NB]), 8) + prev[c] + shl32(q, 7);
            tmp = OC_MAX(-(28 << 17), tmp);
            E[i + c * NB] = (i16)pshr32(tmp, 7);
            prev[c] = prev[c] + shl32(q, 7) - m16(beta, pshr32(q, 8));
        } while (++c < C);
    }
}

/* celt.cpp:3664 */
static void fine_energy(oc_rc *rc, int start, int end, i16 *E, const i32 *fine_quant, int C) {
    int i, c;
    for (i = start; i < end; i++) {
        if (fine_quant[i] <= 0) continue;
        c = 0;
        do {
            i32 q2 = oc_rc_bits(rc, fine_quant[i]);
            i16 offset = (i16)sub16((shl32(q2, 10) + 512) >> fine_quant[i], 512);
            E[i + c * NB] += offset;
        } while (++c < C);
    }
}

/* celt.cpp:3681 */
static void energy_finalise(oc_rc *rc, int start, int end, i16 *E, const i32 *fine_quant,
                            const i32 *fine_priority, int bits_left, int C) {
    int i, prio, c;
    for (prio = 0; prio < 2; prio++) {
        for (i = start; i < end && bits_left >= C; i++) {
            if (fine_quant[i] >= 8 || fine_priority[i] != prio) continue;
            c = 0;
            do {
                i32 q2 = oc_rc_bits(rc, 1);
                i16 offset = (i16)((shl16(q2, 10) - 512) >> (fine_quant[i] + 1));
                E[i + c * NB] += offset;
                bits_left--;
            } while (++c < C);
        }
    }
}

/* celt.cpp:2128 */
static void tf_decode(oc_rc *rc, int start, int end, int transient, i32 *tf_res, int LM) {
    int i, curr = 0, tf_select = 0, tf_changed = 0, tf_select_rsv;
    int logp = transient ? 2 : 4;
    u32 budget = rc->storage * 8, tell = oc_rc_tell(rc);
    tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
    budget -= tf_select_rsv;
    for (i = start; i < end; i++) {
        if (tell + logp <= budget) {
            curr ^= oc_rc_bit_logp(rc, logp);
            tell = oc_rc_tell(rc);
            tf_changed |= curr;
        }
        tf_res[i] = curr;
        logp = transient ? 4 : 5;
    }
    if (tf_select_rsv &&
        tf_select_table[LM][4 * transient + 0 + tf_changed] != tf_select_table[LM][4 * transient + 2 + tf_changed])
        tf_select = oc_rc_bit_logp(rc, 1);
    for (i = start; i < end; i++) tf_res[i] = tf_select_table[LM][4 * transient + 2 * tf_select + tf_res[i]];
}

/* ---- bit allocation ----------------------------------------------------------------------- */
/* celt.cpp:3298 (decoder side) */
static int interp_bits2pulses(oc_rc *rc, int start, int end, int skip_start, const i32 *bits1, const i32 *bits2,
                              const i32 *thresh, const i32 *cap, i32 total, i32 *balance_out, int skip_rsv,
                              i32 *intensity, int intensity_rsv, i32 *dual_stereo, int dual_stereo_rsv, i32 *bits,
                              i32 *ebits, i32 *fine_priority, int C, int LM) {
    i32 psum, left, percoeff, balance;
    int lo = 0, hi = 1 << 6, i, j, done, codedBands;
    int alloc_floor = C << BITRES, stereo = C > 1, logM = LM << BITRES;
    const i32 *eb = rom_eband;
    for (i = 0; i < 6; i++) {
        int mid = (lo + hi) >> 1;
        psum = 0;
        done = 0;
        for (j = end; j-- > start;) {
            i32 tmp = bits1[j] + (mid * (i32)bits2[j] >> 6);
            if (tmp >= thresh[j] || done) {
                done = 1;
                psum += OC_MIN(tmp, cap[j]);
            } else if (tmp >= alloc_floor)
                psum += alloc_floor;
        }
        if (psum > total) hi = mid; else lo = mid;
    }
    psum = 0;
    done = 0;
    for (j = end; j-- > start;) {
        i32 tmp = bits1[j] + ((i32)lo * bits2[j] >> 6);
        if (tmp < thresh[j] && !done)
            tmp = tmp >= alloc_floor ? alloc_floor : 0;
        else
            done = 1;
        tmp = OC_MIN(tmp, cap[j]);
        bits[j] = tmp;
        psum += tmp;
    }
    for (codedBands = end;; codedBands--) {
        i32 band_width, band_bits, rem;
        j = codedBands - 1;
        if (j <= skip_start) {
            total += skip_rsv;
            break;
        }
        left = total - psum;
        percoeff = (i32)((u32)left / (u32)(eb[codedBands] - eb[start]));
        left -= (eb[codedBands] - eb[start]) * percoeff;
        rem = OC_MAX(left - (eb[j] - eb[start]), 0);
        band_width = eb[codedBands] - eb[j];
        band_bits = bits[j] + percoeff * band_width + rem;
        if (band_bits >= OC_MAX(thresh[j], alloc_floor + (1 << BITRES))) {
            if (oc_rc_bit_logp(rc, 1)) break;
            psum += 1 << BITRES;
            band_bits -= 1 << BITRES;
        }
        psum -= bits[j] + intensity_rsv;
        if (intensity_rsv > 0) intensity_rsv = rom_log2_frac[j - start];
        psum += intensity_rsv;
        if (band_bits >= alloc_floor) {
            psum += alloc_floor;
            bits[j] = alloc_floor;
        } else
            bits[j] = 0;
    }
    if (intensity_rsv > 0)
        *intensity = start + oc_rc_uint(rc, codedBands + 1 - start);
    else
        *intensity = 0;
    if (*intensity <= start) {
        total += dual_stereo_rsv;
        dual_stereo_rsv = 0;
    }
    if (dual_stereo_rsv > 0)
        *dual_stereo = oc_rc_bit_logp(rc, 1);
    else
        *dual_stereo = 0;

    left = total - psum;
    percoeff = (i32)((u32)left / (u32)(eb[codedBands] - eb[start]));
    left -= (eb[codedBands] - eb[start]) * percoeff;
    for (j = start; j < codedBands; j++) bits[j] += percoeff * (eb[j + 1] - eb[j]);
    for (j = start; j < codedBands; j++) {
        i32 tmp = OC_MIN(left, (i32)(eb[j + 1] - eb[j]));
        bits[j] += tmp;
        left -= tmp;
    }
    balance = 0;
    for (j = start; j < codedBands; j++) {
        i32 N0 = eb[j + 1] - eb[j], N = N0 << LM, den, offset, NClogN, excess, bit;
        bit = bits[j] + balance;
        if (N > 1) {
            excess = OC_MAX(bit - cap[j], 0);
            bits[j] = bit - excess;
            den = C * N + ((C == 2 && N > 2 && !*dual_stereo && j < *intensity) ? 1 : 0);
            NClogN = den * (rom_logn[j] + logM);
            offset = (NClogN >> 1) - den * 21;
            if (N == 2) offset += den << BITRES >> 2;
            if (bits[j] + offset < den * 2 << BITRES)
                offset += NClogN >> 2;
            else if (bits[j] + offset < den * 3 << BITRES)
                offset += NClogN >> 3;
            ebits[j] = OC_MAX(0, bits[j] + offset + (den << (BITRES - 1)));
            ebits[j] = (i32)((u32)ebits[j] / (u32)den) >> BITRES;
            if (C * ebits[j] > (bits[j] >> BITRES)) ebits[j] = bits[j] >> stereo >> BITRES;
            ebits[j] = OC_MIN(ebits[j], 8);
            fine_priority[j] = ebits[j] * (den << BITRES) >= bits[j] + offset;
            bits[j] -= C * ebits[j] << BITRES;
        } else {
            excess = OC_MAX(0, bit - (C << BITRES));
            bits[j] = bit - excess;
            ebits[j] = 0;
            fine_priority[j] = 1;
        }
        if (excess > 0) {
            i32 extra_fine = OC_MIN(excess >> (stereo + BITRES), 8 - ebits[j]);
            i32 extra_bits;
            ebits[j] += extra_fine;
            extra_bits = extra_fine * C << BITRES;
            fine_priority[j] = extra_bits >= excess - balance;
            excess -= extra_bits;
        }
        balance = excess;
    }
    *balance_out = balance;
    for (; j < end; j++) {
        ebits[j] = bits[j] >> stereo >> BITRES;
        bits[j] = 0;
        fine_priority[j] = ebits[j] < 1;
    }
    return codedBands;
}

/* celt.cpp:3523 */
static int compute_allocation(oc_rc *rc, int start, int end, const i32 *offsets, const i32 *cap, int alloc_trim,
                              i32 *intensity, i32 *dual_stereo, i32 total, i32 *balance, i32 *pulses, i32 *ebits,
                              i32 *fine_priority, int C, int LM) {
    i32 bits1[NB], bits2[NB], thresh[NB], trim_offset[NB];
    int lo, hi, j, skip_start = start, skip_rsv, intensity_rsv = 0, dual_stereo_rsv = 0;
    const i32 *eb = rom_eband;
    total = OC_MAX(total, 0);
    skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
    total -= skip_rsv;
    if (C == 2) {
        intensity_rsv = rom_log2_frac[end - start];
        if (intensity_rsv > total)
            intensity_rsv = 0;
        else {
            total -= intensity_rsv;
            dual_stereo_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
            total -= dual_stereo_rsv;
        }
    }
    for (j = start; j < end; j++) {
        int w = eb[j + 1] - eb[j];
        thresh[j] = OC_MAX(C << BITRES, (3 * w << LM << BITRES) >> 4);
        trim_offset[j] = C * w * (alloc_trim - 5 - LM) * (end - j - 1) * (1 << (LM + BITRES)) >> 6;
        if (w << LM == 1) trim_offset[j] -= C << BITRES;
    }
    lo = 1;
    hi = 11 - 1;
    do {
        int done = 0, mid = (lo + hi) >> 1;
        i32 psum = 0;
        for (j = end; j-- > start;) {
            int w = eb[j + 1] - eb[j];
            i32 bitsj = C * w * rom_band_alloc[mid * NB + j] << LM >> 2;
            if (bitsj > 0) bitsj = OC_MAX(0, bitsj + trim_offset[j]);
            bitsj += offsets[j];
            if (bitsj >= thresh[j] || done) {
                done = 1;
                psum += OC_MIN(bitsj, cap[j]);
            } else if (bitsj >= C << BITRES)
                psum += C << BITRES;
        }
        if (psum > total) hi = mid - 1; else lo = mid + 1;
    } while (lo <= hi);
    hi = lo--;
    for (j = start; j < end; j++) {
        int w = eb[j + 1] - eb[j];
        i32 b1 = C * w * rom_band_alloc[lo * NB + j] << LM >> 2;
        i32 b2 = hi >= 11 ? cap[j] : C * w * rom_band_alloc[hi * NB + j] << LM >> 2;
        if (b1 > 0) b1 = OC_MAX(0, b1 + trim_offset[j]);
        if (b2 > 0) b2 = OC_MAX(0, b2 + trim_offset[j]);
        if (lo > 0) b1 += offsets[j];
        b2 += offsets[j];
        if (offsets[j] > 0) skip_start = j;
        b2 = OC_MAX(0, b2 - b1);
        bits1[j] = b1;
        bits2[j] = b2;
    }
    return interp_bits2pulses(rc, start, end, skip_start, bits1, bits2, thresh, cap, total, balance, skip_rsv,
                              intensity, intensity_rsv, dual_stereo, dual_stereo_rsv, pulses, ebits, fine_priority,
                              C, LM);
}

/* ---- band decoding ------------------------------------------------------------------------ */
typedef struct {
    oc_rc *rc;
    int band, intensity, spread, tf_change, disable_inv;
    i32 remaining_bits;
    u32 seed;
} bandctx;

static inline const u8 *pulse_cache(int band, int LM) { /* celt.h:543 */
    return rom_pulse_bits + rom_pulse_idx[(LM + 1) * NB + band];
}
static int bits2pulses(int band, int LM, int bits) { /* celt.h:537 */
    const u8 *cache = pulse_cache(band, LM);
    int lo = 0, hi = cache[0], i;
    bits--;
    for (i = 0; i < 6; i++) {
        int mid = (lo + hi + 1) >> 1;
        if ((int)cache[mid] >= bits) hi = mid; else lo = mid;
    }
    return bits - (lo == 0 ? -1 : (int)cache[lo]) <= (int)cache[hi] - bits ? lo : hi;
}
static int pulses2bits(int band, int LM, int pulses) { /* celt.h:563 */
    return pulses == 0 ? 0 : pulse_cache(band, LM)[pulses] + 1;
}
static inline int get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); } /* celt.h:533 */
static inline u32 lcg(u32 s) { return 1664525u * s + 1013904223u; }                         /* celt.cpp:921 */

/* celt.cpp:926 */
static i16 bitexact_cos(i16 x) {
    i32 tmp = (4096 + (i32)x * x) >> 13;
    i16 x2 = (i16)tmp;
    x2 = (i16)((32767 - x2) + fmul16(x2, -7651 + fmul16(x2, 8277 + fmul16(-626, x2))));
    return (i16)(1 + x2);
}
/* celt.cpp:937 */
static int bitexact_log2tan(int isin, int icos) {
    int lc = ilog32(icos), ls = ilog32(isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * (1 << 11) + fmul16(isin, fmul16(isin, -2597) + 7932) - fmul16(icos, fmul16(icos, -2597) + 7932);
}

/* celt.cpp:797 */
static void renormalise(i16 *X, int N, i16 gain) {
    i32 E = 1, t;
    int i, k;
    i16 g;
    for (i = 0; i < N; i++) E += m16(X[i], X[i]);
    k = ilog2p(E) >> 1;
    t = vshr32(E, 2 * (k - 7));
    g = (i16)m16_p15(oc_rsqrt_norm(t), gain);
    for (i = 0; i < N; i++) X[i] = (i16)pshr32(m16(g, X[i]), k + 1);
}

/* celt.cpp:782 (decode_pulses :2622, normalise_residual :745, extract_collapse_mask :760) */
static u32 alg_unquant(bandctx *cx, i16 *X, int N, int K, int spread, int B, i16 gain) {
    i32 iy[176 + 4], Ryy, t;
    u32 mask;
    int i, k;
    i16 g;
    Ryy = oc_cwrsi(N, K, oc_rc_uint(cx->rc, oc_pvq_v(N, K)), iy);
    k = ilog2p(Ryy) >> 1;
    t = vshr32(Ryy, 2 * (k - 7));
    g = (i16)m16_p15(oc_rsqrt_norm(t), gain);
    for (i = 0; i < N; i++) X[i] = (i16)pshr32(m16(g, iy[i]), k + 1);
    oc_exp_rotation(X, N, -1, B, K, spread);
    if (B <= 1) return 1;
    {
        int N0 = (int)((u32)N / (u32)B), j;
        mask = 0;
        for (i = 0; i < B; i++) {
            u32 tmp = 0;
            for (j = 0; j < N0; j++) tmp |= iy[i * N0 + j];
            mask |= (u32)(tmp != 0) << i;
        }
    }
    return mask;
}

/* TEST ENTRY (unit known-answer tests of the HIP kernels' leaf decode): alg_unquant with the codeword index given instead of
 * read from the range decoder.  Returns the collapse mask; X receives N values. */
u32 oc_test_pvq_leaf(int N, int K, u32 index, int spread, int B, int gain, i16 *X) {
    i32 iy[176 + 4], Ryy, t;
    u32 mask;
    int i, k;
    i16 g;
    Ryy = oc_cwrsi(N, K, index, iy);
    k = ilog2p(Ryy) >> 1;
    t = vshr32(Ryy, 2 * (k - 7));
    g = (i16)m16_p15(oc_rsqrt_norm(t), (i16)gain);
    for (i = 0; i < N; i++) X[i] = (i16)pshr32(m16(g, iy[i]), k + 1);
    oc_exp_rotation(X, N, -1, B, K, spread);
    if (B <= 1) return 1;
    {
        int N0 = (int)((u32)N / (u32)B), j;
        mask = 0;
        for (i = 0; i < B; i++) {
            u32 tmp = 0;
            for (j = 0; j < N0; j++) tmp |= iy[i * N0 + j];
            mask |= (u32)(tmp != 0) << i;
        }
    }
    return mask;
}

/* celt.cpp:1113 */
static void stereo_merge(i16 *X, i16 *Y, i16 mid, int N) {
    i32 xp = 0, side = 0, El, Er, t, lgain, rgain;
    i16 mid2;
    int j, kl, kr;
    for (j = 0; j < N; j++) { /* dual_inner_prod(Y, X, Y) */
        xp += m16(Y[j], X[j]);
        side += m16(Y[j], Y[j]);
    }
    xp = m16x32_q15(mid, xp);
    mid2 = (i16)(mid >> 1);
    El = m16(mid2, mid2) + side - 2 * xp;
    Er = m16(mid2, mid2) + side + 2 * xp;
    if (Er < 161061 || El < 161061) { /* QCONST32(6e-4f,28) */
        memcpy(Y, X, N * sizeof(*Y));
        return;
    }
    kl = ilog2p(El) >> 1;
    kr = ilog2p(Er) >> 1;
    t = vshr32(El, (kl - 7) << 1);
    lgain = oc_rsqrt_norm(t);
    t = vshr32(Er, (kr - 7) << 1);
    rgain = oc_rsqrt_norm(t);
    if (kl < 7) kl = 7;
    if (kr < 7) kr = 7;
    for (j = 0; j < N; j++) {
        i16 l = (i16)m16_p15(mid, X[j]), r = Y[j];
        X[j] = (i16)pshr32(m16(lgain, sub16(l, r)), kl + 1);
        Y[j] = (i16)pshr32(m16(rgain, add16(l, r)), kr + 1);
    }
}

static const int ordery_table[] = {1, 0, 3, 0, 2, 1, 7, 0, 4, 3, 6, 1, 5, 2, 15, 0, 8, 7, 12, 3,
                                   11, 4, 14, 1, 9, 6, 13, 2, 10, 5}; /* celt.cpp:1160 */

/* celt.cpp:1162 */
static void deinterleave_hadamard(i16 *X, int N0, int stride, int hadamard) {
    i16 tmp[176];
    int i, j, N = N0 * stride;
    if (hadamard) {
        const int *ordery = ordery_table + stride - 2;
        for (i = 0; i < stride; i++)
            for (j = 0; j < N0; j++) tmp[ordery[i] * N0 + j] = X[j * stride + i];
    } else {
        for (i = 0; i < stride; i++)
            for (j = 0; j < N0; j++) tmp[i * N0 + j] = X[j * stride + i];
    }
    memcpy(X, tmp, N * sizeof(*X));
}
/* celt.cpp:1183 */
static void interleave_hadamard(i16 *X, int N0, int stride, int hadamard) {
    i16 tmp[176];
    int i, j, N = N0 * stride;
    if (hadamard) {
        const int *ordery = ordery_table + stride - 2;
        for (i = 0; i < stride; i++)
            for (j = 0; j < N0; j++) tmp[j * stride + i] = X[ordery[i] * N0 + j];
    } else {
        for (i = 0; i < stride; i++)
            for (j = 0; j < N0; j++) tmp[j * stride + i] = X[i * N0 + j];
    }
    memcpy(X, tmp, N * sizeof(*X));
}
/* celt.cpp:1202 */
static void haar1(i16 *X, int N0, int stride) {
    int i, j;
    N0 >>= 1;
    for (i = 0; i < stride; i++)
        for (j = 0; j < N0; j++) {
            i32 t1 = m16(23170, X[stride * 2 * j + i]), t2 = m16(23170, X[stride * (2 * j + 1) + i]);
            X[stride * 2 * j + i] = (i16)pshr32(t1 + t2, 15);
            X[stride * (2 * j + 1) + i] = (i16)pshr32(t1 - t2, 15);
        }
}

/* celt.cpp:1215 */
static int compute_qn(int N, int b, int offset, int pulse_cap, int stereo) {
    static const i16 exp2_table8[8] = {16384, 17866, 19483, 21247, 23170, 25267, 27554, 30048};
    int qn, qb, N2 = 2 * N - 1;
    if (stereo && N == 2) N2--;
    qb = (b + N2 * offset) / N2;
    qb = OC_MIN(b - pulse_cap - (4 << BITRES), qb);
    qb = OC_MIN(8 << BITRES, qb);
    if (qb < (1 << BITRES >> 1))
        qn = 1;
    else {
        qn = exp2_table8[qb & 0x7] >> (14 - (qb >> BITRES));
        qn = (qn + 1) >> 1 << 1;
    }
    return qn;
}

typedef struct { int inv, imid, iside, delta, itheta, qalloc; } splitctx;

/* celt.cpp:1241 (decoder branches only) */
static void compute_theta(bandctx *cx, splitctx *sc, int N, i32 *b, int B, int B0, int LM, int stereo, i32 *fill) {
    oc_rc *rc = cx->rc;
    int qn, itheta = 0, delta, imid, iside, qalloc, pulse_cap, offset, inv = 0, i = cx->band;
    u32 tell;
    pulse_cap = rom_logn[i] + LM * (1 << BITRES);
    offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
    qn = compute_qn(N, *b, offset, pulse_cap, stereo);
    if (stereo && i >= cx->intensity) qn = 1;
    tell = oc_rc_tell_frac(rc);
    if (qn != 1) {
        if (stereo && N > 2) {
            int p0 = 3, x, x0 = qn / 2, ft = p0 * (x0 + 1) + x0, fs;
            fs = oc_rc_decode(rc, ft);
            if (fs < (x0 + 1) * p0)
                x = fs / p0;
            else
                x = x0 + 1 + (fs - (x0 + 1) * p0);
            oc_rc_update(rc, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0,
                         x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
            itheta = x;
        } else if (B0 > 1 || stereo) {
            itheta = oc_rc_uint(rc, qn + 1);
        } else {
            int fs = 1, ft = ((qn >> 1) + 1) * ((qn >> 1) + 1), fl = 0, fm;
            fm = oc_rc_decode(rc, ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
                itheta = (oc_isqrt32(8 * (u32)fm + 1) - 1) >> 1;
                fs = itheta + 1;
                fl = itheta * (itheta + 1) >> 1;
            } else {
                itheta = (2 * (qn + 1) - oc_isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1;
                fs = qn + 1 - itheta;
                fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            }
            oc_rc_update(rc, fl, fl + fs, ft);
        }
        itheta = (int)((u32)(itheta * 16384) / (u32)qn);
    } else if (stereo) {
        if (*b > 2 << BITRES && cx->remaining_bits > 2 << BITRES)
            inv = oc_rc_bit_logp(rc, 2);
        else
            inv = 0;
        if (cx->disable_inv) inv = 0;
        itheta = 0;
    }
    qalloc = oc_rc_tell_frac(rc) - tell;
    *b -= qalloc;
    if (itheta == 0) {
        imid = 32767;
        iside = 0;
        *fill &= (1 << B) - 1;
        delta = -16384;
    } else if (itheta == 16384) {
        imid = 0;
        iside = 32767;
        *fill &= ((1 << B) - 1) << B;
        delta = 16384;
    } else {
        imid = bitexact_cos((i16)itheta);
        iside = bitexact_cos((i16)(16384 - itheta));
        delta = fmul16((N - 1) << 7, bitexact_log2tan(iside, imid));
    }
    sc->inv = inv;
    sc->imid = imid;
    sc->iside = iside;
    sc->delta = delta;
    sc->itheta = itheta;
    sc->qalloc = qalloc;
}

/* celt.cpp:1357 */
static u32 quant_band_n1(bandctx *cx, i16 *X, i16 *Y, i16 *lowband_out) {
    i16 *x = X;
    int c = 0, stereo = Y != NULL;
    do {
        int sign = 0;
        if (cx->remaining_bits >= 1 << BITRES) {
            sign = oc_rc_bits(cx->rc, 1);
            cx->remaining_bits -= 1 << BITRES;
        }
        x[0] = sign ? -16384 : 16384;
        x = Y;
    } while (++c < 1 + stereo);
    if (lowband_out) lowband_out[0] = X[0] >> 4;
    return 1;
}

/* celt.cpp:1382 */
static u32 quant_partition(bandctx *cx, i16 *X, int N, i32 b, int B, i16 *lowband, int LM, i16 gain, i32 fill) {
    const u8 *cache;
    int B0 = B, i = cx->band, spread = cx->spread;
    u32 cm = 0;
    cache = pulse_cache(i, LM);
    if (LM != -1 && b > cache[cache[0]] + 12 && N > 2) {
        i32 mbits, sbits, delta, rebalance;
        int itheta, qalloc;
        splitctx sc;
        i16 *next_lowband2 = NULL, *Y, mid, side;
        N >>= 1;
        Y = X + N;
        LM -= 1;
        if (B == 1) fill = (fill & 1) | (fill << 1);
        B = (B + 1) >> 1;
        compute_theta(cx, &sc, N, &b, B, B0, LM, 0, &fill);
        mid = (i16)sc.imid;
        side = (i16)sc.iside;
        delta = sc.delta;
        itheta = sc.itheta;
        qalloc = sc.qalloc;
        if (B0 > 1 && (itheta & 0x3fff)) {
            if (itheta > 8192)
                delta -= delta >> (4 - LM);
            else
                delta = OC_MIN(0, delta + (N << BITRES >> (5 - LM)));
        }
        mbits = OC_MAX(0, OC_MIN(b, (b - delta) / 2));
        sbits = b - mbits;
        cx->remaining_bits -= qalloc;
        if (lowband) next_lowband2 = lowband + N;
        rebalance = cx->remaining_bits;
        if (mbits >= sbits) {
            cm = quant_partition(cx, X, N, mbits, B, lowband, LM, (i16)m16_p15(gain, mid), fill);
            rebalance = mbits - (rebalance - cx->remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
            cm |= quant_partition(cx, Y, N, sbits, B, next_lowband2, LM, (i16)m16_p15(gain, side), fill >> B)
                  << (B0 >> 1);
        } else {
            cm = quant_partition(cx, Y, N, sbits, B, next_lowband2, LM, (i16)m16_p15(gain, side), fill >> B)
                 << (B0 >> 1);
            rebalance = sbits - (rebalance - cx->remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
            cm |= quant_partition(cx, X, N, mbits, B, lowband, LM, (i16)m16_p15(gain, mid), fill);
        }
    } else {
        int q = bits2pulses(i, LM, b), curr_bits = pulses2bits(i, LM, q), j;
        cx->remaining_bits -= curr_bits;
        while (cx->remaining_bits < 0 && q > 0) {
            cx->remaining_bits += curr_bits;
            q--;
            curr_bits = pulses2bits(i, LM, q);
            cx->remaining_bits -= curr_bits;
        }
        if (q != 0) {
            cm = alg_unquant(cx, X, N, get_pulses(q), spread, B, gain);
        } else {
            u32 cm_mask = (u32)(1UL << B) - 1;
            fill &= cm_mask;
            if (!fill) {
                memset(X, 0, N * sizeof(*X));
            } else {
                if (lowband == NULL) {
                    for (j = 0; j < N; j++) {
                        cx->seed = lcg(cx->seed);
                        X[j] = (i16)((i32)cx->seed >> 20);
                    }
                    cm = cm_mask;
                } else {
                    for (j = 0; j < N; j++) {
                        i16 tmp = 4; /* QCONST16(1/256, 10) */
                        cx->seed = lcg(cx->seed);
                        tmp = (cx->seed & 0x8000) ? tmp : -tmp;
                        X[j] = lowband[j] + tmp;
                    }
                    cm = fill;
                }
                renormalise(X, N, gain);
            }
        }
    }
    return cm;
}

/* celt.cpp:1526 */
static u32 quant_band(bandctx *cx, i16 *X, int N, i32 b, int B, i16 *lowband, int LM, i16 *lowband_out, i16 gain,
                      i16 *lowband_scratch, i32 fill) {
    static const u8 bit_interleave[16] = {0, 1, 1, 1, 2, 3, 3, 3, 2, 3, 3, 3, 2, 3, 3, 3};
    static const u8 bit_deinterleave[16] = {0x00, 0x03, 0x0C, 0x0F, 0x30, 0x33, 0x3C, 0x3F,
                                            0xC0, 0xC3, 0xCC, 0xCF, 0xF0, 0xF3, 0xFC, 0xFF};
    int N0 = N, N_B, N_B0, B0 = B, time_divide = 0, recombine = 0, longBlocks, k, tf_change = cx->tf_change;
    u32 cm;
    longBlocks = B0 == 1;
    N_B = (int)((u32)N / (u32)B);
    if (N == 1) return quant_band_n1(cx, X, NULL, lowband_out);
    if (tf_change > 0) recombine = tf_change;
    if (lowband_scratch && lowband && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
        memcpy(lowband_scratch, lowband, N * sizeof(*lowband));
        lowband = lowband_scratch;
    }
    for (k = 0; k < recombine; k++) {
        if (lowband) haar1(lowband, N >> k, 1 << k);
        fill = bit_interleave[fill & 0xF] | bit_interleave[fill >> 4] << 2;
    }
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        if (lowband) haar1(lowband, N_B, B);
        fill |= fill << B;
        B <<= 1;
        N_B >>= 1;
        time_divide++;
        tf_change++;
    }
    B0 = B;
    N_B0 = N_B;
    if (B0 > 1 && lowband) deinterleave_hadamard(lowband, N_B >> recombine, B0 << recombine, longBlocks);
    cm = quant_partition(cx, X, N, b, B, lowband, LM, gain, fill);
    if (B0 > 1) interleave_hadamard(X, N_B >> recombine, B0 << recombine, longBlocks);
    N_B = N_B0;
    B = B0;
    for (k = 0; k < time_divide; k++) {
        B >>= 1;
        N_B <<= 1;
        cm |= cm >> B;
        haar1(X, N_B, B);
    }
    for (k = 0; k < recombine; k++) {
        cm = bit_deinterleave[cm];
        haar1(X, N0 >> k, 1 << k);
    }
    B <<= recombine;
    if (lowband_out) {
        i16 n = (i16)oc_sqrt(shl32(N0, 22));
        int j;
        for (j = 0; j < N0; j++) lowband_out[j] = (i16)m16_q15(n, X[j]);
    }
    cm &= (1 << B) - 1;
    return cm;
}

/* celt.cpp:1628 */
static u32 quant_band_stereo(bandctx *cx, i16 *X, i16 *Y, int N, i32 b, int B, i16 *lowband, int LM,
                             i16 *lowband_out, i16 *lowband_scratch, i32 fill) {
    int inv, itheta, qalloc;
    i16 mid, side;
    u32 cm;
    i32 mbits, sbits, delta, orig_fill;
    splitctx sc;
    if (N == 1) return quant_band_n1(cx, X, Y, lowband_out);
    orig_fill = fill;
    compute_theta(cx, &sc, N, &b, B, B, LM, 1, &fill);
    inv = sc.inv;
    mid = (i16)sc.imid;
    side = (i16)sc.iside;
    delta = sc.delta;
    itheta = sc.itheta;
    qalloc = sc.qalloc;
    if (N == 2) {
        int c, sign = 0;
        i16 *x2, *y2, tmp;
        mbits = b;
        sbits = 0;
        if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
        mbits -= sbits;
        c = itheta > 8192;
        cx->remaining_bits -= qalloc + sbits;
        x2 = c ? Y : X;
        y2 = c ? X : Y;
        if (sbits) sign = oc_rc_bits(cx->rc, 1);
        sign = 1 - 2 * sign;
        cm = quant_band(cx, x2, N, mbits, B, lowband, LM, lowband_out, 32767, lowband_scratch, orig_fill);
        y2[0] = (i16)(-sign * x2[1]);
        y2[1] = (i16)(sign * x2[0]);
        X[0] = (i16)m16_q15(mid, X[0]);
        X[1] = (i16)m16_q15(mid, X[1]);
        Y[0] = (i16)m16_q15(side, Y[0]);
        Y[1] = (i16)m16_q15(side, Y[1]);
        tmp = X[0];
        X[0] = (i16)sub16(tmp, Y[0]);
        Y[0] = add16(tmp, Y[0]);
        tmp = X[1];
        X[1] = (i16)sub16(tmp, Y[1]);
        Y[1] = add16(tmp, Y[1]);
    } else {
        i32 rebalance;
        mbits = OC_MAX(0, OC_MIN(b, (b - delta) / 2));
        sbits = b - mbits;
        cx->remaining_bits -= qalloc;
        rebalance = cx->remaining_bits;
        if (mbits >= sbits) {
            cm = quant_band(cx, X, N, mbits, B, lowband, LM, lowband_out, 32767, lowband_scratch, fill);
            rebalance = mbits - (rebalance - cx->remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
            cm |= quant_band(cx, Y, N, sbits, B, NULL, LM, NULL, side, NULL, fill >> B);
        } else {
            cm = quant_band(cx, Y, N, sbits, B, NULL, LM, NULL, side, NULL, fill >> B);
            rebalance = sbits - (rebalance - cx->remaining_bits);
            if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
            cm |= quant_band(cx, X, N, mbits, B, lowband, LM, lowband_out, 32767, lowband_scratch, fill);
        }
    }
    if (N != 2) stereo_merge(X, Y, mid, N);
    if (inv) {
        int j;
        for (j = 0; j < N; j++) Y[j] = (i16)(-Y[j]);
    }
    return cm;
}

/* celt.cpp:1754 */
static void quant_all_bands(oc_rc *rc, int start, int end, i16 *X_, i16 *Y_, u8 *collapse_masks, const i32 *pulses,
                            int shortBlocks, int spread, int dual_stereo, int intensity, const i32 *tf_res,
                            i32 total_bits, i32 balance, int LM, int codedBands, u32 *seed, int disable_inv) {
    const i32 *eb = rom_eband;
    i16 normbuf[2 * 8 * 100]; /* C * (M*eBands[nbEBands-1] - norm_offset) */
    i16 *norm, *norm2, *lowband_scratch;
    int i, M = 1 << LM, B = shortBlocks ? M : 1, C = Y_ != NULL ? 2 : 1;
    int norm_offset = M * eb[start], lowband_offset = 0, update_lowband = 1;
    i32 remaining_bits;
    bandctx cx;
    memset(normbuf, 0, sizeof(normbuf)); /* reference leaves this malloc'd; it is never read before written */
    norm = normbuf;
    norm2 = norm + M * eb[NB - 1] - norm_offset;
    lowband_scratch = X_ + M * eb[NB - 1];
    cx.rc = rc;
    cx.intensity = intensity;
    cx.seed = *seed;
    cx.spread = spread;
    cx.disable_inv = disable_inv;
    for (i = start; i < end; i++) {
        i32 tell, b, curr_balance;
        int N, effective_lowband = -1, tf_change, last = (i == end - 1);
        i16 *X, *Y;
        u32 x_cm, y_cm;
        cx.band = i;
        X = X_ + M * eb[i];
        Y = Y_ != NULL ? Y_ + M * eb[i] : NULL;
        N = M * eb[i + 1] - M * eb[i];
        tell = oc_rc_tell_frac(rc);
        if (i != start) balance -= tell;
        remaining_bits = total_bits - tell - 1;
        cx.remaining_bits = remaining_bits;
        if (i <= codedBands - 1) {
            curr_balance = balance / OC_MIN(3, codedBands - i);
            b = OC_MAX(0, OC_MIN(16383, OC_MIN(remaining_bits + 1, pulses[i] + curr_balance)));
        } else
            b = 0;
        if ((M * eb[i] - N >= M * eb[start] || i == start + 1) && (update_lowband || lowband_offset == 0))
            lowband_offset = i;
        if (i == start + 1) { /* special_hybrid_folding celt.cpp:1743 */
            int n1 = M * (eb[start + 1] - eb[start]), n2 = M * (eb[start + 2] - eb[start + 1]);
            if (n2 > n1) {
                memcpy(&norm[n1], &norm[2 * n1 - n2], (n2 - n1) * sizeof(*norm));
                if (dual_stereo) memcpy(&norm2[n1], &norm2[2 * n1 - n2], (n2 - n1) * sizeof(*norm2));
            }
        }
        tf_change = tf_res[i];
        cx.tf_change = tf_change;
        /* i >= effEBands never happens (effEBands == nbEBands == 21) */
        if (last) lowband_scratch = NULL;
        if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
            int fold_start, fold_end, fold_i;
            effective_lowband = OC_MAX(0, M * eb[lowband_offset] - norm_offset - N);
            fold_start = lowband_offset;
            while (M * eb[--fold_start] > effective_lowband + norm_offset)
                ;
            fold_end = lowband_offset - 1;
            while (++fold_end < i && M * eb[fold_end] < effective_lowband + norm_offset + N)
                ;
            x_cm = y_cm = 0;
            fold_i = fold_start;
            do {
                x_cm |= collapse_masks[fold_i * C + 0];
                y_cm |= collapse_masks[fold_i * C + C - 1];
            } while (++fold_i < fold_end);
        } else
            x_cm = y_cm = (1 << B) - 1;
        if (dual_stereo && i == intensity) {
            int j;
            dual_stereo = 0;
            for (j = 0; j < M * eb[i] - norm_offset; j++) norm[j] = (i16)((norm[j] + norm2[j]) >> 1);
        }
        if (dual_stereo) {
            x_cm = quant_band(&cx, X, N, b / 2, B, effective_lowband != -1 ? norm + effective_lowband : NULL, LM,
                              last ? NULL : norm + M * eb[i] - norm_offset, 32767, lowband_scratch, x_cm);
            y_cm = quant_band(&cx, Y, N, b / 2, B, effective_lowband != -1 ? norm2 + effective_lowband : NULL, LM,
                              last ? NULL : norm2 + M * eb[i] - norm_offset, 32767, lowband_scratch, y_cm);
        } else {
            if (Y != NULL)
                x_cm = quant_band_stereo(&cx, X, Y, N, b, B, effective_lowband != -1 ? norm + effective_lowband : NULL,
                                         LM, last ? NULL : norm + M * eb[i] - norm_offset, lowband_scratch,
                                         x_cm | y_cm);
            else
                x_cm = quant_band(&cx, X, N, b, B, effective_lowband != -1 ? norm + effective_lowband : NULL, LM,
                                  last ? NULL : norm + M * eb[i] - norm_offset, 32767, lowband_scratch, x_cm | y_cm);
            y_cm = x_cm;
        }
        collapse_masks[i * C + 0] = (u8)x_cm;
        collapse_masks[i * C + C - 1] = (u8)y_cm;
        balance += pulses[i] + tell;
        update_lowband = b > (N << BITRES);
    }
    *seed = cx.seed;
}

/* celt.cpp:1010 */
static void anti_collapse(i16 *X_, const u8 *collapse_masks, int LM, int C, int size, int start, int end,
                          const i16 *logE, const i16 *prev1logE, const i16 *prev2logE, const i32 *pulses, u32 seed) {
    const i32 *eb = rom_eband;
    int c, i, j, k;
    for (i = start; i < end; i++) {
        int N0 = eb[i + 1] - eb[i], depth, shift;
        i16 thresh, sqrt_1;
        i32 thresh32, t;
        depth = (int)((u32)(1 + pulses[i]) / (u32)N0) >> LM;
        thresh32 = oc_exp2(-shl16(depth, 10 - BITRES)) >> 1;
        thresh = (i16)m16x32_q15(16384, OC_MIN(32767, thresh32));
        t = N0 << LM;
        shift = ilog2p(t) >> 1;
        t = shl32(t, (7 - shift) << 1);
        sqrt_1 = oc_rsqrt_norm(t);
        c = 0;
        do {
            i16 *X, prev1 = prev1logE[c * NB + i], prev2 = prev2logE[c * NB + i], r;
            i32 Ediff;
            int renorm = 0;
            if (C == 1) {
                prev1 = OC_MAX(prev1, prev1logE[NB + i]);
                prev2 = OC_MAX(prev2, prev2logE[NB + i]);
            }
            Ediff = (i32)logE[c * NB + i] - (i32)OC_MIN(prev1, prev2);
            Ediff = OC_MAX(0, Ediff);
            if (Ediff < 16384) {
                i32 r32 = oc_exp2(-(i16)Ediff) >> 1;
                r = (i16)(2 * OC_MIN(16383, r32));
            } else
                r = 0;
            if (LM == 3) r = (i16)m16_q14(23170, OC_MIN(23169, r));
            r = (i16)(OC_MIN(thresh, r) >> 1);
            r = (i16)(m16_q15(sqrt_1, r) >> shift);
            X = X_ + c * size + (eb[i] << LM);
            for (k = 0; k < 1 << LM; k++) {
                if (!(collapse_masks[i * C + c] & 1 << k)) {
                    for (j = 0; j < N0; j++) {
                        seed = lcg(seed);
                        X[(j << LM) + k] = (seed & 0x8000 ? r : -r);
                    }
                    renorm = 1;
                }
            }
            if (renorm) renormalise(X, N0 << LM, 32767);
        } while (++c < C);
    }
}

/* ---- synthesis ---------------------------------------------------------------------------- */
/* celt.cpp:948 (downsample == 1) */
static void denormalise(const i16 *X, i32 *freq, const i16 *bandLogE, int start, int end, int M, int silence) {
    const i32 *eb = rom_eband;
    int i, N = M * 120, bound = M * eb[end];
    i32 *f = freq;
    const i16 *x;
    if (silence) {
        bound = 0;
        start = end = 0;
    }
    x = X + M * eb[start];
    for (i = 0; i < M * eb[start]; i++) *f++ = 0;
    for (i = start; i < end; i++) {
        int j = M * eb[i], band_end = M * eb[i + 1], shift;
        i32 lg32 = (i32)bandLogE[i] + shl32((i32)rom_emeans[i], 6);
        i16 lg = (i16)(lg32 > 32767 ? 32767 : (lg32 < -32768 ? -32768 : lg32)), g;
        shift = 16 - (lg >> 10);
        if (shift > 31) {
            shift = 0;
            g = 0;
        } else
            g = (i16)oc_exp2_frac(lg & 1023);
        if (shift < 0) {
            if (shift <= -2) {
                g = 16384;
                shift = -2;
            }
            do {
                *f++ = shl32(m16(*x++, g), -shift);
            } while (++j < band_end);
        } else
            do {
                *f++ = m16(*x++, g) >> shift;
            } while (++j < band_end);
    }
    memset(&freq[bound], 0, (N - bound) * sizeof(*freq));
}

/* celt.cpp:2057 */
static void synthesis(oc_celt *st, const i16 *X, i32 *out_syn[2], const i16 *bandE, int start, int effEnd, int C,
                      int CC, int transient, int LM, int silence, oc_celt_taps *taps) {
    i32 freq[960];
    int c, i, b, M = 1 << LM, N = 120 << LM, B, NBk, shift;
    (void)st;
    if (transient) {
        B = M;
        NBk = 120;
        shift = 3;
    } else {
        B = 1;
        NBk = 120 << LM;
        shift = 3 - LM;
    }
    if (CC == 2 && C == 1) {
        i32 *freq2;
        denormalise(X, freq, bandE, start, effEnd, M, silence);
        freq2 = out_syn[1] + OC_OVERLAP / 2;
        memcpy(freq2, freq, N * sizeof(*freq2));
        if (taps) { memcpy(taps->freq[0], freq, N * 4); memcpy(taps->freq[1], freq, N * 4); }
        for (b = 0; b < B; b++) oc_imdct(&freq2[b], out_syn[0] + NBk * b, OC_OVERLAP, shift, B);
        for (b = 0; b < B; b++) oc_imdct(&freq[b], out_syn[1] + NBk * b, OC_OVERLAP, shift, B);
    } else if (CC == 1 && C == 2) {
        i32 *freq2 = out_syn[0] + OC_OVERLAP / 2;
        denormalise(X, freq, bandE, start, effEnd, M, silence);
        denormalise(X + N, freq2, bandE + NB, start, effEnd, M, silence);
        for (i = 0; i < N; i++) freq[i] = (freq[i] >> 1) + (freq2[i] >> 1);
        if (taps) memcpy(taps->freq[0], freq, N * 4);
        for (b = 0; b < B; b++) oc_imdct(&freq[b], out_syn[0] + NBk * b, OC_OVERLAP, shift, B);
    } else {
        c = 0;
        do {
            denormalise(X + c * N, freq, bandE + c * NB, start, effEnd, M, silence);
            if (taps) memcpy(taps->freq[c], freq, N * 4);
            for (b = 0; b < B; b++) oc_imdct(&freq[b], out_syn[c] + NBk * b, OC_OVERLAP, shift, B);
        } while (++c < CC);
    }
    c = 0;
    do {
        for (i = 0; i < N; i++) out_syn[c][i] = satsym(out_syn[c][i], OC_SIG_SAT);
    } while (++c < CC);
}

/* celt.cpp:1965, :1988 (downsample 1, no accumulation): 1-pole IIR + rounding to int16 */
static void deemphasis(i32 *in[2], i16 *pcm, int N, int C, i32 *mem) {
    const i16 coef0 = 27853; /* m_CELTMode.preemph[0] celt.cpp:634 */
    int c, j;
    for (c = 0; c < C; c++) {
        i32 m = mem[c];
        const i32 *x = in[c];
        for (j = 0; j < N; j++) {
            i32 tmp = x[j] + m, v;
            m = m16x32_q15(coef0, tmp);
            v = pshr32(tmp, 12); /* sig2word16 celt.h:413 */
            pcm[j * C + c] = (i16)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
        mem[c] = m;
    }
}

/* ---- frame driver (celt.cpp:2162) --------------------------------------------------------- */
int oc_celt_decode(oc_celt *st, oc_rc *rc, i16 *pcm, int frame_size, oc_celt_taps *taps) {
    const i32 *eb = rom_eband;
    const int CC = st->channels, C = st->stream_channels;
    i32 *out_syn[2];
    i16 X[2 * 960];
    i32 tf_res[NB], cap[NB], offsets[NB], fine_quant[NB], pulses[NB], fine_priority[NB];
    u8 collapse_masks[2 * NB];
    i16 *bandE = st->bandE, *logE1 = st->logE1, *logE2 = st->logE2;
    /* the reference ignores its END_BAND request and always decodes 21 bands (Q1); RFC mode sets end_band by bandwidth */
    int c, i, N, LM, M, start = st->start_band, end = (st->end_band > 0 && st->end_band <= NB) ? st->end_band : NB, effEnd = end;
    int spread, shortBlocks, transient, intra, codedBands, alloc_trim, pf_pitch = 0, pf_tapset = 0;
    i16 pf_gain = 0;
    i32 intensity = 0, dual_stereo = 0, total_bits, balance, tell, bits;
    int dynalloc_logp, anti_collapse_rsv, anti_collapse_on = 0, silence;

    for (LM = 0; LM <= 3; LM++)
        if (120 << LM == frame_size) break;
    if (LM > 3) return OC_CELT_BAD_ARG; /* celt.cpp:2211 */
    M = 1 << LM;
    if (rc->storage > 1275 || pcm == NULL) return OC_CELT_BAD_ARG; /* :2216 */
    N = M * 120;
    for (c = 0; c < CC; c++) out_syn[c] = st->syn[c] + OC_HIST;
    if (rc->storage <= 1) return OC_CELT_BAD_ARG; /* :2225 */

    if (C == 1)
        for (i = 0; i < NB; i++) bandE[i] = OC_MAX(bandE[i], bandE[NB + i]);

    total_bits = rc->storage * 8;
    tell = oc_rc_tell(rc);
    if (tell >= total_bits)
        silence = 1;
    else if (tell == 1)
        silence = oc_rc_bit_logp(rc, 15);
    else
        silence = 0;
    if (silence) {
        tell = rc->storage * 8;
        rc->nbits_total += tell - oc_rc_tell(rc);
    }
    if (start == 0 && tell + 16 <= total_bits) {
        if (oc_rc_bit_logp(rc, 1)) {
            int qg, octave = oc_rc_uint(rc, 6);
            pf_pitch = (16 << octave) + oc_rc_bits(rc, 4 + octave) - 1;
            qg = oc_rc_bits(rc, 3);
            if (oc_rc_tell(rc) + 2 <= total_bits) pf_tapset = oc_rc_icdf(rc, tapset_icdf, 2);
            pf_gain = (i16)(3072 * (qg + 1)); /* QCONST16(.09375,15) */
        }
        tell = oc_rc_tell(rc);
    }
    if (LM > 0 && tell + 3 <= total_bits) {
        transient = oc_rc_bit_logp(rc, 3);
        tell = oc_rc_tell(rc);
    } else
        transient = 0;
    shortBlocks = transient ? M : 0;
    intra = tell + 3 <= total_bits ? oc_rc_bit_logp(rc, 3) : 0;
    coarse_energy(rc, start, end, bandE, intra, C, LM);
    tf_decode(rc, start, end, transient, tf_res, LM);
    tell = oc_rc_tell(rc);
    spread = 2;
    if (tell + 4 <= total_bits) spread = oc_rc_icdf(rc, spread_icdf, 5);

    for (i = 0; i < NB; i++) { /* init_caps celt.cpp:911 */
        int Nb = (eb[i + 1] - eb[i]) << LM;
        cap[i] = (rom_pulse_caps[NB * (2 * LM + C - 1) + i] + 64) * C * Nb >> 2;
    }
    dynalloc_logp = 6;
    total_bits <<= BITRES;
    tell = oc_rc_tell_frac(rc);
    for (i = start; i < end; i++) {
        int width = C * (eb[i + 1] - eb[i]) << LM, quanta, loop_logp = dynalloc_logp, boost = 0;
        quanta = OC_MIN(width << BITRES, OC_MAX(6 << BITRES, width));
        while (tell + (loop_logp << BITRES) < total_bits && boost < cap[i]) {
            int flag = oc_rc_bit_logp(rc, loop_logp);
            tell = oc_rc_tell_frac(rc);
            if (!flag) break;
            boost += quanta;
            total_bits -= quanta;
            loop_logp = 1;
        }
        offsets[i] = boost;
        if (boost > 0) dynalloc_logp = OC_MAX(2, dynalloc_logp - 1);
    }
    for (i = 0; i < start; i++) offsets[i] = 0;
    alloc_trim = tell + (6 << BITRES) <= total_bits ? oc_rc_icdf(rc, trim_icdf, 7) : 5;
    bits = (((i32)rc->storage * 8) << BITRES) - oc_rc_tell_frac(rc) - 1;
    anti_collapse_rsv = transient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
    bits -= anti_collapse_rsv;
    memset(pulses, 0, sizeof(pulses));
    memset(fine_quant, 0, sizeof(fine_quant));
    memset(fine_priority, 0, sizeof(fine_priority));
    codedBands = compute_allocation(rc, start, end, offsets, cap, alloc_trim, &intensity, &dual_stereo, bits, &balance,
                                    pulses, fine_quant, fine_priority, C, LM);
    fine_energy(rc, start, end, bandE, fine_quant, C);

    /* The reference shifts its END-anchored history left by N here (celt.cpp:2349).  This buffer is
       START-anchored (out_syn at OC_HIST), so the equivalent shift is done after the frame, by the
       frame's own N (see the end of this function): frames of different sizes may follow each other (Q4). */

    memset(collapse_masks, 0, sizeof(collapse_masks));
    memset(X, 0, sizeof(X)); /* reference: malloc'd; bands below `start` are never read by denormalise */
    quant_all_bands(rc, start, end, X, C == 2 ? X + N : NULL, collapse_masks, pulses, shortBlocks, spread, dual_stereo,
                    intensity, tf_res, rc->storage * (8 << BITRES) - anti_collapse_rsv, balance, LM, codedBands,
                    &st->rng, st->disable_inv);
    if (anti_collapse_rsv > 0) anti_collapse_on = oc_rc_bits(rc, 1);
    energy_finalise(rc, start, end, bandE, fine_quant, fine_priority, rc->storage * 8 - oc_rc_tell(rc), C);
    if (anti_collapse_on)
        anti_collapse(X, collapse_masks, LM, C, N, start, end, bandE, logE1, logE2, pulses, st->rng);
    if (silence)
        for (i = 0; i < C * NB; i++) bandE[i] = -28 * 1024;

    if (taps) {
        taps->valid = 1;
        taps->is_transient = transient; taps->silence = silence; taps->coded_bands = codedBands;
        taps->intensity = intensity; taps->dual_stereo = dual_stereo; taps->spread = spread; taps->LM = LM;
        taps->pf_pitch = pf_pitch; taps->pf_gain = pf_gain; taps->pf_tapset = pf_tapset;
        taps->anti_collapse_on = anti_collapse_on;
        memcpy(taps->pulses, pulses, sizeof(pulses));
        memcpy(taps->fine_quant, fine_quant, sizeof(fine_quant));
        memcpy(taps->tf_res, tf_res, sizeof(tf_res));
        memcpy(taps->X, X, sizeof(X));
        memcpy(taps->bandE, bandE, sizeof(taps->bandE));
    }
    synthesis(st, X, out_syn, bandE, start, effEnd, C, CC, transient, LM, silence, taps);
    if (taps)
        for (c = 0; c < CC; c++) memcpy(taps->syn_pre[c], out_syn[c], (N + OC_OVERLAP) * sizeof(i32));

    c = 0;
    do {
        st->pf_period = OC_MAX(st->pf_period, 15);
        st->pf_period_old = OC_MAX(st->pf_period_old, 15);
        oc_comb_filter(out_syn[c], out_syn[c], st->pf_period_old, st->pf_period, 120, st->pf_gain_old, st->pf_gain,
                       st->pf_tapset_old, st->pf_tapset);
        if (LM != 0)
            oc_comb_filter(out_syn[c] + 120, out_syn[c] + 120, st->pf_period, pf_pitch, N - 120, st->pf_gain, pf_gain,
                           st->pf_tapset, pf_tapset);
    } while (++c < CC);
    st->pf_period_old = st->pf_period;
    st->pf_gain_old = st->pf_gain;
    st->pf_tapset_old = st->pf_tapset;
    st->pf_period = pf_pitch;
    st->pf_gain = pf_gain;
    st->pf_tapset = pf_tapset;
    if (LM != 0) {
        st->pf_period_old = st->pf_period;
        st->pf_gain_old = st->pf_gain;
        st->pf_tapset_old = st->pf_tapset;
    }
    if (taps)
        for (c = 0; c < CC; c++) memcpy(taps->syn_post[c], out_syn[c], N * sizeof(i32));

    if (C == 1) memcpy(&bandE[NB], bandE, NB * sizeof(*bandE));
    if (!transient) {
        memcpy(logE2, logE1, 2 * NB * sizeof(*logE2));
        memcpy(logE1, bandE, 2 * NB * sizeof(*logE1));
        { /* celt.cpp:2411-2418: the noise floor rises by at most 2.4 dB/s, or 1 dB per update after a long loss */
            const i16 inc = st->loss_count < 10 ? (i16)(M * 1) : 1024; /* M * QCONST16(0.001f, DB_SHIFT), QCONST16(1.f, DB_SHIFT) */
            for (i = 0; i < 2 * NB; i++) st->backgroundLogE[i] = OC_MIN((i16)(st->backgroundLogE[i] + inc), bandE[i]);
        }
    } else {
        for (i = 0; i < 2 * NB; i++) logE1[i] = OC_MIN(logE1[i], bandE[i]);
    }
    c = 0;
    do {
        for (i = 0; i < start; i++) {
            bandE[c * NB + i] = 0;
            logE1[c * NB + i] = logE2[c * NB + i] = -28 * 1024;
        }
        for (i = end; i < NB; i++) {
            bandE[c * NB + i] = 0;
            logE1[c * NB + i] = logE2[c * NB + i] = -28 * 1024;
        }
    } while (++c < 2);
    st->rng = rc->rng;
    if (taps) taps->rc_rng_end = rc->rng;

    deemphasis(out_syn, pcm, N, CC, st->deemph_mem);
    st->loss_count = 0;
    for (c = 0; c < CC; c++) /* keep OC_HIST samples of history + the 60-sample overlap tail at out_syn[0..60) */
        memmove(st->syn[c], st->syn[c] + N, (OC_HIST + OC_OVERLAP / 2) * sizeof(i32));
    if (oc_rc_tell(rc) > 8 * (i32)rc->storage) return OC_INTERNAL_ERROR;
    if (rc->error) st->error = 1;
    return frame_size;
}

/* ---- pitch-based concealment (RFC mode) -------------------------------------------------------------------
 * What RFC 6716's decoder does for the first lost frames of a CELT-only stream (celt_decode_lost with loss_count < 5 and
 * start == 0): find the pitch period of the last output, take the LPC residual of the last two periods, repeat it -- decaying
 * by the energy ratio of its two halves per period -- through the LPC synthesis filter continued from the history, keep the
 * result from getting louder than what it continues, and leave an overlap tail (pre-filtered against the post-filter, folded
 * by the window) for the next decoded frame's transform to blend into.  A decoder's concealment is not normative and neither
 * the reference nor this image holds libopus' source: the STRUCTURE follows that decoder, the fixed-point detail below is this
 * repository's own (64-bit accumulators instead of libopus' block-wise shifts; the 1,024 samples of history the decoder keeps
 * anyway instead of 2,048, so the pitch search correlates the last 304 samples against lags 100 .. 720).  PARITY-UNPINNED; the
 * HIP side (og_plc.hpp) makes the same choices and is compared with this, sample by sample. */
#define PLC_LPC 24
#define PLC_PMIN 100
#define PLC_PMAX 720
static int ilog64(unsigned long long x) { /* bits needed: 0 for 0 */
    int n = 0;
    while (x) {
        n++;
        x >>= 1;
    }
    return n;
}
static u32 isqrt64(unsigned long long x) { /* floor(sqrt(x)) for x < 2^62 */
    unsigned long long r = 0, bit = 1ull << 60;
    while (bit > x) bit >>= 2;
    while (bit) {
        if (x >= r + bit) {
            x -= r + bit;
            r = (r >> 1) + bit;
        } else
            r >>= 1;
        bit >>= 2;
    }
    return (u32)r;
}
static i32 mul32_q31(i32 a, i32 b) { return (i32)(((i64)a * b) >> 31); }
static i16 sat16_64(i64 x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : (i16)x); }
#define PLC_ABS(x) ((x) < 0 ? -(x) : (x))
/* sqrt(a / b) in Q15, at most 32767 (a, b >= 0, b > 0; both scaled down together until b < 2^30) */
static i16 plc_ratio_q15(i64 a, i64 b) {
    const int sh = OC_MAX(0, ilog64((unsigned long long)b) - 30);
    u32 r;
    a >>= sh;
    b >>= sh;
    if (b <= 0) return 32767;
    if (a >= b) return 32767;
    r = isqrt64((unsigned long long)(a << 30) / (unsigned long long)b);
    return (i16)OC_MIN((u32)32767, r);
}

/* pitch period of v[0 .. 1024) (16-bit samples, newest last), 100 .. 720 */
static int plc_pitch_search(const i16 *v) {
    i16 lp[512], w[1024];
    int i, L, best = 50, sh, mx = 0;
    i64 bnum = -1, bden = 1;
    /* half rate: (1 2 1) / 4, then scaled so that |lp| < 2^9 (sums of 152 products stay below 2^27) */
    for (i = 0; i < 512; i++) {
        const i32 a = i ? v[2 * i - 1] : 0, b = v[2 * i], c = v[2 * i + 1];
        lp[i] = (i16)((a + 2 * b + c + 2) >> 2);
        if (PLC_ABS(lp[i]) > mx) mx = PLC_ABS(lp[i]);
    }
    sh = OC_MAX(0, ilog64((unsigned long long)mx) - 9);
    for (i = 0; i < 512; i++) lp[i] = (i16)(lp[i] >> sh);
    for (L = PLC_PMIN / 2; L <= PLC_PMAX / 2; L++) { /* the last 152 samples against the 152 samples L earlier */
        i64 xc = 0, en = 1, num;
        for (i = 0; i < 152; i++) {
            xc += (i32)lp[360 + i] * lp[360 - L + i];
            en += (i32)lp[360 - L + i] * lp[360 - L + i];
        }
        if (xc <= 0) continue;
        num = (xc * xc) >> 20;
        if (num * bden > bnum * en) { /* strictly better: the shortest lag wins a tie */
            bnum = num;
            bden = en;
            best = L;
        }
    }
    /* full rate around 2 * best */
    mx = 0;
    for (i = 0; i < 1024; i++)
        if (PLC_ABS(v[i]) > mx) mx = PLC_ABS(v[i]);
    sh = OC_MAX(0, ilog64((unsigned long long)mx) - 9);
    for (i = 0; i < 1024; i++) w[i] = (i16)(v[i] >> sh);
    {
        int P, bestP = 2 * best;
        bnum = -1;
        bden = 1;
        for (P = 2 * best - 1; P <= 2 * best + 1; P++) {
            i64 xc = 0, en = 1, num;
            if (P < PLC_PMIN || P > PLC_PMAX) continue;
            for (i = 0; i < 304; i++) {
                xc += (i32)w[720 + i] * w[720 - P + i];
                en += (i32)w[720 - P + i] * w[720 - P + i];
            }
            if (xc <= 0) continue;
            num = (xc * xc) >> 20;
            if (num * bden > bnum * en) {
                bnum = num;
                bden = en;
                bestP = P;
            }
        }
        return OC_MIN(PLC_PMAX, OC_MAX(PLC_PMIN, bestP));
    }
}

/* order-24 LPC of v[0 .. 1024) in Q12 (autocorrelation, -40 dB noise floor, lag window, Levinson-Durbin) */
static void plc_lpc(const i16 *v, i16 *lpc16) {
    i64 acc[PLC_LPC + 1];
    i32 ac[PLC_LPC + 1], lpc[PLC_LPC], err;
    int i, j, k, sh;
    for (k = 0; k <= PLC_LPC; k++) {
        i64 a = 0;
        for (i = k; i < 1024; i++) a += (i32)v[i] * v[i - k];
        acc[k] = a;
    }
    sh = OC_MAX(0, ilog64((unsigned long long)acc[0]) - 29);
    for (k = 0; k <= PLC_LPC; k++) ac[k] = (i32)(acc[k] >> sh);
    ac[0] += ac[0] >> 13;
    for (k = 1; k <= PLC_LPC; k++) ac[k] -= (i32)(((i64)ac[k] * (2 * k * k)) >> 15);
    for (i = 0; i < PLC_LPC; i++) lpc[i] = 0;
    err = ac[0];
    if (ac[0] > 0)
        for (i = 0; i < PLC_LPC; i++) {
            i64 rr = 0, q;
            i32 r;
            for (j = 0; j < i; j++) rr += mul32_q31(lpc[j], ac[i - j]); /* lpc in Q25 */
            rr += ac[i + 1] >> 6;
            q = -(rr * 64 * 33554432) / err; /* reflection coefficient, Q25 */
            if (q > (1 << 25) - 1) q = (1 << 25) - 1;
            if (q < -(1 << 25) + 1) q = -(1 << 25) + 1;
            r = (i32)q;
            lpc[i] = r;
            for (j = 0; j < (i + 1) >> 1; j++) {
                const i32 t1 = lpc[j], t2 = lpc[i - 1 - j];
                lpc[j] = t1 + (i32)(((i64)r * t2) >> 25);
                lpc[i - 1 - j] = t2 + (i32)(((i64)r * t1) >> 25);
            }
            err -= (i32)(((i64)(i32)(((i64)r * r) >> 25) * err) >> 25);
            if (err < (ac[0] >> 10)) break;
        }
    for (i = 0; i < PLC_LPC; i++) lpc16[i] = sat16(pshr32(lpc[i], 13));
}

static void celt_decode_lost_pitch(oc_celt *st, i16 *pcm, int N) {
    static const i16 gains[3][3] = {{10048, 7112, 4248}, {15200, 8784, 0}, {26208, 3280, 0}};
    const int C = st->channels, len = N + OC_OVERLAP, cmp = OC_MIN(len, 1024);
    i32 *out_syn[2];
    i16 v[2][1024 + PLC_LPC], mono[1024], lpc16[PLC_LPC], e[1024], sy[PLC_LPC + 960 + OC_OVERLAP];
    i32 etmp[OC_OVERLAP];
    int c, i, j, pitch, exc_len;
    for (c = 0; c < C; c++) {
        out_syn[c] = st->syn[c] + OC_HIST;
        for (i = 0; i < PLC_LPC; i++) v[c][i] = 0; /* (nothing older than the history is known) */
        for (i = 0; i < 1024; i++) v[c][PLC_LPC + i] = sat16(pshr32(st->syn[c][i], 12));
    }
    for (i = 0; i < 1024; i++) mono[i] = C == 2 ? (i16)((v[0][PLC_LPC + i] + v[1][PLC_LPC + i]) >> 1) : v[0][PLC_LPC + i];
    if (st->loss_count == 0) st->plc_pitch = plc_pitch_search(mono);
    pitch = st->plc_pitch;
    exc_len = OC_MIN(2 * pitch, 1000);
    for (c = 0; c < C; c++) {
        const i16 *x = v[c] + PLC_LPC; /* x[-24 .. 1024) */
        const i16 fade = st->loss_count == 0 ? 32767 : 26214; /* Q15: 1, 0.8 */
        i64 E1 = 1, E2 = 1, S1 = 0, S2 = 0;
        i32 att;
        i16 decay;
        if (st->loss_count == 0) {
            plc_lpc(x, lpc16);
            for (i = 0; i < PLC_LPC; i++) st->plc_lpc[c][i] = lpc16[i];
        } else
            for (i = 0; i < PLC_LPC; i++) lpc16[i] = st->plc_lpc[c][i];
        /* the residual of the last exc_len samples */
        for (i = 1024 - exc_len; i < 1024; i++) {
            i64 a = 0;
            for (j = 0; j < PLC_LPC; j++) a += (i32)lpc16[j] * x[i - 1 - j];
            e[i] = sat16_64(x[i] + ((a + 2048) >> 12));
        }
        /* how much it decays from its first half to its second */
        for (i = 0; i < exc_len / 2; i++) {
            const i32 a = e[1024 - exc_len / 2 + i], b = e[1024 - 2 * (exc_len / 2) + i];
            E1 += a * a;
            E2 += b * b;
        }
        decay = plc_ratio_q15(OC_MIN(E1, E2), E2);
        /* the period before the end, again and again, a little quieter each time, through the synthesis filter */
        for (i = 0; i < PLC_LPC; i++) sy[i] = x[1024 - PLC_LPC + i];
        att = m16_q15(fade, decay);
        for (i = 0, j = 0; i < len; i++, j++) {
            i64 a;
            int k;
            if (j >= pitch) {
                j -= pitch;
                att = m16_q15(att, decay);
            }
            a = (i64)m16_q15(att, e[1024 - pitch + j]) * 4096;
            for (k = 0; k < PLC_LPC; k++) a -= (i32)lpc16[k] * sy[PLC_LPC + i - 1 - k];
            sy[PLC_LPC + i] = sat16_64((a + 2048) >> 12);
        }
        /* not louder than what it continues */
        for (i = 0; i < cmp; i++) {
            S1 += (i32)x[1024 - cmp + i] * x[1024 - cmp + i];
            S2 += (i32)sy[PLC_LPC + i] * sy[PLC_LPC + i];
        }
        if (!(S1 > (S2 >> 2)))
            for (i = 0; i < len; i++) sy[PLC_LPC + i] = 0;
        else if (S1 < S2) {
            const i16 ratio = plc_ratio_q15((S1 >> 1) + 1, S2 + 1);
            for (i = 0; i < OC_OVERLAP; i++) {
                const i16 g = (i16)(32767 - m16_q15(rom_win120[i], 32767 - ratio));
                sy[PLC_LPC + i] = (i16)m16_q15(g, sy[PLC_LPC + i]);
            }
            for (i = OC_OVERLAP; i < len; i++) sy[PLC_LPC + i] = (i16)m16_q15(ratio, sy[PLC_LPC + i]);
        }
        for (i = 0; i < len; i++) out_syn[c][i] = (i32)sy[PLC_LPC + i] * 4096;
        /* the overlap for the next frame: pre-filtered against the post-filter that frame will run over it, folded by the window */
        {
            const int T = OC_MAX(st->pf_period, 15), tap = st->pf_tapset;
            const i16 g = (i16)-st->pf_gain;
            const i16 g0 = (i16)m16_p15(g, gains[tap][0]), g1 = (i16)m16_p15(g, gains[tap][1]), g2 = (i16)m16_p15(g, gains[tap][2]);
            const i32 *xx = out_syn[c] + N;
            for (i = 0; i < OC_OVERLAP; i++) {
                i32 y = xx[i];
                if (st->pf_gain != 0)
                    y = satsym(y + m16x32_q15(g0, xx[i - T]) + m16x32_q15(g1, xx[i - T + 1] + xx[i - T - 1]) +
                                   m16x32_q15(g2, xx[i - T + 2] + xx[i - T - 2]), OC_SIG_SAT);
                etmp[i] = y;
            }
            for (i = 0; i < OC_OVERLAP / 2; i++)
                out_syn[c][N + i] = m16x32_q15(rom_win120[i], etmp[OC_OVERLAP - 1 - i]) + m16x32_q15(rom_win120[OC_OVERLAP - 1 - i], etmp[i]);
        }
    }
    deemphasis(out_syn, pcm, N, C, st->deemph_mem);
    st->loss_count++;
    for (c = 0; c < C; c++) memmove(st->syn[c], st->syn[c] + N, (OC_HIST + OC_OVERLAP / 2) * sizeof(i32));
}

/* RFC 6716's celt_decode_lost: the pitch-based branch above for the first lost frames of a stream that codes from band 0, else the
 * noise-based branch, then the tail of celt_decode_with_ec for a lost frame (de-emphasis only) */
int oc_celt_decode_lost(oc_celt *st, i16 *pcm, int frame_size) {
    const i32 *eb = rom_eband;
    const int C = st->channels; /* the concealment runs over the decoder's channels, not the last packet's */
    const int start = st->start_band, end = (st->end_band > 0 && st->end_band <= NB) ? st->end_band : NB;
    const int effEnd = OC_MAX(start, OC_MIN(end, NB));
    const i16 decay = st->loss_count == 0 ? 1536 : 512; /* QCONST16(1.5f, DB_SHIFT) : QCONST16(.5f, DB_SHIFT) */
    i32 *out_syn[2];
    i16 X[2 * 960];
    int c, i, j, LM, N;
    u32 seed;
    for (LM = 0; LM <= 3; LM++)
        if (120 << LM == frame_size) break;
    if (LM > 3 || pcm == NULL) return OC_BAD_ARG;
    N = 120 << LM;
    if (st->loss_count < 5 && start == 0) {
        celt_decode_lost_pitch(st, pcm, N);
        return frame_size;
    }
    for (c = 0; c < C; c++) out_syn[c] = st->syn[c] + OC_HIST;
    for (c = 0; c < C; c++)
        for (i = start; i < end; i++)
            st->bandE[c * NB + i] = OC_MAX(st->backgroundLogE[c * NB + i], (i16)(st->bandE[c * NB + i] - decay));
    memset(X, 0, sizeof(X));
    seed = st->rng;
    for (c = 0; c < C; c++)
        for (i = start; i < effEnd; i++) {
            const int boffs = N * c + (eb[i] << LM), blen = (eb[i + 1] - eb[i]) << LM;
            for (j = 0; j < blen; j++) {
                seed = lcg(seed);
                X[boffs + j] = (i16)((i32)seed >> 20);
            }
            renormalise(X + boffs, blen, 32767);
        }
    st->rng = seed;
    synthesis(st, X, out_syn, st->bandE, start, effEnd, C, C, 0, LM, 0, NULL);
    deemphasis(out_syn, pcm, N, C, st->deemph_mem);
    st->loss_count++;
    for (c = 0; c < C; c++) memmove(st->syn[c], st->syn[c] + N, (OC_HIST + OC_OVERLAP / 2) * sizeof(i32));
    return frame_size;
}
