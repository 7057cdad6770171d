/*
 * oc_celt_math.c -- CPU ORACLE (test infrastructure): CELT scalar approximations, PVQ index
 * decode, FFT / IMDCT and the pitch comb filter.  Restates the reference's
 * src/celt.cpp:684-899 (rotation, comb filter), :2545-2620 (cwrsi), :2794-3038 (FFT),
 * :3086-3296 (isqrt, rsqrt, sqrt, cos, rcp, IMDCT).
 */
#include "oc_celt_priv.h"

/* celt.cpp:3086 */
u32 oc_isqrt32(u32 val) {
    u32 g = 0, b;
    int bshift = (ilog32(val) - 1) >> 1;
    b = 1u << bshift;
    do {
        u32 t = ((g << 1) + b) << bshift;
        if (t <= val) {
            g += b;
            val -= t;
        }
        b >>= 1;
        bshift--;
    } while (bshift >= 0);
    return g;
}

/* celt.cpp:3109 -- Q16 in [0.25,1) -> Q14 1/sqrt */
i16 oc_rsqrt_norm(i32 x) {
    i16 n = (i16)(x - 32768);
    i16 r = add16(23557, m16_q15(n, add16(-13490, m16_q15(n, 6713))));
    i16 r2 = (i16)m16_q15(r, r);
    i16 y = shl16(sub16(add16(m16_q15(r2, n), r2), 16384), 1);
    return add16(r, m16_q15(r, m16_q15(y, sub16(m16_q15(y, 12288), 16384))));
}

/* celt.cpp:3131 */
i32 oc_sqrt(i32 x) {
    static const i16 C[5] = {23175, 11561, -3011, 1699, -664};
    int k;
    i16 n;
    i32 rt;
    if (x == 0) return 0;
    if (x >= 1073741824) return 32767;
    k = (ilog2p(x) >> 1) - 7;
    x = vshr32(x, 2 * k);
    n = (i16)(x - 32768);
    rt = add16(C[0], m16_q15(n, add16(C[1], m16_q15(n, add16(C[2], m16_q15(n, add16(C[3], m16_q15(n, C[4]))))))));
    return vshr32(rt, 7 - k);
}

/* celt.cpp:3151 */
static i16 cos_pi_2(i16 x) {
    i16 x2 = (i16)m16_p15(x, x);
    i32 v = (i32)sub16(32767, x2) + m16_p15(x2, -7651 + m16_p15(x2, 8277 + m16_p15(-626, x2)));
    return add16(1, OC_MIN(32766, v));
}

/* celt.cpp:3161 */
i16 oc_cos_norm(i32 x) {
    x &= 0x1ffff;
    if (x > (1 << 16)) x = (1 << 17) - x;
    if (x & 0x7fff) {
        if (x < (1 << 15)) return cos_pi_2((i16)x);
        return (i16)(-cos_pi_2((i16)(65536 - x)));
    }
    if (x & 0xffff) return 0;
    if (x & 0x1ffff) return -32767;
    return 32767;
}

/* celt.cpp:3181 -- Q15 in, Q16 out */
i32 oc_rcp(i32 x) {
    int i = ilog2p(x);
    i16 n = (i16)(vshr32(x, i - 15) - 32768);
    i16 r = add16(30840, m16_q15(-15420, n));
    r = (i16)sub16(r, m16_q15(r, add16(m16_q15(r, n), add16(r, -32768))));
    r = (i16)sub16(r, add16(1, m16_q15(r, add16(m16_q15(r, n), add16(r, -32768)))));
    return vshr32((i32)r, i - 16);
}

/* celt.h:494 */
i32 oc_exp2_frac(i32 x) {
    i16 frac = shl16(x, 4);
    return add16(16383, m16_q15(frac, add16(22804, m16_q15(frac, add16(14819, m16_q15(10204, frac))))));
}

/* celt.h:501 -- Q10 in, Q16 out */
i32 oc_exp2(i32 x_in) {
    i16 x = (i16)x_in;
    i32 integer = x >> 10;
    i16 frac;
    if (integer > 14) return 0x7f000000;
    if (integer < -15) return 0;
    frac = (i16)oc_exp2_frac((i16)(x - shl16(integer, 10)));
    return vshr32((i32)frac, -integer - 2);
}

/* ------------------------------------------------------------------------------------------
 * PVQ codeword index -> pulse vector (celt.cpp:2545).  U(n,k) comes from the dense generated
 * table; U is symmetric so U(min,max) is always addressable.
 */
static inline u32 pvq_u(int a, int b) {
    int lo = a < b ? a : b, hi = a < b ? b : a;
    return rom_pvq_u[lo * ROM_PVQ_COLS + hi];
}
u32 oc_pvq_v(int n, int k) { return pvq_u(n, k) + pvq_u(n, k + 1); } /* CELT_PVQ_V celt.cpp:660 */

i32 oc_cwrsi(int n, int k, u32 i, i32 *y) {
    u32 p, q;
    int s, k0;
    i16 val;
    i32 yy = 0;
    while (n > 2) {
        if (k >= n) { /* many pulses: walk row n along k */
            p = pvq_u(n, k + 1);
            s = -(i >= p);
            i -= p & s;
            k0 = k;
            q = pvq_u(n, n);
            if (q > i) {
                k = n;
                do p = pvq_u(--k, n);
                while (p > i);
            } else {
                for (p = pvq_u(n, k); p > i; p = pvq_u(n, k)) k--;
            }
            i -= p;
            val = (i16)((k0 - k + s) ^ s);
            *y++ = val;
            yy += m16(val, val);
        } else { /* many dimensions */
            p = pvq_u(k, n);
            q = pvq_u(k + 1, n);
            if (p <= i && i < q) {
                i -= p;
                *y++ = 0;
            } else {
                s = -(i >= q);
                i -= q & s;
                k0 = k;
                do p = pvq_u(--k, n);
                while (p > i);
                i -= p;
                val = (i16)((k0 - k + s) ^ s);
                *y++ = val;
                yy += m16(val, val);
            }
        }
        n--;
    }
    /* n == 2 */
    p = 2 * k + 1;
    s = -(i >= p);
    i -= p & s;
    k0 = k;
    k = (i + 1) >> 1;
    if (k) i -= 2 * k - 1;
    val = (i16)((k0 - k + s) ^ s);
    *y++ = val;
    yy += m16(val, val);
    /* n == 1 */
    s = -(int)i;
    val = (i16)((k + s) ^ s);
    *y = val;
    yy += m16(val, val);
    return yy;
}

/* ------------------------------------------------------------------------------------------
 * Spreading rotation (celt.cpp:684, :707)
 */
static void rotate1(i16 *X, int len, int stride, i16 c, i16 s) {
    int i;
    i16 ms = (i16)(-s);
    i16 *p = X;
    for (i = 0; i < len - stride; i++) {
        i16 x1 = p[0], x2 = p[stride];
        p[stride] = (i16)pshr32(m16(c, x2) + m16(s, x1), 15);
        *p++ = (i16)pshr32(m16(c, x1) + m16(ms, x2), 15);
    }
    p = &X[len - 2 * stride - 1];
    for (i = len - 2 * stride - 1; i >= 0; i--) {
        i16 x1 = p[0], x2 = p[stride];
        p[stride] = (i16)pshr32(m16(c, x2) + m16(s, x1), 15);
        *p-- = (i16)pshr32(m16(c, x1) + m16(ms, x2), 15);
    }
}

void oc_exp_rotation(i16 *X, int len, int dir, int stride, int K, int spread) {
    static const int factor_tab[3] = {15, 10, 5};
    int i, factor, stride2 = 0;
    i16 c, s, gain, theta;
    if (2 * K >= len || spread == 0) return;
    factor = factor_tab[spread - 1];
    gain = (i16)m32_q31(m16(32767, len), oc_rcp(len + factor * K)); /* celt_div celt.h:367 */
    theta = (i16)(m16_q15(gain, gain) >> 1);
    c = oc_cos_norm(theta);
    s = oc_cos_norm(sub16(32767, theta));
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    len = (int)((u32)len / (u32)stride);
    for (i = 0; i < stride; i++) {
        if (dir < 0) {
            if (stride2) rotate1(X + i * len, len, stride2, s, c);
            rotate1(X + i * len, len, 1, c, s);
        } else {
            rotate1(X + i * len, len, 1, c, (i16)(-s));
            if (stride2) rotate1(X + i * len, len, stride2, s, (i16)(-c));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Mixed-radix FFT without scaling (celt.cpp:2794-3038).  Complex data is interleaved
 * (re, im) int32; twiddles are rom_fft_tw (re, im) int16 of the 480-point circle.
 */
typedef struct { i32 r, i; } cpx;
#define SMUL(a, b) m16x32_q15((b), (a)) /* S_MUL celt.h:192 */

static inline cpx cmul_tw(cpx a, int tw) { /* C_MUL celt.h:193 */
    i16 wr = rom_fft_tw[2 * tw], wi = rom_fft_tw[2 * tw + 1];
    cpx m;
    m.r = subw(SMUL(a.r, wr), SMUL(a.i, wi));
    m.i = addw(SMUL(a.r, wi), SMUL(a.i, wr));
    return m;
}
static inline cpx cadd(cpx a, cpx b) { cpx r = {addw(a.r, b.r), addw(a.i, b.i)}; return r; }
static inline cpx csub(cpx a, cpx b) { cpx r = {subw(a.r, b.r), subw(a.i, b.i)}; return r; }

/* celt.cpp:2794 (m is always 4) */
static void bfly2(cpx *F, int N) {
    const i16 tw = 23170; /* QCONST16(0.7071067812,15) */
    int i;
    for (i = 0; i < N; i++) {
        cpx *F2 = F + 4, t;
        t = F2[0];
        F2[0] = csub(F[0], t);
        F[0] = cadd(F[0], t);
        t.r = SMUL(addw(F2[1].r, F2[1].i), tw);
        t.i = SMUL(subw(F2[1].i, F2[1].r), tw);
        F2[1] = csub(F[1], t);
        F[1] = cadd(F[1], t);
        t.r = F2[2].i;
        t.i = -F2[2].r;
        F2[2] = csub(F[2], t);
        F[2] = cadd(F[2], t);
        t.r = SMUL(subw(F2[3].i, F2[3].r), tw);
        t.i = SMUL(negw(addw(F2[3].i, F2[3].r)), tw);
        F2[3] = csub(F[3], t);
        F[3] = cadd(F[3], t);
        F += 8;
    }
}

/* celt.cpp:2830 */
static void bfly4(cpx *Fbeg, int fstride, int m, int N, int mm) {
    int i, j;
    if (m == 1) {
        cpx *F = Fbeg;
        for (i = 0; i < N; i++) {
            cpx s0 = csub(F[0], F[2]), s1;
            F[0] = cadd(F[0], F[2]);
            s1 = cadd(F[1], F[3]);
            F[2] = csub(F[0], s1);
            F[0] = cadd(F[0], s1);
            s1 = csub(F[1], F[3]);
            F[1].r = addw(s0.r, s1.i);
            F[1].i = subw(s0.i, s1.r);
            F[3].r = subw(s0.r, s1.i);
            F[3].i = addw(s0.i, s1.r);
            F += 4;
        }
        return;
    }
    for (i = 0; i < N; i++) {
        cpx *F = Fbeg + i * mm;
        int t1 = 0, t2 = 0, t3 = 0;
        for (j = 0; j < m; j++) {
            cpx a = cmul_tw(F[m], t1), b = cmul_tw(F[2 * m], t2), c = cmul_tw(F[3 * m], t3);
            cpx s5 = csub(F[0], b), s3, s4;
            F[0] = cadd(F[0], b);
            s3 = cadd(a, c);
            s4 = csub(a, c);
            F[2 * m] = csub(F[0], s3);
            t1 += fstride;
            t2 += fstride * 2;
            t3 += fstride * 3;
            F[0] = cadd(F[0], s3);
            F[m].r = addw(s5.r, s4.i);
            F[m].i = subw(s5.i, s4.r);
            F[3 * m].r = subw(s5.r, s4.i);
            F[3 * m].i = addw(s5.i, s4.r);
            ++F;
        }
    }
}

/* celt.cpp:2887 */
static void bfly3(cpx *Fbeg, int fstride, int m, int N, int mm) {
    const i16 epi3_i = -28378;
    int i, k;
    for (i = 0; i < N; i++) {
        cpx *F = Fbeg + i * mm;
        int t1 = 0, t2 = 0;
        for (k = 0; k < m; k++) {
            cpx s1 = cmul_tw(F[m], t1), s2 = cmul_tw(F[2 * m], t2);
            cpx s3 = cadd(s1, s2), s0 = csub(s1, s2);
            t1 += fstride;
            t2 += fstride * 2;
            F[m].r = subw(F[0].r, s3.r >> 1);
            F[m].i = subw(F[0].i, s3.i >> 1);
            s0.r = SMUL(s0.r, epi3_i);
            s0.i = SMUL(s0.i, epi3_i);
            F[0] = cadd(F[0], s3);
            F[2 * m].r = addw(F[m].r, s0.i);
            F[2 * m].i = subw(F[m].i, s0.r);
            F[m].r = subw(F[m].r, s0.i);
            F[m].i = addw(F[m].i, s0.r);
            ++F;
        }
    }
}

/* celt.cpp:2930 */
static void bfly5(cpx *Fbeg, int fstride, int m, int N, int mm) {
    const i16 ya_r = 10126, ya_i = -31164, yb_r = -26510, yb_i = -19261;
    int i, u;
    for (i = 0; i < N; i++) {
        cpx *F0 = Fbeg + i * mm, *F1 = F0 + m, *F2 = F0 + 2 * m, *F3 = F0 + 3 * m, *F4 = F0 + 4 * m;
        for (u = 0; u < m; ++u) {
            cpx s0 = *F0;
            cpx s1 = cmul_tw(*F1, u * fstride), s2 = cmul_tw(*F2, 2 * u * fstride);
            cpx s3 = cmul_tw(*F3, 3 * u * fstride), s4 = cmul_tw(*F4, 4 * u * fstride);
            cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
            cpx s5, s6, s11, s12;
            F0->r = addw(F0->r, addw(s7.r, s8.r));
            F0->i = addw(F0->i, addw(s7.i, s8.i));
            s5.r = addw(s0.r, addw(SMUL(s7.r, ya_r), SMUL(s8.r, yb_r)));
            s5.i = addw(s0.i, addw(SMUL(s7.i, ya_r), SMUL(s8.i, yb_r)));
            s6.r = addw(SMUL(s10.i, ya_i), SMUL(s9.i, yb_i));
            s6.i = negw(addw(SMUL(s10.r, ya_i), SMUL(s9.r, yb_i)));
            *F1 = csub(s5, s6);
            *F4 = cadd(s5, s6);
            s11.r = addw(s0.r, addw(SMUL(s7.r, yb_r), SMUL(s8.r, ya_r)));
            s11.i = addw(s0.i, addw(SMUL(s7.i, yb_r), SMUL(s8.i, ya_r)));
            s12.r = subw(SMUL(s9.i, ya_i), SMUL(s10.i, yb_i));
            s12.i = subw(SMUL(s10.r, yb_i), SMUL(s9.r, ya_i));
            *F2 = cadd(s11, s12);
            *F3 = csub(s11, s12);
            ++F0; ++F1; ++F2; ++F3; ++F4;
        }
    }
}

/* factor schedules of the 480/240/120/60-point transforms (celt.cpp:589-626) */
static const i16 fft_factors[4][10] = {
    {5, 96, 3, 32, 4, 8, 2, 4, 4, 1},
    {5, 48, 3, 16, 4, 4, 4, 1, 0, 0},
    {5, 24, 3, 8, 2, 4, 4, 1, 0, 0},
    {5, 12, 3, 4, 4, 1, 0, 0, 0, 0},
};

/* celt.cpp:2997; `shift` selects the transform: 0 -> 480, 1 -> 240, 2 -> 120, 3 -> 60 points */
void oc_fft(int shift, i32 *data) {
    cpx *fout = (cpx *)data;
    const i16 *fac = fft_factors[shift];
    int fstride[8], L = 0, m, m2, p, i;
    fstride[0] = 1;
    do {
        p = fac[2 * L];
        m = fac[2 * L + 1];
        fstride[L + 1] = fstride[L] * p;
        L++;
    } while (m != 1);
    m = fac[2 * L - 1];
    for (i = L - 1; i >= 0; i--) {
        m2 = i != 0 ? fac[2 * i - 1] : 1;
        switch (fac[2 * i]) {
            case 2: bfly2(fout, fstride[i]); break;
            case 4: bfly4(fout, fstride[i] << shift, m, fstride[i], m2); break;
            case 3: bfly3(fout, fstride[i] << shift, m, fstride[i], m2); break;
            case 5: bfly5(fout, fstride[i] << shift, m, fstride[i], m2); break;
        }
        m = m2;
    }
}

/* ------------------------------------------------------------------------------------------
 * Fixed-point inverse MDCT with TDAC windowing (celt.cpp:3204).  N = 1920 >> shift.
 */
void oc_imdct(const i32 *in, i32 *out, int overlap, int shift, int stride) {
    static const i16 *const bitrev_tab[4] = {rom_bitrev480, rom_bitrev240, rom_bitrev120, rom_bitrev60};
    const i16 *trig = rom_mdct_trig;
    int N = 1920, N2, N4, i;
    for (i = 0; i < shift; i++) {
        N >>= 1;
        trig += N;
    }
    N2 = N >> 1;
    N4 = N >> 2;
    { /* pre-rotation, written in digit-reversed order */
        const i32 *xp1 = in, *xp2 = in + stride * (N2 - 1);
        i32 *yp = out + (overlap >> 1);
        const i16 *br = bitrev_tab[shift];
        for (i = 0; i < N4; i++) {
            int rev = br[i];
            i32 yr = addw(SMUL(*xp2, trig[i]), SMUL(*xp1, trig[N4 + i]));
            i32 yi = subw(SMUL(*xp1, trig[i]), SMUL(*xp2, trig[N4 + i]));
            yp[2 * rev + 1] = yr;
            yp[2 * rev] = yi;
            xp1 += 2 * stride;
            xp2 -= 2 * stride;
        }
    }
    oc_fft(shift, out + (overlap >> 1));
    { /* post-rotation from both ends */
        i32 *yp0 = out + (overlap >> 1), *yp1 = out + (overlap >> 1) + N2 - 2;
        for (i = 0; i < (N4 + 1) >> 1; i++) {
            i32 re = yp0[1], im = yp0[0], yr, yi;
            i16 t0 = trig[i], t1 = trig[N4 + i];
            yr = addw(SMUL(re, t0), SMUL(im, t1));
            yi = subw(SMUL(re, t1), SMUL(im, t0));
            re = yp1[1];
            im = yp1[0];
            yp0[0] = yr;
            yp1[1] = yi;
            t0 = trig[N4 - i - 1];
            t1 = trig[N2 - i - 1];
            yr = addw(SMUL(re, t0), SMUL(im, t1));
            yi = subw(SMUL(re, t1), SMUL(im, t0));
            yp1[0] = yr;
            yp0[1] = yi;
            yp0 += 2;
            yp1 -= 2;
        }
    }
    { /* TDAC mirror */
        i32 *xp1 = out + overlap - 1, *yp1 = out;
        const i16 *wp1 = rom_win120, *wp2 = rom_win120 + overlap - 1;
        for (i = 0; i < overlap / 2; i++) {
            i32 x1 = *xp1, x2 = *yp1;
            *yp1++ = subw(m16x32_q15(*wp2, x2), m16x32_q15(*wp1, x1));
            *xp1-- = addw(m16x32_q15(*wp1, x2), m16x32_q15(*wp2, x1));
            wp1++;
            wp2--;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Pitch comb post-filter (celt.cpp:830, :848)
 */
static void comb_const(i32 *y, i32 *x, int T, int N, i16 g10, i16 g11, i16 g12) {
    i32 x4 = x[-T - 2], x3 = x[-T - 1], x2 = x[-T], x1 = x[-T + 1], x0;
    int i;
    for (i = 0; i < N; i++) {
        x0 = x[i - T + 2];
        y[i] = x[i] + m16x32_q15(g10, x2) + m16x32_q15(g11, x1 + x3) + m16x32_q15(g12, x0 + x4);
        y[i] = satsym(y[i], OC_SIG_SAT);
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
}

void oc_comb_filter(i32 *y, i32 *x, int T0, int T1, int N, i16 g0, i16 g1, int tap0, int tap1) {
    /* QCONST16 of {0.3066406250, 0.2170410156, 0.1296386719}, {0.4638671875, 0.2680664062, 0},
       {0.7998046875, 0.1000976562, 0} in Q15 (celt.cpp:854) */
    static const i16 gains[3][3] = {{10048, 7112, 4248}, {15200, 8784, 0}, {26208, 3280, 0}};
    int i, overlap = OC_OVERLAP;
    i16 g00, g01, g02, g10, g11, g12;
    i32 x0, x1, x2, x3, x4;
    if (g0 == 0 && g1 == 0) {
        if (x != y) memmove(y, x, N * sizeof(*y));
        return;
    }
    T0 = OC_MAX(T0, 15);
    T1 = OC_MAX(T1, 15);
    g00 = (i16)m16_p15(g0, gains[tap0][0]);
    g01 = (i16)m16_p15(g0, gains[tap0][1]);
    g02 = (i16)m16_p15(g0, gains[tap0][2]);
    g10 = (i16)m16_p15(g1, gains[tap1][0]);
    g11 = (i16)m16_p15(g1, gains[tap1][1]);
    g12 = (i16)m16_p15(g1, gains[tap1][2]);
    x1 = x[-T1 + 1];
    x2 = x[-T1];
    x3 = x[-T1 - 1];
    x4 = x[-T1 - 2];
    if (g0 == g1 && T0 == T1 && tap0 == tap1) overlap = 0;
    for (i = 0; i < overlap; i++) {
        i16 f;
        x0 = x[i - T1 + 2];
        f = (i16)m16_q15(rom_win120[i], rom_win120[i]);
        y[i] = x[i] + m16x32_q15(m16_q15(32767 - f, g00), x[i - T0]) +
               m16x32_q15(m16_q15(32767 - f, g01), x[i - T0 + 1] + x[i - T0 - 1]) +
               m16x32_q15(m16_q15(32767 - f, g02), x[i - T0 + 2] + x[i - T0 - 2]) +
               m16x32_q15(m16_q15(f, g10), x2) + m16x32_q15(m16_q15(f, g11), x1 + x3) +
               m16x32_q15(m16_q15(f, g12), x0 + x4);
        y[i] = satsym(y[i], OC_SIG_SAT);
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
    if (g1 == 0) {
        if (x != y) memmove(y + overlap, x + overlap, (N - overlap) * sizeof(*y));
        return;
    }
    comb_const(y + i, x + i, T1, N - i, g10, g11, g12);
}
