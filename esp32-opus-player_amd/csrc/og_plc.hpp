// og_plc.hpp -- pitch-based concealment of lost CELT frames (RFC mode only; the reference has no concealment, Q8).
//
// What RFC 6716's decoder does for the first lost frames of a CELT-only stream (celt_decode_lost with loss_count < 5 and
// start == 0): find the pitch period of the last output, take the LPC residual of the last two periods, repeat it -- decaying
// by the energy ratio of its two halves per period -- through the LPC synthesis filter continued from the history, keep the
// result from getting louder than what it continues, and leave an overlap tail (pre-filtered against the post-filter, folded by
// the window) for the next decoded frame's transform to blend into.  A decoder's concealment is not normative and neither the
// reference nor this image holds libopus' source: the structure follows that decoder, the fixed-point detail is the oracle's
// (oracle/oc_celt.c celt_decode_lost_pitch -- 64-bit accumulators, the 1,024 samples of history the decoder keeps anyway, a
// pitch search of the last 304 samples against lags 100 .. 720), and this file computes exactly the same values.  PARITY-UNPINNED.
//
// One frame per wave like everything in RFC mode.  Lost CELT frames are rare and this mode is not the fast one: loops whose
// iterations are independent (correlations per lag, the autocorrelation per lag, the residual per sample) are spread over the
// lanes; the recurrences (Levinson-Durbin, the synthesis filter) run on lane 0.  Every sum is an exact integer, so the order
// in which lanes add their parts does not matter.  Scratch: the SILK synthesis' LDS object, idle in a CELT-only frame.
#pragma once
#include "og_celt.hpp"
#include "og_silk.hpp"

namespace og {

constexpr int PLC_LPC = 24, PLC_PMIN = 100, PLC_PMAX = 720, PLC_HIST = 1024;

// Scratch.  A lost CELT-only frame has the SILK kernels' LDS objects to itself -- with one exception: when the concealment smooths
// a switch from CELT to hybrid (RFC 6716 section 4.5), the new frame's SILK PCM is already waiting in SL().u.out.pcm for its CELT
// layer.  So the scratch keeps out of those 3,840 bytes: three pieces -- A: from behind that PCM to the end of the SILK synthesis'
// object (the up-sampler's buffers, where the concealment's OUTPUT is parked afterwards, and the decoder controls); B: the
// synthesis' look-back rows in front of it; C: the wave-uniform SILK parse's scratch (the head of the CELT spectrum X, dead until
// the concealed frame's PCM planes are written there -- after the search).
struct PlcA {
    i16 xs[PLC_LPC + 960 + OVERLAP];  // first the channel's history in 16 bits behind 24 zeros (x, 1,048 entries; during the search the
                                      // channels' mean), then -- x is dead by then -- the synthesis: 24 samples of history, the frame, its overlap
    i16 e[PLC_HIST];                  // the residual (during the search: the scaled copy)
    i32 etmp[OVERLAP];
    long long acc[PLC_LPC + 1];       // autocorrelation
    i16 lpc16[PLC_LPC];
    i32 scal[8];                      // [0] a search's result, [3] a shift
};
struct PlcB {
    long long part[2][OG_NLANES > 1 ? 64 : 1]; // per-lane partial sums / per-lane best scores
    i32 best_l[OG_NLANES > 1 ? 64 : 1];
};
struct PlcC {
    i16 lp[PLC_HIST / 2];             // the half-rate signal of the search
};
constexpr size_t PLC_A_AT = offsetof(SilkLds, u) + sizeof(i16) * 1920; // behind u.out.pcm
static_assert(offsetof(SilkLds, u) % 8 == 0 && PLC_A_AT % 8 == 0, "scratch alignment");
#if !defined(OG_SILK_LDS_FRAME) && !defined(OG_SILK_TIGHT) // (the synthesis kernels of the split path size SilkLds for themselves: no concealment runs from those layouts)
static_assert(PLC_A_AT + sizeof(PlcA) <= sizeof(SilkLds), "piece A: behind the SILK PCM");
static_assert(sizeof(PlcB) <= offsetof(SilkLds, u), "piece B: in front of the SILK PCM");
#endif
static_assert(sizeof(PlcC) <= sizeof(SilkWaveParseLds), "piece C: the wave-uniform SILK parse's object");
struct PlcLds {
    PlcA &a;
    PlcB &b;
    PlcC &c;
};
OG_DEV PlcA &PLA() { return *reinterpret_cast<PlcA *>(reinterpret_cast<u8 *>(&g_silk_lds) + PLC_A_AT); }
OG_DEV PlcB &PLB() { return *reinterpret_cast<PlcB *>(&g_silk_lds); }
OG_DEV PlcC &PLCc() { return *reinterpret_cast<PlcC *>(&PW()); } // (over the head of X: the search is over before a PCM plane is written there)

OG_DEV int plc_ilog64(unsigned long long x) { // bits needed: 0 for 0
    int n = 0;
    while (x) {
        n++;
        x >>= 1;
    }
    return n;
}
OG_DEV u32 plc_isqrt64(unsigned long long x) { // floor(sqrt(x)) for x < 2^62
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
OG_DEV i32 plc_sat16_64(long long x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : (i32)x); }
// sqrt(a / b) in Q15, at most 32767 (both scaled down together until b < 2^30)
OG_DEV i32 plc_ratio_q15(long long a, long long b) {
    const int sh = OG_MAX(0, plc_ilog64((unsigned long long)b) - 30);
    a >>= sh;
    b >>= sh;
    if (b <= 0 || a >= b) return 32767;
    const u32 r = plc_isqrt64((unsigned long long)(a << 30) / (unsigned long long)b);
    return (i32)OG_MIN((u32)32767, r);
}
// the history as 16-bit samples: sample i of the last 1024 of channel c (sig2word16 without de-emphasis)
OG_DEV i32 plc_hist16(const CeltState *st, int c, int pos, int i) { return sat16(pshr32(st->ring[c][(pos - PLC_HIST + i) & RING_MASK], 12)); }

// better (num / den larger, or equal and the lag shorter)?  -- the order a scan over ascending lags with a strict comparison gives
OG_DEV bool plc_better(long long num, long long den, int l, long long bnum, long long bden, int bl) {
    const long long a = num * bden, b = bnum * den;
    return a > b || (a == b && l < bl);
}

// one round of the search: lags l0 + k * step ... <= l1 over buf (scaled 16-bit samples), window [at, at + n) against [at - lag, ...);
// leaves the best lag in PLCS().scal[0] (or `fallback` when no lag correlates positively)
OG_DEV void plc_search(const i16 *buf, int at, int n, int l0, int l1, int fallback) {
    PlcA &L = PLA(); PlcB &LB = PLB(); PlcC &LC = PLCc(); (void)LB; (void)LC;
    long long bnum = -1, bden = 1;
    int bl = 0x7fffffff;
    for (int l = l0 + OG_LANE; l <= l1; l += OG_NLANES) {
        long long xc = 0, en = 1;
        for (int i = 0; i < n; i++) {
            xc += (i32)buf[at + i] * (i32)buf[at - l + i];
            en += (i32)buf[at - l + i] * (i32)buf[at - l + i];
        }
        if (xc <= 0) continue;
        const long long num = (xc * xc) >> 20;
        if (plc_better(num, en, l, bnum, bden, bl)) {
            bnum = num;
            bden = en;
            bl = l;
        }
    }
    LB.part[0][OG_LANE] = bnum;
    LB.part[1][OG_LANE] = bden;
    LB.best_l[OG_LANE] = bl;
    OG_SYNC();
    if (OG_LANE == 0) {
        bnum = -1, bden = 1, bl = 0x7fffffff;
        for (int t = 0; t < OG_NLANES; t++)
            if (LB.part[0][t] >= 0 && plc_better(LB.part[0][t], LB.part[1][t], LB.best_l[t], bnum, bden, bl)) {
                bnum = LB.part[0][t];
                bden = LB.part[1][t];
                bl = LB.best_l[t];
            }
        L.scal[0] = bnum >= 0 ? bl : fallback;
    }
    OG_SYNC();
}

// largest magnitude of buf[0 .. n) -> the shift that brings it below 2^9
OG_DEV int plc_shift_for(const i16 *buf, int n) {
    PlcA &L = PLA(); PlcB &LB = PLB(); PlcC &LC = PLCc(); (void)LB; (void)LC;
    int mx = 0;
    OG_FOR_LANES(i, n) {
        const int a = buf[i] < 0 ? -(int)buf[i] : (int)buf[i];
        mx = OG_MAX(mx, a);
    }
    LB.best_l[OG_LANE] = mx;
    OG_SYNC();
    if (OG_LANE == 0) {
        for (int t = 1; t < OG_NLANES; t++) mx = OG_MAX(mx, LB.best_l[t]);
        L.scal[3] = OG_MAX(0, plc_ilog64((unsigned long long)mx) - 9);
    }
    OG_SYNC();
    const int sh = L.scal[3];
    OG_SYNC();
    return sh;
}

// PLCS().x holds the signal to search (24 zeros, then 1024 samples): -> pitch period 100 .. 720
OG_DEV int plc_pitch_search() {
    PlcA &L = PLA(); PlcB &LB = PLB(); PlcC &LC = PLCc(); (void)LB; (void)LC;
    const i16 *v = L.xs + PLC_LPC;
    OG_SYNC();
    OG_FOR_LANES(i, PLC_HIST / 2) {
        const i32 a = i ? v[2 * i - 1] : 0, b = v[2 * i], c = v[2 * i + 1];
        LC.lp[i] = (i16)((a + 2 * b + c + 2) >> 2);
    }
    OG_SYNC();
    int sh = plc_shift_for(LC.lp, PLC_HIST / 2);
    OG_FOR_LANES(i, PLC_HIST / 2) LC.lp[i] = (i16)(LC.lp[i] >> sh);
    OG_SYNC();
    plc_search(LC.lp, 360, 152, PLC_PMIN / 2, PLC_PMAX / 2, PLC_PMIN / 2);
    const int best = L.scal[0];
    OG_SYNC();
    sh = plc_shift_for(v, PLC_HIST);
    OG_FOR_LANES(i, PLC_HIST) L.e[i] = (i16)(v[i] >> sh);
    OG_SYNC();
    plc_search(L.e, 720, 304, OG_MAX(PLC_PMIN, 2 * best - 1), OG_MIN(PLC_PMAX, 2 * best + 1), 2 * best);
    const int p = OG_MIN(PLC_PMAX, OG_MAX(PLC_PMIN, L.scal[0]));
    OG_SYNC();
    return p;
}

// order-24 LPC of PLCS().x[24 ..) in Q12 -> PLCS().lpc16
OG_DEV void plc_lpc() {
    PlcA &L = PLA(); PlcB &LB = PLB(); PlcC &LC = PLCc(); (void)LB; (void)LC;
    const i16 *v = L.xs + PLC_LPC;
    OG_SYNC();
    OG_FOR_LANES(k, PLC_LPC + 1) {
        long long a = 0;
        for (int i = k; i < PLC_HIST; i++) a += (i32)v[i] * (i32)v[i - k];
        L.acc[k] = a;
    }
    OG_SYNC();
    if (OG_LANE == 0) {
        i32 ac[PLC_LPC + 1], lpc[PLC_LPC];
        const int sh = OG_MAX(0, plc_ilog64((unsigned long long)L.acc[0]) - 29);
        for (int k = 0; k <= PLC_LPC; k++) ac[k] = (i32)(L.acc[k] >> sh);
        ac[0] += ac[0] >> 13;
        for (int k = 1; k <= PLC_LPC; k++) ac[k] -= (i32)(((long long)ac[k] * (2 * k * k)) >> 15);
        for (int i = 0; i < PLC_LPC; i++) lpc[i] = 0;
        i32 err = ac[0];
        if (ac[0] > 0)
            for (int i = 0; i < PLC_LPC; i++) {
                long long rr = 0;
                for (int j = 0; j < i; j++) rr += (i32)(((long long)lpc[j] * ac[i - j]) >> 31); // lpc in Q25
                rr += ac[i + 1] >> 6;
                long long q = -(rr * 64 * 33554432) / err; // reflection coefficient, Q25
                if (q > (1 << 25) - 1) q = (1 << 25) - 1;
                if (q < -(1 << 25) + 1) q = -(1 << 25) + 1;
                const i32 r = (i32)q;
                lpc[i] = r;
                for (int j = 0; j < (i + 1) >> 1; j++) {
                    const i32 t1 = lpc[j], t2 = lpc[i - 1 - j];
                    lpc[j] = t1 + (i32)(((long long)r * t2) >> 25);
                    lpc[i - 1 - j] = t2 + (i32)(((long long)r * t1) >> 25);
                }
                err -= (i32)(((long long)(i32)(((long long)r * r) >> 25) * err) >> 25);
                if (err < (ac[0] >> 10)) break;
            }
        for (int i = 0; i < PLC_LPC; i++) L.lpc16[i] = (i16)sat16(pshr32(lpc[i], 13));
    }
    OG_SYNC();
}

// sum over lanes of two 64-bit parts -> (a, b), on every lane
OG_DEV void plc_sum2(long long &a, long long &b) {
    PlcA &L = PLA(); PlcB &LB = PLB(); PlcC &LC = PLCc(); (void)LB; (void)LC;
    OG_SYNC();
    LB.part[0][OG_LANE] = a;
    LB.part[1][OG_LANE] = b;
    OG_SYNC();
    a = b = 0;
    for (int t = 0; t < OG_NLANES; t++) {
        a += LB.part[0][t];
        b += LB.part[1][t];
    }
    OG_SYNC();
}

// The concealed frame of every decoder channel: N samples into the history ring, the overlap tail, the de-emphasis memory, the
// PCM planes (as after celt_decode_frame with C == CC).  Returns frame_size.
OG_DEV int celt_decode_lost_pitch(CeltState *st, LossState *loss, int N, int CC) {
    PlcA &L = PLA(); PlcB &LB = PLB(); PlcC &LC = PLCc(); (void)LB; (void)LC;
    const int len = N + OVERLAP, cmp = OG_MIN(len, PLC_HIST), pos = st->ring_pos;
    const int first = loss->celt_loss_count == 0;
    i32 *const SY = syn_buf();
    OG_SYNC();
    OG_FOR_LANES(i, PLC_LPC) L.xs[i] = 0;
    if (first) { // the pitch period: searched once, on the mean of the channels
        OG_FOR_LANES(i, PLC_HIST) L.xs[PLC_LPC + i] = CC == 2 ? (i16)((plc_hist16(st, 0, pos, i) + plc_hist16(st, 1, pos, i)) >> 1) : (i16)plc_hist16(st, 0, pos, i);
        OG_SYNC();
        const int p = plc_pitch_search();
        if (OG_LANE == 0) loss->plc_pitch = p;
        OG_SYNC();
    }
    const int pitch = OG_UNI(loss->plc_pitch), exc_len = OG_MIN(2 * pitch, 1000), half = exc_len / 2;
    for (int c = 0; c < CC; c++) {
        OG_SYNC();
        OG_FOR_LANES(i, PLC_HIST) L.xs[PLC_LPC + i] = (i16)plc_hist16(st, c, pos, i);
        OG_SYNC();
        const i16 *x = L.xs + PLC_LPC; // x[-24 .. 1024)
        if (first) {
            plc_lpc();
            OG_FOR_LANES(i, PLC_LPC) loss->plc_lpc[c][i] = L.lpc16[i];
        } else
            OG_FOR_LANES(i, PLC_LPC) L.lpc16[i] = loss->plc_lpc[c][i];
        OG_SYNC();
        // the residual of the last exc_len samples
        OG_FOR_LANES(k, exc_len) {
            const int i = PLC_HIST - exc_len + k;
            long long a = 0;
            for (int j = 0; j < PLC_LPC; j++) a += (i32)L.lpc16[j] * (i32)x[i - 1 - j];
            L.e[i] = (i16)plc_sat16_64((long long)x[i] + ((a + 2048) >> 12));
        }
        OG_SYNC();
        // how much it decays from its first half to its second
        long long E1 = 0, E2 = 0;
        OG_FOR_LANES(i, half) {
            const i32 a = L.e[PLC_HIST - half + i], b = L.e[PLC_HIST - 2 * half + i];
            E1 += a * a;
            E2 += b * b;
        }
        plc_sum2(E1, E2);
        E1 += 1;
        E2 += 1;
        const i32 decay = plc_ratio_q15(OG_MIN(E1, E2), E2);
        // (the energy of what the concealment continues, while x is still there: the synthesis is written over it)
        long long S1 = 0, S1b = 0;
        OG_FOR_LANES(i, cmp) S1 += (i32)x[PLC_HIST - cmp + i] * (i32)x[PLC_HIST - cmp + i];
        plc_sum2(S1, S1b);
        // the period before the end, again and again, a little quieter each time, through the synthesis filter (a recurrence: lane 0)
        i32 keep = 0; // the filter's memory: the last 24 samples of x move to the front of the same array
        if (OG_LANE < PLC_LPC) keep = x[PLC_HIST - PLC_LPC + OG_LANE];
#ifdef OG_HOST_EMUL
        i16 keep_all[PLC_LPC]; // (one lane stands for all)
        for (int i = 0; i < PLC_LPC; i++) keep_all[i] = x[PLC_HIST - PLC_LPC + i];
#endif
        OG_SYNC();
        i16 *const sy = L.xs;
#ifdef OG_HOST_EMUL
        for (int i = 0; i < PLC_LPC; i++) sy[i] = keep_all[i];
#else
        if (OG_LANE < PLC_LPC) sy[OG_LANE] = (i16)keep;
#endif
        OG_SYNC();
        if (OG_LANE == 0) {
            const i32 fade = first ? 32767 : 26214; // Q15: 1, 0.8
            i32 att = mul16_q15(fade, decay);
            for (int i = 0, j = 0; i < len; i++, j++) {
                if (j >= pitch) {
                    j -= pitch;
                    att = mul16_q15(att, decay);
                }
                long long a = (long long)mul16_q15(att, L.e[PLC_HIST - pitch + j]) * 4096;
                for (int k = 0; k < PLC_LPC; k++) a -= (i32)L.lpc16[k] * (i32)sy[PLC_LPC + i - 1 - k];
                sy[PLC_LPC + i] = (i16)plc_sat16_64((a + 2048) >> 12);
            }
        }
        OG_SYNC();
        // not louder than what it continues
        long long S2 = 0, S2b = 0;
        OG_FOR_LANES(i, cmp) S2 += (i32)sy[PLC_LPC + i] * (i32)sy[PLC_LPC + i];
        plc_sum2(S2, S2b);
        if (!(S1 > (S2 >> 2))) {
            OG_FOR_LANES(i, len) sy[PLC_LPC + i] = 0;
        } else if (S1 < S2) {
            const i32 ratio = plc_ratio_q15((S1 >> 1) + 1, S2 + 1);
            OG_FOR_LANES(i, len) {
                const i32 g = i < OVERLAP ? (i32)(i16)(32767 - mul16_q15(rom_win120[i], 32767 - ratio)) : ratio;
                sy[PLC_LPC + i] = (i16)mul16_q15(g, sy[PLC_LPC + i]);
            }
        }
        OG_SYNC();
        OG_FOR_LANES(i, len) SY[i] = (i32)sy[PLC_LPC + i] * 4096;
        OG_SYNC();
        // the overlap for the next frame: pre-filtered against the post-filter that frame will run over it, folded by the window
        {
            static constexpr i16 gains[3][3] = {{10048, 7112, 4248}, {15200, 8784, 0}, {26208, 3280, 0}};
            const int T = OG_MAX((int)OG_UNI(st->pf_period), 15), tap = OG_UNI(st->pf_tapset);
            const i32 pfg = OG_UNI(st->pf_gain);
            const i32 g = (i32)(i16)-pfg;
            const i32 g0 = (i32)(i16)mul16_p15(g, gains[tap][0]), g1 = (i32)(i16)mul16_p15(g, gains[tap][1]), g2 = (i32)(i16)mul16_p15(g, gains[tap][2]);
            OG_FOR_LANES(i, OVERLAP) {
                i32 y = SY[N + i];
                if (pfg != 0) {
                    const int at = N + i - T;
                    y = clampsym(y + mul16x32_q15(g0, syn_at(st, c, at)) + mul16x32_q15(g1, syn_at(st, c, at + 1) + syn_at(st, c, at - 1)) +
                                     mul16x32_q15(g2, syn_at(st, c, at + 2) + syn_at(st, c, at - 2)), SIG_SAT);
                }
                L.etmp[i] = y;
            }
            OG_SYNC();
            OG_FOR_LANES(i, OVERLAP / 2)
                st->tail[c][i] = mul16x32_q15(rom_win120[i], L.etmp[OVERLAP - 1 - i]) + mul16x32_q15(rom_win120[OVERLAP - 1 - i], L.etmp[i]);
        }
        // de-emphasis (a rounding IIR, serial) into the channel's PCM plane; the frame into the history ring
        OG_SYNC();
        if (OG_LANE == 0) {
            const int plane = pcm_plane(c, CC, CC);
            i32 m = st->deemph[c];
            for (int j = 0; j < N; j++) {
                const i32 tmp = SY[j] + m;
                m = mul16x32_q15(27853, tmp);
                S.v[plane + j] = (i16)sat16(pshr32(tmp, 12));
            }
            st->deemph[c] = m;
        }
        OG_SYNC();
        OG_FOR_LANES(i, N) st->ring[c][(pos + i) & RING_MASK] = SY[i];
        OG_SYNC();
    }
    if (OG_LANE == 0) {
        st->ring_pos = (pos + N) & RING_MASK;
        loss->celt_loss_count += 1;
    }
    OG_SYNC();
    return N;
}

// RFC 6716's celt_decode_lost: the pitch-based branch for the first lost frames of a stream that codes from band 0, else noise
OG_DEV int celt_conceal(CeltState *st, LossState *loss, int frame_size, int CC, int start, int end) {
    if (OG_UNI(loss->celt_loss_count) < 5 && start == 0 &&
        (frame_size == 120 || frame_size == 240 || frame_size == 480 || frame_size == 960))
        return celt_decode_lost_pitch(st, loss, frame_size, CC);
    return celt_decode_lost(st, loss, frame_size, CC, start, end);
}

} // namespace og
