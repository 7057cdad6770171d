// og_silk_loss.hpp -- SILK packet-loss concealment, comfort noise and the smoothing between concealed and decoded frames
// (RFC mode only).
//
// Behaviour reproduced: silk_PLC / silk_PLC_update / silk_PLC_conceal / silk_PLC_glue_frames (src/silk.cpp:2862-3185),
// silk_CNG (:1305-1432), silk_sum_sqr_shift (:3839), silk_bwexpander (:576), silk_SQRT_APPROX (src/silk.h:888).  The
// reference carries this code but never reaches its loss branches (lostFlag == 0 always, Q8); the oracle's RFC mode does
// (oracle/oc_silk.c), and this file is bit-exact to that.
//
// Mapping: all of it is short and serial in time (IIR recurrences, running seeds), and a lost frame is the rare case: ONE LANE
// PER CHANNEL runs it, over the same LDS buffers the regular synthesis uses (included from og_silk.hpp, inside namespace og).
#pragma once

OG_DEV void silk_loss_chan_init(SilkLossChannel *lc) { // the loss half of silk_init_decoder (silk.cpp:2192-2204)
    // all zero: plc_fs_kHz == cng_fs_kHz == 0 makes the first frame run silk_PLC_Reset / silk_CNG_Reset with the real rate
    u32 *w = reinterpret_cast<u32 *>(lc);
    OG_FOR_LANES(i, (int)(sizeof(SilkLossChannel) / 4)) w[i] = 0;
    OG_SYNC();
}

template <class AQ>
OG_DEV void silk_bwexpander16(AQ ar, int d, i32 chirp_Q16) { // silk.cpp:576
    const i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    for (int i = 0; i < d - 1; i++) {
        ar[i] = (i16)rshift_round(chirp_Q16 * (i32)ar[i], 16);
        chirp_Q16 += rshift_round(chirp_Q16 * chirp_minus_one_Q16, 16);
    }
    ar[d - 1] = (i16)rshift_round(chirp_Q16 * (i32)ar[d - 1], 16);
}

OG_DEV void silk_sum_sqr_shift(i32 &energy, i32 &shift, const i16 *x, int len) { // silk.cpp:3839
    int shft = 31 - clz32(len);
    u32 nrg = (u32)len;
    int i;
    for (i = 0; i < len - 1; i += 2) nrg += ((u32)smulbb(x[i], x[i]) + (u32)smulbb(x[i + 1], x[i + 1])) >> shft;
    if (i < len) nrg += (u32)smulbb(x[i], x[i]) >> shft;
    shft = OG_MAX(0, shft + 3 - clz32((i32)nrg));
    nrg = 0;
    for (i = 0; i < len - 1; i += 2) nrg += ((u32)smulbb(x[i], x[i]) + (u32)smulbb(x[i + 1], x[i + 1])) >> shft;
    if (i < len) nrg += (u32)smulbb(x[i], x[i]) >> shft;
    shift = shft;
    energy = (i32)nrg;
}

OG_DEV i32 silk_sqrt_approx(i32 x) { // silk.h:888
    if (x <= 0) return 0;
    const int lz = clz32(x);
    const int rot = 24 - lz; // silk_ROR32(x, 24 - lz) & 0x7f
    const u32 ux = (u32)x;
    const u32 r = rot == 0 ? ux : rot < 0 ? (ux << (u32)(-rot)) | (ux >> (u32)(32 + rot)) : (ux << (u32)(32 - rot)) | (ux >> (u32)rot);
    const i32 frac_Q7 = (i32)(r & 0x7f);
    i32 y = (lz & 1) ? 32768 : 46214;
    y >>= lz >> 1;
    return smlawb(y, y, smulbb(213, frac_Q7));
}

// silk_PLC's rate check (silk.cpp:2873-2876) with silk_PLC_Reset (:2862)
OG_DEV void silk_plc_rate_check(SilkLossChannel *lc, int fs_kHz, int frame_length) {
    if (lc->plc_fs_kHz != fs_kHz) {
        lc->plc_pitchL_Q8 = shl32(frame_length, 7);
        lc->plc_prevGain_Q16[0] = lc->plc_prevGain_Q16[1] = 1 << 16;
        lc->plc_subfr_length = 20;
        lc->plc_nb_subfr = 2;
        lc->plc_fs_kHz = fs_kHz;
    }
}

// silk_PLC_update silk.cpp:2895: after a decoded frame, what a concealment would start from
OG_DEV void silk_plc_update_lane(SilkLossChannel *lc, const SilkCtrl &k, int fs_kHz, int nb_subfr) {
    const int order = fs_kHz == 16 ? 16 : 10, subfr = 5 * fs_kHz;
    i32 LTP_Gain_Q14 = 0;
    if (k.signalType == 2) {
        for (int j = 0; j * subfr < k.pitchL[nb_subfr - 1]; j++) {
            if (j == nb_subfr) break;
            i32 t = 0;
            for (int i = 0; i < 5; i++) t += k.LTPCoef_Q14[(nb_subfr - 1 - j) * 5 + i];
            if (t > LTP_Gain_Q14) {
                LTP_Gain_Q14 = t;
                lc->plc_pitchL_Q8 = shl32(k.pitchL[nb_subfr - 1 - j], 8);
            }
        }
        i32 mid = (i32)(i16)LTP_Gain_Q14; // LTPCoef_Q14[LTP_ORDER / 2], the only tap left
        if (LTP_Gain_Q14 < 11469) // V_PITCH_GAIN_START_MIN_Q14
            mid = smulbb(mid, shl32(11469, 10) / OG_MAX(LTP_Gain_Q14, 1)) >> 10;
        else if (LTP_Gain_Q14 > 15565) // V_PITCH_GAIN_START_MAX_Q14
            mid = smulbb(mid, shl32(15565, 14) / OG_MAX(LTP_Gain_Q14, 1)) >> 14;
        for (int i = 0; i < 5; i++) lc->plc_LTPCoef_Q14[i] = (i16)(i == 2 ? mid : 0);
    } else {
        lc->plc_pitchL_Q8 = shl32(smulbb(fs_kHz, 18), 8);
        for (int i = 0; i < 5; i++) lc->plc_LTPCoef_Q14[i] = 0;
    }
    for (int i = 0; i < order; i++) lc->plc_prevLPC_Q12[i] = k.PredCoef_Q12[1][i];
    lc->plc_prevLTP_scale_Q14 = (i32)(i16)k.LTP_scale_Q14;
    lc->plc_prevGain_Q16[0] = k.Gains_Q16[nb_subfr - 2];
    lc->plc_prevGain_Q16[1] = k.Gains_Q16[nb_subfr - 1];
    lc->plc_subfr_length = subfr;
    lc->plc_nb_subfr = nb_subfr;
}

// silk_PLC_conceal silk.cpp:2973 (with silk_PLC_energy :2956).  Before it, wave-uniform (silk_decode_packet): the previous
// LPC filter cleared after a reset and bandwidth-expanded in the state, and -- first unvoiced loss -- its inverse prediction
// gain (`invGain_Q30`).  History: L.u.core.hist[ch] (the staged outBuf).  Output: L.xq[ch][2..].
OG_DEVN void silk_plc_conceal_lane(SilkChannel *c, SilkLossChannel *lc, int ch, int fs_kHz, int nb_subfr, i32 invGain_Q30) {
    SilkLds &L = SL();
    SilkCtrl &k = L.ctrl[ch];
    const int order = fs_kHz == 16 ? 16 : 10, subfr = 5 * fs_kHz, frame_length = nb_subfr * subfr, ltp_mem = 20 * fs_kHz;
    i16 *xq = &L.xq[ch][2];
    i32 *sLTP_Q14 = L.u.core.sLTP_Q15[ch];
    i16 *sLTP = L.u.core.sLTP[ch];
    const i16 *hist = L.u.core.hist[ch];
    const int lossCnt = lc->lossCnt, att = OG_MIN(1, lossCnt), prevType = c->prevSignalType;
    const i32 prevGain_Q10_0 = lc->plc_prevGain_Q16[0] >> 6, prevGain_Q10_1 = lc->plc_prevGain_Q16[1] >> 6;
    // the quieter of the last two subframes' excitation is the noise source (sLTP is free until the re-whitening)
    for (int i = 0; i < subfr; i++) {
        sLTP[i] = (i16)sat16(smulww(lc->exc_Q14[i + (nb_subfr - 2) * subfr], prevGain_Q10_0) >> 8);
        sLTP[subfr + i] = (i16)sat16(smulww(lc->exc_Q14[i + (nb_subfr - 1) * subfr], prevGain_Q10_1) >> 8);
    }
    i32 energy1, shift1, energy2, shift2;
    silk_sum_sqr_shift(energy1, shift1, sLTP, subfr);
    silk_sum_sqr_shift(energy2, shift2, sLTP + subfr, subfr);
    const int rand_base = (energy1 >> shift2) < (energy2 >> shift1) ? OG_MAX(0, (lc->plc_nb_subfr - 1) * lc->plc_subfr_length - 128)
                                                                     : OG_MAX(0, lc->plc_nb_subfr * lc->plc_subfr_length - 128);
    const i32 *rand_ptr = &lc->exc_Q14[rand_base];
    i32 B[5];
    for (int i = 0; i < 5; i++) B[i] = lc->plc_LTPCoef_Q14[i];
    i32 rand_scale_Q14 = (i32)(i16)lc->plc_randScale_Q14;
    const i32 harm_Gain_Q15 = att ? 31130 : 32440; // HARM_ATT_Q15
    i32 rand_Gain_Q15 = prevType == 2 ? (att ? 26214 : 31130) : (att ? 29491 : 32440); // PLC_RAND_ATTENUATE_V / _UV_Q15
    i16 A_Q12[SILK_MAX_LPC];
    for (int i = 0; i < order; i++) A_Q12[i] = lc->plc_prevLPC_Q12[i];
    if (lossCnt == 0) { // first lost frame
        rand_scale_Q14 = 1 << 14;
        if (prevType == 2) {
            for (int i = 0; i < 5; i++) rand_scale_Q14 = (i32)(i16)(rand_scale_Q14 - B[i]);
            rand_scale_Q14 = OG_MAX(3277, rand_scale_Q14);
            rand_scale_Q14 = (i32)(i16)(smulbb(rand_scale_Q14, lc->plc_prevLTP_scale_Q14) >> 14);
        } else {
            i32 down_scale_Q30 = OG_MIN((1 << 30) >> 3, invGain_Q30); // LOG2_INV_LPC_GAIN_HIGH_THRES
            down_scale_Q30 = OG_MAX((1 << 30) >> 8, down_scale_Q30);  // LOG2_INV_LPC_GAIN_LOW_THRES
            down_scale_Q30 = shl32(down_scale_Q30, 3);
            rand_Gain_Q15 = smulwb(down_scale_Q30, rand_Gain_Q15) >> 14;
        }
    }
    i32 rand_seed = lc->plc_rand_seed, pitchL_Q8 = lc->plc_pitchL_Q8;
    int lag = rshift_round(pitchL_Q8, 8), sLTP_buf_idx = ltp_mem;
    { // re-whiten the LTP state with the concealment's filter (silk_LPC_analysis_filter silk.cpp:2268), then scale it
        const int idx = ltp_mem - lag - order - 2;
        const i16 *in = &hist[idx];
        for (int ix = order; ix < ltp_mem - idx; ix++) {
            i32 acc = 0;
            for (int j = 0; j < order; j++) acc = smlabb(acc, in[ix - 1 - j], A_Q12[j]);
            sLTP[idx + ix] = (i16)sat16(rshift_round(subw(shl32((i32)in[ix], 12), acc), 12));
        }
        i32 inv_gain_Q30 = silk_inverse32_varQ(lc->plc_prevGain_Q16[1], 46);
        inv_gain_Q30 = OG_MIN(inv_gain_Q30, 0x7fffffff >> 1);
        for (int i = idx + order; i < ltp_mem; i++) sLTP_Q14[i] = smulwb(inv_gain_Q30, sLTP[i]);
    }
    for (int sf = 0; sf < nb_subfr; sf++) { // LTP synthesis
        const i32 *p = &sLTP_Q14[sLTP_buf_idx - lag + 2];
        for (int i = 0; i < subfr; i++) {
            i32 LTP_pred_Q12 = 2;
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, p[0], B[0]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, p[-1], B[1]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, p[-2], B[2]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, p[-3], B[3]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, p[-4], B[4]);
            p++;
            rand_seed = (i32)(907633515u + (u32)rand_seed * 196314165u);
            const int ridx = (rand_seed >> 25) & 127; // RAND_BUF_MASK
            sLTP_Q14[sLTP_buf_idx] = shl32(smlawb(LTP_pred_Q12, rand_ptr[ridx], rand_scale_Q14), 2);
            sLTP_buf_idx++;
        }
        for (int j = 0; j < 5; j++) B[j] = (i32)(i16)(smulbb(harm_Gain_Q15, B[j]) >> 15);
        if (prevType != 0) rand_scale_Q14 = (i32)(i16)(smulbb(rand_scale_Q14, rand_Gain_Q15) >> 15);
        pitchL_Q8 = smlawb(pitchL_Q8, pitchL_Q8, 655);                      // PITCH_DRIFT_FAC_Q16
        pitchL_Q8 = OG_MIN(pitchL_Q8, shl32(smulbb(18, fs_kHz), 8));         // MAX_PITCH_LAG_MS
        lag = rshift_round(pitchL_Q8, 8);
    }
    { // LPC synthesis, in place behind the 16 state samples
        i32 *sLPC = &sLTP_Q14[ltp_mem - SILK_MAX_LPC];
        for (int j = 0; j < SILK_MAX_LPC; j++) sLPC[j] = c->sLPC_Q14_buf[j];
        for (int i = 0; i < frame_length; i++) {
            i32 LPC_pred_Q10 = order >> 1;
            for (int j = 0; j < order; j++) LPC_pred_Q10 = smlawb(LPC_pred_Q10, sLPC[SILK_MAX_LPC + i - j - 1], A_Q12[j]);
            const i32 s = add_sat32(sLPC[SILK_MAX_LPC + i], lshift_sat32(LPC_pred_Q10, 4));
            sLPC[SILK_MAX_LPC + i] = s;
            xq[i] = (i16)sat16(rshift_round(smulww(s, prevGain_Q10_1), 8));
        }
        for (int j = 0; j < SILK_MAX_LPC; j++) c->sLPC_Q14_buf[j] = sLPC[frame_length + j];
    }
    for (int i = 0; i < 5; i++) lc->plc_LTPCoef_Q14[i] = (i16)B[i];
    lc->plc_pitchL_Q8 = pitchL_Q8;
    lc->plc_rand_seed = rand_seed;
    lc->plc_randScale_Q14 = rand_scale_Q14;
    for (int i = 0; i < 4; i++) k.pitchL[i] = lag;
    lc->lossCnt = lossCnt + 1;
    c->lagPrev = lag;
}

// silk_CNG silk.cpp:1342, one channel: after a decoded inactive frame the comfort-noise parameters follow the signal; while
// frames are lost, comfort noise is added to the concealment.  `A_Q12`: the smoothed NLSFs as a filter (made wave-uniformly
// by the caller, only looked at while frames are lost).  `frame`: L.xq[ch][2..], already copied to outBuf.
OG_DEVN void silk_cng_lane(SilkChannel *c, SilkLossChannel *lc, int ch, int fs_kHz, int nb_subfr, const i16 *A_Q12) {
    SilkLds &L = SL();
    const SilkCtrl &k = L.ctrl[ch];
    const int order = fs_kHz == 16 ? 16 : 10, subfr = 5 * fs_kHz, length = nb_subfr * subfr;
    i16 *frame = &L.xq[ch][2];
    if (lc->lossCnt == 0 && c->prevSignalType == 0) {
        for (int i = 0; i < order; i++) {
            const i32 v = lc->cng_smth_NLSF_Q15[i];
            lc->cng_smth_NLSF_Q15[i] = (i16)(v + smulwb((i32)c->prevNLSF_Q15[i] - v, 16348)); // CNG_NLSF_SMTH_Q16
        }
        i32 max_Gain_Q16 = 0;
        int sub = 0;
        for (int i = 0; i < nb_subfr; i++)
            if (k.Gains_Q16[i] > max_Gain_Q16) {
                max_Gain_Q16 = k.Gains_Q16[i];
                sub = i;
            }
        for (int i = (nb_subfr - 1) * subfr - 1; i >= 0; i--) lc->cng_exc_buf_Q14[subfr + i] = lc->cng_exc_buf_Q14[i];
        for (int i = 0; i < subfr; i++) lc->cng_exc_buf_Q14[i] = lc->exc_Q14[sub * subfr + i];
        i32 g = lc->cng_smth_Gain_Q16;
        for (int i = 0; i < nb_subfr; i++) g += smulwb(k.Gains_Q16[i] - g, 4634); // CNG_GAIN_SMTH_Q16
        lc->cng_smth_Gain_Q16 = g;
    }
    if (lc->lossCnt) {
        i32 *sig = L.u.core.sLTP_Q15[ch]; // [16 state samples | length]
        const i32 smth = lc->cng_smth_Gain_Q16;
        i32 gain_Q16 = smulww((i32)(i16)lc->plc_randScale_Q14, lc->plc_prevGain_Q16[1]);
        if (gain_Q16 >= (1 << 21) || smth > (1 << 23)) {
            gain_Q16 = (gain_Q16 >> 16) * (gain_Q16 >> 16);
            gain_Q16 = subw((smth >> 16) * (smth >> 16), shl32(gain_Q16, 5));
            gain_Q16 = shl32(silk_sqrt_approx(gain_Q16), 16);
        } else {
            gain_Q16 = smulww(gain_Q16, gain_Q16);
            gain_Q16 = subw(smulww(smth, smth), shl32(gain_Q16, 5));
            gain_Q16 = shl32(silk_sqrt_approx(gain_Q16), 8);
        }
        const i32 gain_Q10 = gain_Q16 >> 6;
        i32 exc_mask = 255, seed = lc->cng_rand_seed; // CNG_BUF_MASK_MAX
        while (exc_mask > length) exc_mask >>= 1;
        for (int i = 0; i < length; i++) { // silk_CNG_exc silk.cpp:1305
            seed = (i32)(907633515u + (u32)seed * 196314165u);
            sig[SILK_MAX_LPC + i] = lc->cng_exc_buf_Q14[(seed >> 24) & exc_mask];
        }
        lc->cng_rand_seed = seed;
        for (int j = 0; j < SILK_MAX_LPC; j++) sig[j] = lc->cng_synth_state[j];
        for (int i = 0; i < length; i++) {
            i32 LPC_pred_Q10 = order >> 1;
            for (int j = 0; j < order; j++) LPC_pred_Q10 = smlawb(LPC_pred_Q10, sig[SILK_MAX_LPC + i - j - 1], A_Q12[j]);
            const i32 s = add_sat32(sig[SILK_MAX_LPC + i], lshift_sat32(LPC_pred_Q10, 4));
            sig[SILK_MAX_LPC + i] = s;
            frame[i] = (i16)sat16((i32)frame[i] + sat16(rshift_round(smulww(s, gain_Q10), 8)));
        }
        for (int j = 0; j < SILK_MAX_LPC; j++) lc->cng_synth_state[j] = sig[length + j];
    } else
        for (int j = 0; j < order; j++) lc->cng_synth_state[j] = 0;
}

// silk_PLC_glue_frames silk.cpp:3138: the first decoded frame after a loss fades in from the concealment's energy
OG_DEVN void silk_glue_lane(SilkLossChannel *lc, int ch, int length) {
    i16 *frame = &SL().xq[ch][2];
    if (lc->lossCnt) {
        i32 e, s;
        silk_sum_sqr_shift(e, s, frame, length);
        lc->plc_conc_energy = e;
        lc->plc_conc_energy_shift = s;
        lc->plc_last_frame_lost = 1;
        return;
    }
    if (lc->plc_last_frame_lost) {
        i32 energy, energy_shift, conc = lc->plc_conc_energy;
        const i32 conc_shift = lc->plc_conc_energy_shift;
        silk_sum_sqr_shift(energy, energy_shift, frame, length);
        if (energy_shift > conc_shift)
            conc >>= energy_shift - conc_shift;
        else if (energy_shift < conc_shift)
            energy >>= conc_shift - energy_shift;
        if (energy > conc) {
            const int LZ = clz32(conc) - 1;
            conc = shl32(conc, LZ);
            energy >>= OG_MAX(24 - LZ, 0);
            const i32 frac_Q24 = conc / OG_MAX(energy, 1);
            i32 gain_Q16 = shl32(silk_sqrt_approx(frac_Q24), 4);
            const i32 slope_Q16 = shl32(((1 << 16) - gain_Q16) / length, 2);
            for (int i = 0; i < length; i++) {
                frame[i] = (i16)smulwb(gain_Q16, frame[i]);
                gain_Q16 += slope_Q16;
                if (gain_Q16 > 1 << 16) break;
            }
        }
        lc->plc_conc_energy = conc;
    }
    lc->plc_last_frame_lost = 0;
}
