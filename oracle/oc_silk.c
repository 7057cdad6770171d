/*
 * oc_silk.c -- CPU ORACLE (test infrastructure): the fixed-point SILK decoder as the reference runs
 * it (lostFlag == 0, API rate 48 kHz, 20 ms payloads, nChannelsAPI == nChannelsInternal).
 * Restates src/silk.cpp of the reference: silk_Decode (:1481), silk_decode_frame (:1974),
 * silk_decode_indices (:708), silk_decode_pulses (:898), silk_decode_parameters (:827),
 * silk_NLSF_decode / _stabilize / _unpack (:2466, :2676, :2762), silk_NLSF2A (:642),
 * silk_LPC_fit (:2314), silk_LPC_inverse_pred_gain (:2359-2443), silk_decode_core (:1806),
 * silk_stereo_* (:592-623, :4028), silk_resampler* (:3451-3713), with the state in oc_silk.
 * The loss path -- silk_PLC (:2862-3185), silk_CNG (:1305-1432), the lostFlag branches of silk_Decode / silk_decode_frame
 * and LBRR (FEC) decoding -- is restated too.  The reference never reaches it (lostFlag == 0 always, Q8): with lossCnt == 0
 * the per-frame state updates (silk_PLC_update, the CNG smoothing, glue_frames) do not touch the PCM, so
 * reference-mode output is unchanged; only the oracle's RFC mode (oc_decoder_set_rfc) calls with lostFlag != 0.
 */
#include "oc_celt_priv.h"

#define MAX_LPC 16
#define MAX_FRAME 320
#define LTP_ORDER 5

typedef struct {
    signed char GainsIndices[4], LTPIndex[4], NLSFIndices[MAX_LPC + 1];
    i16 lagIndex;
    signed char contourIndex, signalType, quantOffsetType, NLSFInterpCoef_Q2, PERIndex, LTP_scaleIndex, Seed;
} side_info;

typedef struct {
    i32 prev_gain_Q16;
    i32 exc_Q14[MAX_FRAME];
    i32 sLPC_Q14_buf[MAX_LPC];
    i16 outBuf[MAX_FRAME + 2 * 80];
    i32 lagPrev;
    signed char LastGainIndex;
    i32 fs_kHz, fs_API_hz, nb_subfr, frame_length, subfr_length, ltp_mem_length, LPC_order;
    i16 prevNLSF_Q15[MAX_LPC];
    i32 first_frame_after_reset;
    i32 nFramesDecoded, nFramesPerPacket;
    i32 ec_prevSignalType;
    i16 ec_prevLagIndex;
    i32 VAD_flags[3], LBRR_flag, LBRR_flags[3];
    side_info idx;
    i32 lossCnt, prevSignalType;
    struct { /* silk_PLC_struct silk.h:680 */
        i32 pitchL_Q8;
        i16 LTPCoef_Q14[LTP_ORDER], prevLPC_Q12[MAX_LPC];
        i32 last_frame_lost, rand_seed;
        i16 randScale_Q14;
        i32 conc_energy, conc_energy_shift;
        i16 prevLTP_scale_Q14;
        i32 prevGain_Q16[2], fs_kHz, nb_subfr, subfr_length;
    } plc;
    struct { /* silk_CNG_struct silk.h:696 */
        i32 exc_buf_Q14[MAX_FRAME];
        i16 smth_NLSF_Q15[MAX_LPC];
        i32 synth_state[MAX_LPC], smth_Gain_Q16, rand_seed, fs_kHz;
    } cng;
} chan_t;

typedef struct {
    i32 sIIR[6];
    i16 sFIR[8];
    i16 delayBuf[48];
    i32 batchSize, invRatio_Q16, Fs_in_kHz, Fs_out_kHz, inputDelay;
} resamp_t;

typedef struct { /* silk_decoder_control_t silk.h:747 */
    i32 pitchL[4], Gains_Q16[4];
    i16 PredCoef_Q12[2][MAX_LPC];
    i16 LTPCoef_Q14[LTP_ORDER * 4];
    i32 LTP_scale_Q14;
} ctrl_t;

struct oc_silk {
    chan_t ch[2];
    resamp_t rs[2];
    i16 pred_prev_Q13[2], sMid[2], sSide[2];
    i32 nChannelsAPI, nChannelsInternal, prev_decode_only_middle;
    ctrl_t ctrl;
    i32 prev_pitch_lag; /* silk_DecControlStruct::prevPitchLag (silk.cpp:1764-1769): exported by every decode call; silk_InitDecoder does not touch it */
};

int oc_silk_sizeof(void) { return (int)sizeof(struct oc_silk); }

/* TEST ENTRY: stage taps of the last SILK frame decoded (single-threaded tests only): per coded channel the decoder
 * control (gains, pitch lags, both sets of LPC coefficients, LTP taps and scale), the signal type and the internal-rate
 * output of the synthesis core, taken before stereo un-mixing rewrites it in place. */
static struct {
    int on;
    i32 valid[2], signalType[2], quantOffsetType[2], frame_length[2], lpc_order[2];
    ctrl_t ctrl[2];
    i16 xq[2][MAX_FRAME];
} g_silk_taps;
void oc_silk_taps_enable(int on) { g_silk_taps.on = on; }
/* what: 0 = {valid, signalType, quantOffsetType, frame_length, LPC order, LTP_scale_Q14} (i32 x 6), 1 = pitchL + Gains_Q16
 * (i32 x 8), 2 = PredCoef_Q12 (i16 x 2 x 16), 3 = LTPCoef_Q14 (i16 x 20), 4 = xq (i16 x frame_length).  Returns bytes. */
int oc_silk_taps_copy(int what, int ch, void *dst) {
    if (ch < 0 || ch > 1) return -1;
    switch (what) {
        case 0: {
            i32 v[6] = {g_silk_taps.valid[ch], g_silk_taps.signalType[ch], g_silk_taps.quantOffsetType[ch],
                        g_silk_taps.frame_length[ch], g_silk_taps.lpc_order[ch], g_silk_taps.ctrl[ch].LTP_scale_Q14};
            memcpy(dst, v, sizeof(v));
            return (int)sizeof(v);
        }
        case 1: memcpy(dst, g_silk_taps.ctrl[ch].pitchL, 8 * sizeof(i32)); return 8 * (int)sizeof(i32);
        case 2: memcpy(dst, g_silk_taps.ctrl[ch].PredCoef_Q12, sizeof(g_silk_taps.ctrl[ch].PredCoef_Q12)); return (int)sizeof(g_silk_taps.ctrl[ch].PredCoef_Q12);
        case 3: memcpy(dst, g_silk_taps.ctrl[ch].LTPCoef_Q14, sizeof(g_silk_taps.ctrl[ch].LTPCoef_Q14)); return (int)sizeof(g_silk_taps.ctrl[ch].LTPCoef_Q14);
        case 4: memcpy(dst, g_silk_taps.xq[ch], sizeof(i16) * (size_t)g_silk_taps.frame_length[ch]); return (int)sizeof(i16) * g_silk_taps.frame_length[ch];
    }
    return -1;
}

/* silk_CNG_Reset silk.cpp:1327 */
static void cng_reset(chan_t *c) {
    int i;
    i32 step = 32767 / (c->LPC_order + 1), acc = 0;
    for (i = 0; i < c->LPC_order; i++) {
        acc += step;
        c->cng.smth_NLSF_Q15[i] = (i16)acc;
    }
    c->cng.smth_Gain_Q16 = 0;
    c->cng.rand_seed = 3176576;
}

/* silk_PLC_Reset silk.cpp:2862 */
static void plc_reset(chan_t *c) {
    c->plc.pitchL_Q8 = shl32(c->frame_length, 8 - 1);
    c->plc.prevGain_Q16[0] = c->plc.prevGain_Q16[1] = 1 << 16;
    c->plc.subfr_length = 20;
    c->plc.nb_subfr = 2;
}

/* silk_init_decoder silk.cpp:2192 */
static void chan_init(chan_t *c) {
    memset(c, 0, sizeof(*c));
    c->first_frame_after_reset = 1;
    c->prev_gain_Q16 = 65536;
    cng_reset(c);
    plc_reset(c);
}

/* silk_InitDecoder silk.cpp:1792: the channel states and the stereo state; NOT the resamplers nor the
 * remembered channel counts (they are re-derived on the next frame) */
void oc_silk_init(oc_silk *s) {
    chan_init(&s->ch[0]);
    chan_init(&s->ch[1]);
    memset(s->pred_prev_Q13, 0, sizeof(s->pred_prev_Q13));
    memset(s->sMid, 0, sizeof(s->sMid));
    memset(s->sSide, 0, sizeof(s->sSide));
    s->prev_decode_only_middle = 0;
}

/* ---- small math (silk.h:913-996, silk.cpp:2248) -------------------------------------------------------- */
static i32 div32_varQ(i32 a32, i32 b32, int Qres) { /* silk_DIV32_varQ */
    int a_headrm = clz32(a32 < 0 ? -a32 : a32) - 1, b_headrm = clz32(b32 < 0 ? -b32 : b32) - 1, lshift;
    i32 a32_nrm = shl32(a32, a_headrm), b32_nrm = shl32(b32, b_headrm);
    i32 b32_inv = (INT32_MAX >> 2) / (b32_nrm >> 16);
    i32 result = smulwb(a32_nrm, b32_inv);
    a32_nrm = subw(a32_nrm, shl32(smmul(b32_nrm, result), 3));
    result = smlawb(result, a32_nrm, b32_inv);
    lshift = 29 + a_headrm - b_headrm - Qres;
    if (lshift < 0) return lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

static i32 inverse32_varQ(i32 b32, int Qres) { /* silk_INVERSE32_varQ */
    int b_headrm = clz32(b32 < 0 ? -b32 : b32) - 1, lshift;
    i32 b32_nrm = shl32(b32, b_headrm);
    i32 b32_inv = (INT32_MAX >> 2) / (b32_nrm >> 16);
    i32 result = shl32(b32_inv, 16);
    i32 err_Q32 = shl32((1 << 29) - smulwb(b32_nrm, b32_inv), 3);
    result = (i32)((u32)result + (u32)smulww(err_Q32, b32_inv)); /* silk_SMLAWW */
    lshift = 61 - b_headrm - Qres;
    if (lshift <= 0) return lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

static i32 log2lin(i32 inLog_Q7) { /* silk.cpp:2248 */
    i32 out, frac_Q7;
    if (inLog_Q7 < 0) return 0;
    if (inLog_Q7 >= 3967) return INT32_MAX;
    out = shl32(1, inLog_Q7 >> 7);
    frac_Q7 = inLog_Q7 & 0x7F;
    if (inLog_Q7 < 2048)
        out = out + ((out * smlawb(frac_Q7, smulbb(frac_Q7, 128 - frac_Q7), -174)) >> 7);
    else
        out = out + (out >> 7) * smlawb(frac_Q7, smulbb(frac_Q7, 128 - frac_Q7), -174);
    return out;
}

/* ---- codebook selection (silk_NLSF_CB_struct silk.h:639; instances silk.cpp:384-427) --------------------- */
typedef struct {
    int order;
    i32 quantStepSize_Q16;
    const u8 *CB1_NLSF_Q8, *CB1_iCDF, *pred_Q8, *ec_sel, *ec_iCDF;
    const i32 *CB1_Wght_Q9, *deltaMin_Q15;
} nlsf_cb;

static void get_cb(nlsf_cb *cb, int wb) {
    if (wb) {
        cb->order = 16;
        cb->quantStepSize_Q16 = 9830; /* SILK_FIX_CONST(0.15, 16) */
        cb->CB1_NLSF_Q8 = rom_silk_wb_cb1_q8;
        cb->CB1_Wght_Q9 = rom_silk_wb_cb1_wght_q9;
        cb->CB1_iCDF = rom_silk_wb_cb1_icdf;
        cb->pred_Q8 = rom_silk_wb_pred_q8;
        cb->ec_sel = rom_silk_wb_cb2_select;
        cb->ec_iCDF = rom_silk_wb_cb2_icdf;
        cb->deltaMin_Q15 = rom_silk_wb_delta_min_q15;
    } else {
        cb->order = 10;
        cb->quantStepSize_Q16 = 11796; /* SILK_FIX_CONST(0.18, 16) */
        cb->CB1_NLSF_Q8 = rom_silk_nb_cb1_q8;
        cb->CB1_Wght_Q9 = rom_silk_nb_cb1_wght_q9;
        cb->CB1_iCDF = rom_silk_nb_cb1_icdf;
        cb->pred_Q8 = rom_silk_nb_pred_q8;
        cb->ec_sel = rom_silk_nb_cb2_select;
        cb->ec_iCDF = rom_silk_nb_cb2_icdf;
        cb->deltaMin_Q15 = rom_silk_nb_delta_min_q15;
    }
}

/* silk_NLSF_unpack silk.cpp:2762 */
static void nlsf_unpack(i16 ec_ix[], u8 pred_Q8[], const nlsf_cb *cb, int CB1_index) {
    const u8 *sel = &cb->ec_sel[CB1_index * cb->order / 2];
    int i;
    for (i = 0; i < cb->order; i += 2) {
        u8 entry = *sel++;
        ec_ix[i] = (i16)smulbb((entry >> 1) & 7, 9);
        pred_Q8[i] = cb->pred_Q8[i + (entry & 1) * (cb->order - 1)];
        ec_ix[i + 1] = (i16)smulbb((entry >> 5) & 7, 9);
        pred_Q8[i + 1] = cb->pred_Q8[i + ((entry >> 4) & 1) * (cb->order - 1) + 1];
    }
}

/* ---- side information (silk_decode_indices silk.cpp:708) ------------------------------------------------- */
static void decode_indices(chan_t *c, oc_rc *rc, int FrameIndex, int decode_LBRR, int condCoding) {
    int i, k, Ix;
    i16 ec_ix[MAX_LPC];
    u8 pred_Q8[MAX_LPC];
    nlsf_cb cb;
    const u8 *lowbits, *contour;
    get_cb(&cb, c->LPC_order == 16);
    if (decode_LBRR || c->VAD_flags[FrameIndex])
        Ix = oc_rc_icdf(rc, rom_silk_type_vad_icdf, 8) + 2;
    else
        Ix = oc_rc_icdf(rc, rom_silk_type_novad_icdf, 8);
    c->idx.signalType = (signed char)(Ix >> 1);
    c->idx.quantOffsetType = (signed char)(Ix & 1);
    if (condCoding == 2)
        c->idx.GainsIndices[0] = (signed char)oc_rc_icdf(rc, rom_silk_delta_gain_icdf, 8);
    else {
        c->idx.GainsIndices[0] = (signed char)(oc_rc_icdf(rc, rom_silk_gain_icdf + 8 * c->idx.signalType, 8) << 3);
        c->idx.GainsIndices[0] += (signed char)oc_rc_icdf(rc, rom_silk_uniform8_icdf, 8);
    }
    for (i = 1; i < c->nb_subfr; i++) c->idx.GainsIndices[i] = (signed char)oc_rc_icdf(rc, rom_silk_delta_gain_icdf, 8);
    c->idx.NLSFIndices[0] = (signed char)oc_rc_icdf(rc, &cb.CB1_iCDF[(c->idx.signalType >> 1) * 32], 8);
    nlsf_unpack(ec_ix, pred_Q8, &cb, c->idx.NLSFIndices[0]);
    for (i = 0; i < cb.order; i++) {
        Ix = oc_rc_icdf(rc, &cb.ec_iCDF[ec_ix[i]], 8);
        if (Ix == 0)
            Ix -= oc_rc_icdf(rc, rom_silk_nlsf_ext_icdf, 8);
        else if (Ix == 8)
            Ix += oc_rc_icdf(rc, rom_silk_nlsf_ext_icdf, 8);
        c->idx.NLSFIndices[i + 1] = (signed char)(Ix - 4);
    }
    if (c->nb_subfr == 4)
        c->idx.NLSFInterpCoef_Q2 = (signed char)oc_rc_icdf(rc, rom_silk_nlsf_interp_icdf, 8);
    else
        c->idx.NLSFInterpCoef_Q2 = 4;
    if (c->idx.signalType == 2) {
        int decode_abs = 1;
        lowbits = c->fs_kHz == 16 ? rom_silk_uniform8_icdf : (c->fs_kHz == 12 ? rom_silk_uniform6_icdf : rom_silk_uniform4_icdf);
        if (c->fs_kHz == 8)
            contour = c->nb_subfr == 4 ? rom_silk_pitch_contour_nb_icdf : rom_silk_pitch_contour_10ms_nb_icdf;
        else
            contour = c->nb_subfr == 4 ? rom_silk_pitch_contour_icdf : rom_silk_pitch_contour_10ms_icdf;
        if (condCoding == 2 && c->ec_prevSignalType == 2) {
            int delta = (i16)oc_rc_icdf(rc, rom_silk_pitch_delta_icdf, 8);
            if (delta > 0) {
                delta -= 9;
                c->idx.lagIndex = (i16)(c->ec_prevLagIndex + delta);
                decode_abs = 0;
            }
        }
        if (decode_abs) {
            c->idx.lagIndex = (i16)((i16)oc_rc_icdf(rc, rom_silk_pitch_lag_icdf, 8) * (c->fs_kHz >> 1));
            c->idx.lagIndex += (i16)oc_rc_icdf(rc, lowbits, 8);
        }
        c->ec_prevLagIndex = c->idx.lagIndex;
        c->idx.contourIndex = (signed char)oc_rc_icdf(rc, contour, 8);
        c->idx.PERIndex = (signed char)oc_rc_icdf(rc, rom_silk_ltp_per_icdf, 8);
        for (k = 0; k < c->nb_subfr; k++) {
            const u8 *t = c->idx.PERIndex == 0 ? rom_silk_ltp_gain_icdf0
                          : (c->idx.PERIndex == 1 ? rom_silk_ltp_gain_icdf1 : rom_silk_ltp_gain_icdf2);
            c->idx.LTPIndex[k] = (signed char)oc_rc_icdf(rc, t, 8);
        }
        if (condCoding == 0)
            c->idx.LTP_scaleIndex = (signed char)oc_rc_icdf(rc, rom_silk_ltpscale_icdf, 8);
        else
            c->idx.LTP_scaleIndex = 0;
    }
    c->ec_prevSignalType = c->idx.signalType;
    c->idx.Seed = (signed char)oc_rc_icdf(rc, rom_silk_uniform4_icdf, 8);
}

/* ---- excitation (silk_decode_pulses :898, silk_shell_decoder :1162, silk_decode_signs :1436) ------------ */
static void split(oc_rc *rc, i16 *c1, i16 *c2, int p, const u8 *table) {
    if (p > 0) {
        c1[0] = (i16)oc_rc_icdf(rc, &table[rom_silk_shell_offsets[p]], 8);
        c2[0] = (i16)(p - c1[0]);
    } else {
        c1[0] = 0;
        c2[0] = 0;
    }
}

static void shell_decode(oc_rc *rc, i16 *p0, int pulses4) {
    i16 p3[2], p2[4], p1[8];
    split(rc, &p3[0], &p3[1], pulses4, rom_silk_shell3);
    split(rc, &p2[0], &p2[1], p3[0], rom_silk_shell2);
    split(rc, &p1[0], &p1[1], p2[0], rom_silk_shell1);
    split(rc, &p0[0], &p0[1], p1[0], rom_silk_shell0);
    split(rc, &p0[2], &p0[3], p1[1], rom_silk_shell0);
    split(rc, &p1[2], &p1[3], p2[1], rom_silk_shell1);
    split(rc, &p0[4], &p0[5], p1[2], rom_silk_shell0);
    split(rc, &p0[6], &p0[7], p1[3], rom_silk_shell0);
    split(rc, &p2[2], &p2[3], p3[1], rom_silk_shell2);
    split(rc, &p1[4], &p1[5], p2[2], rom_silk_shell1);
    split(rc, &p0[8], &p0[9], p1[4], rom_silk_shell0);
    split(rc, &p0[10], &p0[11], p1[5], rom_silk_shell0);
    split(rc, &p1[6], &p1[7], p2[3], rom_silk_shell1);
    split(rc, &p0[12], &p0[13], p1[6], rom_silk_shell0);
    split(rc, &p0[14], &p0[15], p1[7], rom_silk_shell0);
}

static void decode_pulses(oc_rc *rc, i16 pulses[], int signalType, int quantOffsetType, int frame_length) {
    i32 sum_pulses[20], nLshifts[20];
    int i, j, k, iter, RateLevelIndex;
    const u8 *cdf;
    RateLevelIndex = oc_rc_icdf(rc, rom_silk_rate_levels_icdf + 9 * (signalType >> 1), 8);
    iter = frame_length >> 4;
    if (iter * 16 < frame_length) iter++;
    cdf = rom_silk_pulses_per_block_icdf + 18 * RateLevelIndex;
    for (i = 0; i < iter; i++) {
        nLshifts[i] = 0;
        sum_pulses[i] = oc_rc_icdf(rc, cdf, 8);
        while (sum_pulses[i] == 17) {
            nLshifts[i]++;
            sum_pulses[i] = oc_rc_icdf(rc, rom_silk_pulses_per_block_icdf + 18 * 9 + (nLshifts[i] == 10), 8);
        }
    }
    for (i = 0; i < iter; i++) {
        if (sum_pulses[i] > 0)
            shell_decode(rc, &pulses[i * 16], sum_pulses[i]);
        else
            memset(&pulses[i * 16], 0, 16 * sizeof(pulses[0]));
    }
    for (i = 0; i < iter; i++) {
        if (nLshifts[i] > 0) {
            int nLS = nLshifts[i];
            i16 *p = &pulses[i * 16];
            for (k = 0; k < 16; k++) {
                i32 abs_q = p[k];
                for (j = 0; j < nLS; j++) {
                    abs_q = shl32(abs_q, 1);
                    abs_q += oc_rc_icdf(rc, rom_silk_lsb_icdf, 8);
                }
                p[k] = (i16)abs_q;
            }
            sum_pulses[i] |= nLS << 5;
        }
    }
    { /* signs */
        u8 icdf[2];
        const u8 *icdf_ptr = &rom_silk_sign_icdf[7 * (quantOffsetType + (signalType << 1))];
        i16 *q = pulses;
        int length = (frame_length + 8) >> 4;
        icdf[1] = 0;
        for (i = 0; i < length; i++) {
            int p = sum_pulses[i];
            if (p > 0) {
                icdf[0] = icdf_ptr[OC_MIN(p & 0x1F, 6)];
                for (j = 0; j < 16; j++)
                    if (q[j] > 0) q[j] = (i16)(q[j] * ((oc_rc_icdf(rc, icdf, 8) << 1) - 1));
            }
            q += 16;
        }
    }
}

/* ---- NLSF -> LPC ------------------------------------------------------------------------------------------ */
/* silk_NLSF_stabilize silk.cpp:2676 */
static void nlsf_stabilize(i16 *NLSF_Q15, const i32 *NDeltaMin_Q15, int L) {
    int i, I = 0, k, loops;
    i32 diff_Q15, min_diff_Q15, min_center_Q15, max_center_Q15;
    i16 center_freq_Q15;
    for (loops = 0; loops < 20; loops++) {
        min_diff_Q15 = NLSF_Q15[0] - NDeltaMin_Q15[0];
        I = 0;
        for (i = 1; i <= L - 1; i++) {
            diff_Q15 = NLSF_Q15[i] - (NLSF_Q15[i - 1] + NDeltaMin_Q15[i]);
            if (diff_Q15 < min_diff_Q15) {
                min_diff_Q15 = diff_Q15;
                I = i;
            }
        }
        diff_Q15 = (1 << 15) - (NLSF_Q15[L - 1] + NDeltaMin_Q15[L]);
        if (diff_Q15 < min_diff_Q15) {
            min_diff_Q15 = diff_Q15;
            I = L;
        }
        if (min_diff_Q15 >= 0) return;
        if (I == 0)
            NLSF_Q15[0] = NDeltaMin_Q15[0];
        else if (I == L)
            NLSF_Q15[L - 1] = (i16)((1 << 15) - NDeltaMin_Q15[L]);
        else {
            min_center_Q15 = 0;
            for (k = 0; k < I; k++) min_center_Q15 += NDeltaMin_Q15[k];
            min_center_Q15 += NDeltaMin_Q15[I] >> 1;
            max_center_Q15 = 1 << 15;
            for (k = L; k > I; k--) max_center_Q15 -= NDeltaMin_Q15[k];
            max_center_Q15 -= NDeltaMin_Q15[I] >> 1;
            center_freq_Q15 = (i16)limit32(rshift_round((i32)NLSF_Q15[I - 1] + (i32)NLSF_Q15[I], 1), min_center_Q15,
                                           max_center_Q15);
            NLSF_Q15[I - 1] = (i16)(center_freq_Q15 - (NDeltaMin_Q15[I] >> 1));
            NLSF_Q15[I] = (i16)(NLSF_Q15[I - 1] + NDeltaMin_Q15[I]);
        }
    }
    /* fall-back: sort, then enforce the spacing from both ends (silk.cpp:2741) */
    for (i = 1; i < L; i++) {
        i32 value = NLSF_Q15[i];
        int j;
        for (j = i - 1; j >= 0 && value < NLSF_Q15[j]; j--) NLSF_Q15[j + 1] = NLSF_Q15[j];
        NLSF_Q15[j + 1] = (i16)value;
    }
    NLSF_Q15[0] = (i16)OC_MAX((i32)NLSF_Q15[0], (i32)NDeltaMin_Q15[0]);
    for (i = 1; i < L; i++) {
        i32 lo = sat16((i32)NLSF_Q15[i - 1] + NDeltaMin_Q15[i]); /* silk_ADD_SAT16 */
        NLSF_Q15[i] = (i16)OC_MAX((i32)NLSF_Q15[i], lo);
    }
    NLSF_Q15[L - 1] = (i16)OC_MIN((i32)NLSF_Q15[L - 1], (1 << 15) - NDeltaMin_Q15[L]);
    for (i = L - 2; i >= 0; i--) NLSF_Q15[i] = (i16)OC_MIN((i32)NLSF_Q15[i], NLSF_Q15[i + 1] - NDeltaMin_Q15[i + 1]);
}

/* silk_NLSF_decode silk.cpp:2466 (+ residual dequantiser :2445) */
static void nlsf_decode(i16 *pNLSF_Q15, const signed char *NLSFIndices, const nlsf_cb *cb) {
    u8 pred_Q8[MAX_LPC];
    i16 ec_ix[MAX_LPC], res_Q10[MAX_LPC];
    i32 out_Q10 = 0, pred_Q10;
    const u8 *pCB = &cb->CB1_NLSF_Q8[NLSFIndices[0] * cb->order];
    const i32 *pW = &cb->CB1_Wght_Q9[NLSFIndices[0] * cb->order];
    int i;
    nlsf_unpack(ec_ix, pred_Q8, cb, NLSFIndices[0]);
    for (i = cb->order - 1; i >= 0; i--) {
        pred_Q10 = smulbb(out_Q10, (i16)pred_Q8[i]) >> 8;
        out_Q10 = shl32(NLSFIndices[1 + i], 10);
        if (out_Q10 > 0)
            out_Q10 = out_Q10 - 102; /* SILK_FIX_CONST(0.1, 10) */
        else if (out_Q10 < 0)
            out_Q10 = out_Q10 + 102;
        out_Q10 = smlawb(pred_Q10, out_Q10, cb->quantStepSize_Q16);
        res_Q10[i] = (i16)out_Q10;
    }
    for (i = 0; i < cb->order; i++) {
        i32 t = shl32((i32)res_Q10[i], 14) / pW[i] + shl32((i16)pCB[i], 7);
        pNLSF_Q15[i] = (i16)limit32(t, 0, 32767);
    }
    nlsf_stabilize(pNLSF_Q15, cb->deltaMin_Q15, cb->order);
}

/* silk_bwexpander_32 silk.cpp:561 */
static void bwexpander_32(i32 *ar, int d, i32 chirp_Q16) {
    i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    int i;
    for (i = 0; i < d - 1; i++) {
        ar[i] = smulww(chirp_Q16, ar[i]);
        chirp_Q16 += rshift_round(chirp_Q16 * chirp_minus_one_Q16, 16);
    }
    ar[d - 1] = smulww(chirp_Q16, ar[d - 1]);
}

/* silk_bwexpander silk.cpp:576 (16-bit coefficients) */
static void bwexpander(i16 *ar, int d, i32 chirp_Q16) {
    i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    int i;
    for (i = 0; i < d - 1; i++) {
        ar[i] = (i16)rshift_round(chirp_Q16 * ar[i], 16);
        chirp_Q16 += rshift_round(chirp_Q16 * chirp_minus_one_Q16, 16);
    }
    ar[d - 1] = (i16)rshift_round(chirp_Q16 * ar[d - 1], 16);
}

/* silk_LPC_fit silk.cpp:2314 */
static void lpc_fit(i16 *a_QOUT, i32 *a_QIN, int QOUT, int QIN, int d) {
    int i, k, idx = 0;
    i32 maxabs, absval, chirp_Q16;
    for (i = 0; i < 10; i++) {
        maxabs = 0;
        for (k = 0; k < d; k++) {
            absval = a_QIN[k] > 0 ? a_QIN[k] : -a_QIN[k];
            if (absval > maxabs) {
                maxabs = absval;
                idx = k;
            }
        }
        maxabs = rshift_round(maxabs, QIN - QOUT);
        if (maxabs > 32767) {
            maxabs = OC_MIN(maxabs, 163838);
            chirp_Q16 = 65470 - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);
            bwexpander_32(a_QIN, d, chirp_Q16);
        } else
            break;
    }
    if (i == 10) {
        for (k = 0; k < d; k++) {
            a_QOUT[k] = sat16(rshift_round(a_QIN[k], QIN - QOUT));
            a_QIN[k] = shl32((i32)a_QOUT[k], QIN - QOUT);
        }
    } else {
        for (k = 0; k < d; k++) a_QOUT[k] = (i16)rshift_round(a_QIN[k], QIN - QOUT);
    }
}

/* LPC_inverse_pred_gain_QA_c silk.cpp:2359 and its Q12 wrapper :2425 */
static i32 inverse_pred_gain(const i16 *A_Q12, int order) {
    i32 A[MAX_LPC], DC_resp = 0, invGain_Q30, rc_Q31, rc_mult1_Q30, rc_mult2, tmp1, tmp2;
    const i32 A_LIMIT = 16773022; /* SILK_FIX_CONST(0.99975, 24) */
    const i32 MIN_INVGAIN = 107374; /* SILK_FIX_CONST(1/1e4, 30) */
    int k, n, mult2Q;
    for (k = 0; k < order; k++) {
        DC_resp += (i32)A_Q12[k];
        A[k] = shl32((i32)A_Q12[k], 12);
    }
    if (DC_resp >= 4096) return 0;
    invGain_Q30 = 1 << 30;
    for (k = order - 1; k > 0; k--) {
        if (A[k] > A_LIMIT || A[k] < -A_LIMIT) return 0;
        rc_Q31 = -shl32(A[k], 7);
        rc_mult1_Q30 = (1 << 30) - smmul(rc_Q31, rc_Q31);
        invGain_Q30 = shl32(smmul(invGain_Q30, rc_mult1_Q30), 2);
        if (invGain_Q30 < MIN_INVGAIN) return 0;
        mult2Q = 32 - clz32(rc_mult1_Q30 < 0 ? -rc_mult1_Q30 : rc_mult1_Q30);
        rc_mult2 = inverse32_varQ(rc_mult1_Q30, mult2Q + 30);
        for (n = 0; n < (k + 1) >> 1; n++) {
            i64 tmp64;
            tmp1 = A[n];
            tmp2 = A[k - n - 1];
            tmp64 = rshift_round64((i64)sub_sat32(tmp1, (i32)rshift_round64((i64)tmp2 * rc_Q31, 31)) * rc_mult2, mult2Q);
            if (tmp64 > INT32_MAX || tmp64 < INT32_MIN) return 0;
            A[n] = (i32)tmp64;
            tmp64 = rshift_round64((i64)sub_sat32(tmp2, (i32)rshift_round64((i64)tmp1 * rc_Q31, 31)) * rc_mult2, mult2Q);
            if (tmp64 > INT32_MAX || tmp64 < INT32_MIN) return 0;
            A[k - n - 1] = (i32)tmp64;
        }
    }
    if (A[k] > A_LIMIT || A[k] < -A_LIMIT) return 0;
    rc_Q31 = -shl32(A[0], 7);
    rc_mult1_Q30 = (1 << 30) - smmul(rc_Q31, rc_Q31);
    invGain_Q30 = shl32(smmul(invGain_Q30, rc_mult1_Q30), 2);
    if (invGain_Q30 < MIN_INVGAIN) return 0;
    return invGain_Q30;
}

/* silk_NLSF2A_find_poly silk.cpp:626 */
static void find_poly(i32 *out, const i32 *cLSF, int dd) {
    int k, n;
    out[0] = 1 << 16;
    out[1] = -cLSF[0];
    for (k = 1; k < dd; k++) {
        i32 ftmp = cLSF[2 * k];
        out[k + 1] = shl32(out[k - 1], 1) - (i32)rshift_round64((i64)ftmp * out[k], 16);
        for (n = k; n > 1; n--) out[n] += out[n - 2] - (i32)rshift_round64((i64)ftmp * out[n - 1], 16);
        out[1] -= ftmp;
    }
}

/* silk_NLSF2A silk.cpp:642 */
static void nlsf2a(i16 *a_Q12, const i16 *NLSF, int d) {
    static const u8 ordering16[16] = {0, 15, 8, 7, 4, 11, 12, 3, 2, 13, 10, 5, 6, 9, 14, 1};
    static const u8 ordering10[10] = {0, 9, 6, 3, 4, 5, 8, 1, 2, 7};
    const u8 *ordering = d == 16 ? ordering16 : ordering10;
    i32 cos_LSF_QA[MAX_LPC], P[MAX_LPC / 2 + 1], Q[MAX_LPC / 2 + 1], a32_QA1[MAX_LPC];
    int k, i, dd;
    for (k = 0; k < d; k++) {
        i32 f_int = NLSF[k] >> 8, f_frac = NLSF[k] - shl32(f_int, 8);
        i32 cos_val = rom_silk_cos_q12[f_int], delta = rom_silk_cos_q12[f_int + 1] - cos_val;
        cos_LSF_QA[ordering[k]] = rshift_round(shl32(cos_val, 8) + delta * f_frac, 4);
    }
    dd = d >> 1;
    find_poly(P, &cos_LSF_QA[0], dd);
    find_poly(Q, &cos_LSF_QA[1], dd);
    for (k = 0; k < dd; k++) {
        i32 Ptmp = P[k + 1] + P[k], Qtmp = Q[k + 1] - Q[k];
        a32_QA1[k] = -Qtmp - Ptmp;
        a32_QA1[d - k - 1] = Qtmp - Ptmp;
    }
    lpc_fit(a_Q12, a32_QA1, 12, 17, d);
    for (i = 0; inverse_pred_gain(a_Q12, d) == 0 && i < 16; i++) {
        bwexpander_32(a32_QA1, d, 65536 - shl32(2, i));
        for (k = 0; k < d; k++) a_Q12[k] = (i16)rshift_round(a32_QA1[k], 5);
    }
}

/* silk_decode_pitch silk.cpp:2055 */
static void decode_pitch(int lagIndex, int contourIndex, i32 pitch_lags[], int Fs_kHz, int nb_subfr) {
    const signed char *cbk;
    int cbk_size, k, lag, min_lag = smulbb(2, Fs_kHz), max_lag = smulbb(18, Fs_kHz);
    if (Fs_kHz == 8) {
        if (nb_subfr == 4) { cbk = (const signed char *)rom_silk_lags_stage2; cbk_size = 11; }
        else { cbk = (const signed char *)rom_silk_lags_stage2_10ms; cbk_size = 3; }
    } else {
        if (nb_subfr == 4) { cbk = (const signed char *)rom_silk_lags_stage3; cbk_size = 34; }
        else { cbk = (const signed char *)rom_silk_lags_stage3_10ms; cbk_size = 12; }
    }
    lag = min_lag + lagIndex;
    for (k = 0; k < nb_subfr; k++) {
        pitch_lags[k] = lag + cbk[k * cbk_size + contourIndex];
        pitch_lags[k] = limit32(pitch_lags[k], min_lag, max_lag);
    }
}

/* silk_decode_parameters silk.cpp:827 (gains_dequant :2148 inlined) */
static void decode_parameters(chan_t *c, ctrl_t *ct, int condCoding) {
    i16 pNLSF_Q15[MAX_LPC], pNLSF0_Q15[MAX_LPC];
    nlsf_cb cb;
    int i, k;
    get_cb(&cb, c->LPC_order == 16);
    for (k = 0; k < c->nb_subfr; k++) {
        int ind = c->idx.GainsIndices[k], prev = c->LastGainIndex;
        if (k == 0 && condCoding != 2)
            prev = OC_MAX(ind, prev - 16);
        else {
            int ind_tmp = ind - 4, thr = 2 * 36 - 64 + prev;
            if (ind_tmp > thr) prev += shl32(ind_tmp, 1) - thr; else prev += ind_tmp;
        }
        c->LastGainIndex = (signed char)prev; /* *prev_ind is an int8: the += wraps before the clamp */
        c->LastGainIndex = (signed char)limit32(c->LastGainIndex, 0, 63);
        ct->Gains_Q16[k] = log2lin(OC_MIN(smulwb(1907825, c->LastGainIndex) + 2090, 3967));
    }
    nlsf_decode(pNLSF_Q15, c->idx.NLSFIndices, &cb);
    nlsf2a(ct->PredCoef_Q12[1], pNLSF_Q15, c->LPC_order);
    if (c->first_frame_after_reset == 1) c->idx.NLSFInterpCoef_Q2 = 4;
    if (c->idx.NLSFInterpCoef_Q2 < 4) {
        for (i = 0; i < c->LPC_order; i++)
            pNLSF0_Q15[i] = (i16)(c->prevNLSF_Q15[i] + ((c->idx.NLSFInterpCoef_Q2 * (pNLSF_Q15[i] - c->prevNLSF_Q15[i])) >> 2));
        nlsf2a(ct->PredCoef_Q12[0], pNLSF0_Q15, c->LPC_order);
    } else
        memcpy(ct->PredCoef_Q12[0], ct->PredCoef_Q12[1], c->LPC_order * sizeof(i16));
    memcpy(c->prevNLSF_Q15, pNLSF_Q15, c->LPC_order * sizeof(i16));
    if (c->lossCnt) { /* silk.cpp:860-864: after a packet loss do BWE of the LPC coefficients (BWE_AFTER_LOSS_Q16) */
        bwexpander(ct->PredCoef_Q12[0], c->LPC_order, 63570);
        bwexpander(ct->PredCoef_Q12[1], c->LPC_order, 63570);
    }
    if (c->idx.signalType == 2) {
        const signed char *cbk = (const signed char *)(c->idx.PERIndex == 0 ? rom_silk_ltp_vq0
                                                       : (c->idx.PERIndex == 1 ? rom_silk_ltp_vq1 : rom_silk_ltp_vq2));
        decode_pitch(c->idx.lagIndex, c->idx.contourIndex, ct->pitchL, c->fs_kHz, c->nb_subfr);
        for (k = 0; k < c->nb_subfr; k++) {
            int Ix = c->idx.LTPIndex[k];
            for (i = 0; i < LTP_ORDER; i++) ct->LTPCoef_Q14[k * LTP_ORDER + i] = (i16)shl32(cbk[Ix * LTP_ORDER + i], 7);
        }
        ct->LTP_scale_Q14 = rom_silk_ltp_scales_q14[c->idx.LTP_scaleIndex];
    } else {
        memset(ct->pitchL, 0, c->nb_subfr * sizeof(i32));
        memset(ct->LTPCoef_Q14, 0, LTP_ORDER * c->nb_subfr * sizeof(i16));
        c->idx.PERIndex = 0;
        ct->LTP_scale_Q14 = 0;
    }
}

/* silk_LPC_analysis_filter silk.cpp:2268 */
static void lpc_analysis_filter(i16 *out, const i16 *in, const i16 *B, int len, int d) {
    int ix, j;
    for (ix = d; ix < len; ix++) {
        const i16 *p = &in[ix - 1];
        i32 acc = smulbb(p[0], B[0]);
        for (j = 1; j < d; j++) acc = smlabb(acc, p[-j], B[j]);
        acc = subw(shl32((i32)p[1], 12), acc);
        out[ix] = sat16(rshift_round(acc, 12));
    }
    memset(out, 0, d * sizeof(i16));
}

/* silk_decode_core silk.cpp:1806 */
static void decode_core(chan_t *c, ctrl_t *ct, i16 xq[], const i16 pulses[]) {
    i16 sLTP[MAX_FRAME];
    i32 sLTP_Q15[2 * MAX_FRAME], res_Q14[80], sLPC_Q14[80 + MAX_LPC];
    i32 *pexc_Q14, *pres_Q14, rand_seed;
    i16 *pxq, A_Q12_tmp[MAX_LPC];
    int i, k, lag = 0, sLTP_buf_idx, NLSF_interpolation_flag, signalType;
    i32 offset_Q10 = rom_silk_quant_offsets_q10[(c->idx.signalType >> 1) * 2 + c->idx.quantOffsetType];
    memset(sLTP, 0, sizeof(sLTP));
    memset(sLTP_Q15, 0, sizeof(sLTP_Q15)); /* reference leaves it malloc'd; every entry read is written first */
    NLSF_interpolation_flag = c->idx.NLSFInterpCoef_Q2 < 4;
    rand_seed = c->idx.Seed;
    for (i = 0; i < c->frame_length; i++) {
        rand_seed = silk_rand(rand_seed);
        c->exc_Q14[i] = shl32((i32)pulses[i], 14);
        if (c->exc_Q14[i] > 0)
            c->exc_Q14[i] -= 80 << 4;
        else if (c->exc_Q14[i] < 0)
            c->exc_Q14[i] += 80 << 4;
        c->exc_Q14[i] += offset_Q10 << 4;
        if (rand_seed < 0) c->exc_Q14[i] = -c->exc_Q14[i];
        rand_seed = addw(rand_seed, pulses[i]);
    }
    memcpy(sLPC_Q14, c->sLPC_Q14_buf, MAX_LPC * sizeof(i32));
    pexc_Q14 = c->exc_Q14;
    pxq = xq;
    sLTP_buf_idx = c->ltp_mem_length;
    for (k = 0; k < c->nb_subfr; k++) {
        const i16 *A_Q12 = ct->PredCoef_Q12[k >> 1];
        i16 *B_Q14 = &ct->LTPCoef_Q14[k * LTP_ORDER];
        i32 Gain_Q10, inv_gain_Q31, gain_adj_Q16;
        pres_Q14 = res_Q14;
        memcpy(A_Q12_tmp, A_Q12, c->LPC_order * sizeof(i16));
        signalType = c->idx.signalType;
        Gain_Q10 = ct->Gains_Q16[k] >> 6;
        inv_gain_Q31 = inverse32_varQ(ct->Gains_Q16[k], 47);
        if (ct->Gains_Q16[k] != c->prev_gain_Q16) {
            gain_adj_Q16 = div32_varQ(c->prev_gain_Q16, ct->Gains_Q16[k], 16);
            for (i = 0; i < MAX_LPC; i++) sLPC_Q14[i] = smulww(gain_adj_Q16, sLPC_Q14[i]);
        } else
            gain_adj_Q16 = 1 << 16;
        c->prev_gain_Q16 = ct->Gains_Q16[k];
        /* silk.cpp:1869-1876: avoid an abrupt transition from voiced PLC to unvoiced normal decoding */
        if (c->lossCnt && c->prevSignalType == 2 && c->idx.signalType != 2 && k < 2) {
            memset(B_Q14, 0, LTP_ORDER * sizeof(i16));
            B_Q14[LTP_ORDER / 2] = 4096; /* SILK_FIX_CONST(0.25, 14) */
            signalType = 2;
            ct->pitchL[k] = c->lagPrev;
        }
        if (signalType == 2) {
            lag = ct->pitchL[k];
            if (k == 0 || (k == 2 && NLSF_interpolation_flag)) {
                int start_idx = c->ltp_mem_length - lag - c->LPC_order - LTP_ORDER / 2;
                if (k == 2) memcpy(&c->outBuf[c->ltp_mem_length], xq, 2 * c->subfr_length * sizeof(i16));
                lpc_analysis_filter(&sLTP[start_idx], &c->outBuf[start_idx + k * c->subfr_length], A_Q12,
                                    c->ltp_mem_length - start_idx, c->LPC_order);
                if (k == 0) inv_gain_Q31 = shl32(smulwb(inv_gain_Q31, ct->LTP_scale_Q14), 2);
                for (i = 0; i < lag + LTP_ORDER / 2; i++)
                    sLTP_Q15[sLTP_buf_idx - i - 1] = smulwb(inv_gain_Q31, sLTP[c->ltp_mem_length - i - 1]);
            } else if (gain_adj_Q16 != 1 << 16) {
                for (i = 0; i < lag + LTP_ORDER / 2; i++)
                    sLTP_Q15[sLTP_buf_idx - i - 1] = smulww(gain_adj_Q16, sLTP_Q15[sLTP_buf_idx - i - 1]);
            }
        }
        if (signalType == 2) {
            i32 *pred_lag_ptr = &sLTP_Q15[sLTP_buf_idx - lag + LTP_ORDER / 2];
            for (i = 0; i < c->subfr_length; i++) {
                i32 LTP_pred_Q13 = 2;
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, pred_lag_ptr[0], B_Q14[0]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, pred_lag_ptr[-1], B_Q14[1]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, pred_lag_ptr[-2], B_Q14[2]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, pred_lag_ptr[-3], B_Q14[3]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, pred_lag_ptr[-4], B_Q14[4]);
                pred_lag_ptr++;
                pres_Q14[i] = pexc_Q14[i] + shl32(LTP_pred_Q13, 1);
                sLTP_Q15[sLTP_buf_idx] = shl32(pres_Q14[i], 1);
                sLTP_buf_idx++;
            }
        } else
            pres_Q14 = pexc_Q14;
        for (i = 0; i < c->subfr_length; i++) {
            i32 LPC_pred_Q10 = c->LPC_order >> 1;
            int j;
            for (j = 0; j < c->LPC_order; j++)
                LPC_pred_Q10 = smlawb(LPC_pred_Q10, sLPC_Q14[MAX_LPC + i - 1 - j], A_Q12_tmp[j]);
            sLPC_Q14[MAX_LPC + i] = add_sat32(pres_Q14[i], lshift_sat32(LPC_pred_Q10, 4));
            pxq[i] = sat16(rshift_round(smulww(sLPC_Q14[MAX_LPC + i], Gain_Q10), 8));
        }
        memcpy(sLPC_Q14, &sLPC_Q14[c->subfr_length], MAX_LPC * sizeof(i32));
        pexc_Q14 += c->subfr_length;
        pxq += c->subfr_length;
    }
    memcpy(c->sLPC_Q14_buf, sLPC_Q14, MAX_LPC * sizeof(i32));
}

/* ---- loss concealment and comfort noise (silk.cpp:2862-3185, :1305-1432) -------------------------------- */
/* silk_sum_sqr_shift silk.cpp:3839 */
static void sum_sqr_shift(i32 *energy, i32 *shift, const i16 *x, int len) {
    int i, shft = 31 - clz32(len);
    i32 nrg = len;
    u32 t;
    for (i = 0; i < len - 1; i += 2) {
        t = (u32)smulbb(x[i], x[i]);
        t += (u32)smulbb(x[i + 1], x[i + 1]);
        nrg = (i32)((u32)nrg + (t >> shft));
    }
    if (i < len) {
        t = (u32)smulbb(x[i], x[i]);
        nrg = (i32)((u32)nrg + (t >> shft));
    }
    shft = OC_MAX(0, shft + 3 - clz32(nrg));
    nrg = 0;
    for (i = 0; i < len - 1; i += 2) {
        t = (u32)smulbb(x[i], x[i]);
        t += (u32)smulbb(x[i + 1], x[i + 1]);
        nrg = (i32)((u32)nrg + (t >> shft));
    }
    if (i < len) {
        t = (u32)smulbb(x[i], x[i]);
        nrg = (i32)((u32)nrg + (t >> shft));
    }
    *shift = shft;
    *energy = nrg;
}

/* silk_SQRT_APPROX silk.h:888 */
static i32 sqrt_approx(i32 x) {
    i32 y, lz, frac_Q7;
    if (x <= 0) return 0;
    lz = clz32(x);
    frac_Q7 = ror32(x, 24 - lz) & 0x7f;
    y = (lz & 1) ? 32768 : 46214;
    y >>= lz >> 1;
    return smlawb(y, y, smulbb(213, frac_Q7));
}

/* silk_PLC_update silk.cpp:2895 */
static void plc_update(chan_t *c, const ctrl_t *ct) {
    i32 LTP_Gain_Q14 = 0, temp;
    int i, j;
    c->prevSignalType = c->idx.signalType;
    if (c->idx.signalType == 2) {
        for (j = 0; j * c->subfr_length < ct->pitchL[c->nb_subfr - 1]; j++) {
            if (j == c->nb_subfr) break;
            temp = 0;
            for (i = 0; i < LTP_ORDER; i++) temp += ct->LTPCoef_Q14[(c->nb_subfr - 1 - j) * LTP_ORDER + i];
            if (temp > LTP_Gain_Q14) {
                LTP_Gain_Q14 = temp;
                memcpy(c->plc.LTPCoef_Q14, &ct->LTPCoef_Q14[smulbb(c->nb_subfr - 1 - j, LTP_ORDER)], LTP_ORDER * sizeof(i16));
                c->plc.pitchL_Q8 = shl32(ct->pitchL[c->nb_subfr - 1 - j], 8);
            }
        }
        memset(c->plc.LTPCoef_Q14, 0, LTP_ORDER * sizeof(i16));
        c->plc.LTPCoef_Q14[LTP_ORDER / 2] = (i16)LTP_Gain_Q14;
        if (LTP_Gain_Q14 < 11469) { /* V_PITCH_GAIN_START_MIN_Q14 */
            i32 scale_Q10 = shl32(11469, 10) / OC_MAX(LTP_Gain_Q14, 1);
            for (i = 0; i < LTP_ORDER; i++) c->plc.LTPCoef_Q14[i] = (i16)(smulbb(c->plc.LTPCoef_Q14[i], scale_Q10) >> 10);
        } else if (LTP_Gain_Q14 > 15565) { /* V_PITCH_GAIN_START_MAX_Q14 */
            i32 scale_Q14 = shl32(15565, 14) / OC_MAX(LTP_Gain_Q14, 1);
            for (i = 0; i < LTP_ORDER; i++) c->plc.LTPCoef_Q14[i] = (i16)(smulbb(c->plc.LTPCoef_Q14[i], scale_Q14) >> 14);
        }
    } else {
        c->plc.pitchL_Q8 = shl32(smulbb(c->fs_kHz, 18), 8);
        memset(c->plc.LTPCoef_Q14, 0, LTP_ORDER * sizeof(i16));
    }
    memcpy(c->plc.prevLPC_Q12, ct->PredCoef_Q12[1], c->LPC_order * sizeof(i16));
    c->plc.prevLTP_scale_Q14 = (i16)ct->LTP_scale_Q14;
    memcpy(c->plc.prevGain_Q16, &ct->Gains_Q16[c->nb_subfr - 2], 2 * sizeof(i32));
    c->plc.subfr_length = c->subfr_length;
    c->plc.nb_subfr = c->nb_subfr;
}

/* silk_PLC_conceal silk.cpp:2973 (with silk_PLC_energy :2956) */
static void plc_conceal(chan_t *c, ctrl_t *ct, i16 frame[]) {
    static const i16 HARM_ATT_Q15[2] = {32440, 31130}, RAND_ATT_V_Q15[2] = {31130, 26214}, RAND_ATT_UV_Q15[2] = {32440, 29491};
    i32 sLTP_Q14[2 * MAX_FRAME + MAX_LPC], prevGain_Q10[2], energy1, energy2, shift1, shift2, *rand_ptr, *pred_lag_ptr, *sLPC;
    i32 rand_seed, harm_Gain_Q15, rand_Gain_Q15, inv_gain_Q30, lag, idx, sLTP_buf_idx;
    i16 sLTP[MAX_FRAME], exc_buf[2 * 80], A_Q12[MAX_LPC], rand_scale_Q14, *B_Q14 = c->plc.LTPCoef_Q14;
    int i, j, k, att = OC_MIN(1, c->lossCnt);
    memset(sLTP_Q14, 0, sizeof(sLTP_Q14));
    memset(sLTP, 0, sizeof(sLTP));
    prevGain_Q10[0] = c->plc.prevGain_Q16[0] >> 6;
    prevGain_Q10[1] = c->plc.prevGain_Q16[1] >> 6;
    if (c->first_frame_after_reset) memset(c->plc.prevLPC_Q12, 0, sizeof(c->plc.prevLPC_Q12));
    for (k = 0; k < 2; k++)
        for (i = 0; i < c->subfr_length; i++)
            exc_buf[k * c->subfr_length + i] =
                sat16(smulww(c->exc_Q14[i + (k + c->nb_subfr - 2) * c->subfr_length], prevGain_Q10[k]) >> 8);
    sum_sqr_shift(&energy1, &shift1, exc_buf, c->subfr_length);
    sum_sqr_shift(&energy2, &shift2, &exc_buf[c->subfr_length], c->subfr_length);
    if ((energy1 >> shift2) < (energy2 >> shift1))
        rand_ptr = &c->exc_Q14[OC_MAX(0, (c->plc.nb_subfr - 1) * c->plc.subfr_length - 128)];
    else
        rand_ptr = &c->exc_Q14[OC_MAX(0, c->plc.nb_subfr * c->plc.subfr_length - 128)];
    rand_scale_Q14 = c->plc.randScale_Q14;
    harm_Gain_Q15 = HARM_ATT_Q15[att];
    rand_Gain_Q15 = c->prevSignalType == 2 ? RAND_ATT_V_Q15[att] : RAND_ATT_UV_Q15[att];
    bwexpander(c->plc.prevLPC_Q12, c->LPC_order, 64881); /* SILK_FIX_CONST(BWE_COEF = 0.99, 16) */
    memcpy(A_Q12, c->plc.prevLPC_Q12, c->LPC_order * sizeof(i16));
    if (c->lossCnt == 0) { /* first lost frame */
        rand_scale_Q14 = 1 << 14;
        if (c->prevSignalType == 2) {
            for (i = 0; i < LTP_ORDER; i++) rand_scale_Q14 = (i16)(rand_scale_Q14 - B_Q14[i]);
            rand_scale_Q14 = OC_MAX(3277, rand_scale_Q14);
            rand_scale_Q14 = (i16)(smulbb(rand_scale_Q14, c->plc.prevLTP_scale_Q14) >> 14);
        } else {
            i32 invGain_Q30 = inverse_pred_gain(c->plc.prevLPC_Q12, c->LPC_order), down_scale_Q30;
            down_scale_Q30 = OC_MIN((1 << 30) >> 3, invGain_Q30); /* LOG2_INV_LPC_GAIN_HIGH_THRES */
            down_scale_Q30 = OC_MAX((1 << 30) >> 8, down_scale_Q30); /* LOG2_INV_LPC_GAIN_LOW_THRES */
            down_scale_Q30 = shl32(down_scale_Q30, 3);
            rand_Gain_Q15 = smulwb(down_scale_Q30, rand_Gain_Q15) >> 14;
        }
    }
    rand_seed = c->plc.rand_seed;
    lag = rshift_round(c->plc.pitchL_Q8, 8);
    sLTP_buf_idx = c->ltp_mem_length;
    idx = c->ltp_mem_length - lag - c->LPC_order - LTP_ORDER / 2;
    lpc_analysis_filter(&sLTP[idx], &c->outBuf[idx], A_Q12, c->ltp_mem_length - idx, c->LPC_order);
    inv_gain_Q30 = inverse32_varQ(c->plc.prevGain_Q16[1], 46);
    inv_gain_Q30 = OC_MIN(inv_gain_Q30, INT32_MAX >> 1);
    for (i = idx + c->LPC_order; i < c->ltp_mem_length; i++) sLTP_Q14[i] = smulwb(inv_gain_Q30, sLTP[i]);
    for (k = 0; k < c->nb_subfr; k++) {
        pred_lag_ptr = &sLTP_Q14[sLTP_buf_idx - lag + LTP_ORDER / 2];
        for (i = 0; i < c->subfr_length; i++) {
            i32 LTP_pred_Q12 = 2;
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, pred_lag_ptr[0], B_Q14[0]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, pred_lag_ptr[-1], B_Q14[1]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, pred_lag_ptr[-2], B_Q14[2]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, pred_lag_ptr[-3], B_Q14[3]);
            LTP_pred_Q12 = smlawb(LTP_pred_Q12, pred_lag_ptr[-4], B_Q14[4]);
            pred_lag_ptr++;
            rand_seed = silk_rand(rand_seed);
            idx = (rand_seed >> 25) & 127; /* RAND_BUF_MASK */
            sLTP_Q14[sLTP_buf_idx] = shl32(smlawb(LTP_pred_Q12, rand_ptr[idx], rand_scale_Q14), 2);
            sLTP_buf_idx++;
        }
        for (j = 0; j < LTP_ORDER; j++) B_Q14[j] = (i16)(smulbb(harm_Gain_Q15, B_Q14[j]) >> 15);
        if (c->idx.signalType != 0) rand_scale_Q14 = (i16)(smulbb(rand_scale_Q14, rand_Gain_Q15) >> 15);
        c->plc.pitchL_Q8 = smlawb(c->plc.pitchL_Q8, c->plc.pitchL_Q8, 655); /* PITCH_DRIFT_FAC_Q16 */
        c->plc.pitchL_Q8 = OC_MIN(c->plc.pitchL_Q8, shl32(smulbb(18, c->fs_kHz), 8)); /* MAX_PITCH_LAG_MS */
        lag = rshift_round(c->plc.pitchL_Q8, 8);
    }
    sLPC = &sLTP_Q14[c->ltp_mem_length - MAX_LPC];
    memcpy(sLPC, c->sLPC_Q14_buf, MAX_LPC * sizeof(i32));
    for (i = 0; i < c->frame_length; i++) {
        i32 LPC_pred_Q10 = c->LPC_order >> 1;
        for (j = 0; j < c->LPC_order; j++) LPC_pred_Q10 = smlawb(LPC_pred_Q10, sLPC[MAX_LPC + i - j - 1], A_Q12[j]);
        sLPC[MAX_LPC + i] = add_sat32(sLPC[MAX_LPC + i], lshift_sat32(LPC_pred_Q10, 4));
        frame[i] = sat16(rshift_round(smulww(sLPC[MAX_LPC + i], prevGain_Q10[1]), 8));
    }
    memcpy(c->sLPC_Q14_buf, &sLPC[c->frame_length], MAX_LPC * sizeof(i32));
    c->plc.rand_seed = rand_seed;
    c->plc.randScale_Q14 = rand_scale_Q14;
    for (i = 0; i < 4; i++) ct->pitchL[i] = lag;
}

/* silk_PLC silk.cpp:2871 */
static void plc(chan_t *c, ctrl_t *ct, i16 frame[], int lost) {
    if (c->fs_kHz != c->plc.fs_kHz) {
        plc_reset(c);
        c->plc.fs_kHz = c->fs_kHz;
    }
    if (lost) {
        plc_conceal(c, ct, frame);
        c->lossCnt++;
    } else
        plc_update(c, ct);
}

/* silk_PLC_glue_frames silk.cpp:3138 */
static void plc_glue_frames(chan_t *c, i16 frame[], int length) {
    i32 energy, energy_shift;
    int i;
    if (c->lossCnt) {
        sum_sqr_shift(&c->plc.conc_energy, &c->plc.conc_energy_shift, frame, length);
        c->plc.last_frame_lost = 1;
        return;
    }
    if (c->plc.last_frame_lost) {
        sum_sqr_shift(&energy, &energy_shift, frame, length);
        if (energy_shift > c->plc.conc_energy_shift)
            c->plc.conc_energy >>= energy_shift - c->plc.conc_energy_shift;
        else if (energy_shift < c->plc.conc_energy_shift)
            energy >>= c->plc.conc_energy_shift - energy_shift;
        if (energy > c->plc.conc_energy) {
            i32 frac_Q24, gain_Q16, slope_Q16;
            int LZ = clz32(c->plc.conc_energy) - 1;
            c->plc.conc_energy = shl32(c->plc.conc_energy, LZ);
            energy >>= OC_MAX(24 - LZ, 0);
            frac_Q24 = c->plc.conc_energy / OC_MAX(energy, 1);
            gain_Q16 = shl32(sqrt_approx(frac_Q24), 4);
            slope_Q16 = ((1 << 16) - gain_Q16) / length;
            slope_Q16 = shl32(slope_Q16, 2);
            for (i = 0; i < length; i++) {
                frame[i] = (i16)smulwb(gain_Q16, frame[i]);
                gain_Q16 += slope_Q16;
                if (gain_Q16 > 1 << 16) break;
            }
        }
    }
    c->plc.last_frame_lost = 0;
}

/* silk_CNG silk.cpp:1342 (with silk_CNG_exc :1305) */
static void cng(chan_t *c, const ctrl_t *ct, i16 frame[], int length) {
    int i, j, subfr;
    if (c->fs_kHz != c->cng.fs_kHz) {
        cng_reset(c);
        c->cng.fs_kHz = c->fs_kHz;
    }
    if (c->lossCnt == 0 && c->prevSignalType == 0) {
        i32 max_Gain_Q16 = 0;
        for (i = 0; i < c->LPC_order; i++)
            c->cng.smth_NLSF_Q15[i] = (i16)(c->cng.smth_NLSF_Q15[i] + smulwb((i32)c->prevNLSF_Q15[i] - (i32)c->cng.smth_NLSF_Q15[i], 16348));
        subfr = 0;
        for (i = 0; i < c->nb_subfr; i++)
            if (ct->Gains_Q16[i] > max_Gain_Q16) {
                max_Gain_Q16 = ct->Gains_Q16[i];
                subfr = i;
            }
        memmove(&c->cng.exc_buf_Q14[c->subfr_length], c->cng.exc_buf_Q14, (c->nb_subfr - 1) * c->subfr_length * sizeof(i32));
        memcpy(c->cng.exc_buf_Q14, &c->exc_Q14[subfr * c->subfr_length], c->subfr_length * sizeof(i32));
        for (i = 0; i < c->nb_subfr; i++) c->cng.smth_Gain_Q16 += smulwb(ct->Gains_Q16[i] - c->cng.smth_Gain_Q16, 4634);
    }
    if (c->lossCnt) {
        i32 sig_Q14[MAX_FRAME + MAX_LPC], gain_Q16, gain_Q10, seed, exc_mask = 255; /* CNG_BUF_MASK_MAX */
        i16 A_Q12[MAX_LPC];
        gain_Q16 = smulww(c->plc.randScale_Q14, c->plc.prevGain_Q16[1]);
        if (gain_Q16 >= (1 << 21) || c->cng.smth_Gain_Q16 > (1 << 23)) {
            gain_Q16 = (gain_Q16 >> 16) * (gain_Q16 >> 16);
            gain_Q16 = subw((c->cng.smth_Gain_Q16 >> 16) * (c->cng.smth_Gain_Q16 >> 16), shl32(gain_Q16, 5));
            gain_Q16 = shl32(sqrt_approx(gain_Q16), 16);
        } else {
            gain_Q16 = smulww(gain_Q16, gain_Q16);
            gain_Q16 = subw(smulww(c->cng.smth_Gain_Q16, c->cng.smth_Gain_Q16), shl32(gain_Q16, 5));
            gain_Q16 = shl32(sqrt_approx(gain_Q16), 8);
        }
        gain_Q10 = gain_Q16 >> 6;
        while (exc_mask > length) exc_mask >>= 1;
        seed = c->cng.rand_seed;
        for (i = 0; i < length; i++) {
            seed = silk_rand(seed);
            sig_Q14[MAX_LPC + i] = c->cng.exc_buf_Q14[(seed >> 24) & exc_mask];
        }
        c->cng.rand_seed = seed;
        nlsf2a(A_Q12, c->cng.smth_NLSF_Q15, c->LPC_order);
        memcpy(sig_Q14, c->cng.synth_state, MAX_LPC * sizeof(i32));
        for (i = 0; i < length; i++) {
            i32 LPC_pred_Q10 = c->LPC_order >> 1;
            for (j = 0; j < c->LPC_order; j++) LPC_pred_Q10 = smlawb(LPC_pred_Q10, sig_Q14[MAX_LPC + i - j - 1], A_Q12[j]);
            sig_Q14[MAX_LPC + i] = add_sat32(sig_Q14[MAX_LPC + i], lshift_sat32(LPC_pred_Q10, 4));
            frame[i] = sat16((i32)frame[i] + sat16(rshift_round(smulww(sig_Q14[MAX_LPC + i], gain_Q10), 8)));
        }
        memcpy(c->cng.synth_state, &sig_Q14[length], MAX_LPC * sizeof(i32));
    } else
        memset(c->cng.synth_state, 0, c->LPC_order * sizeof(i32));
}

/* silk_decode_frame silk.cpp:1974.  lostFlag: 0 normal, 1 packet lost, 2 decode the LBRR (FEC) copy */
static void decode_frame(chan_t *c, ctrl_t *ct, oc_rc *rc, i16 pOut[], i32 *pN, int lostFlag, int condCoding) {
    i16 pulses[MAX_FRAME + 16];
    int L = c->frame_length, mv_len;
    ct->LTP_scale_Q14 = 0;
    if (lostFlag == 0 || (lostFlag == 2 && c->LBRR_flags[c->nFramesDecoded] == 1)) {
        decode_indices(c, rc, c->nFramesDecoded, lostFlag, condCoding);
        decode_pulses(rc, pulses, c->idx.signalType, c->idx.quantOffsetType, c->frame_length);
        decode_parameters(c, ct, condCoding);
        decode_core(c, ct, pOut, pulses);
        plc(c, ct, pOut, 0);
        c->lossCnt = 0;
        c->prevSignalType = c->idx.signalType;
        c->first_frame_after_reset = 0;
    } else {
        c->idx.signalType = (signed char)c->prevSignalType;
        plc(c, ct, pOut, 1);
    }
    mv_len = c->ltp_mem_length - c->frame_length;
    memmove(c->outBuf, &c->outBuf[c->frame_length], mv_len * sizeof(i16));
    memcpy(&c->outBuf[mv_len], pOut, c->frame_length * sizeof(i16));
    cng(c, ct, pOut, L);
    plc_glue_frames(c, pOut, L);
    c->lagPrev = ct->pitchL[c->nb_subfr - 1];
    *pN = L;
}

/* ---- resampler (silk.cpp:3451-3713), upsampling to 48 kHz only ------------------------------------------ */
static void resampler_init(resamp_t *r, i32 Fs_Hz_in, i32 Fs_Hz_out) {
    int in_id = ((Fs_Hz_in >> 12) - (Fs_Hz_in > 16000)) - 1; /* rateID silk.h:397 for <= 16 kHz */
    memset(r, 0, sizeof(*r));
    r->inputDelay = rom_silk_delay_dec[in_id * 5 + 4]; /* column of 48 kHz */
    r->Fs_in_kHz = Fs_Hz_in / 1000;
    r->Fs_out_kHz = Fs_Hz_out / 1000;
    r->batchSize = r->Fs_in_kHz * 10;
    /* Fs_out > Fs_in and Fs_out != 2*Fs_in: the IIR_FIR path with 2x pre-upsampling */
    r->invRatio_Q16 = shl32(shl32(Fs_Hz_in, 14 + 1) / Fs_Hz_out, 2);
    while (smulww(r->invRatio_Q16, Fs_Hz_out) < shl32(Fs_Hz_in, 1)) r->invRatio_Q16++;
}

/* silk_resampler_private_up2_HQ silk.cpp:3515 */
static void up2_hq(i32 *S, i16 *out, const i16 *in, int len) {
    int k;
    for (k = 0; k < len; k++) {
        i32 in32 = shl32((i32)in[k], 10), Y, X, o1, o2;
        Y = in32 - S[0]; X = smulwb(Y, rom_silk_up2_hq0[0]); o1 = S[0] + X; S[0] = in32 + X;
        Y = o1 - S[1]; X = smulwb(Y, rom_silk_up2_hq0[1]); o2 = S[1] + X; S[1] = o1 + X;
        Y = o2 - S[2]; X = smlawb(Y, Y, rom_silk_up2_hq0[2]); o1 = S[2] + X; S[2] = o2 + X;
        out[2 * k] = sat16(rshift_round(o1, 10));
        Y = in32 - S[3]; X = smulwb(Y, rom_silk_up2_hq1[0]); o1 = S[3] + X; S[3] = in32 + X;
        Y = o1 - S[4]; X = smulwb(Y, rom_silk_up2_hq1[1]); o2 = S[4] + X; S[4] = o1 + X;
        Y = o2 - S[5]; X = smlawb(Y, Y, rom_silk_up2_hq1[2]); o1 = S[5] + X; S[5] = o2 + X;
        out[2 * k + 1] = sat16(rshift_round(o1, 10));
    }
}

/* silk_resampler_private_IIR_FIR silk.cpp:3475 (+ _INTERPOL :3451) */
static i16 *iir_fir(resamp_t *r, i16 *out, const i16 *in, i32 inLen) {
    i16 buf[2 * 160 + 8];
    i32 nSamplesIn, max_index_Q16, index_Q16;
    memcpy(buf, r->sFIR, 8 * sizeof(i16));
    for (;;) {
        nSamplesIn = OC_MIN(inLen, r->batchSize);
        up2_hq(r->sIIR, &buf[8], in, nSamplesIn);
        max_index_Q16 = shl32(nSamplesIn, 17);
        for (index_Q16 = 0; index_Q16 < max_index_Q16; index_Q16 += r->invRatio_Q16) {
            int t = smulwb(index_Q16 & 0xFFFF, 12);
            const i16 *b = &buf[index_Q16 >> 16], *f0 = &rom_silk_frac_fir12[4 * t], *f1 = &rom_silk_frac_fir12[4 * (11 - t)];
            i32 res = smulbb(b[0], f0[0]);
            res = smlabb(res, b[1], f0[1]);
            res = smlabb(res, b[2], f0[2]);
            res = smlabb(res, b[3], f0[3]);
            res = smlabb(res, b[4], f1[3]);
            res = smlabb(res, b[5], f1[2]);
            res = smlabb(res, b[6], f1[1]);
            res = smlabb(res, b[7], f1[0]);
            *out++ = sat16(rshift_round(res, 15));
        }
        in += nSamplesIn;
        inLen -= nSamplesIn;
        if (inLen > 0)
            memcpy(buf, &buf[nSamplesIn << 1], 8 * sizeof(i16));
        else
            break;
    }
    memcpy(r->sFIR, &buf[nSamplesIn << 1], 8 * sizeof(i16));
    return out;
}

/* silk_resampler silk.cpp:3676 */
static void resample(resamp_t *r, i16 *out, const i16 *in, i32 inLen) {
    int nSamples = r->Fs_in_kHz - r->inputDelay;
    memcpy(&r->delayBuf[r->inputDelay], in, nSamples * sizeof(i16));
    iir_fir(r, out, r->delayBuf, r->Fs_in_kHz);
    iir_fir(r, &out[r->Fs_out_kHz], &in[nSamples], inLen - r->Fs_in_kHz);
    memcpy(r->delayBuf, &in[inLen - r->inputDelay], r->inputDelay * sizeof(i16));
}

/* silk_decoder_set_fs silk.cpp:978 (API rate fixed at 48 kHz) */
static void set_fs(chan_t *c, resamp_t *r, int fs_kHz) {
    int frame_length;
    c->subfr_length = 5 * fs_kHz;
    frame_length = c->nb_subfr * c->subfr_length;
    if (c->fs_kHz != fs_kHz || c->fs_API_hz != 48000) {
        resampler_init(r, fs_kHz * 1000, 48000);
        c->fs_API_hz = 48000;
    }
    if (c->fs_kHz != fs_kHz || frame_length != c->frame_length) {
        if (c->fs_kHz != fs_kHz) {
            c->ltp_mem_length = 20 * fs_kHz;
            c->LPC_order = (fs_kHz == 8 || fs_kHz == 12) ? 10 : 16;
            c->first_frame_after_reset = 1;
            c->lagPrev = 100;
            c->LastGainIndex = 10;
            c->prevSignalType = 0;
            memset(c->outBuf, 0, sizeof(c->outBuf));
            memset(c->sLPC_Q14_buf, 0, sizeof(c->sLPC_Q14_buf));
        }
        c->fs_kHz = fs_kHz;
        c->frame_length = frame_length;
    }
}

/* silk_stereo_decode_pred silk.cpp:592 */
static void stereo_decode_pred(oc_rc *rc, i32 pred_Q13[2]) {
    int n, ix[2][3];
    n = oc_rc_icdf(rc, rom_silk_stereo_joint_icdf, 8);
    ix[0][2] = n / 5;
    ix[1][2] = n - 5 * ix[0][2];
    for (n = 0; n < 2; n++) {
        ix[n][0] = oc_rc_icdf(rc, rom_silk_uniform3_icdf, 8);
        ix[n][1] = oc_rc_icdf(rc, rom_silk_uniform5_icdf, 8);
    }
    for (n = 0; n < 2; n++) {
        i32 low_Q13, step_Q13;
        ix[n][0] += 3 * ix[n][2];
        low_Q13 = rom_silk_stereo_pred_q13[ix[n][0]];
        step_Q13 = smulwb(rom_silk_stereo_pred_q13[ix[n][0] + 1] - low_Q13, 6554);
        pred_Q13[n] = smlabb(low_Q13, step_Q13, 2 * ix[n][1] + 1);
    }
    pred_Q13[0] -= pred_Q13[1];
}

/* silk_stereo_MS_to_LR silk.cpp:4028 */
static void ms_to_lr(oc_silk *s, i16 x1[], i16 x2[], const i32 pred_Q13[], int fs_kHz, int frame_length) {
    int n;
    i32 denom_Q16, delta0_Q13, delta1_Q13, sum, diff, pred0_Q13, pred1_Q13;
    memcpy(x1, s->sMid, 2 * sizeof(i16));
    memcpy(x2, s->sSide, 2 * sizeof(i16));
    memcpy(s->sMid, &x1[frame_length], 2 * sizeof(i16));
    memcpy(s->sSide, &x2[frame_length], 2 * sizeof(i16));
    pred0_Q13 = s->pred_prev_Q13[0];
    pred1_Q13 = s->pred_prev_Q13[1];
    denom_Q16 = (1 << 16) / (8 * fs_kHz);
    delta0_Q13 = rshift_round(smulbb(pred_Q13[0] - s->pred_prev_Q13[0], denom_Q16), 16);
    delta1_Q13 = rshift_round(smulbb(pred_Q13[1] - s->pred_prev_Q13[1], denom_Q16), 16);
    for (n = 0; n < 8 * fs_kHz; n++) {
        pred0_Q13 += delta0_Q13;
        pred1_Q13 += delta1_Q13;
        sum = shl32((x1[n] + x1[n + 2]) + shl32(x1[n + 1], 1), 9);
        sum = smlawb(shl32((i32)x2[n + 1], 8), sum, pred0_Q13);
        sum = smlawb(sum, shl32((i32)x1[n + 1], 11), pred1_Q13);
        x2[n + 1] = sat16(rshift_round(sum, 8));
    }
    pred0_Q13 = pred_Q13[0];
    pred1_Q13 = pred_Q13[1];
    for (n = 8 * fs_kHz; n < frame_length; n++) {
        sum = shl32((x1[n] + x1[n + 2]) + shl32(x1[n + 1], 1), 9);
        sum = smlawb(shl32((i32)x2[n + 1], 8), sum, pred0_Q13);
        sum = smlawb(sum, shl32((i32)x1[n + 1], 11), pred1_Q13);
        x2[n + 1] = sat16(rshift_round(sum, 8));
    }
    s->pred_prev_Q13[0] = (i16)pred_Q13[0];
    s->pred_prev_Q13[1] = (i16)pred_Q13[1];
    for (n = 0; n < frame_length; n++) {
        sum = x1[n + 1] + (i32)x2[n + 1];
        diff = x1[n + 1] - (i32)x2[n + 1];
        x1[n + 1] = sat16(sum);
        x2[n + 1] = sat16(diff);
    }
}

/* silk_Decode silk.cpp:1481 with lostFlag = FLAG_DECODE_NORMAL, payloadSize_ms = 20, API 48 kHz,
 * nChannelsAPI = nChannelsInternal = channels (src/opus_decoder.cpp:167-169, :203) */
/* silk_LBRR_flags_2_iCDF / _3_iCDF (RFC 6716 table 4: PDFs {0,53,53,150} and {0,41,20,29,41,15,28,82}) -- only packets of
 * two / three internal frames read them, which the reference never decodes as such */
static const u8 lbrr_flags_2_icdf[3] = {203, 150, 0};
static const u8 lbrr_flags_3_icdf[7] = {215, 195, 166, 125, 110, 82, 0};

int oc_silk_decode(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, i16 *out, i32 *n_out) {
    return oc_silk_decode_ms(s, rc, channels, internal_hz, first, 20, out, n_out);
}

int oc_silk_decode_ms(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, int payload_ms, i16 *out, i32 *n_out) {
    return oc_silk_decode_ex(s, rc, channels, internal_hz, first, payload_ms, 0, out, n_out);
}

/* lostFlag 1 (packet lost: conceal one frame of payload_ms = 10 or 20; the internal rate stays what it is when internal_hz
 * is 0) and 2 (decode the LBRR copy of the frame where there is one, conceal otherwise) follow silk.cpp:1481-1779 */
int oc_silk_decode_ex(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, int payload_ms, int lostFlag, i16 *out,
                      i32 *n_out) {
    i16 tmp[2][MAX_FRAME + 2 + 16], rs_out[960];
    i32 MS_pred_Q13[2] = {0, 0}, nSamplesOutDec = 0;
    int n, i, decode_only_middle = 0, has_side;
    if (first)
        for (n = 0; n < channels; n++) s->ch[n].nFramesDecoded = 0;
    if (channels > s->nChannelsInternal) chan_init(&s->ch[1]);
    if (s->ch[0].nFramesDecoded == 0) {
        for (n = 0; n < channels; n++) {
            int fs_kHz_dec = internal_hz ? (internal_hz >> 10) + 1 : s->ch[0].fs_kHz;
            /* silk.cpp:1522-1540 with the payload duration the reference pins to 20 ms */
            s->ch[n].nFramesPerPacket = payload_ms == 40 ? 2 : payload_ms == 60 ? 3 : 1;
            s->ch[n].nb_subfr = payload_ms == 10 ? 2 : 4;
            if (fs_kHz_dec != 8 && fs_kHz_dec != 12 && fs_kHz_dec != 16) return -200;
            set_fs(&s->ch[n], &s->rs[n], fs_kHz_dec);
        }
    }
    if (channels == 2 && (s->nChannelsAPI == 1 || s->nChannelsInternal == 1)) {
        memset(s->pred_prev_Q13, 0, sizeof(s->pred_prev_Q13));
        memset(s->sSide, 0, sizeof(s->sSide));
        /* the reference's resampler-state copy here is a self-copy (Q10) */
    }
    s->nChannelsAPI = channels;
    s->nChannelsInternal = channels;

    if (lostFlag != 1 && s->ch[0].nFramesDecoded == 0) {
        for (n = 0; n < channels; n++) {
            for (i = 0; i < s->ch[n].nFramesPerPacket; i++) s->ch[n].VAD_flags[i] = oc_rc_bit_logp(rc, 1);
            s->ch[n].LBRR_flag = oc_rc_bit_logp(rc, 1);
        }
        for (n = 0; n < channels; n++) {
            memset(s->ch[n].LBRR_flags, 0, sizeof(s->ch[n].LBRR_flags));
            if (s->ch[n].LBRR_flag) {
                if (s->ch[n].nFramesPerPacket == 1)
                    s->ch[n].LBRR_flags[0] = 1;
                else { /* silk.cpp:1580-1586 */
                    int sym = oc_rc_icdf(rc, s->ch[n].nFramesPerPacket == 2 ? lbrr_flags_2_icdf : lbrr_flags_3_icdf, 8) + 1;
                    for (i = 0; i < s->ch[n].nFramesPerPacket; i++) s->ch[n].LBRR_flags[i] = (sym >> i) & 1;
                }
            }
        }
        /* regular decoding: read past the LBRR data (it still updates the entropy-coding context) */
        if (lostFlag == 0)
            for (i = 0; i < s->ch[0].nFramesPerPacket; i++) {
                for (n = 0; n < channels; n++) {
                    if (s->ch[n].LBRR_flags[i]) {
                        i16 pulses[MAX_FRAME + 16];
                        int condCoding;
                        if (channels == 2 && n == 0) {
                            stereo_decode_pred(rc, MS_pred_Q13);
                            if (s->ch[1].LBRR_flags[i] == 0) decode_only_middle = oc_rc_icdf(rc, rom_silk_mid_only_icdf, 8);
                        }
                        condCoding = (i > 0 && s->ch[n].LBRR_flags[i - 1]) ? 2 : 0;
                        decode_indices(&s->ch[n], rc, i, 1, condCoding);
                        decode_pulses(rc, pulses, s->ch[n].idx.signalType, s->ch[n].idx.quantOffsetType, s->ch[n].frame_length);
                    }
                }
            }
    }
    if (channels == 2) { /* silk.cpp:1620-1637 */
        if (lostFlag == 0 || (lostFlag == 2 && s->ch[0].LBRR_flags[s->ch[0].nFramesDecoded] == 1)) {
            stereo_decode_pred(rc, MS_pred_Q13);
            if ((lostFlag == 0 && s->ch[1].VAD_flags[s->ch[0].nFramesDecoded] == 0) ||
                (lostFlag == 2 && s->ch[1].LBRR_flags[s->ch[0].nFramesDecoded] == 0))
                decode_only_middle = oc_rc_icdf(rc, rom_silk_mid_only_icdf, 8);
            else
                decode_only_middle = 0;
        } else
            for (n = 0; n < 2; n++) MS_pred_Q13[n] = s->pred_prev_Q13[n];
    }
    if (channels == 2 && decode_only_middle == 0 && s->prev_decode_only_middle == 1) {
        memset(s->ch[1].outBuf, 0, sizeof(s->ch[1].outBuf));
        memset(s->ch[1].sLPC_Q14_buf, 0, sizeof(s->ch[1].sLPC_Q14_buf));
        s->ch[1].lagPrev = 100;
        s->ch[1].LastGainIndex = 10;
        s->ch[1].prevSignalType = 0;
        s->ch[1].first_frame_after_reset = 1;
    }
    memset(tmp, 0, sizeof(tmp));
    if (lostFlag == 0)
        has_side = !decode_only_middle;
    else /* silk.cpp:1667-1671 */
        has_side = !s->prev_decode_only_middle || (channels == 2 && lostFlag == 2 && s->ch[1].LBRR_flags[s->ch[1].nFramesDecoded] == 1);
    if (g_silk_taps.on) g_silk_taps.valid[0] = g_silk_taps.valid[1] = 0;
    for (n = 0; n < channels; n++) {
        if (n == 0 || has_side) {
            int FrameIndex = s->ch[0].nFramesDecoded - n, condCoding;
            if (FrameIndex <= 0)
                condCoding = 0;
            else if (lostFlag == 2)
                condCoding = s->ch[n].LBRR_flags[FrameIndex - 1] ? 2 : 0;
            else if (n > 0 && s->prev_decode_only_middle)
                condCoding = 1;
            else
                condCoding = 2;
            decode_frame(&s->ch[n], &s->ctrl, rc, &tmp[n][2], &nSamplesOutDec, lostFlag, condCoding);
            if (g_silk_taps.on) {
                g_silk_taps.valid[n] = 1;
                g_silk_taps.signalType[n] = s->ch[n].idx.signalType;
                g_silk_taps.quantOffsetType[n] = s->ch[n].idx.quantOffsetType;
                g_silk_taps.frame_length[n] = s->ch[n].frame_length;
                g_silk_taps.lpc_order[n] = s->ch[n].LPC_order;
                g_silk_taps.ctrl[n] = s->ctrl;
                memcpy(g_silk_taps.xq[n], &tmp[n][2], sizeof(i16) * (size_t)s->ch[n].frame_length);
            }
        } else
            memset(&tmp[n][2], 0, nSamplesOutDec * sizeof(i16));
        s->ch[n].nFramesDecoded++;
    }
    if (channels == 2)
        ms_to_lr(s, tmp[0], tmp[1], MS_pred_Q13, s->ch[0].fs_kHz, nSamplesOutDec);
    else {
        memcpy(tmp[0], s->sMid, 2 * sizeof(i16));
        memcpy(s->sMid, &tmp[0][nSamplesOutDec], 2 * sizeof(i16));
    }
    *n_out = nSamplesOutDec * 48000 / (s->ch[0].fs_kHz * 1000);
    for (n = 0; n < channels; n++) {
        resample(&s->rs[n], rs_out, &tmp[n][1], nSamplesOutDec);
        if (channels == 2)
            for (i = 0; i < *n_out; i++) out[n + 2 * i] = rs_out[i];
        else
            memcpy(out, rs_out, *n_out * sizeof(i16));
    }
    /* the pitch lag at 48 kHz, for OPUS_GET_PITCH (silk.cpp:1764-1769) */
    s->prev_pitch_lag = s->ch[0].prevSignalType == 2 ? s->ch[0].lagPrev * (s->ch[0].fs_kHz == 8 ? 6 : s->ch[0].fs_kHz == 12 ? 4 : 3) : 0;
    if (lostFlag == 1) /* silk.cpp:1772-1776: no gain clamping across a loss */
        for (i = 0; i < s->nChannelsInternal; i++) s->ch[i].LastGainIndex = 10;
    else
        s->prev_decode_only_middle = decode_only_middle;
    return 0;
}
i32 oc_silk_prev_pitch_lag(const oc_silk *s) { return s->prev_pitch_lag; }
