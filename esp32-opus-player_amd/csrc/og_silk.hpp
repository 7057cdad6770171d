// og_silk.hpp -- SILK frame decoder, one frame (both channels) per wavefront.
//
// Behaviour reproduced: silk_Decode as the reference drives it (src/silk.cpp:1481; lostFlag == 0,
// 20 ms payload, API rate 48 kHz, nChannelsAPI == nChannelsInternal == packet channels, src/opus_decoder.cpp:167-203).
//
// Mapping onto the wave (one frame per wave):
//   [scalar, wave-uniform]   only without the split path (OPUSGPU_SPLIT=0): VAD/LBRR flags, LBRR skip-decode, stereo
//                            predictor, side information, shell/sign pulse decode (inverse-CDF symbols searched by the
//                            whole wave at once: og_range.hpp), parameter dequantisation, NLSF -> LPC.  On the split
//                            path all of that is done one frame per LANE by k_silk_parse (og_silk_parse.hpp, and
//                            silk_params_lane below) and arrives as a SilkRec.
//   [16-lane row per channel] silk_decode_core: excitation, LTP and the order-10/16 LPC synthesis recurrence
//                            (saturating, hence serial in time): lane j of the row holds state sample i-1-j and
//                            coefficient j, the prediction is one multiply + a DPP row all-reduce.
//   [lane = all-pass section] the 2x up-sampler as a systolic array (2 phases x 3 sections per channel).
//   [lane-parallel]          re-whitening FIR, mid/side -> left/right, the 8-tap 12-phase FIR interpolation to 48 kHz.
// Scratch lives in its own LDS object (SilkLds); the resampler buffers overlay the synthesis buffers.
#pragma once
#include "og_celt_math.hpp"
#include "og_silk_parse.hpp"

namespace og {

constexpr int SILK_MAX_LPC = 16;
constexpr int SILK_MAX_FRAME = 320;
// The longest frame the synthesis' LDS buffers hold.  og_silk_nb.hip builds the synthesis kernel of narrowband frames with 160:
// 6.7 KB instead of 10.2 (six instead of eight LDS granules: five frames per SIMD instead of four -- the kernel waits on its frames'
// serial chains, so frames in flight are what it is short of), and per-subframe loops of three instead of five samples per lane.
#ifdef OG_SILK_LDS_FRAME
constexpr int SILK_LDS_FRAME = OG_SILK_LDS_FRAME;
#else
constexpr int SILK_LDS_FRAME = SILK_MAX_FRAME;
#endif
static_assert(SILK_LDS_FRAME % 80 == 0 && SILK_LDS_FRAME <= SILK_MAX_FRAME, "20 ms at 8, 12 or 16 kHz");

struct SilkCtrl { // silk_decoder_control_t (src/silk.h:747) + the frame's side information (:588)
    i32 pitchL[4], Gains_Q16[4];
    i16 PredCoef_Q12[2][SILK_MAX_LPC];
    i16 LTPCoef_Q14[20];
    i32 LTP_scale_Q14;
    i32 signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed, lagIndex, contourIndex, PERIndex, LTP_scaleIndex;
    i32 GainsIndices[4], LTPIndex[4], NLSFIndices[SILK_MAX_LPC + 1];
    i32 coded; // channel has a coded frame this step
};

#ifndef OG_SILK_TIGHT
struct SilkLds {
    i16 xq[2][SILK_LDS_FRAME + 8];      // [0..2) look-back slots, frame at +2 (samplesOut1_tmp, silk.cpp:1657)
    // Two phases share these bytes: the synthesis recurrence (LTP state, whitened history, staged outBuf) and, once
    // that is done, the resampler (48 kHz PCM of both channels, 2x up-sampled signals).  The up-sampler's staged
    // 32-bit input lives in sLTP_Q15[ch][0..frame), i.e. under `pcm`, and is dead before the FIR writes `pcm`.
    union {
        struct {
            i32 sLTP_Q15[2][2 * SILK_LDS_FRAME];
            i16 sLTP[2][SILK_LDS_FRAME];
            i16 hist[2][SILK_LDS_FRAME + SILK_LDS_FRAME / 2]; // outBuf staged from HBM (+ 2 subframes for the mid-frame re-whitening)
        } core;
        struct {
            i16 pcm[1920];                          // SILK output at 48 kHz, interleaved over the packet's channels
            // (the rest of sLTP_Q15 where it is longer than `pcm`: the up-sampler's 32-bit rows end there, see silk_up2_rows)
            i16 raw_tail[16 * SILK_LDS_FRAME > 3840 ? (16 * SILK_LDS_FRAME - 3840) / 2 : 0];
            i16 up[2][8 + 2 * SILK_LDS_FRAME + 8];  // FIR history + 2x up-sampled frame
            i32 sink[4][2][4];                      // where the up-sampler's inner sections "store" (silk_up2_rows)
            u32 taps[48];                           // rom_silk_fir12_taps8 (a lane's phase is its own: a table read per output)
        } out;
    } u;
    SilkCtrl ctrl[2];
};
static_assert(offsetof(SilkLds, u.out.up) - offsetof(SilkLds, u.out.pcm) >= sizeof(((SilkLds *)0)->u.core.sLTP_Q15),
              "the resampler's 16-bit rows start behind the up-sampler's 32-bit rows (sLTP_Q15), which lie under `pcm` (and `raw_tail`)");
#else
// OG_SILK_TIGHT (og_silk_synth.hip, og_silk_nb.hip: the split path's synthesis kernels -- reference mode, one 20 ms frame from the parse
// kernel's record, no loss path): the same working set in 8,936 instead of 10,184 bytes, SEVEN LDS granules instead of eight (the
// narrowband kernel: 6,056 instead of 6,728, five instead of six).  The kernel waits on its frames' serial chains, LDS is what limits the
// frames in flight, and a granule is worth 3 - 5 % of a step.  What is different:
//   * ONE row per channel for the output history and the frame's output, the frame right behind the history (at [ltp_mem]): the
//     mid-frame re-whitening reads the frame's first two subframes where they lie (the copy behind the history is gone), and the
//     look-back slots of the stereo un-mixing are the history's last two entries, dead by then;
//   * once the up-sampler has staged the frame as its 32-bit input, that row is the FIR's input row (`up`: 8 + 2 x frame + 8);
//   * the re-whitening scales its outputs into the LTP state as it makes them (no whitened-history buffer; the subframe's
//     residuals, which lay over it, have 320 bytes of their own per channel).
// The arrays of length zero are named by code that never runs from this layout (the loss path, the one-lane forms).
struct SilkLds {
    i16 hx[2][2 * SILK_LDS_FRAME + 16];
    union {
        struct {
            i32 sLTP_Q15[2][2 * SILK_LDS_FRAME];
            i32 resb[2][SILK_LDS_FRAME / 4];
            i16 sLTP[2][0], hist[2][0];
        } core;
        struct {
            i16 pcm[1920];
            i16 raw_tail[16 * SILK_LDS_FRAME > 3840 ? (16 * SILK_LDS_FRAME - 3840) / 2 : 0];
            i32 sink[4][2][4];
            u32 taps[48];
            i16 up[2][0];
        } out;
    } u;
    SilkCtrl ctrl[2];
    i16 xq[2][0];
};
static_assert(offsetof(SilkLds, u.out.sink) - offsetof(SilkLds, u.out.pcm) >= sizeof(((SilkLds *)0)->u.core.sLTP_Q15),
              "the up-sampler's sink and the FIR's taps lie behind its 32-bit rows (sLTP_Q15), which lie under `pcm` (and `raw_tail`)");
static_assert(sizeof(SilkLds) <= (SILK_LDS_FRAME == 320 ? 8960 : 6400), "seven LDS granules (narrowband: five)");
#endif
// SILK's working set is its own LDS object: only the kernels that run SILK pay for it.
OG_LDS SilkLds g_silk_lds;
OG_DEV SilkLds &SL() { return g_silk_lds; }
// A channel's rows by what they hold (the two layouts above place them differently).  `ltp_mem`: 20 ms at the frame's rate -- the
// frame's length in the tight layout, which only sees 20 ms frames.
//   silk_xq_row    the frame's output: [0..2) look-back slots of the stereo un-mixing, the frame at +2
//   silk_hist_row  the last 20 ms of the channel's output (outBuf), staged for the re-whitening
//   silk_up_row    the FIR interpolator's input: 8 samples of history, the 2x up-sampled frame
#ifdef OG_SILK_TIGHT
OG_DEV i16 *silk_xq_row(int ch, int ltp_mem) { return &SL().hx[ch][ltp_mem - 2]; }
OG_DEV i16 *silk_hist_row(int ch) { return SL().hx[ch]; }
OG_DEV i16 *silk_up_row(int ch) { return SL().hx[ch]; }
#else
OG_DEV i16 *silk_xq_row(int ch, int) { return SL().xq[ch]; }
OG_DEV i16 *silk_hist_row(int ch) { return SL().u.core.hist[ch]; }
OG_DEV i16 *silk_up_row(int ch) { return SL().u.out.up[ch]; }
#endif
// What only the wave-uniform ENTROPY half needs -- the pulse row it decodes into, the scratch of the pulse decoder and of
// the NLSF -> LPC conversion -- is an object of its own: the synthesis kernel of the split path (silk_decode_20ms<true>:
// everything from the parse kernel's record, pulses read where they lie in HBM) never names it, and its LDS footprint
// drops from 12,160 to 10,152 bytes = 8 instead of 10 LDS granules of 1280 bytes: 16 instead of 12 workgroups per CU.
struct SilkWaveParseLds {
    i16 pulses[2][SILK_MAX_FRAME + 16];
    i32 sum_pulses[20], nLshifts[20];
    i32 VAD_flags[2], LBRR_flag[2];
    i16 nlsf[SILK_MAX_LPC], nlsf0[SILK_MAX_LPC], ec_ix[SILK_MAX_LPC], res_Q10[SILK_MAX_LPC];
    i32 pred_Q8[SILK_MAX_LPC];
    i32 cosLSF[SILK_MAX_LPC], P[SILK_MAX_LPC / 2 + 1], Q[SILK_MAX_LPC / 2 + 1], a32[SILK_MAX_LPC], Atmp[SILK_MAX_LPC];
};
// It has no bytes of its own: whenever the wave-uniform SILK parse (or the SILK concealment's LPC scratch) runs, the CELT working
// set's spectrum X is dead -- a frame's SILK layer comes before its CELT layer, a concealment's SILK part before its CELT part, and
// what SILK hands on waits in SilkLds -- so the object lies over the first 2 KB of X.  (k_decode_rfc: 22.5 -> 20.5 KB of LDS, eight
// instead of seven workgroups per CU; k_decode_step likewise.)
static_assert(sizeof(SilkWaveParseLds) <= sizeof(i16) * 1920 && alignof(SilkWaveParseLds) <= 16, "the wave parse's scratch lies over X");
OG_DEV SilkWaveParseLds &PW() { return *reinterpret_cast<SilkWaveParseLds *>(&S.v[V_X]); }

// ---- state ------------------------------------------------------------------------------------------
// `lc`: the channel's loss-concealment state (RFC mode; reference-mode callers pass none and never look at it)
OG_DEV void silk_chan_init(SilkChannel *c, SilkLossChannel *lc = nullptr) { // silk_init_decoder silk.cpp:2192
    u32 *w = reinterpret_cast<u32 *>(c);
    OG_FOR_LANES(i, (int)(sizeof(SilkChannel) / 4)) w[i] = 0;
    if (lc) { // silk_CNG_Reset / silk_PLC_Reset run again, with the real rate, before anything reads what they set
        u32 *wl = reinterpret_cast<u32 *>(lc);
        OG_FOR_LANES(i, (int)(sizeof(SilkLossChannel) / 4)) wl[i] = 0;
    }
    OG_SYNC();
    if (OG_LANE == 0) {
        c->first_frame_after_reset = 1;
        c->prev_gain_Q16 = 65536;
    }
    OG_SYNC();
}

// silk_InitDecoder silk.cpp:1792.  The channel init clears fs_kHz, so the resampler part of the record is
// re-initialised by the next frame exactly when the reference re-initialises its (separate) resampler state.
OG_DEV void silk_init_state(SilkState *s, LossState *loss = nullptr) {
    silk_chan_init(&s->ch[0], loss ? &loss->silk[0] : nullptr);
    silk_chan_init(&s->ch[1], loss ? &loss->silk[1] : nullptr);
    if (OG_LANE == 0) {
        s->pred_prev_Q13[0] = s->pred_prev_Q13[1] = 0;
        s->sMid[0] = s->sMid[1] = 0;
        s->sSide[0] = s->sSide[1] = 0;
        s->prev_decode_only_middle = 0;
    }
    OG_SYNC();
}

// ---- scalar helpers (src/silk.h:913-996, silk.cpp:2248) -------------------------------------------------
OG_DEV i32 iabs(i32 a) { return a < 0 ? -a : a; }

OG_DEV i32 silk_div32_varQ(i32 a32, i32 b32, int Qres) {
    int a_headrm = clz32(iabs(a32)) - 1, b_headrm = clz32(iabs(b32)) - 1;
    i32 a32_nrm = shl32(a32, a_headrm), b32_nrm = shl32(b32, b_headrm);
    i32 b32_inv = (2147483647 >> 2) / (b32_nrm >> 16);
    i32 result = smulwb(a32_nrm, b32_inv);
    a32_nrm = subw(a32_nrm, shl32(smmul(b32_nrm, result), 3));
    result = smlawb(result, a32_nrm, b32_inv);
    int lshift = 29 + a_headrm - b_headrm - Qres;
    if (lshift < 0) return lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

OG_DEV i32 silk_inverse32_varQ(i32 b32, int Qres) {
    int b_headrm = clz32(iabs(b32)) - 1;
    i32 b32_nrm = shl32(b32, b_headrm);
    i32 b32_inv = (2147483647 >> 2) / (b32_nrm >> 16);
    i32 result = shl32(b32_inv, 16);
    i32 err_Q32 = shl32((1 << 29) - smulwb(b32_nrm, b32_inv), 3);
    result = addw(result, smulww(err_Q32, b32_inv));
    int lshift = 61 - b_headrm - Qres;
    if (lshift <= 0) return lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

OG_DEV i32 silk_log2lin(i32 inLog_Q7) {
    if (inLog_Q7 < 0) return 0;
    if (inLog_Q7 >= 3967) return 2147483647;
    i32 out = shl32(1, inLog_Q7 >> 7), frac_Q7 = inLog_Q7 & 0x7F;
    i32 t = smlawb(frac_Q7, smulbb(frac_Q7, 128 - frac_Q7), -174);
    if (inLog_Q7 < 2048) return out + ((out * t) >> 7);
    return out + (out >> 7) * t;
}

// The byte tables of the WAVE-UNIFORM entropy decoder (below) and of the codebooks: in ROM -- or, for a translation unit that says so
// (OG_SILK_TABLES_LDS: og_rfc.hip), in an LDS copy of rom_silk_u8_blob that lies over the CELT working set's folding-history rows,
// which are dead whenever SILK's entropy half runs (silk_wave_tab_load before it).  The wave-uniform decoder reads a table entry per
// lane and symbol, a dependent read each: from ROM that is a trip through the vector memory path per symbol, 1,100 - 3,300 of them
// per packet, and it was what RFC mode's step consisted of.
#if defined(OG_SILK_TABLES_LDS) && !defined(OG_HOST_EMUL)
static_assert((V_TOTAL - V_NORM) * 2 >= SILK_BLOB_SIZE + 64, "the table copy (and the 64 entries a symbol's lanes read) fits behind X");
OG_DEV const u8 *silk_wave_tab() { return reinterpret_cast<const u8 *>(&S.v[V_NORM]); }
OG_DEV void silk_wave_tab_load() { // the whole wave; ends with a sync
    OG_SYNC();
    const u32 *src = reinterpret_cast<const u32 *>(rom_silk_u8_blob);
    u32 *dst = reinterpret_cast<u32 *>(&S.v[V_NORM]);
    OG_FOR_LANES(i, SILK_BLOB_SIZE / 4) dst[i] = src[i];
    OG_SYNC();
}
#define SILK_TAB(name) (silk_wave_tab() + SILK_BLOB_##name)
#else
OG_DEV void silk_wave_tab_load() {}
#define SILK_TAB(name) rom_silk_##name
#endif

// ---- codebooks (silk_NLSF_CB_struct instances silk.cpp:384-427) -------------------------------------------
struct NlsfCb {
    int order;
    i32 quantStepSize_Q16;
    const u8 *CB1_NLSF_Q8, *CB1_iCDF, *pred_Q8, *ec_sel, *ec_iCDF;
    const i32 *CB1_Wght_Q9, *deltaMin_Q15;
};
OG_DEV NlsfCb nlsf_cb(int wb) {
    NlsfCb cb;
    if (wb) {
        cb.order = 16;
        cb.quantStepSize_Q16 = 9830;
        cb.CB1_NLSF_Q8 = SILK_TAB(wb_cb1_q8);
        cb.CB1_Wght_Q9 = rom_silk_wb_cb1_wght_q9;
        cb.CB1_iCDF = SILK_TAB(wb_cb1_icdf);
        cb.pred_Q8 = SILK_TAB(wb_pred_q8);
        cb.ec_sel = SILK_TAB(wb_cb2_select);
        cb.ec_iCDF = SILK_TAB(wb_cb2_icdf);
        cb.deltaMin_Q15 = rom_silk_wb_delta_min_q15;
    } else {
        cb.order = 10;
        cb.quantStepSize_Q16 = 11796;
        cb.CB1_NLSF_Q8 = SILK_TAB(nb_cb1_q8);
        cb.CB1_Wght_Q9 = rom_silk_nb_cb1_wght_q9;
        cb.CB1_iCDF = SILK_TAB(nb_cb1_icdf);
        cb.pred_Q8 = SILK_TAB(nb_pred_q8);
        cb.ec_sel = SILK_TAB(nb_cb2_select);
        cb.ec_iCDF = SILK_TAB(nb_cb2_icdf);
        cb.deltaMin_Q15 = rom_silk_nb_delta_min_q15;
    }
    return cb;
}

OG_DEV void nlsf_unpack(const NlsfCb &cb, int CB1_index) { // silk_NLSF_unpack silk.cpp:2762 -> PW().ec_ix / pred_Q8
    SilkWaveParseLds &L = PW();
    const u8 *sel = &cb.ec_sel[CB1_index * cb.order / 2];
    for (int i = 0; i < cb.order; i += 2) {
        int entry = *sel++;
        L.ec_ix[i] = (i16)(((entry >> 1) & 7) * 9);
        L.pred_Q8[i] = cb.pred_Q8[i + (entry & 1) * (cb.order - 1)];
        L.ec_ix[i + 1] = (i16)(((entry >> 5) & 7) * 9);
        L.pred_Q8[i + 1] = cb.pred_Q8[i + ((entry >> 4) & 1) * (cb.order - 1) + 1];
    }
}

// ---- side information (silk_decode_indices silk.cpp:708) --------------------------------------------------
// nb_subfr: 4 (20 ms frames: all the reference decodes, Q6) or 2 (10 ms frames, RFC mode)
OG_DEV void silk_decode_indices(SilkChannel *c, SilkCtrl &k, Rc &rc, int fs_kHz, int vad, int decode_LBRR, int condCoding,
                                 i32 &ec_prevSignalType, i32 &ec_prevLagIndex, int nb_subfr = 4) {
    SilkWaveParseLds &L = PW();
    (void)c;
    const NlsfCb cb = nlsf_cb(fs_kHz == 16);
    int Ix;
    if (decode_LBRR || vad)
        Ix = rc_icdf(rc, SILK_TAB(type_vad_icdf), 8) + 2;
    else
        Ix = rc_icdf(rc, SILK_TAB(type_novad_icdf), 8);
    k.signalType = Ix >> 1;
    k.quantOffsetType = Ix & 1;
    if (condCoding == 2)
        k.GainsIndices[0] = rc_icdf(rc, SILK_TAB(delta_gain_icdf), 8);
    else {
        k.GainsIndices[0] = rc_icdf(rc, SILK_TAB(gain_icdf) + 8 * k.signalType, 8) << 3;
        k.GainsIndices[0] += rc_icdf(rc, SILK_TAB(uniform8_icdf), 8);
    }
    for (int i = 1; i < nb_subfr; i++) k.GainsIndices[i] = rc_icdf(rc, SILK_TAB(delta_gain_icdf), 8);
    k.NLSFIndices[0] = rc_icdf(rc, &cb.CB1_iCDF[(k.signalType >> 1) * 32], 8);
    nlsf_unpack(cb, k.NLSFIndices[0]);
    for (int i = 0; i < cb.order; i++) {
        Ix = rc_icdf(rc, &cb.ec_iCDF[L.ec_ix[i]], 8);
        if (Ix == 0)
            Ix -= rc_icdf(rc, SILK_TAB(nlsf_ext_icdf), 8);
        else if (Ix == 8)
            Ix += rc_icdf(rc, SILK_TAB(nlsf_ext_icdf), 8);
        k.NLSFIndices[i + 1] = Ix - 4;
    }
    k.NLSFInterpCoef_Q2 = nb_subfr == 4 ? rc_icdf(rc, SILK_TAB(nlsf_interp_icdf), 8) : 4; // silk.cpp:771-776
    if (k.signalType == 2) {
        int decode_abs = 1;
        const u8 *lowbits = fs_kHz == 16 ? SILK_TAB(uniform8_icdf) : (fs_kHz == 12 ? SILK_TAB(uniform6_icdf) : SILK_TAB(uniform4_icdf));
        const u8 *contour = fs_kHz == 8 ? (nb_subfr == 4 ? SILK_TAB(pitch_contour_nb_icdf) : SILK_TAB(pitch_contour_10ms_nb_icdf))
                                         : (nb_subfr == 4 ? SILK_TAB(pitch_contour_icdf) : SILK_TAB(pitch_contour_10ms_icdf));
        if (condCoding == 2 && ec_prevSignalType == 2) {
            int delta = rc_icdf(rc, SILK_TAB(pitch_delta_icdf), 8);
            if (delta > 0) {
                delta -= 9;
                k.lagIndex = tr16(ec_prevLagIndex + delta);
                decode_abs = 0;
            }
        }
        if (decode_abs) {
            k.lagIndex = tr16(rc_icdf(rc, SILK_TAB(pitch_lag_icdf), 8) * (fs_kHz >> 1));
            k.lagIndex = tr16(k.lagIndex + rc_icdf(rc, lowbits, 8));
        }
        ec_prevLagIndex = k.lagIndex;
        k.contourIndex = rc_icdf(rc, contour, 8);
        k.PERIndex = rc_icdf(rc, SILK_TAB(ltp_per_icdf), 8);
        const u8 *t = k.PERIndex == 0 ? SILK_TAB(ltp_gain_icdf0) : (k.PERIndex == 1 ? SILK_TAB(ltp_gain_icdf1) : SILK_TAB(ltp_gain_icdf2));
        for (int j = 0; j < nb_subfr; j++) k.LTPIndex[j] = rc_icdf(rc, t, 8);
        k.LTP_scaleIndex = condCoding == 0 ? rc_icdf(rc, SILK_TAB(ltpscale_icdf), 8) : 0;
    }
    ec_prevSignalType = k.signalType;
    k.Seed = rc_icdf(rc, SILK_TAB(uniform4_icdf), 8);
}

// ---- excitation pulses (silk_decode_pulses :898, silk_shell_decoder :1162, silk_decode_signs :1436) ----------
OG_DEV void shell_split(Rc &rc, int &c1, int &c2, int p, const u8 *table) {
    if (p > 0) {
        c1 = rc_icdf(rc, &table[SILK_TAB(shell_offsets)[p]], 8);
        c2 = p - c1;
    } else {
        c1 = 0;
        c2 = 0;
    }
}

OG_DEV void silk_decode_pulses(Rc &rc, int ch, int signalType, int quantOffsetType, int frame_length) {
    SilkWaveParseLds &L = PW();
    i16 *pulses = L.pulses[ch];
    int iter = frame_length >> 4;
    if (iter * 16 < frame_length) iter++;
    const int RateLevelIndex = rc_icdf(rc, SILK_TAB(rate_levels_icdf) + 9 * (signalType >> 1), 8);
    const u8 *cdf = SILK_TAB(pulses_per_block_icdf) + 18 * RateLevelIndex;
    for (int i = 0; i < iter; i++) {
        int nl = 0, sp = rc_icdf(rc, cdf, 8);
        while (sp == 17) {
            nl++;
            sp = rc_icdf(rc, SILK_TAB(pulses_per_block_icdf) + 18 * 9 + (nl == 10), 8);
        }
        L.nLshifts[i] = nl;
        L.sum_pulses[i] = sp;
    }
    for (int i = 0; i < iter; i++) {
        i16 *p0 = &pulses[i * 16];
        const int sp = L.sum_pulses[i];
        if (sp > 0) {
            // binary shell tree: 16 -> 8 -> 4 -> 2 -> 1, depth-first in the reference's order
            int p3[2], p2[4], p1[8], a, b;
            shell_split(rc, p3[0], p3[1], sp, SILK_TAB(shell3));
            for (int h = 0; h < 2; h++) {
                shell_split(rc, p2[2 * h], p2[2 * h + 1], p3[h], SILK_TAB(shell2));
                for (int q = 0; q < 2; q++) {
                    const int qi = 2 * h + q;
                    shell_split(rc, p1[2 * qi], p1[2 * qi + 1], p2[qi], SILK_TAB(shell1));
                    for (int e = 0; e < 2; e++) {
                        const int ei = 2 * qi + e;
                        shell_split(rc, a, b, p1[ei], SILK_TAB(shell0));
                        p0[2 * ei] = (i16)a;
                        p0[2 * ei + 1] = (i16)b;
                    }
                }
            }
        } else {
            for (int j = 0; j < 16; j++) p0[j] = 0;
        }
    }
    for (int i = 0; i < iter; i++) {
        const int nLS = L.nLshifts[i];
        if (nLS > 0) {
            i16 *p = &pulses[i * 16];
            for (int j = 0; j < 16; j++) {
                i32 abs_q = p[j];
                for (int b = 0; b < nLS; b++) {
                    abs_q = shl32(abs_q, 1);
                    abs_q += rc_icdf(rc, SILK_TAB(lsb_icdf), 8);
                }
                p[j] = (i16)abs_q;
            }
            L.sum_pulses[i] |= nLS << 5;
        }
    }
    const u8 *icdf_ptr = &SILK_TAB(sign_icdf)[7 * (quantOffsetType + (signalType << 1))];
    const int length = (frame_length + 8) >> 4;
    for (int i = 0; i < length; i++) {
        const int p = L.sum_pulses[i];
        if (p > 0) {
            const u32 ic0 = icdf_ptr[OG_MIN(p & 0x1F, 6)];
            i16 *q = &pulses[i * 16];
            for (int j = 0; j < 16; j++) {
                if (q[j] > 0) {
                    // two-symbol iCDF {ic0, 0}, ftb 8
                    u32 s = rc.rng, d = rc.val, r = s >> 8, t = s;
                    int ret = 0;
                    s = r * ic0;
                    if (d < s) {
                        t = s;
                        s = 0;
                        ret = 1;
                    }
                    rc.val = d - s;
                    rc.rng = t - s;
                    rc_renorm(rc);
                    q[j] = (i16)(q[j] * ((ret << 1) - 1));
                }
            }
        }
    }
}

// ---- NLSF decode and NLSF -> LPC ------------------------------------------------------------------------------
// The parameter-decoding functions below are written once against array views and a storage provider, and used in two
// shapes: wave-uniform (arrays contiguous in the wave's SilkLds) and lane-private (one frame per lane in the parse
// kernel, arrays laid out [element][lane] in LDS: og_silk_parse.hpp instantiates them with SilkParLane).
template <class T, int STRIDE>
struct ArrV {
    T *p;
    OG_MEMBER T &operator[](int i) const { return p[i * STRIDE]; }
    OG_MEMBER ArrV at(int o) const { ArrV r = {p + o * STRIDE}; return r; }
};
struct SilkParWave {
    typedef ArrV<i16, 1> A16;
    typedef ArrV<i32, 1> A32;
    static OG_MEMBER A16 nlsf() { A16 r = {PW().nlsf}; return r; }
    static OG_MEMBER A16 nlsf0() { A16 r = {PW().nlsf0}; return r; }
    static OG_MEMBER A16 res_Q10() { A16 r = {PW().res_Q10}; return r; }
    static OG_MEMBER A32 pred_Q8() { A32 r = {PW().pred_Q8}; return r; }
    static OG_MEMBER A32 cosLSF() { A32 r = {PW().cosLSF}; return r; }
    static OG_MEMBER A32 P() { A32 r = {PW().P}; return r; }
    static OG_MEMBER A32 Q() { A32 r = {PW().Q}; return r; }
    static OG_MEMBER A32 a32() { A32 r = {PW().a32}; return r; }
    static OG_MEMBER A32 Atmp() { A32 r = {PW().Atmp}; return r; }
};

template <class A16>
OG_DEV void silk_nlsf_stabilize(A16 NLSF_Q15, const i32 *NDeltaMin_Q15, int Lo) { // silk.cpp:2676
    int I = 0;
    for (int loops = 0; loops < 20; loops++) {
        i32 min_diff = NLSF_Q15[0] - NDeltaMin_Q15[0], diff;
        I = 0;
        for (int i = 1; i <= Lo - 1; i++) {
            diff = NLSF_Q15[i] - (NLSF_Q15[i - 1] + NDeltaMin_Q15[i]);
            if (diff < min_diff) {
                min_diff = diff;
                I = i;
            }
        }
        diff = (1 << 15) - (NLSF_Q15[Lo - 1] + NDeltaMin_Q15[Lo]);
        if (diff < min_diff) {
            min_diff = diff;
            I = Lo;
        }
        if (min_diff >= 0) return;
        if (I == 0)
            NLSF_Q15[0] = NDeltaMin_Q15[0];
        else if (I == Lo)
            NLSF_Q15[Lo - 1] = (i16)((1 << 15) - NDeltaMin_Q15[Lo]);
        else {
            i32 min_center = 0, max_center = 1 << 15;
            for (int k = 0; k < I; k++) min_center += NDeltaMin_Q15[k];
            min_center += NDeltaMin_Q15[I] >> 1;
            for (int k = Lo; k > I; k--) max_center -= NDeltaMin_Q15[k];
            max_center -= NDeltaMin_Q15[I] >> 1;
            i32 center = tr16(limit32(rshift_round((i32)NLSF_Q15[I - 1] + (i32)NLSF_Q15[I], 1), min_center, max_center));
            NLSF_Q15[I - 1] = (i16)(center - (NDeltaMin_Q15[I] >> 1));
            NLSF_Q15[I] = (i16)(NLSF_Q15[I - 1] + NDeltaMin_Q15[I]);
        }
    }
    for (int i = 1; i < Lo; i++) { // fall-back: insertion sort, then spacing from both ends
        i32 value = NLSF_Q15[i];
        int j;
        for (j = i - 1; j >= 0 && value < NLSF_Q15[j]; j--) NLSF_Q15[j + 1] = NLSF_Q15[j];
        NLSF_Q15[j + 1] = (i16)value;
    }
    NLSF_Q15[0] = (i16)OG_MAX((i32)NLSF_Q15[0], (i32)NDeltaMin_Q15[0]);
    for (int i = 1; i < Lo; i++)
        NLSF_Q15[i] = (i16)OG_MAX((i32)NLSF_Q15[i], sat16((i32)NLSF_Q15[i - 1] + NDeltaMin_Q15[i]));
    NLSF_Q15[Lo - 1] = (i16)OG_MIN((i32)NLSF_Q15[Lo - 1], (1 << 15) - NDeltaMin_Q15[Lo]);
    for (int i = Lo - 2; i >= 0; i--) NLSF_Q15[i] = (i16)OG_MIN((i32)NLSF_Q15[i], NLSF_Q15[i + 1] - NDeltaMin_Q15[i + 1]);
}

template <class W>
OG_DEV void silk_nlsf_decode(typename W::A16 pNLSF_Q15, const i32 *NLSFIndices, const NlsfCb &cb) { // silk.cpp:2466, :2445
    const typename W::A16 res_Q10 = W::res_Q10();
    const typename W::A32 pred_Q8 = W::pred_Q8();
    { // silk_NLSF_unpack silk.cpp:2762 (the predictor weights; the entropy tables belong to the parse)
        const u8 *sel = &cb.ec_sel[NLSFIndices[0] * cb.order / 2];
        for (int i = 0; i < cb.order; i += 2) {
            const int entry = *sel++;
            pred_Q8[i] = cb.pred_Q8[i + (entry & 1) * (cb.order - 1)];
            pred_Q8[i + 1] = cb.pred_Q8[i + ((entry >> 4) & 1) * (cb.order - 1) + 1];
        }
    }
    i32 out_Q10 = 0;
    for (int i = cb.order - 1; i >= 0; i--) {
        i32 pred_Q10 = smulbb(out_Q10, pred_Q8[i]) >> 8;
        out_Q10 = shl32(NLSFIndices[1 + i], 10);
        if (out_Q10 > 0)
            out_Q10 -= 102;
        else if (out_Q10 < 0)
            out_Q10 += 102;
        out_Q10 = smlawb(pred_Q10, out_Q10, cb.quantStepSize_Q16);
        res_Q10[i] = (i16)out_Q10;
    }
    const u8 *pCB = &cb.CB1_NLSF_Q8[NLSFIndices[0] * cb.order];
    const i32 *pW = &cb.CB1_Wght_Q9[NLSFIndices[0] * cb.order];
    for (int i = 0; i < cb.order; i++) {
        i32 t = shl32((i32)res_Q10[i], 14) / (i32)pW[i] + shl32((i32)pCB[i], 7);
        pNLSF_Q15[i] = (i16)limit32(t, 0, 32767);
    }
    silk_nlsf_stabilize(pNLSF_Q15, cb.deltaMin_Q15, cb.order);
}

template <class A32>
OG_DEV void silk_bwexpander_32(A32 ar, int d, i32 chirp_Q16) { // silk.cpp:561
    const i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    for (int i = 0; i < d - 1; i++) {
        ar[i] = smulww(chirp_Q16, ar[i]);
        chirp_Q16 += rshift_round(chirp_Q16 * chirp_minus_one_Q16, 16);
    }
    ar[d - 1] = smulww(chirp_Q16, ar[d - 1]);
}

template <class W, class AQ>
OG_DEV i32 silk_inverse_pred_gain(AQ A_Q12, int order) { // silk.cpp:2425 + :2359
    const typename W::A32 A = W::Atmp();
    const i32 A_LIMIT = 16773022, MIN_INVGAIN = 107374;
    i32 DC_resp = 0;
    for (int k = 0; k < order; k++) {
        DC_resp += (i32)A_Q12[k];
        A[k] = shl32((i32)A_Q12[k], 12);
    }
    if (DC_resp >= 4096) return 0;
    i32 invGain_Q30 = 1 << 30, rc_Q31, rc_mult1_Q30;
    int k;
    for (k = order - 1; k > 0; k--) {
        if (A[k] > A_LIMIT || A[k] < -A_LIMIT) return 0;
        rc_Q31 = -shl32(A[k], 7);
        rc_mult1_Q30 = (1 << 30) - smmul(rc_Q31, rc_Q31);
        invGain_Q30 = shl32(smmul(invGain_Q30, rc_mult1_Q30), 2);
        if (invGain_Q30 < MIN_INVGAIN) return 0;
        const int mult2Q = 32 - clz32(iabs(rc_mult1_Q30));
        const i32 rc_mult2 = silk_inverse32_varQ(rc_mult1_Q30, mult2Q + 30);
        for (int n = 0; n < (k + 1) >> 1; n++) {
            const i32 tmp1 = A[n], tmp2 = A[k - n - 1];
            i64 t64 = rshift_round64((i64)sub_sat32(tmp1, (i32)rshift_round64((i64)tmp2 * rc_Q31, 31)) * rc_mult2, mult2Q);
            if (t64 > 2147483647LL || t64 < -2147483648LL) return 0;
            A[n] = (i32)t64;
            t64 = rshift_round64((i64)sub_sat32(tmp2, (i32)rshift_round64((i64)tmp1 * rc_Q31, 31)) * rc_mult2, mult2Q);
            if (t64 > 2147483647LL || t64 < -2147483648LL) return 0;
            A[k - n - 1] = (i32)t64;
        }
    }
    if (A[k] > A_LIMIT || A[k] < -A_LIMIT) return 0;
    rc_Q31 = -shl32(A[0], 7);
    rc_mult1_Q30 = (1 << 30) - smmul(rc_Q31, rc_Q31);
    invGain_Q30 = shl32(smmul(invGain_Q30, rc_mult1_Q30), 2);
    if (invGain_Q30 < MIN_INVGAIN) return 0;
    return invGain_Q30;
}

template <class A32>
OG_DEV void silk_find_poly(A32 out, A32 cLSF, int dd) { // silk.cpp:626
    out[0] = 1 << 16;
    out[1] = -cLSF[0];
    for (int k = 1; k < dd; k++) {
        const i32 ftmp = cLSF[2 * k];
        out[k + 1] = shl32(out[k - 1], 1) - (i32)rshift_round64((i64)ftmp * out[k], 16);
        for (int n = k; n > 1; n--) out[n] += out[n - 2] - (i32)rshift_round64((i64)ftmp * out[n - 1], 16);
        out[1] -= ftmp;
    }
}

template <class W, class AQ>
OG_DEV void silk_nlsf2a(AQ a_Q12, typename W::A16 NLSF, int d) { // silk.cpp:642
    struct { typename W::A32 cosLSF, P, Q, a32; } L = {W::cosLSF(), W::P(), W::Q(), W::a32()};
    // ordering16 = {0,15,8,7,4,11,12,3,2,13,10,5,6,9,14,1}; ordering10 = {0,9,6,3,4,5,8,1,2,7}: one nibble each
    const u64 ord16 = 0x1E965AD23CB478F0ULL; // nibbles (k=0..15): 0,15,8,7,4,11,12,3,2,13,10,5,6,9,14,1
    const u64 o10 = 0x0000007218543690ULL;   // nibbles (k=0..9): 0,9,6,3,4,5,8,1,2,7
    for (int k = 0; k < d; k++) {
        const i32 f_int = NLSF[k] >> 8, f_frac = NLSF[k] - shl32(f_int, 8);
        const i32 cos_val = rom_silk_cos_q12[f_int], delta = rom_silk_cos_q12[f_int + 1] - cos_val;
        const int o = (int)(((d == 16 ? ord16 : o10) >> (4 * k)) & 15);
        L.cosLSF[o] = rshift_round(shl32(cos_val, 8) + delta * f_frac, 4);
    }
    const int dd = d >> 1;
    silk_find_poly(L.P, L.cosLSF, dd);
    silk_find_poly(L.Q, L.cosLSF.at(1), dd);
    for (int k = 0; k < dd; k++) {
        const i32 Ptmp = L.P[k + 1] + L.P[k], Qtmp = L.Q[k + 1] - L.Q[k];
        L.a32[k] = -Qtmp - Ptmp;
        L.a32[d - k - 1] = Qtmp - Ptmp;
    }
    { // silk_LPC_fit(a_Q12, a32, 12, 17, d) silk.cpp:2314
        int i, idx = 0;
        for (i = 0; i < 10; i++) {
            i32 maxabs = 0;
            for (int k = 0; k < d; k++) {
                const i32 absval = iabs(L.a32[k]);
                if (absval > maxabs) {
                    maxabs = absval;
                    idx = k;
                }
            }
            maxabs = rshift_round(maxabs, 5);
            if (maxabs > 32767) {
                maxabs = OG_MIN(maxabs, 163838);
                const i32 chirp_Q16 = 65470 - shl32(maxabs - 32767, 14) / ((maxabs * (idx + 1)) >> 2);
                silk_bwexpander_32(L.a32, d, chirp_Q16);
            } else
                break;
        }
        if (i == 10) {
            for (int k = 0; k < d; k++) {
                a_Q12[k] = (i16)sat16(rshift_round(L.a32[k], 5));
                L.a32[k] = shl32((i32)a_Q12[k], 5);
            }
        } else {
            for (int k = 0; k < d; k++) a_Q12[k] = (i16)rshift_round(L.a32[k], 5);
        }
    }
    for (int i = 0; silk_inverse_pred_gain<W>(a_Q12, d) == 0 && i < 16; i++) {
        silk_bwexpander_32(L.a32, d, 65536 - shl32(2, i));
        for (int k = 0; k < d; k++) a_Q12[k] = (i16)rshift_round(L.a32[k], 5);
    }
}

// silk_decode_parameters silk.cpp:827 (+ silk_gains_dequant :2148, silk_decode_pitch :2055)
// `k`: the frame's indices in, dequantised parameters out (SilkCtrl in LDS, or the record's SilkRecCh in HBM).  The
// stabilised NLSFs are left in W::nlsf() for the caller to store as the next frame's prevNLSF.
template <class W, class K>
OG_DEV void silk_decode_parameters(const i16 *prevNLSF_Q15, K &k, int fs_kHz, int condCoding, i32 &LastGainIndex,
                                   int first_frame_after_reset, int nb_subfr = 4) {
    typedef ArrV<i16, 1> AQ;
    struct { typename W::A16 nlsf, nlsf0; } L = {W::nlsf(), W::nlsf0()};
    const int order = fs_kHz == 16 ? 16 : 10;
    const NlsfCb cb = nlsf_cb(fs_kHz == 16);
    for (int j = 0; j < nb_subfr; j++) {
        const int ind = k.GainsIndices[j];
        i32 prev = LastGainIndex;
        if (j == 0 && condCoding != 2)
            prev = OG_MAX(ind, prev - 16);
        else {
            const int ind_tmp = ind - 4, thr = 2 * 36 - 64 + prev;
            prev += ind_tmp > thr ? shl32(ind_tmp, 1) - thr : ind_tmp;
        }
        prev = (i32)(i8)prev;
        prev = limit32(prev, 0, 63);
        LastGainIndex = prev;
        k.Gains_Q16[j] = silk_log2lin(OG_MIN(smulwb(1907825, prev) + 2090, 3967));
    }
    silk_nlsf_decode<W>(L.nlsf, k.NLSFIndices, cb);
    { AQ a1 = {k.PredCoef_Q12[1]}; silk_nlsf2a<W>(a1, L.nlsf, order); }
    if (first_frame_after_reset == 1) k.NLSFInterpCoef_Q2 = 4;
    if (k.NLSFInterpCoef_Q2 < 4) {
        for (int i = 0; i < order; i++) {
            const i32 pv = prevNLSF_Q15[i];
            L.nlsf0[i] = (i16)(pv + ((k.NLSFInterpCoef_Q2 * ((i32)L.nlsf[i] - pv)) >> 2));
        }
        { AQ a0 = {k.PredCoef_Q12[0]}; silk_nlsf2a<W>(a0, L.nlsf0, order); }
    } else {
        for (int i = 0; i < order; i++) k.PredCoef_Q12[0][i] = k.PredCoef_Q12[1][i];
    }
    if (k.signalType == 2) {
        const i8 *cbk = k.PERIndex == 0 ? rom_silk_ltp_vq0 : (k.PERIndex == 1 ? rom_silk_ltp_vq1 : rom_silk_ltp_vq2);
        // silk_decode_pitch silk.cpp:2055: the 10 ms frames have codebooks of their own
        const i8 *lagcb = fs_kHz == 8 ? (nb_subfr == 4 ? rom_silk_lags_stage2 : rom_silk_lags_stage2_10ms)
                                      : (nb_subfr == 4 ? rom_silk_lags_stage3 : rom_silk_lags_stage3_10ms);
        const int cbk_size = fs_kHz == 8 ? (nb_subfr == 4 ? 11 : 3) : (nb_subfr == 4 ? 34 : 12);
        const int min_lag = 2 * fs_kHz, max_lag = 18 * fs_kHz, lag = min_lag + k.lagIndex;
        for (int j = 0; j < nb_subfr; j++) {
            k.pitchL[j] = limit32(lag + lagcb[j * cbk_size + k.contourIndex], min_lag, max_lag);
            for (int i = 0; i < 5; i++) k.LTPCoef_Q14[j * 5 + i] = (i16)shl32((i32)cbk[k.LTPIndex[j] * 5 + i], 7);
        }
        k.LTP_scale_Q14 = rom_silk_ltp_scales_q14[k.LTP_scaleIndex];
    } else {
        for (int j = 0; j < nb_subfr; j++) k.pitchL[j] = 0;
        for (int j = 0; j < 5 * nb_subfr; j++) k.LTPCoef_Q14[j] = 0;
        k.PERIndex = 0;
        k.LTP_scale_Q14 = 0;
    }
}

// ---- lane-private parameter decoding for the parse kernel (scratch: SilkParLds, og_silk_parse.hpp) -----------------
struct SilkParLane {
    typedef ArrV<i16, OG_PAR_LANES> A16;
    typedef ArrV<i32, OG_PAR_LANES> A32;
    static OG_MEMBER A16 nlsf() { A16 r = {&g_silk_par.nlsf[0][OG_LANE]}; return r; }
    static OG_MEMBER A16 nlsf0() { A16 r = {&g_silk_par.nlsf0[0][OG_LANE]}; return r; }
    static OG_MEMBER A16 res_Q10() { A16 r = {&g_silk_par.res_Q10[0][OG_LANE]}; return r; }
    static OG_MEMBER A32 pred_Q8() { A32 r = {&g_silk_par.pred_Q8[0][OG_LANE]}; return r; }
    static OG_MEMBER A32 cosLSF() { A32 r = {&g_silk_par.cosLSF[0][OG_LANE]}; return r; }
    static OG_MEMBER A32 P() { A32 r = {&g_silk_par.P[0][OG_LANE]}; return r; }
    static OG_MEMBER A32 Q() { A32 r = {&g_silk_par.Q[0][OG_LANE]}; return r; }
    static OG_MEMBER A32 a32() { A32 r = {&g_silk_par.a32[0][OG_LANE]}; return r; }
    static OG_MEMBER A32 Atmp() { A32 r = {&g_silk_par.Atmp[0][OG_LANE]}; return r; }
};
static_assert(sizeof(SilkCtrl) == 4 * (SILK_REC_CTRL_WORDS + 1), "SilkRecCh mirrors SilkCtrl");
static_assert(SILK_LDS_FRAME * 4 * 2 <= 1920 * 2, "up-sampler input staging sits under pcm");

// The parameter half of a SILK-only / hybrid frame on the split path: silk_decode_parameters for the coded channels, from the
// indices the parse lane wrote to the record.  The inputs that live in the stream state (LastGainIndex, first_frame_after_reset,
// previous NLSFs) are taken as the wave kernel will see them after its own (re-)initialisations: decoder init on a CELT ->
// SILK/hybrid switch, channel 1 init when the packet adds a channel, silk_decoder_set_fs on a rate change, side-channel
// restart after a mid-only frame (silk.cpp:1639).  Nothing but the record is written -- and, for pipelined SILK / hybrid steps
// (`shadow`, see SilkShadow), what the NEXT frame's entropy half needs: the values the synthesis kernel will have written to the
// stream's state by the time it is through with this frame (silk_decode_packet: silk_init_state on a switch from CELT,
// silk_chan_init for a channel the packet adds, silk_set_fs, the side channel's restart, the indices' history, the gain index, the
// stabilised NLSFs; decode_frame_wave: prev_mode).
// In two parts per channel, so that k_silk_params can give every (frame, channel) a lane of its own: silk_params_channel READS
// the past and writes the record, silk_params_shadow WRITES the parameter side of the next frame's past -- every read of all lanes
// before any write (a barrier between the two in the kernel; one after the other here for a caller that does both channels itself).
// What the parameter half needs of the ENTROPY side's past comes from the record (SilkRec::par_flags, fs_past), not from the shadow:
// the parse of the stream's next frame may have moved that side on already.
struct SilkParTask {
    int skip;   // the frame ended in an error before anything of the state was touched: the past stays what it is
    int fs_kHz, order, fresh, coded;
    i32 lastGain_past, ffar_past; // the channel's past as the next frame would see it if this one left it alone
};
OG_DEV void silk_params_channel(const SilkParPast &past, int mode, int bandwidth, int channels, SilkRec *rec, int n, SilkParTask &t) {
    t.skip = rec->ret < 0;
    if (t.skip) return;
    int internal_hz = 16000;
    if (mode == MODE_SILK) internal_hz = bandwidth == BW_NB ? 8000 : (bandwidth == BW_MB ? 12000 : 16000);
    const int fs_kHz = (internal_hz >> 10) + 1, order = fs_kHz == 16 ? 16 : 10;
    const int pf = rec->par_flags, fresh_all = pf & 1, fresh_ch1 = (pf >> 1) & 1, prev_dom = (pf >> 2) & 1, dom = rec->decode_only_middle;
    const int fresh = fresh_all || (n == 1 && fresh_ch1); // the channel was (re-)initialised: everything zero, first frame after a reset
    t.fs_kHz = fs_kHz; t.order = order; t.fresh = fresh;
    t.coded = n < channels && !(n == 1 && dom); // (no side channel in a mid-only frame)
    t.lastGain_past = fresh ? 0 : past.lastGain(n);
    t.ffar_past = fresh ? 1 : past.ffar(n);
    if (!t.coded) return;
    const int changed = fresh || rec->fs_past[n] != fs_kHz;
    i32 lastGain = changed ? 10 : past.lastGain(n);
    int ffar = changed ? 1 : past.ffar(n);
    if (n == 1 && channels == 2 && dom == 0 && prev_dom == 1) {
        lastGain = 10;
        ffar = 1;
    }
    SilkRecCh &k = rec->ch[n];
    OG_MARK(53);
    silk_decode_parameters<SilkParLane>(past.prevNLSF(n), k, fs_kHz, 0, lastGain, ffar);
    k.LastGainIndex = lastGain;
    const SilkParLane::A16 nl = SilkParLane::nlsf();
    for (int i = 0; i < order; i++) k.nlsf[i] = nl[i];
}
// the parameter side of the stream's next frame's past (the entropy side: silk_parse_lane)
OG_DEV void silk_params_shadow(const SilkParPast &past, const SilkParTask &t, int channels, const SilkRec *rec, int n, SilkShadow *shadow,
                               u32 epoch) {
    if (t.skip || !shadow) return;
    i32 lastGain = t.lastGain_past, ffar = t.ffar_past;
    i16 nl[SILK_REC_LPC];
    for (int i = 0; i < SILK_REC_LPC; i++) nl[i] = t.fresh ? (i16)0 : past.prevNLSF(n)[i];
    if (n < channels) {
        if (rec->fs_past[n] != t.fs_kHz) { // silk_set_fs
            ffar = 1;
            lastGain = 10;
        }
        if (t.coded) {
            lastGain = rec->ch[n].LastGainIndex;
            ffar = 0;
            for (int i = 0; i < t.order; i++) nl[i] = rec->ch[n].nlsf[i];
        }
    }
    SilkShadow::Ch &o = shadow->ch[n];
    o.LastGainIndex = lastGain;
    o.first_frame_after_reset = ffar;
    for (int i = 0; i < SILK_REC_LPC; i++) o.prevNLSF_Q15[i] = nl[i];
    if (n == 0) shadow->par_epoch = epoch;
}
// both channels by one caller (host emulation; the kernel's lanes take one channel each)
OG_DEV void silk_params_lane(const SilkParPast &past, int mode, int bandwidth, int channels, SilkRec *rec, SilkShadow *shadow = nullptr,
                             u32 epoch = 0) {
    SilkParTask t[2];
    for (int n = 0; n < 2; n++) silk_params_channel(past, mode, bandwidth, channels, rec, n, t[n]);
    for (int n = 1; n >= 0; n--) silk_params_shadow(past, t[n], channels, rec, n, shadow, epoch); // (the epoch last)
}

// ---- synthesis: one lane per channel (silk_decode_core silk.cpp:1806, LPC analysis filter :2268) --------------
// `pulses`: the channel's excitation pulses -- in the parse record in HBM (split path) or in PW().pulses
// `lc` (RFC mode): the excitation is kept for a later concealment, and the first decoded frame after a loss is smoothed
OG_DEVN void silk_decode_core_lane(SilkChannel *c, int ch, int fs_kHz, const i16 *pulses, int nb_subfr = 4,
                                   SilkLossChannel *lc = nullptr) {
    SilkLds &L = SL();
    SilkCtrl &k = L.ctrl[ch];
    const int after_loss = lc && lc->lossCnt && c->prevSignalType == 2 && k.signalType != 2; // silk.cpp:1869
    const int order = fs_kHz == 16 ? 16 : 10, subfr = 5 * fs_kHz, frame_length = nb_subfr * subfr, ltp_mem = 20 * fs_kHz;
    i16 *xq = &L.xq[ch][2];
    i32 *sLTP_Q15 = L.u.core.sLTP_Q15[ch];
    i16 *sLTP = L.u.core.sLTP[ch];
    const i32 offset_Q10 = rom_silk_quant_offsets_q10[(k.signalType >> 1) * 2 + k.quantOffsetType];
    const int interp_flag = k.NLSFInterpCoef_Q2 < 4;
    i32 sLPC[SILK_MAX_LPC]; // sLPC[j] = state sample (i-1-j): most recent first
    for (int j = 0; j < SILK_MAX_LPC; j++) sLPC[j] = c->sLPC_Q14_buf[SILK_MAX_LPC - 1 - j];
    i32 rand_seed = k.Seed;
    i32 prev_gain_Q16 = c->prev_gain_Q16;
    int sLTP_buf_idx = ltp_mem, lag = 0, pos = 0;
    for (int sf = 0; sf < nb_subfr; sf++) {
        const i16 *A_Q12 = k.PredCoef_Q12[sf >> 1];
        const i16 *B_Q14 = &k.LTPCoef_Q14[sf * 5];
        const i32 Gain_Q16 = k.Gains_Q16[sf], Gain_Q10 = Gain_Q16 >> 6;
        i32 inv_gain_Q31 = silk_inverse32_varQ(Gain_Q16, 47), gain_adj_Q16;
        if (Gain_Q16 != prev_gain_Q16) {
            gain_adj_Q16 = silk_div32_varQ(prev_gain_Q16, Gain_Q16, 16);
            for (int j = 0; j < SILK_MAX_LPC; j++) sLPC[j] = smulww(gain_adj_Q16, sLPC[j]);
        } else
            gain_adj_Q16 = 1 << 16;
        prev_gain_Q16 = Gain_Q16;
        int voiced = k.signalType == 2;
        i16 B_smooth[5] = {0, 0, 4096, 0, 0}; // SILK_FIX_CONST(0.25, 14) on the centre tap
        if (after_loss && sf < 2) { // avoid an abrupt transition from voiced concealment to unvoiced decoding (silk.cpp:1869-1876)
            for (int i = 0; i < 5; i++) k.LTPCoef_Q14[sf * 5 + i] = B_smooth[i];
            voiced = 1;
            k.pitchL[sf] = c->lagPrev;
        }
        if (voiced) {
            lag = k.pitchL[sf];
            if (sf == 0 || (sf == 2 && interp_flag)) { // re-whitening
                const int start_idx = ltp_mem - lag - order - 2;
                // input = outBuf history (staged in LDS) followed, for sf == 2, by the two subframes decoded so far
                i16 *hist = L.u.core.hist[ch];
                if (sf == 2)
                    for (int i = 0; i < 2 * subfr; i++) hist[ltp_mem + i] = xq[i];
                const i16 *in = &hist[start_idx + sf * subfr];
                for (int ix = order; ix < ltp_mem - start_idx; ix++) {
                    i32 acc = 0;
                    for (int j = 0; j < order; j++) acc = smlabb(acc, in[ix - 1 - j], A_Q12[j]);
                    sLTP[start_idx + ix] = (i16)sat16(rshift_round(subw(shl32((i32)in[ix], 12), acc), 12));
                }
                for (int j = 0; j < order; j++) sLTP[start_idx + j] = 0;
                if (sf == 0) inv_gain_Q31 = shl32(smulwb(inv_gain_Q31, k.LTP_scale_Q14), 2);
                for (int i = 0; i < lag + 2; i++) sLTP_Q15[sLTP_buf_idx - i - 1] = smulwb(inv_gain_Q31, sLTP[ltp_mem - i - 1]);
            } else if (gain_adj_Q16 != 1 << 16) {
                for (int i = 0; i < lag + 2; i++) sLTP_Q15[sLTP_buf_idx - i - 1] = smulww(gain_adj_Q16, sLTP_Q15[sLTP_buf_idx - i - 1]);
            }
        }
        for (int i = 0; i < subfr; i++) {
            // excitation (silk.cpp:1826-1835)
            rand_seed = (i32)(907633515u + (u32)rand_seed * 196314165u);
            const i32 pl = pulses[pos + i];
            i32 exc = shl32(pl, 14);
            if (exc > 0)
                exc -= 80 << 4;
            else if (exc < 0)
                exc += 80 << 4;
            exc += offset_Q10 << 4;
            if (rand_seed < 0) exc = -exc;
            rand_seed = addw(rand_seed, pl);
            if (lc) lc->exc_Q14[pos + i] = exc;
            i32 res = exc;
            if (voiced) {
                const i32 *p = &sLTP_Q15[sLTP_buf_idx - lag + 2];
                i32 LTP_pred_Q13 = 2;
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[0], B_Q14[0]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-1], B_Q14[1]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-2], B_Q14[2]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-3], B_Q14[3]);
                LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-4], B_Q14[4]);
                res = addw(exc, shl32(LTP_pred_Q13, 1));
                sLTP_Q15[sLTP_buf_idx] = shl32(res, 1);
                sLTP_buf_idx++;
            }
            i32 LPC_pred_Q10 = order >> 1;
#pragma unroll
            for (int j = 0; j < 10; j++) LPC_pred_Q10 = smlawb(LPC_pred_Q10, sLPC[j], A_Q12[j]);
            if (order == 16) {
#pragma unroll
                for (int j = 10; j < 16; j++) LPC_pred_Q10 = smlawb(LPC_pred_Q10, sLPC[j], A_Q12[j]);
            }
            const i32 s = add_sat32(res, lshift_sat32(LPC_pred_Q10, 4));
#pragma unroll
            for (int j = SILK_MAX_LPC - 1; j > 0; j--) sLPC[j] = sLPC[j - 1];
            sLPC[0] = s;
            xq[pos + i] = (i16)sat16(rshift_round(smulww(s, Gain_Q10), 8));
        }
        pos += subfr;
    }
    for (int j = 0; j < SILK_MAX_LPC; j++) c->sLPC_Q14_buf[SILK_MAX_LPC - 1 - j] = sLPC[j];
    c->prev_gain_Q16 = prev_gain_Q16;
    (void)frame_length; // outBuf update happens lane-parallel in the caller
    c->lagPrev = k.pitchL[nb_subfr - 1];
    c->prevSignalType = k.signalType;
    c->first_frame_after_reset = 0;
}

#ifndef OG_HOST_EMUL
// The same synthesis with one 16-lane ROW per channel (row 0 = channel 0, row 1 = channel 1; wave64 = 4 DPP rows).
// The order-10/16 LPC recurrence is a chain of 16 dependent multiply-adds per sample when one lane runs it; here lane j
// of the row keeps state sample (i-1-j) and coefficient j, the prediction is one multiply plus a 4-step DPP row
// all-reduce, and the state shifts by one lane per sample (row_shr:1).  Everything else (excitation, LTP, gains) is
// computed redundantly by the 16 lanes of the row; the re-whitening FIR and the LTP-state scaling are spread over them.
// Must be entered by all 64 lanes.  Same arithmetic as silk_decode_core_lane (silk_decode_core silk.cpp:1806).
#define OG_ROW_SYNC()                                                   \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          \
        __builtin_amdgcn_wave_barrier();                                \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          \
    } while (0)
OG_DEV i32 row_sum16(i32 v) { // sum over the 16 lanes of a DPP row, result in every lane of the row
    v += OG_DPP_ROR(v, 8);
    v += OG_DPP_ROR(v, 4);
    v += OG_DPP_ROR(v, 2);
    v += OG_DPP_ROR(v, 1);
    return v;
}
OG_DEV i32 row_lane0(i32 v) { // lane 0 of every 16-lane row, in all lanes of that row
    // keep lane 0's value, zero elsewhere, then sum over the row
    return row_sum16((OG_LANE & 15) == 0 ? v : 0);
}
// `nb_subfr`: 4, or 2 for a 10 ms frame (RFC mode); `loss` (RFC mode): the stream's loss-concealment state -- the frame's excitation
// is kept there for a later concealment (silk.cpp:1835), and the first two subframes after a voiced concealment are smoothed
// (:1869-1876).
// (LOSS = false -- reference mode: no loss state, four subframes -- is a compile-time case of its own: written as run-time values
// the kernel of the split path came out with 147 instead of 90 registers, three waves per SIMD instead of four)
template <bool LOSS>
OG_DEV void silk_decode_core_rows(SilkState *st, int fs_kHz, int channels, const i16 *pulses0, const i16 *pulses1, int nb_subfr_arg = 4,
                                  LossState *loss_arg = nullptr) {
    const int nb_subfr = LOSS ? nb_subfr_arg : 4;
    LossState *const loss = LOSS ? loss_arg : nullptr;
    SilkLds &L = SL();
    const int row = OG_LANE >> 4, j = OG_LANE & 15;
    if (row >= channels) return;
    const int ch = row;
    const int order = fs_kHz == 16 ? 16 : 10, subfr = 5 * fs_kHz, ltp_mem = 20 * fs_kHz, frame_length = nb_subfr * subfr;
    i16 *xq = silk_xq_row(ch, ltp_mem) + 2;
    if (!L.ctrl[ch].coded) {
        for (int i = j; i < frame_length; i += 16) xq[i] = 0;
        return;
    }
    SilkChannel *c = &st->ch[ch];
    SilkCtrl &k = L.ctrl[ch];
    SilkLossChannel *const lc = loss ? &loss->silk[ch] : nullptr;
    const int after_loss = LOSS && lc && lc->lossCnt && c->prevSignalType == 2 && k.signalType != 2; // silk.cpp:1869
    const int lag_prev = LOSS ? c->lagPrev : 0;
    const i16 *pulses = ch ? pulses1 : pulses0;
    i32 *sLTP_Q15 = L.u.core.sLTP_Q15[ch];
#ifndef OG_SILK_TIGHT
    i16 *sLTP = L.u.core.sLTP[ch];
#endif
    const i32 offset_Q10 = rom_silk_quant_offsets_q10[(k.signalType >> 1) * 2 + k.quantOffsetType];
    const int interp_flag = k.NLSFInterpCoef_Q2 < 4, voiced_frame = k.signalType == 2;
    i32 sLPC = c->sLPC_Q14_buf[SILK_MAX_LPC - 1 - j]; // state sample (i-1-j)
    i32 rand_seed = k.Seed;
    i32 prev_gain_Q16 = c->prev_gain_Q16;
    int sLTP_buf_idx = ltp_mem, lag = 0, pos = 0;
    for (int sf = 0; sf < nb_subfr; sf++) {
        OG_MARK(60);
        int voiced = voiced_frame;
        if (after_loss && sf < 2) { // avoid an abrupt transition from voiced concealment to unvoiced decoding (silk.cpp:1869-1876)
            if (j < 5) k.LTPCoef_Q14[sf * 5 + j] = (i16)(j == 2 ? 4096 : 0); // SILK_FIX_CONST(0.25, 14) on the centre tap
            if (j == 0) k.pitchL[sf] = lag_prev;
            voiced = 1;
            OG_ROW_SYNC();
        }
        const i16 *A_Q12 = k.PredCoef_Q12[sf >> 1];
        const i16 *B_Q14 = &k.LTPCoef_Q14[sf * 5];
        const i32 A_j = j < order ? (i32)A_Q12[j] : 0;
        const i32 Gain_Q16 = k.Gains_Q16[sf], Gain_Q10 = Gain_Q16 >> 6;
        i32 inv_gain_Q31 = silk_inverse32_varQ(Gain_Q16, 47), gain_adj_Q16;
        if (Gain_Q16 != prev_gain_Q16) {
            gain_adj_Q16 = silk_div32_varQ(prev_gain_Q16, Gain_Q16, 16);
            sLPC = smulww(gain_adj_Q16, sLPC);
        } else
            gain_adj_Q16 = 1 << 16;
        prev_gain_Q16 = Gain_Q16;
        if (voiced) {
            lag = k.pitchL[sf];
            if (sf == 0 || (sf == 2 && interp_flag)) { // re-whitening: a FIR, one output per lane and step
                const int start_idx = ltp_mem - lag - order - 2;
                i16 *hist = silk_hist_row(ch);
#ifdef OG_SILK_TIGHT
                // (the frame's first two subframes lie right behind the history: hist[ltp_mem + i] IS xq[i].)  Of the filter's
                // lag + order + 2 outputs the LTP state takes the last lag + 2, scaled: output ix (>= order) goes to
                // sLTP_Q15[sLTP_buf_idx - (lag + order + 2) + ix] as it is made -- the same values the two passes over a buffer of
                // whitened history left there (silk.cpp:1893-1905)
                const i16 *in = &hist[start_idx + sf * subfr];
                if (sf == 0) inv_gain_Q31 = shl32(smulwb(inv_gain_Q31, k.LTP_scale_Q14), 2);
                i32 *const dst = sLTP_Q15 + (sLTP_buf_idx - (lag + order + 2));
                for (int ix = order + j; ix < lag + order + 2; ix += 16) {
                    i32 acc = 0;
                    for (int t = 0; t < order; t++) acc = smlabb(acc, in[ix - 1 - t], A_Q12[t]);
                    dst[ix] = smulwb(inv_gain_Q31, sat16(rshift_round(subw(shl32((i32)in[ix], 12), acc), 12)));
                }
                OG_ROW_SYNC();
#else
                if (sf == 2) {
                    for (int i = j; i < 2 * subfr; i += 16) hist[ltp_mem + i] = xq[i];
                    OG_ROW_SYNC();
                }
                const i16 *in = &hist[start_idx + sf * subfr];
                for (int ix = order + j; ix < ltp_mem - start_idx; ix += 16) {
                    i32 acc = 0;
                    for (int t = 0; t < order; t++) acc = smlabb(acc, in[ix - 1 - t], A_Q12[t]);
                    sLTP[start_idx + ix] = (i16)sat16(rshift_round(subw(shl32((i32)in[ix], 12), acc), 12));
                }
                if (j < order) sLTP[start_idx + j] = 0;
                if (sf == 0) inv_gain_Q31 = shl32(smulwb(inv_gain_Q31, k.LTP_scale_Q14), 2);
                OG_ROW_SYNC();
                for (int i = j; i < lag + 2; i += 16) sLTP_Q15[sLTP_buf_idx - i - 1] = smulwb(inv_gain_Q31, sLTP[ltp_mem - i - 1]);
                OG_ROW_SYNC();
#endif
            } else if (gain_adj_Q16 != 1 << 16) {
                for (int i = j; i < lag + 2; i += 16) sLTP_Q15[sLTP_buf_idx - i - 1] = smulww(gain_adj_Q16, sLTP_Q15[sLTP_buf_idx - i - 1]);
                OG_ROW_SYNC();
            }
        }
        // The subframe in three phases, so that only what must wait for the previous output sample does (silk.cpp:1826-1873):
        //  1. excitation: the dither seed absorbs each pulse -- a chain of affine maps, folded per lane and scanned over the row;
        //  2. voiced: the LTP prediction reads its own state at least lag - 2 samples back -- min(16, lag - 2) samples at a
        //     time, one per lane;
        //  3. the LPC recurrence.  silk_SMLAWB wraps, so the order of its additions is free: taps 2 .. order of the NEXT
        //     sample do not involve the sample being computed and are reduced over the row beside it; one multiply-add and
        //     the saturating update stay on the dependent chain.  The row sum leaves the same value in every lane, so each
        //     lane again keeps the samples it owns; the output scaling happens after the loop, in parallel.
        OG_MARK(61);
#ifdef OG_SILK_TIGHT
        i32 *resb = L.u.core.resb[ch]; // residuals of this subframe
#else
        i32 *resb = reinterpret_cast<i32 *>(sLTP); // residuals of this subframe (the whitened history is dead by now)
#endif
        enum { OWN = (SILK_LDS_FRAME / 4 + 15) / 16 };
        {
            // The dither seed: r <- a r + c, used for the sign, then r <- r + pulse: one affine map mod 2^32 per sample, and
            // affine maps compose exactly.  Lane j folds its own `per` consecutive samples into one map, an inclusive scan
            // over the row (four DPP steps) composes the maps of the lanes before it, and the lane replays only its own
            // samples from its start seed.
            const u32 LCG_A = 196314165u, LCG_C = 907633515u;
            const int per = (subfr + 15) >> 4, i0 = per * j;
            u32 M = 1u, B = 0u;
            i32 pl_own[OWN];
#pragma unroll
            for (int t = 0; t < OWN; t++) {
                pl_own[t] = 0;
                if (t < per && i0 + t < subfr) {
                    pl_own[t] = pulses[pos + i0 + t];
                    M *= LCG_A;
                    B = B * LCG_A + LCG_C + (u32)pl_own[t];
                }
            }
            // (M, B) <- (M, B) o (M', B') of lane j - d, d = 1, 2, 4, 8; the identity where there is no such lane
#define OG_AFFINE_SCAN_STEP(d)                                                                                      \
    do {                                                                                                            \
        const u32 Mp = (u32)__builtin_amdgcn_update_dpp(1, (i32)M, 0x110 + (d) /* row_shr:d */, 0xf, 0xf, false);   \
        const u32 Bp = (u32)__builtin_amdgcn_update_dpp(0, (i32)B, 0x110 + (d), 0xf, 0xf, true);                    \
        B = M * Bp + B;                                                                                             \
        M = M * Mp;                                                                                                 \
    } while (0)
            OG_AFFINE_SCAN_STEP(1);
            OG_AFFINE_SCAN_STEP(2);
            OG_AFFINE_SCAN_STEP(4);
            OG_AFFINE_SCAN_STEP(8);
#undef OG_AFFINE_SCAN_STEP
            const u32 r0 = (u32)rand_seed;
            const u32 Me = (u32)__builtin_amdgcn_update_dpp(1, (i32)M, 0x111, 0xf, 0xf, false);
            const u32 Be = (u32)__builtin_amdgcn_update_dpp(0, (i32)B, 0x111, 0xf, 0xf, true); // (bound_ctrl: a lane without a source reads 0 -- no register to preset)
            u32 r = Me * r0 + Be;                                              // seed before the lane's first sample
            rand_seed = row_sum16(j == 15 ? (i32)(M * r0 + B) : 0);            // seed after the subframe, in every lane
#pragma unroll
            for (int t = 0; t < OWN; t++) {
                if (t < per && i0 + t < subfr) {
                    r = LCG_C + r * LCG_A;
                    const i32 pl = pl_own[t];
                    i32 exc = shl32(pl, 14);
                    if (exc > 0)
                        exc -= 80 << 4;
                    else if (exc < 0)
                        exc += 80 << 4;
                    exc += offset_Q10 << 4;
                    if ((i32)r < 0) exc = -exc;
                    r += (u32)pl;
                    resb[i0 + t] = exc;
                    if (lc) lc->exc_Q14[pos + i0 + t] = exc;
                }
            }
        }
        OG_ROW_SYNC();
        OG_MARK(62);
        if (voiced) {
            const int chunk = OG_MIN(16, lag - 2);
            for (int i0 = 0; i0 < subfr; i0 += chunk) {
                const int i = i0 + j;
                if (j < chunk && i < subfr) {
                    const i32 *p = &sLTP_Q15[sLTP_buf_idx + i - lag + 2];
                    i32 LTP_pred_Q13 = 2;
                    LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[0], B_Q14[0]);
                    LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-1], B_Q14[1]);
                    LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-2], B_Q14[2]);
                    LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-3], B_Q14[3]);
                    LTP_pred_Q13 = smlawb(LTP_pred_Q13, p[-4], B_Q14[4]);
                    const i32 res = addw(resb[i], shl32(LTP_pred_Q13, 1));
                    sLTP_Q15[sLTP_buf_idx + i] = shl32(res, 1);
                    resb[i] = res;
                }
                OG_ROW_SYNC();
            }
            sLTP_buf_idx += subfr;
        }
        OG_MARK(63);
        {
            // Transposed form of the predictor: lane j carries R_j = sum over u > j of A_u * s[n - (u - j)] -- what the samples
            // decoded so far contribute to the prediction j samples ahead -- so lane 0 holds the whole prediction of the next
            // sample.  A new sample costs every lane one product (its own tap times the sample, broadcast over the row) and one
            // add that also passes the accumulator one lane down: ten vector instructions per sample where the direct form
            // -- one product per lane and a four-step row reduction beside the saturating update -- took nineteen, in a kernel
            // that is bound by vector-ALU issue.  The sums are the same products added in another order, and silk_SMLAWB
            // wraps, so the order is free.
            //   R_j = sum_k smulwb(h_k, A[j + k]),  h_k = the k-th last output (lane k of sLPC), A[] = A_Q12, zero past `order`
            i32 R = 0, Arot = A_j;
#define OG_ROW_BCAST(v, n) __builtin_amdgcn_update_dpp(0, (v), 0x150 + (n) /* row_newbcast:n */, 0xf, 0xf, true)
#define OG_ACC_STEP(kk)                                                                             \
    if ((kk) < order) {                                                                             \
        R = addw(R, smulwb(OG_ROW_BCAST(sLPC, kk), Arot));                                          \
        Arot = __builtin_amdgcn_update_dpp(0, Arot, 0x101 /* row_shl:1 */, 0xf, 0xf, true);         \
    }
            OG_ACC_STEP(0) OG_ACC_STEP(1) OG_ACC_STEP(2) OG_ACC_STEP(3) OG_ACC_STEP(4) OG_ACC_STEP(5) OG_ACC_STEP(6) OG_ACC_STEP(7)
            OG_ACC_STEP(8) OG_ACC_STEP(9) OG_ACC_STEP(10) OG_ACC_STEP(11) OG_ACC_STEP(12) OG_ACC_STEP(13) OG_ACC_STEP(14) OG_ACC_STEP(15)
#undef OG_ACC_STEP
            const i32 bias = order >> 1; // (the rounding offset the prediction starts from, silk.cpp:1937)
            // silk_SMULWB(sn, A_j) = (sn * A_j) >> 16 = the high word of sn * (A_j << 16): ONE instruction (v_mul_hi_i32) where two
            // 24-bit multiplies, a shift and an add stood; the shifted accumulator arrives through the add's own DPP operand.  Four
            // samples per trip (a subframe is 40, 60 or 80): their residuals come as one 16-byte read a trip ahead, their outputs
            // leave as one 16-byte write.  7 vector instructions per sample (12 before), in the kernel's longest serial loop.
            // The rounding offset rides in the accumulators: every R_j carries it, and lane 15 -- whose shifted-in neighbour is the
            // zero beyond the row -- gets it back through the product's 64-bit addend (v_mad_i64_i32: the high word of
            // sn * A16 + (bias << 32) is mul_hi + bias), so the prediction needs no add of its own: 6 instructions per sample.
            const i32 A16 = shl32(A_j, 16);
            const i64 bias_in = (i64)(j == 15 ? bias : 0) << 32;
            R = addw(R, bias);
            typedef i32 i32x4 __attribute__((ext_vector_type(4)));
            i32x4 rq = *reinterpret_cast<const i32x4 *>(&resb[0]);
#pragma unroll 2
            for (int i = 0; i < subfr; i += 4) {
                const i32x4 nx = *reinterpret_cast<const i32x4 *>(&resb[i + 4 < subfr ? i + 4 : i]); // (its latency is off the chain)
                i32x4 sv;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const i32 sn0 = __builtin_elementwise_add_sat(rq[t], lshift_sat32(R, 4)); // (lane 0's R is the prediction, offset included)
                    const i32 sn = OG_ROW_BCAST(sn0, 0);
                    sv[t] = sn; // the residual is consumed: its slot keeps the sample (output scaling, next history); every lane
                                // of the row stores the same value to the same word
                    i64 prod;
                    u64 carry;
                    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(prod), "=s"(carry) : "v"(sn), "v"(A16), "v"(bias_in));
                    R = addw(__builtin_amdgcn_update_dpp(0, R, 0x101 /* row_shl:1 */, 0xf, 0xf, true), (i32)(prod >> 32));
                }
                *reinterpret_cast<i32x4 *>(&resb[i]) = sv;
                rq = nx;
            }
#undef OG_ROW_BCAST
        }
        OG_ROW_SYNC();
        for (int i = j; i < subfr; i += 16) xq[pos + i] = (i16)sat16(rshift_round(smulww(resb[i], Gain_Q10), 8));
        sLPC = resb[subfr - 1 - j]; // the last sixteen samples, most recent in lane 0: the next subframe's (or frame's) history
        pos += subfr;
        if (voiced) OG_ROW_SYNC(); // this subframe's sLTP_Q15 / xq writes before the next subframe's reads
    }
    OG_MARK(34);
    c->sLPC_Q14_buf[SILK_MAX_LPC - 1 - j] = sLPC;
    if (j == 0) {
        c->prev_gain_Q16 = prev_gain_Q16;
        c->lagPrev = k.pitchL[nb_subfr - 1];
        c->prevSignalType = k.signalType;
        c->first_frame_after_reset = 0;
    }
}
#endif

// 2x all-pass up-sampler for one channel, one lane (silk_resampler_private_up2_HQ silk.cpp:3515); the input
// stream is [delayBuf (inputDelay old samples) | in[0 .. inLen - inputDelay)] (silk_resampler silk.cpp:3676)
OG_DEVN void silk_up2_lane(SilkChannel *c, int ch, int inLen) {
    SilkLds &L = SL();
    const i16 *in = &L.xq[ch][1];
    i16 *up = L.u.out.up[ch];
    const int delay = c->rs_inputDelay;
    i32 S0 = c->rs_sIIR[0], S1 = c->rs_sIIR[1], S2 = c->rs_sIIR[2], S3 = c->rs_sIIR[3], S4 = c->rs_sIIR[4], S5 = c->rs_sIIR[5];
    const i32 a0 = rom_silk_up2_hq0[0], a1 = rom_silk_up2_hq0[1], a2 = rom_silk_up2_hq0[2];
    const i32 b0 = rom_silk_up2_hq1[0], b1 = rom_silk_up2_hq1[1], b2 = rom_silk_up2_hq1[2];
    for (int j = 0; j < 8; j++) up[j] = c->rs_sFIR[j];
    for (int t = 0; t < inLen; t++) {
        const i32 x = t < delay ? (i32)c->rs_delayBuf[t] : (i32)in[t - delay];
        const i32 in32 = shl32(x, 10);
        i32 Y, X, o1, o2;
        Y = in32 - S0; X = smulwb(Y, a0); o1 = S0 + X; S0 = in32 + X;
        Y = o1 - S1; X = smulwb(Y, a1); o2 = S1 + X; S1 = o1 + X;
        Y = o2 - S2; X = smlawb(Y, Y, a2); o1 = S2 + X; S2 = o2 + X;
        up[8 + 2 * t] = (i16)sat16(rshift_round(o1, 10));
        Y = in32 - S3; X = smulwb(Y, b0); o1 = S3 + X; S3 = in32 + X;
        Y = o1 - S4; X = smulwb(Y, b1); o2 = S4 + X; S4 = o1 + X;
        Y = o2 - S5; X = smlawb(Y, Y, b2); o1 = S5 + X; S5 = o2 + X;
        up[8 + 2 * t + 1] = (i16)sat16(rshift_round(o1, 10));
    }
    c->rs_sIIR[0] = S0; c->rs_sIIR[1] = S1; c->rs_sIIR[2] = S2; c->rs_sIIR[3] = S3; c->rs_sIIR[4] = S4; c->rs_sIIR[5] = S5;
    for (int j = 0; j < 8; j++) c->rs_sFIR[j] = up[2 * inLen + j];
    for (int j = 0; j < delay; j++) c->rs_delayBuf[j] = in[inLen - delay + j];
}

// silk_decoder_set_fs + silk_resampler_init for the 20 ms / 48 kHz case (silk.cpp:978, :3590)
OG_DEV void silk_set_fs(SilkChannel *c, int fs_kHz) {
    if (c->fs_kHz == fs_kHz) return; // fs_API_hz is constant, frame length follows fs_kHz
    OG_SYNC();
    OG_FOR_LANES(i, 320) c->outBuf[i] = 0;
    OG_FOR_LANES(i, 16) {
        c->sLPC_Q14_buf[i] = 0;
        c->rs_delayBuf[i] = 0;
    }
    OG_FOR_LANES(i, 8) c->rs_sFIR[i] = 0;
    OG_FOR_LANES(i, 6) c->rs_sIIR[i] = 0;
    OG_SYNC();
    if (OG_LANE == 0) {
        c->first_frame_after_reset = 1;
        c->lagPrev = 100;
        c->LastGainIndex = 10;
        c->prevSignalType = 0;
        const i32 Fs_in = fs_kHz * 1000, Fs_out = 48000;
        const int in_id = ((Fs_in >> 12) - (Fs_in > 16000)) - 1; // rateID silk.h:397
        c->rs_inputDelay = rom_silk_delay_dec[in_id * 5 + 4];
        c->rs_fs_in_kHz = fs_kHz;
        i32 inv = shl32(shl32(Fs_in, 15) / Fs_out, 2);
        while (smulww(inv, Fs_out) < shl32(Fs_in, 1)) inv++;
        c->rs_invRatio_Q16 = inv;
        c->fs_kHz = fs_kHz;
    }
    OG_SYNC();
}

#ifndef OG_HOST_EMUL
// The same up-sampler as a systolic array: the two output phases of a channel are two cascades of three first-order all-pass
// sections; lane (phase, section) owns one section's state, takes its input from the lane before it (row_shr:1, the previous
// step's output) and works on sample t = step - section.  inLen + 2 steps of one section each instead of inLen steps of six.
// Must be entered by all 64 lanes.  Phase 0 of channel n runs in lanes 0 - 2 of DPP row n, phase 1 in lanes 0 - 2 of row n + 2:
// every cascade's first section is then lane 0 of a row -- the one lane row_shr:1 has no source for, which keeps the value the
// register held before: the input sample.
// A step is six vector instructions (round 3: seventeen, and an exec-mask branch around the store): shift-or-input in one DPP
// move; the section's product as the high word of Y * (coef << 16) (silk_SMULWB in one v_mul_hi_i32); two three-operand adds; the
// last sections store their 32-bit outputs as they are -- phase 0 over the input samples it has consumed, phase 1 into the row's
// other half -- and the rounding to 16 bits is done afterwards by all 64 lanes at once; the inner sections, which have nothing
// to store, write to a sink instead of branching.
OG_DEV i32 og_add3(i32 a, i32 b, i32 c) {
    i32 r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
OG_DEV void silk_up2_rows(SilkState *st, int channels, int inLen) {
    SilkLds &L = SL();
    const int row = OG_LANE >> 4, j = OG_LANE & 15;
    if (row < channels) {
        SilkChannel *c = &st->ch[row];
        const i16 *in = silk_xq_row(row, inLen) + 1; // (the tight layout only sees 20 ms frames: inLen is its ltp_mem)
        const int delay = c->rs_inputDelay;
        if (j < 8) silk_up_row(row)[j] = c->rs_sFIR[j];
        // the input stream [delayBuf | in] as 32-bit Q10 values, staged once (the sLTP_Q15 row of this channel is free by now)
        i32 *in32 = L.u.core.sLTP_Q15[row];
        if (j < delay) in32[j] = shl32((i32)c->rs_delayBuf[j], 10); // (delay is 0, 4 or 7: rom_silk_delay_dec)
        for (int t = j; t < inLen - delay; t += 16) in32[delay + t] = shl32((i32)in[t], 10);
        // (the next frame's delay line, taken now: in the tight layout the FIR's input rows are written over the frame below)
        if (j < delay) c->rs_delayBuf[j] = in[inLen - delay + j];
    }
    OG_SYNC();
    const int ch = row & 1, ph = row >> 1;
    if (ch < channels && j < 3) {
        SilkChannel *c = &st->ch[ch];
        const int sec = j;
        const bool first = sec == 0, last = sec == 2;
        const i32 coef = ph ? rom_silk_up2_hq1[sec] : rom_silk_up2_hq0[sec], coef16 = shl32(coef, 16);
        const i32 m2 = last ? -1 : 0;
        i32 S = c->rs_sIIR[3 * ph + sec], out = 0;
        const i32 *in32 = L.u.core.sLTP_Q15[ch];
        i32 *const raw = L.u.core.sLTP_Q15[ch] + ph * SILK_LDS_FRAME; // the last section's outputs, sample t at [t]
        // Section `sec` handles input sample t = u - sec in step u.  The two steps that fill the pipeline and the two that drain it
        // run the general body (a section without a sample keeps its state); the inLen - 2 steps in between have every section
        // at work and carry no such bookkeeping.
        auto edge_step = [&](int u) {
            const i32 prev_out = __builtin_amdgcn_update_dpp(0, out, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
            const int t = u - sec;
            const bool live = (unsigned)t < (unsigned)inLen;
            const i32 v = first ? in32[live ? t : 0] : prev_out;
            const i32 Y = v - S;
            const i32 X = smmul(Y, coef16) + (Y & m2);
            const i32 o = S + X;
            out = live ? o : out;
            S = live ? v + X : S;
            if (live && last) raw[t] = o;
        };
#define OG_UP2_STEP(in_u, slot)                                                                                          \
    do {                                                                                                                 \
        const i32 v = __builtin_amdgcn_update_dpp((in_u), out, 0x111 /* row_shr:1 */, 0xf, 0xf, false);                  \
        const i32 Y = v - S, P = smmul(Y, coef16), Ym = Y & m2;                                                          \
        out = og_add3(S, P, Ym);                                                                                         \
        S = og_add3(v, P, Ym);                                                                                           \
        wr[slot] = out;                                                                                                  \
    } while (0)
        edge_step(0);
        edge_step(1);
        {
            // steps 2 .. inLen - 1 (inLen is 160, 240 or 320): two single ones, then four per trip with the trip's inputs read as
            // one 16-byte word a trip ahead.  wr: where sample u - 2 goes -- only the last sections' pointer moves.
            typedef i32 i32x4 __attribute__((ext_vector_type(4)));
            i32 *wr = last ? raw : &L.u.out.sink[row][sec][0];
            const int adv = last ? 4 : 0;
            OG_UP2_STEP(in32[2], 0);
            OG_UP2_STEP(in32[3], 1);
            wr += adv >> 1;
            i32x4 cur = *reinterpret_cast<const i32x4 *>(&in32[4]);
            for (int u = 4; u < inLen; u += 4) {
                const i32x4 nxt = *reinterpret_cast<const i32x4 *>(&in32[u + 4 < inLen ? u + 4 : u]);
                OG_UP2_STEP(cur[0], 0);
                OG_UP2_STEP(cur[1], 1);
                OG_UP2_STEP(cur[2], 2);
                OG_UP2_STEP(cur[3], 3);
                wr += adv;
                cur = nxt;
            }
        }
#undef OG_UP2_STEP
        edge_step(inLen);
        edge_step(inLen + 1);
        c->rs_sIIR[3 * ph + sec] = S;
    }
    OG_SYNC();
    // the 32-bit outputs to the FIR's 16-bit input rows, phases interleaved (silk.cpp:3515: silk_SAT16(silk_RSHIFT_ROUND(out, 10)))
    for (int n = 0; n < channels; n++) {
        const i32 *in32 = L.u.core.sLTP_Q15[n];
        i16 *up = silk_up_row(n) + 8;
        OG_FOR_LANES(m, 2 * inLen) up[m] = (i16)sat16(rshift_round(in32[(m & 1) * SILK_LDS_FRAME + (m >> 1)], 10));
    }
    OG_SYNC();
    if (row < channels) {
        SilkChannel *c = &st->ch[row];
        const i16 *up = silk_up_row(row);
        if (j < 8) c->rs_sFIR[j] = up[2 * inLen + j];
    }
}
#endif

#include "og_silk_loss.hpp"

OG_DEV void silk_stereo_decode_pred(Rc &rc, i32 pred_Q13[2]) { // silk.cpp:592
    int ix[2][3];
    int n = rc_icdf(rc, rom_silk_stereo_joint_icdf, 8);
    ix[0][2] = n / 5;
    ix[1][2] = n - 5 * ix[0][2];
    for (n = 0; n < 2; n++) {
        ix[n][0] = rc_icdf(rc, rom_silk_uniform3_icdf, 8);
        ix[n][1] = rc_icdf(rc, rom_silk_uniform5_icdf, 8);
    }
    for (n = 0; n < 2; n++) {
        ix[n][0] += 3 * ix[n][2];
        const i32 low_Q13 = rom_silk_stereo_pred_q13[ix[n][0]];
        const i32 step_Q13 = smulwb(rom_silk_stereo_pred_q13[ix[n][0] + 1] - low_Q13, 6554);
        pred_Q13[n] = smlabb(low_Q13, step_Q13, 2 * ix[n][1] + 1);
    }
    pred_Q13[0] -= pred_Q13[1];
}

// Decode one 20 ms SILK frame (mid + side / mono) into S.pcm_silk (48 kHz, interleaved when stereo).
// Returns 0 or a non-zero error (wave-uniform).
// `rec` != null: the frame's entropy half comes from the parse kernel's record and `rc` is not touched.
// silk_Decode silk.cpp:1481 for one Opus frame: the packet header, then its internal SILK frames one after the other.
// payload_ms: 20 in reference mode (the reference pins it, Q6); RFC mode also 10 (two subframes), 40 and 60 (two / three
// internal frames of 20 ms; silk.cpp:1522-1540 derives both from payloadSize_ms).  After every internal frame its 48 kHz PCM
// (interleaved over `channels`) is in SL().u.out.pcm and `emit(frame index, samples per channel)` is called.
// REC_ONLY: the instantiation of the split path's synthesis kernel -- `rec` is always there (one 20 ms frame), none of the
// wave-uniform entropy decoding is compiled in, and neither is its LDS object (PW()).
// silk_LBRR_flags_2_iCDF / _3_iCDF (RFC 6716 table 4); only packets of two / three internal frames read them
OPUS_ROM uint8_t rom_silk_lbrr_flags_2_icdf[3] = {203, 150, 0};
OPUS_ROM uint8_t rom_silk_lbrr_flags_3_icdf[7] = {215, 195, 166, 125, 110, 82, 0};

// `loss` (RFC mode): the stream's loss-concealment state.  With it every decoded frame also leaves what a later
// concealment starts from (silk_PLC_update, the comfort-noise estimate) and is smoothed when it follows a loss; `lost` = 1 is
// silk_Decode's lostFlag == FLAG_PACKET_LOST (silk.cpp:1481-1779): nothing is read from `rc`, one frame of payload_ms (10 or
// 20) is concealed at the rate of the last decoded frame (internal_hz == 0); `lost` = 2 is FLAG_DECODE_LBRR: the packet's
// forward error correction data -- every internal frame's LBRR copy where the packet carries one, a concealment where not.
// The synthesis then runs one lane per channel.
template <bool REC_ONLY, class Emit>
OG_DEV int silk_decode_packet(SilkState *s, Rc &rc, int channels, int internal_hz, int payload_ms, const SilkRec *rec, Emit &&emit,
                              LossState *loss = nullptr, int lost = 0) {
    SilkLds &L = SL();
    const int fs_kHz = internal_hz ? (internal_hz >> 10) + 1 : (int)OG_UNI(s->ch[0].fs_kHz);
    if (fs_kHz != 8 && fs_kHz != 12 && fs_kHz != 16) return -200;
    const int nF = (REC_ONLY || rec) ? 1 : payload_ms == 40 ? 2 : payload_ms == 60 ? 3 : 1;
    const int nb_subfr = (REC_ONLY || rec) ? 4 : payload_ms == 10 ? 2 : 4;
    const int frame_length = nb_subfr * 5 * fs_kHz, ltp_mem = 20 * fs_kHz;
    OG_SYNC();
    // first frame of the packet: nFramesDecoded = 0 for the coded channels
    if (channels > s->nChannelsInternal) silk_chan_init(&s->ch[1], loss ? &loss->silk[1] : nullptr);
    for (int n = 0; n < channels; n++) silk_set_fs(&s->ch[n], fs_kHz);
    if (channels == 2 && (s->nChannelsAPI == 1 || s->nChannelsInternal == 1)) {
        OG_SYNC();
        if (OG_LANE == 0) {
            s->pred_prev_Q13[0] = s->pred_prev_Q13[1] = 0;
            s->sSide[0] = s->sSide[1] = 0;
        }
        OG_SYNC();
    }
    // persistent scalars of both channels into registers (uniform)
    i32 ecType[2], ecLag[2], lastGain[2], ffar[2];
    for (int n = 0; n < 2; n++) {
        ecType[n] = s->ch[n].ec_prevSignalType;
        ecLag[n] = s->ch[n].ec_prevLagIndex;
        lastGain[n] = s->ch[n].LastGainIndex;
        ffar[n] = s->ch[n].first_frame_after_reset;
    }
    int vad[2][3] = {{0, 0, 0}, {0, 0, 0}}, lbrr[2][3] = {{0, 0, 0}, {0, 0, 0}};
    OG_MARK(30);
    i32 MS_pred_Q13[2] = {0, 0};
    int decode_only_middle = 0;
    if (REC_ONLY || rec) { // the entropy half was done by the lane-per-frame parse kernel (og_silk_parse.hpp)
        MS_pred_Q13[0] = OG_UNI(rec->MS_pred_Q13[0]);
        MS_pred_Q13[1] = OG_UNI(rec->MS_pred_Q13[1]);
        decode_only_middle = OG_UNI(rec->decode_only_middle);
        for (int n = 0; n < 2; n++) {
            ecType[n] = OG_UNI(rec->ch[n].ec_prevSignalType);
            ecLag[n] = OG_UNI(rec->ch[n].ec_prevLagIndex);
        }
    } else if (lost == 1) { // silk.cpp:1632-1635: the predictors stay; decode_only_middle is silk_Decode's local, zero
        MS_pred_Q13[0] = s->pred_prev_Q13[0];
        MS_pred_Q13[1] = s->pred_prev_Q13[1];
    } else if constexpr (!REC_ONLY) {
        int lbrr_flag[2] = {0, 0};
        for (int n = 0; n < channels; n++) { // silk.cpp:1568-1573: every channel's VAD flags and LBRR flag first
            for (int i = 0; i < nF; i++) vad[n][i] = rc_bit_logp(rc, 1);
            lbrr_flag[n] = rc_bit_logp(rc, 1);
        }
        for (int n = 0; n < channels; n++) // then, per channel, which of its frames carry LBRR data (silk.cpp:1576-1586)
            if (lbrr_flag[n]) {
                if (nF == 1)
                    lbrr[n][0] = 1;
                else {
                    const int sym = rc_icdf(rc, nF == 2 ? rom_silk_lbrr_flags_2_icdf : rom_silk_lbrr_flags_3_icdf, 8) + 1;
                    for (int i = 0; i < nF; i++) lbrr[n][i] = (sym >> i) & 1;
                }
            }
        for (int i = 0; i < nF && !lost; i++) // regular decoding reads past the LBRR frames (silk.cpp:1590-1616)
            for (int n = 0; n < channels; n++)
                if (lbrr[n][i]) {
                    if (channels == 2 && n == 0) {
                        silk_stereo_decode_pred(rc, MS_pred_Q13);
                        if (lbrr[1][i] == 0) decode_only_middle = rc_icdf(rc, rom_silk_mid_only_icdf, 8);
                    }
                    const int condCoding = (i > 0 && lbrr[n][i - 1]) ? 2 : 0;
                    silk_decode_indices(&s->ch[n], L.ctrl[n], rc, fs_kHz, vad[n][i], 1, condCoding, ecType[n], ecLag[n], nb_subfr);
                    silk_decode_pulses(rc, n, L.ctrl[n].signalType, L.ctrl[n].quantOffsetType, frame_length);
                }
    }
    int prev_dom = s->prev_decode_only_middle;
    for (int fi = 0; fi < nF; fi++) {
        if constexpr (!REC_ONLY) {
            if (!rec && !lost && channels == 2) {
                silk_stereo_decode_pred(rc, MS_pred_Q13);
                decode_only_middle = vad[1][fi] == 0 ? rc_icdf(rc, rom_silk_mid_only_icdf, 8) : 0;
            } else if (lost == 2) { // silk.cpp:1620-1637 with FLAG_DECODE_LBRR (decode_only_middle is a local of the call: zero)
                decode_only_middle = 0;
                if (channels == 2 && lbrr[0][fi]) {
                    silk_stereo_decode_pred(rc, MS_pred_Q13);
                    if (lbrr[1][fi] == 0) decode_only_middle = rc_icdf(rc, rom_silk_mid_only_icdf, 8);
                } else {
                    MS_pred_Q13[0] = s->pred_prev_Q13[0];
                    MS_pred_Q13[1] = s->pred_prev_Q13[1];
                }
            }
        }
        const int restart_side = channels == 2 && decode_only_middle == 0 && prev_dom == 1;
        if (restart_side) { // side channel restarts (silk.cpp:1639)
            SilkChannel *c1 = &s->ch[1];
            OG_SYNC();
            OG_FOR_LANES(i, 320) c1->outBuf[i] = 0;
            OG_FOR_LANES(i, 16) c1->sLPC_Q14_buf[i] = 0;
            if (OG_LANE == 0) {
                c1->lagPrev = 100;
                c1->prevSignalType = 0;
            }
            OG_SYNC();
            lastGain[1] = 10;
            ffar[1] = 1;
        }
        const int has_side = lost ? (!prev_dom || (channels == 2 && lost == 2 && lbrr[1][fi])) : !decode_only_middle; // silk.cpp:1664-1671
        i32 plc_invGain_Q30[2] = {0, 0};
        int conceal[2] = {0, 0}; // the channel has no data to decode for this frame (silk_decode_frame silk.cpp:1987, :2021)
        for (int n = 0; n < channels; n++) {
            L.ctrl[n].coded = (n == 0 || has_side);
            conceal[n] = lost == 1 || (lost == 2 && !lbrr[n][fi]);
            if (L.ctrl[n].coded && conceal[n]) {
                if constexpr (!REC_ONLY) {
                    // a concealed frame: the serial part runs one lane per channel below; what needs the wave's scratch runs here
                    // (silk_PLC silk.cpp:2871-2877; silk_PLC_conceal :2992, :3015-3039)
                    SilkLossChannel *lc = &loss->silk[n];
                    const int order = fs_kHz == 16 ? 16 : 10;
                    OG_SYNC();
                    if (OG_LANE == 0) {
                        silk_plc_rate_check(lc, fs_kHz, frame_length);
                        if (ffar[n])
                            for (int i = 0; i < SILK_MAX_LPC; i++) lc->plc_prevLPC_Q12[i] = 0;
                        ArrV<i16, 1> a = {lc->plc_prevLPC_Q12};
                        silk_bwexpander16(a, order, 64881); // SILK_FIX_CONST(BWE_COEF = 0.99, 16)
                        L.ctrl[n].signalType = s->ch[n].prevSignalType;
                    }
                    OG_SYNC();
                    if (OG_UNI(lc->lossCnt) == 0 && OG_UNI(s->ch[n].prevSignalType) != 2) {
                        ArrV<i16, 1> a = {lc->plc_prevLPC_Q12};
                        plc_invGain_Q30[n] = silk_inverse_pred_gain<SilkParWave>(a, order);
                    }
                    OG_SYNC();
                }
            } else if (L.ctrl[n].coded) {
                // FrameIndex = channel 0's nFramesDecoded - n, and channel 0's count has been stepped by the time channel 1
                // gets here (silk.cpp:1676-1700): the frame's index for both; <= 0 -> independent coding
                const int FrameIndex = fi;
                const int condCoding = FrameIndex <= 0 ? 0 : lost == 2 ? (lbrr[n][fi - 1] ? 2 : 0) : (n > 0 && prev_dom) ? 1 : 2;
                OG_MARK(30);
                if (REC_ONLY || rec) { // parameters + indices (68 words laid out like SilkCtrl) from the record
                    OG_SYNC();
                    const i32 *src = rec->ch[n].pitchL;
                    i32 *dst = L.ctrl[n].pitchL;
                    OG_FOR_LANES(i, SILK_REC_CTRL_WORDS) dst[i] = src[i];
                    lastGain[n] = OG_UNI(rec->ch[n].LastGainIndex);
                    OG_FOR_LANES(i, fs_kHz == 16 ? 16 : 10) s->ch[n].prevNLSF_Q15[i] = rec->ch[n].nlsf[i];
                    OG_SYNC();
                } else if constexpr (!REC_ONLY) {
                    silk_decode_indices(&s->ch[n], L.ctrl[n], rc, fs_kHz, vad[n][fi], lost == 2, condCoding, ecType[n], ecLag[n], nb_subfr);
                    OG_MARK(31);
                    silk_decode_pulses(rc, n, L.ctrl[n].signalType, L.ctrl[n].quantOffsetType, frame_length);
                    OG_MARK(32);
                    silk_decode_parameters<SilkParWave>(s->ch[n].prevNLSF_Q15, L.ctrl[n], fs_kHz, condCoding, lastGain[n], ffar[n],
                                                        nb_subfr);
                    OG_SYNC();
                    OG_FOR_LANES(i, fs_kHz == 16 ? 16 : 10) s->ch[n].prevNLSF_Q15[i] = PW().nlsf[i];
                    if (loss && OG_UNI(loss->silk[n].lossCnt) && OG_LANE == 0) { // silk.cpp:860-864: BWE_AFTER_LOSS_Q16
                        ArrV<i16, 1> a0 = {L.ctrl[n].PredCoef_Q12[0]}, a1 = {L.ctrl[n].PredCoef_Q12[1]};
                        silk_bwexpander16(a0, fs_kHz == 16 ? 16 : 10, 63570);
                        silk_bwexpander16(a1, fs_kHz == 16 ? 16 : 10, 63570);
                    }
                    OG_SYNC();
                }
                ffar[n] = 0; // (the synthesis below clears it in the state)
                OG_MARK(33);
            }
        }
        OG_SYNC();
        if (OG_LANE == 0) {
            for (int n = 0; n < 2; n++) {
                s->ch[n].ec_prevSignalType = ecType[n];
                s->ch[n].ec_prevLagIndex = ecLag[n];
                s->ch[n].LastGainIndex = lastGain[n];
            }
            if (restart_side) s->ch[1].first_frame_after_reset = 1;
        }
        OG_SYNC();
        // ---- synthesis: stage the output history, then lane n = channel n
        OG_MARK(34);
        // where the channels' pulses are: in the parse record (read where they lie, in HBM) or in the row they were decoded into
        const i16 *pulse_row[2];
        if constexpr (REC_ONLY) {
            pulse_row[0] = rec->ch[0].pulses;
            pulse_row[1] = rec->ch[1].pulses;
        } else {
            pulse_row[0] = rec ? rec->ch[0].pulses : PW().pulses[0];
            pulse_row[1] = rec ? rec->ch[1].pulses : PW().pulses[1];
        }
        for (int n = 0; n < channels; n++)
            if (L.ctrl[n].coded) OG_FOR_LANES(i, ltp_mem) silk_hist_row(n)[i] = s->ch[n].outBuf[i];
        OG_SYNC();
        // one lane per channel: the decoded frame's synthesis, or a lost frame's concealment
        auto lane_synth = [&](int n) {
            if (!L.ctrl[n].coded) {
                for (int i = 0; i < frame_length; i++) silk_xq_row(n, ltp_mem)[2 + i] = 0;
                return;
            }
            if constexpr (!REC_ONLY) {
                if (loss) {
                    SilkLossChannel *lc = &loss->silk[n];
                    if (conceal[n])
                        silk_plc_conceal_lane(&s->ch[n], lc, n, fs_kHz, nb_subfr, plc_invGain_Q30[n]);
                    else { // silk_decode_frame silk.cpp:2008-2015: the core, then silk_PLC(lost = 0), then lossCnt = 0
                        silk_decode_core_lane(&s->ch[n], n, fs_kHz, pulse_row[n], nb_subfr, lc);
                        silk_plc_rate_check(lc, fs_kHz, frame_length);
                        silk_plc_update_lane(lc, L.ctrl[n], fs_kHz, nb_subfr);
                        lc->lossCnt = 0;
                    }
                    return;
                }
            }
            silk_decode_core_lane(&s->ch[n], n, fs_kHz, pulse_row[n], nb_subfr);
        };
#ifdef OG_HOST_EMUL
        OG_FOR_LANES(n, channels) lane_synth(n);
#else
        bool rows = true; // the row form takes decoded frames; a concealed channel (RFC mode: lost packets, missing LBRR data) the one-lane form
        if constexpr (!REC_ONLY) // (the split path's synthesis kernel must not even reference the one-lane form: a call to it sizes
                                 // the kernel's register file -- 147 instead of 90, three waves per SIMD instead of four)
            for (int n = 0; n < channels; n++) rows = rows && !(L.ctrl[n].coded && conceal[n]);
        if constexpr (REC_ONLY)
            silk_decode_core_rows<false>(s, fs_kHz, channels, pulse_row[0], pulse_row[1]);
        else if (rows) {
            if (nb_subfr == 4 && !loss)
                silk_decode_core_rows<false>(s, fs_kHz, channels, pulse_row[0], pulse_row[1]);
            else
                silk_decode_core_rows<true>(s, fs_kHz, channels, pulse_row[0], pulse_row[1], nb_subfr, loss);
            if constexpr (!REC_ONLY) {
                if (loss) { // silk_decode_frame silk.cpp:2008-2015: behind the core, silk_PLC(lost = 0), then lossCnt = 0
                    OG_SYNC();
                    OG_FOR_LANES(n, channels) {
                        if (L.ctrl[n].coded) {
                            SilkLossChannel *lc = &loss->silk[n];
                            silk_plc_rate_check(lc, fs_kHz, frame_length);
                            silk_plc_update_lane(lc, L.ctrl[n], fs_kHz, nb_subfr);
                            lc->lossCnt = 0;
                        }
                    }
                }
            }
        } else
            OG_FOR_LANES(n, channels) lane_synth(n);
#endif
        OG_SYNC();
        OG_TAP(40); // decoder control + core output of every coded channel (host emulation only)
        OG_MARK(35);
        // outBuf update (silk.cpp:2031-2034): the last ltp_mem_length samples -- for 20 ms frames exactly this frame, for 10 ms
        // frames the second half of the old history (still staged in LDS) followed by this frame
        for (int n = 0; n < channels; n++)
            if (L.ctrl[n].coded) {
                const int keep = ltp_mem - frame_length;
                OG_FOR_LANES(i, ltp_mem) s->ch[n].outBuf[i] = i < keep ? silk_hist_row(n)[frame_length + i] : silk_xq_row(n, ltp_mem)[2 + i - keep];
            }
        OG_SYNC();
        if constexpr (!REC_ONLY) {
            if (loss) { // silk_decode_frame silk.cpp:2036-2044: comfort noise estimate / generation, then the glue to the last frame
                const int order = fs_kHz == 16 ? 16 : 10;
                for (int n = 0; n < channels; n++) {
                    if (!L.ctrl[n].coded) continue;
                    SilkLossChannel *lc = &loss->silk[n];
                    if (OG_LANE == 0 && lc->cng_fs_kHz != fs_kHz) { // silk_CNG_Reset silk.cpp:1327
                        const i32 step = 32767 / (order + 1);
                        for (int i = 0; i < order; i++) lc->cng_smth_NLSF_Q15[i] = (i16)((i + 1) * step);
                        lc->cng_smth_Gain_Q16 = 0;
                        lc->cng_rand_seed = 3176576;
                        lc->cng_fs_kHz = fs_kHz;
                    }
                    OG_SYNC();
                    if (OG_UNI(lc->lossCnt)) { // the smoothed NLSFs as a filter, into the (free) whitening row of the channel
                        OG_FOR_LANES(i, order) PW().nlsf[i] = lc->cng_smth_NLSF_Q15[i];
                        OG_SYNC();
                        ArrV<i16, 1> a = {L.u.core.sLTP[n]};
                        silk_nlsf2a<SilkParWave>(a, SilkParWave::nlsf(), order);
                        OG_SYNC();
                    }
                }
                OG_FOR_LANES(n, channels) {
                    if (L.ctrl[n].coded) {
                        silk_cng_lane(&s->ch[n], &loss->silk[n], n, fs_kHz, nb_subfr, L.u.core.sLTP[n]);
                        silk_glue_lane(&loss->silk[n], n, frame_length);
                    }
                }
                OG_SYNC();
            }
        }
        // ---- stereo un-mixing (silk_stereo_MS_to_LR silk.cpp:4028) or mono look-back buffering (:1705)
        if (channels == 2) {
            i16 *x1 = silk_xq_row(0, ltp_mem), *x2 = silk_xq_row(1, ltp_mem);
            if (OG_LANE == 0) {
                x1[0] = s->sMid[0]; x1[1] = s->sMid[1];
                x2[0] = s->sSide[0]; x2[1] = s->sSide[1];
                s->sMid[0] = x1[frame_length]; s->sMid[1] = x1[frame_length + 1];
                s->sSide[0] = x2[frame_length]; s->sSide[1] = x2[frame_length + 1];
            }
            const i32 pp0 = s->pred_prev_Q13[0], pp1 = s->pred_prev_Q13[1];
            const i32 denom_Q16 = (1 << 16) / (8 * fs_kHz);
            const i32 delta0 = rshift_round(smulbb(MS_pred_Q13[0] - pp0, denom_Q16), 16);
            const i32 delta1 = rshift_round(smulbb(MS_pred_Q13[1] - pp1, denom_Q16), 16);
            OG_SYNC();
            i32 side_new[(SILK_LDS_FRAME + OG_NLANES - 1) / OG_NLANES]; // each lane holds its own results until every lane has read the old side signal
            int cnt = 0;
            OG_FOR_LANES(n, frame_length) {
                const i32 p0 = n < 8 * fs_kHz ? pp0 + (n + 1) * delta0 : MS_pred_Q13[0];
                const i32 p1 = n < 8 * fs_kHz ? pp1 + (n + 1) * delta1 : MS_pred_Q13[1];
                i32 sum = shl32(((i32)x1[n] + (i32)x1[n + 2]) + shl32((i32)x1[n + 1], 1), 9);
                sum = smlawb(shl32((i32)x2[n + 1], 8), sum, p0);
                sum = smlawb(sum, shl32((i32)x1[n + 1], 11), p1);
                side_new[cnt++] = sat16(rshift_round(sum, 8));
            }
            OG_SYNC();
            cnt = 0;
            OG_FOR_LANES(n, frame_length) {
                const i32 m = x1[n + 1], sd = side_new[cnt++];
                x1[n + 1] = (i16)sat16(m + sd);
                x2[n + 1] = (i16)sat16(m - sd);
            }
            if (OG_LANE == 0) {
                s->pred_prev_Q13[0] = tr16(MS_pred_Q13[0]);
                s->pred_prev_Q13[1] = tr16(MS_pred_Q13[1]);
            }
            OG_SYNC();
        } else {
            if (OG_LANE == 0) {
                i16 *x1 = silk_xq_row(0, ltp_mem);
                x1[0] = s->sMid[0]; x1[1] = s->sMid[1];
                s->sMid[0] = x1[frame_length]; s->sMid[1] = x1[frame_length + 1];
            }
            OG_SYNC();
        }
        // ---- resample to 48 kHz: 2x all-pass per channel (serial in time), then lane-parallel FIR interpolation
        OG_MARK(36);
#ifdef OG_HOST_EMUL
        OG_FOR_LANES(n, channels) silk_up2_lane(&s->ch[n], n, frame_length);
#else
        silk_up2_rows(s, channels, frame_length);
#endif
        OG_FOR_LANES(i, 48) L.u.out.taps[i] = rom_silk_fir12_taps8[i];
        OG_SYNC();
        OG_MARK(37);
        int out_total = 0;
        {
            const i32 inv = s->ch[0].rs_invRatio_Q16; // both channels run at the same rate
            // batches of the reference: [0, fs_kHz), then chunks of 10*fs_kHz (silk.cpp:3676, :3475)
            int t0 = 0, out0 = 0;
            while (t0 < frame_length) {
                const int nIn = t0 == 0 ? fs_kHz : OG_MIN(frame_length - t0, 10 * fs_kHz);
                const i32 max_index_Q16 = shl32(nIn, 17);
                const int count = (int)udiv((u32)max_index_Q16 + (u32)inv - 1u, (u32)inv);
                for (int n = 0; n < channels; n++) // (a loop per channel: splitting one index by `count` costs a software division per output)
                OG_FOR_LANES(m, count) {
                    const i32 index_Q16 = m * inv;
                    const int t = smulwb(index_Q16 & 0xFFFF, 12);
                    const i16 *b = silk_up_row(n) + 2 * t0 + (index_Q16 >> 16);
#ifdef OG_HOST_EMUL
                    const i16 *f0 = &rom_silk_frac_fir12[4 * t], *f1 = &rom_silk_frac_fir12[4 * (11 - t)];
                    i32 res = smulbb(b[0], f0[0]);
                    res = smlabb(res, b[1], f0[1]);
                    res = smlabb(res, b[2], f0[2]);
                    res = smlabb(res, b[3], f0[3]);
                    res = smlabb(res, b[4], f1[3]);
                    res = smlabb(res, b[5], f1[2]);
                    res = smlabb(res, b[6], f1[1]);
                    res = smlabb(res, b[7], f1[0]);
#else
                    // the eight products two at a time (v_dot2_i32_i16 adds like silk_SMLABB, mod 2^32): the samples as the four
                    // words they lie in, the phase's taps packed the same way (rom_silk_fir12_taps8)
                    typedef u32 u32x4u __attribute__((ext_vector_type(4), aligned(2)));
                    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4u x = *reinterpret_cast<const u32x4u *>(b);
                    const u32x4 f = *reinterpret_cast<const u32x4 *>(&L.u.out.taps[4 * t]); // (from LDS: the ROM's latency was the loop's)
                    i32 res;
                    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(res) : "v"(x[0]), "v"(f[0]));
                    asm("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(res) : "v"(x[1]), "v"(f[1]));
                    asm("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(res) : "v"(x[2]), "v"(f[2]));
                    asm("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(res) : "v"(x[3]), "v"(f[3]));
#endif
                    SL().u.out.pcm[(out0 + m) * channels + n] = (i16)sat16(rshift_round(res, 15));
                }
                t0 += nIn;
                out0 += count;
            }
            out_total = out0;
        }
        OG_SYNC();
        OG_MARK(38);
        if (lost != 1) { // silk.cpp:1772-1778
            prev_dom = decode_only_middle;
            if (OG_LANE == 0) s->prev_decode_only_middle = decode_only_middle;
        }
        emit(fi, out_total);
        OG_SYNC();
    }
    if (OG_LANE == 0) {
        s->nChannelsAPI = channels;
        s->nChannelsInternal = channels;
        if (lost == 1) // no gain clamping across a loss (silk.cpp:1772-1776)
            for (int n = 0; n < channels; n++) s->ch[n].LastGainIndex = 10;
    }
    OG_SYNC();
    return 0;
}

// one 20 ms frame per packet: what the reference decodes (Q6)
template <bool REC_ONLY = false>
OG_DEV int silk_decode_20ms(SilkState *s, Rc &rc, int channels, int internal_hz, const SilkRec *rec = nullptr) {
    return silk_decode_packet<REC_ONLY>(s, rc, channels, internal_hz, 20, rec, [](int, int) {});
}

} // namespace og
