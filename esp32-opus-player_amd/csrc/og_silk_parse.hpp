// og_silk_parse.hpp -- SILK entropy decoding with ONE FRAME PER LANE (k_silk_parse).
//
// SILK's side information and excitation pulses are a long chain of inverse-CDF symbols (several hundred per frame
// at 20 ms stereo) with no data-parallel work in between: run one frame per wave it keeps a 64-lane SIMD busy with
// scalar code.  Here each lane is an ordinary serial decoder for its own frame; the 8-bit probability tables sit in
// LDS (one copy per workgroup, rom_silk_u8_blob), packet bytes arrive through RcLane's word windows.  The lane writes
// a SilkRec (indices of both channels, pulses, stereo predictor, flags) to HBM; the frame-per-wave kernel picks it
// up for everything that is arithmetic on decoded values (parameter dequantisation, NLSF -> LPC, LTP / LPC synthesis,
// stereo un-mixing, resampling: og_silk.hpp).  For hybrid frames the lane also leaves the live coder state in the
// frame's SilkHandoff for the CELT parse kernel.
//
// Reference behaviour: silk_Decode's entropy half (src/silk.cpp:1481-1700), silk_decode_indices (:708),
// silk_decode_pulses (:898) with silk_shell_decoder (:1162) and silk_decode_signs (:1436), silk_stereo_decode_pred (:592).
#pragma once
#include "og_celt_split.hpp"

namespace og {

constexpr int SILK_REC_LPC = 16, SILK_REC_FRAME = 320;

struct SilkRecCh {
    // the first 68 words mirror SilkCtrl (og_silk.hpp) so the wave kernel takes them over with one copy:
    // dequantised parameters (silk_params_lane), then the indices (silk_parse_indices)
    i32 pitchL[4], Gains_Q16[4];
    i16 PredCoef_Q12[2][SILK_REC_LPC];
    i16 LTPCoef_Q14[20];
    i32 LTP_scale_Q14;
    i32 signalType, quantOffsetType, NLSFInterpCoef_Q2, Seed, lagIndex, contourIndex, PERIndex, LTP_scaleIndex;
    i32 GainsIndices[4], LTPIndex[4], NLSFIndices[SILK_REC_LPC + 1];
    // state after the frame, written back by the wave kernel
    i32 ec_prevSignalType, ec_prevLagIndex, LastGainIndex;
    i32 pad;
    i16 nlsf[SILK_REC_LPC]; // stabilised NLSFs of this frame (the next frame's prevNLSF_Q15)
    i16 pulses[SILK_REC_FRAME + 16];
};
constexpr int SILK_REC_CTRL_WORDS = 4 + 4 + SILK_REC_LPC + 10 + 1 + 8 + 4 + 4 + SILK_REC_LPC + 1;
struct alignas(128) SilkRec { // (a record starts on a line boundary of the memory system: the synthesis wave fetches it whole)
    i32 ret; // 0, or the negative code the frame ends with (then nothing else is valid)
    i32 decode_only_middle;
    i32 MS_pred_Q13[2];
    // The stream's prev_mode when the step began.  The synthesis kernel takes it from here, not from the stream state: on
    // a hybrid frame the CELT reconstruction kernel -- which may run concurrently -- writes the new value there.
    i32 prev_mode;
    // What the parameter half (k_silk_params) needs of the stream's past besides its own values, left here by the entropy half --
    // whose run for the stream's NEXT frame may overwrite them in the shadow before the parameter half of this one has run:
    // bit 0 the SILK decoder is re-initialised at this frame (the frame before was CELT-only), bit 1 channel 1 is (the packet adds
    // it), bit 2 the frame before was mid-only; and the channels' internal rate before this frame (0 after a re-initialisation)
    i32 par_flags, fs_past[2];
    SilkRecCh ch[2];
};
static_assert(sizeof(SilkRec) % 16 == 0, "record alignment");

// What the ENTROPY half of a SILK / hybrid frame needs of the frames before it, and nothing else: a copy the parse kernel keeps
// per stream for pipelined SILK-only steps (opusgpu_set_pipeline, og_api.hip), where the parse of step k + 1 runs next to the
// synthesis of step k and so cannot wait for that kernel to write these values into the stream's state.  The parse kernel
// computes what the synthesis WILL leave there (the same values: silk_shadow_next) and hands it to its own next run.
// `epoch`: the copy counts only if it carries the context's current epoch -- every step that is not such a pipelined step, every
// reset and mode change advances the epoch on the host (whatever they do to the state, the copy is then stale by definition),
// and the parse falls back to the state itself, which nothing in flight is writing then.
// Two writers since round 5: k_silk_parse keeps the ENTROPY side (epoch, prev_mode, nChannelsInternal, prev_decode_only_middle,
// the channels' indices history and rate), k_silk_params the PARAMETER side (LastGainIndex, first_frame_after_reset, the
// stabilised NLSFs; `par_epoch` says whether that side is current) -- so that the parse of step k + 1 waits for the parse of
// step k only, not for its parameter half.
struct SilkShadow {
    u32 epoch;
    i32 prev_mode, nChannelsInternal, prev_decode_only_middle;
    struct Ch {
        i32 ec_prevSignalType, ec_prevLagIndex, fs_kHz, LastGainIndex, first_frame_after_reset;
        i16 prevNLSF_Q15[SILK_REC_LPC];
    } ch[2];
    u32 par_epoch;
    i32 pad[1];
};
static_assert(sizeof(SilkShadow) == 128, "one shadow per 128 bytes");

// the entropy half's view of the frames before: the shadow when it is current, the stream's state otherwise
struct SilkPast {
    const StreamState *st;
    const SilkShadow *sh; // null: the state
    OG_MEMBER SilkPast(const StreamState *st_, const SilkShadow *shadow, u32 epoch) : st(st_), sh(shadow && shadow->epoch == epoch ? shadow : nullptr) {}
    OG_MEMBER i32 prev_mode() const { return sh ? sh->prev_mode : st->prev_mode; }
    OG_MEMBER i32 nChannelsInternal() const { return sh ? sh->nChannelsInternal : st->silk.nChannelsInternal; }
    OG_MEMBER i32 prev_dom() const { return sh ? sh->prev_decode_only_middle : st->silk.prev_decode_only_middle; }
    OG_MEMBER i32 ecType(int n) const { return sh ? sh->ch[n].ec_prevSignalType : st->silk.ch[n].ec_prevSignalType; }
    OG_MEMBER i32 ecLag(int n) const { return sh ? sh->ch[n].ec_prevLagIndex : st->silk.ch[n].ec_prevLagIndex; }
    OG_MEMBER i32 fs_kHz(int n) const { return sh ? sh->ch[n].fs_kHz : st->silk.ch[n].fs_kHz; }
};
// ... and the parameter half's: the shadow's parameter side when THAT is current
struct SilkParPast {
    const StreamState *st;
    const SilkShadow *sh;
    OG_MEMBER SilkParPast(const StreamState *st_, const SilkShadow *shadow, u32 epoch) : st(st_), sh(shadow && shadow->par_epoch == epoch ? shadow : nullptr) {}
    OG_MEMBER i32 lastGain(int n) const { return sh ? sh->ch[n].LastGainIndex : st->silk.ch[n].LastGainIndex; }
    OG_MEMBER i32 ffar(int n) const { return sh ? sh->ch[n].first_frame_after_reset : st->silk.ch[n].first_frame_after_reset; }
    OG_MEMBER const i16 *prevNLSF(int n) const { return sh ? sh->ch[n].prevNLSF_Q15 : st->silk.ch[n].prevNLSF_Q15; }
};

OG_LDS u8 g_silk_tab[SILK_BLOB_SIZE]; // LDS copy of rom_silk_u8_blob
OG_DEV void silk_tables_load() {      // cooperative, whole workgroup; ends with a barrier
    const u32 *src = reinterpret_cast<const u32 *>(rom_silk_u8_blob);
    u32 *dst = reinterpret_cast<u32 *>(g_silk_tab);
    OG_FOR_LANES(i, SILK_BLOB_SIZE / 4) dst[i] = src[i];
    OG_FULL_SYNC();
}

// inverse-CDF symbol from the LDS table blob (ec_dec_icdf celt.cpp:2727, ftb = 8 throughout SILK).  `n`: entries of
// the table including its terminating 0.  The reference scans linearly; the table is monotone, so the same symbol is
// found by bisection -- log2(n) dependent LDS reads, and (almost) the same trip count in every lane of the wave.
// The two products the update needs -- r * icdf[symbol] and r * icdf[symbol - 1] -- are the last ones the search compared on either
// side, so they are kept instead of read again (two dependent LDS round trips less per symbol than in round 4).
OG_DEV int rc_icdf_tab(RcLane &rc, int off, int n) {
    const u32 d = rc.val, r = rc.rng >> 8;
    int lo = 0, hi = n - 1; // smallest index with val >= r * icdf[index]; the last entry (0) always qualifies
    u32 s = 0, t = rc.rng;  // r * icdf[hi] (the last entry is 0), r * icdf[lo - 1] (or the whole range at lo == 0)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const u32 p = r * (u32)g_silk_tab[off + mid];
        if (d >= p) {
            hi = mid;
            s = p;
        } else {
            lo = mid + 1;
            t = p;
        }
    }
    rc.val = d - s;
    rc.rng = t - s;
    rc_renorm(rc);
    return lo;
}

// silk_decode_indices silk.cpp:708 (20 ms: nb_subfr == 4).  Indices go straight to the record.
// STORE = false: the frame is read past, not decoded -- an LBRR frame in regular decoding (silk.cpp:1590-1616) -- the same symbols
// leave the coder, nothing is kept but what later symbols depend on; `sig` returns signalType | quantOffsetType << 2.
template <bool STORE = true>
OG_DEV void silk_parse_indices(RcLane &rc, SilkRecCh *o, int fs_kHz, int vad, int decode_LBRR, int condCoding, i32 &ec_prevSignalType,
                               i32 &ec_prevLagIndex, int *sig = nullptr) {
#define OG_KEEP(field, value)                    \
    do {                                         \
        const int keep_ = (value);               \
        if constexpr (STORE) o->field = keep_;   \
    } while (0)
    const int wb = fs_kHz == 16, order = wb ? 16 : 10;
    const int CB1_iCDF = wb ? SILK_BLOB_wb_cb1_icdf : SILK_BLOB_nb_cb1_icdf, ec_sel = wb ? SILK_BLOB_wb_cb2_select : SILK_BLOB_nb_cb2_select;
    const int ec_iCDF = wb ? SILK_BLOB_wb_cb2_icdf : SILK_BLOB_nb_cb2_icdf;
    int Ix;
    if (decode_LBRR || vad)
        Ix = rc_icdf_tab(rc, SILK_BLOB_type_vad_icdf, 4) + 2;
    else
        Ix = rc_icdf_tab(rc, SILK_BLOB_type_novad_icdf, 2);
    const int signalType = Ix >> 1;
    OG_KEEP(signalType, signalType);
    OG_KEEP(quantOffsetType, Ix & 1);
    if (sig) *sig = signalType | (Ix & 1) << 2;
    if (condCoding == 2)
        OG_KEEP(GainsIndices[0], rc_icdf_tab(rc, SILK_BLOB_delta_gain_icdf, 41));
    else {
        int g = rc_icdf_tab(rc, SILK_BLOB_gain_icdf + 8 * signalType, 8) << 3;
        g += rc_icdf_tab(rc, SILK_BLOB_uniform8_icdf, 8);
        OG_KEEP(GainsIndices[0], g);
    }
    for (int i = 1; i < 4; i++) OG_KEEP(GainsIndices[i], rc_icdf_tab(rc, SILK_BLOB_delta_gain_icdf, 41));
    const int cb1 = rc_icdf_tab(rc, CB1_iCDF + (signalType >> 1) * 32, 32);
    OG_KEEP(NLSFIndices[0], cb1);
    for (int i = 0; i < order; i++) { // silk_NLSF_unpack silk.cpp:2762: the entropy table of coefficient i
        const int entry = g_silk_tab[ec_sel + cb1 * order / 2 + (i >> 1)];
        const int ec_ix = ((entry >> (1 + 4 * (i & 1))) & 7) * 9;
        Ix = rc_icdf_tab(rc, ec_iCDF + ec_ix, 9);
        if (Ix == 0)
            Ix -= rc_icdf_tab(rc, SILK_BLOB_nlsf_ext_icdf, 7);
        else if (Ix == 8)
            Ix += rc_icdf_tab(rc, SILK_BLOB_nlsf_ext_icdf, 7);
        OG_KEEP(NLSFIndices[i + 1], Ix - 4);
    }
    OG_KEEP(NLSFInterpCoef_Q2, rc_icdf_tab(rc, SILK_BLOB_nlsf_interp_icdf, 5));
    if (signalType == 2) {
        int decode_abs = 1, lagIndex = 0;
        const int lowbits = fs_kHz == 16 ? SILK_BLOB_uniform8_icdf : (fs_kHz == 12 ? SILK_BLOB_uniform6_icdf : SILK_BLOB_uniform4_icdf);
        const int lowbits_n = fs_kHz >> 1;
        const int contour = fs_kHz == 8 ? SILK_BLOB_pitch_contour_nb_icdf : SILK_BLOB_pitch_contour_icdf;
        const int contour_n = fs_kHz == 8 ? 11 : 34;
        if (condCoding == 2 && ec_prevSignalType == 2) {
            int delta = rc_icdf_tab(rc, SILK_BLOB_pitch_delta_icdf, 21);
            if (delta > 0) {
                delta -= 9;
                lagIndex = tr16(ec_prevLagIndex + delta);
                decode_abs = 0;
            }
        }
        if (decode_abs) {
            lagIndex = tr16(rc_icdf_tab(rc, SILK_BLOB_pitch_lag_icdf, 32) * (fs_kHz >> 1));
            lagIndex = tr16(lagIndex + rc_icdf_tab(rc, lowbits, lowbits_n));
        }
        OG_KEEP(lagIndex, lagIndex);
        ec_prevLagIndex = lagIndex;
        OG_KEEP(contourIndex, rc_icdf_tab(rc, contour, contour_n));
        const int per = rc_icdf_tab(rc, SILK_BLOB_ltp_per_icdf, 3);
        OG_KEEP(PERIndex, per);
        const int t = per == 0 ? SILK_BLOB_ltp_gain_icdf0 : (per == 1 ? SILK_BLOB_ltp_gain_icdf1 : SILK_BLOB_ltp_gain_icdf2);
        for (int j = 0; j < 4; j++) OG_KEEP(LTPIndex[j], rc_icdf_tab(rc, t, 8 << per));
        OG_KEEP(LTP_scaleIndex, condCoding == 0 ? rc_icdf_tab(rc, SILK_BLOB_ltpscale_icdf, 3) : 0);
    }
    ec_prevSignalType = signalType;
    OG_KEEP(Seed, rc_icdf_tab(rc, SILK_BLOB_uniform4_icdf, 4));
#undef OG_KEEP
}

OG_DEV void shell_split_tab(RcLane &rc, int &c1, int &c2, int p, int table) {
    if (p > 0) {
        c1 = rc_icdf_tab(rc, table + g_silk_tab[SILK_BLOB_shell_offsets + p], p + 1);
        c2 = p - c1;
    } else {
        c1 = 0;
        c2 = 0;
    }
}

// Width of the [element][lane] arrays of the SILK parse kernels: a wave's 64 lanes.  k_silk_parse64 fills them -- 64 frames per
// wave, for pipelined steps and large batches, where the kernel's ISSUE SLOTS are what the step pays (a wave's instruction stream
// is nearly the same for 64 frames as for 32) -- k_silk_parse uses the first 32 columns: 32 frames per wave for small in-order
// steps, where the latency of a wave's serial chain is what the step pays and more, shorter-lived waves hide it better.  (Round 4
// measured 64 frames per wave as a wash; with the parameter half gone -- 77 registers, 7.8 KB of LDS -- it wins: DESIGN.md 6e.)
#ifndef OG_SP_LANES
#define OG_SP_LANES (OG_NLANES >= 64 ? 64 : OG_NLANES)
#endif
// Lane-private scratch, [element][lane]: a lane's walk along its column and the wave's access to a row fall on different banks.
// k_silk_parse keeps only the pulse decoder's block bookkeeping here (2.5 KB per 32 frames; with the table blob 5.4 KB per workgroup):
//   blk      sum_pulses | nLshifts << 5 per 16-sample block;
//   nzmask   silk_skip_pulses: which of a block's 16 coefficients are not zero.
struct SilkBlkLds {
    u16 blk[SILK_REC_FRAME / 16][OG_SP_LANES];
    u16 nzmask[SILK_REC_FRAME / 16][OG_SP_LANES];
};
OG_LDS SilkBlkLds g_silk_blkl;
#define g_silk_blk g_silk_blkl.blk
// The parameter dequantisation (silk_decode_parameters: NLSF decode and stabilisation, NLSF -> LPC, gains, pitch, LTP) is a kernel
// of its own since round 5, k_silk_params: ONE (FRAME, CHANNEL) PER LANE -- 32 frames x 2 channels per wave, every lane busy on a
// stereo frame, where the parse kernel's lane did its two channels one after the other with the wave's upper half idle.  Its
// scratch, arrays that are never live together sharing storage:
//   res_Q10  belongs to silk_nlsf_decode and is dead when the interpolated NLSFs (nlsf0) are made;
//   pred_Q8  belongs to silk_nlsf_decode too and is dead when silk_nlsf2a starts; cosLSF is dead once P and Q are formed,
//            P and Q once a32 is, and Atmp is only used after that (silk_inverse_pred_gain, at the end of silk_nlsf2a).
#define OG_PAR_LANES (OG_NLANES >= 64 ? 64 : OG_NLANES)
struct SilkParLds {
    i16 nlsf[SILK_REC_LPC][OG_PAR_LANES];
    union {
        i16 nlsf0[SILK_REC_LPC][OG_PAR_LANES];
        i16 res_Q10[SILK_REC_LPC][OG_PAR_LANES];
    };
    union {
        i32 cosLSF[SILK_REC_LPC][OG_PAR_LANES];
        i32 a32[SILK_REC_LPC][OG_PAR_LANES];
    };
    union {
        struct {
            i32 P[SILK_REC_LPC / 2 + 1][OG_PAR_LANES], Q[SILK_REC_LPC / 2 + 1][OG_PAR_LANES];
        };
        i32 pred_Q8[SILK_REC_LPC][OG_PAR_LANES];
        i32 Atmp[SILK_REC_LPC][OG_PAR_LANES];
    };
};
OG_LDS SilkParLds g_silk_par;

// silk_decode_pulses silk.cpp:898.  A channel's pulses go to the record (HBM) block by block: 16 coefficients = 32 bytes.
// Between the passes of the bitstream (all shell trees, then all LSBs, then all signs) a block waits in its own 32 bytes of the
// record -- as 16 magnitudes of one byte each (they are at most 16), written with ONE 16-byte store and read back with one load,
// where round 3 stored and re-read sixteen 2-byte values one by one (each a memory instruction whose 32 lanes touch 32 records);
// only a block with LSBs (rare) is widened to 16 bits before the sign pass.  The sign pass reads one symbol per NON-ZERO
// coefficient: it walks the bits of the block's non-zero mask -- the wave goes round as often as its busiest lane has non-zero
// coefficients, not sixteen times -- and then builds the signed 16-bit values two at a time (packed 16-bit arithmetic).
#ifndef OG_HOST_EMUL
typedef u32 og_u32x4 __attribute__((ext_vector_type(4)));
OG_DEV u32 og_pk_sub_i16(u32 a, u32 b) {
    u32 r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif
OG_DEV void silk_parse_pulses(RcLane &rc, i16 *pulses, int signalType, int quantOffsetType, int frame_length) {
    int iter = frame_length >> 4;
    if (iter * 16 < frame_length) iter++;
    const int RateLevelIndex = rc_icdf_tab(rc, SILK_BLOB_rate_levels_icdf + 9 * (signalType >> 1), 9);
    const int cdf = SILK_BLOB_pulses_per_block_icdf + 18 * RateLevelIndex;
    OG_MARK(46);
    u32 lsb_blocks = 0; // bit i: block i has LSBs (rare: the lane's own list -- the wave does not go through every block for them)
    for (int i = 0; i < iter; i++) {
        int nl = 0, sp = rc_icdf_tab(rc, cdf, 18);
        while (sp == 17) {
            nl++;
            sp = rc_icdf_tab(rc, SILK_BLOB_pulses_per_block_icdf + 18 * 9 + (nl == 10), 18 - (nl == 10));
        }
        g_silk_blk[i][OG_LANE] = (u16)(sp | nl << 5);
        lsb_blocks |= (u32)(nl > 0) << i;
    }
    OG_MARK(47);
    for (int i = 0; i < iter; i++) {
        i16 *p0 = &pulses[i * 16];
        const int sp = g_silk_blk[i][OG_LANE] & 31;
#ifdef OG_HOST_EMUL
        u8 *pb = reinterpret_cast<u8 *>(p0);
        for (int j = 0; j < 32; j++) pb[j] = 0;
#else
        u32 w[4] = {0, 0, 0, 0};
#endif
        if (sp > 0) {
            // binary shell tree: 16 -> 8 -> 4 -> 2 -> 1, depth-first in the reference's order
            int p3[2], p2[4], p1[8], a, b;
            shell_split_tab(rc, p3[0], p3[1], sp, SILK_BLOB_shell3);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                shell_split_tab(rc, p2[2 * h], p2[2 * h + 1], p3[h], SILK_BLOB_shell2);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int qi = 2 * h + q;
                    shell_split_tab(rc, p1[2 * qi], p1[2 * qi + 1], p2[qi], SILK_BLOB_shell1);
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int ei = 2 * qi + e;
                        shell_split_tab(rc, a, b, p1[ei], SILK_BLOB_shell0);
#ifdef OG_HOST_EMUL
                        pb[2 * ei] = (u8)a;
                        pb[2 * ei + 1] = (u8)b;
#else
                        w[ei >> 1] |= ((u32)a | (u32)b << 8) << (16 * (ei & 1));
#endif
                    }
                }
            }
        }
#ifndef OG_HOST_EMUL
        // (all 32 bytes: a block without pulses and without LSBs is final as it is -- zeros in either form)
        og_u32x4 *dst = reinterpret_cast<og_u32x4 *>(p0);
        const og_u32x4 lo = {w[0], w[1], w[2], w[3]}, zero = {0, 0, 0, 0};
        dst[0] = lo;
        if (sp == 0) dst[1] = zero;
#endif
    }
    OG_MARK(48);
    while (lsb_blocks) { // widen the block, then its LSBs
        const int i = __builtin_ctz(lsb_blocks);
        lsb_blocks &= lsb_blocks - 1;
        const int nLS = g_silk_blk[i][OG_LANE] >> 5;
        i16 *p = &pulses[i * 16];
        const u8 *pb = reinterpret_cast<const u8 *>(p);
        u8 mag[16];
        for (int j = 0; j < 16; j++) mag[j] = pb[j];
        for (int j = 0; j < 16; j++) {
            i32 abs_q = mag[j];
            for (int b = 0; b < nLS; b++) {
                abs_q = shl32(abs_q, 1);
                abs_q += rc_icdf_tab(rc, SILK_BLOB_lsb_icdf, 2);
            }
            p[j] = (i16)abs_q;
        }
    }
    OG_MARK(49);
    const int icdf_ptr = SILK_BLOB_sign_icdf + 7 * (quantOffsetType + (signalType << 1));
    const int length = (frame_length + 8) >> 4;
    for (int i = 0; i < length; i++) {
        const int blk = g_silk_blk[i][OG_LANE];
        // the reference tests sum_pulses[i] > 0 after OR-ing nLshifts << 5 into it (silk.cpp:966)
        if (blk > 0) {
            const u32 ic0 = g_silk_tab[icdf_ptr + OG_MIN(blk & 0x1F, 6)];
            i16 *q = &pulses[i * 16];
            auto sign_symbol = [&]() -> int { // two-symbol iCDF {ic0, 0}, ftb 8
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
                return ret;
            };
            if ((blk >> 5) > 0) { // a widened block: sixteen 16-bit magnitudes
                i32 cur = q[0];
                for (int j = 0; j < 16; j++) {
                    const i32 nxt = j < 15 ? q[j + 1] : 0; // requested one step ahead of its use
                    if (cur > 0) q[j] = (i16)(cur * ((sign_symbol() << 1) - 1));
                    cur = nxt;
                }
                continue;
            }
#ifdef OG_HOST_EMUL
            u8 mag[16];
            for (int j = 0; j < 16; j++) mag[j] = reinterpret_cast<const u8 *>(q)[j];
            for (int j = 0; j < 16; j++) q[j] = mag[j] > 0 ? (i16)((i32)mag[j] * ((sign_symbol() << 1) - 1)) : (i16)0;
#else
            const og_u32x4 w = *reinterpret_cast<const og_u32x4 *>(q);
            // bit j of nz: magnitude j is not zero (a byte is at most 16: adding 0x7f carries into its top bit exactly then)
            u32 nz = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const u32 t = (w[d] + 0x7f7f7f7fu) & 0x80808080u;
                nz |= (((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 15u) << (4 * d);
            }
            u32 neg = 0; // bit j: coefficient j is negative
            while (nz) {
                const int j = __builtin_ctz(nz);
                nz &= nz - 1;
                neg |= (u32)(sign_symbol() ^ 1) << j;
            }
            og_u32x4 o[2];
#pragma unroll
            for (int pr = 0; pr < 8; pr++) { // coefficients 2 pr, 2 pr + 1 as two 16-bit lanes: (m ^ s) - s with s = 0 / -1
                const u32 m2 = __builtin_amdgcn_perm(0u, w[pr >> 1], (pr & 1) ? 0x0c030c02u : 0x0c010c00u);
                const u32 b2 = (neg >> (2 * pr)) & 3u, s2 = og_pk_sub_i16(0u, (b2 | b2 << 15) & 0x00010001u);
                o[pr >> 2][pr & 3] = og_pk_sub_i16(m2 ^ s2, s2);
            }
            og_u32x4 *dst = reinterpret_cast<og_u32x4 *>(q);
            dst[0] = o[0];
            dst[1] = o[1];
#endif
        }
    }
}

// The same symbols without the pulses: an LBRR frame that regular decoding only reads past (silk.cpp:1590-1616).  What the later
// symbols depend on is kept and nothing else -- per block the pulse count and LSB depth (as above) and WHICH coefficients are not
// zero (16 bits, in LDS): the LSB pass can only turn zeros into non-zeros, and the sign pass reads one symbol per non-zero
// coefficient, all from the block's one table -- as many symbols as the mask has bits, not sixteen masked attempts.  Nothing goes to
// HBM and nothing comes back.
OG_DEV void silk_skip_pulses(RcLane &rc, int signalType, int quantOffsetType, int frame_length) {
    int iter = frame_length >> 4;
    if (iter * 16 < frame_length) iter++;
    const int RateLevelIndex = rc_icdf_tab(rc, SILK_BLOB_rate_levels_icdf + 9 * (signalType >> 1), 9);
    const int cdf = SILK_BLOB_pulses_per_block_icdf + 18 * RateLevelIndex;
    OG_MARK(46);
    u32 lsb_blocks = 0;
    for (int i = 0; i < iter; i++) {
        int nl = 0, sp = rc_icdf_tab(rc, cdf, 18);
        while (sp == 17) {
            nl++;
            sp = rc_icdf_tab(rc, SILK_BLOB_pulses_per_block_icdf + 18 * 9 + (nl == 10), 18 - (nl == 10));
        }
        g_silk_blk[i][OG_LANE] = (u16)(sp | nl << 5);
        lsb_blocks |= (u32)(nl > 0) << i;
    }
    OG_MARK(47);
    for (int i = 0; i < iter; i++) {
        const int sp = g_silk_blk[i][OG_LANE] & 31;
        u32 nz = 0;
        if (sp > 0) {
            int p3[2], p2[4], p1[8], a, b;
            shell_split_tab(rc, p3[0], p3[1], sp, SILK_BLOB_shell3);
            for (int h = 0; h < 2; h++) {
                shell_split_tab(rc, p2[2 * h], p2[2 * h + 1], p3[h], SILK_BLOB_shell2);
                for (int q = 0; q < 2; q++) {
                    const int qi = 2 * h + q;
                    shell_split_tab(rc, p1[2 * qi], p1[2 * qi + 1], p2[qi], SILK_BLOB_shell1);
                    for (int e = 0; e < 2; e++) {
                        const int ei = 2 * qi + e;
                        shell_split_tab(rc, a, b, p1[ei], SILK_BLOB_shell0);
                        nz |= (u32)(a > 0) << (2 * ei) | (u32)(b > 0) << (2 * ei + 1);
                    }
                }
            }
        }
        g_silk_blkl.nzmask[i][OG_LANE] = (u16)nz;
    }
    OG_MARK(48);
    while (lsb_blocks) {
        const int i = __builtin_ctz(lsb_blocks);
        lsb_blocks &= lsb_blocks - 1;
        const int nLS = g_silk_blk[i][OG_LANE] >> 5;
        u32 nz = g_silk_blkl.nzmask[i][OG_LANE];
        for (int j = 0; j < 16; j++)
            for (int b = 0; b < nLS; b++) nz |= (u32)rc_icdf_tab(rc, SILK_BLOB_lsb_icdf, 2) << j;
        g_silk_blkl.nzmask[i][OG_LANE] = (u16)nz;
    }
    OG_MARK(49);
    const int icdf_ptr = SILK_BLOB_sign_icdf + 7 * (quantOffsetType + (signalType << 1));
    const int length = (frame_length + 8) >> 4;
    for (int i = 0; i < length; i++) {
        const int blk = g_silk_blk[i][OG_LANE];
        if (blk > 0) {
            const u32 ic0 = g_silk_tab[icdf_ptr + OG_MIN(blk & 0x1F, 6)];
            for (int cnt = __builtin_popcount((u32)g_silk_blkl.nzmask[i][OG_LANE]); cnt > 0; cnt--) {
                // two-symbol iCDF {ic0, 0}, ftb 8: the sign itself is not needed
                u32 s = rc.rng, d = rc.val, r = s >> 8, t = s;
                s = r * ic0;
                if (d < s) {
                    t = s;
                    s = 0;
                }
                rc.val = d - s;
                rc.rng = t - s;
                rc_renorm(rc);
            }
        }
    }
}

OG_DEV void silk_parse_stereo_pred(RcLane &rc, i32 pred_Q13[2]) { // silk_stereo_decode_pred silk.cpp:592
    int n = rc_icdf_tab(rc, SILK_BLOB_stereo_joint_icdf, 25);
    const int ix02 = n / 5, ix12 = n - 5 * ix02;
    int ix00 = rc_icdf_tab(rc, SILK_BLOB_uniform3_icdf, 3);
    const int ix01 = rc_icdf_tab(rc, SILK_BLOB_uniform5_icdf, 5);
    int ix10 = rc_icdf_tab(rc, SILK_BLOB_uniform3_icdf, 3);
    const int ix11 = rc_icdf_tab(rc, SILK_BLOB_uniform5_icdf, 5);
    ix00 += 3 * ix02;
    ix10 += 3 * ix12;
    {
        const i32 low_Q13 = rom_silk_stereo_pred_q13[ix00];
        const i32 step_Q13 = smulwb(rom_silk_stereo_pred_q13[ix00 + 1] - low_Q13, 6554);
        pred_Q13[0] = smlabb(low_Q13, step_Q13, 2 * ix01 + 1);
    }
    {
        const i32 low_Q13 = rom_silk_stereo_pred_q13[ix10];
        const i32 step_Q13 = smulwb(rom_silk_stereo_pred_q13[ix10 + 1] - low_Q13, 6554);
        pred_Q13[1] = smlabb(low_Q13, step_Q13, 2 * ix11 + 1);
    }
    pred_Q13[0] -= pred_Q13[1];
}

// The entropy half of one SILK-only or hybrid frame, lane-private (decode_frame_wave's head + silk_Decode's).
// Reads the stream's state, writes only the record and the hand-off.
// `shadow` (pipelined SILK / hybrid steps): where the lane leaves the entropy side of the stream's NEXT frame's past -- what the
// synthesis kernel will have written to the state by the time it is through with this frame (silk_decode_packet: silk_init_state on
// a switch from CELT, silk_chan_init for a channel the packet adds, silk_set_fs, the indices' history; decode_frame_wave: prev_mode
// -- `mode_after`) -- over the one it has just read; nothing is written for a frame that ends in an error (the past stays).
OG_DEV void silk_parse_lane(const SilkPast &past, const u8 *payload, int len, int mode, int bandwidth, int channels, SilkRec *rec,
                            SilkHandoff *handoff, SilkShadow *shadow = nullptr, u32 epoch = 0, int mode_after = -1) {
    handoff->valid = 0;
    rec->prev_mode = past.prev_mode();
    if (len < 0 || len > 1275) {
        rec->ret = BAD_ARG;
        return;
    }
    int internal_hz = 16000;
    if (mode == MODE_SILK) internal_hz = bandwidth == BW_NB ? 8000 : (bandwidth == BW_MB ? 12000 : 16000);
    const int fs_kHz = (internal_hz >> 10) + 1;
    if (fs_kHz != 8 && fs_kHz != 12 && fs_kHz != 16) {
        rec->ret = INTERNAL_ERROR;
        return;
    }
    const int frame_length = 20 * fs_kHz;
    RcLane rc;
    OG_MARK(50);
    rc_lane_attach(rc, payload, (u32)len);
    rc_init(rc, (u32)len);
    // entropy-side state, as the wave kernel will see it after its own (re-)initialisations:
    // silk_init_state on a CELT -> SILK/hybrid switch, channel 1 init when the packet adds a channel
    const int fresh_all = past.prev_mode() == MODE_CELT, fresh_ch1 = channels > past.nChannelsInternal();
    const i32 fs_past0 = fresh_all ? 0 : past.fs_kHz(0), fs_past1 = (fresh_all || fresh_ch1) ? 0 : past.fs_kHz(1);
    rec->par_flags = fresh_all | fresh_ch1 << 1 | ((fresh_all ? 0 : past.prev_dom()) != 0) << 2;
    rec->fs_past[0] = fs_past0;
    rec->fs_past[1] = fs_past1;
    i32 ecType0 = fresh_all ? 0 : past.ecType(0), ecLag0 = fresh_all ? 0 : past.ecLag(0);
    i32 ecType1 = (fresh_all || fresh_ch1) ? 0 : past.ecType(1), ecLag1 = (fresh_all || fresh_ch1) ? 0 : past.ecLag(1);
    int vad0 = rc_bit_logp(rc, 1), lbrr0 = rc_bit_logp(rc, 1), vad1 = 0, lbrr1 = 0;
    if (channels == 2) {
        vad1 = rc_bit_logp(rc, 1);
        lbrr1 = rc_bit_logp(rc, 1);
    }
    i32 MS_pred_Q13[2] = {0, 0};
    int decode_only_middle = 0;
    // regular decoding reads past the LBRR frames (silk.cpp:1590-1616); their content is discarded
    OG_MARK(45);
    if (lbrr0) {
        if (channels == 2) {
            silk_parse_stereo_pred(rc, MS_pred_Q13);
            if (lbrr1 == 0) decode_only_middle = rc_icdf_tab(rc, SILK_BLOB_mid_only_icdf, 2);
        }
        int sig;
        silk_parse_indices<false>(rc, nullptr, fs_kHz, vad0, 1, 0, ecType0, ecLag0, &sig);
        silk_skip_pulses(rc, sig & 3, sig >> 2, frame_length);
        OG_MARK(45);
    }
    if (channels == 2 && lbrr1) {
        int sig;
        silk_parse_indices<false>(rc, nullptr, fs_kHz, vad1, 1, 0, ecType1, ecLag1, &sig);
        silk_skip_pulses(rc, sig & 3, sig >> 2, frame_length);
    }
    OG_MARK(50);
    if (channels == 2) {
        silk_parse_stereo_pred(rc, MS_pred_Q13);
        decode_only_middle = vad1 == 0 ? rc_icdf_tab(rc, SILK_BLOB_mid_only_icdf, 2) : 0;
    }
    const int has_side = !decode_only_middle;
    OG_MARK(51);
    silk_parse_indices(rc, &rec->ch[0], fs_kHz, vad0, 0, 0, ecType0, ecLag0);
    OG_MARK(52);
    silk_parse_pulses(rc, rec->ch[0].pulses, rec->ch[0].signalType, rec->ch[0].quantOffsetType, frame_length);
    if (channels == 2 && has_side) {
        OG_MARK(51);
        silk_parse_indices(rc, &rec->ch[1], fs_kHz, vad1, 0, 0, ecType1, ecLag1);
        OG_MARK(52);
        silk_parse_pulses(rc, rec->ch[1].pulses, rec->ch[1].signalType, rec->ch[1].quantOffsetType, frame_length);
    }
    OG_MARK(54);
    rec->ch[0].ec_prevSignalType = ecType0;
    rec->ch[0].ec_prevLagIndex = ecLag0;
    rec->ch[1].ec_prevSignalType = ecType1;
    rec->ch[1].ec_prevLagIndex = ecLag1;
    rec->decode_only_middle = decode_only_middle;
    rec->MS_pred_Q13[0] = MS_pred_Q13[0];
    rec->MS_pred_Q13[1] = MS_pred_Q13[1];
    rec->ret = 0;
    // opus_decode_frame src/opus_decoder.cpp:218-221: hybrid redundancy flag, read and ignored (Q2)
    if (rc_tell(rc) + 17 + 20 * (mode == MODE_HYBRID) <= 8 * len) {
        if (mode == MODE_HYBRID) (void)rc_bit_logp(rc, 12);
    }
    handoff->storage = rc.storage; handoff->end_offs = rc.end_offs; handoff->end_window = rc.end_window;
    handoff->nend_bits = rc.nend_bits; handoff->nbits_total = rc.nbits_total; handoff->offs = rc.offs;
    handoff->rng = rc.rng; handoff->val = rc.val; handoff->ext = rc.ext; handoff->rem = rc.rem; handoff->error = rc.error;
    handoff->valid = 1;
    if (shadow) { // (every input of the past has been consumed)
        shadow->ch[0].ec_prevSignalType = ecType0;
        shadow->ch[0].ec_prevLagIndex = ecLag0;
        shadow->ch[1].ec_prevSignalType = ecType1;
        shadow->ch[1].ec_prevLagIndex = ecLag1;
        shadow->ch[0].fs_kHz = fs_kHz;                                // silk_set_fs for the packet's channels
        shadow->ch[1].fs_kHz = channels == 2 ? fs_kHz : fs_past1;
        shadow->prev_mode = mode_after >= 0 ? mode_after : mode;
        shadow->nChannelsInternal = channels;
        shadow->prev_decode_only_middle = decode_only_middle; // (0 for a mono packet: silk_decode_packet stores its local, which only stereo packets set)
        shadow->epoch = epoch;
    }
}

} // namespace og
