// TEST-ONLY host emulation of the HIP decode kernels (see og_common.hpp, OG_HOST_EMUL).
// Compiles the kernel headers for the CPU with a one-lane "wave" so the device source can be fuzzed
// against the oracle and run under ASan/UBSan without a GPU.  Never linked into libopusgpu.so.
#define OG_HOST_EMUL 1
#include "og_decode.hpp"
#include "og_output.hpp"

#ifdef OG_STATS
long long og_stats[64];
extern "C" const long long *emu_stats(void) { return og_stats; }
#endif

extern "C" {
int emu_state_size(void) { return (int)sizeof(og::StreamState); }
void emu_stream_init(void *st, int channels) { og::stream_init((og::StreamState *)st, channels); }
void emu_stream_reset(void *st) { og::stream_reset((og::StreamState *)st); }
// mode 0 = the empty frame of a stream that has had no packet yet (descriptor bit 11, og_state.hpp): coded as hybrid, prev_mode
// stays 0.  Returns what to pass as mode_after (-1: the mode itself).
static int emu_no_mode(int &mode) {
    if (mode != 0) return -1;
    mode = og::MODE_HYBRID;
    return 0;
}
// the single-kernel path (every mode)
int emu_decode_frame_single(void *st, const uint8_t *payload, int len, int mode, int bw, int ch, int16_t *pcm) {
    const int after = emu_no_mode(mode);
    return og::decode_frame_wave<true>((og::StreamState *)st, payload, len, mode, bw, ch, pcm, nullptr, nullptr, 0, after);
}
// what the library dispatches: CELT-only frames take the split path (parse per lane, reconstruct per wave, post per
// channel); hybrid frames decode their SILK half on the single-kernel path and hand the CELT half over
static og::ParseRec rec; // the last frame's parse record (emu_last_leaf_geom)
// the (position, n, k, blocks) words of the last CELT / hybrid frame's PVQ leaves, in the order the reconstruction takes them
int emu_last_leaf_geom(unsigned *out, int cap) {
    const int n = rec.n_leaves < cap ? rec.n_leaves : cap;
    for (int i = 0; i < n; i++) out[i] = rec.leaf[i].geom;
    return rec.n_leaves;
}
int emu_last_leaf_idx(unsigned *out, int cap) { // ... and their codeword indices
    const int n = rec.n_leaves < cap ? rec.n_leaves : cap;
    for (int i = 0; i < n; i++) out[i] = rec.leaf[i].idx;
    return rec.n_leaves;
}
unsigned emu_last_rec_flags(void) { return rec.flags; } // ... and the frame's record flags (RF_*: the spread decision among them)
int emu_decode_frame(void *stv, const uint8_t *payload, int len, int mode, int bw, int ch, int16_t *pcm) {
    og::StreamState *st = (og::StreamState *)stv;
    static og::SilkHandoff handoff;
    static og::SilkRec srec;
    const og::SilkHandoff *h = nullptr;
    const int after = emu_no_mode(mode);
    if (mode != og::MODE_CELT) { // SILK entropy half per lane, then the frame-per-wave kernel from the record
        og::silk_tables_load();
        const og::SilkPast past(st, nullptr, 0);
        og::silk_parse_lane(past, payload, len, mode, bw, ch, &srec, &handoff);
        og::silk_params_lane(og::SilkParPast(st, nullptr, 0), mode, bw, ch, &srec);
        int r = og::decode_frame_wave<false>(st, payload, len, mode, bw, ch, pcm, &handoff, &srec);
        if (r == og::CONTINUE_Q4) return og::decode_frame_wave<true>(st, payload, len, mode, bw, ch, pcm, &handoff, &srec, 1);
        if (r != og::CONTINUE_SPLIT) return r;
        h = &handoff;
    }
    og::parse_tables_load();
    og::celt_parse_lane(st, payload, len, ch, &rec, h);
    const int ret = og::celt_recon_wave(st, &rec, mode, ch, og::RECON_ALL, after);
    for (int c = 0; c < st->channels; c++) og::celt_post(st, &rec, ret, c, pcm, h ? h->pcm : nullptr, ch);
    return ret;
}
// A SILK-only / hybrid frame the way pipelined SILK-only steps decode it: the entropy half takes the past from `shadow` (128 bytes,
// og::SilkShadow) when that carries `epoch`, from the state otherwise, and leaves the next frame's past there.
int emu_decode_frame_shadowed(void *stv, void *shadow, unsigned epoch, const uint8_t *payload, int len, int mode, int bw, int ch, int16_t *pcm) {
    og::StreamState *st = (og::StreamState *)stv;
    static og::SilkHandoff handoff;
    static og::SilkRec srec;
    if (mode == og::MODE_CELT) return -1000;
    const int after = emu_no_mode(mode);
    og::silk_tables_load();
    const og::SilkPast past(st, (const og::SilkShadow *)shadow, epoch);
    const og::SilkParPast par_past(st, (const og::SilkShadow *)shadow, epoch); // (chosen before the parse moves the entropy side on)
    og::silk_parse_lane(past, payload, len, mode, bw, ch, &srec, &handoff, (og::SilkShadow *)shadow, epoch, after);
    og::silk_params_lane(par_past, mode, bw, ch, &srec, (og::SilkShadow *)shadow, epoch);
    int r = og::decode_frame_wave<false>(st, payload, len, mode, bw, ch, pcm, &handoff, &srec);
    if (r == og::CONTINUE_Q4) return og::decode_frame_wave<true>(st, payload, len, mode, bw, ch, pcm, &handoff, &srec, 1);
    if (r != og::CONTINUE_SPLIT) return r;
    og::parse_tables_load();
    og::celt_parse_lane(st, payload, len, ch, &rec, &handoff);
    const int ret = og::celt_recon_wave(st, &rec, mode, ch, og::RECON_ALL, after);
    for (int c = 0; c < st->channels; c++) og::celt_post(st, &rec, ret, c, pcm, handoff.pcm, ch);
    return ret;
}
// 0 when `shadow` holds exactly what the state holds of the entropy half's past (it must, after every frame that decoded);
// otherwise the number of the first field that differs
int emu_shadow_vs_state(const void *stv, const void *shadow) {
    const og::StreamState *st = (const og::StreamState *)stv;
    const og::SilkShadow *sh = (const og::SilkShadow *)shadow;
    if (sh->prev_mode != st->prev_mode) return 1;
    if (sh->nChannelsInternal != st->silk.nChannelsInternal) return 2;
    if (sh->prev_decode_only_middle != st->silk.prev_decode_only_middle) return 3;
    for (int n = 0; n < 2; n++) {
        const og::SilkChannel &c = st->silk.ch[n];
        const og::SilkShadow::Ch &o = sh->ch[n];
        if (o.ec_prevSignalType != c.ec_prevSignalType) return 10 + 10 * n;
        if (o.ec_prevLagIndex != c.ec_prevLagIndex) return 11 + 10 * n;
        if (o.fs_kHz != c.fs_kHz) return 12 + 10 * n;
        if (o.LastGainIndex != c.LastGainIndex) return 13 + 10 * n;
        if (o.first_frame_after_reset != c.first_frame_after_reset) return 14 + 10 * n;
        for (int i = 0; i < 16; i++)
            if (o.prevNLSF_Q15[i] != c.prevNLSF_Q15[i]) return 100 + 16 * n + i;
    }
    return 0;
}
// RFC mode (opt-in): one Opus frame at its true duration through the single-kernel code (og_decode.hpp, decode_frame_rfc)
int emu_decode_frame_rfc(void *st, const uint8_t *payload, int len, int mode, int bw, int ch, int16_t *pcm, int frame_size) {
    return og::decode_frame_rfc((og::StreamState *)st, payload, len, mode, bw, ch, pcm, frame_size);
}
// ... its forward error correction data instead (RFC 6716's decode_fec; descriptor flag bit 10)
int emu_decode_frame_rfc_fec(void *st, const uint8_t *payload, int len, int mode, int bw, int ch, int16_t *pcm, int frame_size) {
    return og::decode_frame_rfc((og::StreamState *)st, payload, len, mode, bw, ch, pcm, frame_size, 1);
}
int emu_last_record_words(void) { return 0; }
}

// ---- unit entry: the kernels' PVQ leaf decode (index -> pulses -> scaled -> rotation undone -> collapse mask) ---------
extern "C" unsigned emu_pvq_leaf(int n, int k, unsigned index, int B, int gain, int spread, int16_t *X) {
    for (int i = 0; i < n; i++) og::S.v[og::V_X + i] = 0; // the leaf function relies on a cleared spectrum
    og::pvq_tab_load();
    const unsigned cm = og::pvq_leaf_lane(og::S.v, og::pvq_lds(), n, k, index, og::V_X, B, gain, spread);
    memcpy(X, &og::S.v[og::V_X], sizeof(int16_t) * n);
    return cm;
}

// ---- stage taps -------------------------------------------------------------------------------
static int16_t tap_X[1920], tap_bandE[42];
static int32_t tap_syn_pre[2][1080], tap_syn_post[2][1080];
static og::SilkCtrl tap_silk_ctrl[2];
static int16_t tap_silk_xq[2][og::SILK_MAX_FRAME];
extern "C" void og_emul_tap(int id) {
    if (id == 40) { // SILK: decoder control and core output, before stereo un-mixing rewrites xq
        for (int c = 0; c < 2; c++) {
            tap_silk_ctrl[c] = og::SL().ctrl[c];
            memcpy(tap_silk_xq[c], &og::SL().xq[c][2], sizeof(tap_silk_xq[c]));
        }
        return;
    }
    if (id == 1) { memcpy(tap_X, &og::S.v[og::V_X], sizeof(tap_X)); memcpy(tap_bandE, og::S.bandE, sizeof(tap_bandE)); }
    // synthesis taps come once per channel: id = 2 (IMDCT output) or 3 (comb filter output), + 16 * channel
    if ((id & 15) == 2) memcpy(tap_syn_pre[id >> 4], og::syn_buf(), 1080 * 4);
    if ((id & 15) == 3) memcpy(tap_syn_post[id >> 4], og::syn_buf(), 1080 * 4);
}
extern "C" {
// what as in the oracle's oc_silk_taps_copy: 0 = {coded, signalType, quantOffsetType, -, -, LTP_scale_Q14}, 1 = pitchL + Gains_Q16,
// 2 = PredCoef_Q12, 3 = LTPCoef_Q14, 4 = xq (SILK_MAX_FRAME values)
int emu_tap_silk(int what, int ch, void *dst) {
    const og::SilkCtrl &k = tap_silk_ctrl[ch];
    switch (what) {
        case 0: { int32_t v[6] = {k.coded, k.signalType, k.quantOffsetType, 0, 0, k.LTP_scale_Q14}; memcpy(dst, v, sizeof v); return sizeof v; }
        case 1: memcpy(dst, k.pitchL, 8 * sizeof(int32_t)); return 8 * sizeof(int32_t);
        case 2: memcpy(dst, k.PredCoef_Q12, sizeof k.PredCoef_Q12); return sizeof k.PredCoef_Q12;
        case 3: memcpy(dst, k.LTPCoef_Q14, sizeof k.LTPCoef_Q14); return sizeof k.LTPCoef_Q14;
        case 4: memcpy(dst, tap_silk_xq[ch], sizeof tap_silk_xq[ch]); return sizeof tap_silk_xq[ch];
    }
    return -1;
}
const int16_t *emu_tap_X(void) { return tap_X; }
const int16_t *emu_tap_bandE(void) { return tap_bandE; }
const int32_t *emu_tap_syn_pre(int c) { return tap_syn_pre[c]; }
const int32_t *emu_tap_syn_post(int c) { return tap_syn_post[c]; }
}

// output stage (og_output.hpp): all words of one block, as the kernel computes them
extern "C" int emu_output_block(const int16_t *blk, int valid, int volume, int force_mono, int bits, int channels, uint32_t *out) {
    og::OutputCfg c;
    c.volume = (uint8_t)volume; c.force_mono = (uint8_t)force_mono; c.bits = (uint8_t)bits; c.channels = (uint8_t)channels;
    const int n = og::output_words(c, valid);
    for (int w = 0; w < n; w++) out[w] = og::output_word(blk, w, c);
    return n;
}
