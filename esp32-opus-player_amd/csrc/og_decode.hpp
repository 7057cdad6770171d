// og_decode.hpp -- per-frame mode dispatch of the wave decoder: the device-side equivalent of the
// reference's opus_decode_frame (src/opus_decoder.cpp:154-278).  One call decodes one 20 ms frame of
// one stream: range-decoder init, SILK (SILK-only / hybrid), CELT (CELT-only / hybrid, plus the
// reference's hybrid->SILK transition quirk Q4), saturating mix, state update, PCM write-out.
#pragma once
#include "og_celt.hpp"
#include "og_celt_split.hpp"
#ifndef OG_NO_SILK
#include "og_silk.hpp"
#include "og_plc.hpp"
#endif

namespace og {

// fresh stream == opus_multistream_decoder_init (src/opus_decoder.cpp:742 -> :82): everything zero,
// then the CELT/SILK reset values.  Lane-parallel over the record's words.
OG_DEV void stream_init(StreamState *st, int channels) {
    u32 *w = reinterpret_cast<u32 *>(st);
    const int nw = (int)(sizeof(StreamState) / 4);
    OG_FOR_LANES(i, nw) w[i] = 0;
    OG_SYNC();
    OG_FOR_LANES(i, 2 * NBANDS) st->celt.logE1[i] = st->celt.logE2[i] = (i16)(-28 * 1024);
    if (OG_LANE == 0) st->channels = channels;
#ifndef OG_NO_SILK
    silk_init_state(&st->silk);
#endif
    OG_SYNC();
}

// OPUS_RESET_STATE semantics (src/opus_decoder.cpp:382-390): partial CELT reset (Q5), full SILK init
OG_DEV void stream_reset(StreamState *st) {
    celt_reset_state(&st->celt);
    if (OG_LANE == 0) {
        st->prev_mode = 0;
        st->range_final = 0;
    }
#ifndef OG_NO_SILK
    silk_init_state(&st->silk);
#endif
    OG_SYNC();
}

// Decode one frame.  `payload` points at the frame's bytes in HBM; `pcm` at 960*channels int16 in HBM.
// Returns samples per channel (960) or a negative OPUS_* code; the value is wave-uniform.
// With the split path enabled (`handoff`, `srec` != null) SILK-only and hybrid frames arrive with their entropy half
// already decoded by the lane-per-frame parse kernel: `srec` holds the SILK indices / pulses, `handoff` the live coder
// state.  A hybrid frame's CELT half is then left to the split path: the SILK PCM goes to handoff->pcm and
// CONTINUE_SPLIT is returned (no result, no bookkeeping yet).
//
// WITH_CELT = false is the split path's SILK synthesis kernel: it contains no CELT code at all (and so needs neither
// the CELT working set in LDS nor its registers).  SILK-only frames finish there -- their PCM is the SILK PCM -- except
// the Q4 transition frame (SILK-only right after hybrid), whose 120-sample CELT frame needs the CELT decoder: that one
// is parked (handoff->valid = 2, CONTINUE_Q4) and finished by a second pass of the full kernel with q4_resume = 1.
enum { CONTINUE_SPLIT = 1, CONTINUE_Q4 = 2 };
template <bool WITH_CELT>
OG_DEV int decode_frame_wave(StreamState *st, const u8 *payload, int len, int mode, int bandwidth, int ch, i16 *pcm,
                             SilkHandoff *handoff = nullptr, const SilkRec *srec = nullptr, int q4_resume = 0, int mode_after = -1) {
    const int audiosize = 960;
    if (mode_after < 0) mode_after = mode; // (what prev_mode becomes: desc_mode_after)
    const int CC = st->channels;
    if (len < 0 || len > 1275) return BAD_ARG;
    // (split path: the value the parse kernel saw when the step began; see SilkRec::prev_mode)
    const int prev_mode = srec ? (int)OG_UNI(srec->prev_mode) : st->prev_mode;
    Rc rc;
    if (srec) {
        const int r0 = OG_UNI(srec->ret);
        if (r0 < 0) return r0;
        // the coder state after the SILK half (+ redundancy flag); the packet itself is only needed again by Q4
        rc.storage = (u32)OG_UNI(handoff->storage); rc.end_offs = (u32)OG_UNI(handoff->end_offs);
        rc.end_window = (u32)OG_UNI(handoff->end_window); rc.nend_bits = OG_UNI(handoff->nend_bits);
        rc.nbits_total = OG_UNI(handoff->nbits_total); rc.offs = (u32)OG_UNI(handoff->offs); rc.rng = (u32)OG_UNI(handoff->rng);
        rc.val = (u32)OG_UNI(handoff->val); rc.ext = (u32)OG_UNI(handoff->ext); rc.rem = OG_UNI(handoff->rem);
        rc.error = OG_UNI(handoff->error);
        if (WITH_CELT && mode == MODE_SILK && prev_mode == MODE_HYBRID) {
            OG_SYNC();
            OG_FOR_LANES(i, len) S.pkt[i] = payload[i];
            if (q4_resume) { // the SILK half ran in the synthesis kernel: take its PCM back
                const u32 *src = reinterpret_cast<const u32 *>(handoff->pcm);
                u32 *dst = reinterpret_cast<u32 *>(SL().u.out.pcm);
                OG_FOR_LANES(i, audiosize * ch / 2) dst[i] = src[i];
            }
            OG_SYNC();
        }
    } else if (WITH_CELT) {
        OG_SYNC();
        OG_FOR_LANES(i, len) S.pkt[i] = payload[i];
        OG_SYNC();
        rc_init(rc, (u32)len);
    } else
        return INTERNAL_ERROR; // the synthesis kernel only exists behind the parse kernel
    int celt_ret = 0;

#ifndef OG_NO_SILK
    if (mode != MODE_CELT && !q4_resume) {
        if (prev_mode == MODE_CELT) silk_init_state(&st->silk);
        int internal_hz = 16000;
        if (mode == MODE_SILK) internal_hz = bandwidth == BW_NB ? 8000 : (bandwidth == BW_MB ? 12000 : 16000);
        int ret = silk_decode_20ms<!WITH_CELT>(&st->silk, rc, ch, internal_hz, srec); // fills g_pcm_silk (48 kHz, interleaved)
        if (ret) return INTERNAL_ERROR;
    }
#else
    if (mode != MODE_CELT) return INTERNAL_ERROR;
#endif
    int start_band = 0;
    if (!srec && mode != MODE_CELT && rc_tell(rc) + 17 + 20 * (mode == MODE_HYBRID) <= 8 * len) {
        if (mode == MODE_HYBRID) (void)rc_bit_logp(rc, 12); // redundancy flag read and ignored (Q2)
    }
    if (mode != MODE_CELT) start_band = 17;
    const int disable_inv = CC == 1;
#ifndef OG_NO_SILK
    if (handoff && mode == MODE_HYBRID) {
        OG_SYNC();
        {
            const u32 *src = reinterpret_cast<const u32 *>(SL().u.out.pcm);
            u32 *dst = reinterpret_cast<u32 *>(handoff->pcm);
            OG_FOR_LANES(i, audiosize * ch / 2) dst[i] = src[i];
        }
        OG_SYNC();
        return CONTINUE_SPLIT;
    }
    if (!WITH_CELT) { // SILK-only frame on the split path
        if (mode != MODE_SILK) return INTERNAL_ERROR;
        OG_SYNC();
        if (prev_mode == MODE_HYBRID) { // Q4: park the frame for the full kernel
            const u32 *src = reinterpret_cast<const u32 *>(SL().u.out.pcm);
            u32 *dst = reinterpret_cast<u32 *>(handoff->pcm);
            OG_FOR_LANES(i, audiosize * ch / 2) dst[i] = src[i];
            if (OG_LANE == 0) handoff->valid = 2;
            OG_SYNC();
            return CONTINUE_Q4;
        }
        // PCM = SAT16(0 + pcm_silk) over the first 960 * ch interleaved entries, zero beyond (Q3)
        {
            const u32 *src = reinterpret_cast<const u32 *>(SL().u.out.pcm);
            u32 *dst = reinterpret_cast<u32 *>(pcm);
            OG_FOR_LANES(i, audiosize * CC / 2) dst[i] = i < audiosize * ch / 2 ? src[i] : 0u;
        }
        if (OG_LANE == 0) {
            st->prev_mode = mode;
            st->frames_decoded += 1;
            st->range_final = rc.rng;
        }
        OG_SYNC();
        return audiosize;
    }
#endif

    if constexpr (WITH_CELT) {
    // One CELT call site (the whole CELT decoder is inlined into it): the regular frame, or -- Q4 -- the 2.5 ms
    // frame the reference decodes from the live range decoder on a hybrid -> SILK-only transition.
    int do_celt = 0, celt_n = audiosize, celt_start = start_band;
    if (mode != MODE_SILK) {
        if (mode != prev_mode && prev_mode > 0) {
            celt_reset_state(&st->celt);
            OG_SYNC();
        }
        do_celt = 1;
    } else {
        OG_SYNC();
        OG_FOR_LANES(i, audiosize * 2) S.v[V_X + i] = 0;
        OG_SYNC();
        if (prev_mode == MODE_HYBRID) { // Q4: start band 0, 120*CC samples at the head of the PCM staging area
            do_celt = 1;
            celt_n = 120;
            celt_start = 0;
        }
    }
    if (do_celt) {
        const int r = celt_decode_frame(&st->celt, rc, celt_n, ch, CC, celt_start, disable_inv);
        if (mode != MODE_SILK)
            celt_ret = r;
        else { // the reference ignores the return value here (src/opus_decoder.cpp:267).  Only the 120 decoded
               // samples per channel survive; the rest of the PCM planes (which shared X with CELT's work) is zero.
            OG_SYNC();
            for (int c = 0; c < CC; c++) OG_FOR_LANES(j, 120) S.v[V_IY + 120 * c + j] = S.v[pcm_plane(c, ch, CC) + j];
            OG_SYNC();
            OG_FOR_LANES(i, audiosize * 2) S.v[V_X + i] = 0;
            OG_SYNC();
            for (int c = 0; c < CC; c++) OG_FOR_LANES(j, 120) S.v[pcm_plane(c, ch, CC) + j] = S.v[V_IY + 120 * c + j];
            OG_SYNC();
        }
    }
#ifndef OG_NO_SILK
    if (mode != MODE_CELT) { // SAT16(outbuf + pcm_silk) over audiosize*stream_channels entries (Q3)
        OG_SYNC();
        OG_FOR_LANES(i, audiosize * ch) { // i indexes the interleaved PCM; sample j of channel c lives in plane c
            const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i, at = pcm_plane(c, ch, CC) + j;
            S.v[at] = (i16)sat16((i32)S.v[at] + (i32)SL().u.out.pcm[i]);
        }
        OG_SYNC();
    }
#endif
    if (OG_LANE == 0) {
        st->prev_mode = mode_after;
        st->frames_decoded += 1;
        st->range_final = rc.rng;
    }
    if (celt_ret < 0) return celt_ret;
    // PCM: LDS staging -> HBM, two samples per lane-store, coalesced
    OG_SYNC();
    pcm_store(pcm, audiosize, ch, CC);
    OG_SYNC();
    return audiosize;
    } else
        return INTERNAL_ERROR;
}

// RFC mode (opt-in; the oracle's oc_decoder_set_rfc): one Opus frame at the duration its TOC names -- CELT 2.5 /
// 5 / 10 / 20 ms, SILK 10 / 20 / 40 / 60 ms, hybrid 10 / 20 ms -- CELT's last band by bandwidth, and the two-byte silence frame
// instead of Q4's stale-coder frame when SILK-only follows hybrid.  Everything else as decode_frame_wave has it (Q2, Q3, Q5).
// Runs on the single-kernel path only (wave-uniform entropy decoding): the mode exists for completeness, the lane-per-frame
// parse kernels and the 8 KB reconstruction kernel stay 20 ms.  `pcm`: audiosize * channels int16 in HBM.
// Returns audiosize or a negative code (wave-uniform).
OG_DEV int rfc_end_band(int bandwidth) { // RFC 6716 section 4.3: NB 13, WB 17, SWB 19, FB 21 bands
    return bandwidth == BW_NB ? 13 : (bandwidth == BW_MB || bandwidth == BW_WB) ? 17 : bandwidth == BW_SWB ? 19 : NBANDS;
}
// A frame with nothing to decode (RFC mode): the packet was lost (len == 0) or the frame is a DTX frame of at
// most one byte.  RFC 6716's opus_decode_frame with data == NULL as the oracle restates it (conceal_frame, oracle/oc_packet.c):
// the mode is the previous frame's; more than 20 ms goes in chunks of 20 ms; SILK conceals 10 or 20 ms (a 2.5 / 5 ms request takes
// the head of a 10 ms concealment); CELT -- and hybrid's CELT layer from band 17 -- conceals with celt_decode_lost, its last band
// what the last decoded frame made it.  `ch`: the channel count of the last packet (the descriptor's).
// `hold` (the transition smoothing of decode_frame_rfc): the audio goes to this LDS buffer of audiosize * channels entries instead
// of `pcm`, zero where a SILK-only concealment of a mono packet in a stereo decoder defines nothing (Q3).
OG_DEV int conceal_chunk_rfc(StreamState *st, int ch, i16 *pcm, int audiosize, i16 *hold = nullptr) { // at most 20 ms
    // the last used mode: CELT if the last frame ended with CELT redundancy
    const int CC = st->channels, mode = st->loss.prev_redundancy ? (int)MODE_CELT : st->prev_mode;
    if (mode == 0) { // nothing decoded yet: zeros
        OG_FOR_LANES(i, audiosize * CC) (hold ? hold : pcm)[i] = 0;
        OG_SYNC();
        return audiosize;
    }
    const int nmix = audiosize * (ch < CC ? ch : CC);
#ifndef OG_NO_SILK
    if (mode != MODE_CELT) {
        Rc none;
        rc_init(none, 0u);
        int taken = 0; // the SILK PCM of the frame, kept in SL().u.out.pcm (one internal frame: the first call's output is what counts)
        const int ret = silk_decode_packet<false>(&st->silk, none, ch, 0, audiosize >= 960 ? 20 : 10, nullptr,
                                                  [&](int, int n48) { taken += n48; }, &st->loss, 1);
        if (ret) return INTERNAL_ERROR;
        if (mode == MODE_SILK) { // PCM = SAT16(0 + pcm_silk) over the frame's first nmix linear entries
            OG_SYNC();
            if (hold)
                OG_FOR_LANES(i, audiosize * CC) hold[i] = i < nmix ? SL().u.out.pcm[i] : (i16)0;
            else
                OG_FOR_LANES(i, nmix) pcm[i] = SL().u.out.pcm[i];
            OG_SYNC();
        }
    }
#else
    if (mode != MODE_CELT) return INTERNAL_ERROR;
#endif
    int celt_ret = 0;
    if (mode != MODE_SILK) {
        celt_ret = celt_conceal(&st->celt, &st->loss, audiosize, CC, mode == MODE_CELT ? 0 : 17, st->loss.celt_end_band);
#ifndef OG_NO_SILK
        if (mode == MODE_HYBRID && celt_ret >= 0) {
            OG_SYNC();
            OG_FOR_LANES(i, nmix) {
                const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i, at = pcm_plane(c, CC, CC) + j;
                S.v[at] = (i16)sat16((i32)S.v[at] + (i32)SL().u.out.pcm[i]);
            }
            OG_SYNC();
        }
#endif
        if (celt_ret >= 0) {
            OG_SYNC();
            if (hold)
                OG_FOR_LANES(i, audiosize * CC) {
                    const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i;
                    hold[i] = S.v[pcm_plane(c, CC, CC) + j];
                }
            else
                pcm_store(pcm, audiosize, CC, CC);
            OG_SYNC();
        }
    }
    OG_SYNC();
    if (OG_LANE == 0) {
        st->prev_mode = mode;
        st->loss.prev_redundancy = 0;
        st->frames_decoded += 1;
        st->range_final = 0;
    }
    OG_SYNC();
    return celt_ret < 0 ? celt_ret : audiosize;
}
OG_DEV int conceal_frame_rfc(StreamState *st, int ch, i16 *pcm, int audiosize) {
    const int CC = st->channels;
    for (int done = 0; done < audiosize; done += 960) {
        const int r = conceal_chunk_rfc(st, ch, pcm + done * CC, OG_MIN(audiosize - done, 960));
        if (r < 0) return r;
    }
    return audiosize;
}

// `fec` (descriptor flag bit 10): RFC 6716's decode_fec -- the frame is the first one of the packet AFTER a lost packet: SILK
// decodes its LBRR copies (the lost frame's audio) where the packet carries them and conceals where not, CELT has no FEC and
// conceals (hybrid: its layer from band 17).  The host never sets it for CELT-only frames (plain concealment instead).
// Redundancy (RFC 6716 section 4.5.1; the reference reads the flag and ignores it, Q2): a hybrid frame may flag, and a SILK-only
// frame carries in whatever follows its SILK data, a redundant 5 ms CELT frame for a mode transition.  CELT -> SILK: it is
// decoded first, from the running CELT state; its first 2.5 ms replace the frame's start, the rest fades out into the frame.
// SILK -> CELT: it is decoded last, from a reset CELT state, and its second half fades in over the frame's last 2.5 ms (the CELT
// frame that follows then continues from that state instead of a reset one).  The fades weigh with the squared CELT window.
OG_DEV void rfc_smooth_fade(i16 *pcm, int at, const i16 *red, int red_at, int CC, bool into_pcm) {
    // out = (w * in2 + (Q15ONE - w) * in1) >> 15 over 120 samples per channel; in1 fades out, in2 fades in.
    // into_pcm: in1 = the frame (pcm[at..]), in2 = the redundant audio (red[red_at..]); otherwise the other way round.
    OG_FOR_LANES(i, 120 * CC) {
        const int j = CC == 2 ? (i >> 1) : i;
        const i32 w = mul16_q15(rom_win120[j], rom_win120[j]);
        const i32 p = pcm[at + i], r = red[red_at + i];
        const i32 in1 = into_pcm ? p : r, in2 = into_pcm ? r : p;
        pcm[at + i] = (i16)((mul16(w, in2) + mul16(32767 - w, in1)) >> 15);
    }
}
OG_DEV int decode_frame_rfc(StreamState *st, const u8 *payload, int len, int mode, int bandwidth, int ch, i16 *pcm, int audiosize,
                            int fec = 0) {
    const int CC = st->channels;
    if (len < 0 || len > 1275) return BAD_ARG;
    if (len <= 1) return conceal_frame_rfc(st, ch, pcm, audiosize);
    const int prev_mode = st->prev_mode, prev_redundancy = st->loss.prev_redundancy;
    // the reference's mix loop runs over audiosize * stream_channels LINEAR entries of the interleaved output (Q3); the oracle's
    // RFC mode keeps that and stops at the frame's own end
    const int nmix = audiosize * (ch < CC ? ch : CC);
    const int disable_inv = CC == 1, end_band = rfc_end_band(bandwidth);
    // RFC 6716 section 4.5: a switch between CELT-only and the SILK modes that no redundant frame covers is smoothed with 5 ms of
    // concealment from the OLD mode (as much as the frame is long, for a 2.5 ms frame), cross-faded into the new frame at the
    // end.  The SILK modes' concealment runs before anything of the new frame is decoded, CELT's behind the frame's SILK layer.
    // It waits in the SILK up-sampler's buffers (dead whenever it is needed; a frame with a redundant frame has no transition).
    int transition = 0;
#ifndef OG_NO_SILK
    transition = prev_mode > 0 && ((mode == MODE_CELT && prev_mode != MODE_CELT && !prev_redundancy) || (mode != MODE_CELT && prev_mode == MODE_CELT));
    i16 *const tr_hold = &SL().u.out.up[0][0];
    const int tr_size = OG_MIN(240, audiosize);
    if (transition && mode == MODE_CELT) {
        const int r = conceal_chunk_rfc(st, ch, nullptr, tr_size, tr_hold);
        if (r < 0) return r;
    }
#endif
    OG_SYNC();
    OG_FOR_LANES(i, len) S.pkt[i] = payload[i];
    OG_SYNC();
    Rc rc;
    rc_init(rc, (u32)len);
    int celt_ret = 0, redundancy = 0, celt_to_silk = 0, redundancy_bytes = 0, celt_lost = 0, main_len = len;
    u32 redundant_rng = 0;
#ifndef OG_NO_SILK
    if (mode != MODE_CELT) {
        if (prev_mode == MODE_CELT) silk_init_state(&st->silk, &st->loss);
        silk_wave_tab_load(); // (og_rfc.hip: the entropy decoder's tables into LDS, over rows that nothing uses until the CELT layer)
        int internal_hz = 16000;
        if (mode == MODE_SILK) internal_hz = bandwidth == BW_NB ? 8000 : (bandwidth == BW_MB ? 12000 : 16000);
        int base = 0; // linear position of the internal frame in the packet's SILK PCM
        const int ret = silk_decode_packet<false>(&st->silk, rc, ch, internal_hz, audiosize / 48, nullptr, [&](int, int n48) {
            if (mode == MODE_SILK) { // PCM = SAT16(outbuf + pcm_silk), outbuf zero but for the fade-out added below: straight to HBM
                OG_FOR_LANES(i, n48 * ch) {
                    const int at = base + i;
                    if (at < nmix) pcm[at] = SL().u.out.pcm[i];
                }
                base += n48 * ch;
            }
        }, &st->loss, fec ? 2 : 0);
        if (ret) return INTERNAL_ERROR;
    }
    if (!fec && mode != MODE_CELT && rc_tell(rc) + 17 + 20 * (mode == MODE_HYBRID) <= 8 * len) {
        redundancy = mode == MODE_HYBRID ? rc_bit_logp(rc, 12) : 1;
        if (redundancy) {
            celt_to_silk = rc_bit_logp(rc, 1);
            redundancy_bytes = mode == MODE_HYBRID ? (int)rc_uint(rc, 256) + 2 : len - ((rc_tell(rc) + 7) >> 3);
            main_len = len - redundancy_bytes;
            if (main_len * 8 < rc_tell(rc)) { // (never for a valid packet; what happens then is not normative)
                main_len = 0;
                redundancy_bytes = 0;
                redundancy = 0;
                celt_lost = 1; // RFC 6716's decoder: a CELT frame of len <= 1 is a lost one
            }
            rc.storage -= (u32)redundancy_bytes; // the raw bits end where the redundant frame starts
        }
    }
    i16 *const red_hold = &SL().u.out.up[0][0]; // 240 * CC samples: the 2x up-sampler's buffers are dead once SILK's PCM is out
    if (redundancy) transition = 0;
    if (transition && mode != MODE_CELT) { // (before this frame's last band is recorded: the concealment keeps the old one)
        // (a hybrid frame's SILK PCM waits in SL().u.out.pcm for the CELT layer: the pitch-based concealment of the old mode, CELT,
        // keeps its scratch out of those bytes -- og_plc.hpp)
        const int r = conceal_chunk_rfc(st, ch, nullptr, tr_size, tr_hold);
        if (r < 0) return r;
        if (mode == MODE_HYBRID) { // the concealment's synthesis ran over the packet buffer: the frame's bytes again
            OG_SYNC();
            OG_FOR_LANES(i, len) S.pkt[i] = payload[i];
            OG_SYNC();
        }
    }
#else
    if (mode != MODE_CELT) return INTERNAL_ERROR;
    i16 *const red_hold = nullptr;
#endif
    if (OG_LANE == 0) st->loss.celt_end_band = end_band; // (what a later concealment's last band is)
    // the redundant frame: its bytes to the front of the packet buffer, a range decoder of its own, 5 ms from band 0
    auto decode_redundant = [&]() {
        OG_SYNC();
        OG_FOR_LANES(i, redundancy_bytes) S.pkt[i] = payload[main_len + i];
        OG_SYNC();
        Rc rr;
        rc_init(rr, (u32)redundancy_bytes);
        (void)celt_decode_frame(&st->celt, rr, 240, ch, CC, 0, disable_inv, end_band, &st->loss);
        OG_SYNC();
        redundant_rng = (u32)OG_UNI(st->celt.rng);
    };
    if (redundancy && celt_to_silk) {
        decode_redundant();
        OG_FOR_LANES(i, 240 * CC) {
            const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i;
            red_hold[i] = S.v[pcm_plane(c, ch, CC) + j];
        }
        OG_SYNC();
        if (mode == MODE_HYBRID) { // the main frame's bytes again: its coder is still running
            OG_FOR_LANES(i, main_len) S.pkt[i] = payload[i];
            OG_SYNC();
        }
    }
    if (mode != MODE_SILK) {
        if (mode != prev_mode && prev_mode > 0 && !prev_redundancy) {
            celt_reset_state(&st->celt);
            OG_SYNC();
        }
        OG_SYNC();
        const int lost = fec || celt_lost;
        const int Cp = lost ? CC : ch; // (a concealment runs over the decoder's channels: its PCM planes are laid out for C == CC)
        if (lost)
            celt_ret = celt_conceal(&st->celt, &st->loss, audiosize, CC, mode == MODE_CELT ? 0 : 17, end_band);
        else
            celt_ret = celt_decode_frame(&st->celt, rc, audiosize, ch, CC, mode == MODE_CELT ? 0 : 17, disable_inv, end_band, &st->loss);
#ifndef OG_NO_SILK
        if (mode == MODE_HYBRID && celt_ret >= 0) {
            OG_SYNC();
            OG_FOR_LANES(i, nmix) { // i indexes the interleaved PCM; sample j of channel c lives in plane c
                const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i, at = pcm_plane(c, Cp, CC) + j;
                S.v[at] = (i16)sat16((i32)S.v[at] + (i32)SL().u.out.pcm[i]);
            }
            OG_SYNC();
        }
#endif
        if (celt_ret >= 0) {
            OG_SYNC();
            pcm_store(pcm, audiosize, Cp, CC);
            OG_SYNC();
        }
    }
#ifndef OG_NO_SILK
    else if (prev_mode == MODE_HYBRID && !(redundancy && celt_to_silk && prev_redundancy)) {
        // RFC 6716 section 4.5.2: the MDCT fades out through a silence frame of 2.5 ms from band 0 (its return value is not looked
        // at); its 120 samples per channel are added to the SILK PCM already in HBM
        OG_SYNC();
        OG_FOR_LANES(i, 2) S.pkt[i] = 0xFF;
        OG_SYNC();
        Rc rs;
        rc_init(rs, 2u);
        (void)celt_decode_frame(&st->celt, rs, 120, ch, CC, 0, disable_inv, end_band, &st->loss);
        OG_SYNC();
        OG_FOR_LANES(i, 120 * CC) {
            const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i;
            const i32 v = S.v[pcm_plane(c, ch, CC) + j];
            pcm[i] = (i16)(i < nmix ? sat16(v + (i32)pcm[i]) : v);
        }
        OG_SYNC();
    }
    if (redundancy && !celt_to_silk) { // SILK -> CELT: a fresh CELT state, the redundant frame, its second half over the frame's end
        celt_reset_state(&st->celt);
        OG_SYNC();
        decode_redundant();
        if (celt_ret >= 0) {
            OG_FOR_LANES(i, 120 * CC) { // (the redundant audio still lies in the PCM planes: interleave its second half first)
                const int c = CC == 2 ? (i & 1) : 0, j = CC == 2 ? (i >> 1) : i;
                red_hold[i] = S.v[pcm_plane(c, ch, CC) + 120 + j];
            }
            OG_SYNC();
            rfc_smooth_fade(pcm, CC * (audiosize - 120), red_hold, 0, CC, true);
        }
    } else if (redundancy && celt_ret >= 0) { // CELT -> SILK
        OG_SYNC();
        OG_FOR_LANES(i, 120 * CC) pcm[i] = red_hold[i];
        rfc_smooth_fade(pcm, CC * 120, red_hold, CC * 120, CC, false);
    }
    if (transition && celt_ret >= 0) { // 2.5 ms of the concealment as it is, then 2.5 ms of cross-fade (a 2.5 ms frame: from its start)
        const int head = audiosize >= 240 ? 120 : 0;
        OG_SYNC();
        OG_FOR_LANES(i, head * CC) pcm[i] = tr_hold[i];
        rfc_smooth_fade(pcm, CC * head, tr_hold, CC * head, CC, false);
    }
#endif
    OG_SYNC();
    if (OG_LANE == 0) {
        st->prev_mode = mode;
        st->loss.prev_redundancy = redundancy && !celt_to_silk;
        st->frames_decoded += 1;
        st->range_final = main_len <= 1 ? 0u : rc.rng ^ redundant_rng;
    }
    OG_SYNC();
    return celt_ret < 0 ? celt_ret : audiosize;
}

} // namespace og
