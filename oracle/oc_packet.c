/*
 * oc_packet.c -- CPU ORACLE (test infrastructure): Opus packet layer and per-frame mode dispatch.
 * Restates the reference's src/opus_decoder.cpp: TOC helpers (:135-152, :460-556), frame
 * splitting (:559-680), opus_decode_frame (:154-278), opus_decode_native (:280-348), init and
 * OPUS_RESET_STATE (:82-118, :382-390).  Quirks kept: audiosize fixed at 960 (Q6), redundancy flag
 * decoded and ignored (Q2), hybrid->SILK transition decodes CELT from the live range coder (Q4).
 */
#include <stdlib.h>
#include "oc_celt_priv.h"

int oc_packet_mode(const u8 *d) { /* opus_decoder.cpp:135 */
    if (d[0] & 0x80) return OC_MODE_CELT;
    if ((d[0] & 0x60) == 0x60) return OC_MODE_HYBRID;
    return OC_MODE_SILK;
}

int oc_packet_bandwidth(const u8 *d) { /* opus_decoder.cpp:460 */
    int bw;
    if (d[0] & 0x80) {
        bw = OC_BW_MB + ((d[0] >> 5) & 0x3);
        if (bw == OC_BW_MB) bw = OC_BW_NB;
    } else if ((d[0] & 0x60) == 0x60)
        bw = (d[0] & 0x10) ? OC_BW_FB : OC_BW_SWB;
    else
        bw = OC_BW_NB + ((d[0] >> 5) & 0x3);
    return bw;
}

int oc_packet_channels(const u8 *d) { return (d[0] & 0x4) ? 2 : 1; } /* :474 */

int oc_packet_samples_per_frame(const u8 *d, i32 Fs) { /* :541 */
    int a;
    if (d[0] & 0x80) {
        a = (d[0] >> 3) & 0x3;
        a = (Fs << a) / 400;
    } else if ((d[0] & 0x60) == 0x60)
        a = (d[0] & 0x08) ? Fs / 50 : Fs / 100;
    else {
        a = (d[0] >> 3) & 0x3;
        a = a == 3 ? Fs * 60 / 1000 : (Fs << a) / 100;
    }
    return a;
}

static int parse_size(const u8 *d, i32 len, i16 *size) { /* :524 */
    if (len < 1) {
        *size = -1;
        return -1;
    }
    if (d[0] < 252) {
        *size = d[0];
        return 1;
    }
    if (len < 2) {
        *size = -1;
        return -1;
    }
    *size = 4 * d[1] + d[0];
    return 2;
}

int oc_packet_parse(const u8 *data, i32 len, int self_delimited, u8 *out_toc, i16 size[48], int *payload_offset,
                    i32 *packet_offset) { /* :559 */
    int i, bytes, count, cbr = 0, framesize;
    u8 ch, toc;
    i32 last_size, pad = 0;
    const u8 *data0 = data;
    if (size == NULL || len < 0) return OC_BAD_ARG;
    if (len == 0) return OC_INVALID_PACKET;
    framesize = oc_packet_samples_per_frame(data, 48000);
    toc = *data++;
    len--;
    last_size = len;
    switch (toc & 0x3) {
        case 0: count = 1; break;
        case 1:
            count = 2;
            cbr = 1;
            if (!self_delimited) {
                if (len & 0x1) return OC_INVALID_PACKET;
                last_size = len / 2;
                size[0] = (i16)last_size;
            }
            break;
        case 2:
            count = 2;
            bytes = parse_size(data, len, size);
            len -= bytes;
            if (size[0] < 0 || size[0] > len) return OC_INVALID_PACKET;
            data += bytes;
            last_size = len - size[0];
            break;
        default:
            if (len < 1) return OC_INVALID_PACKET;
            ch = *data++;
            count = ch & 0x3F;
            if (count <= 0 || framesize * (i32)count > 5760) return OC_INVALID_PACKET;
            len--;
            if (ch & 0x40) {
                int p;
                do {
                    int tmp;
                    if (len <= 0) return OC_INVALID_PACKET;
                    p = *data++;
                    len--;
                    tmp = p == 255 ? 254 : p;
                    len -= tmp;
                    pad += tmp;
                } while (p == 255);
            }
            if (len < 0) return OC_INVALID_PACKET;
            cbr = !(ch & 0x80);
            if (!cbr) {
                last_size = len;
                for (i = 0; i < count - 1; i++) {
                    bytes = parse_size(data, len, size + i);
                    len -= bytes;
                    if (size[i] < 0 || size[i] > len) return OC_INVALID_PACKET;
                    data += bytes;
                    last_size -= bytes + size[i];
                }
                if (last_size < 0) return OC_INVALID_PACKET;
            } else if (!self_delimited) {
                last_size = len / count;
                if (last_size * count != len) return OC_INVALID_PACKET;
                for (i = 0; i < count - 1; i++) size[i] = (i16)last_size;
            }
            break;
    }
    if (self_delimited) {
        bytes = parse_size(data, len, size + count - 1);
        len -= bytes;
        if (size[count - 1] < 0 || size[count - 1] > len) return OC_INVALID_PACKET;
        data += bytes;
        if (cbr) {
            if (size[count - 1] * count > len) return OC_INVALID_PACKET;
            for (i = 0; i < count - 1; i++) size[i] = size[count - 1];
        } else if (bytes + size[count - 1] > last_size)
            return OC_INVALID_PACKET;
    } else {
        if (last_size > 1275) return OC_INVALID_PACKET;
        size[count - 1] = (i16)last_size;
    }
    if (payload_offset) *payload_offset = (int)(data - data0);
    for (i = 0; i < count; i++) data += size[i];
    if (packet_offset) *packet_offset = pad + (i32)(data - data0);
    if (out_toc) *out_toc = toc;
    return count;
}

/* ---- decoder object ----------------------------------------------------------------------- */
void oc_decoder_set_rfc(oc_decoder *d, int on) { d->rfc = on ? 1 : 0; }

void oc_decoder_init(oc_decoder *d, int channels) { /* opus_decoder.cpp:82 */
    oc_silk *silk = d->silk;
    oc_celt_taps *taps = d->taps;
    const i32 rfc = d->rfc;
    memset(d, 0, sizeof(*d));
    d->rfc = rfc;
    d->silk = silk;
    d->taps = taps;
    d->channels = d->stream_channels = channels;
    oc_silk_init(d->silk);
    oc_celt_init(&d->celt, channels);
    d->prev_mode = 0;
    d->frame_size = 48000 / 400;
}

void oc_decoder_reset(oc_decoder *d) { /* opus_decoder.cpp:382 */
    d->stream_channels = d->bandwidth = d->mode = d->prev_mode = d->frame_size = 0;
    d->prev_redundancy = 0;
    d->last_packet_duration = 0;
    d->range_final = 0;
    oc_celt_reset(&d->celt);
    oc_silk_init(d->silk);
    d->stream_channels = d->channels;
    d->frame_size = 48000 / 400;
}

u32 oc_decoder_final_range(const oc_decoder *d) { return d->range_final; } /* (a tap of the range decoder; see oc_opus.h) */
u32 oc_decoder_ctl_final_range(const oc_decoder *d) { (void)d; return 0; } /* st->rangeFinal: declared :58, cleared :90, returned :380, never assigned */
int oc_decoder_ctl_pitch(const oc_decoder *d, i32 *value) { /* opus_decoder.cpp:399-407 */
    if (d->prev_mode == OC_MODE_CELT) return -5; /* celt_decoder_ctl((int32_t)value): no such request, celt.cpp:2532-2541 */
    *value = oc_silk_prev_pitch_lag(d->silk);
    return 0;
}

oc_decoder *oc_decoder_create(int channels) {
    oc_decoder *d = (oc_decoder *)calloc(1, sizeof(*d));
    if (!d) return NULL;
    d->silk = (oc_silk *)calloc(1, oc_silk_sizeof());
    if (!d->silk) {
        free(d);
        return NULL;
    }
    oc_decoder_init(d, channels);
    return d;
}

void oc_decoder_destroy(oc_decoder *d) {
    if (!d) return;
    free(d->silk);
    free(d);
}

/* RFC mode: CELT's last band by audio bandwidth (RFC 6716 section 4.3: NB 13, WB 17, SWB 19, FB 21 bands; the CELT-only
 * "medium band" code point does not exist, and hybrid is SWB or FB) */
static int rfc_end_band(int bandwidth) {
    switch (bandwidth) {
        case OC_BW_NB: return 13;
        case OC_BW_MB:
        case OC_BW_WB: return 17;
        case OC_BW_SWB: return 19;
        default: return 21;
    }
}

/* RFC mode, a frame with nothing to decode (the packet was lost: inbuf NULL; or a DTX frame: at most one payload byte):
 * RFC 6716's opus_decode_frame with data == NULL.  The mode is the previous frame's; `frame_size` is what is to be concealed
 * (120 << k, or a multiple of 960): more than 20 ms goes in chunks of 20 ms; SILK conceals 10 or 20 ms (2.5 / 5 ms requests take
 * the head of a 10 ms concealment); CELT (and hybrid's CELT layer, from band 17) conceals with oc_celt_decode_lost.  No
 * redundancy (a concealed frame carries none). */
static int conceal_frame(oc_decoder *d, i16 *out, int frame_size) {
    /* the last used mode: CELT if the last frame ended with CELT redundancy */
    const int mode = d->prev_redundancy ? OC_MODE_CELT : d->prev_mode, ch = d->stream_channels, CC = d->channels;
    int audiosize = frame_size, i, nmix, celt_ret = 0;
    i16 pcm_silk[960 * 2];
    if (mode == 0) { /* nothing decoded yet: all we can do is return zeros */
        for (i = 0; i < audiosize * CC; i++) out[i] = 0;
        return audiosize;
    }
    if (audiosize > 960) {
        int done = 0;
        do {
            int ret = conceal_frame(d, out + done * CC, OC_MIN(audiosize - done, 960));
            if (ret < 0) return ret;
            done += ret;
        } while (done < audiosize);
        return frame_size;
    }
    /* no concealment of sizes other than 2.5 (CELT), 5 (CELT), 10 or 20 ms: e.g. 12.5 ms conceals 10 and the caller comes back */
    if (audiosize < 960) {
        if (audiosize > 480)
            audiosize = 480;
        else if (mode != OC_MODE_SILK && audiosize > 240 && audiosize < 480)
            audiosize = 240;
    }
    nmix = audiosize * (ch < CC ? ch : CC);
    if (mode != OC_MODE_CELT) {
        int decoded = 0;
        i16 *p = pcm_silk;
        const int payload_ms = OC_MAX(10, audiosize / 48);
        do {
            i32 n = 0;
            if (oc_silk_decode_ex(d->silk, &d->rc, ch, 0, decoded == 0, payload_ms, 1, p, &n)) return OC_INTERNAL_ERROR;
            p += n * ch;
            decoded += n;
        } while (decoded < audiosize);
    }
    d->celt.start_band = mode != OC_MODE_CELT ? 17 : 0; /* the last band stays what the last frame made it */
    if (mode != OC_MODE_SILK)
        celt_ret = oc_celt_decode_lost(&d->celt, out, audiosize);
    else
        for (i = 0; i < nmix; i++) out[i] = 0;
    if (mode != OC_MODE_CELT)
        for (i = 0; i < nmix; i++) out[i] = sat16((i32)out[i] + pcm_silk[i]);
    d->prev_mode = mode;
    d->prev_redundancy = 0;
    d->range_final = 0;
    return celt_ret < 0 ? celt_ret : audiosize;
}

/* opus_decoder.cpp:154.  Reference mode: audiosize is 960 whatever the TOC says (Q6).  RFC mode: the TOC's duration. */
/* fec (RFC mode only): RFC 6716's decode_fec -- SILK decodes the LBRR copy of the frame BEFORE this packet where the packet
 * carries one (lostFlag 2), conceals otherwise; CELT has no FEC and conceals (hybrid: its layer from band 17). */
static int decode_frame(oc_decoder *d, const u8 *inbuf, i32 len, i16 *out, int fec) {
    const int mode = d->mode, ch = d->stream_channels, audiosize = d->rfc ? d->frame_size : 960;
    const int payload_ms = d->rfc ? audiosize / 48 : 20;
    /* The reference zeroes / mixes 960 * stream_channels entries of `out` even when the decoder has fewer channels (Q3): what
     * lies beyond the frame's own 960 * channels entries is the next frame's space or past the caller's buffer.  RFC mode
     * keeps the arithmetic and stays inside the frame (packets of 120 ms fill the buffer to its last entry). */
    const int nmix = d->rfc ? audiosize * (ch < d->channels ? ch : d->channels) : audiosize * ch;
    int i, c, celt_ret = 0, start_band, redundancy = 0, celt_to_silk = 0, celt_lost = 0, transition = 0;
    i32 redundancy_bytes = 0;
    u32 redundant_rng = 0;
    i16 pcm_silk[2880 * 2], redundant_audio[240 * 2], pcm_transition[240 * 2];
    oc_rc *rc = &d->rc;

    if (d->rfc && (inbuf == NULL || len <= 1)) return conceal_frame(d, out, d->frame_size);
    /* RFC 6716 section 4.5 (RFC mode; the reference has none of it): a switch between CELT-only and the SILK modes that no
     * redundant frame covers is smoothed with 5 ms of concealment from the OLD mode, cross-faded into the new frame.  The
     * concealment of the SILK modes runs before anything of the new frame is decoded, CELT's after the frame's SILK layer. */
    if (d->rfc && d->prev_mode > 0 &&
        ((mode == OC_MODE_CELT && d->prev_mode != OC_MODE_CELT && !d->prev_redundancy) || (mode != OC_MODE_CELT && d->prev_mode == OC_MODE_CELT)))
        transition = 1;
    /* (a SILK-only concealment of a mono packet in a stereo decoder defines only the first half of its entries, Q3: the rest
     * cross-fades from zero) */
    if (transition) memset(pcm_transition, 0, sizeof(pcm_transition));
    if (transition && mode == OC_MODE_CELT) {
        int ret = conceal_frame(d, pcm_transition, OC_MIN(240, audiosize));
        if (ret < 0) return ret;
    }
    oc_rc_init(rc, inbuf, len);
    if (mode != OC_MODE_CELT) {
        int decoded = 0, internal_hz;
        i16 *p = pcm_silk;
        if (d->prev_mode == OC_MODE_CELT) oc_silk_init(d->silk);
        if (mode == OC_MODE_SILK) {
            if (d->bandwidth == OC_BW_NB) internal_hz = 8000;
            else if (d->bandwidth == OC_BW_MB) internal_hz = 12000;
            else internal_hz = 16000;
        } else
            internal_hz = 16000;
        do {
            i32 n = 0;
            int ret = oc_silk_decode_ex(d->silk, rc, ch, internal_hz, decoded == 0, payload_ms, fec ? 2 : 0, p, &n);
            if (ret) return OC_INTERNAL_ERROR;
            p += n * ch;
            decoded += n;
        } while (decoded < audiosize);
    }
    start_band = 0;
    if (!fec && mode != OC_MODE_CELT && oc_rc_tell(rc) + 17 + 20 * (mode == OC_MODE_HYBRID) <= 8 * len) {
        if (!d->rfc) {
            if (mode == OC_MODE_HYBRID) (void)oc_rc_bit_logp(rc, 12); /* redundancy flag: ignored (Q2) */
        } else {
            /* RFC 6716 section 4.5.1: a redundant 5 ms CELT frame for the mode transitions.  Hybrid frames flag it; in a SILK-only
             * frame whatever follows the SILK data is one.  Then: which side of the frame it belongs to, and its size. */
            redundancy = mode == OC_MODE_HYBRID ? oc_rc_bit_logp(rc, 12) : 1;
            if (redundancy) {
                celt_to_silk = oc_rc_bit_logp(rc, 1);
                redundancy_bytes = mode == OC_MODE_HYBRID ? (i32)oc_rc_uint(rc, 256) + 2 : len - ((oc_rc_tell(rc) + 7) >> 3);
                len -= redundancy_bytes;
                if (len * 8 < oc_rc_tell(rc)) { /* (never for a valid packet; what happens then is not normative) */
                    len = 0;
                    redundancy_bytes = 0;
                    redundancy = 0;
                    celt_lost = 1; /* RFC 6716's decoder: a CELT frame of len <= 1 is a lost one */
                }
                rc->storage -= redundancy_bytes; /* the raw bits end where the redundant frame starts */
            }
        }
    }
    if (mode != OC_MODE_CELT) start_band = 17;
    if (redundancy) transition = 0;
    if (transition && mode != OC_MODE_CELT) { /* (before the frame's own last band is set: the concealment keeps the old one) */
        int ret = conceal_frame(d, pcm_transition, OC_MIN(240, audiosize));
        if (ret < 0) return ret;
    }
    if (d->bandwidth) d->celt.stream_channels = ch; /* END_BAND request has no effect (Q1) */
    d->celt.end_band = d->rfc ? rfc_end_band(d->bandwidth) : OC_NBANDS;

    if (redundancy && celt_to_silk) { /* the 5 ms redundant frame of a CELT -> SILK transition: decoded first, heard first */
        oc_rc rr;
        d->celt.start_band = 0;
        oc_rc_init(&rr, inbuf + len, redundancy_bytes);
        (void)oc_celt_decode(&d->celt, &rr, redundant_audio, 240, NULL);
        redundant_rng = d->celt.rng;
    }
    d->celt.start_band = start_band;

    if (mode != OC_MODE_SILK) {
        if (mode != d->prev_mode && d->prev_mode > 0 && !d->prev_redundancy) oc_celt_reset(&d->celt);
        if (fec || celt_lost)
            celt_ret = oc_celt_decode_lost(&d->celt, out, audiosize);
        else
            celt_ret = oc_celt_decode(&d->celt, rc, out, audiosize, d->taps); /* (CELT frames are 2.5 - 20 ms) */
    } else {
        for (i = 0; i < nmix; i++) out[i] = 0;
        if (d->prev_mode == OC_MODE_HYBRID && !(redundancy && celt_to_silk && d->prev_redundancy)) {
            d->celt.start_band = 0;
            if (d->rfc) { /* RFC 6716 section 4.5.2: let the MDCT fade out by decoding a silence frame */
                static const u8 silence[2] = {0xFF, 0xFF};
                oc_rc rs;
                oc_rc_init(&rs, silence, 2);
                (void)oc_celt_decode(&d->celt, &rs, out, 120, NULL);
            } else /* Q4: the reference runs the 2.5 ms frame off the coder SILK has just used */
                (void)oc_celt_decode(&d->celt, rc, out, 120, NULL);
        }
    }
    if (mode != OC_MODE_CELT)
        for (i = 0; i < nmix; i++) out[i] = sat16((i32)out[i] + pcm_silk[i]);
    if (redundancy) { /* RFC 6716 section 4.5.1.4 (smooth_fade: the squared CELT window over 2.5 ms) */
        const int CC = d->channels;
        i16 *a, *b, *o;
        if (!celt_to_silk) { /* SILK -> CELT: the redundant frame starts a fresh CELT state; its second half fades in at the end */
            oc_rc rr;
            oc_celt_reset(&d->celt);
            d->celt.start_band = 0;
            oc_rc_init(&rr, inbuf + len, redundancy_bytes);
            (void)oc_celt_decode(&d->celt, &rr, redundant_audio, 240, NULL);
            redundant_rng = d->celt.rng;
            a = out + CC * (audiosize - 120);
            b = redundant_audio + CC * 120;
            o = a;
        } else { /* CELT -> SILK: its first half replaces the frame's start, its second half fades out into the frame */
            for (i = 0; i < 120 * CC; i++) out[i] = redundant_audio[i];
            a = redundant_audio + CC * 120;
            b = out + CC * 120;
            o = b;
        }
        for (c = 0; c < CC; c++)
            for (i = 0; i < 120; i++) {
                const i32 w = m16_q15(rom_win120[i], rom_win120[i]);
                o[i * CC + c] = (i16)((m16(w, b[i * CC + c]) + m16(32767 - w, a[i * CC + c])) >> 15);
            }
    }
    if (transition) { /* 2.5 ms of the concealment as it is, then 2.5 ms of cross-fade; a 2.5 ms frame is cross-faded from its start */
        const int CC = d->channels, head = audiosize >= 240 ? 120 : 0;
        for (i = 0; i < head * CC; i++) out[i] = pcm_transition[i];
        for (c = 0; c < CC; c++)
            for (i = 0; i < 120; i++) {
                const i32 w = m16_q15(rom_win120[i], rom_win120[i]);
                const int at = (head + i) * CC + c;
                out[at] = (i16)((m16(w, out[at]) + m16(32767 - w, pcm_transition[at])) >> 15);
            }
    }
    d->range_final = len <= 1 ? 0 : rc->rng ^ redundant_rng;
    d->prev_mode = mode;
    d->prev_redundancy = redundancy && !celt_to_silk;
    d->last_redundancy = redundancy | celt_to_silk << 1;
    return celt_ret < 0 ? celt_ret : audiosize;
}

/* opus_decoder.cpp:280 (self_delimited = 0; data != NULL) */
int oc_decode(oc_decoder *d, const u8 *data, i32 len, i16 *pcm, int frame_size) {
    int i, nb_samples = 0, count, offset, pfs, pbw, pmode, pch;
    i16 size[48];
    u8 toc;
    if (frame_size <= 0) return OC_BAD_ARG;
    if (d->rfc && (len == 0 || data == NULL)) { /* a lost packet: conceal frame_size samples, a frame of the last size at a time */
        int done = 0;
        if (frame_size % 120) return OC_BAD_ARG;
        while (done < frame_size) {
            const int want = OC_MIN(frame_size - done, d->frame_size);
            int ret = conceal_frame(d, pcm + done * d->channels, want);
            if (ret < 0) return ret;
            done += ret;
        }
        d->last_packet_duration = done;
        return done;
    }
    /* opus_decoder.cpp:290-308: an empty packet.  The reference has no concealment, but this branch is live: it runs
     * opus_decode_frame(st, NULL, 0, ...) -- a frame of NO bytes in the decoder's LAST mode / bandwidth / channel count (mode 0
     * before the first packet and after OPUS_RESET_STATE, which :175 / :249 treat like hybrid: SILK at 16 kHz, then CELT) --
     * 960 samples per pass (:161; the room left over is ignored) until frame_size is filled.  SILK-only: 960 samples of PCM per
     * pass and the state moves on; hybrid or no mode yet: the SILK half runs, then celt_decode_with_ec refuses the empty frame
     * (celt.cpp:2225, -18) with prev_mode updated all the same (:276); CELT-only: -18, nothing but prev_mode touched.
     * The caller needs room for the whole passes: ceil(frame_size / 960) * 960 samples (the reference writes them too, and
     * its assert at :306 fires when frame_size is no multiple of 960). */
    if (len == 0 || data == NULL) {
        int pcm_count = 0;
        if (frame_size % 120) return OC_BAD_ARG; /* :290 */
        do {
            int ret = decode_frame(d, NULL, 0, pcm + pcm_count * d->channels, 0);
            if (ret < 0) return ret;
            pcm_count += ret;
        } while (pcm_count < frame_size);
        d->last_packet_duration = pcm_count;
        return pcm_count;
    }
    if (len < 0) return OC_BAD_ARG; /* :309 */
    pmode = oc_packet_mode(data);
    pbw = oc_packet_bandwidth(data);
    pfs = oc_packet_samples_per_frame(data, 48000);
    pch = oc_packet_channels(data);
    count = oc_packet_parse(data, len, 0, &toc, size, &offset, NULL);
    if (count < 0) return count;
    data += offset;
    if (count * pfs > frame_size) return OC_BUFFER_TOO_SMALL;
    d->mode = pmode;
    d->bandwidth = pbw;
    d->frame_size = pfs;
    d->stream_channels = pch;
    for (i = 0; i < count; i++) {
        int ret = decode_frame(d, data, size[i], pcm + nb_samples * d->channels, 0);
        if (ret < 0) return ret;
        data += size[i];
        nb_samples += ret;
    }
    d->last_packet_duration = nb_samples;
    return nb_samples;
}

/* RFC mode: opus_decode(decode_fec = 1) of RFC 6716's decoder -- the packet BEFORE `data` was lost; conceal frame_size samples
 * (what the lost packet carried), the last packet_frame_size of them from the forward error correction data (SILK's LBRR
 * frames) in the first frame of `data`, where there is any.  A CELT-only packet (or predecessor) has none: plain concealment.
 * The caller decodes `data` itself afterwards, normally. */
int oc_decode_fec(oc_decoder *d, const u8 *data, i32 len, i16 *pcm, int frame_size) {
    int count, offset, pfs, pbw, pmode, pch, ret, dur;
    i16 size[48];
    u8 toc;
    if (!d->rfc || frame_size <= 0 || frame_size % 120) return OC_BAD_ARG;
    if (len <= 0 || data == NULL) return oc_decode(d, NULL, 0, pcm, frame_size);
    pmode = oc_packet_mode(data);
    pbw = oc_packet_bandwidth(data);
    pfs = oc_packet_samples_per_frame(data, 48000);
    pch = oc_packet_channels(data);
    count = oc_packet_parse(data, len, 0, &toc, size, &offset, NULL);
    if (count < 0) return count;
    data += offset;
    if (frame_size < pfs || pmode == OC_MODE_CELT || d->mode == OC_MODE_CELT) return oc_decode(d, NULL, 0, pcm, frame_size);
    dur = d->last_packet_duration;
    if (frame_size - pfs != 0) {
        ret = oc_decode(d, NULL, 0, pcm, frame_size - pfs);
        if (ret < 0) {
            d->last_packet_duration = dur;
            return ret;
        }
    }
    d->mode = pmode;
    d->bandwidth = pbw;
    d->frame_size = pfs;
    d->stream_channels = pch;
    ret = decode_frame(d, data, size[0], pcm + d->channels * (frame_size - pfs), 1);
    if (ret < 0) return ret;
    d->last_packet_duration = frame_size;
    return frame_size;
}
