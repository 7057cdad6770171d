// og_compat.cpp -- the reference's C++ call surface (include/opus_decoder.h, include/opusfile.h) implemented
// over the C ABI (include/opusgpu.h).  Host logic only: framing, channel mapping, ctl bookkeeping, the Ogg
// container reader.  All codec arithmetic happens in opusgpu_decode_packets(), i.e. on the GPU.
#include <new>
#include <stdlib.h>
#include <string.h>
#include "og_container.hpp" // before opusfile.h: its OP_* macros would otherwise rewrite the enum of the same names
#include "og_packet.hpp"
#include "../../include/opus_decoder.h"
#include "../../include/opusfile.h"
#include "../../include/opusgpu.h"

// ---- packet helpers -------------------------------------------------------------------------------------
int opus_packet_get_bandwidth(uint8_t *data) { return ogh::toc_bandwidth(data[0]); }
int opus_packet_get_samples_per_frame(uint8_t *data, int32_t Fs) { return ogh::toc_samples_per_frame(data[0], Fs); }
int opus_packet_get_nb_channels(uint8_t *data) { return ogh::toc_channels(data[0]); }
int opus_packet_get_nb_frames(uint8_t packet[], int32_t len) {
    if (len < 1) return OPUS_BAD_ARG;
    int count = packet[0] & 3;
    if (count == 0) return 1;
    if (count != 3) return 2;
    if (len < 2) return OPUS_INVALID_PACKET;
    return packet[1] & 0x3F;
}
int opus_packet_get_nb_samples(uint8_t packet[], int32_t len, int32_t Fs) {
    int count = opus_packet_get_nb_frames(packet, len);
    if (count < 0) return count;
    int samples = count * opus_packet_get_samples_per_frame(packet, Fs);
    return samples * 25 > Fs * 3 ? OPUS_INVALID_PACKET : samples;
}
int opus_packet_parse(uint8_t *data, int32_t len, unsigned char *out_toc, uint8_t *frames[48], int16_t size[48],
                      int *payload_offset) {
    return opus_packet_parse_impl(data, len, 0, out_toc, frames, size, payload_offset, nullptr);
}
int opus_packet_parse_impl(uint8_t *data, int32_t len, int self_delimited, unsigned char *out_toc, uint8_t *frames[48],
                           int16_t size[48], int *payload_offset, int32_t *packet_offset) { // src/opus_decoder.cpp:559-680
    int off = 0;
    int n = ogh::parse_packet(data, len, self_delimited, out_toc, size, &off, packet_offset);
    if (n < 0) return n;
    if (payload_offset) *payload_offset = off;
    if (frames) {
        uint8_t *p = data + off;
        for (int i = 0; i < n; i++) {
            frames[i] = p;
            p += size[i];
        }
    }
    return n;
}

// ---- one GPU-backed elementary decoder ---------------------------------------------------------------------
struct OpusDecoder {
    opusgpu_ctx *ctx;
    int channels;
    int32_t Fs;
    int decode_gain;
    int bandwidth, last_packet_duration;
    // silk_DecControlStruct::prevPitchLag as the reference keeps it: exported at the end of every silk_Decode (src/silk.cpp:1764-1769)
    // and by nothing else -- OPUS_RESET_STATE leaves it, so what OPUS_GET_PITCH reports after a reset is the last value.  Read from
    // the device when asked (or before a reset), if a decode call has run since.
    int32_t prev_pitch_lag;
    bool pitch_stale;
};

static void pitch_refresh(OpusDecoder *d) {
    if (!d->pitch_stale) return;
    int32_t v[4];
    if (opusgpu_stream_pitch_get(d->ctx, 0, v) == OPUSGPU_OK && (v[3] == 8 || v[3] == 12 || v[3] == 16))
        d->prev_pitch_lag = v[1] == 2 ? v[2] * (v[3] == 8 ? 6 : v[3] == 12 ? 4 : 3) : 0; // the lag at 48 kHz (mult_tab, silk.cpp:1765)
    d->pitch_stale = false;
}

static int dec_open(OpusDecoder *d, int32_t Fs, int channels) {
    if ((Fs != 48000 && Fs != 24000 && Fs != 16000 && Fs != 12000 && Fs != 8000) || (channels != 1 && channels != 2))
        return OPUS_BAD_ARG;
    if (Fs != 48000) return OPUS_UNIMPLEMENTED; // the reference pins the output rate to 48 kHz (src/opus_decoder.cpp:169)
    memset(d, 0, sizeof(*d));
    int rc = opusgpu_ctx_create(0, &d->ctx);
    if (rc != OPUSGPU_OK) return rc == OPUSGPU_ALLOC_FAIL ? OPUS_ALLOC_FAIL : OPUS_INTERNAL_ERROR;
    rc = opusgpu_streams_alloc(d->ctx, 1, channels);
    if (rc != OPUSGPU_OK) {
        opusgpu_ctx_destroy(d->ctx);
        d->ctx = nullptr;
        return OPUS_ALLOC_FAIL;
    }
    d->channels = channels;
    d->Fs = Fs;
    return OPUS_OK;
}

static int dec_decode(OpusDecoder *d, const uint8_t *data, int32_t len, int16_t *pcm, int frame_size) {
    if (!d || !d->ctx || frame_size <= 0) return OPUS_BAD_ARG; // src/opus_decoder.cpp:351
    if (len == 0 || data == nullptr) {
        // The reference's empty-packet branch (src/opus_decoder.cpp:290-308; include/opusgpu.h "EMPTY PACKETS"): passes of
        // opus_decode_frame(NULL, 0) -- 960 samples each, in the decoder's last mode -- until frame_size is filled or one fails.
        if (frame_size % 120) return OPUS_BAD_ARG; // :290
        // A frame_size that is no multiple of 960 makes the reference write the whole last pass past what the caller offered and
        // trip its assert (:306): refused here instead -- the one deliberate difference of this branch.
        if (frame_size % 960) return OPUS_BUFFER_TOO_SMALL;
        const int passes = frame_size / 960;
        int32_t id = 0, res = 0, zero = 0;
        const uint8_t *none = nullptr;
        int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)passes * 960 * d->channels);
        if (!buf) return OPUS_ALLOC_FAIL;
        int rc = opusgpu_decode_packets(d->ctx, 1, &id, &none, &zero, buf, passes, &res);
        if (rc != OPUSGPU_OK) res = OPUS_INTERNAL_ERROR;
        d->pitch_stale = true;
        if (res > 0) {
            memcpy(pcm, buf, sizeof(int16_t) * (size_t)res * d->channels);
            d->last_packet_duration = res; // :307
        }
        free(buf);
        return res;
    }
    if (len < 0) return OPUS_BAD_ARG; // :309
    const int pfs = ogh::toc_samples_per_frame(data[0], 48000);
    uint8_t toc;
    int16_t size[48];
    const int count = ogh::parse_packet(data, len, 0, &toc, size, nullptr, nullptr);
    if (count < 0) return count;
    if (count * pfs > frame_size) return OPUS_BUFFER_TOO_SMALL;
    d->bandwidth = ogh::toc_bandwidth(toc); // :328: with the packet accepted, whatever its frames return
    // the GPU path writes 960 samples per frame (Q6); decode into a scratch block sized for `count` frames
    int32_t id = 0, res = 0;
    const uint8_t *pk = data;
    // (room, in 20 ms blocks, for what the C ABI checks: the frames at 960 samples each AND the duration the TOC names -- a 60 ms
    // TOC passes :323 with frame_size 2880 and then decodes 960 samples)
    const int cap = count * pfs > count * 960 ? (count * pfs + 959) / 960 : count;
    int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)cap * 960 * d->channels);
    if (!buf) return OPUS_ALLOC_FAIL;
    int rc = opusgpu_decode_packets(d->ctx, 1, &id, &pk, &len, buf, cap, &res);
    if (rc != OPUSGPU_OK) res = OPUS_INTERNAL_ERROR;
    d->pitch_stale = true; // (frames may have run a SILK decode, whatever they returned)
    if (res > 0) {
        int n = res < frame_size ? res : frame_size;
        memcpy(pcm, buf, sizeof(int16_t) * (size_t)n * d->channels);
        d->last_packet_duration = res;
    }
    free(buf);
    return res;
}

static int dec_ctl(OpusDecoder *d, int request, va_list ap) {
    switch (request) {
        case OPUS_GET_BANDWIDTH_REQUEST: { int32_t *v = va_arg(ap, int32_t *); if (!v) return OPUS_BAD_ARG; *v = d->bandwidth; return OPUS_OK; }
        case OPUS_GET_SAMPLE_RATE_REQUEST: { int32_t *v = va_arg(ap, int32_t *); if (!v) return OPUS_BAD_ARG; *v = d->Fs; return OPUS_OK; }
        case OPUS_GET_GAIN_REQUEST: { int32_t *v = va_arg(ap, int32_t *); if (!v) return OPUS_BAD_ARG; *v = d->decode_gain; return OPUS_OK; }
        case OPUS_SET_GAIN_REQUEST: { int32_t v = va_arg(ap, int32_t); if (v < -32768 || v > 32767) return OPUS_BAD_ARG; d->decode_gain = v; return OPUS_OK; } // stored, never applied (Q7)
        case OPUS_GET_LAST_PACKET_DURATION_REQUEST: { int32_t *v = va_arg(ap, int32_t *); if (!v) return OPUS_BAD_ARG; *v = d->last_packet_duration; return OPUS_OK; }
        case OPUS_GET_FINAL_RANGE_REQUEST: {
            // The reference's OpusDecoder::rangeFinal is declared (src/opus_decoder.cpp:58), cleared with the struct (:90) and
            // returned here (:375-380) -- and never assigned: opus_decode_frame ends at :276 without the `rangeFinal = dec.rng ^
            // redundant_rng` of RFC 6716's decoder.  The ctl reports 0, always; so does this.  (The range decoder's last range of a
            // stream is in its state record all the same -- opusgpu_stream_state_get, word 3 -- and parity tests compare it.)
            uint32_t *v = va_arg(ap, uint32_t *);
            if (!v) return OPUS_BAD_ARG;
            *v = 0;
            return OPUS_OK;
        }
        case OPUS_GET_PITCH_REQUEST: {
            // src/opus_decoder.cpp:399-407.  After a CELT-only frame the reference calls celt_decoder_ctl with the POINTER as the
            // request number: no case of that switch is a pointer's value, it answers OPUS_UNIMPLEMENTED (src/celt.cpp:2532-2541).
            // Otherwise -- SILK-only / hybrid, or no frame yet -- the SILK decoder's last exported pitch lag at 48 kHz.
            int32_t *v = va_arg(ap, int32_t *);
            if (!v) return OPUS_BAD_ARG;
            int32_t st[4];
            if (opusgpu_stream_pitch_get(d->ctx, 0, st) != OPUSGPU_OK) return OPUS_INTERNAL_ERROR;
            if (st[0] == MODE_CELT_ONLY) return OPUS_UNIMPLEMENTED;
            pitch_refresh(d);
            *v = d->prev_pitch_lag;
            return OPUS_OK;
        }
        case OPUS_RESET_STATE:
            pitch_refresh(d); // (what the reset does not clear: see prev_pitch_lag)
            d->bandwidth = 0;
            d->last_packet_duration = 0;
            return opusgpu_streams_reset(d->ctx, 0, 1, 0) == OPUSGPU_OK ? OPUS_OK : OPUS_INTERNAL_ERROR;
        default: return OPUS_UNIMPLEMENTED;
    }
}

int opus_decoder_get_size(int channels) { return (channels < 1 || channels > 2) ? 0 : (int)sizeof(OpusDecoder); }
int opus_decoder_init(OpusDecoder *st, int32_t Fs, int channels) { return st ? dec_open(st, Fs, channels) : OPUS_BAD_ARG; }
int opus_decode_native(OpusDecoder *st, uint8_t *data, int32_t len, int16_t *pcm, int frame_size, int self_delimited,
                       int32_t *packet_offset) { // src/opus_decoder.cpp:280-348 (no check of frame_size here: opus_decode's, :351)
    if (self_delimited) return OPUS_UNIMPLEMENTED;
    if (frame_size <= 0) { // (opus_decode refuses these, :351; here the reference goes on: a packet fails its size check, :323 -- after
                           // its parse, :318 -- and an empty packet's loop would run into memory it was not given: refused)
        if (len == 0 || data == nullptr || len < 0) return OPUS_BAD_ARG;
        uint8_t toc;
        int16_t size[48];
        const int count = ogh::parse_packet(data, len, 0, &toc, size, nullptr, nullptr);
        return count < 0 ? count : OPUS_BUFFER_TOO_SMALL;
    }
    const int ret = dec_decode(st, data, len, pcm, frame_size);
    if (packet_offset && len > 0 && data) { // what opus_packet_parse_impl reports for the packet (:676-677): all of it
        uint8_t toc;
        int16_t size[48];
        int32_t po = 0;
        if (ogh::parse_packet(data, len, 0, &toc, size, nullptr, &po) >= 0) *packet_offset = po;
    }
    return ret;
}
int opus_decode(OpusDecoder *st, uint8_t *data, int32_t len, int16_t *pcm, int frame_size) {
    return dec_decode(st, data, len, pcm, frame_size);
}
// opus_decoder_get_nb_samples src/opus_decoder.cpp:507: the packet's sample count at the decoder's rate
int opus_decoder_get_nb_samples(const OpusDecoder *dec, uint8_t packet[], int32_t len) {
    return opus_packet_get_nb_samples(packet, len, dec->Fs);
}
int opus_decoder_ctl(OpusDecoder *st, int request, ...) {
    if (!st) return OPUS_BAD_ARG;
    va_list ap;
    va_start(ap, request);
    int r = dec_ctl(st, request, ap);
    va_end(ap);
    return r;
}
void opus_decoder_destroy(OpusDecoder *st) {
    if (!st) return;
    if (st->ctx) opusgpu_ctx_destroy(st->ctx);
    st->ctx = nullptr;
}

// ---- multistream wrapper ---------------------------------------------------------------------------------------
struct MSImpl {
    OpusMSDecoder_t pub; // must be first: callers see an OpusMSDecoder_t*
    OpusDecoder dec;
};

int32_t opus_multistream_decoder_get_size(int streams, int coupled) {
    if (streams < 1 || coupled > streams || coupled < 0) return 0;
    return (int32_t)sizeof(MSImpl);
}

int opus_multistream_decoder_init(OpusMSDecoder_t *st, int32_t Fs, int channels, int streams, int coupled_streams,
                                  const uint8_t *mapping) {
    if (!st || channels > 255 || channels < 1 || coupled_streams > streams || streams < 1 || coupled_streams < 0 ||
        streams > 255 - coupled_streams)
        return OPUS_BAD_ARG;
    if (streams != 1) return OPUS_UNIMPLEMENTED; // the reference cannot really run >1 stream either (global codec state)
    MSImpl *m = reinterpret_cast<MSImpl *>(st);
    for (int i = 0; i < channels; i++) {
        if (mapping[i] >= streams + coupled_streams && mapping[i] != 255) return OPUS_BAD_ARG; // validate_layout
    }
    const int dch = coupled_streams ? 2 : 1;
    if (m->dec.ctx && m->dec.channels == dch) { // re-init of an existing object: fresh codec state
        pitch_refresh(&m->dec); // (the SILK control structure is not part of what an init clears)
        if (opusgpu_streams_reset(m->dec.ctx, 0, 1, 1) != OPUSGPU_OK) return OPUS_INTERNAL_ERROR;
        m->dec.bandwidth = m->dec.last_packet_duration = 0;
    } else {
        if (m->dec.ctx) opus_decoder_destroy(&m->dec);
        int rc = dec_open(&m->dec, Fs, dch);
        if (rc != OPUS_OK) return rc;
    }
    st->nb_channels = channels;
    st->nb_streams = streams;
    st->nb_coupled_streams = coupled_streams;
    for (int i = 0; i < channels; i++) st->mapping[i] = mapping[i];
    return OPUS_OK;
}

OpusMSDecoder_t *opus_multistream_decoder_create(int32_t Fs, int channels, int streams, int coupled_streams,
                                                 const uint8_t *mapping, int *error) {
    MSImpl *m = (MSImpl *)calloc(1, sizeof(MSImpl));
    if (!m) {
        if (error) *error = OPUS_ALLOC_FAIL;
        return nullptr;
    }
    int ret = opus_multistream_decoder_init(&m->pub, Fs, channels, streams, coupled_streams, mapping);
    if (error) *error = ret;
    if (ret != OPUS_OK) {
        free(m);
        return nullptr;
    }
    return &m->pub;
}

static void copy_channel_out_short(void *dst, int dst_stride, int dst_channel, const int16_t *src, int src_stride, int frame_size,
                                   void *) { // opus_copy_channel_out_short, src/opus_decoder.cpp:916-928
    int16_t *d = (int16_t *)dst;
    for (int i = 0; i < frame_size; i++) d[i * dst_stride + dst_channel] = src ? src[i * src_stride] : (int16_t)0;
}
int opus_multistream_decode(OpusMSDecoder_t *st, uint8_t *data, int32_t len, int16_t *pcm, int frame_size) {
    return opus_multistream_decode_native(st, data, len, pcm, copy_channel_out_short, frame_size);
}
int opus_multistream_decode_native(OpusMSDecoder_t *st, uint8_t *data, int32_t len, void *pcm, opus_copy_channel_out_func copy_channel_out,
                                   int frame_size) {
    // opus_multistream_decode_native (src/opus_decoder.cpp:826-913), one stream: frame_size <= 0 -> -1 (:836); frame_size capped at
    // 120 ms (:841); len < 0 -> -1 (:848); len == 0 is the empty packet of opus_decode_native (do_plc, :847), len >= 1 passes
    // :851's `len < 2 * nb_streams - 1`; opus_multistream_packet_validate's answers (:854-860) are those of dec_decode's own parse
    // and size check; `ret <= 0` comes back as it is (:876)
    if (!st || !copy_channel_out || frame_size <= 0 || len < 0) return OPUS_BAD_ARG;
    MSImpl *m = reinterpret_cast<MSImpl *>(st);
    if (frame_size > 5760) frame_size = 5760;
    const int dch = m->dec.channels;
    int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)frame_size);
    if (!buf) return OPUS_ALLOC_FAIL;
    int ret = dec_decode(&m->dec, data, len, buf, frame_size);
    if (ret > 0) { // the caller's copy function by mapping (src/opus_decoder.cpp:881-910: left, right / mono, then the muted channels)
        // A packet of more short frames than frame_size has room for at 960 samples each returns more than frame_size
        // (Q6): only what the caller has room for -- and what `buf` holds -- is handed out (dec_decode did the same).
        const int n = ret < frame_size ? ret : frame_size;
        for (int side = 0; side < (dch == 2 ? 2 : 1); side++)
            for (int c = 0; c < st->nb_channels; c++)
                if (st->mapping[c] == side) copy_channel_out(pcm, st->nb_channels, c, buf + side, dch, n, nullptr);
        for (int c = 0; c < st->nb_channels; c++)
            if (st->mapping[c] == 255) copy_channel_out(pcm, st->nb_channels, c, nullptr, 0, n, nullptr);
    }
    free(buf);
    return ret;
}

int opus_multistream_decoder_ctl(OpusMSDecoder_t *st, int request, ...) {
    va_list ap;
    va_start(ap, request);
    const int r = opus_multistream_decoder_ctl_va_list(st, request, ap);
    va_end(ap);
    return r;
}
int opus_multistream_decoder_ctl_va_list(OpusMSDecoder_t *st, int request, va_list ap) { // src/opus_decoder.cpp:936-1030
    if (!st) return OPUS_BAD_ARG;
    MSImpl *m = reinterpret_cast<MSImpl *>(st);
    int r;
    if (request == OPUS_MULTISTREAM_GET_DECODER_STATE_REQUEST) {
        int32_t id = va_arg(ap, int32_t);
        OpusDecoder **v = va_arg(ap, OpusDecoder **);
        if (id != 0 || !v) r = OPUS_BAD_ARG;
        else { *v = &m->dec; r = OPUS_OK; }
    } else if (request == OPUS_GET_PITCH_REQUEST)
        r = OPUS_UNIMPLEMENTED; // (not among the requests the reference's multistream ctl passes on, src/opus_decoder.cpp:945-1026)
    else
        r = dec_ctl(&m->dec, request, ap);
    return r;
}

void opus_multistream_decoder_destroy(OpusMSDecoder_t *st) {
    if (!st) return;
    MSImpl *m = reinterpret_cast<MSImpl *>(st);
    opus_decoder_destroy(&m->dec);
    free(m);
}

// ---- the player's container entry points -------------------------------------------------------------------
struct OggOpusFile {
    ogc::OpusFile *of;
    OpusMSDecoder_t *od;
};
static OggOpusFile g_player = {nullptr, nullptr};

static int player_decode(void *user, const uint8_t *pkt, int32_t len, int16_t *pcm, int frame_size) {
    OggOpusFile *p = (OggOpusFile *)user;
    return opus_multistream_decode(p->od, const_cast<uint8_t *>(pkt), len, pcm, frame_size);
}

int opus_head_parse(OpusHead_t *_head, uint8_t *_data, size_t _len) {
    ogc::Head h;
    int r = ogc::parse_head(&h, _data, _len);
    if (r < 0 || !_head) return r;
    _head->version = h.version; _head->channel_count = h.channel_count; _head->pre_skip = h.pre_skip;
    _head->input_sample_rate = h.input_sample_rate; _head->output_gain = h.output_gain;
    _head->mapping_family = h.mapping_family; _head->stream_count = h.stream_count; _head->coupled_count = h.coupled_count;
    memcpy(_head->mapping, h.mapping, sizeof(_head->mapping));
    return 0;
}

void opus_close_decoder() {
    delete g_player.of;
    g_player.of = nullptr;
    if (g_player.od) opus_multistream_decoder_destroy(g_player.od);
    g_player.od = nullptr;
}

OggOpusFile_t *opus_init_decoder() {
    opus_close_decoder();
    if (!SD_read) return nullptr;
    g_player.of = new (std::nothrow) ogc::OpusFile(SD_read, player_decode, &g_player);
    if (!g_player.of) return nullptr;
    if (g_player.of->open() < 0) {
        opus_close_decoder();
        return nullptr;
    }
    const ogc::Head &h = g_player.of->head();
    int err = 0;
    g_player.od = opus_multistream_decoder_create(48000, h.channel_count, h.stream_count, h.coupled_count, h.mapping, &err);
    if (!g_player.od) {
        opus_close_decoder();
        return nullptr;
    }
    return &g_player;
}

int op_read_stereo(int16_t *_pcm, int _buf_size) {
    if (!g_player.of) return OP_EINVAL;
    return g_player.of->read_stereo(_pcm, _buf_size);
}
