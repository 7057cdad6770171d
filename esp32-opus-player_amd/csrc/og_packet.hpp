// og_packet.hpp -- host-side Opus packet framing (RFC 6716 section 3) for the batcher.
// Behaviour follows the reference's opus_packet_parse_impl and TOC helpers
// (src/opus_decoder.cpp:135-152, :460-474, :524-556, :559-680); results feed opusgpu_frame_desc.
#pragma once
#include <stdint.h>
#include "../../include/opusgpu.h"

namespace ogh {

enum { MODE_SILK = 1000, MODE_HYBRID = 1001, MODE_CELT = 1002 };
enum { BW_NB = 1101 };

inline int toc_mode(uint8_t toc) {
    if (toc & 0x80) return MODE_CELT;
    if ((toc & 0x60) == 0x60) return MODE_HYBRID;
    return MODE_SILK;
}
inline int toc_bandwidth(uint8_t toc) { // 1101..1105
    if (toc & 0x80) {
        int bw = 1102 + ((toc >> 5) & 3);
        return bw == 1102 ? 1101 : bw;
    }
    if ((toc & 0x60) == 0x60) return (toc & 0x10) ? 1105 : 1104;
    return 1101 + ((toc >> 5) & 3);
}
inline int toc_channels(uint8_t toc) { return (toc & 4) ? 2 : 1; }
inline int toc_samples_per_frame(uint8_t toc, int32_t Fs) {
    if (toc & 0x80) return (Fs << ((toc >> 3) & 3)) / 400;
    if ((toc & 0x60) == 0x60) return (toc & 0x08) ? Fs / 50 : Fs / 100;
    int a = (toc >> 3) & 3;
    return a == 3 ? Fs * 60 / 1000 : (Fs << a) / 100;
}
inline int32_t toc_flags(uint8_t toc) {
    return (toc_mode(toc) - MODE_SILK) | ((toc_bandwidth(toc) - BW_NB) << 2) | ((toc & 4) ? 32 : 0);
}
// Reference mode, the descriptor flags of an EMPTY packet's frames for a stream that has had no packet since its reset: the
// reference's decoder is in mode 0 then (src/opus_decoder.cpp:82 / :382 clear it), which opus_decode_frame runs like a hybrid
// frame -- SILK at 16 kHz on st->channels channels (:175-201), then CELT, which refuses the empty frame -- except that prev_mode
// stays 0 (:276).  Hybrid, bandwidth irrelevant (SWB), the decoder's channel count, bit 11 (og_state.hpp desc_mode_after).
inline int32_t empty_flags_no_packet_yet(int decoder_channels) {
    return (MODE_HYBRID - MODE_SILK) | 3 << 2 | (decoder_channels == 2 ? 32 : 0) | OPUSGPU_DESC_NO_MODE;
}
// RFC mode: the frame's duration (bits 6-8) and the mode bit (bit 9) on top of toc_flags
inline int32_t toc_flags_rfc(uint8_t toc) {
    const int n = toc_samples_per_frame(toc, 48000);
    const int code = n == 120 ? 1 : n == 240 ? 2 : n == 480 ? 3 : n == 1920 ? 4 : n == 2880 ? 5 : 0;
    return toc_flags(toc) | code << 6 | 1 << 9;
}

// the frame duration a descriptor's flags name, in samples at 48 kHz (bits 6 - 8; 960 in reference mode)
inline int flags_frame_size(int32_t f) {
    const int d = (f >> 6) & 7;
    return d == 1 ? 120 : d == 2 ? 240 : d == 3 ? 480 : d == 4 ? 1920 : d == 5 ? 2880 : 960;
}

inline int read_size(const uint8_t *d, int32_t len, int16_t *size) {
    if (len < 1) { *size = -1; return -1; }
    if (d[0] < 252) { *size = d[0]; return 1; }
    if (len < 2) { *size = -1; return -1; }
    *size = (int16_t)(4 * d[1] + d[0]);
    return 2;
}

// Returns the number of frames; size[i] and *payload_offset as in the reference.  self_delimited = 0/1.
inline int parse_packet(const uint8_t *data, int32_t len, int self_delimited, uint8_t *out_toc, int16_t size[48],
                        int *payload_offset, int32_t *packet_offset) {
    const uint8_t *data0 = data;
    int32_t pad = 0, last_size;
    int count, cbr = 0, bytes;
    if (size == nullptr || len < 0) return OPUSGPU_BAD_ARG;
    if (len == 0) return OPUSGPU_INVALID_PACKET;
    const int framesize = toc_samples_per_frame(data[0], 48000);
    const uint8_t toc = *data++;
    len--;
    last_size = len;
    switch (toc & 3) {
        case 0: count = 1; break;
        case 1:
            count = 2;
            cbr = 1;
            if (!self_delimited) {
                if (len & 1) return OPUSGPU_INVALID_PACKET;
                last_size = len / 2;
                size[0] = (int16_t)last_size;
            }
            break;
        case 2:
            count = 2;
            bytes = read_size(data, len, size);
            len -= bytes;
            if (size[0] < 0 || size[0] > len) return OPUSGPU_INVALID_PACKET;
            data += bytes;
            last_size = len - size[0];
            break;
        default: {
            if (len < 1) return OPUSGPU_INVALID_PACKET;
            const uint8_t ch = *data++;
            count = ch & 0x3F;
            if (count <= 0 || framesize * (int32_t)count > 5760) return OPUSGPU_INVALID_PACKET;
            len--;
            if (ch & 0x40) {
                int p;
                do {
                    if (len <= 0) return OPUSGPU_INVALID_PACKET;
                    p = *data++;
                    len--;
                    const int tmp = p == 255 ? 254 : p;
                    len -= tmp;
                    pad += tmp;
                } while (p == 255);
            }
            if (len < 0) return OPUSGPU_INVALID_PACKET;
            cbr = !(ch & 0x80);
            if (!cbr) {
                last_size = len;
                for (int i = 0; i < count - 1; i++) {
                    bytes = read_size(data, len, size + i);
                    len -= bytes;
                    if (size[i] < 0 || size[i] > len) return OPUSGPU_INVALID_PACKET;
                    data += bytes;
                    last_size -= bytes + size[i];
                }
                if (last_size < 0) return OPUSGPU_INVALID_PACKET;
            } else if (!self_delimited) {
                last_size = len / count;
                if (last_size * count != len) return OPUSGPU_INVALID_PACKET;
                for (int i = 0; i < count - 1; i++) size[i] = (int16_t)last_size;
            }
        } break;
    }
    if (self_delimited) {
        bytes = read_size(data, len, size + count - 1);
        len -= bytes;
        if (size[count - 1] < 0 || size[count - 1] > len) return OPUSGPU_INVALID_PACKET;
        data += bytes;
        if (cbr) {
            if (size[count - 1] * count > len) return OPUSGPU_INVALID_PACKET;
            for (int i = 0; i < count - 1; i++) size[i] = size[count - 1];
        } else if (bytes + size[count - 1] > last_size)
            return OPUSGPU_INVALID_PACKET;
    } else {
        if (last_size > 1275) return OPUSGPU_INVALID_PACKET;
        size[count - 1] = (int16_t)last_size;
    }
    if (payload_offset) *payload_offset = (int)(data - data0);
    for (int i = 0; i < count; i++) data += size[i];
    if (packet_offset) *packet_offset = pad + (int32_t)(data - data0);
    if (out_toc) *out_toc = toc;
    return count;
}

} // namespace ogh
