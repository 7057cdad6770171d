// TEST PROGRAM for the reference-compatible decoder surface (include/opus_decoder.h over libopusgpu.so): what a caller of
// opus_decode / opus_multistream_decode / the ctls / the packet helpers would do, driven by a script file so that the
// Python test can compare every result with the CPU oracle.
//   script:  records  'D' u32 frame_size u32 len bytes   decode (single-stream decoder and multistream wrapper, both)
//                     'N' i32 frame_size                  decode the last packet's bytes with len = -1 (both)
//                     'R'                                 OPUS_RESET_STATE on both
//                     'Q'                                 ctl queries + packet helpers of the last packet
//                     'F'                                 OPUS_GET_FINAL_RANGE of both decoders
//                     'P'                                 OPUS_GET_PITCH of both decoders
//                     'X'                                 the lower-level entry points on the last packet (see below)
//                     'V'                                 the argument checks of the create / init / get_size functions: 19 x i32
//   output:  per 'D': i32 ret_single, i32 ret_ms, then min(ret, frame_size) * 2 int16 of the single-stream PCM if ret > 0
//            per 'Q': 8 x i32;  per 'F': 2 x u32;  per 'P': i32 ret_single, i32 value, i32 ret_ms;  per 'X': 6 x i32
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "opus_decoder.h"

static const int16_t GUARD = 0x7A7A;

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    if (!in || !out) return 2;
    OpusDecoder *st = (OpusDecoder *)malloc((size_t)opus_decoder_get_size(2));
    if (!st || opus_decoder_init(st, 48000, 2) != OPUS_OK) { fprintf(stderr, "opus_decoder_init failed\n"); return 1; }
    int err = 1;
    const uint8_t mapping[2] = {0, 1};
    OpusMSDecoder_t *ms = opus_multistream_decoder_create(48000, 2, 1, 1, mapping, &err);
    if (!ms || err != OPUS_OK) { fprintf(stderr, "opus_multistream_decoder_create failed (%d)\n", err); return 1; }
    std::vector<uint8_t> last;
    int cmd, guards_ok = 1, decodes = 0;
    while ((cmd = fgetc(in)) != EOF) {
        if (cmd == 'D') {
            int32_t fs = 0; // (as it is: zero and negative frame sizes are calls too)
            uint32_t len = 0;
            if (fread(&fs, 4, 1, in) != 1 || fread(&len, 4, 1, in) != 1) return 2;
            std::vector<uint8_t> cur(len);
            if (len && fread(cur.data(), 1, len, in) != len) return 2;
            if (len) last = cur; // (the packet helpers of 'Q' look at the last packet that HAD bytes)
            const size_t room = fs > 0 ? (size_t)fs * 2 : 0, guard = 4096;
            std::vector<int16_t> a(room + guard, GUARD), b(room + guard, GUARD);
            const int32_t ra = opus_decode(st, cur.data(), (int32_t)len, a.data(), (int)fs);
            const int32_t rb = opus_multistream_decode(ms, cur.data(), (int32_t)len, b.data(), (int)fs);
            for (size_t i = room; i < room + guard; i++) guards_ok &= a[i] == GUARD && b[i] == GUARD;
            fwrite(&ra, 4, 1, out);
            fwrite(&rb, 4, 1, out);
            if (ra > 0) {
                const size_t n = (size_t)(ra < (int32_t)fs ? ra : (int32_t)fs) * 2;
                fwrite(a.data(), 2, n, out);
                if (rb != ra || memcmp(a.data(), b.data(), n * 2) != 0) { fprintf(stderr, "single-stream and multistream disagree\n"); return 1; }
            }
            decodes++;
        } else if (cmd == 'N') {
            int32_t fs = 0;
            if (fread(&fs, 4, 1, in) != 1) return 2;
            std::vector<int16_t> a(5760 * 2 + 64, GUARD);
            const int32_t ra = opus_decode(st, last.data(), -1, a.data(), fs), rb = opus_multistream_decode(ms, last.data(), -1, a.data(), fs);
            fwrite(&ra, 4, 1, out);
            fwrite(&rb, 4, 1, out);
        } else if (cmd == 'R') {
            if (opus_decoder_ctl(st, OPUS_RESET_STATE) != OPUS_OK || opus_multistream_decoder_ctl(ms, OPUS_RESET_STATE) != OPUS_OK) return 1;
        } else if (cmd == 'F') {
            uint32_t v[2] = {0xdeadbeefu, 0xdeadbeefu};
            if (opus_decoder_ctl(st, OPUS_GET_FINAL_RANGE_REQUEST, &v[0]) != OPUS_OK || opus_multistream_decoder_ctl(ms, OPUS_GET_FINAL_RANGE_REQUEST, &v[1]) != OPUS_OK) return 1;
            fwrite(v, 4, 2, out);
        } else if (cmd == 'V') {
            int32_t v[19];
            int k = 0;
            auto create = [&](int32_t Fs, int channels, int streams, int coupled, const uint8_t *map) {
                int e = 12345;
                OpusMSDecoder_t *d = opus_multistream_decoder_create(Fs, channels, streams, coupled, map, &e);
                if (d) opus_multistream_decoder_destroy(d);
                return d ? (e == OPUS_OK ? 0 : -9999) : (e == OPUS_OK ? -9998 : e);
            };
            const uint8_t m01[2] = {0, 1}, m02[2] = {0, 2}, m0m[2] = {0, 255}, m0[1] = {0};
            v[k++] = create(48000, 0, 1, 1, m01);    // channels < 1
            v[k++] = create(48000, 256, 1, 1, m01);  // channels > 255
            v[k++] = create(48000, 2, 0, 0, m01);    // streams < 1
            v[k++] = create(48000, 2, 1, 2, m01);    // coupled_streams > streams
            v[k++] = create(48000, 2, 1, -1, m01);   // coupled_streams < 0
            v[k++] = create(48000, 2, 1, 1, m02);    // validate_layout: a channel mapped past the decoded ones
            v[k++] = create(48000, 2, 1, 1, m0m);    // a muted channel is fine
            v[k++] = create(48000, 1, 1, 0, m0);     // one mono stream
            v[k++] = create(44100, 2, 1, 1, m01);    // opus_decoder_init's rate check
            v[k++] = create(24000, 2, 1, 1, m01);    // (this library: 48 kHz only)
            v[k++] = create(48000, 2, 2, 0, m01);    // (this library: one elementary stream)
            v[k++] = opus_multistream_decoder_get_size(0, 0);
            v[k++] = opus_multistream_decoder_get_size(1, 2);
            v[k++] = opus_multistream_decoder_get_size(1, -1);
            v[k++] = opus_multistream_decoder_get_size(1, 1) > 0;
            v[k++] = opus_decoder_get_size(0);
            v[k++] = opus_decoder_get_size(3);
            v[k++] = opus_decoder_get_size(1) > 0 && opus_decoder_get_size(2) > 0;
            OpusDecoder *bad = (OpusDecoder *)malloc((size_t)opus_decoder_get_size(2));
            v[k++] = opus_decoder_init(bad, 48000, 3); // channels
            free(bad);
            fwrite(v, 4, 19, out);
        } else if (cmd == 'X') {
            // opus_packet_parse_impl (not self-delimited: count, payload offset, packet offset), opus_decode_native with self_delimited
            // 1 (unimplemented here) and with frame_size 0 on the last packet, opus_multistream_decode_native through a copy function
            // of the caller's that negates what it copies (on two fresh decoders, so that the two under test keep their state)
            int32_t v[6] = {0, 0, 0, 0, 0, 0};
            if (!last.empty()) {
                unsigned char toc = 0;
                int16_t size[48];
                uint8_t *frames[48];
                int payload_offset = -1;
                int32_t packet_offset = -1;
                v[0] = opus_packet_parse_impl(last.data(), (int32_t)last.size(), 0, &toc, frames, size, &payload_offset, &packet_offset);
                v[1] = payload_offset;
                v[2] = packet_offset;
                std::vector<int16_t> a(5760 * 2 + 64, GUARD);
                int32_t po = -1;
                v[3] = opus_decode_native(st, last.data(), (int32_t)last.size(), a.data(), 960, 1, &po);
                v[4] = opus_decode_native(st, last.data(), (int32_t)last.size(), a.data(), 0, 0, &po);
                // (two FRESH decoders: OPUS_RESET_STATE is the reference's partial reset, Q5 -- a decoder that has decoded the packet once
                // does not decode it the same way again)
                int e2 = 1, e3 = 1;
                OpusMSDecoder_t *ms2 = opus_multistream_decoder_create(48000, 2, 1, 1, mapping, &e2);
                OpusMSDecoder_t *ms3 = opus_multistream_decoder_create(48000, 2, 1, 1, mapping, &e3);
                if (!ms2 || !ms3) return 1;
                auto neg = [](void *dst, int dst_stride, int dst_channel, const int16_t *src, int src_stride, int frame_size, void *) {
                    int16_t *d = (int16_t *)dst;
                    for (int i = 0; i < frame_size; i++) d[i * dst_stride + dst_channel] = src ? (int16_t)~src[i * src_stride] : (int16_t)0;
                };
                std::vector<int16_t> b(5760 * 2 + 64, GUARD), c(5760 * 2 + 64, GUARD);
                const int32_t r1 = opus_multistream_decode_native(ms2, last.data(), (int32_t)last.size(), b.data(), neg, 5760);
                const int32_t r2 = opus_multistream_decode(ms3, last.data(), (int32_t)last.size(), c.data(), 5760);
                opus_multistream_decoder_destroy(ms2);
                opus_multistream_decoder_destroy(ms3);
                v[5] = r1;
                if (r1 != r2) v[5] = -9999;
                for (int i = 0; r1 > 0 && i < 2 * r1; i++)
                    if (b[i] != (int16_t)~c[i]) v[5] = -9998;
            }
            fwrite(v, 4, 6, out);
        } else if (cmd == 'P') {
            int32_t v[3] = {0, -777, 0};
            v[0] = opus_decoder_ctl(st, OPUS_GET_PITCH_REQUEST, &v[1]);
            int32_t unused = -777;
            v[2] = opus_multistream_decoder_ctl(ms, OPUS_GET_PITCH_REQUEST, &unused);
            fwrite(v, 4, 3, out);
        } else if (cmd == 'Q') {
            int32_t v[8] = {0};
            opus_decoder_ctl(st, OPUS_GET_SAMPLE_RATE_REQUEST, &v[0]);
            opus_decoder_ctl(st, OPUS_GET_BANDWIDTH_REQUEST, &v[1]);
            opus_decoder_ctl(st, OPUS_GET_LAST_PACKET_DURATION_REQUEST, &v[2]);
            if (!last.empty()) {
                v[3] = opus_packet_get_nb_frames(last.data(), (int32_t)last.size());
                v[4] = opus_packet_get_nb_samples(last.data(), (int32_t)last.size(), 48000);
                v[5] = opus_packet_get_nb_channels(last.data());
                v[6] = opus_packet_get_bandwidth(last.data());
                v[7] = opus_packet_get_samples_per_frame(last.data(), 48000);
            }
            fwrite(v, 4, 8, out);
        } else
            return 2;
    }
    printf("decodes=%d guards=%s\n", decodes, guards_ok ? "intact" : "OVERRUN");
    opus_multistream_decoder_destroy(ms);
    opus_decoder_destroy(st);
    fclose(out);
    return guards_ok ? 0 : 1;
}
