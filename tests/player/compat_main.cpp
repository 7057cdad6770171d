// TEST PROGRAM for the reference-compatible decoder surface (include/opus_decoder.h over libopusgpu.so): what a caller of
// opus_decode / opus_multistream_decode / the ctls / the packet helpers would do, driven by a script file so that the
// Python test can compare every result with the CPU oracle.
//   script:  records  'D' u32 frame_size u32 len bytes   decode (single-stream decoder and multistream wrapper, both)
//                     'N' i32 frame_size                  decode the last packet's bytes with len = -1 (both)
//                     'R'                                 OPUS_RESET_STATE on both
//                     'Q'                                 ctl queries + packet helpers of the last packet
//                     'F'                                 OPUS_GET_FINAL_RANGE of both decoders
//                     'P'                                 OPUS_GET_PITCH of both decoders
//   output:  per 'D': i32 ret_single, i32 ret_ms, then min(ret, frame_size) * 2 int16 of the single-stream PCM if ret > 0
//            per 'Q': 8 x i32;  per 'F': 2 x u32;  per 'P': i32 ret_single, i32 value, i32 ret_ms
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
