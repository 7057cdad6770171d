// TEST-ONLY: the whole host stack above the C ABI -- include/opusfile.h entry points, csrc/og_compat.cpp (opus_decoder.h
// surface, channel mapping, ctl bookkeeping) and csrc/og_container.hpp -- linked against a TEST DOUBLE of the C ABI whose
// decode is the CPU oracle, so that it can be fuzzed with crafted files under AddressSanitizer / UBSan on a machine
// without a GPU (tools/fuzz_stack_asan.py).  The double mirrors opusgpu_decode_packets' framing and size checks
// (og_api.hip); it is never linked into the product.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "og_packet.hpp"
#include "../../include/opusgpu.h"
#include "../../include/opusfile.h"
#include "oc_opus.h"

struct opusgpu_ctx {
    int channels = 0;
    std::vector<oc_decoder *> dec;
};
extern "C" {
int opusgpu_ctx_create(int, opusgpu_ctx **out) { *out = new opusgpu_ctx; return OPUSGPU_OK; }
void opusgpu_ctx_destroy(opusgpu_ctx *c) {
    if (!c) return;
    for (auto d : c->dec) oc_decoder_destroy(d);
    delete c;
}
int opusgpu_streams_alloc(opusgpu_ctx *c, int n, int channels) {
    if (!c || n <= 0 || (channels != 1 && channels != 2)) return OPUSGPU_BAD_ARG;
    for (auto d : c->dec) oc_decoder_destroy(d);
    c->dec.clear();
    c->channels = channels;
    for (int i = 0; i < n; i++) {
        oc_decoder *d = oc_decoder_create(channels);
        oc_decoder_init(d, channels);
        c->dec.push_back(d);
    }
    return OPUSGPU_OK;
}
int opusgpu_streams_reset(opusgpu_ctx *c, int first, int count, int full) {
    if (!c || first < 0 || count < 0 || first + count > (int)c->dec.size()) return OPUSGPU_BAD_ARG;
    for (int i = first; i < first + count; i++) {
        if (full) oc_decoder_init(c->dec[i], c->channels);
        else oc_decoder_reset(c->dec[i]);
    }
    return OPUSGPU_OK;
}
int opusgpu_stream_state_get(opusgpu_ctx *, int, void *, size_t) { return OPUSGPU_UNIMPLEMENTED; }
int opusgpu_decode_packets(opusgpu_ctx *c, int n, const int32_t *ids, const uint8_t *const *packets, const int32_t *lens,
                           int16_t *pcm, int frame_capacity, int32_t *result) {
    if (!c || n < 0 || c->dec.empty()) return OPUSGPU_BAD_ARG;
    if (!ids || !packets || !lens || !pcm || !result || frame_capacity <= 0) return OPUSGPU_BAD_ARG;
    std::vector<int16_t> tmp((size_t)48 * 960 * 2);
    const size_t block = (size_t)frame_capacity * 960 * c->channels;
    for (int i = 0; i < n; i++) {
        result[i] = 0;
        if (ids[i] < 0 || ids[i] >= (int)c->dec.size() || lens[i] < 0) { result[i] = OPUSGPU_BAD_ARG; continue; }
        if (!packets[i] || lens[i] == 0) { // an empty packet: the reference's own branch for it (oc_decode, src/opus_decoder.cpp:290-308)
            const int r = oc_decode(c->dec[ids[i]], nullptr, 0, tmp.data(), 960 * frame_capacity);
            result[i] = r;
            if (r > 0) memcpy(pcm + (size_t)i * block, tmp.data(), sizeof(int16_t) * (size_t)r * c->channels);
            continue;
        }
        int16_t size[48];
        uint8_t toc;
        const int count = ogh::parse_packet(packets[i], lens[i], 0, &toc, size, nullptr, nullptr);
        if (count < 0) { result[i] = count; continue; }
        const int pfs = ogh::toc_samples_per_frame(packets[i][0], 48000);
        if ((int64_t)count * pfs > (int64_t)frame_capacity * 960 || count > frame_capacity) { result[i] = OPUSGPU_BUFFER_TOO_SMALL; continue; }
        const int r = oc_decode(c->dec[ids[i]], packets[i], lens[i], tmp.data(), 960 * frame_capacity);
        result[i] = r;
        if (r > 0) memcpy(pcm + (size_t)i * block, tmp.data(), sizeof(int16_t) * (size_t)r * c->channels);
    }
    return OPUSGPU_OK;
}
}

// ---- the application side: SD_read over a memory buffer -------------------------------------------------------------
static const unsigned char *g_data;
static size_t g_len, g_pos;
static int g_eof_code = -1;
int SD_read(unsigned char *buff, int nbytes) {
    const size_t left = g_len - g_pos;
    if (left == 0) return g_eof_code;
    const size_t k = left < (size_t)nbytes ? left : (size_t)nbytes;
    memcpy(buff, g_data + g_pos, k);
    g_pos += k;
    return (int)k;
}
extern "C" {
int st_open(const unsigned char *data, size_t len, int eof_code) {
    g_data = data; g_len = len; g_pos = 0; g_eof_code = eof_code;
    return opus_init_decoder() != nullptr;
}
int st_read(int16_t *pcm, int buf_size) { return op_read_stereo(pcm, buf_size); }
void st_close(void) { opus_close_decoder(); }
}
