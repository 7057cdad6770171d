// TEST-ONLY: the host container reader (csrc/og_container.hpp -- no codec arithmetic in it) driven with the
// CPU ORACLE as its decode callback, so the Ogg/opusfile bookkeeping can be checked on a machine without a
// GPU (survey KAT 3).  The shipped library wires the same reader to the GPU decoder instead (og_compat.cpp).
#include <stdlib.h>
#include <string.h>
#include "og_container.hpp"
#include "oc_opus.h"

static const unsigned char *g_data;
static size_t g_len, g_pos;
static int g_eof_code = -1; // the reference player's SD_read returns -1 when nothing more can be read (main.cpp:267)
static oc_decoder *g_dec;
static ogc::OpusFile *g_of;

static int mem_read(unsigned char *buf, int n) {
    size_t left = g_len - g_pos;
    if (left == 0) return g_eof_code;
    size_t k = left < (size_t)n ? left : (size_t)n;
    memcpy(buf, g_data + g_pos, k);
    g_pos += k;
    return (int)k;
}
static int oracle_decode(void *, const uint8_t *pkt, int32_t len, int16_t *pcm, int frame_size) {
#ifdef CT_STUB_DECODE // fuzzing the container logic alone: "decode" = the packet's duration, silence
    const int d = ogc::packet_duration(pkt, len);
    if (d <= 0) return -4;
    if (d > frame_size) return -2;
    memset(pcm, 0, sizeof(int16_t) * (size_t)d * g_of->head().channel_count);
    return d;
#else
    if (!g_dec) g_dec = oc_decoder_create(g_of->head().channel_count);
    return oc_decode(g_dec, pkt, len, pcm, frame_size);
#endif
}

extern "C" {
int ct_open(const unsigned char *data, size_t len, int eof_code) {
    g_data = data; g_len = len; g_pos = 0; g_eof_code = eof_code;
    if (g_dec) { oc_decoder_destroy(g_dec); g_dec = nullptr; }
    delete g_of;
    g_of = new ogc::OpusFile(mem_read, oracle_decode, nullptr);
    return g_of->open();
}
int ct_read_stereo(int16_t *pcm, int buf_size) { return g_of->read_stereo(pcm, buf_size); }
int ct_channels(void) { return g_of->head().channel_count; }
int ct_pre_skip(void) { return (int)g_of->head().pre_skip; }
uint32_t ct_crc(const unsigned char *p, size_t n) { return ogc::crc_update(0, p, n); }
}
