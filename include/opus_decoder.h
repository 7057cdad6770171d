// opus_decoder.h -- reference-compatible packet-decode API (C++ linkage, as in the reference's
// src/opus_decoder.h:156-218) implemented on top of the C ABI in opusgpu.h.  Every decode runs on the GPU;
// each decoder object owns one stream record in HBM (the reference keeps ONE global codec state that all
// decoder objects alias -- src/opusfile.cpp:786-787 -- so several decoders are a new capability here).
#pragma once
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>

#define OPUS_OK 0
#define OPUS_BAD_ARG -1
#define OPUS_BUFFER_TOO_SMALL -2
#define OPUS_INTERNAL_ERROR -3
#define OPUS_INVALID_PACKET -4
#define OPUS_UNIMPLEMENTED -5
#define OPUS_INVALID_STATE -6
#define OPUS_ALLOC_FAIL -7
/* celt_decode_with_ec's own refusals (reference src/opus_decoder.h:43-65, enum; src/celt.cpp:2211,2216,2225): a CELT-only or
 * hybrid frame of <= 1 byte comes back from opus_decode / opus_multistream_decode as this value, not as OPUS_BAD_ARG */
#define ERR_OPUS_CELT_BAD_ARG -18

#define OPUS_GET_BANDWIDTH_REQUEST 4009
#define OPUS_RESET_STATE 4028
#define OPUS_GET_SAMPLE_RATE_REQUEST 4029
#define OPUS_GET_FINAL_RANGE_REQUEST 4031 /* reports 0, always, as the reference does: its rangeFinal is never assigned (src/opus_decoder.cpp:58, :375-380) */
#define OPUS_GET_PITCH_REQUEST 4033       /* OPUS_UNIMPLEMENTED after a CELT-only frame and through the multistream ctl; else the SILK decoder's last lag at 48 kHz (:399-407) */
#define OPUS_SET_GAIN_REQUEST 4034
#define OPUS_GET_GAIN_REQUEST 4045 /* sic: the reference's value */
#define OPUS_GET_LAST_PACKET_DURATION_REQUEST 4039
#define OPUS_SET_PHASE_INVERSION_DISABLED_REQUEST 4046
#define OPUS_GET_PHASE_INVERSION_DISABLED_REQUEST 4047
#define OPUS_MULTISTREAM_GET_DECODER_STATE_REQUEST 5122

#define OPUS_BANDWIDTH_NARROWBAND 1101
#define OPUS_BANDWIDTH_MEDIUMBAND 1102
#define OPUS_BANDWIDTH_WIDEBAND 1103
#define OPUS_BANDWIDTH_SUPERWIDEBAND 1104
#define OPUS_BANDWIDTH_FULLBAND 1105
#define MODE_SILK_ONLY 1000
#define MODE_HYBRID 1001
#define MODE_CELT_ONLY 1002

typedef struct OpusDecoder OpusDecoder;

typedef struct OpusMSDecoder { // public layout of the reference (src/opus_decoder.h:156-161); private state follows it
    int nb_channels;
    int nb_streams;
    int nb_coupled_streams;
    unsigned char mapping[256];
} OpusMSDecoder_t;

// ---- packet helpers (host only; reference src/opus_decoder.cpp:460-509, :541-556, :683) ------------------
int opus_packet_parse(uint8_t *data, int32_t len, unsigned char *out_toc, uint8_t *frames[48], int16_t size[48],
                      int *payload_offset);
int opus_packet_parse_impl(uint8_t *data, int32_t len, int self_delimited, unsigned char *out_toc, uint8_t *frames[48],
                           int16_t size[48], int *payload_offset, int32_t *packet_offset); // reference src/opus_decoder.h:202-204
int opus_packet_get_bandwidth(uint8_t *data);
int opus_packet_get_samples_per_frame(uint8_t *data, int32_t Fs);
int opus_packet_get_nb_channels(uint8_t *data);
int opus_packet_get_nb_frames(uint8_t packet[], int32_t len);
int opus_packet_get_nb_samples(uint8_t packet[], int32_t len, int32_t Fs);
int opus_decoder_get_nb_samples(const OpusDecoder *dec, uint8_t packet[], int32_t len); // reference src/opus_decoder.h:175

// ---- single-stream decoder (reference :66-118, :351-457) ---------------------------------------------------
int opus_decoder_get_size(int channels);
int opus_decoder_init(OpusDecoder *st, int32_t Fs, int channels);
int opus_decode(OpusDecoder *st, uint8_t *data, int32_t len, int16_t *pcm, int frame_size);
// (reference src/opus_decoder.h:182: what opus_decode and the multistream wrapper call.  self_delimited != 0 -- the framing of all
// but the last stream of a multistream packet -- answers OPUS_UNIMPLEMENTED here, like a multistream decoder of more than one stream)
int opus_decode_native(OpusDecoder *st, uint8_t *data, int32_t len, int16_t *pcm, int frame_size, int self_delimited,
                       int32_t *packet_offset);
int opus_decoder_ctl(OpusDecoder *st, int request, ...);
void opus_decoder_destroy(OpusDecoder *st);

// ---- multistream wrapper the container layer calls (reference :729-1045) ---------------------------------
int32_t opus_multistream_decoder_get_size(int streams, int coupled_streams);
OpusMSDecoder_t *opus_multistream_decoder_create(int32_t Fs, int channels, int streams, int coupled_streams,
                                                 const uint8_t *mapping, int *error);
int opus_multistream_decoder_init(OpusMSDecoder_t *st, int32_t Fs, int channels, int streams, int coupled_streams,
                                  const uint8_t *mapping);
int opus_multistream_decode(OpusMSDecoder_t *st, uint8_t *data, int32_t len, int16_t *pcm, int frame_size);
// (reference src/opus_decoder.h:153, :205-207)
typedef void (*opus_copy_channel_out_func)(void *dst, int dst_stride, int dst_channel, const int16_t *src, int src_stride,
                                           int frame_size, void *user_data);
int opus_multistream_decode_native(OpusMSDecoder_t *st, uint8_t *data, int32_t len, void *pcm,
                                   opus_copy_channel_out_func copy_channel_out, int frame_size);
int opus_multistream_decoder_ctl_va_list(OpusMSDecoder_t *st, int request, va_list ap);
int opus_multistream_decoder_ctl(OpusMSDecoder_t *st, int request, ...);
void opus_multistream_decoder_destroy(OpusMSDecoder_t *st);
