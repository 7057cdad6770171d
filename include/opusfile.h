// opusfile.h -- the three touch-points of the reference player's main.cpp (src/main.cpp:9, :264, :273, :308):
//     #include "opusfile.h";  int SD_read(unsigned char*, int) [defined by the application];
//     OggOpusFile_t* opus_init_decoder();  int op_read_stereo(int16_t* pcm, int buf_size);
// Same names, argument meaning and return values as the reference's src/opusfile.h:19, :144, :156 (C++ linkage).
// Ogg demux and granule bookkeeping run on the host; every packet is decoded on the GPU through opusgpu.h.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "opus_decoder.h"

extern __attribute__((weak)) int SD_read(unsigned char *buff, int nbytes); // >0 bytes read, 0 only at clean EOF, <0 error

#define OP_FALSE (-1)
#define OP_EOF (-2)
#define OP_HOLE (-3)
#define OP_EREAD (-128)
#define OP_EFAULT (-129)
#define OP_EIMPL (-130)
#define OP_EINVAL (-131)
#define OP_ENOTFORMAT (-132)
#define OP_EBADHEADER (-133)
#define OP_EVERSION (-134)
#define OP_EBADPACKET (-136)
#define OP_EBADLINK (-137)
#define OP_EBADTIMESTAMP (-139)

typedef struct OpusHead {
    int version;
    int channel_count;
    unsigned pre_skip;
    uint32_t input_sample_rate;
    int output_gain;
    int mapping_family;
    int stream_count;
    int coupled_count;
    uint8_t mapping[8];
} OpusHead_t;

typedef struct OggOpusFile OggOpusFile_t; // opaque here (the reference exposes its fields; main.cpp ignores the value)

int opus_head_parse(OpusHead_t *_head, uint8_t *_data, size_t _len);
OggOpusFile_t *opus_init_decoder();              // NULL on failure (reference src/opusfile.cpp:784)
int op_read_stereo(int16_t *_pcm, int _buf_size); // samples per channel (<= _buf_size/2), 0 at clean EOF, <0 OP_* error
void opus_close_decoder();                        // additive: releases the GPU context of the implicit player
