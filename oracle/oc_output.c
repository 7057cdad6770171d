/* CPU ORACLE -- TEST INFRASTRUCTURE, not a product path (see oc_opus.h).
 * Output stage of the player: what src/main.cpp does to a decoded block between the decoder and the I2S
 * peripheral -- playChunk (:148-259) walks m_outBuff by bit depth / channel count / force-mono, playSample (:226-256)
 * expands 8-bit samples, halves the input for headroom and applies Gain (:137-146), which packs one 32-bit I2S word per
 * output frame.  i2s_channel_write becomes a store into `i2s`.  Restated in the reference's own loop shape.
 * Not applied by the reference and therefore not here: the OpusHead output gain (op_update_gain is commented out,
 * src/opusfile.cpp:704).  Parity of this file is pinned by hand-derived known answers only (tests/test_output_stage.py):
 * the reference's main.cpp needs <Arduino.h> and has no test vectors of its own. */
#include <stdint.h>

enum { LEFTCHANNEL = 0, RIGHTCHANNEL = 1 };

/* Gain, main.cpp:137-146 (l = r = 0 there).  v << 16 on a negative or > 16-bit v is what the target's compiler makes of
 * it: a shift of the two's-complement pattern. */
static uint32_t oc_gain(const int16_t s[2], int vol) {
    int32_t v[2];
    v[LEFTCHANNEL] = (s[LEFTCHANNEL] * vol) >> 6;
    v[RIGHTCHANNEL] = (s[RIGHTCHANNEL] * vol) >> 6;
    return ((uint32_t)v[RIGHTCHANNEL] << 16) | ((uint32_t)v[LEFTCHANNEL] & 0xffffu);
}

/* playSample, main.cpp:226-256 */
static void oc_play_sample(int16_t sample[2], int bits, int vol, uint32_t *i2s, long *n) {
    if (bits == 8) { /* unsigned 8 bits -> signed 16 bits (:231-234) */
        sample[LEFTCHANNEL] = (int16_t)(((sample[LEFTCHANNEL] & 0xff) - 128) * 256);
        sample[RIGHTCHANNEL] = (int16_t)(((sample[RIGHTCHANNEL] & 0xff) - 128) * 256);
    }
    sample[LEFTCHANNEL] = sample[LEFTCHANNEL] >> 1; /* :236-237 */
    sample[RIGHTCHANNEL] = sample[RIGHTCHANNEL] >> 1;
    i2s[(*n)++] = oc_gain(sample, vol); /* :239-242 */
}

/* playChunk, main.cpp:148-224.  out_buff: m_outBuff, valid_samples: m_validSamples.  Returns the number of I2S words
 * written, -1 for a bit depth that is neither 8 nor 16 (:222-223). */
long oc_output_stage(const int16_t *out_buff, int valid_samples, int bits, int channels, int force_mono, int vol, uint32_t *i2s) {
    int16_t sample[2];
    long n = 0;
    int cur = 0;
    if (bits == 8) {
        if (channels == 1) {
            while (valid_samples) {
                uint8_t x = out_buff[cur] & 0x00FF;
                uint8_t y = (out_buff[cur] & 0xFF00) >> 8;
                sample[LEFTCHANNEL] = x;
                sample[RIGHTCHANNEL] = x;
                oc_play_sample(sample, bits, vol, i2s, &n);
                sample[LEFTCHANNEL] = y;
                sample[RIGHTCHANNEL] = y;
                oc_play_sample(sample, bits, vol, i2s, &n);
                valid_samples--;
                cur++;
            }
        }
        if (channels == 2) {
            while (valid_samples) {
                uint8_t x = out_buff[cur] & 0x00FF;
                uint8_t y = (out_buff[cur] & 0xFF00) >> 8;
                if (!force_mono) {
                    sample[LEFTCHANNEL] = x;
                    sample[RIGHTCHANNEL] = y;
                } else {
                    uint8_t xy = (x + y) / 2;
                    sample[LEFTCHANNEL] = xy;
                    sample[RIGHTCHANNEL] = xy;
                }
                oc_play_sample(sample, bits, vol, i2s, &n);
                valid_samples--;
                cur++;
            }
        }
        return n;
    }
    if (bits == 16) {
        if (channels == 1) {
            while (valid_samples) {
                sample[LEFTCHANNEL] = out_buff[cur];
                sample[RIGHTCHANNEL] = out_buff[cur];
                oc_play_sample(sample, bits, vol, i2s, &n);
                valid_samples--;
                cur++;
            }
        }
        if (channels == 2) {
            while (valid_samples) {
                if (!force_mono) {
                    sample[LEFTCHANNEL] = out_buff[cur * 2];
                    sample[RIGHTCHANNEL] = out_buff[cur * 2 + 1];
                } else {
                    int16_t xy = (int16_t)((out_buff[cur * 2] + out_buff[cur * 2 + 1]) / 2);
                    sample[LEFTCHANNEL] = xy;
                    sample[RIGHTCHANNEL] = xy;
                }
                oc_play_sample(sample, bits, vol, i2s, &n);
                valid_samples--;
                cur++;
            }
        }
        return n;
    }
    return -1;
}
