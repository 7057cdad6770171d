// Output stage of the player: decoded int16 PCM -> 32-bit I2S words, as src/main.cpp does between the
// decoder and the peripheral: playChunk (:148-224) picks the samples of an output frame by bit depth, channel count and
// force-mono; playSample (:226-242) expands 8-bit samples, halves for headroom and applies Gain (:137-146), which packs
// (right << 16) | (left & 0xffff).  One output word depends on one or two input words only, so the loops of the reference
// become a function of the word index.  Shared by the kernel (og_api.hip) and the host emulation (tests/emul).
#pragma once
#include "og_common.hpp"

namespace og {

struct OutputCfg { // mirrors opusgpu_output_cfg
    u8 volume;     // m_vol: 64 = unity (main.cpp:39)
    u8 force_mono; // m_f_forceMono
    u8 bits;       // 16 or 8 (setBitsPerSample :119-123)
    u8 channels;   // 1 or 2 (setChannels :128-132)
};

// I2S words a block of `valid` samples (m_validSamples) makes: 8-bit mono plays both bytes of every word (:153-169);
// playChunk emits nothing for other bit depths (:222) or channel counts
OG_DEV int output_words(OutputCfg c, int valid) {
    if (valid <= 0 || (c.bits != 8 && c.bits != 16) || (c.channels != 1 && c.channels != 2)) return 0;
    return c.bits == 8 && c.channels == 1 ? 2 * valid : valid;
}

// playSample + Gain on one frame whose two samples are already chosen (:231-242, :142-145)
OG_DEV u32 output_pack(i32 l, i32 r, bool bits8, i32 vol) {
    if (bits8) {
        l = ((l & 0xff) - 128) * 256;
        r = ((r & 0xff) - 128) * 256;
    }
    l = (i32)(i16)l >> 1;
    r = (i32)(i16)r >> 1;
    const i32 vl = (l * vol) >> 6, vr = (r * vol) >> 6;
    return (u32)vr << 16 | ((u32)vl & 0xffffu);
}

// word w (< output_words) of the block whose m_outBuff is `blk`
OG_DEV u32 output_word(const i16 *blk, int w, OutputCfg c) {
    i32 l, r;
    if (c.bits == 16) {
        if (c.channels == 1) {
            l = r = blk[w]; // :196-197
        } else {
            l = blk[2 * w]; // :209-210
            r = blk[2 * w + 1];
            if (c.force_mono) l = r = (i32)(i16)((l + r) / 2); // :213 (C division: towards zero)
        }
    } else {
        const u32 word = (u16)blk[c.channels == 1 ? w >> 1 : w];
        const i32 x = (i32)(word & 0xffu), y = (i32)(word >> 8); // :154-155, :172-173
        if (c.channels == 1) {
            l = r = (w & 1) ? y : x; // :156-165: the low byte first, then the high byte
        } else if (c.force_mono) {
            l = r = (i32)(u8)((x + y) / 2); // :179
        } else {
            l = x; // :175-176
            r = y;
        }
    }
    return output_pack(l, r, c.bits == 8, (i32)c.volume);
}

} // namespace og
