// og_state.hpp -- HBM-resident per-stream decoder state, per-step frame descriptors and the
// per-wave LDS working set.
//
// HBM layout: one StreamState record per stream, array-of-structures.  With one frame per
// wavefront the 64 lanes of a wave read/write consecutive words of ONE record, so AoS is the
// coalesced layout here.  The CELT synthesis history is a ring (2048 samples per channel) so a
// frame appends its 960 new samples instead of shifting 1084 (the reference memmoves 17 KB per
// frame, src/celt.cpp:2349); only what the pitch comb filter reaches back to (<= 1024 samples,
// src/celt.cpp:833, :2262) is ever read again.
#pragma once
#include "og_common.hpp"

namespace og {

constexpr int NBANDS = 21;     // src/celt.cpp:632
constexpr int OVERLAP = 120;   // src/celt.cpp:631
constexpr int RING = 2048;
constexpr int RING_MASK = RING - 1;

struct CeltState {             // replaces CELTDecoder_t + trailing arrays (src/celt.h:150-171, celt.cpp:2202)
    alignas(128) i32 ring[2][RING]; // comb-filtered synthesis output history (the live part of _decode_mem); on a line boundary of the
                               // memory system: k_celt_post and the comb filter read it in 128-byte pieces (misaligned, every piece
                               // was two lines: 12.8 KB fetched per frame for 7.7 KB read)
    i32 tail[2][64];           // IMDCT overlap tail out_syn[N..N+60) carried to the next frame
    i32 deemph[2];             // preemph_memD
    u32 rng;
    i32 ring_pos;              // ring index of out_syn[0] of the NEXT frame
    i32 pf_period, pf_period_old, pf_gain, pf_gain_old, pf_tapset, pf_tapset_old;
    i32 error;
    i32 reserved;
    i16 bandE[2 * NBANDS], logE1[2 * NBANDS], logE2[2 * NBANDS]; // oldBandE / oldLogE / oldLogE2
    i16 pad[2];
};

// SILK per-channel state: the live subset of silk_decoder_state_t (src/silk.h:705-741) plus the
// channel's resampler (src/silk.h:654-670).  PLC/CNG state and exc_Q14 are not kept: with
// lostFlag == 0 they never reach the PCM; for 20 ms frames outBuf is just the last
// frame's output (ltp_mem_length == frame_length).
struct SilkChannel {
    i32 prev_gain_Q16;
    i32 sLPC_Q14_buf[16];
    i32 lagPrev;
    i32 LastGainIndex;          // int8 in the reference
    i32 fs_kHz;                 // 0 after init -> next frame re-derives everything (silk_decoder_set_fs)
    i32 first_frame_after_reset;
    i32 ec_prevSignalType, ec_prevLagIndex;
    i32 prevSignalType;
    i32 nFramesDecoded;
    // resampler
    i32 rs_sIIR[6];
    i32 rs_invRatio_Q16, rs_inputDelay, rs_fs_in_kHz;
    i16 rs_sFIR[8];
    i16 rs_delayBuf[16];
    i16 prevNLSF_Q15[16];
    i16 outBuf[320];
};

struct SilkState {              // silk_decoder_t (src/silk.h:758-764) incl. stereo state (:672-676)
    SilkChannel ch[2];
    i32 pred_prev_Q13[2];       // int16 in the reference
    i16 sMid[2], sSide[2];
    i32 nChannelsAPI, nChannelsInternal, prev_decode_only_middle;
    i32 reserved;
};

// RFC mode only (loss concealment): what the reference keeps for a path it never takes (lostFlag == 0, Q8) --
// a channel's last excitation, silk_PLC_struct and silk_CNG_struct (src/silk.h:680-703, :716, :738), and CELT's noise floor
// and loss counter (src/celt.h:161, src/celt.cpp:2206).  Reference-mode kernels never touch these bytes: they sit at the END
// of the record, and the record's hot part keeps its layout.
struct SilkLossChannel {
    i32 exc_Q14[320];
    i32 cng_exc_buf_Q14[320];
    i32 cng_synth_state[16];
    i32 lossCnt;
    i32 plc_pitchL_Q8, plc_last_frame_lost, plc_rand_seed, plc_randScale_Q14, plc_conc_energy, plc_conc_energy_shift;
    i32 plc_prevLTP_scale_Q14, plc_prevGain_Q16[2], plc_fs_kHz, plc_nb_subfr, plc_subfr_length;
    i32 cng_smth_Gain_Q16, cng_rand_seed, cng_fs_kHz;
    i16 plc_LTPCoef_Q14[6];    // 5 taps
    i16 plc_prevLPC_Q12[16];
    i16 cng_smth_NLSF_Q15[16];
    i32 reserved;
};
struct LossState {
    SilkLossChannel silk[2];
    i16 backgroundLogE[2 * NBANDS];
    i16 pad[2];
    i32 celt_loss_count;
    i32 celt_end_band;         // last band by the bandwidth of the last decoded frame (what a concealment fills up to)
    i32 prev_redundancy;       // the last frame ended with a redundant CELT frame: a SILK -> CELT transition is under way
    i32 plc_pitch;             // pitch-based CELT concealment (og_plc.hpp): the period found at the first lost frame,
    i16 plc_lpc[2][24];        // and each channel's LPC filter of that frame, kept for the losses that follow
};

struct StreamState {
    i32 channels;              // decoder channels (OpusHead)
    i32 prev_mode;             // OpusDecoder.prev_mode (src/opus_decoder.cpp:54)
    i32 frames_decoded;
    u32 range_final;
    CeltState celt;
    SilkState silk;
    LossState loss;            // RFC mode only
};

// One frame of work for one stream in one decode step (host-built, SoA-free: 16 bytes).
struct FrameDesc {
    i32 stream;                // index into the StreamState array
    i32 offset;                // byte offset of the frame payload (after TOC/size bytes) in the arena
    i32 len;                   // payload bytes (<= 1275)
    i32 flags;                 // bits 0-1: 0 SILK, 1 hybrid, 2 CELT; bits 2-4: bandwidth - 1101; bit 5: stereo;
                               // bits 6-8: frame duration (0: 20 ms, 1: 2.5, 2: 5, 3: 10, 4: 40, 5: 60); bit 9: RFC mode;
                               // bit 10 (RFC mode): decode the frame's forward error correction data (the frame before it)
                               // bit 11 (reference mode, an EMPTY frame of a stream that has had no packet since its reset): the
                               // reference's decoder is in mode 0 there, which opus_decode_frame runs like hybrid (SILK at 16 kHz,
                               // then CELT's refusal of the empty frame) but leaves as prev_mode (src/opus_decoder.cpp:156,175,276)
};
OG_DEV int desc_mode(i32 f) { return MODE_SILK + (f & 3); }
OG_DEV int desc_bandwidth(i32 f) { return BW_NB + ((f >> 2) & 7); }
OG_DEV int desc_channels(i32 f) { return (f & 32) ? 2 : 1; }
// a SILK-only frame at 8 kHz: the frames of k_silk_synth_nb (og_silk_nb.hip)
OG_DEV bool desc_silk_nb_only(i32 f) { return (f & 3) == 0 && ((f >> 2) & 7) == 0; }
// what the frame leaves as the stream's prev_mode: its mode -- or 0 for the mode-0 frame (bit 11: coded as hybrid, see above)
OG_DEV int desc_mode_after(i32 f) { return ((f >> 11) & 1) && (f & 3) == 1 ? 0 : desc_mode(f); }
// RFC mode (opt-in, opusgpu_set_mode): the frame decodes at the duration its TOC names; reference mode: always 960 (Q6)
OG_DEV int desc_rfc(i32 f) { return (f >> 9) & 1; }
OG_DEV int desc_fec(i32 f) { return (f >> 10) & 1; }
OG_DEV int desc_frame_size(i32 f) {
    const int d = (f >> 6) & 7;
    return d == 1 ? 120 : d == 2 ? 240 : d == 3 ? 480 : d == 4 ? 1920 : d == 5 ? 2880 : 960;
}

// ---- per-wave LDS working set ---------------------------------------------------------------------
#ifndef OG_RECON_TIGHT
// i16 vector arena: [X (2 x 960)] [norm (2 x 624)] [iy (192)] [tmp (192)]
constexpr int V_X = 0;
constexpr int V_NORM = 1920;
constexpr int V_IY = V_NORM + 1248;
constexpr int V_TMP = V_IY + 192;
constexpr int V_TOTAL = V_TMP + 192;
constexpr int V_SYN = V_NORM;  // the i32 synthesis buffer of one channel starts here
constexpr int SYN_LEN = 1088;  // 960 + 120 (+8 pad)

// Two phases share the bytes behind X: while the bands are decoded they hold the folding history, the pulse /
// scratch rows and the packet (or, on the split path, the per-leaf collapse masks); during synthesis they hold the
// i32 IMDCT / output buffer of ONE channel at a time (syn_buf(), og_celt.hpp).  After a channel's synthesis its PCM
// goes to a half of the (by then dead) X region as a plane of 960 samples.
struct FrameLds {
    alignas(16) i16 v[V_TOTAL];
    u8 pkt[1344];              // packet bytes (<= 1275); the split path keeps its jobs' collapse masks here
#ifndef OG_NO_SPLIT_LDS        // (og_rfc.hip: a translation unit that never instantiates the split path's reconstruction)
    u32 win[64];               // split path: window of the parse record's word stream
#endif
    i32 pulses[NBANDS], fine_quant[NBANDS], fine_prio[NBANDS], tf_res[NBANDS], offsets[NBANDS];
    i32 bits1[NBANDS], bits2[NBANDS];
    i16 bandE[2 * NBANDS], logE1[2 * NBANDS], logE2[2 * NBANDS];
    i16 dn_g[2 * NBANDS], dn_shift[2 * NBANDS];
    u8 cmask[2 * NBANDS];
    u8 bin2band[120];
    OG_MEMBER i16 *bandE_row() { return bandE; }
    OG_MEMBER i16 *logE1_row() { return logE1; }
    OG_MEMBER i16 *logE2_row() { return logE2; }
    OG_MEMBER u8 *cmask_row() { return cmask; }
    OG_MEMBER i32 *pulses_row() { return pulses; }
    OG_MEMBER u32 *job_mask_row() { return reinterpret_cast<u32 *>(&pkt[0]); } // 2 * NBANDS words
#ifndef OG_NO_SPLIT_LDS
    OG_MEMBER u32 *word_window() { return win; }
    OG_MEMBER u8 *rot_marker() { return reinterpret_cast<u8 *>(win); } // (64 bytes of the leaf pass: the window is the band loop's)
#else
    OG_MEMBER u32 *word_window() { return reinterpret_cast<u32 *>(pkt); } // (never called there)
    OG_MEMBER u8 *rot_marker() { return pkt; }
#endif
    OG_MEMBER i16 *dn_g_row() { return dn_g; }
    OG_MEMBER i16 *dn_shift_row() { return dn_shift; }
    OG_MEMBER u8 *bin2band_row() { return bin2band; }
};
static_assert((V_TOTAL - V_NORM) * 2 + 1344 >= SYN_LEN * 4, "the synthesis buffer overlays norm | iy | tmp | pkt");
#else
// The working set of the reconstruction kernel of 20 ms frames (og_recon.hip): 6,312 bytes = FIVE 1280-byte granules of LDS, so
// that SIX waves fit a SIMD (its 80 registers allow as many) -- and, what counts in pipelined steps, thirteen instead of eleven
// such workgroups fit a CU next to two of the parse kernel's.  LDS is handed out in granules of 1280 bytes (measured with a
// residency census: 9968 B -> 16 workgroups per CU, 8080 B -> 18, 6000 B -> 25).  Measured by padding: the 10 KB layout above went
// with 1 / (waves per SIMD) (2.69 ms at three, 2.09 at four); round 4's 7,552 bytes (six granules) padded by one granule cost the
// CELT step 7 % (round 5).  What makes it fit:
//   * the folding history is made on demand (phase-major band loop), so the bytes behind X hold, one after the other, the
//     PVQ table (leaf pass), the band loop's tables + two 200-entry scratch rows, the synthesis buffer;
//   * the synthesis buffer of a channel starts INSIDE X, over the second channel's spectrum: that channel is synthesised
//     first, its spectrum read into registers before the buffer is written (og_celt.hpp), and the first channel's
//     spectrum is still in place when its turn comes;
//   * the mode codes 800 of a channel's 960 coefficients (eband[21] = 100 bins of 8): the top 160 entries of each channel's
//     spectrum are never written by a leaf or a band and only ever read with a gain of zero (the IMDCT's front), so they are
//     320 bytes of storage each -- the PVQ table's short rows 9 - 14 and the leaf pass's rotation marker while the leaves are
//     decoded, the synthesis' per-band gains (which must outlive the first channel's transform) afterwards;
//   * the leaves' collapse masks are ORed into one word per JOB (168 bytes) instead of kept per leaf (832);
//   * the synthesis' per-bin gain table (480 bytes) lies INSIDE the buffer, in its last 120 words: a channel's transform has
//     read every gain (and every coefficient) before it stores its first word (og_celt.hpp);
//   * arrays only the entropy-decoding half needs are gone (zero-length here: the code that names them is never run from
//     this layout, see og_recon.hip).
constexpr int V_X = 0;
constexpr int X_TOP0 = 800;             // 160 entries no band reaches: first channel (see above)
constexpr int X_TOP1 = 960 + 800;       // ... second channel
constexpr int V_NORM = 1920;            // the band loop's tables (PmLds) -- or the PVQ table's rows 4 - 8 during the leaf pass
constexpr int V_IY = V_NORM + 576;      // scratch row (folding source), 200 entries
constexpr int V_TMP = V_IY + 200;       // scratch row (Hadamard), 200 entries
// Once the stereo merges are done the two scratch rows hold the frame's small arrays: the bands' collapse masks, and --
// staged from the record and the stream state only now -- pulses, band energies and the two energy histories, which
// anti-collapse and the start of the synthesis read.  (The synthesis buffer later runs over them: they are dead by then.)
constexpr int V_LATE = V_IY;
constexpr int V_WIN = V_TMP + 200;      // window of the record's word stream (64 x u32; fill jobs)
constexpr int V_JOBM = V_NORM + 1152;   // the jobs' collapse masks (2 * NBANDS x u32), from the leaf pass to the band loop: behind
                                        // the PVQ table's rows 4 - 8 + row bases (1134 + 16 entries) and behind the window
constexpr int V_TOTAL = V_JOBM + 4 * NBANDS;
constexpr int V_SYN = 960;              // the synthesis buffer: second channel's spectrum + what lies behind X
constexpr int SYN_LEN = 1088;
struct FrameLds {
    alignas(16) i16 v[V_TOTAL];
    // (named by code that this layout never runs)
    u8 pkt[0];
    i32 fine_quant[0], fine_prio[0], tf_res[0], offsets[0], bits1[0], bits2[0];
    OG_MEMBER u8 *cmask_row() { return reinterpret_cast<u8 *>(&v[V_LATE]); }                 // 42 bytes
    OG_MEMBER i32 *pulses_row() { return reinterpret_cast<i32 *>(&v[V_LATE + 24]); }          // 21 words
    OG_MEMBER i16 *bandE_row() { return &v[V_LATE + 24 + 2 * NBANDS]; }
    OG_MEMBER i16 *logE1_row() { return &v[V_LATE + 24 + 4 * NBANDS]; }
    OG_MEMBER i16 *logE2_row() { return &v[V_LATE + 24 + 6 * NBANDS]; }
    OG_MEMBER u32 *job_mask_row() { return reinterpret_cast<u32 *>(&v[V_JOBM]); }
    OG_MEMBER u32 *word_window() { return reinterpret_cast<u32 *>(&v[V_WIN]); }
    OG_MEMBER u8 *rot_marker() { return reinterpret_cast<u8 *>(&v[X_TOP1 + 104]); }  // 64 bytes behind the PVQ table's rows 12 - 14 (leaf pass)
    OG_MEMBER i16 *dn_g_row() { return &v[X_TOP0]; }                                  // synthesis only: 2 * NBANDS each
    OG_MEMBER i16 *dn_shift_row() { return &v[X_TOP0 + 2 * NBANDS]; }
    OG_MEMBER u8 *bin2band_row() { return reinterpret_cast<u8 *>(&v[X_TOP0 + 4 * NBANDS]); } // 120 bytes
};
static_assert(V_SYN * 2 + SYN_LEN * 4 <= V_TOTAL * 2, "the synthesis buffer fits");
static_assert(X_TOP0 + 4 * NBANDS + 60 <= 960, "the synthesis' per-band gains and the bin -> band table fit the first channel's unused top");
static_assert(V_LATE + 24 + 8 * NBANDS <= V_WIN && (V_LATE + 24) % 2 == 0, "the late-staged arrays fit the scratch rows");
static_assert(V_NORM % 8 == 0 && V_IY % 8 == 0 && V_WIN % 2 == 0 && V_WIN + 128 <= V_JOBM && V_JOBM % 2 == 0 && X_TOP1 % 8 == 0, "alignment of the overlays");
static_assert(sizeof(FrameLds) <= 6400, "five 1280-byte LDS granules: 25 workgroups per CU");
#endif

} // namespace og
