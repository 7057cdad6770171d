/*
 * oc_opus.h -- CPU ORACLE for the Opus decode hot path (test infrastructure only).
 *
 * A plain-C restatement of the reference's opus_multistream_decode -> opus_decode_frame ->
 * {range decoder, CELT, SILK} path (src/opus_decoder.cpp, src/celt.cpp, src/silk.cpp under
 * /root/reference), with the codec state held in an explicit per-stream struct instead of the
 * reference's file-scope globals.  "Bit-exact to the reference" includes its quirks
 * (SURVEY.md appendix A, Q1-Q14).
 *
 * PARITY PIN: the reference needs <Arduino.h>, which this image lacks, so it cannot be built
 * here without a stand-in header (not allowed); there is no oracle/_ref.  The reference has no
 * tests or vectors of its own.  This restatement is pinned by the outputs of the reference
 * recorded during the survey session (KAT 1-3: PCM hashes and samples),
 * checked in tests/test_oracle_kat.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef OC_OPUS_H
#define OC_OPUS_H
#include "oc_math.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OC_OK 0
#define OC_BAD_ARG (-1)
#define OC_BUFFER_TOO_SMALL (-2)
#define OC_INTERNAL_ERROR (-3)
#define OC_INVALID_PACKET (-4)
/* what celt_decode_with_ec returns for a frame it refuses (src/celt.cpp:2211,2216,2225: ERR_OPUS_CELT_BAD_ARG,
 * src/opus_decoder.h:55); opus_decode_frame / opus_decode_native pass it up unchanged (src/opus_decoder.cpp:277,333-337) */
#define OC_CELT_BAD_ARG (-18)

#define OC_MODE_SILK 1000
#define OC_MODE_HYBRID 1001
#define OC_MODE_CELT 1002

#define OC_BW_NB 1101
#define OC_BW_MB 1102
#define OC_BW_WB 1103
#define OC_BW_SWB 1104
#define OC_BW_FB 1105

/* ---- range decoder (celt.h:73-86, celt.cpp:2627-2792) ------------------------------------ */
typedef struct {
    const u8 *buf;
    u32 storage, end_offs, end_window;
    i32 nend_bits, nbits_total;
    u32 offs, rng, val, ext;
    i32 rem, error;
} oc_rc;

void oc_rc_init(oc_rc *rc, const u8 *buf, u32 len);
u32 oc_rc_decode(oc_rc *rc, u32 ft);
u32 oc_rc_decode_bin(oc_rc *rc, unsigned bits);
void oc_rc_update(oc_rc *rc, u32 fl, u32 fh, u32 ft);
int oc_rc_bit_logp(oc_rc *rc, unsigned logp);
int oc_rc_icdf(oc_rc *rc, const u8 *icdf, unsigned ftb);
u32 oc_rc_uint(oc_rc *rc, u32 ft);
u32 oc_rc_bits(oc_rc *rc, unsigned bits);
u32 oc_rc_tell_frac(const oc_rc *rc);
int oc_rc_laplace(oc_rc *rc, u32 fs, int decay);
static inline i32 oc_rc_tell(const oc_rc *rc) { return rc->nbits_total - ilog32(rc->rng); }

/* ---- CELT (celt.h:150-171 + trailing arrays celt.cpp:2202-2206) -------------------------- */
#define OC_NBANDS 21
#define OC_OVERLAP 120
#define OC_HIST 1024             /* comb filter reach: x[-T-2], T <= 1022 (celt.cpp:833, :2262) */
#define OC_SYNLEN (OC_HIST + 960 + OC_OVERLAP)

typedef struct {
    i32 channels;          /* CC: output channels of the decoder */
    i32 stream_channels;   /* C : channels coded in the packet */
    i32 start_band;
    i32 end_band;          /* 21 in reference mode (Q1); by bandwidth in RFC mode */
    i32 disable_inv;
    u32 rng;
    i32 error;
    i32 pf_period, pf_period_old;
    i16 pf_gain, pf_gain_old;
    i32 pf_tapset, pf_tapset_old;
    i32 deemph_mem[2];
    i32 syn[2][OC_SYNLEN]; /* [0,OC_HIST) = history, out_syn starts at OC_HIST */
    i16 bandE[2 * OC_NBANDS], logE1[2 * OC_NBANDS], logE2[2 * OC_NBANDS];
    i16 backgroundLogE[2 * OC_NBANDS]; /* celt.cpp:2206, :2413-2418: the noise floor the concealment decays to */
    i32 loss_count;                    /* celt.h:161 */
    i32 plc_pitch;                     /* pitch-based concealment: the period found at the first lost frame, */
    i16 plc_lpc[2][24];                /* and each channel's LPC filter of that frame (kept for the losses that follow) */
} oc_celt;

/* optional stage taps for parity tests of the HIP kernels (filled when non-NULL) */
typedef struct {
    i32 valid;
    i32 is_transient, silence, coded_bands, intensity, dual_stereo, spread, LM;
    i32 pf_pitch, pf_gain, pf_tapset, anti_collapse_on;
    i32 pulses[OC_NBANDS], fine_quant[OC_NBANDS], tf_res[OC_NBANDS];
    i16 X[2 * 960];        /* normalised coefficients after quant_all_bands (+ anti-collapse) */
    i16 bandE[2 * OC_NBANDS];
    i32 freq[2][960];      /* denormalised MDCT input per channel */
    i32 syn_pre[2][960 + OC_OVERLAP];  /* IMDCT output before the comb filter */
    i32 syn_post[2][960];  /* after the comb filter */
    u32 rc_rng_end;
} oc_celt_taps;

void oc_celt_init(oc_celt *st, int channels);           /* celt_decoder_init  celt.cpp:1933 */
void oc_celt_reset(oc_celt *st);                        /* OPUS_RESET_STATE   celt.cpp:2479 (partial, Q5) */
int oc_celt_decode(oc_celt *st, oc_rc *rc, i16 *pcm, int frame_size, oc_celt_taps *taps); /* celt.cpp:2162 */
/* Concealment of a lost CELT frame (RFC mode only; the reference has none, Q8), after RFC 6716's decoder (celt_decode_lost): the
 * first five lost frames of a stream that codes from band 0 are extrapolated from the pitch period of the last output
 * (celt_decode_lost_pitch in oc_celt.c: the structure of that decoder, fixed-point detail of this repository's own); after that,
 * and for hybrid's CELT layer, the noise-based branch: band energies decay towards backgroundLogE, every band is filled with
 * renormalised LCG noise, one long MDCT, no post-filter.  No reference output pins either (PARITY-UNPINNED). */
int oc_celt_decode_lost(oc_celt *st, i16 *pcm, int frame_size);

/* stage functions exported for unit tests of the HIP kernels */
void oc_imdct(const i32 *in, i32 *out, int overlap, int shift, int stride);     /* celt.cpp:3204 */
void oc_fft(int shift, i32 *cpx);                                               /* celt.cpp:2997 */
void oc_comb_filter(i32 *y, i32 *x, int T0, int T1, int N, i16 g0, i16 g1, int tap0, int tap1); /* :848 */
i32 oc_cwrsi(int n, int k, u32 i, i32 *y);                                      /* celt.cpp:2545 */
void oc_exp_rotation(i16 *X, int len, int dir, int stride, int K, int spread);  /* celt.cpp:707 */

/* ---- SILK (silk.h:705-764) ---------------------------------------------------------------- */
struct oc_silk;
typedef struct oc_silk oc_silk;
int oc_silk_sizeof(void);
void oc_silk_init(oc_silk *s);                           /* silk_InitDecoder silk.cpp:1792 */
/* silk_Decode silk.cpp:1481 with lostFlag=0, API rate 48 kHz, 20 ms payload; first = first frame in packet */
int oc_silk_decode(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, i16 *out, i32 *n_out);
/* the same with the packet's true payload duration (RFC mode): 10 ms -> two subframes, 40 / 60 ms -> two / three internal
 * frames per packet, one call each (silk.cpp:1522-1540 sets nFramesPerPacket / nb_subfr from payloadSize_ms; the reference
 * pins that to 20).  NOT pinned by any reference output: written from RFC 6716 section 4.2. */
int oc_silk_decode_ms(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, int payload_ms, i16 *out, i32 *n_out);
/* ... and with the reference's lostFlag (silk.cpp:1483): 0 normal, 1 packet lost (conceal payload_ms = 10 or 20; rc unused;
 * internal_hz 0 keeps the rate of the last frame), 2 decode the LBRR (FEC) copy.  The reference never calls with 1 or 2 (Q8):
 * the code is its silk_PLC / silk_CNG (:2862-3185, :1305-1432), reachable only in RFC mode; NOT pinned by any reference output. */
int oc_silk_decode_ex(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, int payload_ms, int lostFlag, i16 *out,
                      i32 *n_out);

/* ---- packet layer (opus_decoder.cpp) ------------------------------------------------------ */
typedef struct oc_decoder {
    i32 channels;
    i32 rfc;               /* 0: reference-exact (every frame decodes as 20 ms, Q6); 1: RFC mode, see oc_decoder_set_rfc */
    i32 stream_channels, bandwidth, mode, prev_mode, frame_size, last_packet_duration;
    i32 prev_redundancy;   /* RFC mode: the last frame ended with a redundant CELT frame (a SILK -> CELT transition is under way) */
    i32 last_redundancy;   /* test hook: what the last decoded frame carried -- 0 none, 1 SILK -> CELT, 3 CELT -> SILK redundancy */
    u32 range_final;
    oc_rc rc;              /* the reference's global s_ec: survives between frames (Q4) */
    oc_celt celt;
    oc_silk *silk;         /* allocated by oc_decoder_create */
    oc_celt_taps *taps;    /* optional */
} oc_decoder;

oc_decoder *oc_decoder_create(int channels);             /* fresh state == opus_multistream_decoder_init */
void oc_decoder_destroy(oc_decoder *d);
void oc_decoder_init(oc_decoder *d, int channels);       /* opus_decoder_init opus_decoder.cpp:82 */
void oc_decoder_reset(oc_decoder *d);                    /* OPUS_RESET_STATE  opus_decoder.cpp:382 */
/* TEST TAP, not the reference's ctl: the range decoder's last range (^ a redundant frame's in RFC mode), what RFC 6716's decoder
 * reports as its final range.  The reference's OPUS_GET_FINAL_RANGE (opus_decoder.cpp:375-380) returns OpusDecoder::rangeFinal,
 * which nothing ever assigns: 0, always (oc_decoder_ctl_final_range). */
u32 oc_decoder_final_range(const oc_decoder *d);
u32 oc_decoder_ctl_final_range(const oc_decoder *d);      /* OPUS_GET_FINAL_RANGE as the reference answers it: 0 */
/* OPUS_GET_PITCH (opus_decoder.cpp:399-407): after a CELT-only frame the reference hands the POINTER to celt_decoder_ctl as the
 * request number and gets OPUS_UNIMPLEMENTED (-5: returned here); else the SILK decoder's last exported lag at 48 kHz in *value */
int oc_decoder_ctl_pitch(const oc_decoder *d, i32 *value);
i32 oc_silk_prev_pitch_lag(const oc_silk *s);
/* RFC mode (PARITY UNPINNED -- the reference cannot do this and no libopus exists in the image): frames decode
 * at the duration their TOC names (CELT 2.5 / 5 / 10 / 20 ms, SILK 10 / 20 / 40 / 60 ms, hybrid 10 / 20 ms), multi-frame
 * packets accordingly; CELT's last band follows the bandwidth (13 / 17 / 19 / 21: Q1 fixed); a SILK-only frame after a hybrid
 * one fades the CELT layer out with the two-byte silence frame (RFC 6716 section 4.5.2) instead of Q4's stale-coder frame;
 * the redundant 5 ms CELT frames of mode transitions are decoded and cross-faded in (section 4.5.1; Q2 fixed); lost packets,
 * DTX frames and forward error correction as in RFC 6716's decoder (oc_decode with data == NULL, oc_decode_fec; Q8 fixed).
 * Everything else stays as the reference has it (Q3, Q5, Q7).  Survives oc_decoder_init / _reset. */
void oc_decoder_set_rfc(oc_decoder *d, int on);
/* opus_decode_native (opus_decoder.cpp:280) for one elementary stream; returns samples/channel or <0 */
int oc_decode(oc_decoder *d, const u8 *data, i32 len, i16 *pcm, int frame_size);
/* RFC mode only, PARITY UNPINNED: opus_decode(decode_fec = 1) as RFC 6716's decoder has it (the reference has neither the flag
 * nor the path, Q8).  `data` is the packet AFTER a lost one; frame_size samples are concealed, the last frame's worth of them from
 * the LBRR frames in `data` where it carries any (SILK / hybrid). */
int oc_decode_fec(oc_decoder *d, const u8 *data, i32 len, i16 *pcm, int frame_size);
int oc_packet_parse(const u8 *data, i32 len, int self_delimited, u8 *out_toc, i16 size[48],
                    int *payload_offset, i32 *packet_offset);               /* opus_decoder.cpp:559 */
int oc_packet_mode(const u8 *data);
int oc_packet_bandwidth(const u8 *data);
int oc_packet_samples_per_frame(const u8 *data, i32 Fs);
int oc_packet_channels(const u8 *data);

#ifdef __cplusplus
}
#endif
#endif
