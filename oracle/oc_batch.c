/*
 * oc_batch.c -- CPU ORACLE (test infrastructure): decode many independent streams in one call.
 * Used by tests (reference PCM for large batches) and by bench.py's cpu_baseline leg, where it is
 * the thing being TIMED as the CPU baseline ("port" of the reference path), never the product.
 */
#include <stdlib.h>
#include "oc_opus.h"

/* payloads: [n_frames][n_streams][L] bytes; packet = toc + payload (code 0).  Streams [s0, s1) are decoded
 * with a fresh decoder each; pcm (may be NULL): [n_streams][n_frames][960][channels].  Returns the number of
 * frames decoded successfully. */
long oc_batch_decode(int channels, int toc, const u8 *payloads, int n_streams, int n_frames, int L, int s0, int s1,
                     i16 *pcm) {
    oc_decoder *d = oc_decoder_create(channels);
    i16 *tmp = (i16 *)malloc(sizeof(i16) * 5760 * 2);
    u8 pkt[1300];
    long ok = 0;
    int s, f, i;
    if (!d || !tmp || L > 1275) return -1;
    for (s = s0; s < s1; s++) {
        oc_decoder_init(d, channels);
        for (f = 0; f < n_frames; f++) {
            const u8 *p = payloads + ((size_t)f * n_streams + s) * L;
            int r;
            pkt[0] = (u8)toc;
            for (i = 0; i < L; i++) pkt[1 + i] = p[i];
            r = oc_decode(d, pkt, L + 1, tmp, 5760);
            if (r == 960) {
                ok++;
                if (pcm) memcpy(pcm + (((size_t)s * n_frames + f) * 960) * channels, tmp, sizeof(i16) * 960 * channels);
            }
        }
    }
    free(tmp);
    oc_decoder_destroy(d);
    return ok;
}

/* Arbitrary packets: frame f of stream s is the packet arena[offs[f * n_streams + s] .. + lens[f * n_streams + s]) (TOC
 * first, any frame count code).  Streams [s0, s1) are decoded with a fresh decoder each and carry their state across
 * frames.  rets: [n_streams][n_frames] return value of every call; pcm: [n_streams][n_frames][960][channels], written
 * for calls that return 960 (the first 960 * channels entries of the decoder's output).  Returns the number of calls that
 * returned 960. */
long oc_batch_decode_var_cap(int channels, const u8 *arena, const long long *offs, const i32 *lens, int n_streams, int n_frames,
                             int s0, int s1, i16 *pcm, i32 *rets, int cap_frames);
long oc_batch_decode_var(int channels, const u8 *arena, const long long *offs, const i32 *lens, int n_streams, int n_frames,
                         int s0, int s1, i16 *pcm, i32 *rets) {
    return oc_batch_decode_var_cap(channels, arena, offs, lens, n_streams, n_frames, s0, s1, pcm, rets, 1);
}
/* The same with room for cap_frames 20 ms frames per call (multi-frame packets): the decoder is called with
 * frame_size = 960 * cap_frames like opus_multistream_decode's caller would; pcm: [n_streams][n_frames][960 * cap_frames]
 * [channels], the first r samples written when the call returns r > 0.  Returns the number of calls with r > 0. */
long oc_batch_decode_var_cap(int channels, const u8 *arena, const long long *offs, const i32 *lens, int n_streams, int n_frames,
                             int s0, int s1, i16 *pcm, i32 *rets, int cap_frames) {
    oc_decoder *d = oc_decoder_create(channels);
    /* Q6: the reference checks count * (samples per frame from the TOC) against frame_size but decodes EVERY frame as 960
     * samples: a packet of many short frames passes the check and writes up to 48 * 960 samples.  Room for that. */
    i16 *tmp = (i16 *)malloc(sizeof(i16) * 48 * 960 * 2);
    long ok = 0;
    int s, f;
    if (!d || !tmp || cap_frames < 1 || cap_frames > 6) return -1;
    for (s = s0; s < s1; s++) {
        oc_decoder_init(d, channels);
        for (f = 0; f < n_frames; f++) {
            const size_t k = (size_t)f * n_streams + s;
            const int r = oc_decode(d, arena + offs[k], lens[k], tmp, 960 * cap_frames);
            rets[(size_t)s * n_frames + f] = r;
            if (r > 0 && r <= 960 * cap_frames) {
                ok++;
                memcpy(pcm + (((size_t)s * n_frames + f) * 960 * cap_frames) * channels, tmp, sizeof(i16) * (size_t)r * channels);
            }
        }
    }
    free(tmp);
    oc_decoder_destroy(d);
    return ok;
}

/* RFC mode (oc_decoder_set_rfc), with lost packets and forward error correction: entry k = f * n_streams + s is what stream s does
 * at step f -- ops[k] 0: decode the packet arena[offs[k] .. + lens[k]);  1: the packet was LOST and is concealed for as long as
 * the stream's last packet was (oc_decode(NULL), 20 ms before the first packet);  2: the packet was lost and is RECOVERED from the
 * forward error correction data of the stream's NEXT packet, which is what offs[k] / lens[k] name (oc_decode_fec over the last
 * packet's duration).  Streams [s0, s1) run on a fresh decoder each, state carried across steps.  rets: [n_streams][n_frames]
 * return values; pcm_last (may be NULL): [n_streams][5760][channels], the output of every stream's LAST step.  Returns the
 * number of calls that produced samples. */
long oc_batch_decode_rfc(int channels, const u8 *arena, const long long *offs, const i32 *lens, const u8 *ops, int n_streams,
                         int n_frames, int s0, int s1, i16 *pcm_last, i32 *rets) {
    oc_decoder *d = oc_decoder_create(channels);
    i16 *tmp = (i16 *)malloc(sizeof(i16) * 5760 * 2);
    long ok = 0;
    int s, f;
    if (!d || !tmp) return -1;
    for (s = s0; s < s1; s++) {
        int last_dur = 960;
        oc_decoder_init(d, channels);
        oc_decoder_set_rfc(d, 1);
        for (f = 0; f < n_frames; f++) {
            const size_t k = (size_t)f * n_streams + s;
            int r;
            if (ops[k] == 1)
                r = oc_decode(d, NULL, 0, tmp, last_dur);
            else if (ops[k] == 2)
                r = oc_decode_fec(d, arena + offs[k], lens[k], tmp, last_dur);
            else {
                r = oc_decode(d, arena + offs[k], lens[k], tmp, 5760);
                if (r > 0) last_dur = r;
            }
            rets[(size_t)s * n_frames + f] = r;
            if (r > 0) ok++;
            if (pcm_last && f == n_frames - 1 && r > 0 && r <= 5760)
                memcpy(pcm_last + (size_t)s * 5760 * channels, tmp, sizeof(i16) * (size_t)r * channels);
        }
    }
    free(tmp);
    oc_decoder_destroy(d);
    return ok;
}

/* stage taps for parity tests of the HIP kernels: enable once, then copy after each oc_decode() call */
int oc_taps_enable(oc_decoder *d) {
    if (!d->taps) d->taps = (oc_celt_taps *)calloc(1, sizeof(oc_celt_taps));
    return d->taps != NULL;
}
/* what: 0 = X (i16[1920]), 1 = bandE (i16[42]), 2 = syn_pre ch c (i32[1080]), 3 = syn_post ch c (i32[960]),
 * 4 = the frame's header as i32[75]: {is_transient, silence, coded_bands, intensity, dual_stereo, spread, LM, pf_pitch, pf_gain,
 * pf_tapset, anti_collapse_on, rng after the frame} then pulses[21], fine_quant[21], tf_res[21] */
int oc_taps_copy(const oc_decoder *d, int what, int c, void *dst) {
    const oc_celt_taps *t = d->taps;
    if (!t || !t->valid) return -1;
    switch (what) {
        case 4: {
            i32 v[75] = {t->is_transient, t->silence, t->coded_bands, t->intensity, t->dual_stereo, t->spread, t->LM, t->pf_pitch,
                         t->pf_gain, t->pf_tapset, t->anti_collapse_on, (i32)t->rc_rng_end};
            memcpy(v + 12, t->pulses, sizeof(t->pulses));
            memcpy(v + 33, t->fine_quant, sizeof(t->fine_quant));
            memcpy(v + 54, t->tf_res, sizeof(t->tf_res));
            memcpy(dst, v, sizeof(v));
            return (int)sizeof(v);
        }
        case 0: memcpy(dst, t->X, sizeof(t->X)); return (int)sizeof(t->X);
        case 1: memcpy(dst, t->bandE, sizeof(t->bandE)); return (int)sizeof(t->bandE);
        case 2: memcpy(dst, t->syn_pre[c], sizeof(t->syn_pre[c])); return (int)sizeof(t->syn_pre[c]);
        case 3: memcpy(dst, t->syn_post[c], sizeof(t->syn_post[c])); return (int)sizeof(t->syn_post[c]);
    }
    return -1;
}

/* TEST ENTRY: the mode of the last frame (0 before the first one): what a concealment would run in */
int oc_decoder_prev_mode(const oc_decoder *d) { return d->prev_redundancy ? OC_MODE_CELT : d->prev_mode; }
/* TEST ENTRY: 0 none, 1 SILK -> CELT, 3 CELT -> SILK: the redundancy of the last frame decoded (RFC mode) */
int oc_decoder_last_redundancy(const oc_decoder *d) { return d->last_redundancy; }
