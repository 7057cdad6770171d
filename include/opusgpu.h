/*
 * opusgpu.h -- C ABI of the MI355X batched Opus decoder (libopusgpu.so).
 *
 * This is the drop-in boundary for the reference's decode hot path
 *   opus_multistream_decode -> opus_decode_native -> opus_decode_frame -> {ec_*, silk_Decode, celt_decode_with_ec}
 *   (reference: src/opus_decoder.cpp:931, :280, :154; src/silk.cpp:1481; src/celt.cpp:2162).
 * The reference decodes ONE stream per process with its codec state in file-scope globals; this
 * library keeps one state record per stream in HBM and decodes one 20 ms frame of every submitted
 * stream per step, one frame per wavefront.  PCM is bit-exact to the reference's fixed-point decoder.
 *
 * Plain C: opaque handle, plain pointers and sizes, negative OPUS_* error codes (reference values,
 * src/opus_decoder.h:70-77).  No exceptions cross this boundary.  One host thread per context.
 * The reference-compatible C++ entry points (opus_multistream_decode, op_read_stereo, ...) declared in
 * include/opus_decoder.h and include/opusfile.h are implemented on top of this ABI.
 */
#ifndef OPUSGPU_H
#define OPUSGPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPUSGPU_OK 0
#define OPUSGPU_BAD_ARG (-1)          /* OPUS_BAD_ARG */
#define OPUSGPU_BUFFER_TOO_SMALL (-2) /* OPUS_BUFFER_TOO_SMALL */
#define OPUSGPU_INTERNAL_ERROR (-3)   /* OPUS_INTERNAL_ERROR */
#define OPUSGPU_INVALID_PACKET (-4)   /* OPUS_INVALID_PACKET */
#define OPUSGPU_UNIMPLEMENTED (-5)    /* OPUS_UNIMPLEMENTED */
#define OPUSGPU_ALLOC_FAIL (-7)       /* OPUS_ALLOC_FAIL */
#define OPUSGPU_CELT_BAD_ARG (-18)    /* ERR_OPUS_CELT_BAD_ARG: a CELT-only / hybrid frame of <= 1 byte (src/celt.cpp:2225) */
#define OPUSGPU_ERR_NO_DEVICE (-100)  /* no usable HIP device / kernel image: there is NO CPU fallback */
#define OPUSGPU_ERR_HIP (-101)        /* a HIP runtime call failed; see opusgpu_last_error() */

#define OPUSGPU_FRAME_SAMPLES 960     /* the reference decodes 20 ms at 48 kHz only (src/opus_decoder.cpp:161) */
#define OPUSGPU_MAX_FRAME_BYTES 1275

typedef struct opusgpu_ctx opusgpu_ctx;

/* One frame of work for one stream in one decode step (16 bytes, device layout).
 * flags: bits 0-1 mode (0 SILK-only, 1 hybrid, 2 CELT-only); bits 2-4 bandwidth (0 NB .. 4 FB); bit 5 stereo;
 * bits 6-10: frame duration, the RFC bit and the FEC bit, zero in reference mode (see OPUSGPU_MODE_RFC);
 * bit 11 OPUSGPU_DESC_NO_MODE: see EMPTY PACKETS below.
 * These are the TOC fields opus_decode_native derives (src/opus_decoder.cpp:312-315).
 *
 * EMPTY PACKETS in reference mode (data == NULL or len == 0; src/opus_decoder.cpp:290-308).  The reference conceals nothing, but
 * its branch for them is live: it calls opus_decode_frame(st, NULL, 0, ...) -- a frame of NO bytes in the decoder's LAST mode,
 * bandwidth and channel count (what the last accepted packet's TOC set, :327-331), always 960 samples (:161) -- again and again
 * until `frame_size` samples exist, and stops at the first pass that fails.  What a pass does follows from opus_decode_frame:
 *   last packet SILK-only  -> the SILK decoder runs off a coder that reads zeros: 960 samples of PCM, the state moves on;
 *   last packet hybrid     -> the SILK half runs (and its state moves on), then celt_decode_with_ec refuses the empty frame:
 *                             ERR_OPUS_CELT_BAD_ARG = -18 (src/celt.cpp:2225); prev_mode is updated all the same (:276);
 *   last packet CELT-only  -> -18, nothing but prev_mode touched;
 *   no packet since the stream was created or reset -> st->mode is 0, which :175 / :249 run like hybrid (SILK at 16 kHz on the
 *                             decoder's channel count, then -18), and prev_mode stays 0.
 * Here: opusgpu_decode_packets takes packets[i] == NULL or lens[i] == 0 as such a packet with frame_size = frame_capacity x 960
 * and returns in result[i] the samples produced or the failing pass's code (the library remembers every stream's last accepted
 * TOC).  On the device path the caller passes, per pass, a descriptor with len 0 and the flags of the stream's last accepted
 * packet -- or, before the stream's first packet, opusgpu_empty_packet_to_frames' flags (hybrid, the decoder's channel count,
 * OPUSGPU_DESC_NO_MODE).  opusgpu_empty_packet_to_frames builds the descriptors either way.  (The library's memory of a stream's
 * last accepted TOC is kept by opusgpu_decode_packets and cleared by opusgpu_streams_alloc / _reset: a caller that decodes a
 * stream through the device path keeps that stream's last flags itself, and should not mix the two paths on one stream around an
 * empty packet.) */
#define OPUSGPU_DESC_NO_MODE (1 << 11)
typedef struct opusgpu_frame_desc {
    int32_t stream;  /* stream index in the context */
    int32_t offset;  /* byte offset of the frame payload inside the packet arena */
    int32_t len;     /* payload bytes, 0..1275 */
    int32_t flags;
} opusgpu_frame_desc;

/* ---- context -------------------------------------------------------------------------------- */
int opusgpu_version(void);
/* Binds to HIP device `device` (>=0).  Fails with OPUSGPU_ERR_NO_DEVICE when no GPU is usable. */
int opusgpu_ctx_create(int device, opusgpu_ctx **out);
void opusgpu_ctx_destroy(opusgpu_ctx *ctx);
const char *opusgpu_last_error(const opusgpu_ctx *ctx);

/* ---- RFC mode (opt-in, off by default) ---------------------------------------------------------
 * The reference decodes every frame as 20 ms whatever its TOC says (src/opus_decoder.cpp:161, :186, :341; Q6).  With
 * OPUSGPU_MODE_RFC set, frames decode at the duration the TOC names -- CELT 2.5 / 5 / 10 / 20 ms, SILK 10 / 20 / 40 / 60 ms
 * (src/silk.cpp:1522-1540 with the real payload duration), hybrid 10 / 20 ms -- multi-frame packets (codes 1 - 3) accordingly;
 * CELT's last band follows the bandwidth (Q1 fixed) and a SILK-only frame after a hybrid one fades the CELT layer out with the
 * two-byte silence frame of RFC 6716 section 4.5.2 instead of Q4's frame off the stale coder; the redundant 5 ms CELT frames of
 * mode transitions (section 4.5.1) are decoded and cross-faded in (Q2 fixed), and a switch between CELT-only and the SILK modes
 * that no redundant frame covers starts with 5 ms of the old mode's concealment, cross-faded into the new frame (section 4.5).
 * Everything else stays as the reference has it (Q3 mixing, Q5 partial reset).  The reference cannot produce these outputs and
 * no libopus exists:
 * this mode is bit-exact to oracle/'s RFC mode (oc_decoder_set_rfc), which is PARITY-UNPINNED.
 * RFC-mode frames run on a kernel of their own (wave-uniform entropy decoding): the mode is for completeness, not speed.
 * LOSS PATH (the reference has none, Q8 -- in reference mode an empty packet does what the reference's
 * empty-packet branch does, see EMPTY PACKETS above: no concealment):
 *   - opusgpu_decode_packets: packets[i] == NULL or lens[i] == 0 is a LOST packet: it is concealed for as long as the stream's
 *     last packet was (frame count x frame duration; 20 ms of zeros before the stream's first packet or after a reset), what
 *     opus_decode(data = NULL, frame_size = last duration) gives; result[i] = the samples concealed;
 *   - a frame of at most one payload byte (DTX) is concealed for its TOC's duration;
 *   - opusgpu_decode_step_device: a descriptor with len <= 1 conceals the duration in its flags; mode and bandwidth come from
 *     the stream's state, the stereo bit should be that of the stream's last packet.
 *   SILK conceals with the reference's own (unreachable) silk_PLC / silk_CNG code restated (src/silk.cpp:2862-3185, :1305-1432),
 *   CELT like RFC 6716's decoder (celt_decode_lost): the first five lost frames of a CELT-only stream are extrapolated from the
 *   pitch period of the last output (pitch search, order-24 LPC, the residual of the last two periods repeated through the
 *   synthesis filter with a decay, an overlap tail for the next frame's transform), later ones and hybrid's CELT layer are
 *   noise at the decaying band energies; hybrid conceals with both coders.
 *   Two things to know (a decoder's concealment is not normative, and the oracle's RFC mode makes the same choices, so GPU and
 *   oracle agree with each other, not sample by sample with libopus): (1) the pitch-based branch follows that decoder's
 *   STRUCTURE with fixed-point detail of this repository's own (64-bit accumulators, the 1,024 samples of history the decoder
 *   keeps, csrc/og_plc.hpp); (2) a concealment is cut into pieces of the stream's LAST frame duration (what opus_decode(NULL)
 *   is asked for), with a remainder of 30 / 50 ms as 20 / 40 + 10 ms.  Use one mode per stream from its (re)set on:
 *   the state the concealment needs is only kept by RFC-mode frames.
 * Applies to opusgpu_packet_to_frames_mode / opusgpu_decode_packets / opusgpu_decode_step_device calls made after it is set:
 *   - descriptors carry the duration and the mode bit (frame_desc.flags bits 6 - 9); one without the bit is OPUSGPU_BAD_ARG;
 *   - opusgpu_decode_packets: result[i] and the PCM block of packet i hold the packet's true sample count (<= 5760);
 *   - opusgpu_decode_step_device: d_pcm is [n][OPUSGPU_RFC_FRAME_SAMPLES * channels] (room for a 60 ms frame), d_result[f]
 *     the frame's sample count. */
#define OPUSGPU_MODE_REFERENCE 0
#define OPUSGPU_MODE_RFC 1
#define OPUSGPU_RFC_FRAME_SAMPLES 2880
int opusgpu_set_mode(opusgpu_ctx *ctx, int mode);
int opusgpu_get_mode(const opusgpu_ctx *ctx);

/* Pipelined decode steps (off by default).  A stream's frames are sequential, but only two things tie step k+1 to step k
 * before its reconstruction: the range decoder of a CELT frame predicts the band energies from the previous frame's, and
 * the SILK half reads the SILK state.  With pipelining on, the library carries the band energies in the parse kernel
 * (k_celt_parse) and runs that kernel for step k+1's CELT-only frames on a stream of its own, NEXT TO step k's
 * reconstruction (k_celt_recon_fb / k_celt_post), into one of three rotating sets of parse records; the reconstruction runs on another
 * stream of the library's own and never touches the caller's buffers, the step's own stream waits for it before the
 * de-emphasis (k_celt_post) writes PCM and result codes.  Results are bit-identical to the in-order flow (tests/test_gpu_pipeline.py); the step's stream
 * still completes everything the step launched, so the caller synchronises exactly as before.
 * What the caller additionally guarantees while it is on, for opusgpu_decode_step_device:
 *   - d_descs and d_arena of a call are COMPLETE in device memory when the call is made (uploaded and synchronised, or
 *     produced by work that has finished) -- not merely queued on `hip_stream` ahead of the call;
 *   - consecutive steps use the same stream (a change of stream is honoured by draining both, i.e. no overlap).
 * Which steps run ahead: a step the caller declares CELT-only (opusgpu_decode_step_device_modes, OPUSGPU_HAS_CELT alone) as
 * described; a step declared free of CELT-only frames (OPUSGPU_HAS_SILK, OPUSGPU_HAS_HYBRID or both) runs its SILK parse -- and a
 * hybrid frame's CELT parse behind it -- for step k+1 next to step k's SILK synthesis: the SILK parse keeps a copy of its own of
 * what the entropy half needs of the frames before (indices' history, gain index, NLSFs, rate, channel count, prev_mode), which is
 * exactly what it can compute itself (tests/test_emul_vs_oracle.py checks the copy against the state frame by frame); only a
 * step that may hold hybrid frames waits for a step that may have held SILK-only ones (the hybrid -> SILK-only transition frame
 * decodes a CELT frame in the step's last kernel).  A step that is undeclared or mixes CELT-only frames with the others runs in
 * order (cut into two halves on two streams when it has SILK frames).  Going from one kind of step to another drains the device.
 * opusgpu_decode_packets (whose uploads are queued by the call itself) runs in order regardless.  Switching synchronises
 * the device.  Reference: the per-packet call sequence this replaces is src/opus_decoder.cpp:931 -> :280 -> :154 ->
 * src/celt.cpp:2162, one packet at a time; there is nothing to pipeline there. */
int opusgpu_set_pipeline(opusgpu_ctx *ctx, int on);
int opusgpu_get_pipeline(const opusgpu_ctx *ctx);

/* ---- streams -------------------------------------------------------------------------------- */
/* Allocates `n_streams` per-stream state records in HBM (replacing any previous set) and gives each
 * the fresh state of opus_multistream_decoder_init(48000, channels, 1, channels-1, {0,1})
 * (src/opus_decoder.cpp:742).  channels is 1 or 2. */
int opusgpu_streams_alloc(opusgpu_ctx *ctx, int n_streams, int channels);
/* full != 0: fresh state again (decoder_init).  full == 0: OPUS_RESET_STATE semantics of
 * opus_multistream_decoder_ctl (src/opus_decoder.cpp:976 -> :382), which is NOT a full reset (CELT keeps
 * its synthesis history, band energies and de-emphasis memory: src/celt.cpp:2479-2498). */
int opusgpu_streams_reset(opusgpu_ctx *ctx, int first, int count, int full);
int opusgpu_stream_count(const opusgpu_ctx *ctx);
int opusgpu_stream_channels(const opusgpu_ctx *ctx);
size_t opusgpu_stream_state_bytes(void);

/* ---- host-buffer path: the batched equivalent of opus_multistream_decode (src/opus_decoder.cpp:931) --- */
/* Decodes packets[i] (lens[i] bytes, TOC first) for stream stream_ids[i], i < n; a stream may appear
 * at most once per call.  pcm receives n blocks of `frame_capacity` * 960 * channels interleaved int16;
 * result[i] = samples per channel decoded for packet i (frames * 960) or a negative OPUS_* code.
 * Packets with several frames (codes 1-3) are decoded frame after frame like opus_decode_native.
 * One deliberate difference from the reference: it checks the room as frames * (frame duration from the TOC) but
 * decodes every frame as 960 samples (src/opus_decoder.cpp:161, :323), so a packet of more short frames than
 * `frame_capacity` passes its check and overruns the caller's buffer.  Here such a packet gets
 * OPUSGPU_BUFFER_TOO_SMALL and nothing is written.
 * Returns OPUSGPU_OK or a context-level error. */
int opusgpu_decode_packets(opusgpu_ctx *ctx, int n, const int32_t *stream_ids, const uint8_t *const *packets,
                           const int32_t *lens, int16_t *pcm, int frame_capacity, int32_t *result);

/* RFC mode only (OPUSGPU_BAD_ARG otherwise): forward error correction, opus_decode(decode_fec = 1) of RFC 6716's decoder -- the
 * reference has neither the flag nor the path.  packets[i] is the packet that FOLLOWS a lost packet of stream stream_ids[i]: the
 * lost packet's duration (that of the stream's last packet) is produced -- the last frame's worth of it from the LBRR frames in
 * packets[i]'s first frame where it is a SILK-only or hybrid packet (SILK conceals the channels / internal frames without an LBRR
 * copy, hybrid's CELT layer conceals), everything before that by concealment; a CELT-only packet or predecessor carries no such
 * data: plain concealment.  result[i] = samples produced.  Call opusgpu_decode_packets with the same packets afterwards: the
 * packet itself is not decoded here.  On the device path: descriptor flags bit 10 on the packet's first frame. */
int opusgpu_decode_packets_fec(opusgpu_ctx *ctx, int n, const int32_t *stream_ids, const uint8_t *const *packets,
                               const int32_t *lens, int16_t *pcm, int frame_capacity, int32_t *result);

/* Splits one packet into frame descriptors exactly as opus_packet_parse_impl + the TOC helpers do
 * (src/opus_decoder.cpp:559, :135, :460, :474).  descs[k].offset is relative to the packet start.
 * Returns the frame count (1..48) or a negative OPUS_* code.  Pure host code. */
int opusgpu_packet_to_frames(const uint8_t *packet, int32_t len, int32_t stream, opusgpu_frame_desc descs[48]);
/* The same for a given mode (OPUSGPU_MODE_*): in RFC mode the descriptors carry the frames' duration (flags bits 6 - 8: 0 20 ms,
 * 1 2.5, 2 5, 3 10, 4 40, 5 60) and the RFC bit (bit 9). */
int opusgpu_packet_to_frames_mode(const uint8_t *packet, int32_t len, int32_t stream, int mode, opusgpu_frame_desc descs[48]);
/* (Both mirror opus_packet_parse_impl, whose answer to len == 0 is OPUS_INVALID_PACKET, src/opus_decoder.cpp:567: an empty
 * packet has no TOC to take descriptors from.)  Reference mode, the frames of an EMPTY packet for the device path (EMPTY
 * PACKETS above): `last_flags` = the flags of the stream's last accepted packet (descs[0].flags of opusgpu_packet_to_frames),
 * or a negative value when the stream has had none since it was created / reset (`decoder_channels` then names the stream's
 * channel count); `frame_size` as opus_decode's.  Writes ceil(frame_size / 960) descriptors of len 0 -- one per pass of the
 * reference's loop, to be run one step each until a pass returns a negative code -- and returns their count; OPUSGPU_BAD_ARG
 * when frame_size <= 0 or no multiple of 120 (:290, :351) or more than 48 passes. */
int opusgpu_empty_packet_to_frames(int32_t stream, int32_t last_flags, int decoder_channels, int frame_size, opusgpu_frame_desc descs[48]);

/* ---- device-resident path (inputs and outputs stay in HBM; used by bench.py and on-device consumers) -- */
int opusgpu_dev_alloc(opusgpu_ctx *ctx, size_t bytes, void **dptr); /* hipMalloc of bytes + 16: aligned and tailed as a packet arena must be */
int opusgpu_dev_free(opusgpu_ctx *ctx, void *dptr);
int opusgpu_memcpy_h2d(opusgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
int opusgpu_memcpy_d2h(opusgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
/* One decode step: n frames described by d_descs (device array of opusgpu_frame_desc), payload bytes in
 * d_arena, PCM to d_pcm[n][960*channels] int16, per-frame result to d_result[n] int32.  Asynchronous on the
 * context's stream (or on `hip_stream` if not NULL: a hipStream_t).
 * The tables are in device memory and are NOT validated beyond what a frame's own kernel can see.  The caller guarantees:
 *   - a stream appears at most once in a step (frames of one stream are sequential: one step each);
 *   - 0 <= offset and offset + len lies inside the arena; d_arena is 16-byte aligned (checked: OPUSGPU_BAD_ARG) and the
 *     ALLOCATION extends at least to the next multiple of 16 past the last frame's end (the kernels fetch packets as aligned
 *     16-byte pieces; allocating 16 bytes more than the packed bytes, as every producer in this repository does, is enough);
 *   - d_descs, d_arena, d_pcm and d_result stay valid and unmodified until the step has completed on its stream.
 * Checked on the device, per frame, and reported in d_result: stream index out of range -> OPUSGPU_BAD_ARG; len outside
 * 0..1275 -> OPUSGPU_BAD_ARG; every decode error the reference would return for the frame.
 * opusgpu_packet_to_frames, opusgpu_decode_packets and opusgpu_pages_demux produce tables that satisfy all of this. */
int opusgpu_decode_step_device(opusgpu_ctx *ctx, int n, const void *d_descs, const void *d_arena, void *d_pcm,
                               void *d_result, void *hip_stream);
/* The same, for a caller that knows which kinds of frame the step contains (whoever framed the packets does: the TOC byte).
 * `modes`: bit 0 SILK-only, bit 1 hybrid, bit 2 CELT-only frames MAY be present (1 .. 7; plus OPUSGPU_STEP_KEEPS_MODE).  The kernels of modes ruled out are
 * not launched -- on a step of 65,536 frames the launches that find nothing to do cost 1 - 2 % -- and a pipelined step
 * (opusgpu_set_pipeline) without SILK-only and hybrid frames also starts its reconstruction while the previous step's
 * de-emphasis still runs.  A frame of a mode that was ruled out is reported as OPUSGPU_BAD_ARG in d_result and not decoded. */
#define OPUSGPU_HAS_SILK 1
#define OPUSGPU_HAS_HYBRID 2
#define OPUSGPU_HAS_CELT 4
/* May be OR-ed into `modes` of a pipelined step (opusgpu_set_pipeline): the caller's word that NO STREAM OF THIS STEP HAS DECODED A
 * FRAME OF ANOTHER MODE (SILK-only, hybrid, CELT-only) SINCE ITS LAST RESET -- SURVEY 8d config 5: a stream's mode is fixed.  What it
 * buys: (1) a step with frames of every mode (modes 7) runs ahead like a declared one -- its entropy kernels next to the arithmetic
 * kernels of the step before -- where without the flag it runs in order: a CELT-only frame changes what the SILK half of the SAME
 * stream's next frame must see, which the library cannot rule out by itself; (2) declared steps of the two pipelined kinds
 * (CELT-only; SILK-only / hybrid) follow each other without the drain that a stream crossing from one to the other would need.
 * Results with a true promise are bit-identical to the in-order flow; with a false one they are undefined for the streams that
 * broke it (never for others). */
#define OPUSGPU_STEP_KEEPS_MODE 8
int opusgpu_decode_step_device_modes(opusgpu_ctx *ctx, int n, const void *d_descs, const void *d_arena, void *d_pcm,
                                     void *d_result, void *hip_stream, int modes);
/* A WINDOW of consecutive decode steps in one call: step j has n[j] frames and the tables d_descs[j] / d_arena[j], and writes
 * d_pcm[j] / d_result[j]; everything opusgpu_decode_step_device_modes says holds per step (`modes`: 0 = not known), the tables of
 * every step are complete in device memory at the call.  Same results as n_steps single calls.  The point is pipelined steps
 * (opusgpu_set_pipeline) of CELT-only frames (SILK-only and hybrid steps pipeline the same way with either entry): knowing the step that follows, the library orders the kernels of neighbouring
 * steps by real dependencies -- the next step's parse is placed before this step's reconstruction, that before the previous
 * step's de-emphasis, each held by a stream memory wait on a count of started workgroups -- where a single call has to leave
 * the order to the hardware queues (round 2 used a spin-wait kernel and an unused LDS request for it; both are gone). */
int opusgpu_decode_steps_device(opusgpu_ctx *ctx, int n_steps, const int32_t *n, const void *const *d_descs, const void *const *d_arena,
                                void *const *d_pcm, void *const *d_result, void *hip_stream, int modes);
int opusgpu_synchronize(opusgpu_ctx *ctx);
/* HIP events on the context's stream, for timing from hosts without HIP headers. */
int opusgpu_event_create(opusgpu_ctx *ctx, void **event);
int opusgpu_event_record(opusgpu_ctx *ctx, void *event);
int opusgpu_event_elapsed_ms(opusgpu_ctx *ctx, void *start, void *stop, float *ms); /* synchronises on stop */
int opusgpu_event_destroy(opusgpu_ctx *ctx, void *event);
int opusgpu_event_synchronize(opusgpu_ctx *ctx, void *event); /* the host waits for it */
/* Uploads NEXT TO the decode (config 5: the ingest of the next batch of Ogg pages under the decode of this one).  The copy runs on
 * a stream of the context's own, not on the decode stream: one host thread demuxes batch b + 1 (opusgpu_pages_demux) and queues
 * its step tables and packet bytes with opusgpu_upload_async, then records a fence; the thread that decodes lets its stream wait
 * for that fence (opusgpu_stream_wait_event; hip_stream NULL = the context's stream -- and with it the streams pipelined steps run
 * ahead on, so that tables behind the fence count as complete in device memory for opusgpu_set_pipeline) before batch b + 1's first step.  `src` must
 * stay valid until the fence has passed (opusgpu_event_synchronize).  These four calls may come from a second host thread. */
int opusgpu_upload_async(opusgpu_ctx *ctx, void *dst, const void *src, size_t bytes);
int opusgpu_upload_fence(opusgpu_ctx *ctx, void *event);
int opusgpu_stream_wait_event(opusgpu_ctx *ctx, void *event, void *hip_stream);
/* Page-lock a caller-owned host buffer in place (and undo it): transfers from / into it need no staging copy. */
int opusgpu_host_register(opusgpu_ctx *ctx, void *ptr, size_t bytes);
int opusgpu_host_unregister(opusgpu_ctx *ctx, void *ptr);
/* Copies stream `index`'s raw state record to the host (tests / checkpointing). */
int opusgpu_stream_state_get(opusgpu_ctx *ctx, int index, void *dst, size_t bytes);
/* What the reference's OPUS_GET_PITCH ctl looks at (src/opus_decoder.cpp:399-407, src/silk.cpp:1764-1769), for stream `index`:
 * out[0] = prev_mode (0 before the first frame, else 1000 / 1001 / 1002), out[1] = the first SILK channel's prevSignalType
 * (2 = voiced), out[2] = its lagPrev, out[3] = its internal rate in kHz (0 before the first SILK frame).  include/opus_decoder.h's
 * ctl is built on it (csrc/og_compat.cpp).  Synchronises the context's stream. */
int opusgpu_stream_pitch_get(opusgpu_ctx *ctx, int index, int32_t out[4]);

/* TEST / DEBUG ENTRY: what the kernels of the last opusgpu_decode_step_device call (reference mode, split path) left BETWEEN
 * the stages, for slot `slot` of that step and the stream it belongs to -- so that a parity test can tell which kernel a
 * difference comes from instead of seeing it only in the PCM (tests/test_gpu_stage_taps.py compares every field with the
 * oracle's taps).  Synchronises the context's stream. */
typedef struct opusgpu_stage_taps {
    /* k_celt_parse's record (CELT-only and hybrid frames; celt_valid = 0 otherwise): the frame's header as parsed */
    int32_t celt_valid, celt_ret, silence, transient, lm, spread, dual_stereo, anti_collapse_on, intensity, pf_pitch, pf_gain,
        pf_tapset, n_leaves;
    uint32_t celt_rng_final;
    int16_t bandE[42];   /* final band energies (coarse + fine + finalise), both channels */
    int16_t pulses[21];
    int8_t tf_res[21];
    int8_t pad0[3];
    /* k_celt_recon / k_celt_recon_fb: the stream's state after the step -- the frame's synthesis output after the comb filter
     * (the newest 960 samples of the history ring), the IMDCT overlap tail carried to the next frame, energies, rng */
    int32_t syn_post[2][960];
    int32_t overlap_tail[2][60];
    int16_t state_bandE[42], state_logE1[42], state_logE2[42], pad1;
    uint32_t state_rng;
    int32_t pf_period, pf_gain_state, pf_tapset_state;
    /* k_silk_parse's record (SILK-only and hybrid frames; silk_valid = 0 otherwise): dequantised parameters per coded channel */
    int32_t silk_valid, silk_ret, decode_only_middle, ms_pred_q13[2];
    struct {
        int32_t pitchL[4], Gains_Q16[4];
        int16_t PredCoef_Q12[2][16];
        int16_t LTPCoef_Q14[20];
        int32_t LTP_scale_Q14, signalType, quantOffsetType;
    } silk_ch[2];
    /* k_silk_synth: the channels' state after the step -- the synthesis core's output at the internal rate (for 20 ms frames
     * the output history IS the frame), the LPC state */
    int16_t silk_out[2][320];
    int32_t silk_sLPC_Q14[2][16];
    int32_t silk_fs_kHz[2];
} opusgpu_stage_taps;
int opusgpu_debug_stage_taps(opusgpu_ctx *ctx, int slot, opusgpu_stage_taps *out);

/* ---- Ogg page ingest at scale (host only, no GPU involved) --------------------------------------
 * The batched equivalent of what the reference does one page at a time: page sync + header checks
 * (ogg_sync_pageseek src/ogg.cpp:839-923), the page CRC (ogg_page_checksum_set :439-480), lacing values -> packets
 * (ogg_stream_pagein / ogg_stream_packetout :969-1097, :1192; op_collect_audio_packets src/opusfile.cpp:424-466) and
 * the TOC split of every packet (opusgpu_packet_to_frames above).  Input: n complete Ogg pages, each tagged by the
 * caller with the decoder stream it belongs to (the caller owns the serial-number -> stream mapping).  Output: decode
 * steps.  Step k holds one descriptor per page that has a k-th 20 ms frame, ready for opusgpu_decode_step_device after
 * the arena and the step's table are copied to the device; frames of one page land in consecutive steps, and a second
 * page of the same stream in the same call starts where the first one ends, so a stream never appears twice in a step.
 * A page must carry whole packets: one that starts with the tail or ends with the head of a spanning packet is
 * reported (OPUSGPU_PAGE_SPANS) and contributes nothing; the file surface (opusfile.h) handles such streams.
 * Every frame becomes its own step entry: if a frame of a multi-frame packet fails on the device (in practice a CELT or
 * hybrid frame of at most one byte, which the reference's CELT decoder rejects), the later frames of that packet are
 * still decoded.  opus_decode_native -- and opusgpu_decode_packets -- stop at a packet's first failing frame. */
#define OPUSGPU_PAGE_BAD_CAPTURE (-200) /* no "OggS", stream structure version != 0, or shorter than its header says */
#define OPUSGPU_PAGE_BAD_CRC (-201)     /* only with OPUSGPU_PAGES_VERIFY_CRC */
#define OPUSGPU_PAGE_SPANS (-202)
#define OPUSGPU_PAGE_BAD_PACKET (-203)  /* a packet with a valid duration fails the frame split (opus_packet_parse_impl): the
                                           reference decodes the page's earlier packets and then reports the error; here
                                           the whole page is dropped.  Packets whose TOC sequence has no valid duration
                                           are skipped like the reference does (src/opusfile.cpp:453-459). */
#define OPUSGPU_PAGE_BAD_STREAM (-204)  /* negative stream id */

#define OPUSGPU_PAGES_VERIFY_CRC 1
#define OPUSGPU_PAGES_GROUP_BY_MODE 2 /* order each step's table SILK-only, hybrid, CELT-only (stable): uniform waves */
#define OPUSGPU_PAGES_ORDER_BY_HEADER 4 /* (implies the grouping) within the SILK-only and the hybrid group: by the frames' LBRR flags -- bits 6 and
                                          4 of a frame's first byte (src/silk.cpp:1568-1573) -- stable otherwise: the 32 frames of a parse wave
                                          then agree on how many frames of forward-error-correction data they read past (:1590-1616) */

typedef struct opusgpu_page_info { /* 32 bytes */
    int32_t status;      /* >= 0: 20 ms frames the page contributes; < 0: OPUSGPU_PAGE_* */
    int32_t packets;     /* packets on the page */
    int32_t first_step;  /* step of the page's first frame */
    int32_t header_type; /* bit 0 continued, bit 1 first page of the stream, bit 2 last page */
    uint32_t serial, seqno;
    int64_t granulepos;
} opusgpu_page_info;

typedef struct opusgpu_page_batch opusgpu_page_batch; /* owns the step tables and the packet arena of one demux call */

/* threads <= 0: one.  info (n_pages entries) may be NULL.  Returns OPUSGPU_OK (bad pages are reported per page, not as
 * a failure of the call), OPUSGPU_BAD_ARG or OPUSGPU_ALLOC_FAIL. */
int opusgpu_pages_demux(int n_pages, const uint8_t *const *pages, const int32_t *page_lens, const int32_t *stream_ids,
                        int flags, int threads, opusgpu_page_info *info, opusgpu_page_batch **out);
/* The same demux with the step tables and the packet arena placed in the CALLER's memory (16-byte aligned; page-locked memory makes
 * the batch uploadable as it lies, in one copy): out_mem = [descriptors of all steps, step after step | padding to a multiple of 256
 * | arena + 16 bytes].  *out_need (may be NULL) receives the bytes needed; OPUSGPU_BUFFER_TOO_SMALL when out_cap is less (nothing is
 * written then; n_pages * 16 * 255 + the pages' bytes + 512 always suffices).  The batch object still owns the per-slot page
 * indices; opusgpu_page_batch_step / _arena point into out_mem, which must outlive the batch.  Descriptor offsets index the arena,
 * which begins opusgpu_page_batch_arena_offset(batch) bytes into out_mem. */
int opusgpu_pages_demux_into(int n_pages, const uint8_t *const *pages, const int32_t *page_lens, const int32_t *stream_ids,
                             int flags, int threads, opusgpu_page_info *info, void *out_mem, size_t out_cap, size_t *out_need,
                             opusgpu_page_batch **out);
size_t opusgpu_page_batch_arena_offset(const opusgpu_page_batch *b);
int opusgpu_page_batch_steps(const opusgpu_page_batch *b);
/* Step `step`: returns its descriptor count and points *descs at the table and *slot_pages (may be NULL) at the index
 * of the input page each descriptor came from (the PCM block of slot s belongs to page (*slot_pages)[s]). */
int opusgpu_page_batch_step(const opusgpu_page_batch *b, int step, const opusgpu_frame_desc **descs,
                            const int32_t **slot_pages);
const uint8_t *opusgpu_page_batch_arena(const opusgpu_page_batch *b, size_t *bytes); /* descriptor offsets index this */
void opusgpu_page_batch_free(opusgpu_page_batch *b);

/* Page checksums on the GPU ("optional GPU CRC"): for pages that are already in HBM -- e.g. raw pages
 * delivered by the work-queue scatter -- the page CRC of ogg_page_checksum_set (src/ogg.cpp:439-480) is recomputed by a
 * kernel, one page per lane, and compared with the stored one.  d_blob: the pages' bytes; d_offsets (int64[n_pages]) and
 * d_lens (int32[n_pages]): where page i lies in d_blob; d_status (int32[n_pages]) receives 1 = checksum matches,
 * 0 = mismatch, OPUSGPU_PAGE_BAD_CAPTURE = not a complete Ogg page (capture pattern, version, lengths: the checks of
 * opusgpu_pages_demux).  The caller guarantees that every [offset, offset + len) lies inside d_blob.  A host demux of
 * pages verified this way can drop OPUSGPU_PAGES_VERIFY_CRC.  Asynchronous on the context's stream (or `hip_stream`). */
int opusgpu_pages_crc_device(opusgpu_ctx *ctx, int n_pages, const void *d_blob, const void *d_offsets, const void *d_lens,
                             void *d_status, void *hip_stream);

/* Output stage of the player: decoded PCM -> the 32-bit words src/main.cpp hands to the I2S peripheral.
 * Replaces playChunk (src/main.cpp:148-224), playSample (:226-256) and Gain (:137-146) for a whole step at once: every
 * block of d_pcm is one m_outBuff.  Per output frame: the two samples are picked by bit depth / channel count /
 * force-mono, 8-bit samples are expanded ((x - 128) << 8), both are halved for headroom (>> 1), scaled
 * ((s * volume) >> 6) and packed as (right << 16) | (left & 0xffff).  The OpusHead output gain is NOT applied -- the
 * reference does not apply it either (op_update_gain is commented out, src/opusfile.cpp:704). */
typedef struct opusgpu_output_cfg {
    uint8_t volume;     /* m_vol (src/main.cpp:39): 64 = unity; above 127 the 16-bit halves wrap as they do there */
    uint8_t force_mono; /* m_f_forceMono: both channels play (left + right) / 2 */
    uint8_t bits;       /* setBitsPerSample: 16, or 8 (every int16 of the block holds two unsigned 8-bit samples) */
    uint8_t channels;   /* setChannels: 2 (interleaved), or 1 */
} opusgpu_output_cfg;

/* Block b: int16 samples at d_pcm + b * pcm_stride (stride in int16 units; a decode step's PCM has stride
 * 2 * 960 * frame_capacity), m_validSamples = d_valid[b] (int32; e.g. the step's result array: a negative entry plays
 * nothing) or valid_all when d_valid is NULL, at most block_samples.  Settings: d_cfgs[b] (device array) or `cfg` when
 * d_cfgs is NULL; `cfg` with other bits / channels than the setters accept is OPUSGPU_BAD_ARG, such a d_cfgs entry makes
 * its block play nothing (as playChunk does).  Output: uint32 words at d_i2s + b * i2s_stride, `valid` of them -- 2 * valid
 * for 8-bit mono, so i2s_stride must hold 2 * block_samples there; words past a block's count are left untouched.
 * 16-byte aligned bases with pcm_stride % 8 == 0 and i2s_stride % 4 == 0 take the fast path; anything else works too.
 * Asynchronous on the context's stream (or `hip_stream`). */
int opusgpu_output_stage_device(opusgpu_ctx *ctx, int n_blocks, int block_samples, const void *d_pcm, long long pcm_stride,
                                const void *d_valid, int valid_all, const void *d_cfgs, opusgpu_output_cfg cfg, void *d_i2s,
                                long long i2s_stride, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
