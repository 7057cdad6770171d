// og_api.hip -- HIP kernels and the C ABI (include/opusgpu.h) of libopusgpu.so, gfx950 only.
//
// Launch shape: one workgroup == one wavefront (64 threads) == one 20 ms frame of one stream.  A
// decode step over n streams launches n workgroups (n >> 256 CUs x 8 waves for the BASELINE configs),
// each touching only its own stream record, so there is no inter-workgroup communication at all and
// block -> XCD placement cannot matter for correctness; the per-stream records are private, so L2
// affinity is not a concern either.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <new>
#include <mutex>
#include <thread>
#include <vector>
#include "og_decode.hpp"
#include "og_packet.hpp"
#include "og_output.hpp"
#include "og_debug.hpp"

using namespace og;

static_assert(sizeof(opusgpu_frame_desc) == sizeof(FrameDesc), "descriptor layout");
enum { OPUSGPU_COPY_PIECES = 16, OPUSGPU_COPY_THREADS = 8 };

// ---- kernels --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_stream_init(StreamState *st, int first, int count, int channels, int full) {
    const int s = first + (int)blockIdx.x;
    if ((int)blockIdx.x >= count) return;
    if (full)
        stream_init(&st[s], channels);
    else
        stream_reset(&st[s]);
}

// Ogg page checksums (ogg_page_checksum_set src/ogg.cpp:439-480: CRC-32, polynomial 0x04c11db7, MSB first, zero start,
// no final inversion, the four checksum bytes taken as zero), ONE PAGE PER LANE, eight 256-entry tables in LDS.
// A lane walking its own page with its own loads would make every wave-level load touch 64 cache lines and come back to
// each line eight times, long after L1 has dropped it (measured: 0.55 TB/s).  Instead the wave moves whole 64-byte lines:
// per step, four lanes fetch one line of one page (a load instruction covers 16 pages) into an LDS tile, then every lane
// consumes its own 64-byte row eight bytes at a time.  The chunks are the MEMORY's 64-byte lines, not the page's: a
// zero-start CRC ignores leading zero bytes, so a page that starts z bytes into a line is taken as z zeros followed by the
// page; every 16-byte load is aligned (pages ending on line boundaries measured 18-25 % faster than pages at arbitrary
// offsets when the chunks were counted from the page's end instead), no load passes the 16-byte block that holds the
// page's last byte, and only the last chunk of a page is partial (eight-byte steps, then at most seven single bytes).
// status: 1 match, 0 mismatch, OPUSGPU_PAGE_BAD_CAPTURE malformed.
// Chunk size and prefetch depth were measured (786,432 pages, shuffled / grouped sizes; the same
// build varies by up to 10 % between runs on the grouped input, so only the last line is a real difference):
//   64-byte chunks, one chunk ahead   2.70 - 2.72 / 3.08 - 3.45 TB/s   HBM traffic (FETCH_SIZE) 1.53 / 1.69 x the page bytes
//   64-byte chunks, two chunks ahead  2.60 / 3.19
//   128-byte chunks, two ahead        2.40 / 2.93        traffic 1.18 x (the second half of a 128-byte line is no longer
//                                                        fetched again after the L2 dropped it), but 12 instead of 20
//                                                        waves per CU, and the table lookups in LDS are what binds
enum {
    CRC_CHUNK = 64,               // bytes of a page per step
    CRC_DEPTH = 1,                // chunks requested ahead of the one being worked on (1 or 2)
    CRC_LPL = CRC_CHUNK / 16,     // lanes per line: each fetches 16 bytes
    CRC_PPI = 64 / CRC_LPL,       // pages covered by one load instruction of the wave
    CRC_NL = 64 / CRC_PPI,        // load instructions per chunk of the wave's 64 pages
    CRC_ROW = CRC_CHUNK / 4 + 1,  // tile row stride in words (+ 1 word: conflict-free column walks)
    CRC_STEPS = CRC_CHUNK / 8     // slice-by-8 steps per full chunk
};
__global__ void __launch_bounds__(256) k_pages_crc(const u8 *__restrict__ blob, const long long *__restrict__ offs,
                                                    const i32 *__restrict__ lens, i32 *__restrict__ status, int n,
                                                    const u32 *__restrict__ tables) {
    __shared__ u32 T[8][256];
    __shared__ u32 tile[4][64 * CRC_ROW];
    __shared__ long long pg_start[4][64]; // blob offset of the first (virtual) byte of the page's first chunk: offs - z
    __shared__ i32 pg_meta[4][64];        // z | chunks << 8
    for (int i = (int)threadIdx.x; i < 8 * 256; i += 256) T[i >> 8][i & 255] = tables[i];
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int p = (int)blockIdx.x * 256 + (int)threadIdx.x;
    // ---- per lane: is this a complete page, and how long is it?
    int total = 0, st = OPUSGPU_PAGE_BAD_CAPTURE;
    u32 want = 0;
    long long at0 = 0;
    if (p < n) {
        // The lanes' pages lie far apart, so every load here is a memory round trip of its own: the 27 header bytes come as
        // two 16-byte loads and the lacing values 16 at a time (byte loads only where a wide one could pass the page's end).
        const u8 *pg = blob + offs[p];
        const int len = lens[p];
        if (len >= 27) {
            u32 h[8];
            if (len >= 32) {
                const uint4 a = *reinterpret_cast<const uint4 *>(pg), b = *reinterpret_cast<const uint4 *>(pg + 16);
                h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
            } else {
                for (int i = 0; i < 8; i++) h[i] = 0;
                for (int i = 0; i < 27; i++) h[i >> 2] |= (u32)pg[i] << (8 * (i & 3));
            }
            const int nseg = (int)(h[6] >> 16) & 255, hdr = 27 + nseg; // byte 26
            if (h[0] == 0x5367674fu /* "OggS" */ && (h[1] & 255u) == 0 && len >= hdr) {
                u32 body = 0;
                for (int i = 0; i < nseg; i += 16) {
                    u32 v[4];
                    if (27 + i + 16 <= len) {
                        const uint4 q = *reinterpret_cast<const uint4 *>(pg + 27 + i);
                        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                    } else {
                        v[0] = v[1] = v[2] = v[3] = 0;
                        for (int b = 0; b < 16 && i + b < nseg; b++) v[b >> 2] |= (u32)pg[27 + i + b] << (8 * (b & 3));
                    }
                    const int keep = nseg - i; // lacing values in this group: the rest of the 16 bytes is page body
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const int kd = keep - 4 * d;
                        const u32 m = kd >= 4 ? 0xffffffffu : kd <= 0 ? 0u : (1u << (8 * kd)) - 1u;
                        body = __builtin_amdgcn_sad_u8(v[d] & m, 0u, body); // sum of the four bytes
                    }
                }
                if (len >= hdr + (int)body) {
                    total = hdr + (int)body;
                    st = 0;
                    want = h[5] >> 16 | h[6] << 16; // bytes 22 .. 25, little-endian
                }
            }
        }
        at0 = offs[p];
    }
    // z: where in its 64-byte line the page starts (by ADDRESS: the blob itself may start anywhere)
    const int z = st < 0 ? 0 : (int)((reinterpret_cast<unsigned long long>(blob) + (unsigned long long)at0) & (unsigned long long)(CRC_CHUNK - 1));
    const int nfull = (z + total) / CRC_CHUNK, tail = (z + total) % CRC_CHUNK, chunks = nfull + (tail != 0);
    pg_start[wv][lane] = at0 - z;
    pg_meta[wv][lane] = z | total << 8; // total <= 27 + 255 + 255 * 255
    __syncthreads();
    int max_chunks = chunks;
    for (int d = 32; d; d >>= 1) max_chunks = max(max_chunks, __shfl_xor(max_chunks, d, 64));
    u32 crc = 0;
    const int grp = lane / CRC_LPL, quarter = lane % CRC_LPL; // this lane fetches 16-byte part `quarter` of the line of page CRC_PPI i + grp
    // chunk k of the wave's 64 pages into registers: CRC_LPL lanes per line, CRC_NL lines per lane
    auto fetch = [&](int k, u32 (&w)[CRC_NL][4]) {
#pragma unroll
        for (int i = 0; i < CRC_NL; i++) {
            const int q = CRC_PPI * i + grp;
            const i32 meta = pg_meta[wv][q];
            const int idx = CRC_CHUNK * k + 16 * quarter - (meta & 255); // page byte index of the quarter's first byte
            w[i][0] = w[i][1] = w[i][2] = w[i][3] = 0u;
            if (idx > -16 && idx < (meta >> 8)) { // the quarter holds at least one byte of the page
                const uint4 v = *reinterpret_cast<const uint4 *>(
                    __builtin_assume_aligned(blob + (pg_start[wv][q] + CRC_CHUNK * k + 16 * quarter), 16));
                w[i][0] = v.x; w[i][1] = v.y; w[i][2] = v.z; w[i][3] = v.w;
                if (idx < 26) { // near the page start: what precedes the page counts as zeros; so do bytes 22 .. 25 (the checksum)
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        u32 m = 0;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int ib = idx + 4 * d + c;
                            if (ib >= 0 && !(ib >= 22 && ib <= 25)) m |= 0xffu << (8 * c);
                        }
                        w[i][d] &= m;
                    }
                }
            }
        }
    };
    // one chunk of the wave's pages from registers through the tile into the lanes' checksums
    auto consume = [&](int k, u32 (&w)[CRC_NL][4]) {
#pragma unroll
        for (int i = 0; i < CRC_NL; i++) {
            u32 *dst = &tile[wv][(CRC_PPI * i + grp) * CRC_ROW + 4 * quarter];
            dst[0] = w[i][0]; dst[1] = w[i][1]; dst[2] = w[i][2]; dst[3] = w[i][3];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the registers are free again: request a later chunk before this one is worked on, so that its memory latency
        // runs under the CRC work
        if (k + CRC_DEPTH < max_chunks) fetch(k + CRC_DEPTH, w);
        // ---- every lane consumes its own row: all of it, or (last chunk) the bytes up to the page's end
        const u32 *rowp = &tile[wv][lane * CRC_ROW];
        const int steps = k < nfull ? CRC_STEPS : k == nfull ? tail >> 3 : 0;
#pragma unroll
        for (int j = 0; j < CRC_STEPS; j++) {
            if (j < steps) {
                const u32 lo = rowp[2 * j], hi = rowp[2 * j + 1]; // message bytes b0 .. b3 | b4 .. b7, first byte lowest
                const u32 a = crc ^ ((lo & 0xffu) << 24 | (lo & 0xff00u) << 8 | (lo >> 8 & 0xff00u) | lo >> 24);
                crc = T[7][a >> 24] ^ T[6][(a >> 16) & 255] ^ T[5][(a >> 8) & 255] ^ T[4][a & 255] ^ T[3][hi & 255] ^
                      T[2][(hi >> 8) & 255] ^ T[1][(hi >> 16) & 255] ^ T[0][hi >> 24];
            }
        }
        if (k == nfull && (tail & 7)) {
            const int base = tail & ~7;
            for (int b = 0; b < (tail & 7); b++) {
                const int i = base + b;
                crc = crc << 8 ^ T[0][crc >> 24 ^ (rowp[i >> 2] >> (8 * (i & 3)) & 255u)];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    u32 wa[CRC_NL][4], wb[CRC_NL][4]; // chunks on their way (wb: only with CRC_DEPTH == 2)
    if (max_chunks > 0) fetch(0, wa);
    if (CRC_DEPTH == 2) {
        if (max_chunks > 1) fetch(1, wb);
        for (int k = 0; k < max_chunks; k += 2) {
            consume(k, wa);
            if (k + 1 < max_chunks) consume(k + 1, wb);
        }
    } else {
        for (int k = 0; k < max_chunks; k++) consume(k, wa);
    }
    if (p < n) status[p] = st < 0 ? st : (i32)(crc == want);
}

// Output stage (og_output.hpp): a thread makes four consecutive I2S words of one block.  Plain streaming work: 4 bytes in,
// 4 bytes out per word in the usual 16-bit stereo case, which takes the 16-byte path when the caller's layout allows it.
static_assert(sizeof(opusgpu_output_cfg) == sizeof(OutputCfg) && sizeof(OutputCfg) == 4, "output cfg layout");
__global__ void __launch_bounds__(256) k_output_stage(const i16 *__restrict__ pcm, long long pcm_stride, const i32 *__restrict__ valid_of,
                                                       int valid_all, int block_samples, const OutputCfg *__restrict__ cfg_of,
                                                       OutputCfg cfg_all, u32 *__restrict__ out, long long out_stride,
                                                       int units_per_block, long long n_units, int vec_ok) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_units) return;
    const int b = (int)(t / units_per_block), w0 = 4 * (int)(t - (long long)b * units_per_block);
    const OutputCfg c = cfg_of ? cfg_of[b] : cfg_all;
    int valid = valid_of ? valid_of[b] : valid_all; // a decode result: negative = the frame failed, nothing to play
    valid = valid > block_samples ? block_samples : valid;
    const int count = output_words(c, valid);
    if (w0 >= count) return;
    const i16 *blk = pcm + (size_t)b * pcm_stride;
    u32 *dst = out + (size_t)b * out_stride + w0;
    if (vec_ok && c.bits == 16 && c.channels == 2 && w0 + 4 <= count) {
        const uint4 v = *reinterpret_cast<const uint4 *>(blk + 2 * w0); // four stereo samples
        const u32 in[4] = {v.x, v.y, v.z, v.w};
        u32 o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            i32 l = (i32)(i16)(in[j] & 0xffffu), r = (i32)(i16)(in[j] >> 16);
            if (c.force_mono) l = r = (i32)(i16)((l + r) / 2);
            o[j] = output_pack(l, r, false, (i32)c.volume);
        }
        *reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        for (int j = 0; j < 4 && w0 + j < count; j++) dst[j] = output_word(blk, w0 + j, c);
    }
}

#ifndef OG_WAVES_PER_SIMD
#define OG_WAVES_PER_SIMD 1
#endif
#ifndef OG_RECON_WAVES
#define OG_RECON_WAVES 2
#endif
#ifndef OG_SILK_WAVES
#define OG_SILK_WAVES 2
#endif
__global__ void __launch_bounds__(64, OG_WAVES_PER_SIMD) k_decode_step(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena,
                                                   StreamState *st, i16 *pcm, i32 *result, int n, int n_streams,
                                                   int pcm_stride, int skip_celt, SilkHandoff *handoff, const SilkRec *srecs,
                                                   int q4_only) {
    // First pass (or the only one): one workgroup per slot.  Second pass of the split path (q4_only): it almost never has a
    // frame, so a workgroup looks at 64 slots -- one per lane -- and decodes the few that were parked for it one after the
    // other: 1 / 64 of the workgroups to start and end for nothing.
    unsigned long long todo = 1ull;
    int base = (int)blockIdx.x;
    if (q4_only) {
        base = (int)blockIdx.x * 64;
        const int f0 = base + (int)threadIdx.x;
        bool mine = false;
        if (f0 < n) {
            const FrameDesc d0 = descs[f0];
            mine = d0.stream >= 0 && d0.stream < n_streams && !desc_rfc(d0.flags) && desc_mode(d0.flags) == MODE_SILK && handoff[f0].valid == 2;
        }
        todo = __ballot(mine);
    }
    while (todo) {
        const int f = base + (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        if (f >= n) return;
        const FrameDesc d = descs[f];
        int ret;
        if (d.stream < 0 || d.stream >= n_streams) {
            ret = BAD_ARG;
        } else if (desc_rfc(d.flags)) {
            continue; // RFC-mode frames belong to k_decode_rfc (og_rfc.hip)
        } else if (skip_celt && desc_mode(d.flags) == MODE_CELT) {
            continue; // CELT-only frames take the split path (k_celt_parse + k_celt_recon)
        } else {
            StreamState *s = &st[d.stream];
#ifdef OG_PROF_SINGLE // profiling builds: time the sections of the single-kernel path
            OG_PROF_INIT();
#endif
            ret = decode_frame_wave<true>(s, arena + d.offset, d.len, desc_mode(d.flags), desc_bandwidth(d.flags),
                                          desc_channels(d.flags), pcm + (size_t)f * pcm_stride, handoff ? &handoff[f] : nullptr,
                                          srecs ? &srecs[f] : nullptr, q4_only, desc_mode_after(d.flags));
#ifdef OG_PROF_SINGLE
            OG_PROF_FLUSH();
#endif
            if (ret == CONTINUE_SPLIT) continue; // the split path finishes this frame and reports its result
        }
        if (threadIdx.x == 0) result[f] = ret;
        __syncthreads();
    }
}

// SILK-only and hybrid frames on the split path, arithmetic half: k_silk_synth (og_silk_synth.hip) and, for narrowband SILK-only
// frames, k_silk_synth_nb (og_silk_nb.hip) -- translation units of their own: their LDS working set has a layout of its own
extern "C" void og_launch_silk_synth(hipStream_t s, const void *descs, const void *arena, void *streams, void *pcm, void *result, int n,
                                     int n_streams, int pcm_stride, void *handoff, const void *srecs, int nb_elsewhere);
extern "C" void og_launch_silk_synth_nb(hipStream_t s, const void *descs, const void *arena, void *streams, void *pcm, void *result, int n,
                                        int n_streams, int pcm_stride, void *handoff, const void *srecs, int nb_elsewhere);

// SILK-only and hybrid frames, entropy half: ONE FRAME PER LANE (og_silk_parse.hpp).  Lane l < LANES of workgroup g decodes the side
// information and pulses of frame LANES g + l into srecs[frame] and leaves the coder state in handoff[frame].  Twice, like the CELT
// parse: k_silk_parse with 32 frames per wave (the upper lanes idle) for small in-order steps, k_silk_parse64 with 64 for pipelined
// steps and large batches (og_silk_parse.hpp, OG_SP_LANES).
#ifndef OG_SPARSE_WAVES
#define OG_SPARSE_WAVES 4
#endif
// `shadow` (null in in-order steps): the per-stream copies of what the entropy half needs of the past, kept by this kernel for
// pipelined SILK / hybrid steps (SilkShadow, og_silk_parse.hpp); `epoch`: the context's current one.
template <int LANES>
OG_DEV void silk_parse_kernel_body(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena, const StreamState *st, SilkRec *srecs,
                                   SilkHandoff *handoff, int n, int n_streams, SilkShadow *shadow, u32 epoch) {
    silk_tables_load();
    if ((int)threadIdx.x >= LANES) return;
    const int f = (int)blockIdx.x * LANES + (int)threadIdx.x;
    if (f >= n) return;
    const FrameDesc d = descs[f];
    const int mode = desc_mode(d.flags);
    if (d.stream < 0 || d.stream >= n_streams || mode == MODE_CELT || desc_rfc(d.flags)) return;
#ifdef OG_PROF_SPARSE // profiling builds: time the sections of the SILK parse kernel (full batches only)
    OG_PROF_INIT();
#endif
    SilkShadow *const sh = shadow ? &shadow[d.stream] : nullptr;
    const SilkPast past(&st[d.stream], sh, epoch);
    silk_parse_lane(past, arena + d.offset, d.len, mode, desc_bandwidth(d.flags), desc_channels(d.flags), &srecs[f], &handoff[f], sh, epoch,
                    desc_mode_after(d.flags));
#ifdef OG_PROF_SPARSE
    OG_PROF_FLUSH();
#endif
}
__global__ void __launch_bounds__(64, OG_SPARSE_WAVES) k_silk_parse(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena,
                                                      const StreamState *st, SilkRec *srecs, SilkHandoff *handoff, int n,
                                                      int n_streams, SilkShadow *shadow, u32 epoch) {
    silk_parse_kernel_body<32>(descs, arena, st, srecs, handoff, n, n_streams, shadow, epoch);
}
__global__ void __launch_bounds__(64, OG_SPARSE_WAVES) k_silk_parse64(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena,
                                                        const StreamState *st, SilkRec *srecs, SilkHandoff *handoff, int n,
                                                        int n_streams, SilkShadow *shadow, u32 epoch) {
    silk_parse_kernel_body<64>(descs, arena, st, srecs, handoff, n, n_streams, shadow, epoch);
}

// ... and their parameter half (silk_decode_parameters), ONE (FRAME, CHANNEL) PER LANE: lane l of workgroup g takes channel l / 32
// of frame 32 g + l % 32 -- the record's indices in, the dequantised parameters out, and for pipelined steps the entropy half's
// past of the stream's next frame (`shadow`).  Behind k_silk_parse on the same stream.
#ifndef OG_SPARAMS_WAVES
#define OG_SPARAMS_WAVES 4
#endif
__global__ void __launch_bounds__(64, OG_SPARAMS_WAVES) k_silk_params(const FrameDesc *__restrict__ descs, const StreamState *st, SilkRec *srecs,
                                                                       int n, int n_streams, SilkShadow *shadow, u32 epoch) {
    constexpr int FR = OG_PAR_LANES / 2;
    const int ch = (int)threadIdx.x / FR, f = (int)blockIdx.x * FR + (int)threadIdx.x % FR;
    bool act = f < n;
    FrameDesc d = {0, 0, 0, 0};
    if (act) {
        d = descs[f];
        act = !(d.stream < 0 || d.stream >= n_streams || desc_mode(d.flags) == MODE_CELT || desc_rfc(d.flags));
    }
    SilkShadow *const sh = act && shadow ? &shadow[d.stream] : nullptr;
    const SilkParPast past(act ? &st[d.stream] : st, sh, epoch);
    const int mode = desc_mode(d.flags), channels = desc_channels(d.flags);
    SilkParTask t;
    t.skip = 1;
    if (act) silk_params_channel(past, mode, desc_bandwidth(d.flags), channels, &srecs[f], ch, t);
    OG_FULL_SYNC(); // every lane has read what it needs of the past: now the lanes of a frame may overwrite it
    if (act) silk_params_shadow(past, t, channels, &srecs[f], ch, sh, epoch);
}

#define OG_PARSE_KERNEL_NAME k_celt_parse
#include "og_parse_kernel.hpp"
// ... and with 64 frames per wave, for pipelined steps (og_parse64.hip)
extern "C" void og_launch_celt_parse64(hipStream_t s, int grid, const void *descs, const void *arena, void *streams, void *recs, int n,
                                       int n_streams, const void *handoff, int which, int groups, unsigned *started);
extern "C" int og_celt_parse64_frames(void); // frames per group of that kernel

// Split CELT path, second half: one frame per wave, driven by the parse record.
// (20 ms CELT-only frames are reconstructed by k_celt_recon_fb, og_recon.hip; `rest_only`: skip what that kernel took)
extern "C" void og_launch_celt_recon_fb(hipStream_t s, const void *descs, void *streams, const void *recs, void *rout, int n,
                                        int n_streams, int hybrid, unsigned *started);
extern "C" int og_celt_recon_fb_signals(int n); // how often a launch over n frames bumps `started`
// RFC mode (opt-in): every frame of a step, at its true duration, incl. the loss path (og_rfc.hip)
extern "C" void og_launch_decode_rfc(hipStream_t s, const void *descs, const void *arena, void *streams, void *pcm, void *result, int n,
                                     int n_streams, int pcm_stride);
__global__ void __launch_bounds__(64, OG_RECON_WAVES) k_celt_recon(const FrameDesc *__restrict__ descs, StreamState *st,
                                                                      const ParseRec *recs, ReconOut *rout, int n,
                                                                      int n_streams, int hybrid, int rest_only) {
    // rest_only (k_celt_recon_fb ran before): what is left -- records that overflowed -- is almost nothing, so a workgroup
    // looks at 64 slots, one per lane, and reconstructs the few left to it one after the other (see k_decode_step)
    unsigned long long todo = 1ull;
    int base = (int)blockIdx.x;
    if (rest_only) {
        base = (int)blockIdx.x * 64;
        const int f0 = base + (int)threadIdx.x;
        bool mine = false;
        if (f0 < n) {
            const FrameDesc d0 = descs[f0];
            const int m0 = desc_mode(d0.flags);
            if (d0.stream >= 0 && d0.stream < n_streams && (m0 == MODE_CELT || (m0 == MODE_HYBRID && hybrid)) && !desc_rfc(d0.flags)) {
                const u32 fl = recs[f0].flags; // (the lane's own look at recon_fast_eligible's conditions)
                const bool fast = !(fl & (RF_SKIP | RF_BAD_CELT)) && ((fl >> RF_LM_SHIFT) & 3) == 3 && recs[f0].n_words < REC_MAX_WORDS &&
                                  recs[f0].n_leaves <= FAST_MAX_LEAVES;
                mine = !fast && !(m0 == MODE_HYBRID && (fl & RF_SKIP));
            }
        }
        todo = __ballot(mine);
    }
    while (todo) {
        const int f = base + (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        if (f >= n) return;
        const FrameDesc d = descs[f];
        const int mode = desc_mode(d.flags);
        if (d.stream < 0 || d.stream >= n_streams || !(mode == MODE_CELT || (mode == MODE_HYBRID && hybrid)) || desc_rfc(d.flags)) continue;
        if (mode == MODE_HYBRID && (recs[f].flags & RF_SKIP)) continue; // the single-kernel path already reported this frame
#if !defined(OG_PROF_PARSE) && !defined(OG_PROF_SINGLE) && !defined(OG_PROF_SPARSE) && !defined(OG_PROF_SSYNTH)
        OG_PROF_INIT();
#endif
        const int pos = OG_UNI(st[d.stream].celt.ring_pos); // where the frame's first sample goes
        const int ret = celt_recon_wave(&st[d.stream], &recs[f], mode, desc_channels(d.flags), rest_only ? RECON_REST_ONLY : RECON_ALL, desc_mode_after(d.flags));
        if (ret != RECON_NOT_MINE && threadIdx.x == 0) rout[f] = ReconOut{ret, pos};
#if !defined(OG_PROF_PARSE) && !defined(OG_PROF_SINGLE) && !defined(OG_PROF_SPARSE) && !defined(OG_PROF_SSYNTH)
        OG_PROF_FLUSH();
#endif
        __syncthreads();
    }
}

// Split CELT path, third step: de-emphasis (a rounding IIR: strictly serial per channel) and int16 PCM, one
// (frame, channel) per lane, from the samples k_celt_recon appended to the history ring.
// It also hands the reconstruction's result codes (ReconOut) to the caller's array, and -- for a step whose caller named the
// modes it contains (`modes`: bit 0 SILK-only, bit 1 hybrid, bit 2 CELT-only; the kernels of absent modes were not launched,
// `others_ran` = 0 if that includes the kernels that report stream-index errors) -- reports what those kernels would have.
// two int16 lanes of a word added with saturation (v_pk_add_i16 ... clamp)
static __device__ __forceinline__ i32 pk_add_sat_i16(i32 a, i32 b) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b));
    return __builtin_bit_cast(i32, r);
}
__global__ void __launch_bounds__(64) k_celt_post(const FrameDesc *__restrict__ descs, StreamState *st, const ParseRec *recs,
                                                  const ReconOut *__restrict__ rout, i32 *__restrict__ result, i16 *pcm, int n,
                                                  int n_streams, int channels, int pcm_stride, const SilkHandoff *handoff,
                                                  int modes, int others_ran) {
    // Fast path (two-channel decoder, every row of the wave live, ring positions on a 32-sample boundary): the 64 rows'
    // next 32 samples are fetched as full 128-byte lines by the whole wave (lane = an eighth of a row's line), transposed
    // through LDS to one row per lane for the recurrence, and the PCM goes out the same way (lane = 16 bytes of a frame's
    // interleaved output, a frame's 128 bytes by eight neighbouring lanes; a hybrid frame's SILK PCM comes in likewise).  Anything
    // else takes the row-per-lane path with its 16-byte accesses (celt_post_lane).
    // (Round 3 moved 64-byte pieces: the memory system fetches 128-byte lines, and the other half of a row's line had left the L2
    // again by the time its turn came -- 36 KB of traffic per hybrid frame for 15 KB needed, in a kernel that does nothing but move.)
    struct RowInfo {
        const i32 *ring;
        i16 *pcm;
        const i16 *silk;
        int pos, silk_n;
    };
    constexpr int CS = 32, PCS = CS / 4, RSTEP = 64 / PCS; // samples per chunk and row; 16-byte pieces per row; rows between a lane's pieces
    __shared__ RowInfo rows[64];
    // (one buffer is enough: a chunk's rows are all read before the barrier that follows the PCM staging, the next chunk's are
    // written behind it)
    __shared__ __attribute__((aligned(16))) i32 tin[64][CS + 4];  // CS samples per row, rows padded by 16 bytes
    __shared__ __attribute__((aligned(16))) i16 tout[64][CS + 8]; // CS outputs per row, rows padded by 16 bytes
    // (16 KB per workgroup next to kernels that are short of LDS: measured by padding, 5 KB more cost the CELT step 0.8 %, mixed pages 0.6 %)
    const int lane = (int)threadIdx.x;
    const int t = (int)blockIdx.x * 64 + lane;
    const int f = channels == 2 ? t >> 1 : t, c = channels == 2 ? t & 1 : 0;
    bool live = false, emit = false;
    StreamState *ss = nullptr;
    const i16 *silk = nullptr;
    int silk_n = 0;
    int pos = 0;
    if (f < n) {
        const FrameDesc d = descs[f];
        const int mode = desc_mode(d.flags);
        const bool stream_ok = d.stream >= 0 && d.stream < n_streams;
        if (!(modes >> (d.flags & 3) & 1) || (!stream_ok && !others_ran)) { // (a frame the caller's mode set left out is an error, not a skip)
            if (c == 0) result[f] = BAD_ARG;
        } else if (stream_ok && (mode == MODE_CELT || (mode == MODE_HYBRID && handoff)) && !desc_rfc(d.flags)) {
            const u32 rf = recs[f].flags;
            if (!(mode == MODE_HYBRID && (rf & RF_SKIP))) { // (those the single-kernel path has reported already)
                const ReconOut ro = rout[f];
                if (c == 0) result[f] = ro.ret;
                if (!(rf & (RF_SKIP | RF_BAD_CELT))) {
                    live = true;
                    ss = &st[d.stream];
                    emit = ro.ret >= 0;
                    pos = ro.pos & RING_MASK;
                    if (mode == MODE_HYBRID) {
                        silk = handoff[f].pcm;
                        silk_n = 960 * desc_channels(d.flags);
                    }
                }
            }
        }
    }
    i16 *out = pcm + (size_t)(f < n ? f : 0) * pcm_stride;
    const bool fast = channels == 2 && __all(live && emit && (pos & (CS - 1)) == 0);
    if (!fast) {
        if (live) celt_post_lane(&ss->celt, c, channels, 960, pos, emit ? out : nullptr, silk, silk_n);
        return;
    }
    rows[lane].ring = ss->celt.ring[c];
    rows[lane].pcm = out;
    rows[lane].silk = silk;
    rows[lane].pos = pos;
    rows[lane].silk_n = silk_n;
    __syncthreads();
    // this lane's share of the cooperative traffic: piece q of rows r0, r0 + RSTEP, ..
    const int q = lane % PCS, r0 = lane / PCS;
    const i32 *src[PCS];
    int spos[PCS];
#pragma unroll
    for (int k = 0; k < PCS; k++) {
        src[k] = rows[r0 + RSTEP * k].ring;
        spos[k] = rows[r0 + RSTEP * k].pos + 4 * q;
    }
    og_v4i v[PCS];
#pragma unroll
    for (int k = 0; k < PCS; k++) v[k] = *reinterpret_cast<const og_v4i *>(src[k] + (spos[k] & RING_MASK));
    i32 m = ss->celt.deemph[c];
    for (int ch = 0; ch < 960 / CS; ch++) {
#pragma unroll
        for (int k = 0; k < PCS; k++) *reinterpret_cast<og_v4i *>(&tin[r0 + RSTEP * k][4 * q]) = v[k];
        if (ch + 1 < 960 / CS) {
#pragma unroll
            for (int k = 0; k < PCS; k++) v[k] = *reinterpret_cast<const og_v4i *>(src[k] + ((spos[k] + CS * (ch + 1)) & RING_MASK));
        }
        __syncthreads();
        // the recurrence on this lane's own row (celt.cpp:1965-2055, sig2word16 celt.h:413)
#pragma unroll
        for (int g8 = 0; g8 < CS / 8; g8++) {
            i16 o[8];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const og_v4i sv = *reinterpret_cast<const og_v4i *>(&tin[lane][8 * g8 + 4 * h]);
                // (a hybrid frame's SILK PCM is added where the PCM leaves, below: there a frame's pieces are read by neighbouring
                // lanes as whole lines; read here, one row per lane, every 16 bytes would come from a line of their own)
                i32 tt = sv.x + m;
                m = mul16x32_q15(27853, tt);
                o[4 * h + 0] = (i16)sat16(pshr32(tt, 12));
                tt = sv.y + m;
                m = mul16x32_q15(27853, tt);
                o[4 * h + 1] = (i16)sat16(pshr32(tt, 12));
                tt = sv.z + m;
                m = mul16x32_q15(27853, tt);
                o[4 * h + 2] = (i16)sat16(pshr32(tt, 12));
                tt = sv.w + m;
                m = mul16x32_q15(27853, tt);
                o[4 * h + 3] = (i16)sat16(pshr32(tt, 12));
            }
            og_v4i w;
            w.x = (i32)((u32)(u16)o[0] | (u32)(u16)o[1] << 16);
            w.y = (i32)((u32)(u16)o[2] | (u32)(u16)o[3] << 16);
            w.z = (i32)((u32)(u16)o[4] | (u32)(u16)o[5] << 16);
            w.w = (i32)((u32)(u16)o[6] | (u32)(u16)o[7] << 16);
            *reinterpret_cast<og_v4i *>(&tout[lane][8 * g8]) = w;
        }
        __syncthreads();
        // PCM: frame fr's CS samples x 2 channels = 128 contiguous bytes; this lane writes piece q (samples 4q .. 4q+3) of frames
        // r0, r0 + RSTEP, ..
#pragma unroll
        for (int k = 0; k < PCS / 2; k++) {
            const int fr = r0 + RSTEP * k; // frame within the wave: rows 2 fr (left) and 2 fr + 1 (right)
            const og_v2u L = *reinterpret_cast<const og_v2u *>(&tout[2 * fr][4 * q]);
            const og_v2u R = *reinterpret_cast<const og_v2u *>(&tout[2 * fr + 1][4 * q]);
            og_v4i w;
            w.x = (i32)((L.x & 0xffffu) | R.x << 16);
            w.y = (i32)(L.x >> 16 | (R.x & 0xffff0000u));
            w.z = (i32)((L.y & 0xffffu) | R.y << 16);
            w.w = (i32)(L.y >> 16 | (R.y & 0xffff0000u));
            const int at = (CS * ch + 4 * q) * 2; // the piece's place in the frame's interleaved PCM -- and in its SILK PCM (Q3: by linear index)
            const i16 *const sk = rows[2 * fr].silk;
            if (sk && at < rows[2 * fr].silk_n) { // SAT16(celt + silk), two samples per saturating packed add
                const og_v4i a = *reinterpret_cast<const og_v4i *>(sk + at);
                w.x = pk_add_sat_i16(w.x, a.x);
                w.y = pk_add_sat_i16(w.y, a.y);
                w.z = pk_add_sat_i16(w.z, a.z);
                w.w = pk_add_sat_i16(w.w, a.w);
            }
            *reinterpret_cast<og_v4i *>(rows[2 * fr].pcm + at) = w;
        }
    }
    ss->celt.deemph[c] = m;
}

// ---- context ----------------------------------------------------------------------------------------
// Worker threads of the host-buffer path, started once per context: a large opusgpu_decode_packets call hands them ranges of its
// packets four times (scan, place per part, delivery); starting 16 threads each time cost 0.6 ms per hand-over at 65,536 packets.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable wake, done;
    std::function<void(int)> job;
    int generation = 0, want = 0, pending = 0;
    bool quit = false;
    ~HostPool() {
        {
            std::lock_guard<std::mutex> l(m);
            quit = true;
        }
        wake.notify_all();
        for (auto &t : th) t.join();
    }
    void worker(int id) {
        int seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> l(m);
                wake.wait(l, [&] { return quit || (generation != seen && id < want); });
                if (quit) return;
                seen = generation;
                f = job;
            }
            f(id);
            {
                std::lock_guard<std::mutex> l(m);
                if (--pending == 0) done.notify_all();
            }
        }
    }
    // f(t) for t = 0 .. count - 1, t = 0 on the calling thread; returns when all are through
    void run(int count, const std::function<void(int)> &f) {
        if (count <= 1) {
            f(0);
            return;
        }
        while ((int)th.size() < count - 1) {
            const int id = (int)th.size();
            th.emplace_back([this, id] { worker(id); });
        }
        {
            std::lock_guard<std::mutex> l(m);
            job = [&f](int id) { f(id + 1); };
            want = count - 1;
            pending = count - 1;
            generation++;
        }
        wake.notify_all();
        f(0);
        std::unique_lock<std::mutex> l(m);
        done.wait(l, [&] { return pending == 0; });
    }
};

#ifndef OG_SILK_SETS
#define OG_SILK_SETS 3 // sets of SILK records and hand-offs that pipelined SILK / hybrid steps rotate through
#endif
struct opusgpu_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    StreamState *d_streams = nullptr;
    int n_streams = 0, channels = 0;
    // staging for the host-buffer path
    void *d_descs = nullptr, *d_arena = nullptr, *d_pcm = nullptr, *d_result = nullptr;
    size_t cap_descs = 0, cap_arena = 0, cap_pcm = 0, cap_result = 0;
    // pinned host landing zone of the host-buffer path's PCM and result codes (DMA at full PCIe rate, no zero-filling of
    // a fresh temporary per call); the caller's pageable buffer is filled from it by a few host threads
    void *h_pcm = nullptr, *h_res = nullptr;
    size_t cap_h_pcm = 0, cap_h_res = 0;
    // ... and of the way in: the call's packet bytes and step table are gathered in page-locked memory that lives as long as the
    // context (a fresh 10 MB allocation per call is 2,600 page faults in front of the first upload, and a copy from pageable
    // memory holds the calling thread until the runtime has staged it)
    void *h_arena = nullptr, *h_descs = nullptr;
    size_t cap_h_arena = 0, cap_h_descs = 0;
    HostPool pool;
    u32 *d_crc_tables = nullptr; // 8 x 256 words, made on first use (opusgpu_pages_crc_device)
    hipEvent_t ev_piece[OPUSGPU_COPY_PIECES] = {}; // one per piece of the PCM's way back to the host (opusgpu_decode_packets)
    // large batches on the host-buffer path run in parts: a part's PCM travels back (on a stream of its own) while the next
    // part's kernels run
    hipStream_t copy_stream = nullptr;
    std::mutex registered_mutex;
    std::vector<std::pair<uintptr_t, size_t>> registered; // host ranges page-locked through opusgpu_host_register
    hipEvent_t ev_part[OPUSGPU_COPY_PIECES] = {};
    int host_parts = 8; // OPUSGPU_HOST_PARTS=1: one batch, copy after the kernels (A/B measurements); 2, 4, 8, 16
    // parse records of the split CELT path (one per frame of a step), grown on demand
    // (five sets: pipelined CELT-only steps rotate through 0 - 2 -- in-order steps use 0 --, pipelined SILK-only / hybrid steps
    // alternate 3 and 4: steps of the two kinds may be in flight together, OPUSGPU_STEP_KEEPS_MODE)
    void *d_recs[6] = {}, *d_rout[6] = {}; // (sets 0 - 2: pipelined CELT-only steps and everything in order; 3 - 5: pipelined SILK / hybrid steps)
    size_t cap_recs[6] = {}, cap_rout[6] = {};
    // (OG_SILK_SETS sets: pipelined SILK / hybrid steps rotate; everything else uses set 0.  Three since round 5: with two the parse
    // of step k + 1 had to wait for the synthesis of step k - 1 to let go of its set, and the chain parse -> parameters of a small
    // step -- 0.58 + 0.45 ms at 65,536 SILK-NB frames -- was then longer than the synthesis it should have hidden under)
    void *d_handoff[OG_SILK_SETS] = {}, *d_srecs[OG_SILK_SETS] = {};
    size_t cap_handoff[OG_SILK_SETS] = {}, cap_srecs[OG_SILK_SETS] = {};
    const void *last_srecs = nullptr; // the SILK records of the last step (opusgpu_debug_stage_taps)
    // Pipelined SILK-only steps (a step the caller declares SILK-only): the parse kernel keeps what its next run needs of the past
    // in d_shadow (SilkShadow per stream, og_silk_parse.hpp) and runs for step k + 1 on parse_stream next to step k's synthesis.
    void *d_shadow = nullptr;
    unsigned shadow_epoch = 1; // advanced by everything else that may change a stream's SILK state: stale copies are ignored
    int silk_slot = 0, sdone_recorded[OG_SILK_SETS] = {}, last_silk_mask = 0, last_kind = 0; // last_kind: 0 in order, 1 pipelined CELT-only, 2 pipelined SILK-only
    bool last_kind2_celt = false; // the last step of kind 2 held CELT-only frames too (enter_step_kind)
    hipEvent_t ev_sparsed = nullptr, ev_sdone[OG_SILK_SETS] = {}, ev_sp = nullptr, ev_spar = nullptr, ev_hrecon = nullptr; // ev_sp: a step's SILK parse is done; ev_spar: its parameter half; ev_hrecon: its CELT reconstruction
    int split_celt = 1;   // OPUSGPU_SPLIT=0 forces the single-kernel path for every mode (A/B measurements)
    int split_hybrid = 1; // OPUSGPU_SPLIT_HYBRID=0 keeps SILK-only and hybrid frames entirely on the single-kernel path
    int fast_recon = 1;   // OPUSGPU_FAST_RECON=0: every CELT frame through the general reconstruction kernel (A/B measurements)
    int mode = OPUSGPU_MODE_REFERENCE; // opusgpu_set_mode
    // opusgpu_set_pipeline: the parse of step k + 1's CELT-only frames runs on parse_stream, next to step k's reconstruction
    // and its reconstruction on recon_stream; parse records and the reconstruction's per-frame output (d_recs, d_rout) rotate
    int pipeline = 0, slot = 0, front_recorded = 0, post_recorded[3] = {};
    hipStream_t parse_stream = nullptr, recon_stream = nullptr, last_step_stream = nullptr;
    hipStream_t side_stream = nullptr;                // in-order steps with SILK frames: the second half's chain (decode_step_impl)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_front = nullptr;  // step k: its front kernels have finished (on the step's stream)
    hipEvent_t ev_parsed = nullptr; // step k: its early parse has finished (on parse_stream)
    hipEvent_t ev_recon = nullptr;  // step k: its reconstruction has finished (on recon_stream)
    hipEvent_t ev_post[3] = {};     // by slot: k_celt_post of the last step that used it has finished (on the step's stream)
    const void *last_recs = nullptr;
    // Steps queued as a window (opusgpu_decode_steps_device): the kernels of neighbouring steps are placed in the order that
    // works -- the next step's parse, then this step's reconstruction, then the de-emphasis of the step before -- by stream
    // memory waits on two counters the parse / reconstruction workgroups bump when they start (device words, 64 bytes apart;
    // the host keeps the totals they will reach).  Round 2 got that order from a spin-wait kernel watching the wall clock.
    u32 *d_started = nullptr; // [0] early-parse workgroups started, [16] every 64th reconstruction workgroup started
    u32 parse_started_total = 0, recon_started_total = 0;
    u32 window_parse_target = 0, window_recon_target = 0; // the counts the last queued step of an unfinished window waits for (0: none)
    // (OPUSGPU_PARSE_GROUPS) groups of frames per workgroup of the early parse, one after the other.  Round 2 measured two as the
    // best (half as many parse workgroups resident for twice as long: 2.545 / 2.50 / 2.52 / 2.97 ms per step at 1 / 2 / 3 / 4).  Round 4:
    // a group takes a parse wave 0.85 ms, so two groups are a chain of 1.7 ms -- which had become the step (a reconstruction doing
    // 40 % of its work: still 1.69 ms).  With one group the parse is done after 1.0 ms of the step and what counts is how many of the
    // reconstruction's waves fit a CU next to it: the parse kernel's LDS went from 46 KB to 36 KB per 128 frames for that
    // (og_celt_split.hpp: ParseLds), 1.83 -> 1.74 ms.
    int parse_groups = 1;
    // the last decode step's tables, for opusgpu_debug_stage_taps
    const void *last_descs = nullptr;
    int last_n = 0, last_had_silk_recs = 0;
    // RFC mode, host side of the loss path: per stream, the frame count and descriptor flags of the last packet framed by
    // opusgpu_decode_packets -- what a lost packet of that stream is concealed as (0 frames: nothing framed yet)
    std::vector<int32_t> last_count, last_flags;
    char err[256] = {0};
};

static int fail(opusgpu_ctx *ctx, int code, const char *what, hipError_t e) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, hipGetErrorString(e));
    return code;
}
#define HIPCHK(ctx, call)                                                \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return fail(ctx, OPUSGPU_ERR_HIP, #call, e_); \
    } while (0)

// OPUSGPU_HOST_TIMING=1: wall time of the phases of opusgpu_decode_packets on stderr (adds a stream synchronise after the
// kernels so that decode and copy-back can be told apart; for tuning only)
struct HostPhaseTimer {
    bool on, light; // on: OPUSGPU_HOST_TIMING=1, the one-batch flow with a wait after the kernels; light (=2): the flow as it is
    std::chrono::steady_clock::time_point t;
    HostPhaseTimer() : on(og_debug().host_timing == 1), light(og_debug().host_timing == 2), t(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on && !light) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[opusgpu_decode_packets] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

extern "C" {

static int grow(opusgpu_ctx *ctx, void **p, size_t *cap, size_t need);

int opusgpu_version(void) { return 100; }

int opusgpu_ctx_create(int device, opusgpu_ctx **out) {
    if (!out || device < 0) return OPUSGPU_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) return OPUSGPU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return OPUSGPU_ERR_NO_DEVICE;
    // the code object is gfx950-only: make sure the kernel image is loadable on this device
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k_decode_step)) != hipSuccess) {
        (void)hipGetLastError();
        return OPUSGPU_ERR_NO_DEVICE;
    }
    opusgpu_ctx *ctx = new (std::nothrow) opusgpu_ctx();
    if (!ctx) return OPUSGPU_ALLOC_FAIL;
    ctx->device = device;
    ctx->split_celt = og_debug().split; // (og_debug.hpp: A/B switches, read from the environment once per process)
    ctx->split_hybrid = og_debug().split_hybrid;
    ctx->fast_recon = og_debug().fast_recon;
    ctx->parse_groups = og_debug().parse_groups;
    ctx->host_parts = og_debug().host_parts;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return OPUSGPU_ERR_HIP;
    }
    *out = ctx;
    return OPUSGPU_OK;
}

void opusgpu_ctx_destroy(opusgpu_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_streams);
    (void)hipFree(ctx->d_descs);
    (void)hipFree(ctx->d_arena);
    (void)hipFree(ctx->d_pcm);
    (void)hipFree(ctx->d_result);
    if (ctx->parse_stream) (void)hipStreamSynchronize(ctx->parse_stream);
    if (ctx->recon_stream) (void)hipStreamSynchronize(ctx->recon_stream);
    for (int i = 0; i < 6; i++) {
        (void)hipFree(ctx->d_recs[i]);
        (void)hipFree(ctx->d_rout[i]);
        if (i < 3 && ctx->ev_post[i]) (void)hipEventDestroy(ctx->ev_post[i]);
    }
    for (int i = 0; i < OG_SILK_SETS; i++) {
        (void)hipFree(ctx->d_handoff[i]);
        (void)hipFree(ctx->d_srecs[i]);
        if (ctx->ev_sdone[i]) (void)hipEventDestroy(ctx->ev_sdone[i]);
    }
    if (ctx->ev_sparsed) (void)hipEventDestroy(ctx->ev_sparsed);
    if (ctx->ev_sp) (void)hipEventDestroy(ctx->ev_sp);
    if (ctx->ev_spar) (void)hipEventDestroy(ctx->ev_spar);
    if (ctx->ev_hrecon) (void)hipEventDestroy(ctx->ev_hrecon);
    (void)hipFree(ctx->d_shadow);
    (void)hipHostFree(ctx->h_pcm);
    (void)hipHostFree(ctx->h_res);
    (void)hipHostFree(ctx->h_arena);
    (void)hipHostFree(ctx->h_descs);
    (void)hipFree(ctx->d_crc_tables);
    (void)hipFree(ctx->d_started);
    for (hipEvent_t e : ctx->ev_piece)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_part)
        if (e) (void)hipEventDestroy(e);
    if (ctx->side_stream) {
        (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamDestroy(ctx->side_stream);
    }
    if (ctx->ev_fork) {
        (void)hipEventDestroy(ctx->ev_fork);
        (void)hipEventDestroy(ctx->ev_join);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->parse_stream) (void)hipStreamDestroy(ctx->parse_stream);
    if (ctx->recon_stream) (void)hipStreamDestroy(ctx->recon_stream);
    if (ctx->ev_front) (void)hipEventDestroy(ctx->ev_front);
    if (ctx->ev_parsed) (void)hipEventDestroy(ctx->ev_parsed);
    if (ctx->ev_recon) (void)hipEventDestroy(ctx->ev_recon);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *opusgpu_last_error(const opusgpu_ctx *ctx) { return ctx ? ctx->err : "no context"; }
int opusgpu_set_mode(opusgpu_ctx *ctx, int mode) {
    if (!ctx || (mode != OPUSGPU_MODE_REFERENCE && mode != OPUSGPU_MODE_RFC)) return OPUSGPU_BAD_ARG;
    ctx->mode = mode;
    return OPUSGPU_OK;
}
int opusgpu_get_mode(const opusgpu_ctx *ctx) { return ctx ? ctx->mode : OPUSGPU_MODE_REFERENCE; }
int opusgpu_set_pipeline(opusgpu_ctx *ctx, int on) {
    if (!ctx) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->parse_stream) {
        // the early parse is a single round of long-running workgroups: it is placed first (highest priority), the
        // reconstruction it runs next to fills the slots around it
        int least = 0, greatest = 0;
        HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (!og_debug().parse_priority) greatest = least;
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->parse_stream, hipStreamNonBlocking, greatest));
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->recon_stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_front, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_parsed, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_recon, hipEventDisableTiming));
        for (int i = 0; i < 3; i++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_post[i], hipEventDisableTiming));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_started, 128));
        HIPCHK(ctx, hipMemset(ctx->d_started, 0, 128));
    }
    if ((on != 0) != (ctx->pipeline != 0)) { // switching: from an idle device (steps of either kind may be queued on any stream)
        HIPCHK(ctx, hipDeviceSynchronize());
        ctx->front_recorded = ctx->post_recorded[0] = ctx->post_recorded[1] = ctx->post_recorded[2] = 0;
        for (int i = 0; i < OG_SILK_SETS; i++) ctx->sdone_recorded[i] = 0;
        ctx->last_kind = 0;
        ctx->shadow_epoch++;
    }
    ctx->pipeline = on ? 1 : 0;
    return OPUSGPU_OK;
}
int opusgpu_get_pipeline(const opusgpu_ctx *ctx) { return ctx ? ctx->pipeline : 0; }
size_t opusgpu_stream_state_bytes(void) { return sizeof(StreamState); }
int opusgpu_stream_count(const opusgpu_ctx *ctx) { return ctx ? ctx->n_streams : 0; }
int opusgpu_stream_channels(const opusgpu_ctx *ctx) { return ctx ? ctx->channels : 0; }

int opusgpu_streams_reset(opusgpu_ctx *ctx, int first, int count, int full) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->n_streams) return OPUSGPU_BAD_ARG;
    if (count == 0) return OPUSGPU_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->parse_stream) { // whatever pipelined steps still have in flight works on the state this resets
        HIPCHK(ctx, hipStreamSynchronize(ctx->parse_stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->recon_stream));
        if (ctx->last_step_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->last_step_stream));
    }
    ctx->front_recorded = ctx->post_recorded[0] = ctx->post_recorded[1] = ctx->post_recorded[2] = 0; // (the reset below is synchronous)
    for (int i = 0; i < OG_SILK_SETS; i++) ctx->sdone_recorded[i] = 0;
    ctx->shadow_epoch++; // (the parse kernel's copies of these streams' past are stale now)
    hipLaunchKernelGGL(k_stream_init, dim3(count), dim3(64), 0, ctx->stream, ctx->d_streams, first, count, ctx->channels,
                       full ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // (both kinds of reset forget the last packet -- src/opus_decoder.cpp:382-390 clears frame_size and the like -- a loss right
    // after one conceals 20 ms, of zeros)
    for (int i = first; i < first + count; i++) ctx->last_count[i] = ctx->last_flags[i] = 0;
    return OPUSGPU_OK;
}

int opusgpu_streams_alloc(opusgpu_ctx *ctx, int n_streams, int channels) {
    if (!ctx || n_streams <= 0 || (channels != 1 && channels != 2)) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->parse_stream) { // pipelined steps still in flight work on the state that is about to be freed
        HIPCHK(ctx, hipStreamSynchronize(ctx->parse_stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->recon_stream));
        if (ctx->last_step_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->last_step_stream));
    }
    if (ctx->d_streams) {
        HIPCHK(ctx, hipFree(ctx->d_streams));
        ctx->d_streams = nullptr;
        ctx->n_streams = 0;
    }
    hipError_t e = hipMalloc((void **)&ctx->d_streams, sizeof(StreamState) * (size_t)n_streams);
    if (e != hipSuccess) return fail(ctx, OPUSGPU_ALLOC_FAIL, "hipMalloc(streams)", e);
    if (ctx->d_shadow) HIPCHK(ctx, hipFree(ctx->d_shadow));
    ctx->d_shadow = nullptr;
    e = hipMalloc(&ctx->d_shadow, sizeof(SilkShadow) * (size_t)n_streams);
    if (e != hipSuccess) return fail(ctx, OPUSGPU_ALLOC_FAIL, "hipMalloc(shadow)", e);
    HIPCHK(ctx, hipMemset(ctx->d_shadow, 0, sizeof(SilkShadow) * (size_t)n_streams)); // (epoch 0: never current)
    ctx->shadow_epoch++;
    ctx->n_streams = n_streams;
    ctx->channels = channels;
    ctx->last_count.assign((size_t)n_streams, 0);
    ctx->last_flags.assign((size_t)n_streams, 0);
    return opusgpu_streams_reset(ctx, 0, n_streams, 1);
}

int opusgpu_dev_alloc(opusgpu_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // (16 bytes more than asked for: an arena made with this call then has the tail the kernels' 16-byte packet fetches may touch)
    if (bytes > SIZE_MAX - 16) return OPUSGPU_ALLOC_FAIL;
    hipError_t e = hipMalloc(dptr, bytes + 16);
    if (e != hipSuccess) return fail(ctx, OPUSGPU_ALLOC_FAIL, "hipMalloc", e);
    return OPUSGPU_OK;
}
int opusgpu_dev_free(opusgpu_ctx *ctx, void *dptr) {
    if (!ctx) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipFree(dptr));
    return OPUSGPU_OK;
}
int opusgpu_memcpy_h2d(opusgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return OPUSGPU_OK;
}
// Whatever pipelined steps still have in flight -- the last step's reconstruction on recon_stream, an early parse, the step's
// own stream when the caller supplied one -- works on state, records and ReconOut: a read-back waits for all of it, not only
// for the context's stream.
static int sync_in_flight(opusgpu_ctx *ctx) {
    if (ctx->parse_stream) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->parse_stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->recon_stream));
    }
    if (ctx->last_step_stream && ctx->last_step_stream != ctx->stream) HIPCHK(ctx, hipStreamSynchronize(ctx->last_step_stream));
    return OPUSGPU_OK;
}
int opusgpu_memcpy_d2h(opusgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_in_flight(ctx)) return rc;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return OPUSGPU_OK;
}

// `tables_resident`: the step's descriptors and payload bytes are complete in device memory now (the public entry's contract
// when pipelining is on); false when this call's own uploads are still queued on the step's stream (opusgpu_decode_packets):
// such a step does not run ahead of anything.
// OPUSGPU_LAUNCH_DELAY_US (og_debug.hpp): the host dawdles before the launches of a decode step -- what a loaded host, a slow
// event hop or another thread's launches would do -- so that tools/launch_jitter.py can show the step time does not depend on it
#ifndef OG_SILK_PARSE_WIDE_MIN
#define OG_SILK_PARSE_WIDE_MIN 98304 // frames of an in-order launch from which the SILK parse runs with 64 frames per wave
#endif
#ifndef OG_HALVES_MIN
#define OG_HALVES_MIN 4096 // frames per half below which an in-order step is not cut in two
#endif
static void launch_jitter() {
    if (const int us = og_debug().launch_delay_us) std::this_thread::sleep_for(std::chrono::microseconds(us));
}
// `next_n` (steps queued as a window, opusgpu_decode_steps_device): the number of frames of the step that the same call queues
// right behind this one with the same mode mask, 0 when there is none or it is not known.
// `slices` (opusgpu_decode_packets: PCM that leaves in pieces): the step's entropy kernels run once over all n frames -- they wait
// on latency, a fraction of the frames takes them as long as all -- and the arithmetic kernels slice by slice, frames
// [bounds[i], bounds[i + 1]); after_slice(i) is called behind slice i's last launch (to queue that slice's copies).
// A step is of one of three kinds: in order (0), pipelined CELT-only (1), pipelined SILK-only (2).  Going into or out of a run of
// pipelined SILK-only steps happens from an idle device (their parse reads the stream state when it has no current copy of its
// own, and whatever follows them reads what their last kernels write); every step of another kind ends the epoch of the parse
// kernel's copies (it may write SILK state, or prev_mode, behind that kernel's back).
// keeps_kind (OPUSGPU_STEP_KEEPS_MODE): the caller's word that no stream of this step has decoded a frame of another mode (SILK-only,
// hybrid, CELT-only) since its last reset.  Such a step shares no stream with anything of the other kind that is still in flight,
// so going from one pipelined kind to the other needs no drain, and a CELT-only step leaves the SILK parse kernel's copies (of
// other streams) as current as they were.
// ... as long as the two kinds really are about different streams: a kind-2 step that carries CELT-only frames along (any mix under
// OPUSGPU_STEP_KEEPS_MODE, `celt_frames`) reconstructs them on ITS stream, ordered only against other kind-2 steps, while a kind-1
// step's reconstruction runs on recon_stream and waits only for kind-1 steps.  Next to each other the two would work on the same
// CELT-only streams' state with nothing in between: that change of kind drains like an undeclared one.
static int enter_step_kind(opusgpu_ctx *ctx, int kind, hipStream_t s, bool keeps_kind = false, bool celt_frames = false) {
    const bool shares_celt = (kind == 1 && ctx->last_kind == 2 && ctx->last_kind2_celt) || (kind == 2 && celt_frames && ctx->last_kind == 1);
    const bool disjoint = keeps_kind && kind != 0 && ctx->last_kind != 0 && !shares_celt;
    if ((kind == 2) != (ctx->last_kind == 2) && !disjoint) {
        if (int rc = sync_in_flight(ctx)) return rc;
        HIPCHK(ctx, hipStreamSynchronize(s));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->front_recorded = ctx->post_recorded[0] = ctx->post_recorded[1] = ctx->post_recorded[2] = 0;
        for (int i = 0; i < OG_SILK_SETS; i++) ctx->sdone_recorded[i] = 0;
        ctx->last_silk_mask = 0;
    }
    if (kind != 2 && !(keeps_kind && kind == 1)) ctx->shadow_epoch++;
    ctx->last_kind = kind;
    if (kind == 2) ctx->last_kind2_celt = celt_frames;
    return OPUSGPU_OK;
}

// Records and reconstruction output of slot `par` for a step of `need` frames.
static int grow_step_slot(opusgpu_ctx *ctx, int par, size_t need) {
    int rc;
    if (ctx->cap_recs[par] < sizeof(ParseRec) * need && (rc = grow(ctx, &ctx->d_recs[par], &ctx->cap_recs[par], sizeof(ParseRec) * need)))
        return rc;
    if (ctx->cap_rout[par] < sizeof(ReconOut) * need && (rc = grow(ctx, &ctx->d_rout[par], &ctx->cap_rout[par], sizeof(ReconOut) * need)))
        return rc;
    return OPUSGPU_OK;
}

// A window that ends early (a HIP error between two of its steps): the placement waits of the last step queued -- for workgroups
// of a step that will not come -- are let go by writing the counts they wait for; the waits are placement only (events carry the
// data dependencies), so whatever is queued completes.  Then the device drains and the counters restart from zero.
static void release_window_waits(opusgpu_ctx *ctx) {
    if (!ctx->d_started || (!ctx->window_parse_target && !ctx->window_recon_target)) return;
    hipStream_t q = nullptr;
    if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess) {
        if (ctx->window_parse_target) (void)hipStreamWriteValue32(q, ctx->d_started, ctx->window_parse_target, 0);
        if (ctx->window_recon_target) (void)hipStreamWriteValue32(q, ctx->d_started + 16, ctx->window_recon_target, 0);
        (void)hipStreamSynchronize(q);
        (void)hipStreamDestroy(q);
    }
    (void)sync_in_flight(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemset(ctx->d_started, 0, 128);
    ctx->parse_started_total = ctx->recon_started_total = 0;
    ctx->window_parse_target = ctx->window_recon_target = 0;
    (void)hipGetLastError();
}

struct StepSlices {
    int count = 0;
    const size_t *bounds = nullptr;
    std::function<int(int)> after_slice;
};
static int decode_step_impl(opusgpu_ctx *ctx, int n, const void *d_descs, const void *d_arena, void *d_pcm, void *d_result,
                            void *hip_stream, bool tables_resident, int modes = 7, int next_n = 0, const StepSlices *slices = nullptr) {
    if (!ctx || n < 0 || !ctx->d_streams) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!d_descs || !d_arena || !d_pcm || !d_result) return OPUSGPU_BAD_ARG;
    if ((uintptr_t)d_arena & 15) return OPUSGPU_BAD_ARG; // (the parse kernels fetch packets as aligned 16-byte pieces, og_range.hpp)
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    const int pcm_stride = (ctx->mode == OPUSGPU_MODE_RFC ? OPUSGPU_RFC_FRAME_SAMPLES : OPUSGPU_FRAME_SAMPLES) * ctx->channels;
    ctx->last_descs = d_descs;
    ctx->last_n = ctx->mode == OPUSGPU_MODE_RFC || !ctx->split_celt ? 0 : n;
    ctx->last_had_silk_recs = ctx->split_celt && ctx->split_hybrid;
    if (ctx->mode == OPUSGPU_MODE_RFC) { // every frame on the one kernel of that mode (og_rfc.hip)
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if (int rc = enter_step_kind(ctx, 0, s)) return rc;
        og_launch_decode_rfc(s, d_descs, d_arena, ctx->d_streams, d_pcm, d_result, n, ctx->n_streams, pcm_stride);
        HIPCHK(ctx, hipGetLastError());
        if (ctx->pipeline) { // (a later pipelined step's early parse waits for all of this one)
            HIPCHK(ctx, hipEventRecord(ctx->ev_front, s));
            ctx->front_recorded = 1;
        }
        ctx->last_step_stream = s;
        return OPUSGPU_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->split_celt) { // OPUSGPU_SPLIT=0 (A/B measurements): every frame through the single kernel, in order
        if (int rc = enter_step_kind(ctx, 0, s)) return rc;
        hipLaunchKernelGGL(k_decode_step, dim3(n), dim3(64), 0, s, (const FrameDesc *)d_descs, (const u8 *)d_arena, ctx->d_streams,
                           (i16 *)d_pcm, (i32 *)d_result, n, ctx->n_streams, pcm_stride, 0, nullptr, nullptr, 0);
        HIPCHK(ctx, hipGetLastError());
        ctx->last_step_stream = s;
        if (slices) // (one kernel for the whole step: every slice's PCM is there behind it)
            for (int i = 0; i < slices->count; i++)
                if (int rc = slices->after_slice(i)) return rc;
        return OPUSGPU_OK;
    }
    // `modes` (bit 0 SILK-only, 1 hybrid, 2 CELT-only frames may be present; 7 = not known): the kernels of modes the caller
    // rules out are not launched; k_celt_post reports a frame of such a mode as OPUSGPU_BAD_ARG
    const bool keeps_kind = (modes & OPUSGPU_STEP_KEEPS_MODE) != 0;
    modes &= 7;
    if (!modes) modes = 7;
    const bool any_silk = (modes & 3) != 0, any_celt = (modes & 6) != 0;
    // Only a step the caller declares CELT-only runs ahead of the step before it.  (Round 2 also ran the CELT-only part of a
    // mixed step's parse ahead: 4 % on the mixed-pages workload.  Cutting such a step into two halves -- below -- gains 6 %, and
    // the two do not combine: a mixed or undeclared step takes the halves.)
    const bool pipe = ctx->pipeline && tables_resident && modes == 4;
    const bool window = pipe && next_n > 0; // the next step is queued by this very call: see PLACEMENT
    // ... and a step declared free of CELT-only frames runs its parse kernels ahead (PIPELINED SILK / HYBRID STEPS below)
    // -- or, with the caller's word that no stream of the step ever changes its mode (OPUSGPU_STEP_KEEPS_MODE), a step of ANY mix with
    // SILK-only / hybrid frames in it: its CELT-only frames' parse carries the band energies like a pipelined CELT-only step's, and a
    // CELT-only frame cannot make another stream's SILK copy stale
    const bool pipe_silk = ctx->pipeline && tables_resident && ((modes & 4) == 0 || keeps_kind) && (modes & 3) != 0 && ctx->split_hybrid && !slices &&
                           (modes == 1 ? og_debug().silk_pipeline : og_debug().hybrid_pipeline);
    if (int rc = enter_step_kind(ctx, pipe ? 1 : pipe_silk ? 2 : 0, s, keeps_kind, (modes & 4) != 0)) return rc;
    if (ctx->pipeline && ctx->last_step_stream && ctx->last_step_stream != s) {
        // consecutive steps on different streams: nothing orders them but the caller, so nothing may run ahead either
        HIPCHK(ctx, hipStreamSynchronize(ctx->last_step_stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->parse_stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->recon_stream));
        ctx->front_recorded = ctx->post_recorded[0] = ctx->post_recorded[1] = ctx->post_recorded[2] = 0;
    }
    ctx->last_step_stream = s;
    // The records, the reconstruction's per-frame output and the hand-off buffers only grow; growing frees the old one, which
    // waits for the device to go idle.  Records and reconstruction output exist twice: pipelined steps alternate.
    if (pipe) ctx->slot = (ctx->slot + 1) % 3;
    if (pipe_silk) ctx->silk_slot = (ctx->silk_slot + 1) % OG_SILK_SETS;
    const int sset = pipe_silk ? ctx->silk_slot : 0; // the set of SILK records and hand-offs this step uses (and of CELT records with them)
    const int par = pipe ? ctx->slot : pipe_silk ? 3 + sset : 0, par2 = (par + 1) % 3; // this step's slot; (pipelined CELT-only steps:) the slot of the step two before it
    if (int rc = grow_step_slot(ctx, par, (size_t)n)) return rc; // (a window's slots were sized before its first launch)
    if (ctx->split_hybrid && any_silk) {
        int rc;
        if (ctx->cap_handoff[sset] < sizeof(SilkHandoff) * (size_t)n || ctx->cap_srecs[sset] < sizeof(SilkRec) * (size_t)n) {
            if (pipe_silk) { // (growing frees: not under kernels of the other set that are still in flight)
                if ((rc = sync_in_flight(ctx))) return rc;
                HIPCHK(ctx, hipStreamSynchronize(s));
            }
            if (ctx->cap_handoff[sset] < sizeof(SilkHandoff) * (size_t)n &&
                (rc = grow(ctx, &ctx->d_handoff[sset], &ctx->cap_handoff[sset], sizeof(SilkHandoff) * (size_t)n)))
                return rc;
            if (ctx->cap_srecs[sset] < sizeof(SilkRec) * (size_t)n &&
                (rc = grow(ctx, &ctx->d_srecs[sset], &ctx->cap_srecs[sset], sizeof(SilkRec) * (size_t)n)))
                return rc;
        }
    }
    ParseRec *const recs = (ParseRec *)ctx->d_recs[par];
    ReconOut *const rout = (ReconOut *)ctx->d_rout[par];
    SilkHandoff *const handoff = ctx->split_hybrid && any_silk ? (SilkHandoff *)ctx->d_handoff[sset] : nullptr;
    SilkRec *const srecs = ctx->split_hybrid && any_silk ? (SilkRec *)ctx->d_srecs[sset] : nullptr;
    ctx->last_recs = recs;
    ctx->last_srecs = srecs;
    ctx->last_had_silk_recs = srecs != nullptr;
    const dim3 parse_block(64 * OG_PL_WAVES);
    // The in-order chain of frames [f0, f0 + cnt) of the step, in two halves: the ENTROPY kernels (one frame per lane) ...
    // `pq` (pipelined SILK / hybrid steps): the stream the parameter half runs on -- behind this step's SILK parse and the parameter
    // half of the step before, but off the entropy chain: the parse of the next step does not wait for it (SilkShadow's two sides)
    auto front = [&](hipStream_t q, size_t f0, int cnt, SilkShadow *shadow = nullptr, u32 epoch = 0, hipStream_t pq = nullptr) {
        const FrameDesc *dd = (const FrameDesc *)d_descs + f0;
        if (srecs) {
            // (64 frames per wave where the kernel's issue slots are what counts: pipelined steps, large batches; 32 for a small
            // in-order step, whose time is the latency of one wave's serial chain -- same-box: SILK-NB in order 1.36 / 1.41 ms)
            if (shadow || cnt >= OG_SILK_PARSE_WIDE_MIN || og_debug().parse_wide == 2)
                hipLaunchKernelGGL(k_silk_parse64, dim3((cnt + 63) / 64), dim3(64), 0, q, dd, (const u8 *)d_arena,
                                   (const StreamState *)ctx->d_streams, srecs + f0, handoff + f0, cnt, ctx->n_streams, shadow, epoch);
            else
                hipLaunchKernelGGL(k_silk_parse, dim3((cnt + 31) / 32), dim3(64), 0, q, dd, (const u8 *)d_arena,
                                   (const StreamState *)ctx->d_streams, srecs + f0, handoff + f0, cnt, ctx->n_streams, shadow, epoch);
            if (pq && pq != q) {
                (void)hipEventRecord(ctx->ev_sp, q);
                (void)hipStreamWaitEvent(pq, ctx->ev_sp, 0);
            }
            hipLaunchKernelGGL(k_silk_params, dim3((cnt + OG_PAR_LANES / 2 - 1) / (OG_PAR_LANES / 2)), dim3(64), 0, pq ? pq : q, dd,
                               (const StreamState *)ctx->d_streams, srecs + f0, cnt, ctx->n_streams, shadow, epoch);
            if (pq && pq != q) (void)hipEventRecord(ctx->ev_spar, pq);
        }
        if (any_celt && ((shadow && og_debug().parse_wide) || og_debug().parse_wide == 2)) { // (a pipelined step: the wide parse, like pipelined CELT-only steps)
            const int fr = og_celt_parse64_frames();
            og_launch_celt_parse64(q, (cnt + fr - 1) / fr, dd, d_arena, ctx->d_streams, recs + f0, cnt, ctx->n_streams,
                                   handoff ? handoff + f0 : nullptr, (int)PARSE_ALL, 1, nullptr);
        } else if (any_celt)
            hipLaunchKernelGGL(k_celt_parse, dim3((cnt + OG_PL_FRAMES - 1) / OG_PL_FRAMES), parse_block, 0, q, dd, (const u8 *)d_arena,
                               ctx->d_streams, recs + f0, cnt, ctx->n_streams, (const SilkHandoff *)(handoff ? handoff + f0 : nullptr),
                               (int)PARSE_ALL, 1, (u32 *)nullptr);
    };
    // ... and the ARITHMETIC ones (one frame per wave), which also write the PCM and the result codes
    // `rq` (pipelined SILK / hybrid steps): the stream the CELT reconstruction runs on, NEXT TO the SILK synthesis instead of behind
    // it -- a hybrid frame's two halves share nothing until the de-emphasis adds them (the synthesis takes prev_mode from the record,
    // SilkRec::prev_mode), and the reconstruction of step k touches nothing the de-emphasis of step k - 1 still reads (it appends to
    // the history ring; k_celt_post reads at the position the reconstruction recorded, as in pipelined CELT-only steps)
    auto back_half = [&](hipStream_t q, size_t f0, int cnt, hipStream_t rq = nullptr) {
        const FrameDesc *dd = (const FrameDesc *)d_descs + f0;
        i16 *pp = (i16 *)d_pcm + f0 * (size_t)pcm_stride;
        i32 *rr = (i32 *)d_result + f0;
        const SilkHandoff *hh = handoff ? handoff + f0 : nullptr;
        bool others = false; // (the kernels that report stream-index errors for every mode)
        if (srecs) { // SILK-only frames and the SILK half of hybrid frames
            // (a step that may hold SILK-only frames: the narrowband ones in the kernel whose LDS is sized for them, og_silk_nb.hip)
            const int nb = (modes & 1) && og_debug().silk_nb_kernel;
            if (nb) og_launch_silk_synth_nb(q, dd, d_arena, ctx->d_streams, pp, rr, cnt, ctx->n_streams, pcm_stride, handoff + f0, srecs + f0, 1);
            og_launch_silk_synth(q, dd, d_arena, ctx->d_streams, pp, rr, cnt, ctx->n_streams, pcm_stride, handoff + f0, srecs + f0, nb);
            others = true;
        } else if (any_silk) { // every frame that is not CELT-only (OPUSGPU_SPLIT_HYBRID=0)
            hipLaunchKernelGGL(k_decode_step, dim3(cnt), dim3(64), 0, q, dd, (const u8 *)d_arena, ctx->d_streams, pp, rr, cnt, ctx->n_streams,
                               pcm_stride, 1, nullptr, nullptr, 0);
            others = true;
        }
        if (any_celt) {
            hipStream_t const r = rq ? rq : q;
            if (rq) (void)hipStreamWaitEvent(rq, ctx->ev_sparsed, 0); // (this step's CELT parse)
            if (ctx->fast_recon) og_launch_celt_recon_fb(r, dd, ctx->d_streams, recs + f0, rout + f0, cnt, ctx->n_streams, handoff ? 1 : 0, nullptr);
            hipLaunchKernelGGL(k_celt_recon, dim3(ctx->fast_recon ? (cnt + 63) / 64 : cnt), dim3(64), 0, r, dd, ctx->d_streams,
                               (const ParseRec *)(recs + f0), rout + f0, cnt, ctx->n_streams, handoff ? 1 : 0, ctx->fast_recon);
            if (rq) {
                (void)hipEventRecord(ctx->ev_hrecon, rq);
                (void)hipStreamWaitEvent(q, ctx->ev_hrecon, 0);
            }
        }
        if (any_celt || !others || modes != 7)
            hipLaunchKernelGGL(k_celt_post, dim3((cnt * ctx->channels + 63) / 64), dim3(64), 0, q, dd, ctx->d_streams, (const ParseRec *)(recs + f0),
                               (const ReconOut *)(rout + f0), rr, pp, cnt, ctx->n_streams, ctx->channels, pcm_stride, hh, modes, others ? 1 : 0);
        if (srecs && (modes & 1)) // the rare hybrid -> SILK-only transition frames (Q4), parked by k_silk_synth, through the full kernel
            hipLaunchKernelGGL(k_decode_step, dim3((cnt + 63) / 64), dim3(64), 0, q, dd, (const u8 *)d_arena, ctx->d_streams, pp, rr, cnt,
                               ctx->n_streams, pcm_stride, 1, handoff + f0, (const SilkRec *)(srecs + f0), 1);
    };
    if (pipe_silk) {
        // PIPELINED SILK / HYBRID STEPS (no CELT-only frames).  k_silk_parse waits on latency (0.9 ms of one lane's serial chain for 0.27 ms of issue time at
        // 65,536 frames), k_silk_synth is bound by issue: they fit next to each other, but within a step the second needs the
        // first.  Across steps the parse needs of step k only what step k's parse already knows -- the indices' history, the gain
        // index, the NLSFs, the rate and channel count, prev_mode: all of it entropy-side -- so it keeps a copy of its own
        // (SilkShadow) and runs for step k + 1 on parse_stream while step k's synthesis is under way on the step's stream -- for
        // hybrid frames followed by their CELT parse, which resumes its range decoder and carries the band energies itself as in
        // pipelined CELT-only steps.  Two sets of records and hand-offs alternate; the parse of step k + 1 waits for the last kernel of step k - 1 (its set's
        // last reader -- and with it for every write to the state of streams it may have no current copy of).
        if (!ctx->ev_sparsed) {
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_sparsed, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_sp, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_spar, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_hrecon, hipEventDisableTiming));
            for (int i = 0; i < OG_SILK_SETS; i++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_sdone[i], hipEventDisableTiming));
        }
        if (ctx->sdone_recorded[sset]) HIPCHK(ctx, hipStreamWaitEvent(ctx->parse_stream, ctx->ev_sdone[sset], 0));
        // One thing of the step before is not entropy-side: a SILK-only frame right behind a hybrid one (Q4) decodes a 2.5 ms CELT
        // frame in the step's LAST kernel (the full kernel's second pass), which writes the band energies a hybrid frame's CELT parse
        // predicts from.  So a step that may hold hybrid frames does not run ahead of a step that may have held SILK-only ones.
        // (a stream that keeps its mode has no such frame: OPUSGPU_STEP_KEEPS_MODE)
        const int prev_set = (sset + OG_SILK_SETS - 1) % OG_SILK_SETS; // (the step before this one)
        if (!keeps_kind && (modes & 2) && (ctx->last_silk_mask & 1) && ctx->sdone_recorded[prev_set])
            HIPCHK(ctx, hipStreamWaitEvent(ctx->parse_stream, ctx->ev_sdone[prev_set], 0));
        ctx->last_silk_mask = modes;
        // (Tried: the step's frames in chunks, the CELT parse of chunk c on the reconstruction's idle stream next to the SILK parse of
        // chunk c + 1, so that the two entropy kernels do not run one after the other: hybrid-256k 12.7 -> 13.1 / 13.3 / 18.7 ms with
        // 2 / 4 / 8 chunks.  The step is bound by what all its kernels issue together, not by the length of the entropy chain.)
        front(ctx->parse_stream, 0, n, (SilkShadow *)ctx->d_shadow, (u32)ctx->shadow_epoch, og_debug().silk_params_aside ? ctx->recon_stream : nullptr);
        HIPCHK(ctx, hipEventRecord(ctx->ev_sparsed, ctx->parse_stream));
        HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_sparsed, 0));
        if (og_debug().silk_params_aside) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_spar, 0));
        // (for steps without CELT-only frames: hybrid-256k 9.50 -> 9.35 ms; with them -- a mixed step's reconstruction is three times
        // the work -- next to the synthesis it loses: mixed pages 7.02 -> 7.22 ms)
        back_half(s, 0, n, (og_debug().hybrid_recon_aside == 2 || (og_debug().hybrid_recon_aside && (modes & 6) == 2)) ? ctx->recon_stream : nullptr);
        HIPCHK(ctx, hipEventRecord(ctx->ev_sdone[sset], s));
        ctx->sdone_recorded[sset] = 1;
        HIPCHK(ctx, hipGetLastError());
        return OPUSGPU_OK;
    }
    if (!pipe) {
        if (!slices && n >= 2 * OG_HALVES_MIN && og_debug().halves) {
            // TWO HALVES.  A step with SILK-only / hybrid frames runs in order -- k_silk_parse reads state the step's later kernels
            // write, so nothing of the next step can start early -- and its kernels are of two kinds: the lane-per-frame parse
            // kernels wait on latency with 13 % of their lanes active (k_silk_parse: 3.97 of a 15.4 ms step of 262,144 hybrid
            // frames), the wave-per-frame ones are bound by vector-instruction issue.  The frames of a step belong to different
            // streams and share nothing, so the step is cut in two and the halves' chains run on two streams: while one half's
            // synthesis fills the SIMDs the other half parses in its gaps.  No state changes hands: each half is the in-order chain
            // of its own frames over its own part of the records; the caller's stream forks the second one and joins it.
            // (CELT-only steps too since round 5: 2.155 -> 2.07 ms per step of 65,536 -- the floor of an in-order step is one lane's
            // parse, 0.85 ms whatever the batch, plus the reconstruction; only steps queued ahead hide the parse, opusgpu_set_pipeline)
            if (!ctx->ev_fork) {
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            // (the second chain's stream: the one pipelined steps reconstruct on when there is one -- it is idle here, the caller's
            // stream has waited for everything on it -- rather than one more: measured with a fourth stream of the context, the two
            // chains no longer overlapped at all, 2.14 instead of 1.89 ms per SILK-NB step; the hardware queues are few)
            if (!ctx->recon_stream && !ctx->side_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
            hipStream_t const side = ctx->recon_stream ? ctx->recon_stream : ctx->side_stream;
            // (Both chains start together.  Staggered -- the second one behind the first one's parse kernels, so that one half parses
            // while the other synthesises from the start -- was measured SLOWER, 14.9 against 14.5 ms per step of 262,144 hybrid
            // frames and 2.44 against 1.90 ms per SILK-NB step: half a batch's parse takes as long as a whole batch's.)
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, s));
            HIPCHK(ctx, hipStreamWaitEvent(side, ctx->ev_fork, 0));
            const int h = (n / 2 + 63) / 64 * 64; // (a multiple of the parse kernels' frames per workgroup)
            front(s, 0, h);
            back_half(s, 0, h);
            front(side, (size_t)h, n - h);
            back_half(side, (size_t)h, n - h);
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, side));
            HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
        } else if (slices && slices->count > 1) {
            front(s, 0, n);
            for (int i = 0; i < slices->count; i++) {
                const size_t lo = slices->bounds[i], hi = slices->bounds[i + 1];
                if (hi > lo) back_half(s, lo, (int)(hi - lo));
                if (int rc = slices->after_slice(i)) return rc;
            }
        } else {
            front(s, 0, n);
            back_half(s, 0, n);
        }
        HIPCHK(ctx, hipGetLastError());
        if (ctx->pipeline) { // (a step that ran in order: whatever a later pipelined step runs ahead waits for all of it)
            HIPCHK(ctx, hipEventRecord(ctx->ev_front, s));
            ctx->front_recorded = 1;
        }
        return OPUSGPU_OK;
    }
    // The kernels of a step and what orders them:
    //   FRONT   k_silk_parse  k_celt_parse  k_silk_synth (or the full kernel)  k_decode_step[Q4]
    //   BACK    k_celt_recon_fb  k_celt_recon  ->  k_celt_post
    // In order (the default): all on the step's stream.  After k_silk_parse the SILK synthesis and the CELT parse +
    // reconstruction are independent (SilkRec::prev_mode, og_silk_parse.hpp); running them on two streams was measured
    // (DESIGN.md section 6): next to k_celt_recon the synthesis gains nothing; next to k_celt_parse it gains 5 % on mixed-mode
    // steps but costs 13 % on CELT-only steps.
    // Pipelined (opusgpu_set_pipeline, a step the caller declares CELT-only; the tables are resident, so nothing here waits for
    // the caller's earlier work):
    //   parse_stream   [the last in-order step, post of step k-3]  k_celt_parse
    //   recon_stream   [the parse, post of step k-2; in a window: every workgroup of the parse of step k+1]  k_celt_recon_fb  k_celt_recon
    //   step's stream  [reconstruction of step k; in a window: the first round of the reconstruction of step k+1]  k_celt_post
    // The entropy half reads one thing of the stream's state, the band energies, and writes them itself (celt_parse_lane): the
    // parse of step k+1 depends on the parse of step k only.  The reconstruction touches neither the caller's buffers (its result
    // codes go through ReconOut) nor anything k_celt_post reads of the step BEFORE (the history ring is written 960 samples
    // further on; the ring position travels in ReconOut), so the reconstruction of step k+1 starts while k_celt_post of step k
    // runs; two steps on, it waits for it (the ring holds two frames; records and ReconOut rotate through three sets).
    // A step that is not declared CELT-only runs in order (ev_front: a later pipelined step's parse waits for all of it).
    // PLACEMENT.  The three kernels compete for LDS: the parse is one round of 14 KB workgroups that live ~1 ms, the
    // reconstruction 65,536 workgroups of 7.5 KB that live ~0.15 ms, the de-emphasis 10 KB ones that nothing waits for.  A parse
    // workgroup that arrives when the CUs are full of reconstruction workgroups finds no hole that fits it (3.1 ms per step
    // instead of 2.3), so the order that works is: parse of step k+1, THEN reconstruction of step k, THEN de-emphasis of step
    // k-1.  Events cannot say "that kernel's workgroups have started"; round 2 approximated it with a wave that watched the wall
    // clock.  When the caller queues a window of steps (opusgpu_decode_steps_device) the next step is known, and the order is a
    // real dependency: the parse and reconstruction workgroups count themselves in when they start, and the stream that
    // launches the dependent kernel waits on that count (hipStreamWaitValue32) -- placement does not depend on how long a launch
    // or an event takes to arrive.  A single step (opusgpu_decode_step_device) cannot know whether another follows: its kernels
    // are released by their data dependencies alone.
    // (from here on: a pipelined step -- CELT-only frames, no SILK records, no hand-off)
    // the early parse: behind the front of the step before and its own slot's last user (three steps back)
    if (ctx->front_recorded) HIPCHK(ctx, hipStreamWaitEvent(ctx->parse_stream, ctx->ev_front, 0));
    if (ctx->post_recorded[par]) HIPCHK(ctx, hipStreamWaitEvent(ctx->parse_stream, ctx->ev_post[par], 0));
    const int wide = og_debug().parse_wide ? og_celt_parse64_frames() : OG_PL_FRAMES; // frames per group of the early parse
    {
        const int grid = (n + wide * ctx->parse_groups - 1) / (wide * ctx->parse_groups);
        launch_jitter();
        if (og_debug().parse_wide) // (64 frames per wave: next to the reconstruction the parse costs its issue slots, not its latency)
            og_launch_celt_parse64(ctx->parse_stream, grid, d_descs, d_arena, ctx->d_streams, recs, n, ctx->n_streams, nullptr, (int)PARSE_CELT_ONLY,
                                   ctx->parse_groups, ctx->d_started);
        else
            hipLaunchKernelGGL(k_celt_parse, dim3(grid), parse_block, 0, ctx->parse_stream, (const FrameDesc *)d_descs, (const u8 *)d_arena,
                               ctx->d_streams, recs, n, ctx->n_streams, (const SilkHandoff *)nullptr, (int)PARSE_CELT_ONLY, ctx->parse_groups,
                               ctx->d_started);
        ctx->parse_started_total += (u32)grid;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_parsed, ctx->parse_stream));
    hipStream_t const back = ctx->recon_stream;
    HIPCHK(ctx, hipStreamWaitEvent(back, ctx->ev_parsed, 0));
    if (ctx->post_recorded[par2]) HIPCHK(ctx, hipStreamWaitEvent(back, ctx->ev_post[par2], 0)); // (the ring: 2 x 960 of 2048)
    if (window) { // ... and every workgroup of the next step's parse has its place
        const int next_grid = (next_n + wide * ctx->parse_groups - 1) / (wide * ctx->parse_groups);
        ctx->window_parse_target = ctx->parse_started_total + (u32)next_grid;
        HIPCHK(ctx, hipStreamWaitValue32(back, ctx->d_started, ctx->window_parse_target, hipStreamWaitValueGte, 0xffffffffu));
    }
    // reconstruct (one frame per wave) ...
    if (ctx->fast_recon) {
        launch_jitter();
        og_launch_celt_recon_fb(back, d_descs, ctx->d_streams, recs, rout, n, ctx->n_streams, 0, ctx->d_started + 16);
        ctx->recon_started_total += (u32)og_celt_recon_fb_signals(n);
    }
    hipLaunchKernelGGL(k_celt_recon, dim3(ctx->fast_recon ? (n + 63) / 64 : n), dim3(64), 0, back, (const FrameDesc *)d_descs, ctx->d_streams,
                       (const ParseRec *)recs, rout, n, ctx->n_streams, 0, ctx->fast_recon);
    HIPCHK(ctx, hipEventRecord(ctx->ev_recon, back));
    HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_recon, 0));
    // Nothing waits for the de-emphasis for two steps, and placed before the next step's reconstruction its 10 KB workgroups
    // take room that kernel -- the critical one -- would use: in a window it is held until the first round of that
    // reconstruction has started (its count of started workgroups, one in 64 counted)
    if (window && ctx->fast_recon) {
        const int first_round = OG_MIN(og_celt_recon_fb_signals(next_n), 32);
        ctx->window_recon_target = ctx->recon_started_total + (u32)first_round;
        HIPCHK(ctx, hipStreamWaitValue32(s, ctx->d_started + 16, ctx->window_recon_target, hipStreamWaitValueGte, 0xffffffffu));
    }
    // ... -> de-emphasis and PCM (one (frame, channel) per lane); the result codes
    launch_jitter();
    hipLaunchKernelGGL(k_celt_post, dim3((n * ctx->channels + 63) / 64), dim3(64), 0, s, (const FrameDesc *)d_descs, ctx->d_streams,
                       (const ParseRec *)recs, (const ReconOut *)rout, (i32 *)d_result, (i16 *)d_pcm, n, ctx->n_streams, ctx->channels,
                       pcm_stride, (const SilkHandoff *)nullptr, modes, 0);
    HIPCHK(ctx, hipEventRecord(ctx->ev_post[par], s));
    ctx->post_recorded[par] = 1;
    HIPCHK(ctx, hipGetLastError());
    return OPUSGPU_OK;
}

int opusgpu_decode_step_device(opusgpu_ctx *ctx, int n, const void *d_descs, const void *d_arena, void *d_pcm,
                               void *d_result, void *hip_stream) {
    return decode_step_impl(ctx, n, d_descs, d_arena, d_pcm, d_result, hip_stream, true);
}
int opusgpu_decode_step_device_modes(opusgpu_ctx *ctx, int n, const void *d_descs, const void *d_arena, void *d_pcm,
                                     void *d_result, void *hip_stream, int modes) {
    if (modes <= 0 || modes > 15 || (modes & 7) == 0) return OPUSGPU_BAD_ARG;
    return decode_step_impl(ctx, n, d_descs, d_arena, d_pcm, d_result, hip_stream, true, modes);
}
int opusgpu_decode_steps_device(opusgpu_ctx *ctx, int n_steps, const int32_t *n, const void *const *d_descs, const void *const *d_arena,
                                void *const *d_pcm, void *const *d_result, void *hip_stream, int modes) {
    if (!ctx || n_steps < 0 || modes < 0 || modes > 7) return OPUSGPU_BAD_ARG;
    if (n_steps == 0) return OPUSGPU_OK;
    if (!n || !d_descs || !d_arena || !d_pcm || !d_result) return OPUSGPU_BAD_ARG;
    if (ctx->d_started && (ctx->parse_started_total > 0x70000000u || ctx->recon_started_total > 0x70000000u)) {
        // the start counters only grow: long before they could wrap they restart from zero, on an idle device
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if (int rc = sync_in_flight(ctx)) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemset(ctx->d_started, 0, 128));
        ctx->parse_started_total = ctx->recon_started_total = 0;
    }
    // Everything that can refuse a step is looked at BEFORE the first launch: step k of a window holds its reconstruction and its
    // de-emphasis until workgroups of step k + 1 have started (hipStreamWaitValue32 below), so a call that stopped between the two
    // would leave waits nothing satisfies.
    if (!ctx->d_streams) return OPUSGPU_BAD_ARG;
    int max_n = 0;
    for (int k = 0; k < n_steps; k++) {
        if (n[k] < 0) return OPUSGPU_BAD_ARG;
        if (n[k] > 0 && (!d_descs[k] || !d_arena[k] || ((uintptr_t)d_arena[k] & 15) || !d_pcm[k] || !d_result[k])) return OPUSGPU_BAD_ARG;
        max_n = OG_MAX(max_n, n[k]);
    }
    if (max_n == 0) return OPUSGPU_OK;
    const int m = modes ? modes : 7;
    if (ctx->pipeline && m == 4 && ctx->split_celt && ctx->mode != OPUSGPU_MODE_RFC) {
        // ... and so is every allocation: the record slots only grow, growing frees the old buffer, and hipFree waits for ALL
        // streams of the device -- among them the one whose head is such a wait for a parse that this thread has yet to launch.
        // All three slots take the window's largest step now, while nothing of the window is queued.
        HIPCHK(ctx, hipSetDevice(ctx->device));
        for (int par = 0; par < 3; par++)
            if (int rc = grow_step_slot(ctx, par, (size_t)max_n)) return rc;
    }
    for (int k = 0; k < n_steps; k++) {
        const int next_n = k + 1 < n_steps ? n[k + 1] : 0;
        const int rc = decode_step_impl(ctx, n[k], d_descs[k], d_arena[k], d_pcm[k], d_result[k], hip_stream, true, m, next_n > 0 ? next_n : 0);
        if (rc) { // (a HIP error in the middle of a window: let go of what the steps before it wait for, then report it)
            release_window_waits(ctx);
            return rc;
        }
    }
    ctx->window_parse_target = ctx->window_recon_target = 0; // (every wait of this window has its kernel queued behind it)
    return OPUSGPU_OK;
}

#ifdef OG_PROF
// profiling builds only: per-section wave-cycle totals of k_celt_recon (see OG_MARK), optionally cleared after the read
extern "C" int og_recon_fb_prof(unsigned long long *out64, int reset);
extern "C" int og_rfc_prof(unsigned long long *out64, int reset);
extern "C" int og_ssynth_prof(unsigned long long *out64, int reset);
extern "C" int og_ssynth_nb_prof(unsigned long long *out64, int reset);
int opusgpu_debug_prof(unsigned long long *out64, int reset) {
    unsigned long long fb[64], rf[64];
    if (og_recon_fb_prof(fb, reset) != 0 || og_rfc_prof(rf, reset) != 0) return -1;
    for (int i = 0; i < 64; i++) fb[i] += rf[i];
    if (og_ssynth_prof(rf, reset) != 0) return -1;
    for (int i = 0; i < 64; i++) fb[i] += rf[i];
    if (og_ssynth_nb_prof(rf, reset) != 0) return -1;
    for (int i = 0; i < 64; i++) fb[i] += rf[i];
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    for (int i = 0; i < 64; i++) out64[i] += fb[i];
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int opusgpu_pages_crc_device(opusgpu_ctx *ctx, int n_pages, const void *d_blob, const void *d_offsets, const void *d_lens,
                             void *d_status, void *hip_stream) {
    if (!ctx || n_pages < 0) return OPUSGPU_BAD_ARG;
    if (n_pages == 0) return OPUSGPU_OK;
    if (!d_blob || !d_offsets || !d_lens || !d_status) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    if (!ctx->d_crc_tables) { // t[0] = the byte table; t[j][i] = t[j-1][i] advanced by one zero byte
        static u32 t[8][256];
        for (u32 i = 0; i < 256; i++) {
            u32 r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            t[0][i] = r;
        }
        for (int j = 1; j < 8; j++)
            for (u32 i = 0; i < 256; i++) t[j][i] = (t[j - 1][i] << 8) ^ t[0][t[j - 1][i] >> 24];
        if (hipMalloc((void **)&ctx->d_crc_tables, sizeof(t)) != hipSuccess) return OPUSGPU_ALLOC_FAIL;
        HIPCHK(ctx, hipMemcpy(ctx->d_crc_tables, t, sizeof(t), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_pages_crc, dim3((n_pages + 255) / 256), dim3(256), 0, s, (const u8 *)d_blob, (const long long *)d_offsets,
                       (const i32 *)d_lens, (i32 *)d_status, n_pages, (const u32 *)ctx->d_crc_tables);
    HIPCHK(ctx, hipGetLastError());
    return OPUSGPU_OK;
}

int opusgpu_output_stage_device(opusgpu_ctx *ctx, int n_blocks, int block_samples, const void *d_pcm, long long pcm_stride,
                                const void *d_valid, int valid_all, const void *d_cfgs, opusgpu_output_cfg cfg, void *d_i2s,
                                long long i2s_stride, void *hip_stream) {
    if (!ctx || n_blocks < 0 || block_samples < 0) return OPUSGPU_BAD_ARG;
    if (n_blocks == 0 || block_samples == 0) return OPUSGPU_OK;
    if (!d_pcm || !d_i2s || pcm_stride < 0 || i2s_stride < 0) return OPUSGPU_BAD_ARG;
    if (!d_valid && (valid_all < 0 || valid_all > block_samples)) return OPUSGPU_BAD_ARG;
    // the reference's setters refuse anything else (setBitsPerSample / setChannels, src/main.cpp:119-132)
    if (!d_cfgs && ((cfg.bits != 8 && cfg.bits != 16) || (cfg.channels != 1 && cfg.channels != 2))) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    // words a block can make: 8-bit mono plays two per sample; with per-block settings any block might
    const int max_words = (d_cfgs || (cfg.bits == 8 && cfg.channels == 1)) ? 2 * block_samples : block_samples;
    const int units = (max_words + 3) / 4;
    const long long n_units = (long long)units * n_blocks;
    const int vec_ok = ((uintptr_t)d_pcm % 16 == 0 && pcm_stride % 8 == 0 && (uintptr_t)d_i2s % 16 == 0 && i2s_stride % 4 == 0) ? 1 : 0;
    OutputCfg c;
    memcpy(&c, &cfg, sizeof c);
    hipLaunchKernelGGL(k_output_stage, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, s, (const i16 *)d_pcm, pcm_stride,
                       (const i32 *)d_valid, valid_all, block_samples, (const OutputCfg *)d_cfgs, c, (u32 *)d_i2s, i2s_stride, units,
                       n_units, vec_ok);
    HIPCHK(ctx, hipGetLastError());
    return OPUSGPU_OK;
}

int opusgpu_synchronize(opusgpu_ctx *ctx) {
    if (!ctx) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return OPUSGPU_OK;
}

int opusgpu_event_create(opusgpu_ctx *ctx, void **event) {
    if (!ctx || !event) return OPUSGPU_BAD_ARG;
    hipEvent_t e;
    HIPCHK(ctx, hipEventCreate(&e));
    *event = (void *)e;
    return OPUSGPU_OK;
}
int opusgpu_event_record(opusgpu_ctx *ctx, void *event) {
    if (!ctx || !event) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return OPUSGPU_OK;
}
int opusgpu_event_elapsed_ms(opusgpu_ctx *ctx, void *start, void *stop, float *ms) {
    if (!ctx || !start || !stop || !ms) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(ctx, hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return OPUSGPU_OK;
}
int opusgpu_event_destroy(opusgpu_ctx *ctx, void *event) {
    if (!ctx || !event) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipEventDestroy((hipEvent_t)event));
    return OPUSGPU_OK;
}

// ---- uploads NEXT TO the decode (the ingest of the next batch of pages under the decode of this one).
// The upload runs on the context's copy stream, never on the decode stream: a host thread demuxes batch b + 1 and queues its
// tables and packet bytes here while the caller's thread has batch b's steps in flight; the fence event orders the two.
static std::mutex g_copy_stream_mutex;
static int copy_stream_of(opusgpu_ctx *ctx, hipStream_t *out) {
    std::lock_guard<std::mutex> lock(g_copy_stream_mutex);
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    *out = ctx->copy_stream;
    return OPUSGPU_OK;
}
int opusgpu_upload_async(opusgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t cs;
    if (int rc = copy_stream_of(ctx, &cs)) return rc;
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
    return OPUSGPU_OK;
}
int opusgpu_upload_fence(opusgpu_ctx *ctx, void *event) {
    if (!ctx || !event) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t cs;
    if (int rc = copy_stream_of(ctx, &cs)) return rc;
    HIPCHK(ctx, hipEventRecord((hipEvent_t)event, cs));
    return OPUSGPU_OK;
}
int opusgpu_stream_wait_event(opusgpu_ctx *ctx, void *event, void *hip_stream) {
    if (!ctx || !event) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamWaitEvent(hip_stream ? (hipStream_t)hip_stream : ctx->stream, (hipEvent_t)event, 0));
    // the context's own stream: the streams pipelined steps run ahead on wait too -- what is behind the event (an upload of step
    // tables, opusgpu_upload_fence) is then as good as resident for every kernel of the steps queued after this call
    if ((!hip_stream || (hipStream_t)hip_stream == ctx->stream) && ctx->parse_stream) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->parse_stream, (hipEvent_t)event, 0));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->recon_stream, (hipEvent_t)event, 0));
    }
    return OPUSGPU_OK;
}
int opusgpu_event_synchronize(opusgpu_ctx *ctx, void *event) {
    if (!ctx || !event) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize((hipEvent_t)event));
    return OPUSGPU_OK;
}
// Caller-owned host buffers made DMA-able in place (page-locked): uploads from and PCM copies into registered memory run at
// the full PCIe rate without the staging copy (opusgpu_decode_packets notices registered PCM buffers by itself).
int opusgpu_host_register(opusgpu_ctx *ctx, void *ptr, size_t bytes) {
    if (!ctx || !ptr || !bytes) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(ctx, OPUSGPU_ALLOC_FAIL, "hipHostRegister", e);
    std::lock_guard<std::mutex> lock(ctx->registered_mutex);
    ctx->registered.emplace_back((uintptr_t)ptr, bytes);
    return OPUSGPU_OK;
}
int opusgpu_host_unregister(opusgpu_ctx *ctx, void *ptr) {
    if (!ctx || !ptr) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> lock(ctx->registered_mutex);
        for (size_t i = 0; i < ctx->registered.size(); i++)
            if (ctx->registered[i].first == (uintptr_t)ptr) {
                ctx->registered.erase(ctx->registered.begin() + i);
                break;
            }
    }
    HIPCHK(ctx, hipHostUnregister(ptr));
    return OPUSGPU_OK;
}

int opusgpu_stream_state_get(opusgpu_ctx *ctx, int index, void *dst, size_t bytes) {
    if (!ctx || !dst || index < 0 || index >= ctx->n_streams || bytes > sizeof(StreamState)) return OPUSGPU_BAD_ARG;
    return opusgpu_memcpy_d2h(ctx, dst, &ctx->d_streams[index], bytes);
}

int opusgpu_stream_pitch_get(opusgpu_ctx *ctx, int index, int32_t out[4]) {
    if (!ctx || !out || index < 0 || index >= ctx->n_streams) return OPUSGPU_BAD_ARG;
    int32_t head[2], ch0[3]; // (channels, prev_mode); (lagPrev .. fs_kHz are not adjacent: three small copies, one synchronisation)
    if (int rc = opusgpu_memcpy_d2h(ctx, head, &ctx->d_streams[index], sizeof(head))) return rc;
    const SilkChannel *c = &ctx->d_streams[index].silk.ch[0];
    HIPCHK(ctx, hipMemcpy(&ch0[0], &c->prevSignalType, 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(&ch0[1], &c->lagPrev, 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(&ch0[2], &c->fs_kHz, 4, hipMemcpyDeviceToHost));
    out[0] = head[1];
    out[1] = ch0[0];
    out[2] = ch0[1];
    out[3] = ch0[2];
    return OPUSGPU_OK;
}

int opusgpu_debug_stage_taps(opusgpu_ctx *ctx, int slot, opusgpu_stage_taps *out) {
    if (!ctx || !out || slot < 0 || slot >= ctx->last_n || !ctx->last_descs) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_in_flight(ctx)) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof(*out));
    FrameDesc d;
    HIPCHK(ctx, hipMemcpy(&d, (const FrameDesc *)ctx->last_descs + slot, sizeof(d), hipMemcpyDeviceToHost));
    if (d.stream < 0 || d.stream >= ctx->n_streams) return OPUSGPU_BAD_ARG;
    const int mode = MODE_SILK + (d.flags & 3);
    std::unique_ptr<StreamState> st(new (std::nothrow) StreamState);
    if (!st) return OPUSGPU_ALLOC_FAIL;
    HIPCHK(ctx, hipMemcpy(st.get(), &ctx->d_streams[d.stream], sizeof(StreamState), hipMemcpyDeviceToHost));
    if (mode != MODE_SILK && ctx->last_recs) {
        std::unique_ptr<ParseRec> r(new (std::nothrow) ParseRec);
        if (!r) return OPUSGPU_ALLOC_FAIL;
        HIPCHK(ctx, hipMemcpy(r.get(), (const ParseRec *)ctx->last_recs + slot, sizeof(ParseRec), hipMemcpyDeviceToHost));
        out->celt_valid = 1;
        out->celt_ret = r->ret;
        out->silence = (r->flags & RF_SILENCE) != 0;
        out->transient = (r->flags & RF_TRANSIENT) != 0;
        out->lm = (int)(r->flags >> RF_LM_SHIFT) & 3;
        out->spread = (int)(r->flags >> RF_SPREAD_SHIFT) & 3;
        out->dual_stereo = (r->flags & RF_DUAL) != 0;
        out->anti_collapse_on = (r->flags & RF_ANTI_COLLAPSE) != 0;
        out->intensity = r->intensity;
        out->pf_pitch = r->pf_pitch;
        out->pf_gain = r->pf_gain;
        out->pf_tapset = r->pf_tapset;
        out->n_leaves = r->n_leaves;
        out->celt_rng_final = r->rng_final;
        memcpy(out->bandE, r->bandE, sizeof(out->bandE));
        memcpy(out->pulses, r->pulses, sizeof(out->pulses));
        memcpy(out->tf_res, r->tf_res, sizeof(out->tf_res));
    }
    const CeltState &c = st->celt;
    for (int ch = 0; ch < 2; ch++) {
        for (int i = 0; i < 960; i++) out->syn_post[ch][i] = c.ring[ch][(c.ring_pos - 960 + i) & RING_MASK];
        for (int i = 0; i < 60; i++) out->overlap_tail[ch][i] = c.tail[ch][i];
    }
    memcpy(out->state_bandE, c.bandE, sizeof(out->state_bandE));
    memcpy(out->state_logE1, c.logE1, sizeof(out->state_logE1));
    memcpy(out->state_logE2, c.logE2, sizeof(out->state_logE2));
    out->state_rng = c.rng;
    out->pf_period = c.pf_period;
    out->pf_gain_state = c.pf_gain;
    out->pf_tapset_state = c.pf_tapset;
    if (mode != MODE_CELT && ctx->last_had_silk_recs && ctx->last_srecs) {
        std::unique_ptr<SilkRec> r(new (std::nothrow) SilkRec);
        if (!r) return OPUSGPU_ALLOC_FAIL;
        HIPCHK(ctx, hipMemcpy(r.get(), (const SilkRec *)ctx->last_srecs + slot, sizeof(SilkRec), hipMemcpyDeviceToHost));
        out->silk_valid = 1;
        out->silk_ret = r->ret;
        out->decode_only_middle = r->decode_only_middle;
        out->ms_pred_q13[0] = r->MS_pred_Q13[0];
        out->ms_pred_q13[1] = r->MS_pred_Q13[1];
        for (int ch = 0; ch < 2; ch++) {
            const SilkRecCh &k = r->ch[ch];
            memcpy(out->silk_ch[ch].pitchL, k.pitchL, sizeof(k.pitchL));
            memcpy(out->silk_ch[ch].Gains_Q16, k.Gains_Q16, sizeof(k.Gains_Q16));
            memcpy(out->silk_ch[ch].PredCoef_Q12, k.PredCoef_Q12, sizeof(k.PredCoef_Q12));
            memcpy(out->silk_ch[ch].LTPCoef_Q14, k.LTPCoef_Q14, sizeof(k.LTPCoef_Q14));
            out->silk_ch[ch].LTP_scale_Q14 = k.LTP_scale_Q14;
            out->silk_ch[ch].signalType = k.signalType;
            out->silk_ch[ch].quantOffsetType = k.quantOffsetType;
        }
    }
    for (int ch = 0; ch < 2; ch++) {
        memcpy(out->silk_out[ch], st->silk.ch[ch].outBuf, sizeof(out->silk_out[ch]));
        memcpy(out->silk_sLPC_Q14[ch], st->silk.ch[ch].sLPC_Q14_buf, sizeof(out->silk_sLPC_Q14[ch]));
        out->silk_fs_kHz[ch] = st->silk.ch[ch].fs_kHz;
    }
    return OPUSGPU_OK;
}

int opusgpu_packet_to_frames(const uint8_t *packet, int32_t len, int32_t stream, opusgpu_frame_desc descs[48]) {
    return opusgpu_packet_to_frames_mode(packet, len, stream, OPUSGPU_MODE_REFERENCE, descs);
}

int opusgpu_packet_to_frames_mode(const uint8_t *packet, int32_t len, int32_t stream, int mode, opusgpu_frame_desc descs[48]) {
    if (!packet || !descs) return OPUSGPU_BAD_ARG;
    if (len <= 0) return len == 0 ? OPUSGPU_INVALID_PACKET : OPUSGPU_BAD_ARG;
    int16_t size[48];
    uint8_t toc;
    int offset = 0;
    const int count = ogh::parse_packet(packet, len, 0, &toc, size, &offset, nullptr);
    if (count < 0) return count;
    const int32_t flags = mode == OPUSGPU_MODE_RFC ? ogh::toc_flags_rfc(toc) : ogh::toc_flags(toc);
    for (int i = 0; i < count; i++) {
        descs[i].stream = stream;
        descs[i].offset = offset;
        descs[i].len = size[i];
        descs[i].flags = flags;
        offset += size[i];
    }
    return count;
}

int opusgpu_empty_packet_to_frames(int32_t stream, int32_t last_flags, int decoder_channels, int frame_size, opusgpu_frame_desc descs[48]) {
    if (!descs || frame_size <= 0 || frame_size % 120 || (decoder_channels != 1 && decoder_channels != 2)) return OPUSGPU_BAD_ARG;
    const int count = (frame_size + OPUSGPU_FRAME_SAMPLES - 1) / OPUSGPU_FRAME_SAMPLES;
    if (count > 48) return OPUSGPU_BAD_ARG;
    const int32_t flags = last_flags >= 0 ? (last_flags & 63) : ogh::empty_flags_no_packet_yet(decoder_channels);
    for (int i = 0; i < count; i++) descs[i] = opusgpu_frame_desc{stream, 0, 0, flags};
    return count;
}

static int grow_pinned(opusgpu_ctx *ctx, void **p, size_t *cap, size_t need) {
    if (*cap >= need) return OPUSGPU_OK;
    if (*p) HIPCHK(ctx, hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4;
    if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) return OPUSGPU_ALLOC_FAIL;
    *cap = want;
    return OPUSGPU_OK;
}

static int grow(opusgpu_ctx *ctx, void **p, size_t *cap, size_t need) {
    if (*cap >= need) return OPUSGPU_OK;
    if (*p) HIPCHK(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 2 + 256;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) return fail(ctx, OPUSGPU_ALLOC_FAIL, "hipMalloc(staging)", e);
    *cap = want;
    return OPUSGPU_OK;
}

// How a concealment of `total` samples is cut into device frames (valid duration codes): a frame of the last packet's size at a
// time like opus_decode(NULL) (src/opus_decoder.cpp:294-308 has the loop), what is left over (30 / 50 ms) as 20 / 40 ms + 10 ms.
static int conceal_pieces(int total, int last_fs, int32_t base_flags, int32_t out_flags[48]) {
    static const int kDur[6] = {2880, 1920, 960, 480, 240, 120}, kCode[6] = {5, 4, 0, 3, 2, 1};
    int n = 0;
    while (total > 0) {
        int w = total < last_fs ? total : last_fs;
        total -= w;
        while (w > 0) {
            int j = 0;
            while (kDur[j] > w) j++;
            if (n == 48) return -1;
            out_flags[n++] = (base_flags & ~(7 << 6) & ~(1 << 10)) | kCode[j] << 6 | 1 << 9;
            w -= kDur[j];
        }
    }
    return n;
}

// CPUs this process may run on (its affinity mask: a container's share, not the machine's), at most `most`.
static int host_cpus(int most) {
    cpu_set_t set;
    int c = 8;
    if (sched_getaffinity(0, sizeof set, &set) == 0) c = CPU_COUNT(&set);
    return c < 1 ? 1 : (c > most ? most : c);
}

// Is [p, p + bytes) page-locked host memory the device can write?  ONE page-locked range must cover all of it: either one the
// caller registered through opusgpu_host_register (the context keeps the list), or one allocation / registration the runtime knows
// (its start and size are asked for: two ends that are each page-locked may have pageable memory between them).
static bool host_range_is_pinned(opusgpu_ctx *ctx, const void *p, size_t bytes) {
    if (!p || !bytes) return false;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    {
        std::lock_guard<std::mutex> lock(ctx->registered_mutex);
        for (const auto &r : ctx->registered)
            if (lo >= r.first && hi <= r.first + r.second) return true;
    }
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess || a.type != hipMemoryTypeHost) {
        (void)hipGetLastError(); // (pageable memory is reported as an error: not one of ours)
        return false;
    }
    void *start = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)p) != hipSuccess ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return lo >= (uintptr_t)start && hi <= (uintptr_t)start + size;
}

static int decode_packets_impl(opusgpu_ctx *ctx, int n, const int32_t *stream_ids, const uint8_t *const *packets,
                               const int32_t *lens, int16_t *pcm, int frame_capacity, int32_t *result, const bool fec) {
    if (!ctx || n < 0 || !ctx->d_streams) return OPUSGPU_BAD_ARG;
    if (n == 0) return OPUSGPU_OK;
    if (!stream_ids || !packets || !lens || !pcm || !result || frame_capacity <= 0) return OPUSGPU_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->pipeline) // (pipelined device-resident steps may still be in flight on the library's streams, which the parts below use)
        if (int rc0 = sync_in_flight(ctx)) return rc0;
    const int CC = ctx->channels;
    const bool rfc = ctx->mode == OPUSGPU_MODE_RFC;
    if (fec && !rfc) return OPUSGPU_BAD_ARG;
    // one frame's block in the device PCM buffer: 20 ms, or room for a 60 ms frame in RFC mode
    const size_t frame_pcm = (size_t)(rfc ? OPUSGPU_RFC_FRAME_SAMPLES : OPUSGPU_FRAME_SAMPLES) * CC;
    const size_t cap_pcm = (size_t)frame_capacity * OPUSGPU_FRAME_SAMPLES * CC; // the caller's block per packet
    // 1. frame the packets on the host (opus_decode_native, src/opus_decoder.cpp:280-348).  Large batches by ranges of
    //    packets on a few threads, in two passes: frame counts and sizes, then (after the prefix sums that place every
    //    packet) descriptors and packet bytes.
    HostPhaseTimer timer;
    const int host_threads = n >= 4096 ? host_cpus(16) : 1;
    auto on_subranges = [&](int from, int to, auto &&f) { // f(lo, hi) over [from, to), on the host threads
        if (host_threads == 1 || to - from < 1024) {
            f(from, to);
            return;
        }
        const int64_t w = to - from;
        ctx->pool.run(host_threads, [&](int t) { f(from + (int)(w * t / host_threads), from + (int)(w * (t + 1) / host_threads)); });
    };
    auto on_ranges = [&](auto &&f) { on_subranges(0, n, f); };
    std::vector<int> first(n + 1, 0), nframes(n, 0);
    std::vector<uint8_t> is_lost(n, 0);
    // The common large call is REGULAR: every packet holds one frame (frame-count code 0) of a stream that exists, of a size and
    // duration the call has room for.  One look at the TOC bytes settles that, and then nothing of the first framing pass is
    // needed: packet i is frame i of the one step, its bytes lie at the running sum of the lengths, and the (only) framing pass
    // runs part by part next to the device (0.63 + 0.17 ms of host work less in front of the first kernel at 65,536 packets).
    bool regular = !rfc && !fec && n >= 4096 && ctx->host_parts > 1 && !timer.on;
    if (regular) {
        std::atomic<int> irregular{0};
        on_ranges([&](int lo, int hi) {
            for (int i = lo; i < hi; i++) {
                const uint8_t *p = packets[i];
                if (stream_ids[i] < 0 || stream_ids[i] >= ctx->n_streams || !p || lens[i] < 1 || lens[i] > 1276 || (p[0] & 3) != 0 ||
                    ogh::toc_samples_per_frame(p[0], 48000) > frame_capacity * OPUSGPU_FRAME_SAMPLES) {
                    irregular.store(1, std::memory_order_relaxed);
                    return;
                }
                result[i] = 0;
                nframes[i] = 1;
                ctx->last_count[stream_ids[i]] = 1; // (what an empty packet of this stream will be decoded as: below)
                ctx->last_flags[stream_ids[i]] = ogh::toc_flags(p[0]);
            }
        });
        regular = irregular.load() == 0;
        if (!regular) std::fill(nframes.begin(), nframes.end(), 0);
    }
    if (!regular) on_ranges([&](int lo, int hi) {
        for (int i = lo; i < hi; i++) {
            result[i] = 0;
            if (stream_ids[i] < 0 || stream_ids[i] >= ctx->n_streams || lens[i] < 0) {
                result[i] = OPUSGPU_BAD_ARG;
                continue;
            }
            if (!packets[i] || lens[i] == 0) {
                if (!rfc) {
                    // The reference has no concealment, but opus_decode_native's empty-packet branch is live (src/opus_decoder.cpp:
                    // 290-308): opus_decode_frame(st, NULL, 0) -- a frame of no bytes in the stream's LAST mode / bandwidth / channel
                    // count, 960 samples per pass -- until frame_size (here frame_capacity x 960, a multiple of 120) is filled or a
                    // pass fails: SILK-only decodes (the coder reads zeros), hybrid runs its SILK half and ends in CELT's -18
                    // (src/celt.cpp:2225), CELT-only in -18; a stream without a packet since its reset is in mode 0 (descriptor bit 11)
                    nframes[i] = frame_capacity;
                    is_lost[i] = 1;
                    continue;
                }
                // RFC mode: a lost packet is concealed as long as the stream's last packet was (one 20 ms frame if there was none)
                const int count = ctx->last_count[stream_ids[i]] ? ctx->last_count[stream_ids[i]] : 1;
                const int fs = ogh::flags_frame_size(ctx->last_flags[stream_ids[i]]);
                if ((int64_t)count * fs > (int64_t)frame_capacity * OPUSGPU_FRAME_SAMPLES) {
                    result[i] = OPUSGPU_BUFFER_TOO_SMALL;
                    continue;
                }
                nframes[i] = count;
                is_lost[i] = 1;
                continue;
            }
            opusgpu_frame_desc d[48];
            const int count = opusgpu_packet_to_frames_mode(packets[i], lens[i], stream_ids[i], ctx->mode, d);
            if (count < 0) {
                result[i] = count;
                continue;
            }
            if (fec) { // the packet BEFORE this one was lost (opus_decode with decode_fec = 1): its duration is concealed, the last
                       // frame's worth of it from this packet's first frame where SILK data is there to carry LBRR frames
                const int sid = stream_ids[i], lc = ctx->last_count[sid];
                const int last_fs = lc ? ogh::flags_frame_size(ctx->last_flags[sid]) : 120;
                const int lost_dur = lc ? lc * last_fs : OPUSGPU_FRAME_SAMPLES;
                const int pfs = ogh::toc_samples_per_frame(packets[i][0], 48000);
                const bool celt = (d[0].flags & 3) == 2 || (lc && (ctx->last_flags[sid] & 3) == 2);
                if (lost_dur > frame_capacity * OPUSGPU_FRAME_SAMPLES) {
                    result[i] = OPUSGPU_BUFFER_TOO_SMALL;
                    continue;
                }
                int32_t fl[48];
                const bool use = !(lost_dur < pfs || celt);
                const int np = conceal_pieces(use ? lost_dur - pfs : lost_dur, last_fs, lc ? ctx->last_flags[sid] : d[0].flags, fl);
                if (np < 0 || np + (use ? 1 : 0) > 48) {
                    result[i] = OPUSGPU_BAD_ARG;
                    continue;
                }
                nframes[i] = np + (use ? 1 : 0);
                is_lost[i] = use ? 2 : 3; // 2: concealment + the FEC frame, 3: concealment only
                continue;
            }
            // count * packet_frame_size > frame_size -> OPUS_BUFFER_TOO_SMALL (src/opus_decoder.cpp:323)
            const int pfs = ogh::toc_samples_per_frame(packets[i][0], 48000);
            // (RFC mode decodes the durations the check is about: no second condition)
            if ((int64_t)count * pfs > (int64_t)frame_capacity * OPUSGPU_FRAME_SAMPLES || (!rfc && count > frame_capacity)) {
                result[i] = OPUSGPU_BUFFER_TOO_SMALL;
                continue;
            }
            nframes[i] = count;
            // what a later empty packet of the stream decodes / conceals as: st->mode, bandwidth, stream_channels (src/opus_decoder.cpp:
            // 327-331: set once the packet has passed the checks above, whatever its frames return).  (A stream appears at most once
            // per call: no two threads write the same entry.)
            ctx->last_count[stream_ids[i]] = count;
            ctx->last_flags[stream_ids[i]] = d[0].flags;
        }
    });
    timer.mark("framing pass 1 (counts)");
    std::vector<size_t> base(n + 1, 0); // where packet i lies in the arena
    int max_frames = 0;
    for (int i = 0; i < n; i++) {
        first[i + 1] = first[i] + nframes[i];
        base[i + 1] = base[i] + (nframes[i] && !(is_lost[i] == 1 || is_lost[i] == 3) ? (size_t)lens[i] : 0);
        if (nframes[i] > max_frames) max_frames = nframes[i];
    }
    timer.mark("prefix sums");
    if (first[n] == 0) return OPUSGPU_OK;
    if (base[n] > 0x7fffffffu) return OPUSGPU_BAD_ARG; // descriptor offsets are 32-bit: split the call
    if (int rc0 = grow_pinned(ctx, &ctx->h_descs, &ctx->cap_h_descs, sizeof(opusgpu_frame_desc) * (size_t)first[n])) return rc0;
    if (int rc0 = grow_pinned(ctx, &ctx->h_arena, &ctx->cap_h_arena, base[n] + 1)) return rc0;
    opusgpu_frame_desc *const all = (opusgpu_frame_desc *)ctx->h_descs; // frames in (packet, frame) order
    uint8_t *const arena = (uint8_t *)ctx->h_arena;
    timer.mark("staging");
    auto place = [&](int lo, int hi) { // framing pass 2: descriptors and packet bytes of packets [lo, hi) to their places
        for (int i = lo; i < hi; i++) {
            if (!nframes[i]) continue;
            if (is_lost[i] == 1) { // nothing to read: len 0, the flags of the stream's last packet (RFC mode: RFC bit and duration included)
                const int32_t fresh = rfc ? (int32_t)((ogh::MODE_CELT - ogh::MODE_SILK) | 4 << 2 | (CC == 2 ? 32 : 0) | 1 << 9)
                                          : ogh::empty_flags_no_packet_yet(CC);
                const int32_t fl = ctx->last_count[stream_ids[i]] ? ctx->last_flags[stream_ids[i]] : fresh;
                for (int k = 0; k < nframes[i]; k++) all[first[i] + k] = opusgpu_frame_desc{stream_ids[i], 0, 0, fl};
                continue;
            }
            if (rfc && is_lost[i] >= 2) { // decode_fec: the concealment frames, then (2) the packet's first frame with the FEC bit
                const int sid = stream_ids[i], lc = ctx->last_count[sid];
                opusgpu_frame_desc d0[48];
                (void)opusgpu_packet_to_frames_mode(packets[i], lens[i], sid, ctx->mode, d0);
                int32_t fl[48];
                const int np = nframes[i] - (is_lost[i] == 2 ? 1 : 0);
                const int last_fs = lc ? ogh::flags_frame_size(ctx->last_flags[sid]) : 120;
                const int lost_dur = lc ? lc * last_fs : OPUSGPU_FRAME_SAMPLES;
                (void)conceal_pieces(is_lost[i] == 2 ? lost_dur - ogh::flags_frame_size(d0[0].flags) : lost_dur, last_fs,
                                     lc ? ctx->last_flags[sid] : d0[0].flags, fl);
                for (int k = 0; k < np; k++) all[first[i] + k] = opusgpu_frame_desc{sid, 0, 0, fl[k]};
                if (is_lost[i] == 2) {
                    memcpy(arena + base[i], packets[i], (size_t)lens[i]);
                    d0[0].offset += (int32_t)base[i];
                    d0[0].flags |= 1 << 10;
                    all[first[i] + np] = d0[0];
                }
                continue;
            }
            opusgpu_frame_desc d[48];
            (void)opusgpu_packet_to_frames_mode(packets[i], lens[i], stream_ids[i], ctx->mode, d);
            memcpy(arena + base[i], packets[i], (size_t)lens[i]);
            for (int k = 0; k < nframes[i]; k++) {
                d[k].offset += (int32_t)base[i];
                all[first[i] + k] = d[k];
            }
        }
    };
    // The common large call -- one frame per packet -- is pipelined: the frames in packet order ARE the step table, so the call
    // goes in parts of packets, each placed (pass 2), uploaded and launched while the device works on the part before it and that
    // part's PCM travels back.  Anything else: everything placed and uploaded first, then one step per frame index.
    const bool pipelined = max_frames == 1 && first[n] >= 4096 && ctx->host_parts > 1 && !timer.on;
    int rc = grow(ctx, &ctx->d_arena, &ctx->cap_arena, base[n] + 16);
    if (rc) return rc;
    if (!pipelined) {
        on_ranges(place);
        timer.mark("prefix + framing pass 2 (place)");
        // 2. upload the arena once; run one step per frame index (frames of one packet are sequential)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_arena, arena, base[n], hipMemcpyHostToDevice, ctx->stream));
        timer.mark("arena upload (enqueue)");
    }
    std::vector<opusgpu_frame_desc> step;
    std::vector<int> owner;
    std::vector<int32_t> placed(rfc ? n : 0, 0); // RFC mode: samples of packet i delivered so far (frames may differ in duration)
    for (int k = 0; k < max_frames; k++) {
        step.clear();
        owner.clear();
        if (regular) {
            // (frame j belongs to packet j)
        } else if (pipelined) {
            owner.reserve(first[n]);
            for (int i = 0; i < n; i++)
                if (nframes[i]) owner.push_back(i);
        } else
            for (int i = 0; i < n; i++)
                if (nframes[i] > k && result[i] >= 0) {
                    step.push_back(all[first[i] + k]);
                    owner.push_back(i);
                }
        const int m = regular ? n : (int)owner.size();
        if (m == 0) break;
        timer.mark("step table");
        if ((rc = grow(ctx, &ctx->d_descs, &ctx->cap_descs, sizeof(opusgpu_frame_desc) * m))) return rc;
        if ((rc = grow(ctx, &ctx->d_pcm, &ctx->cap_pcm, frame_pcm * 2 * m))) return rc;
        if ((rc = grow(ctx, &ctx->d_result, &ctx->cap_result, sizeof(int32_t) * m))) return rc;
        if (!pipelined)
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_descs, step.data(), sizeof(opusgpu_frame_desc) * m, hipMemcpyHostToDevice,
                                       ctx->stream));
        if ((rc = grow_pinned(ctx, &ctx->h_pcm, &ctx->cap_h_pcm, frame_pcm * 2 * m))) return rc;
        if ((rc = grow_pinned(ctx, &ctx->h_res, &ctx->cap_h_res, sizeof(int32_t) * m))) return rc;
        const int16_t *h_pcm = (const int16_t *)ctx->h_pcm;
        const int32_t *h_res = (const int32_t *)ctx->h_res;
        // 3. kernels, then results and PCM back to the host.  The PCM comes in pieces, each followed by an event: every packet
        //    owns its own block of the caller's buffer, and the threads that fill the blocks start on a piece as soon as it has
        //    landed, while the later pieces are still on their way.  A large batch goes in parts: a part's pieces travel (on the
        //    copy stream) while the next part's kernels run.  More parts start the copy earlier but pay the parse kernels' fixed
        //    latency once per part: two is the measured optimum at 65,536 frames (9.5 ms; one 11.8, four 10.7, eight 13.2).
        // the modes a range of this step's frames contains (the kernels of the others are not launched).  The range indexes the
        // table that is uploaded: `all` itself in the pipelined flow (single-frame packets in packet order ARE the step table),
        // `step` otherwise -- step[j] = all[first[owner[j]] + k], a different set of frames than all[lo .. hi) as soon as one packet
        // of the call has more than one frame.
        auto modes_of = [&](size_t lo, size_t hi) {
            const opusgpu_frame_desc *table = pipelined ? all : step.data();
            int mask = 0;
            for (size_t f = lo; f < hi && mask != 7; f++) mask |= 1 << (table[f].flags & 3);
            return mask & 7;
        };
        const int pieces = m >= 4096 ? OPUSGPU_COPY_PIECES : 1;
        // (slices cost one more launch of the arithmetic kernels each; parts that are steps of their own pay the entropy kernels'
        // latency each: two of those at most)
        const bool sliced = pipelined && og_debug().host_slices;
        const int parts = !(pieces > 1 && !timer.on) ? 1 : (sliced ? ctx->host_parts : OG_MIN(ctx->host_parts, 2));
        for (int t = 0; t < pieces; t++)
            if (!ctx->ev_piece[t]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_piece[t], hipEventDisableTiming));
        // frames [bound[t], bound[t + 1]) make piece t; a part is pieces / parts consecutive pieces.  Pipelined: a part is a range
        // of PACKETS (its frames: first[] of the range's ends), cut evenly into its pieces.
        size_t bound[OPUSGPU_COPY_PIECES + 1];
        for (int h = 0; h < parts; h++) {
            const int t0 = h * pieces / parts, t1 = (h + 1) * pieces / parts;
            const size_t flo = pipelined ? (size_t)first[(int64_t)n * h / parts] : (size_t)((int64_t)m * t0 / pieces);
            const size_t fhi = pipelined ? (size_t)first[(int64_t)n * (h + 1) / parts] : (size_t)((int64_t)m * t1 / pieces);
            for (int t = t0; t <= t1; t++) bound[t] = flo + (size_t)((int64_t)(fhi - flo) * (t - t0) / (t1 - t0));
        }
        auto piece_lo = [&](int t) { return bound[t]; };
        // DIRECT: the caller's PCM buffer is page-locked (opusgpu_host_register, hipHostMalloc, hipHostRegister) and the step table is
        // the packets in order, one 20 ms block each: the pieces travel straight into it -- no landing zone, no host copy behind it.
        const bool direct = pipelined && m == n && frame_capacity == 1 && !rfc && host_range_is_pinned(ctx, pcm, (size_t)n * cap_pcm * 2);
        auto copy_pieces = [&](hipStream_t cs, int t0, int t1) -> int { // results of the pieces' frames first, then the pieces
            const size_t flo = piece_lo(t0), fhi = piece_lo(t1);
            HIPCHK(ctx, hipMemcpyAsync((int32_t *)ctx->h_res + flo, (const int32_t *)ctx->d_result + flo, sizeof(int32_t) * (fhi - flo),
                                       hipMemcpyDeviceToHost, cs));
            for (int t = t0; t < t1; t++) {
                const size_t lo = piece_lo(t), hi = piece_lo(t + 1);
                HIPCHK(ctx, hipMemcpyAsync((direct ? (uint8_t *)pcm : (uint8_t *)ctx->h_pcm) + lo * frame_pcm * 2,
                                           (const uint8_t *)ctx->d_pcm + lo * frame_pcm * 2, (hi - lo) * frame_pcm * 2, hipMemcpyDeviceToHost, cs));
                HIPCHK(ctx, hipEventRecord(ctx->ev_piece[t], cs));
            }
            return OPUSGPU_OK;
        };
        if (parts > 1) {
            hipStream_t made;
            if (int rc = copy_stream_of(ctx, &made)) return rc; // (opusgpu_upload_async makes it too, from another thread)
            for (int h = 0; h < parts; h++)
                if (!ctx->ev_part[h]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_part[h], hipEventDisableTiming));
        }
        if (parts > 1 && sliced) {
            // SLICES: everything placed and uploaded at once (0.2 ms of host work at 65,536 packets), the entropy kernels once over
            // the whole table, the arithmetic kernels slice by slice with the slice's PCM leaving behind them
            on_ranges(place);
            timer.mark("  all packets placed");
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_arena, arena, base[n], hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_descs, all, sizeof(opusgpu_frame_desc) * m, hipMemcpyHostToDevice, ctx->stream));
            size_t cut[OPUSGPU_COPY_PIECES + 1];
            for (int h = 0; h <= parts; h++) cut[h] = piece_lo(h * pieces / parts);
            StepSlices sl;
            sl.count = parts;
            sl.bounds = cut;
            sl.after_slice = [&](int h) -> int {
                HIPCHK(ctx, hipEventRecord(ctx->ev_part[h], ctx->stream));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_part[h], 0));
                return copy_pieces(ctx->copy_stream, h * pieces / parts, (h + 1) * pieces / parts);
            };
            rc = decode_step_impl(ctx, m, ctx->d_descs, ctx->d_arena, ctx->d_pcm, ctx->d_result, nullptr, false, modes_of(0, m), 0, &sl);
            if (rc) return rc;
            timer.mark("  uploaded, launched, copies queued");
        } else if (parts > 1) {
            // (A/B flow, OPUSGPU_HOST_SLICES=0: every part its own in-order step -- the entropy kernels' latency is paid per part.
            //  Measured also: the parts' chains alternating between two streams, 6.8 - 7.0 ms per 65,536 packets like the slices.)
            for (int h = 0; h < parts; h++) {
                const int t0 = h * pieces / parts, t1 = (h + 1) * pieces / parts;
                const size_t flo = piece_lo(t0), fhi = piece_lo(t1);
                hipStream_t const q = ctx->stream;
                if (pipelined) { // this part's packets: place, upload
                    const int plo = (int)((int64_t)n * h / parts), phi = (int)((int64_t)n * (h + 1) / parts);
                    on_subranges(plo, phi, place);
                    timer.mark("  part placed");
                    if (base[phi] > base[plo])
                        HIPCHK(ctx, hipMemcpyAsync((uint8_t *)ctx->d_arena + base[plo], arena + base[plo], base[phi] - base[plo],
                                                   hipMemcpyHostToDevice, q));
                    if (fhi > flo)
                        HIPCHK(ctx, hipMemcpyAsync((opusgpu_frame_desc *)ctx->d_descs + flo, all + flo,
                                                   sizeof(opusgpu_frame_desc) * (fhi - flo), hipMemcpyHostToDevice, q));
                }
                rc = decode_step_impl(ctx, (int)(fhi - flo), (const opusgpu_frame_desc *)ctx->d_descs + flo, ctx->d_arena,
                                      (uint8_t *)ctx->d_pcm + flo * frame_pcm * 2, (int32_t *)ctx->d_result + flo, nullptr, false, modes_of(flo, fhi));
                if (rc) return rc; // (an empty part launches nothing; its pieces' events are still recorded below)
                HIPCHK(ctx, hipEventRecord(ctx->ev_part[h], q));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_part[h], 0));
                if ((rc = copy_pieces(ctx->copy_stream, t0, t1))) return rc;
                timer.mark("  part uploaded, launched, copies queued");
            }

            timer.mark("table upload + kernels + copy-back in parts (enqueue)");
        } else {
            rc = decode_step_impl(ctx, m, ctx->d_descs, ctx->d_arena, ctx->d_pcm, ctx->d_result, nullptr, false, modes_of(0, m));
            if (rc) return rc;
            timer.mark("table upload + kernels (enqueue)");
            if (timer.on) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                timer.mark("kernels (wait)");
            }
            if ((rc = copy_pieces(ctx->stream, 0, pieces))) return rc;
        }
        timer.mark("copy-back (enqueue)");
        // every thread takes its share of every piece: the work left when the last piece lands is 1 / pieces of the PCM,
        // spread over all threads
        const int threads = pieces == 1 ? 1 : OPUSGPU_COPY_THREADS;
        hipError_t thread_err[OPUSGPU_COPY_THREADS];
        auto deliver = [&](int t) {
            thread_err[t] = hipSuccess;
            for (int p = 0; p < pieces; p++) {
                if ((thread_err[t] = hipEventSynchronize(ctx->ev_piece[p])) != hipSuccess) return;
                const int64_t plo = (int64_t)bound[p], phi = (int64_t)bound[p + 1];
                const int lo = (int)(plo + (phi - plo) * t / threads), hi = (int)(plo + (phi - plo) * (t + 1) / threads);
                for (int j = lo; j < hi; j++) {
                    const int i = regular ? j : owner[j];
                    if (h_res[j] < 0) {
                        result[i] = h_res[j];
                        if (direct) memset(pcm + (size_t)i * cap_pcm, 0, frame_pcm * 2); // (whatever the device buffer held: not the caller's)
                        continue;
                    }
                    if (direct) { // the PCM is in place already
                        result[i] += h_res[j];
                        continue;
                    }
                    if (rfc) { // a packet appears once per step: nobody else touches placed[i]
                        memcpy(pcm + (size_t)i * cap_pcm + (size_t)placed[i] * CC, &h_pcm[(size_t)j * frame_pcm], (size_t)h_res[j] * CC * 2);
                        placed[i] += h_res[j];
                    } else
                        memcpy(pcm + (size_t)i * cap_pcm + (size_t)k * frame_pcm, &h_pcm[(size_t)j * frame_pcm], frame_pcm * 2);
                    result[i] += h_res[j];
                }
            }
        };
        if (threads == 1)
            deliver(0);
        else {
            ctx->pool.run(threads, deliver);
        }
        for (int t = 0; t < threads; t++)
            if (thread_err[t] != hipSuccess) return fail(ctx, OPUSGPU_ERR_HIP, "hipEventSynchronize (PCM piece)", thread_err[t]);
        timer.mark("copy-back + delivery (wait)");
    }
    return OPUSGPU_OK;
}

int opusgpu_decode_packets(opusgpu_ctx *ctx, int n, const int32_t *stream_ids, const uint8_t *const *packets,
                           const int32_t *lens, int16_t *pcm, int frame_capacity, int32_t *result) {
    return decode_packets_impl(ctx, n, stream_ids, packets, lens, pcm, frame_capacity, result, false);
}
int opusgpu_decode_packets_fec(opusgpu_ctx *ctx, int n, const int32_t *stream_ids, const uint8_t *const *packets,
                               const int32_t *lens, int16_t *pcm, int frame_capacity, int32_t *result) {
    return decode_packets_impl(ctx, n, stream_ids, packets, lens, pcm, frame_capacity, result, true);
}

} // extern "C"
