// og_pages.cpp -- batched Ogg page ingest on the host (include/opusgpu.h, "Ogg page ingest at scale").
//
// What the reference does for one page at a time in src/ogg.cpp (ogg_sync_pageseek :839-923, the page checksum
// :439-480, ogg_stream_pagein / ogg_stream_packetout :969-1097, :1192) and src/opusfile.cpp (op_collect_audio_packets
// :424-466), done here for hundreds of thousands of independent pages per call and laid out as decode steps for the
// GPU.  No codec arithmetic: header fields, a CRC, the lacing walk and the TOC split (og_packet.hpp).
//
// Three passes: (1) per page, in parallel: validate, CRC, count packets / frames / body bytes; (2) sequential and
// cheap: chain pages of one stream, prefix sums, per-step (and per-mode) slot assignment; (3) per page, in parallel:
// copy the body into the arena and write the frame descriptors into their slots.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <new>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unordered_map>
#include <vector>
#include "og_packet.hpp"
#include "og_debug.hpp"
#include "../../include/opusgpu.h"

namespace {

// Ogg CRC-32 (polynomial 0x04c11db7, MSB first, zero initial value, ogg.cpp:439), eight bytes per round
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            t[0][i] = r;
        }
        for (int j = 1; j < 8; j++)
            for (uint32_t i = 0; i < 256; i++) t[j][i] = (t[j - 1][i] << 8) ^ t[0][t[j - 1][i] >> 24];
    }
};
const CrcTables g_crc;

uint32_t crc_run(uint32_t crc, const uint8_t *p, size_t n) {
    while (n >= 8) {
        const uint32_t a = crc ^ ((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]);
        crc = g_crc.t[7][a >> 24] ^ g_crc.t[6][(a >> 16) & 255] ^ g_crc.t[5][(a >> 8) & 255] ^ g_crc.t[4][a & 255] ^
              g_crc.t[3][p[4]] ^ g_crc.t[2][p[5]] ^ g_crc.t[1][p[6]] ^ g_crc.t[0][p[7]];
        p += 8;
        n -= 8;
    }
    while (n--) crc = (crc << 8) ^ g_crc.t[0][(crc >> 24) ^ *p++];
    return crc;
}

// the page CRC covers the whole page with the four checksum bytes (22..25) taken as zero (ogg.cpp:451-470)
uint32_t page_crc(const uint8_t *page, size_t len) {
    static const uint8_t zero[4] = {0, 0, 0, 0};
    uint32_t c = crc_run(0, page, 22);
    c = crc_run(c, zero, 4);
    return crc_run(c, page + 26, len - 26);
}

// duration in 48 kHz samples of an Opus packet, <= 0 if the TOC sequence is invalid (op_get_packet_duration,
// opusfile.cpp: frames * samples per frame, at most 120 ms)
int packet_duration(const uint8_t *p, int32_t len) {
    if (len < 1) return -1;
    const int spf = ogh::toc_samples_per_frame(p[0], 48000);
    const int count = (p[0] & 3) == 0 ? 1 : ((p[0] & 3) != 3 ? 2 : (len < 2 ? -1 : (p[1] & 0x3F)));
    if (count < 0) return -1;
    const int samples = count * spf;
    return samples * 25 > 48000 * 3 ? -1 : samples;
}

struct PageScan {
    int32_t status = 0; // frames (>= 0) or OPUSGPU_PAGE_*
    int32_t packets = 0;
    int32_t header_len = 0, body_len = 0;
    uint64_t modes = 0; // mode (2 bits) of the page's first 32 frames: enough for the mode grouping of ordinary pages
};

// Uninitialised storage for the large outputs: a zero-filling resize would touch (page-fault) every byte on one thread
// before the parallel pass that fills them gets to run.
template <class T>
struct RawBuf {
    T *p = nullptr;
    size_t n = 0;
    bool borrowed = false; // caller's memory (opusgpu_pages_demux_into): not freed here
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() {
        if (!borrowed) free(p);
    }
    void borrow(void *mem, size_t count) {
        if (!borrowed) free(p);
        p = static_cast<T *>(mem);
        n = count;
        borrowed = true;
    }
    void alloc(size_t count) {
        if (!borrowed) free(p);
        borrowed = false;
        p = nullptr;
        n = count;
        const size_t bytes = (count ? count : 1) * sizeof(T), huge = (size_t)2 << 20;
        if (bytes >= 4 * huge) { // first touch of a large buffer is mostly page faults: ask for 2 MB pages where the host allows
            void *q = nullptr;
            const size_t rounded = (bytes + huge - 1) / huge * huge;
            if (posix_memalign(&q, huge, rounded) == 0) {
                (void)madvise(q, rounded, MADV_HUGEPAGE);
                p = static_cast<T *>(q);
            }
        }
        if (!p) p = static_cast<T *>(malloc(bytes));
        if (!p) throw std::bad_alloc();
    }
    T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) const { return p[i]; }
};

// Header checks, CRC, lacing walk.  With `emit`: called per frame as emit(frame k of the page, offset in body, len, flags).
template <class Emit>
PageScan scan_page(const uint8_t *pg, int32_t len, int flags, opusgpu_page_info *info, Emit emit) {
    PageScan r;
    if (len < 27 || memcmp(pg, "OggS", 4) != 0 || pg[4] != 0) { // capture pattern, version (ogg.cpp:856-866)
        r.status = OPUSGPU_PAGE_BAD_CAPTURE;
        return r;
    }
    const int nseg = pg[26];
    r.header_len = 27 + nseg;
    if (len < r.header_len) {
        r.status = OPUSGPU_PAGE_BAD_CAPTURE;
        return r;
    }
    const uint8_t *lace = pg + 27;
    for (int i = 0; i < nseg; i++) r.body_len += lace[i];
    if (len < r.header_len + r.body_len) { // a truncated page never syncs (ogg.cpp:880)
        r.status = OPUSGPU_PAGE_BAD_CAPTURE;
        return r;
    }
    if (info) {
        info->header_type = pg[5];
        uint64_t g = 0;
        for (int i = 7; i >= 0; i--) g = g << 8 | pg[6 + i];
        info->granulepos = (int64_t)g;
        info->serial = (uint32_t)pg[14] | (uint32_t)pg[15] << 8 | (uint32_t)pg[16] << 16 | (uint32_t)pg[17] << 24;
        info->seqno = (uint32_t)pg[18] | (uint32_t)pg[19] << 8 | (uint32_t)pg[20] << 16 | (uint32_t)pg[21] << 24;
    }
    if (flags & OPUSGPU_PAGES_VERIFY_CRC) { // stored little-endian at 22..25 (ogg.cpp:472-476)
        const uint32_t want = (uint32_t)pg[22] | (uint32_t)pg[23] << 8 | (uint32_t)pg[24] << 16 | (uint32_t)pg[25] << 24;
        if (page_crc(pg, (size_t)r.header_len + r.body_len) != want) {
            r.status = OPUSGPU_PAGE_BAD_CRC;
            return r;
        }
    }
    // whole packets only: not a continuation (header_type bit 0), last packet terminated (last lacing value < 255)
    if ((pg[5] & 1) || (nseg > 0 && lace[nseg - 1] == 255)) {
        r.status = OPUSGPU_PAGE_SPANS;
        return r;
    }
    const uint8_t *body = pg + r.header_len;
    int32_t at = 0, frames = 0;
    for (int i = 0; i < nseg;) { // a packet = lacing values up to and including the first one < 255 (ogg.cpp:1040-1075)
        int32_t plen = 0;
        while (i < nseg && lace[i] == 255) plen += lace[i++];
        plen += lace[i++];
        r.packets++;
        if (packet_duration(body + at, plen) <= 0) { // "Ignore packets with an invalid TOC sequence" (opusfile.cpp:453-459)
            at += plen;
            continue;
        }
        int16_t size[48];
        uint8_t toc;
        int off = 0;
        const int count = ogh::parse_packet(body + at, plen, 0, &toc, size, &off, nullptr);
        if (count < 0) { // the reference would decode the page's earlier packets and fail on this one; here the page is dropped
            r.status = OPUSGPU_PAGE_BAD_PACKET;
            return r;
        }
        const int32_t fl = ogh::toc_flags(toc);
        for (int k = 0; k < count; k++) {
            emit(frames + k, at + off, (int32_t)size[k], fl);
            if (frames + k < 32) r.modes |= (uint64_t)(fl & 3) << (2 * (frames + k));
            off += size[k];
        }
        frames += count;
        at += plen;
    }
    r.status = frames;
    return r;
}

// OPUSGPU_PAGES_TIMING=1: wall time of every phase of a demux call on stderr (for tuning; off by default)
struct PhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseTimer() : on(og_debug().pages_timing != 0), t(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[opusgpu_pages_demux] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

template <class F>
void parallel_for(int n, int threads, F f, int grain = 256) { // f(begin, end); at least `grain` items per thread
    threads = threads < 1 ? 1 : threads;
    if (threads > n / grain + 1) threads = n / grain + 1;
    if (threads <= 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) {
        const int b = (int)((int64_t)n * t / threads), e = (int)((int64_t)n * (t + 1) / threads);
        th.emplace_back([=] { f(b, e); });
    }
    for (auto &x : th) x.join();
}

} // namespace

struct opusgpu_page_batch {
    RawBuf<opusgpu_frame_desc> descs; // all steps, step after step
    RawBuf<int32_t> slot_pages;       // parallel to descs
    std::vector<size_t> step_begin;   // n_steps + 1
    RawBuf<uint8_t> arena;
    size_t arena_offset = 0; // opusgpu_pages_demux_into: where the arena begins in the caller's memory
};

static int pages_demux_impl(int n_pages, const uint8_t *const *pages, const int32_t *page_lens, const int32_t *stream_ids, int flags,
                            int threads, opusgpu_page_info *info, opusgpu_page_batch **out, void *out_mem, size_t out_cap, size_t *out_need);

int opusgpu_pages_demux(int n_pages, const uint8_t *const *pages, const int32_t *page_lens, const int32_t *stream_ids,
                        int flags, int threads, opusgpu_page_info *info, opusgpu_page_batch **out) {
    return pages_demux_impl(n_pages, pages, page_lens, stream_ids, flags, threads, info, out, nullptr, 0, nullptr);
}
int opusgpu_pages_demux_into(int n_pages, const uint8_t *const *pages, const int32_t *page_lens, const int32_t *stream_ids,
                             int flags, int threads, opusgpu_page_info *info, void *out_mem, size_t out_cap, size_t *out_need,
                             opusgpu_page_batch **out) {
    if (!out_mem || ((uintptr_t)out_mem & 15)) return OPUSGPU_BAD_ARG;
    return pages_demux_impl(n_pages, pages, page_lens, stream_ids, flags, threads, info, out, out_mem, out_cap, out_need);
}
size_t opusgpu_page_batch_arena_offset(const opusgpu_page_batch *b) { return b ? b->arena_offset : 0; }

static int pages_demux_impl(int n_pages, const uint8_t *const *pages, const int32_t *page_lens, const int32_t *stream_ids, int flags,
                            int threads, opusgpu_page_info *info, opusgpu_page_batch **out, void *out_mem, size_t out_cap, size_t *out_need) {
    if (!out) return OPUSGPU_BAD_ARG;
    *out = nullptr;
    if (out_need) *out_need = 0;
    if (n_pages < 0 || (n_pages > 0 && (!pages || !page_lens || !stream_ids))) return OPUSGPU_BAD_ARG;
    try {
        opusgpu_page_batch *b = new opusgpu_page_batch;
        PhaseTimer timer;
        std::vector<PageScan> scan((size_t)n_pages);
        timer.mark("allocate page scans");
        // pass 1: validate and count
        parallel_for(n_pages, threads, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++) {
                opusgpu_page_info *pi = info ? &info[i] : nullptr;
                if (pi) memset(pi, 0, sizeof *pi);
                if (stream_ids[i] < 0 || !pages[i]) {
                    scan[i].status = stream_ids[i] < 0 ? OPUSGPU_PAGE_BAD_STREAM : OPUSGPU_PAGE_BAD_CAPTURE;
                } else
                    scan[i] = scan_page(pages[i], page_lens[i], flags, pi, [](int, int32_t, int32_t, int32_t) {});
                if (pi) {
                    pi->status = scan[i].status;
                    pi->packets = scan[i].packets;
                }
            }
        });
        timer.mark("pass 1: validate");
        // pass 2: chain the pages of one stream, size the steps, place every page's body in the arena
        std::vector<int32_t> first_step((size_t)n_pages, 0);
        std::vector<size_t> arena_at((size_t)n_pages, 0);
        // a stream's running step count: a table indexed by stream id when the ids are dense enough for one (the usual
        // case: ids 0 .. n_streams - 1), a hash map otherwise -- at a quarter of a million pages the map alone took
        // longer than the parallel passes
        int32_t max_id = -1;
        for (int i = 0; i < n_pages; i++)
            if (scan[i].status > 0 && stream_ids[i] > max_id) max_id = stream_ids[i];
        const bool dense = (int64_t)max_id < 16 * (int64_t)n_pages + 4096;
        std::vector<int32_t> next_dense(dense ? (size_t)max_id + 1 : 0, 0);
        std::unordered_map<int32_t, int32_t> next_sparse;
        if (!dense) next_sparse.reserve((size_t)n_pages);
        int32_t n_steps = 0;
        size_t arena_bytes = 0;
        for (int i = 0; i < n_pages; i++) {
            if (scan[i].status <= 0) continue;
            int32_t &ns = dense ? next_dense[(size_t)stream_ids[i]] : next_sparse[stream_ids[i]];
            first_step[i] = ns;
            ns += scan[i].status;
            if (ns > n_steps) n_steps = ns;
            arena_at[i] = arena_bytes;
            arena_bytes += (size_t)scan[i].body_len;
            if (info) info[i].first_step = first_step[i];
        }
        timer.mark("pass 2: chain streams");
        // slots: per step, pages in input order -- or grouped by mode first (three stable groups)
        const bool group = (flags & (OPUSGPU_PAGES_GROUP_BY_MODE | OPUSGPU_PAGES_ORDER_BY_HEADER)) != 0;
        // ORDER_BY_HEADER: within a step's SILK-only group and its hybrid group, frames in the order of their LBRR flags -- the range
        // coder's second and fourth symbol, each of probability 1/2, i.e. bits 6 and 4 of the frame's first byte (reference
        // src/silk.cpp:1568-1573; a mono frame has only the first) -- stable otherwise.  An LBRR frame is a whole extra frame of side
        // information and pulses to read past (:1590-1616): 32 frames that agree on it make a parse wave that skips those passes
        // together.  The flags are four sub-keys of the counting sort below (SILK 0..3, hybrid 4..7, CELT 8), not a pass of their own.
        // (Bits 6 and 4 are where a frame DECODED AS 20 ms has them: one VAD bit, then the LBRR flag, per channel.  Reference mode
        // decodes every frame so, whatever duration its TOC names (Q6), and these steps exist in reference mode only; a 40 / 60 ms frame
        // decoded at its true duration would have two / three VAD bits in front of the flag.  The order never changes a result.)
        const bool by_header = (flags & OPUSGPU_PAGES_ORDER_BY_HEADER) != 0;
        const int G = group ? (by_header ? 9 : 3) : 1;
        RawBuf<uint8_t> mode_of; // sort key within a step (mode 0..2, or the nine keys above) per frame of every page, only when grouping
        std::vector<size_t> frame_at((size_t)n_pages + 1, 0);
        for (int i = 0; i < n_pages; i++) {
            frame_at[i + 1] = frame_at[i] + (scan[i].status > 0 ? scan[i].status : 0);
        }
        const size_t total = frame_at[n_pages];
        if (total > 0xffffffffull) { // slots are 32-bit: split the call
            delete b;
            return OPUSGPU_BAD_ARG;
        }
        if (group) {
            mode_of.alloc(total);
            parallel_for(n_pages, threads, [&](int lo, int hi) {
                for (int i = lo; i < hi; i++) {
                    if (scan[i].status <= 0) continue;
                    if (by_header) { // the key needs every frame's first byte: the lacing walk again (no checksum this time)
                        const uint8_t *body = pages[i] + scan[i].header_len;
                        scan_page(pages[i], page_lens[i], 0, nullptr, [&](int k, int32_t off, int32_t len, int32_t fl) {
                            const int mode = fl & 3;
                            int key = mode == 2 ? 8 : 4 * mode;
                            if (mode != 2 && len > 0) {
                                const uint8_t b0 = body[off];
                                key += (int)((b0 >> 6) & 1) | ((fl & 32) ? (int)((b0 >> 4) & 1) << 1 : 0);
                            }
                            mode_of[frame_at[i] + k] = (uint8_t)key;
                        });
                        continue;
                    }
                    const int nf = scan[i].status < 32 ? scan[i].status : 32;
                    for (int k = 0; k < nf; k++) mode_of[frame_at[i] + k] = (uint8_t)((scan[i].modes >> (2 * k)) & 3);
                    if (scan[i].status > 32) // rare: scan the page again for the rest
                        scan_page(pages[i], page_lens[i], 0, nullptr, [&](int k, int32_t, int32_t, int32_t fl) {
                            mode_of[frame_at[i] + k] = (uint8_t)(fl & 3);
                        });
                }
            });
        }
        timer.mark("frame index + modes");
        // A counting sort of the frames by (step, group), stable in input order, done by page ranges: every range counts
        // its frames per key, the ranges' counts are chained per key, and every range then numbers its own frames.
        const size_t keys = (size_t)n_steps * G;
        int chunks = threads < 1 ? 1 : threads;
        if (chunks > n_pages / 256 + 1) chunks = n_pages / 256 + 1;
        if (keys * (size_t)chunks > ((size_t)1 << 22)) chunks = 1; // (very many steps: not worth a table per range)
        auto chunk_lo = [&](int c) { return (int)((int64_t)n_pages * c / chunks); };
        auto key_of = [&](int i, int k) { return (size_t)(first_step[i] + k) * G + (group ? mode_of[frame_at[i] + k] : 0); };
        std::vector<size_t> cur((size_t)chunks * keys, 0); // [range][key]: frames counted, then the next free slot
        parallel_for(chunks, chunks, [&](int lo, int hi) {
            for (int c = lo; c < hi; c++) {
                size_t *h = cur.data() + (size_t)c * keys;
                for (int i = chunk_lo(c); i < chunk_lo(c + 1); i++)
                    for (int k = 0; k < (scan[i].status > 0 ? scan[i].status : 0); k++) h[key_of(i, k)]++;
            }
        }, 1);
        std::vector<size_t> count(keys + 1, 0); // count[j] = first slot of (step, group) j
        {
            size_t at = 0;
            for (size_t j = 0; j < keys; j++) {
                count[j] = at;
                for (int c = 0; c < chunks; c++) {
                    const size_t mine = cur[(size_t)c * keys + j];
                    cur[(size_t)c * keys + j] = at;
                    at += mine;
                }
            }
            count[keys] = at;
        }
        RawBuf<uint32_t> slot_of;
        slot_of.alloc(total);
        parallel_for(chunks, chunks, [&](int lo, int hi) {
            for (int c = lo; c < hi; c++) {
                size_t *h = cur.data() + (size_t)c * keys;
                for (int i = chunk_lo(c); i < chunk_lo(c + 1); i++)
                    for (int k = 0; k < (scan[i].status > 0 ? scan[i].status : 0); k++)
                        slot_of[frame_at[i] + k] = (uint32_t)h[key_of(i, k)]++;
            }
        }, 1);
        timer.mark("slot numbering");
        b->step_begin.resize((size_t)n_steps + 1);
        for (int s = 0; s <= n_steps; s++) b->step_begin[s] = count[(size_t)s * G];
        b->slot_pages.alloc(total);
        if (out_mem) { // the caller's memory: [descriptors of all steps | padding to 256 | arena + 16]
            const size_t arena_off = ((size_t)total * sizeof(opusgpu_frame_desc) + 255) / 256 * 256, need = arena_off + arena_bytes + 16;
            if (out_need) *out_need = need;
            if (need > out_cap) {
                delete b;
                return OPUSGPU_BUFFER_TOO_SMALL;
            }
            b->descs.borrow(out_mem, total);
            b->arena.borrow(static_cast<uint8_t *>(out_mem) + arena_off, arena_bytes + 16);
            b->arena_offset = arena_off;
        } else {
            b->descs.alloc(total);
            b->arena.alloc(arena_bytes + 16); // the kernels read packets through aligned 32-bit words: keep a tail
        }
        memset(b->arena.data() + arena_bytes, 0, 16);
        timer.mark("allocate outputs");
        // pass 3: bodies and descriptors
        parallel_for(n_pages, threads, [&](int lo, int hi) {
            for (int i = lo; i < hi; i++) {
                if (scan[i].status <= 0) continue;
                memcpy(b->arena.data() + arena_at[i], pages[i] + scan[i].header_len, (size_t)scan[i].body_len);
                scan_page(pages[i], page_lens[i], 0, nullptr, [&](int k, int32_t off, int32_t len, int32_t fl) {
                    const size_t slot = slot_of[frame_at[i] + k];
                    opusgpu_frame_desc &d = b->descs[slot];
                    d.stream = stream_ids[i];
                    d.offset = (int32_t)(arena_at[i] + (size_t)off);
                    d.len = len;
                    d.flags = fl;
                    b->slot_pages[slot] = i;
                });
            }
        });
        timer.mark("pass 3: bodies + descriptors");
        if (arena_bytes > 0x7fffffffu) { // descriptor offsets are 32-bit: split the call
            delete b;
            return OPUSGPU_BAD_ARG;
        }
        *out = b;
        return OPUSGPU_OK;
    } catch (const std::bad_alloc &) {
        return OPUSGPU_ALLOC_FAIL;
    }
}

int opusgpu_page_batch_steps(const opusgpu_page_batch *b) { return b ? (int)b->step_begin.size() - 1 : OPUSGPU_BAD_ARG; }

int opusgpu_page_batch_step(const opusgpu_page_batch *b, int step, const opusgpu_frame_desc **descs,
                            const int32_t **slot_pages) {
    if (!b || step < 0 || step + 1 >= (int)b->step_begin.size()) return OPUSGPU_BAD_ARG;
    const size_t at = b->step_begin[step];
    if (descs) *descs = b->descs.data() + at;
    if (slot_pages) *slot_pages = b->slot_pages.data() + at;
    return (int)(b->step_begin[step + 1] - at);
}

const uint8_t *opusgpu_page_batch_arena(const opusgpu_page_batch *b, size_t *bytes) {
    if (!b) return nullptr;
    if (bytes) *bytes = b->arena.size();
    return b->arena.data();
}

void opusgpu_page_batch_free(opusgpu_page_batch *b) { delete b; }
