// og_container.hpp -- host-side Ogg Opus container reader (the reference keeps this on the host too).
//
// Re-implements, for the unseekable streaming case the player uses, what the reference's
// src/ogg.cpp (page sync + CRC :439-480, :839-923; lacing -> packets :969-1097, :1192) and
// src/opusfile.cpp (header fetch :154-241, first-page timestamping :486-632, per-page granule
// bookkeeping and end-trimming :835-1133, pre-skip + decode-into-scratch :1171-1291, op_read_stereo
// :1293-1331, OpusHead :1333-1385) do.  No codec arithmetic lives here: packets are handed to a decode
// callback with the semantics of opus_multistream_decode; the shipped callback runs on the GPU.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace ogc {

enum { OP_FALSE = -1, OP_EOF = -2, OP_HOLE = -3, OP_EREAD = -128, OP_EFAULT = -129, OP_EIMPL = -130, OP_EINVAL = -131,
       OP_ENOTFORMAT = -132, OP_EBADHEADER = -133, OP_EVERSION = -134, OP_EBADPACKET = -136, OP_EBADTIMESTAMP = -139 };

typedef int (*read_fn)(unsigned char *buf, int nbytes);   // SD_read: >0 bytes, 0 clean EOF, <0 error
typedef int (*decode_fn)(void *user, const uint8_t *pkt, int32_t len, int16_t *pcm, int frame_size);

struct Head {
    int version = 0, channel_count = 0;
    unsigned pre_skip = 0;
    uint32_t input_sample_rate = 0;
    int output_gain = 0, mapping_family = 0, stream_count = 0, coupled_count = 0;
    uint8_t mapping[8] = {0, 1, 0, 0, 0, 0, 0, 0};
};

inline int parse_head(Head *out, const uint8_t *d, size_t len) { // opus_head_parse opusfile.cpp:1333
    Head h;
    if (len < 8 || memcmp(d, "OpusHead", 8) != 0) return OP_ENOTFORMAT;
    if (len < 9) return OP_EBADHEADER;
    h.version = d[8];
    if (h.version > 15) return OP_EVERSION;
    if (len < 19) return OP_EBADHEADER;
    h.channel_count = d[9];
    h.pre_skip = d[10] | d[11] << 8;
    h.input_sample_rate = d[12] | (uint32_t)d[13] << 8 | (uint32_t)d[14] << 16 | (uint32_t)d[15] << 24;
    int g = d[16] | d[17] << 8;
    h.output_gain = (g ^ 0x8000) - 0x8000;
    h.mapping_family = d[18];
    if (h.mapping_family == 0) {
        if (h.channel_count < 1 || h.channel_count > 2) return OP_EBADHEADER;
        if (h.version <= 1 && len > 19) return OP_EBADHEADER;
        h.stream_count = 1;
        h.coupled_count = h.channel_count - 1;
        h.mapping[0] = 0;
        h.mapping[1] = 1;
    } else if (h.mapping_family == 1) {
        if (h.channel_count < 1 || h.channel_count > 8) return OP_EBADHEADER;
        size_t size = 21 + h.channel_count;
        if (len < size || (h.version <= 1 && len > size)) return OP_EBADHEADER;
        h.stream_count = d[19];
        if (h.stream_count < 1) return OP_EBADHEADER;
        h.coupled_count = d[20];
        if (h.coupled_count > h.stream_count) return OP_EBADHEADER;
        for (int c = 0; c < h.channel_count; c++) {
            if (d[21 + c] >= h.stream_count + h.coupled_count && d[21 + c] != 255) return OP_EBADHEADER;
            h.mapping[c] = d[21 + c];
        }
    } else if (h.mapping_family == 255)
        return OP_EIMPL;
    else
        return OP_EBADHEADER;
    if (out) *out = h;
    return 0;
}

// Ogg CRC-32: polynomial 0x04c11db7, MSB first, zero initial value (ogg.cpp:439)
inline uint32_t crc_update(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            table[i] = r;
        }
        init = true;
    }
    while (n--) crc = (crc << 8) ^ table[((crc >> 24) & 0xff) ^ *p++];
    return crc;
}

// duration in 48 kHz samples of an Opus packet, <0 if the TOC sequence is invalid (op_get_packet_duration)
inline int packet_duration(const uint8_t *p, int32_t len) {
    if (len < 1) return -1;
    int spf;
    const uint8_t t = p[0];
    if (t & 0x80) spf = (48000 << ((t >> 3) & 3)) / 400;
    else if ((t & 0x60) == 0x60) spf = (t & 0x08) ? 960 : 480;
    else { int a = (t >> 3) & 3; spf = a == 3 ? 2880 : (48000 << a) / 100; }
    int count = (t & 3) == 0 ? 1 : ((t & 3) != 3 ? 2 : (len < 2 ? -4 : (p[1] & 0x3F)));
    if (count < 0) return count;
    int samples = count * spf;
    return samples * 25 > 48000 * 3 ? -4 : samples;
}

// Granule positions (op_granpos_add / _diff / _cmp, opusfile.cpp:299-400).  A granule position is a 64-bit counter whose
// negative values come AFTER the positive ones, with -1 (= 2^64 - 1) meaning "none": stated here in unsigned arithmetic,
// which has no undefined overflow.  All three tolerate -1 as an operand (the reference asserts against it) so that a
// hostile file cannot reach undefined behaviour.
inline bool gp_add(int64_t *dst, int64_t src, int32_t delta) { // false: the result would cross -1 or drop below 0
    const uint64_t u = (uint64_t)src;
    if (delta >= 0) {
        if (u > UINT64_MAX - 1 - (uint64_t)delta) return false;
        *dst = (int64_t)(u + (uint64_t)delta);
    } else {
        const uint64_t m = (uint64_t)(-(int64_t)delta);
        if (u < m) return false;
        *dst = (int64_t)(u - m);
    }
    return true;
}
inline bool gp_diff(int64_t *delta, int64_t a, int64_t b) { // a - b in counter order; false: it does not fit 64 signed bits
    const uint64_t ua = (uint64_t)a, ub = (uint64_t)b;
    if (ua >= ub) {
        const uint64_t d = ua - ub;
        if (d > (uint64_t)INT64_MAX) return false;
        *delta = (int64_t)d;
    } else {
        const uint64_t d = ub - ua;
        if (d > (uint64_t)INT64_MAX + 1) return false;
        *delta = d == (uint64_t)INT64_MAX + 1 ? INT64_MIN : -(int64_t)d;
    }
    return true;
}
inline int gp_cmp(int64_t a, int64_t b) { return ((uint64_t)a > (uint64_t)b) - ((uint64_t)b > (uint64_t)a); }
// samples still to trim from a packet of `dur` samples when `diff` samples remain before the page's granule position
// (the overflow guard of opusfile.cpp:1066-1070: a hugely negative diff means "far too many")
inline int64_t gp_trim(int dur, int64_t diff) { return diff < 0 && INT64_MAX + diff < dur ? (int64_t)dur + 1 : dur - diff; }

struct Page {
    int header_type = 0;
    int64_t granulepos = -1;
    uint32_t serial = 0, seqno = 0;
    std::vector<uint8_t> lacing, body;
    int header_len = 0;
    bool bos() const { return header_type & 2; }
    bool eos() const { return header_type & 4; }
    bool continued() const { return header_type & 1; }
};

struct Packet {
    std::vector<uint8_t> data;
    int64_t granulepos = -1;
    bool e_o_s = false;
};

class OpusFile {
  public:
    OpusFile(read_fn rd, decode_fn dec, void *user) : rd_(rd), dec_(dec), user_(user) {}
    const Head &head() const { return head_; }
    bool ready() const { return ready_; }

    // opus_init_decoder's container half: headers + first audio page (opusfile.cpp:730-767)
    int open() {
        Page og;
        int64_t r = next_page(og);
        if (r < 0) return r == OP_FALSE ? OP_ENOTFORMAT : (int)r;
        // BOS pages: find the Opus stream (op_fetch_headers_impl :154)
        bool have_head = false;
        while (og.bos()) {
            if (!have_head) {
                serial_ = og.serial;
                reset_stream();
                std::vector<Packet> pk;
                page_in(og, pk, nullptr);
                if (!pk.empty()) {
                    int ret = parse_head(&head_, pk[0].data.data(), pk[0].data.size());
                    if (ret >= 0) have_head = true;
                    else if (ret != OP_ENOTFORMAT) return ret;
                }
            }
            if (next_page(og) < 0) return have_head ? OP_EBADHEADER : OP_ENOTFORMAT;
        }
        if (!have_head) return OP_ENOTFORMAT;
        // comment header: first packet after the BOS pages on our stream (its content is not parsed)
        std::vector<Packet> pk;
        for (;;) {
            if (og.serial == serial_) {
                bool hole = false;
                page_in(og, pk, &hole);
                if (hole) return OP_EBADHEADER;
                if (!pk.empty()) break;
            } else if (og.bos())
                return OP_EBADHEADER;
            if (next_page(og) < 0) return OP_EBADHEADER;
        }
        // first page with completed audio packets (op_find_initial_pcm_offset :486)
        int ret = find_initial_pcm_offset();
        if (ret < 0) return ret;
        ready_ = true;
        return 0;
    }

    // op_read_stereo opusfile.cpp:1293 (with op_read_native :1171 inlined for _buf_size == 0)
    int read_stereo(int16_t *pcm, int buf_size) {
        if (!ready_) return OP_EINVAL;
        int ret = fill();
        if (ret < 0) return ret;
        int avail = od_size_ - od_pos_;
        if (avail <= 0) return 0;
        const int nch = head_.channel_count;
        int n = avail < (buf_size >> 1) ? avail : (buf_size >> 1);
        const int16_t *src = od_.data() + (size_t)nch * od_pos_;
        if (nch == 2)
            memcpy(pcm, src, (size_t)n * 2 * sizeof(int16_t));
        else
            for (int i = 0; i < n; i++) pcm[2 * i] = pcm[2 * i + 1] = src[i];
        od_pos_ += n;
        return n;
    }

  private:
    read_fn rd_;
    decode_fn dec_;
    void *user_;
    Head head_;
    bool ready_ = false;
    // sync layer
    std::vector<uint8_t> buf_;
    size_t rpos_ = 0;
    // stream layer
    uint32_t serial_ = 0;
    int64_t expect_seq_ = -1;
    std::vector<uint8_t> partial_;
    bool have_partial_ = false;
    // opusfile layer
    std::vector<Packet> op_;
    size_t op_pos_ = 0;
    int32_t cur_discard_ = 0;
    int64_t prev_packet_gp_ = -1, pcm_start_ = 0;
    std::vector<int16_t> od_;
    int od_pos_ = 0, od_size_ = 0;

    void reset_stream() {
        expect_seq_ = -1;
        partial_.clear();
        have_partial_ = false;
    }

    // ogg_sync_pageseek + op_get_next_page (ogg.cpp:839, opusfile.cpp:63): next CRC-valid page, or
    // OP_FALSE at a clean end of data, OP_EREAD on a read error.
    int64_t next_page(Page &og) {
        for (;;) {
            size_t avail = buf_.size() - rpos_;
            const uint8_t *p = buf_.data() + rpos_;
            bool need_more = true;
            if (avail >= 27) {
                if (memcmp(p, "OggS", 4) != 0) {
                    const void *nx = memchr(p + 1, 'O', avail - 1);
                    rpos_ = nx ? (size_t)((const uint8_t *)nx - buf_.data()) : buf_.size();
                    continue;
                }
                const int nseg = p[26];
                const size_t hlen = 27 + (size_t)nseg;
                if (avail >= hlen) {
                    size_t blen = 0;
                    for (int i = 0; i < nseg; i++) blen += p[27 + i];
                    if (avail >= hlen + blen) {
                        uint8_t hdr[27 + 255];
                        memcpy(hdr, p, hlen);
                        const uint32_t want = hdr[22] | (uint32_t)hdr[23] << 8 | (uint32_t)hdr[24] << 16 | (uint32_t)hdr[25] << 24;
                        hdr[22] = hdr[23] = hdr[24] = hdr[25] = 0;
                        uint32_t crc = crc_update(0, hdr, hlen);
                        crc = crc_update(crc, p + hlen, blen);
                        if (crc != want || p[4] != 0) { // lost sync: look for the next capture pattern
                            rpos_ += 1;
                            continue;
                        }
                        og.header_type = p[5];
                        int64_t gp = 0;
                        for (int i = 7; i >= 0; i--) gp = (int64_t)(((uint64_t)gp << 8) | p[6 + i]);
                        og.granulepos = gp;
                        og.serial = p[14] | (uint32_t)p[15] << 8 | (uint32_t)p[16] << 16 | (uint32_t)p[17] << 24;
                        og.seqno = p[18] | (uint32_t)p[19] << 8 | (uint32_t)p[20] << 16 | (uint32_t)p[21] << 24;
                        og.lacing.assign(p + 27, p + hlen);
                        og.body.assign(p + hlen, p + hlen + blen);
                        og.header_len = (int)hlen;
                        rpos_ += hlen + blen;
                        if (rpos_ > (1u << 16)) { // compact
                            buf_.erase(buf_.begin(), buf_.begin() + rpos_);
                            rpos_ = 0;
                        }
                        return 0;
                    }
                }
            }
            if (need_more) {
                const int chunk = 2048; // OP_READ_SIZE
                size_t old = buf_.size();
                buf_.resize(old + chunk);
                int n = rd_ ? rd_(buf_.data() + old, chunk) : -1;
                if (n < 0) {
                    buf_.resize(old);
                    return OP_EREAD;
                }
                buf_.resize(old + (size_t)n);
                if (n == 0) return OP_FALSE;
            }
        }
    }

    // ogg_stream_pagein + packetout (ogg.cpp:969, :1192): completed packets of this page, in order.
    void page_in(const Page &og, std::vector<Packet> &out, bool *hole) {
        out.clear();
        size_t seg = 0, off = 0;
        const size_t nseg = og.lacing.size();
        bool gap = expect_seq_ >= 0 && (int64_t)og.seqno != expect_seq_;
        expect_seq_ = (int64_t)og.seqno + 1;
        if (gap) {
            partial_.clear();
            have_partial_ = false;
            if (hole) *hole = true;
        }
        if (og.continued()) {
            if (!have_partial_) { // continuation of a packet we never saw the start of: skip it
                while (seg < nseg) {
                    const int l = og.lacing[seg++];
                    off += l;
                    if (l < 255) break;
                }
            }
        } else if (have_partial_) { // the previous packet was never finished
            partial_.clear();
            have_partial_ = false;
        }
        while (seg < nseg) {
            const int l = og.lacing[seg++];
            partial_.insert(partial_.end(), og.body.begin() + off, og.body.begin() + off + l);
            have_partial_ = true;
            off += l;
            if (l < 255) {
                Packet p;
                p.data.swap(partial_);
                partial_.clear();
                have_partial_ = false;
                out.push_back(std::move(p));
            }
        }
        if (!out.empty()) { // granule position and EOS belong to the last packet completed on the page
            out.back().granulepos = og.granulepos;
            if (og.eos() && !have_partial_) out.back().e_o_s = true;
        }
    }

    // op_collect_audio_packets opusfile.cpp:424: drop packets with an invalid TOC sequence
    int32_t collect(std::vector<Packet> &pk, std::vector<int> &dur) {
        int32_t total = 0;
        std::vector<Packet> keep;
        dur.clear();
        for (auto &p : pk) {
            int d = packet_duration(p.data.data(), (int32_t)p.data.size());
            if (d > 0) {
                total += d;
                dur.push_back(d);
                keep.push_back(std::move(p));
            } else if (!keep.empty()) {
                keep.back().granulepos = p.granulepos;
                keep.back().e_o_s = keep.back().e_o_s || p.e_o_s;
            }
        }
        pk.swap(keep);
        return total;
    }

    int find_initial_pcm_offset() {
        Page og;
        std::vector<Packet> pk;
        std::vector<int> dur;
        int32_t total = 0;
        for (;;) {
            int64_t r = next_page(og);
            if (r < 0) {
                if (r < OP_FALSE) return (int)r;
                if (head_.pre_skip > 0) return OP_EBADTIMESTAMP;
                op_.clear();
                return 0;
            }
            if (og.bos()) return head_.pre_skip > 0 ? OP_EBADTIMESTAMP : 0;
            if (og.serial != serial_) continue;
            page_in(og, pk, nullptr);
            total = collect(pk, dur);
            if (!pk.empty()) break;
        }
        const int64_t cur_page_gp = pk.back().granulepos;
        if (cur_page_gp == -1) return OP_EBADTIMESTAMP;
        const bool eos = pk.back().e_o_s;
        int64_t pcm_start;
        if (!gp_add(&pcm_start, cur_page_gp, -total)) { // less audio before the granule position than the page carries
            if (!eos) return OP_EBADTIMESTAMP;
            pcm_start = 0; // end trimming: the stream starts at zero by definition
            if (gp_cmp(cur_page_gp, (int64_t)head_.pre_skip) < 0) return OP_EBADTIMESTAMP;
        }
        int64_t prev = pcm_start;
        size_t pi;
        for (pi = 0; pi < pk.size(); pi++) {
            if (eos) {
                int64_t diff;
                if (!gp_diff(&diff, cur_page_gp, prev)) return OP_EBADTIMESTAMP; // (the reference reads an unset variable here)
                diff = gp_trim(dur[pi], diff);
                if (diff > 0) {
                    if (diff > dur[pi]) break;
                    pk[pi].granulepos = prev = cur_page_gp;
                    pk[pi].e_o_s = true;
                    continue;
                }
            }
            int64_t g;
            if (gp_add(&g, prev, dur[pi])) pk[pi].granulepos = g; // (on overflow the packet keeps what the page gave it)
            prev = pk[pi].granulepos;
        }
        pk.resize(pi);
        op_.swap(pk);
        op_pos_ = 0;
        cur_discard_ = (int32_t)head_.pre_skip;
        prev_packet_gp_ = pcm_start_ = pcm_start;
        return 0;
    }

    // one more page of our stream -> timestamped packets (op_fetch_and_process_page opusfile.cpp:835)
    int fetch_page() {
        for (;;) {
            Page og;
            int64_t r = next_page(og);
            if (r < 0) return r < OP_FALSE ? (int)r : OP_EOF;
            if (og.serial != serial_) {
                if (!og.bos()) continue;
                return OP_EOF; // a new link: chained streams are not followed by this reader
            }
            std::vector<Packet> pk;
            std::vector<int> dur;
            bool hole = false;
            page_in(og, pk, &hole);
            int32_t total = collect(pk, dur);
            if (hole) prev_packet_gp_ = -1;
            if (!pk.empty()) {
                int64_t cur_page_gp = pk.back().granulepos;
                const bool eos = pk.back().e_o_s;
                int64_t prev = prev_packet_gp_;
                if (prev == -1) { // after a hole: restart the timeline from this page
                    if (eos) {
                        if (hole) return OP_HOLE;
                        continue;
                    }
                    prev = pcm_start_; // an unusable granule position: count on from the start of the stream
                    if (cur_page_gp != -1) (void)gp_add(&prev, cur_page_gp, -total);
                    cur_discard_ = 80 * 48;
                }
                // completed packets but no granule position on the page (illegal): count forwards from the previous page
                if (cur_page_gp == -1 && !gp_add(&cur_page_gp, prev, total)) return OP_EBADTIMESTAMP; // timeline exhausted
                size_t pi;
                int64_t diff;
                if (eos && gp_diff(&diff, cur_page_gp, prev) && diff < total) { // end trimming
                    int64_t cur = prev;
                    for (pi = 0; pi < pk.size(); pi++) {
                        diff = gp_trim(dur[pi], diff);
                        if (diff > 0) {
                            if (diff > dur[pi]) break;
                            cur = cur_page_gp;
                            pk[pi].e_o_s = true;
                        } else
                            (void)gp_add(&cur, cur, dur[pi]);
                        pk[pi].granulepos = cur;
                        (void)gp_diff(&diff, cur_page_gp, cur);
                    }
                } else { // timestamps backwards from the page's granule position; an underflowing start counts as 0
                    if (!gp_add(&prev, cur_page_gp, -total)) prev = 0;
                    int32_t left = total;
                    for (pi = 0; pi < pk.size(); pi++) {
                        int64_t cur;
                        if (!gp_add(&cur, cur_page_gp, -left)) cur = 0;
                        left -= dur[pi];
                        (void)gp_add(&cur, cur, dur[pi]);
                        pk[pi].granulepos = cur;
                    }
                }
                prev_packet_gp_ = prev;
                pk.resize(pi);
                op_.swap(pk);
                op_pos_ = 0;
            } else
                op_.clear(), op_pos_ = 0;
            if (hole) return OP_HOLE;
            if (!op_.empty()) return 0;
        }
    }

    // make sure decoded samples are buffered (op_read_native with _buf_size == 0)
    int fill() {
        for (;;) {
            if (od_size_ - od_pos_ > 0) return 0;
            if (op_pos_ < op_.size()) {
                const Packet &pop = op_[op_pos_++];
                const int nch = head_.channel_count;
                const int duration = packet_duration(pop.data.data(), (int32_t)pop.data.size());
                int trimmed = duration;
                if (pop.e_o_s) { // end trimming (opusfile.cpp:1220-1227)
                    int64_t diff;
                    if (gp_cmp(pop.granulepos, prev_packet_gp_) <= 0) trimmed = 0;
                    else if (gp_diff(&diff, pop.granulepos, prev_packet_gp_) && diff < trimmed) trimmed = (int)diff;
                }
                prev_packet_gp_ = pop.granulepos;
                if (od_.size() < (size_t)nch * 5760) od_.resize((size_t)nch * 5760); // 120 ms at 48 kHz
                int ret = dec_(user_, pop.data.data(), (int32_t)pop.data.size(), od_.data(), duration);
                if (ret < 0) return OP_EBADPACKET;
                int skip = trimmed < cur_discard_ ? trimmed : cur_discard_;
                cur_discard_ -= skip;
                od_pos_ = skip;
                od_size_ = trimmed;
                continue;
            }
            int ret = fetch_page();
            if (ret == OP_EOF) return 0;
            if (ret < 0) return ret;
        }
    }
};

} // namespace ogc
