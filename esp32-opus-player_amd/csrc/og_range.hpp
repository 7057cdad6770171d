// og_range.hpp -- the Opus range decoder, wave-uniform.
//
// Every lane of the wave runs this code on identical values (the packet bytes sit in LDS at
// S.pkt, the coder state in registers), so the compiler keeps it on the scalar path.  Behaviour
// follows the reference's ec_* functions (src/celt.cpp:2627-2792, :3041-3083); the state here is
// a per-wave value instead of the reference's process-wide `s_ec`.
#pragma once
#include "og_state.hpp"

namespace og {

OG_LDS FrameLds S; // the wave's LDS working set (one workgroup == one wave == one frame)

struct Rc {
    u32 storage, end_offs, end_window;
    i32 nend_bits, nbits_total;
    u32 offs, rng, val, ext;
    i32 rem, error;
};
// Lane-private flavour for the one-frame-per-LANE parse kernel: the same coder state, packet bytes read from
// HBM through a per-lane pointer (every rc_* function below is a template over the two).
// On the GPU a byte fetch straight from HBM would put a full memory latency on the decoder's critical path at every
// renormalisation, so the packet is read through two small windows -- one walking forward from the first byte (range-coded
// data), one walking backward from the last (raw bits) -- each holding the current piece and, already requested, the next one in
// its direction.  Only pieces that overlap the packet are ever loaded.
// The forward window moves in aligned 16-BYTE pieces (round 4; 4-byte words before).  The 64 lanes of a parse wave cross their
// piece boundaries at 64 different moments, each crossing requests the lane's next piece, and the wait for a piece is a wait for
// EVERY load of the wave in flight (the counter is the wave's, in order): with 4-byte pieces a wave of 160-byte packets crossed
// 658 times per 64 frames, nearly every crossing waited for a load another lane had requested a moment before, and the parse
// kernel sat in such waits for 47 % of its life (profiles/r04: SQ_WAIT_ANY 3,343 of 7,050 wave cycles per frame).  The backward
// window (a few raw bits per band) keeps 4-byte words.
typedef u32 og_u32x4 __attribute__((ext_vector_type(4)));
struct RcLane : Rc {
    const u8 *buf;
#ifndef OG_HOST_EMUL
    // buf rounded down to a 16- / 4-byte boundary.  Explicitly global-memory pointers: through generic ones these prefetches
    // become flat loads, which also count against lgkmcnt -- every LDS wait would then wait for the prefetch as well.
    const __attribute__((address_space(1))) og_u32x4 *chunks;
    const __attribute__((address_space(1))) u32 *words;
    u32 fshift, shift;            // buf - (const u8 *)chunks, buf - (const u8 *)words
    i32 last_chunk, last_word;    // index of the last piece of either size that overlaps the packet
    og_u32x4 f_cur, f_next;
    u32 b_cur, b_next;
    i32 f_idx, b_idx; // piece indices of f_cur / b_cur
#endif
};

OG_DEV int rc_next_byte(Rc &rc) { return rc.offs < rc.storage ? S.pkt[rc.offs++] : 0; }          // :2642
OG_DEV int rc_next_byte_end(Rc &rc) {                                                           // :2644
    return rc.end_offs < rc.storage ? S.pkt[rc.storage - ++rc.end_offs] : 0;
}
#ifdef OG_HOST_EMUL
OG_DEV void rc_lane_attach(RcLane &rc, const u8 *buf, u32 len) {
    rc.buf = buf;
    (void)len;
}
OG_DEV void rc_lane_resume(RcLane &rc) {}
OG_DEV int rc_next_byte(RcLane &rc) { return rc.offs < rc.storage ? rc.buf[rc.offs++] : 0; }
OG_DEV int rc_next_byte_end(RcLane &rc) { return rc.end_offs < rc.storage ? rc.buf[rc.storage - ++rc.end_offs] : 0; }
#else
OG_DEV u32 rc_lane_word(const RcLane &rc, i32 w) { return (w >= 0 && w <= rc.last_word) ? rc.words[w] : 0u; }
OG_DEV og_u32x4 rc_lane_chunk(const RcLane &rc, i32 w) { return (w >= 0 && w <= rc.last_chunk) ? rc.chunks[w] : og_u32x4{0u, 0u, 0u, 0u}; }
OG_DEV void rc_lane_attach(RcLane &rc, const u8 *buf, u32 len) { // call before rc_init
    rc.buf = buf;
    const unsigned long long a = (unsigned long long)buf;
    rc.chunks = (const __attribute__((address_space(1))) og_u32x4 *)(a & ~15ull);
    rc.words = (const __attribute__((address_space(1))) u32 *)(a & ~3ull);
    rc.fshift = (u32)(a & 15ull);
    rc.shift = (u32)(a & 3ull);
    rc.last_chunk = len ? (i32)((rc.fshift + len - 1) >> 4) : -1;
    rc.last_word = len ? (i32)((rc.shift + len - 1) >> 2) : -1;
    rc.f_idx = 0;
    rc.f_cur = rc_lane_chunk(rc, 0);
    rc.f_next = rc_lane_chunk(rc, 1);
    rc.b_idx = rc.last_word;
    rc.b_cur = rc_lane_word(rc, rc.b_idx);
    rc.b_next = rc_lane_word(rc, rc.b_idx - 1);
}
// after the coder state (offs, end_offs, ...) was restored from elsewhere: re-position both windows
OG_DEV void rc_lane_resume(RcLane &rc) {
    rc.f_idx = (i32)((rc.offs + rc.fshift) >> 4);
    rc.f_cur = rc_lane_chunk(rc, rc.f_idx);
    rc.f_next = rc_lane_chunk(rc, rc.f_idx + 1);
    rc.b_idx = rc.end_offs < rc.storage ? (i32)((rc.storage - rc.end_offs - 1 + rc.shift) >> 2) : rc.last_word;
    rc.b_cur = rc_lane_word(rc, rc.b_idx);
    rc.b_next = rc_lane_word(rc, rc.b_idx - 1);
}
OG_DEV int rc_next_byte(RcLane &rc) {
    if (rc.offs >= rc.storage) return 0;
    const u32 pos = rc.offs++ + rc.fshift;
    const i32 w = (i32)(pos >> 4);
    if (w != rc.f_idx) { // sequential: w == f_idx + 1
        rc.f_cur = rc.f_next;
        rc.f_idx = w;
        rc.f_next = rc_lane_chunk(rc, w + 1);
    }
    // byte pos & 15 of the piece: the half it lies in, then one byte permute (selector 0x0c: a zero byte)
    const bool up = (pos & 8u) != 0;
    const u32 lo = up ? rc.f_cur.z : rc.f_cur.x, hi = up ? rc.f_cur.w : rc.f_cur.y;
    return (int)__builtin_amdgcn_perm(hi, lo, 0x0c0c0c00u | (pos & 7u));
}
OG_DEV int rc_next_byte_end(RcLane &rc) {
    if (rc.end_offs >= rc.storage) return 0;
    const u32 pos = rc.storage - ++rc.end_offs + rc.shift;
    const i32 w = (i32)(pos >> 2);
    if (w != rc.b_idx) { // sequential: w == b_idx - 1
        rc.b_cur = rc.b_next;
        rc.b_idx = w;
        rc.b_next = rc_lane_word(rc, w - 1);
    }
    return (int)((rc.b_cur >> (8 * (pos & 3))) & 255u);
}
#endif

template <class R>
OG_DEV void rc_renorm(R &rc) { // ec_dec_normalize :2649
    while (rc.rng <= (1u << 23)) {
        rc.nbits_total += 8;
        rc.rng <<= 8;
        int sym = rc.rem;
        rc.rem = rc_next_byte(rc);
        sym = (sym << 8 | rc.rem) >> 1; // EC_SYM_BITS - EC_CODE_EXTRA = 1
        rc.val = ((rc.val << 8) + (255u & ~(u32)sym)) & 0x7FFFFFFFu;
    }
}

template <class R>
OG_DEV void rc_init(R &rc, u32 len) { // ec_dec_init :2666
    rc.storage = len;
    rc.end_offs = 0;
    rc.end_window = 0;
    rc.nend_bits = 0;
    rc.nbits_total = 9;
    rc.offs = 0;
    rc.rng = 128;
    rc.rem = rc_next_byte(rc);
    rc.val = rc.rng - 1 - (rc.rem >> 1);
    rc.ext = 0;
    rc.error = 0;
    rc_renorm(rc);
}

template <class R>
OG_DEV i32 rc_tell(const R &rc) { return rc.nbits_total - ilog(rc.rng); } // celt.h:420

template <class R>
OG_DEV u32 rc_tell_frac(const R &rc) { // :2627
    u32 nbits = (u32)rc.nbits_total << 3;
    int l = ilog(rc.rng);
    u32 r = rc.rng >> (l - 16);
    u32 b = (r >> 12) - 8;
    // thresholds 2^(k/8+15.x): {35733, 38967, 42495, 46340, 50535, 55109, 60097, 65535}
    u32 corr = b == 0 ? 35733u : b == 1 ? 38967u : b == 2 ? 42495u : b == 3 ? 46340u
             : b == 4 ? 50535u : b == 5 ? 55109u : b == 6 ? 60097u : 65535u;
    b += r > corr;
    return nbits - (u32)((l << 3) + b);
}

template <class R>
OG_DEV u32 rc_decode(R &rc, u32 ft) { // ec_decode :2683
    rc.ext = rc.rng / ft;
    u32 s = rc.val / rc.ext;
    return ft - OG_MIN(s + 1, ft);
}

template <class R>
OG_DEV u32 rc_decode_bin(R &rc, unsigned bits) { // :2690
    rc.ext = rc.rng >> bits;
    u32 s = rc.val / rc.ext, top = 1u << bits;
    return top - OG_MIN(s + 1u, top);
}

template <class R>
OG_DEV void rc_update(R &rc, u32 fl, u32 fh, u32 ft) { // ec_dec_update :2697
    u32 s = rc.ext * (ft - fh);
    rc.val -= s;
    rc.rng = fl > 0 ? rc.ext * (fh - fl) : rc.rng - s;
    rc_renorm(rc);
}

template <class R>
OG_DEV int rc_bit_logp(R &rc, unsigned logp) { // :2712
    u32 r = rc.rng, d = rc.val, s = r >> logp;
    int ret = d < s;
    if (!ret) rc.val = d - s;
    rc.rng = ret ? s : r - s;
    rc_renorm(rc);
    return ret;
}

// inverse-CDF symbol (:2727).  `icdf` is a ROM table (global/constant memory, uniform address).
template <class R>
OG_DEV int rc_icdf(R &rc, const u8 *icdf, unsigned ftb) {
    u32 s = rc.rng, d = rc.val, r = s >> ftb, t;
    int ret = -1;
    do {
        t = s;
        s = r * icdf[++ret];
    } while (d < s);
    rc.val = d - s;
    rc.rng = t - s;
    rc_renorm(rc);
    return ret;
}

#ifndef OG_HOST_EMUL
// Wave-uniform decoder, SILK's workhorse: the reference's linear search is a chain of dependent table loads.  Here lane l
// evaluates entry l (tables have <= 42 entries and 64 bytes of padding behind them, tools/gen_rom_tables.py): the symbol
// is the first entry with val >= r * icdf[entry] -- one coalesced load, one ballot.  All 64 lanes must be active.
OG_DEV int rc_icdf(Rc &rc, const u8 *icdf, unsigned ftb) {
    const u32 d = rc.val, r = rc.rng >> ftb;
    const u32 s_l = r * (u32)icdf[OG_LANE];
    const u64 hit = __ballot(d >= s_l); // the terminating 0 entry always hits
    const int ret = (int)__builtin_ctzll(hit);
    const u32 s = (u32)__builtin_amdgcn_readlane((int)s_l, ret);
    const u32 t = ret ? (u32)__builtin_amdgcn_readlane((int)s_l, ret - 1) : rc.rng;
    rc.val = d - s;
    rc.rng = t - s;
    rc_renorm(rc);
    return ret;
}
#endif

template <class R>
OG_DEV u32 rc_bits(R &rc, unsigned bits) { // ec_dec_bits :2773
    u32 window = rc.end_window;
    int available = rc.nend_bits;
    if ((u32)available < bits) {
        do {
            window |= (u32)rc_next_byte_end(rc) << available;
            available += 8;
        } while (available <= 24);
    }
    u32 ret = window & ((1u << bits) - 1u);
    rc.end_window = window >> bits;
    rc.nend_bits = available - (int)bits;
    rc.nbits_total += bits;
    return ret;
}

template <class R>
OG_DEV u32 rc_uint(R &rc, u32 ft_in) { // ec_dec_uint :2747
    ft_in--;
    int ftb = ilog(ft_in);
    if (ftb > 8) {
        ftb -= 8;
        u32 ft = (ft_in >> ftb) + 1;
        u32 s = rc_decode(rc, ft);
        rc_update(rc, s, s + 1, ft);
        u32 t = s << ftb | rc_bits(rc, ftb);
        if (t <= ft_in) return t;
        rc.error = 1;
        return ft_in;
    }
    ft_in++;
    u32 s = rc_decode(rc, ft_in);
    rc_update(rc, s, s + 1, ft_in);
    return s;
}

template <class R>
OG_DEV int rc_laplace(R &rc, u32 fs, int decay) { // ec_laplace_decode :3047
    int val = 0;
    u32 fl = 0, fm = rc_decode_bin(rc, 15);
    if (fm >= fs) {
        val++;
        fl = fs;
        fs = ((32768u - 32u - fs) * (u32)(16384 - decay) >> 15) + 1; // ec_laplace_get_freq1 :3041
        while (fs > 1 && fm >= fl + 2 * fs) {
            fs *= 2;
            fl += fs;
            fs = ((fs - 2) * (u32)decay) >> 15;
            fs += 1;
            val++;
        }
        if (fs <= 1) {
            int di = (int)((fm - fl) >> 1);
            val += di;
            fl += 2 * di;
        }
        if (fm < fl + fs)
            val = -val;
        else
            fl += fs;
    }
    rc_update(rc, fl, OG_MIN(fl + fs, 32768u), 32768u);
    return val;
}

} // namespace og
