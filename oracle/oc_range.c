/*
 * oc_range.c -- CPU ORACLE (test infrastructure): the Opus range decoder.
 * Restates src/celt.cpp:2627-2792 and :3041-3083 of the reference with an explicit context.
 */
#include "oc_opus.h"

#define SYM_BITS 8u
#define CODE_BITS 32u
#define SYM_MAX 255u
#define CODE_TOP (1u << 31)
#define CODE_BOT (CODE_TOP >> SYM_BITS)
#define CODE_EXTRA 7u /* (32-2) % 8 + 1, celt.cpp:66 */

static int next_byte(oc_rc *rc) { return rc->offs < rc->storage ? rc->buf[rc->offs++] : 0; } /* :2642 */
static int next_byte_end(oc_rc *rc) {                                                     /* :2644 */
    return rc->end_offs < rc->storage ? rc->buf[rc->storage - ++rc->end_offs] : 0;
}

/* celt.cpp:2649 */
static void renorm(oc_rc *rc) {
    while (rc->rng <= CODE_BOT) {
        int sym;
        rc->nbits_total += SYM_BITS;
        rc->rng <<= SYM_BITS;
        sym = rc->rem;
        rc->rem = next_byte(rc);
        sym = (sym << SYM_BITS | rc->rem) >> (SYM_BITS - CODE_EXTRA);
        rc->val = ((rc->val << SYM_BITS) + (SYM_MAX & ~(u32)sym)) & (CODE_TOP - 1);
    }
}

/* celt.cpp:2666 */
void oc_rc_init(oc_rc *rc, const u8 *buf, u32 len) {
    rc->buf = buf;
    rc->storage = len;
    rc->end_offs = 0;
    rc->end_window = 0;
    rc->nend_bits = 0;
    rc->nbits_total = CODE_BITS + 1 - ((CODE_BITS - CODE_EXTRA) / SYM_BITS) * SYM_BITS; /* = 9 */
    rc->offs = 0;
    rc->rng = 1u << CODE_EXTRA;
    rc->rem = next_byte(rc);
    rc->val = rc->rng - 1 - (rc->rem >> (SYM_BITS - CODE_EXTRA));
    rc->ext = 0;
    rc->error = 0;
    renorm(rc);
}

/* celt.cpp:2683 */
u32 oc_rc_decode(oc_rc *rc, u32 ft) {
    u32 s;
    rc->ext = rc->rng / ft;
    s = rc->val / rc->ext;
    return ft - OC_MIN(s + 1, ft);
}

/* celt.cpp:2690 */
u32 oc_rc_decode_bin(oc_rc *rc, unsigned bits) {
    u32 s, top = 1u << bits;
    rc->ext = rc->rng >> bits;
    s = rc->val / rc->ext;
    return top - OC_MIN(s + 1u, top);
}

/* celt.cpp:2697 */
void oc_rc_update(oc_rc *rc, u32 fl, u32 fh, u32 ft) {
    u32 s = rc->ext * (ft - fh);
    rc->val -= s;
    rc->rng = fl > 0 ? rc->ext * (fh - fl) : rc->rng - s;
    renorm(rc);
}

/* celt.cpp:2712 */
int oc_rc_bit_logp(oc_rc *rc, unsigned logp) {
    u32 r = rc->rng, d = rc->val, s = r >> logp;
    int ret = d < s;
    if (!ret) rc->val = d - s;
    rc->rng = ret ? s : r - s;
    renorm(rc);
    return ret;
}

/* celt.cpp:2727 */
int oc_rc_icdf(oc_rc *rc, const u8 *icdf, unsigned ftb) {
    u32 s = rc->rng, d = rc->val, r = s >> ftb, t;
    int ret = -1;
    do {
        t = s;
        s = r * icdf[++ret];
    } while (d < s);
    rc->val = d - s;
    rc->rng = t - s;
    renorm(rc);
    return ret;
}

/* celt.cpp:2773 */
u32 oc_rc_bits(oc_rc *rc, unsigned bits) {
    u32 window = rc->end_window, ret;
    int available = rc->nend_bits;
    if ((u32)available < bits) {
        do {
            window |= (u32)next_byte_end(rc) << available;
            available += SYM_BITS;
        } while (available <= 32 - (int)SYM_BITS);
    }
    ret = window & ((1u << bits) - 1u);
    window >>= bits;
    available -= bits;
    rc->end_window = window;
    rc->nend_bits = available;
    rc->nbits_total += bits;
    return ret;
}

/* celt.cpp:2747 */
u32 oc_rc_uint(oc_rc *rc, u32 ft_in) {
    u32 ft, s, t;
    int ftb;
    ft_in--;
    ftb = ilog32(ft_in);
    if (ftb > 8) {
        ftb -= 8;
        ft = (ft_in >> ftb) + 1;
        s = oc_rc_decode(rc, ft);
        oc_rc_update(rc, s, s + 1, ft);
        t = s << ftb | oc_rc_bits(rc, ftb);
        if (t <= ft_in) return t;
        rc->error = 1;
        return ft_in;
    }
    ft_in++;
    s = oc_rc_decode(rc, ft_in);
    oc_rc_update(rc, s, s + 1, ft_in);
    return s;
}

/* celt.cpp:2627 */
u32 oc_rc_tell_frac(const oc_rc *rc) {
    static const u32 correction[8] = {35733, 38967, 42495, 46340, 50535, 55109, 60097, 65535};
    u32 nbits = (u32)rc->nbits_total << 3, r, b;
    int l = ilog32(rc->rng);
    r = rc->rng >> (l - 16);
    b = (r >> 12) - 8;
    b += r > correction[b];
    l = (l << 3) + b;
    return nbits - l;
}

/* celt.cpp:3041 */
static u32 laplace_freq1(u32 fs0, int decay) {
    u32 ft = 32768 - (2 * 16) - fs0; /* LAPLACE_MINP*(2*LAPLACE_NMIN) = 32 */
    return ft * (i32)(16384 - decay) >> 15;
}

/* celt.cpp:3047 */
int oc_rc_laplace(oc_rc *rc, u32 fs, int decay) {
    int val = 0;
    u32 fl = 0, fm = oc_rc_decode_bin(rc, 15);
    if (fm >= fs) {
        val++;
        fl = fs;
        fs = laplace_freq1(fs, decay) + 1;
        while (fs > 1 && fm >= fl + 2 * fs) {
            fs *= 2;
            fl += fs;
            fs = ((fs - 2) * (i32)decay) >> 15;
            fs += 1;
            val++;
        }
        if (fs <= 1) {
            int di = (fm - fl) >> 1;
            val += di;
            fl += 2 * di;
        }
        if (fm < fl + fs)
            val = -val;
        else
            fl += fs;
    }
    oc_rc_update(rc, fl, OC_MIN(fl + fs, 32768u), 32768u);
    return val;
}
