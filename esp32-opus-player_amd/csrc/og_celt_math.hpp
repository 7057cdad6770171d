// og_celt_math.hpp -- scalar fixed-point approximations and the PVQ codeword enumeration used by
// the CELT wave decoder.  Each function reproduces, bit for bit, the result of the reference
// function named beside it (src/celt.cpp, src/celt.h).
#pragma once
#include "og_range.hpp"
#include "rom_tables.h"

namespace og {

OG_DEV u32 isqrt32(u32 val) { // celt.cpp:3086
    u32 g = 0;
    int bshift = (ilog(val) - 1) >> 1;
    u32 b = 1u << bshift;
    do {
        u32 t = ((g << 1) + b) << bshift;
        if (t <= val) {
            g += b;
            val -= t;
        }
        b >>= 1;
        bshift--;
    } while (bshift >= 0);
    return g;
}

OG_DEV i32 rsqrt_norm(i32 x) { // celt_rsqrt_norm celt.cpp:3109 (Q16 -> Q14)
    i32 n = tr16(x - 32768);
    i32 r = add16(23557, mul16_q15(n, add16(-13490, mul16_q15(n, 6713))));
    i32 r2 = tr16(mul16_q15(r, r));
    i32 y = shl16(sub16(add16(mul16_q15(r2, n), r2), 16384), 1);
    return add16(r, mul16_q15(r, mul16_q15(y, sub16(mul16_q15(y, 12288), 16384))));
}

OG_DEV i32 celt_sqrt(i32 x) { // celt.cpp:3131
    if (x == 0) return 0;
    if (x >= 1073741824) return 32767;
    int k = (ilog2(x) >> 1) - 7;
    x = vshr32(x, 2 * k);
    i32 n = tr16(x - 32768);
    i32 rt = add16(23175, mul16_q15(n, add16(11561, mul16_q15(n, add16(-3011, mul16_q15(n, add16(1699, mul16_q15(n, -664))))))));
    return vshr32(rt, 7 - k);
}

OG_DEV i32 cos_quarter(i32 x) { // _celt_cos_pi_2 celt.cpp:3151
    i32 x2 = tr16(mul16_p15(x, x));
    i32 v = sub16(32767, x2) + mul16_p15(x2, -7651 + mul16_p15(x2, 8277 + mul16_p15(-626, x2)));
    return add16(1, OG_MIN(32766, v));
}

OG_DEV i32 cos_norm(i32 x) { // celt_cos_norm celt.cpp:3161
    x &= 0x1ffff;
    if (x > (1 << 16)) x = (1 << 17) - x;
    if (x & 0x7fff) {
        if (x < (1 << 15)) return cos_quarter(tr16(x));
        return tr16(-cos_quarter(tr16(65536 - x)));
    }
    if (x & 0xffff) return 0;
    if (x & 0x1ffff) return -32767;
    return 32767;
}

OG_DEV i32 celt_rcp(i32 x) { // celt.cpp:3181 (Q15 -> Q16)
    int i = ilog2(x);
    i32 n = tr16(vshr32(x, i - 15) - 32768);
    i32 r = add16(30840, mul16_q15(-15420, n));
    r = tr16(sub16(r, mul16_q15(r, add16(mul16_q15(r, n), add16(r, -32768)))));
    r = tr16(sub16(r, add16(1, mul16_q15(r, add16(mul16_q15(r, n), add16(r, -32768))))));
    return vshr32(r, i - 16);
}

OG_DEV i32 exp2_frac(i32 x) { // celt_exp2_frac celt.h:494
    i32 frac = shl16(x, 4);
    return add16(16383, mul16_q15(frac, add16(22804, mul16_q15(frac, add16(14819, mul16_q15(10204, frac))))));
}

OG_DEV i32 celt_exp2(i32 x_in) { // celt.h:501 (Q10 -> Q16)
    i32 x = tr16(x_in);
    i32 integer = x >> 10;
    if (integer > 14) return 0x7f000000;
    if (integer < -15) return 0;
    i32 frac = tr16(exp2_frac(tr16(x - shl16(integer, 10))));
    return vshr32(frac, -integer - 2);
}

OG_DEV i32 bitexact_cos(i32 x) { // celt.cpp:926
    x = tr16(x);
    i32 x2 = tr16((4096 + x * x) >> 13);
    x2 = tr16((32767 - x2) + frac_mul16(x2, -7651 + frac_mul16(x2, 8277 + frac_mul16(-626, x2))));
    return tr16(1 + x2);
}

OG_DEV i32 bitexact_log2tan(i32 isin, i32 icos) { // celt.cpp:937
    int lc = ilog((u32)icos), ls = ilog((u32)isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls - lc) * (1 << 11) + frac_mul16(isin, frac_mul16(isin, -2597) + 7932) -
           frac_mul16(icos, frac_mul16(icos, -2597) + 7932);
}

// ---- linear congruential generator used for noise filling (celt_lcg_rand celt.cpp:921) ----------
OG_DEV u32 lcg_next(u32 s) { return 1664525u * s + 1013904223u; }
// n steps at once: compose the affine map with itself by squaring (exact modulo 2^32)
OG_DEV u32 lcg_skip(u32 s, u32 n) {
    u32 ra = 1u, rc = 0u, ba = 1664525u, bc = 1013904223u;
    while (n) {
        if (n & 1u) {
            ra = ba * ra;
            rc = ba * rc + bc;
        }
        bc = ba * bc + bc;
        ba = ba * ba;
        n >>= 1;
    }
    return ra * s + rc;
}

} // namespace og
