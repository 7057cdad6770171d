// og_celt_bands.hpp -- CELT band decoding for one frame per wavefront: PVQ shape decode,
// spreading rotation ("pulse de-spread"), folding / noise fill, time-frequency Haar and Hadamard
// reordering, stereo split and merge, anti-collapse.
//
// Vectors live in the LDS arena S.v and are addressed by index (X, the folding history `norm`, the
// pulse vector and a scratch row).  The bit-budget bookkeeping and every range-decoder read are
// wave-uniform scalar code; the per-coefficient loops are split over the 64 lanes.
//
// Structure (round-1 rework): everything here is force-inlined into ONE function so that the range
// decoder state and the band context stay in (scalar) registers -- the first version passed them by
// reference through non-inlined functions, which put them in scratch memory (~9 k flat accesses per
// frame).  To keep the code size finite without out-of-line calls, the reference's call tree is turned
// into loops with a single instance of each body:
//   * the two quant_band() calls of a stereo / dual-stereo band become a loop over "jobs";
//   * quant_partition()'s recursion (depth <= 4 splits) becomes an iterative walk with an explicit
//     stack of split frames in LDS;
//   * the PVQ codebook-size table U(n,k) sits in 48 VGPRs spread across the wave
//     (lane l of register (row, seg) holds U(row, 64*seg + l)); a lookup is a v_readlane, and the search
//     "largest k' <= k with U(n,k') <= i" along a row is three wave ballots instead of a dependent walk.
// Reference behaviour: src/celt.cpp:684-815 (rotation, residual normalisation, collapse mask,
// renormalise), :1010-1082 (anti-collapse), :1113-1213 (stereo merge, Hadamard, Haar),
// :1215-1355 (theta), :1357-1741 (band / partition decode), :1754-1924 (band loop), :2545-2620 (cwrsi).
#pragma once
#include "og_celt_math.hpp"

// Every barrier in the CELT path orders LDS traffic between the lanes of the single wave only.
#undef OG_SYNC
#define OG_SYNC() OG_LSYNC()

namespace og {

constexpr int BITRES = 3;

// The small ROM tables of the entropy-decoding half sit behind a provider type, so the same code can read them from
// global memory (single-kernel path) or from a per-workgroup LDS copy (parse kernel, where a table lookup on the
// critical path of 64 serial decoders should cost an LDS access, not a trip to L2).
struct RomGlobal {
    static OG_MEMBER i32 eband(int i) { return rom_eband[i]; }
    static OG_MEMBER i32 logn(int i) { return rom_logn[i]; }
    static OG_MEMBER i32 pulse_idx(int i) { return rom_pulse_idx[i]; }
    static OG_MEMBER i32 pulse_bits(int i) { return rom_pulse_bits[i]; }
    static OG_MEMBER i32 band_alloc(int i) { return rom_band_alloc[i]; }
    static OG_MEMBER i32 pulse_caps(int i) { return rom_pulse_caps[i]; }
    static OG_MEMBER i32 log2_frac(int i) { return rom_log2_frac[i]; }
    static OG_MEMBER i32 eprob(int i) { return rom_eprob[i]; }
};

// pulse cache of (band, LM): entry 0 = number of entries, entry q = bits (minus one) for q pulses (celt.h:537)
template <class T>
OG_DEV int pulse_cache(int band, int LM) { return T::pulse_idx((LM + 1) * NBANDS + band); }
template <class T>
OG_DEV int pulse_cache_max(int band, int LM) { // cache[cache[0]]
    const int base = pulse_cache<T>(band, LM);
    return T::pulse_bits(base + T::pulse_bits(base));
}
template <class T>
OG_DEV int bits2pulses(int band, int LM, int bits) { // celt.h:537
    const int base = pulse_cache<T>(band, LM);
    int lo = 0, hi = T::pulse_bits(base);
    bits--;
    for (int i = 0; i < 6; i++) {
        int mid = (lo + hi + 1) >> 1;
        if (T::pulse_bits(base + mid) >= bits) hi = mid; else lo = mid;
    }
    return bits - (lo == 0 ? -1 : T::pulse_bits(base + lo)) <= T::pulse_bits(base + hi) - bits ? lo : hi;
}
template <class T>
OG_DEV int pulses2bits(int band, int LM, int pulses) { return pulses == 0 ? 0 : T::pulse_bits(pulse_cache<T>(band, LM) + pulses) + 1; }
OG_DEV int get_pulses(int i) { return i < 8 ? i : (8 + (i & 7)) << ((i >> 3) - 1); } // celt.h:533

// ---- PVQ codebook-size table in registers -----------------------------------------------------------
#ifndef OG_HOST_EMUL
typedef u32 v16u __attribute__((ext_vector_type(16)));
#endif

struct PvqTab {
#ifndef OG_HOST_EMUL
    v16u t0, t1, t2; // row r (0..14) of U: columns [0,64) in t0[r], [64,128) in t1[r], [128,192) in t2[r]
#endif
    OG_MEMBER void load() {
#ifndef OG_HOST_EMUL
        const int l = OG_LANE;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            t0[r] = rom_pvq_u192[r * 192 + l];
            t1[r] = rom_pvq_u192[r * 192 + 64 + l];
            t2[r] = rom_pvq_u192[r * 192 + 128 + l];
        }
#endif
    }
    // U(a,b), symmetric; min(a,b) <= 14, max(a,b) <= 176
    OG_MEMBER u32 u(int a, int b) const {
        const int lo = a < b ? a : b, hi = a < b ? b : a;
#ifdef OG_HOST_EMUL
        return rom_pvq_u192[lo * 192 + hi];
#else
        const u32 v0 = t0[lo], v1 = t1[lo], v2 = t2[lo];
        const int seg = hi >> 6;
        const u32 v = seg == 0 ? v0 : (seg == 1 ? v1 : v2);
        return (u32)__builtin_amdgcn_readlane((int)v, hi & 63);
#endif
    }
    // largest k' in [0, k] with U(n, k') <= i (row n <= 14; U(n, .) is non-decreasing and U(n,0) == 0 for n > 0)
    OG_MEMBER int row_search(int n, int k, u32 i) const {
#ifdef OG_HOST_EMUL
        int kk = k;
        while (rom_pvq_u192[n * 192 + kk] > i) kk--;
        return kk;
#else
        const u32 v0 = t0[n], v1 = t1[n], v2 = t2[n];
        u64 m0 = __ballot(v0 <= i), m1 = __ballot(v1 <= i), m2 = __ballot(v2 <= i);
        // keep columns <= k only
        const int kc = k + 1; // number of admissible columns
        m0 &= kc >= 64 ? ~0ull : ((1ull << kc) - 1);
        m1 &= kc >= 128 ? ~0ull : (kc > 64 ? ((1ull << (kc - 64)) - 1) : 0ull);
        m2 &= kc >= 192 ? ~0ull : (kc > 128 ? ((1ull << (kc - 128)) - 1) : 0ull);
        if (m2) return 128 + 63 - __builtin_clzll(m2);
        if (m1) return 64 + 63 - __builtin_clzll(m1);
        return 63 - __builtin_clzll(m0 | 1ull);
#endif
    }
};

// Codeword index -> signed pulse vector in S.v[V_IY ...]; returns the sum of squares (cwrsi celt.cpp:2545).
OG_DEV i32 pvq_decode_index(const PvqTab &T, int n, int k, u32 i) {
    int pos = V_IY;
    i32 yy = 0;
    while (n > 2) {
        if (k >= n) { // "lots of pulses": everything needed is on row n (U is symmetric, n <= 14 here)
            const u32 p1 = T.u(n, k + 1);
            const int s = -(int)(i >= p1);
            i -= p1 & (u32)s;
            const int k0 = k;
            const u32 q = T.u(n, n);
            k = T.row_search(n, q > i ? n - 1 : k, i);
            i -= T.u(n, k);
            const int val = tr16((k0 - k + s) ^ s);
            S.v[pos++] = (i16)val;
            yy += val * val;
        } else { // "lots of dimensions": rows k and k+1, column n
            const u32 p = T.u(k, n), q = T.u(k + 1, n);
            if (p <= i && i < q) {
                i -= p;
                S.v[pos++] = 0;
            } else {
                const int s = -(int)(i >= q);
                i -= q & (u32)s;
                const int k0 = k;
                u32 pp;
                do pp = T.u(--k, n);
                while (pp > i);
                i -= pp;
                const int val = tr16((k0 - k + s) ^ s);
                S.v[pos++] = (i16)val;
                yy += val * val;
            }
        }
        n--;
    }
    {
        const u32 p = 2 * (u32)k + 1;
        int s = -(int)(i >= p);
        i -= p & (u32)s;
        const int k0 = k;
        k = (int)((i + 1) >> 1);
        if (k) i -= 2 * (u32)k - 1;
        int val = tr16((k0 - k + s) ^ s);
        S.v[pos++] = (i16)val;
        yy += val * val;
        s = -(int)i;
        val = tr16((k + s) ^ s);
        S.v[pos] = (i16)val;
        yy += val * val;
    }
    return yy;
}

// ---- lane-parallel vector helpers ----------------------------------------------------------------
// scale X[0..N) so that its norm becomes `gain` (renormalise_vector celt.cpp:797)
OG_DEV void renormalise(int x, int N, i32 gain) {
    OG_SYNC();
    i32 part = 0;
    OG_FOR_LANES(j, N) part += mul16(S.v[x + j], S.v[x + j]);
    i32 E = 1 + wave_sum(part);
    int k = ilog2(E) >> 1;
    i32 t = vshr32(E, 2 * (k - 7));
    i32 g = tr16(mul16_p15(rsqrt_norm(t), gain));
    OG_FOR_LANES(j, N) S.v[x + j] = (i16)pshr32(mul16(g, S.v[x + j]), k + 1);
    OG_SYNC();
}

// One pass of the 2-tap lattice along chains x[r], x[r+stride], ... (exp_rotation1 celt.cpp:684).
// Inside one block of `len` samples the chains r = 0..stride-1 are independent of each other, and the
// `nblk` blocks are independent too: one lane per (block, chain).  Along a chain the recurrence is serial;
// its running value is carried in a register (forward: the freshly rotated x[i+stride]; backward: x[i]),
// so each step has one LDS load that does not depend on the previous step.
OG_DEV void rotate_pass(int x, int len, int nblk, int stride, i32 c, i32 s) {
    OG_SYNC();
    const i32 ms = tr16(-s);
    const int nchain = nblk * stride;
    OG_FOR_LANES(id, nchain) {
        const int blk = id / stride, r = id - blk * stride;
        const int base = x + blk * len;
        int i = r;
        if (i < len - stride) { // forward: i = r, r+stride, ... while i < len - stride
            i32 x1 = S.v[base + i];
            for (; i < len - stride; i += stride) {
                const i32 x2 = S.v[base + i + stride];
                S.v[base + i] = (i16)pshr32(mul16(c, x1) + mul16(ms, x2), 15);
                x1 = tr16(pshr32(mul16(c, x2) + mul16(s, x1), 15));
            }
            S.v[base + i] = (i16)x1; // i is now the chain's last index
        }
        const int last = len - 2 * stride - 1; // backward: i = last .. 0 on this chain's residue
        if (last >= r) {
            const int top = last - (last - r) % stride;
            i32 x2 = S.v[base + top + stride];
            for (i = top; i >= 0; i -= stride) {
                const i32 x1 = S.v[base + i];
                S.v[base + i + stride] = (i16)pshr32(mul16(c, x2) + mul16(s, x1), 15);
                x2 = tr16(pshr32(mul16(c, x1) + mul16(ms, x2), 15));
            }
            S.v[base + r] = (i16)x2; // i + stride == r after the loop
        }
    }
    OG_SYNC();
}

// inverse spreading rotation (exp_rotation celt.cpp:707 with dir = -1)
OG_DEV void unspread(int x, int len, int stride, int K, int spread) {
    if (2 * K >= len || spread == 0) return;
    int factor = spread == 1 ? 15 : (spread == 2 ? 10 : 5);
    i32 gain = tr16(mul32_q31(mul16(32767, len), celt_rcp(len + factor * K))); // celt_div celt.h:367
    i32 theta = tr16(mul16_q15(gain, gain) >> 1);
    i32 c = cos_norm(theta);
    i32 s = cos_norm(sub16(32767, theta));
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    int blen = (int)udiv((u32)len, (u32)stride);
    if (stride2) rotate_pass(x, blen, stride, stride2, s, c);
    rotate_pass(x, blen, stride, 1, c, s);
}

OG_DEV void haar1(int x, int N0, int stride) { // celt.cpp:1202
    N0 >>= 1;
    OG_SYNC();
    OG_FOR_LANES(id, N0 * stride) {
        int j = id / stride, i = id - j * stride;
        int a = x + stride * 2 * j + i, b = x + stride * (2 * j + 1) + i;
        i32 t1 = mul16(23170, S.v[a]), t2 = mul16(23170, S.v[b]);
        S.v[a] = (i16)pshr32(t1 + t2, 15);
        S.v[b] = (i16)pshr32(t1 - t2, 15);
    }
    OG_SYNC();
}

// natural <-> "ordery" Hadamard index tables for stride 2, 4, 8, 16 (celt.cpp:1160)
OG_DEV int ordery(int stride, int i) {
    const u64 t2 = 0x01ULL, t4 = 0x1203ULL /* 3,0,2,1 */, t8 = 0x25163407ULL /* 7,0,4,3,6,1,5,2 */;
    const u64 t16 = 0x5A2D619E4B3C780FULL; /* 15,0,8,7,12,3,11,4,14,1,9,6,13,2,10,5 */
    u64 t = stride == 2 ? t2 : (stride == 4 ? t4 : (stride == 8 ? t8 : t16));
    return (int)((t >> (4 * i)) & 15);
}

// hadamard reorder, dir = 0: de-interleave (celt.cpp:1162), dir = 1: interleave (:1183)
OG_DEV void hadamard_reorder(int x, int N0, int stride, int hadamard, int dir) {
    int N = N0 * stride;
    OG_SYNC();
    OG_FOR_LANES(id, N) {
        int i = id / N0, j = id - i * N0;
        int blocked = (hadamard ? ordery(stride, i) : i) * N0 + j, inter = j * stride + i;
        if (dir == 0)
            S.v[V_TMP + blocked] = S.v[x + inter];
        else
            S.v[V_TMP + inter] = S.v[x + blocked];
    }
    OG_SYNC();
    OG_FOR_LANES(id, N) S.v[x + id] = S.v[V_TMP + id];
    OG_SYNC();
}

OG_DEV void stereo_merge(int x, int y, i32 mid, int N) { // celt.cpp:1113
    OG_SYNC();
    i32 pxp = 0, pside = 0;
    OG_FOR_LANES(j, N) {
        pxp += mul16(S.v[y + j], S.v[x + j]);
        pside += mul16(S.v[y + j], S.v[y + j]);
    }
    i32 xp = wave_sum(pxp), side = wave_sum(pside);
    xp = mul16x32_q15(mid, xp);
    i32 mid2 = tr16(mid >> 1);
    i32 El = mul16(mid2, mid2) + side - 2 * xp;
    i32 Er = mul16(mid2, mid2) + side + 2 * xp;
    if (Er < 161061 || El < 161061) { // QCONST32(6e-4f, 28)
        OG_FOR_LANES(j, N) S.v[y + j] = S.v[x + j];
        OG_SYNC();
        return;
    }
    int kl = ilog2(El) >> 1, kr = ilog2(Er) >> 1;
    i32 lgain = rsqrt_norm(vshr32(El, (kl - 7) << 1));
    i32 rgain = rsqrt_norm(vshr32(Er, (kr - 7) << 1));
    if (kl < 7) kl = 7;
    if (kr < 7) kr = 7;
    OG_FOR_LANES(j, N) {
        i32 l = tr16(mul16_p15(mid, S.v[x + j])), r = S.v[y + j];
        S.v[x + j] = (i16)pshr32(mul16(lgain, sub16(l, r)), kl + 1);
        S.v[y + j] = (i16)pshr32(mul16(rgain, add16(l, r)), kr + 1);
    }
    OG_SYNC();
}

// ---- split angle ----------------------------------------------------------------------------------
OG_DEV int compute_qn(int N, int b, int offset, int pulse_cap, int stereo) { // celt.cpp:1215
    int N2 = 2 * N - 1;
    if (stereo && N == 2) N2--;
    int qb = (b + N2 * offset) / N2;
    qb = OG_MIN(b - pulse_cap - (4 << BITRES), qb);
    qb = OG_MIN(8 << BITRES, qb);
    if (qb < (1 << BITRES >> 1)) return 1;
    int f = qb & 7; // exp2_table8
    int e = f == 0 ? 16384 : f == 1 ? 17866 : f == 2 ? 19483 : f == 3 ? 21247 : f == 4 ? 23170 : f == 5 ? 25267
          : f == 6 ? 27554 : 30048;
    int qn = e >> (14 - (qb >> BITRES));
    return (qn + 1) >> 1 << 1;
}

struct Split { int inv, imid, iside, delta, itheta, qalloc; };

// compute_theta celt.cpp:1241 (decoder branches)
template <class T, class R>
OG_DEV void compute_theta(R &rc, int band, int intensity, int disable_inv, i32 remaining_bits, Split &sc, int N, i32 &b, int B,
                          int B0, int LM, int stereo, i32 &fill) {
    int itheta = 0, inv = 0;
    int pulse_cap = T::logn(band) + LM * (1 << BITRES);
    int offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
    int qn = compute_qn(N, b, offset, pulse_cap, stereo);
    if (stereo && band >= intensity) qn = 1;
    u32 tell = rc_tell_frac(rc);
    if (qn != 1) {
        if (stereo && N > 2) {
            int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0, x;
            int fs = (int)rc_decode(rc, ft);
            if (fs < (x0 + 1) * p0) x = fs / p0; else x = x0 + 1 + (fs - (x0 + 1) * p0);
            rc_update(rc, x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0,
                      x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0, ft);
            itheta = x;
        } else if (B0 > 1 || stereo) {
            itheta = (int)rc_uint(rc, qn + 1);
        } else {
            int fs, fl, ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fm = (int)rc_decode(rc, ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
                itheta = (int)((isqrt32(8 * (u32)fm + 1) - 1) >> 1);
                fs = itheta + 1;
                fl = itheta * (itheta + 1) >> 1;
            } else {
                itheta = (int)((2u * (u32)(qn + 1) - isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1);
                fs = qn + 1 - itheta;
                fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            }
            rc_update(rc, fl, fl + fs, ft);
        }
        itheta = (int)udiv((u32)(itheta * 16384), (u32)qn);
    } else if (stereo) {
        if (b > 2 << BITRES && remaining_bits > 2 << BITRES) inv = rc_bit_logp(rc, 2);
        if (disable_inv) inv = 0;
    }
    int qalloc = (int)(rc_tell_frac(rc) - tell);
    b -= qalloc;
    int imid, iside, delta;
    if (itheta == 0) {
        imid = 32767;
        iside = 0;
        fill &= (1 << B) - 1;
        delta = -16384;
    } else if (itheta == 16384) {
        imid = 0;
        iside = 32767;
        fill &= ((1 << B) - 1) << B;
        delta = 16384;
    } else {
        imid = bitexact_cos(itheta);
        iside = bitexact_cos(16384 - itheta);
        delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
    }
    sc.inv = inv;
    sc.imid = imid;
    sc.iside = iside;
    sc.delta = delta;
    sc.itheta = itheta;
    sc.qalloc = qalloc;
}

// ---- partitions -----------------------------------------------------------------------------------
// Leaf of the partition tree: PVQ pulses, or (no pulses) zero / noise / folded-spectrum fill (celt.cpp:1463-1520,
// alg_unquant :782).  `low` < 0 means "no folding source".
OG_DEV u32 partition_leaf(Rc &rc, const PvqTab &T, int band, int spread, u32 &seed_io, i32 &remaining_bits, int x, int N, i32 b,
                          int B, int low, int LM, i32 gain, i32 fill) {
    int q = bits2pulses<RomGlobal>(band, LM, b), curr_bits = pulses2bits<RomGlobal>(band, LM, q);
    remaining_bits -= curr_bits;
    while (remaining_bits < 0 && q > 0) {
        remaining_bits += curr_bits;
        q--;
        curr_bits = pulses2bits<RomGlobal>(band, LM, q);
        remaining_bits -= curr_bits;
    }
    if (q != 0) {
        const int K = get_pulses(q);
        const i32 Ryy = pvq_decode_index(T, N, K, rc_uint(rc, T.u(N, K) + T.u(N, K + 1)));
        const int k = ilog2(Ryy) >> 1;
        const i32 t = vshr32(Ryy, 2 * (k - 7));
        const i32 g = tr16(mul16_p15(rsqrt_norm(t), gain));
        OG_SYNC();
        OG_FOR_LANES(j, N) S.v[x + j] = (i16)pshr32(mul16(g, S.v[V_IY + j]), k + 1); // normalise_residual :745
        unspread(x, N, B, K, spread);
        if (B <= 1) return 1;
        const u32 N0 = udiv((u32)N, (u32)B); // extract_collapse_mask :760
        u32 m = 0;
        OG_FOR_LANES(j, N) m |= (u32)(S.v[V_IY + j] != 0) << udiv((u32)j, N0);
        m = wave_or(m);
        OG_SYNC();
        return m;
    }
    const u32 cm_mask = (u32)((1ull << B) - 1);
    fill &= (i32)cm_mask;
    OG_SYNC();
    if (!fill) {
        OG_FOR_LANES(j, N) S.v[x + j] = 0;
        OG_SYNC();
        return 0;
    }
    const u32 seed = seed_io;
    u32 cm;
    if (low < 0) { // noise
        OG_FOR_LANES(j, N) S.v[x + j] = (i16)((i32)lcg_skip(seed, (u32)j + 1) >> 20);
        cm = cm_mask;
    } else { // folded spectrum, +-1/256 dither
        OG_FOR_LANES(j, N) {
            const u32 sj = lcg_skip(seed, (u32)j + 1);
            S.v[x + j] = (i16)(S.v[low + j] + ((sj & 0x8000) ? 4 : -4));
        }
        cm = (u32)fill;
    }
    seed_io = lcg_skip(seed, (u32)N);
    renormalise(x, N, gain);
    return cm;
}

// One split node of quant_partition (celt.cpp:1400-1462), kept in LDS while its children run.
struct SplitFrame {
    i32 x, N, B, B0, LM, low, low2, gain_mid, gain_side, fill, mbits, sbits, itheta, rebal, mid_first, stage, cm;
};
OG_LDS SplitFrame g_split[6];

// quant_partition celt.cpp:1382: the reference recurses (depth <= 4 splits); here the walk is iterative.
OG_DEV u32 partition_tree(Rc &rc, const PvqTab &T, int band, int spread, u32 &seed, i32 &remaining_bits, int x, int N, i32 b, int B,
                          int low, int LM, i32 gain, i32 fill) {
    int depth = 0;
    for (;;) {
        // ---- descend: split as long as the node asks for it
        for (;;) {
            if (!(LM != -1 && b > pulse_cache_max<RomGlobal>(band, LM) + 12 && N > 2)) break;
            const int B0 = B;
            Split sc;
            N >>= 1;
            LM -= 1;
            if (B == 1) fill = (fill & 1) | (fill << 1);
            B = (B + 1) >> 1;
            compute_theta<RomGlobal>(rc, band, 0, 0, remaining_bits, sc, N, b, B, B0, LM, 0, fill);
            i32 delta = sc.delta;
            const int itheta = sc.itheta;
            if (B0 > 1 && (itheta & 0x3fff)) {
                if (itheta > 8192)
                    delta -= delta >> (4 - LM);
                else
                    delta = OG_MIN(0, delta + (N << BITRES >> (5 - LM)));
            }
            const i32 mbits = OG_MAX(0, OG_MIN(b, (b - delta) / 2));
            const i32 sbits = b - mbits;
            remaining_bits -= sc.qalloc;
            SplitFrame &F = g_split[depth];
            F.x = x; F.N = N; F.B = B; F.B0 = B0; F.LM = LM; F.low = low; F.low2 = low >= 0 ? low + N : -1;
            F.gain_mid = tr16(mul16_p15(gain, sc.imid)); F.gain_side = tr16(mul16_p15(gain, sc.iside));
            F.fill = fill; F.mbits = mbits; F.sbits = sbits; F.itheta = itheta; F.rebal = remaining_bits;
            F.mid_first = mbits >= sbits; F.stage = 1; F.cm = 0;
            depth++;
            if (mbits >= sbits) { // first child: mid
                b = mbits;
                gain = tr16(mul16_p15(gain, sc.imid));
            } else {               // first child: side
                x = x + N;
                b = sbits;
                low = low >= 0 ? low + N : -1;
                gain = tr16(mul16_p15(gain, sc.iside));
                fill = fill >> B;
            }
        }
        // ---- leaf
        u32 cm = partition_leaf(rc, T, band, spread, seed, remaining_bits, x, N, b, B, low, LM, gain, fill);
        // ---- return to the parents
        for (;;) {
            if (depth == 0) return cm;
            SplitFrame &F = g_split[depth - 1];
            const int B0 = OG_UNI(F.B0), Bc = OG_UNI(F.B), stage = OG_UNI(F.stage), mid_first = OG_UNI(F.mid_first);
            if (stage == 1) { // first child done -> rebalance and run the second child
                const u32 c1 = mid_first ? cm : cm << (B0 >> 1);
                i32 mbits = OG_UNI(F.mbits), sbits = OG_UNI(F.sbits);
                const int itheta = OG_UNI(F.itheta);
                i32 rebalance = (mid_first ? mbits : sbits) - (OG_UNI(F.rebal) - remaining_bits);
                if (mid_first) {
                    if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
                } else {
                    if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
                }
                F.cm = (i32)c1;
                F.stage = 2;
                N = OG_UNI(F.N);
                B = Bc;
                LM = OG_UNI(F.LM);
                if (mid_first) { // second child: side
                    x = OG_UNI(F.x) + N;
                    b = sbits;
                    low = OG_UNI(F.low2);
                    gain = OG_UNI(F.gain_side);
                    fill = OG_UNI(F.fill) >> Bc;
                } else {          // second child: mid
                    x = OG_UNI(F.x);
                    b = mbits;
                    low = OG_UNI(F.low);
                    gain = OG_UNI(F.gain_mid);
                    fill = OG_UNI(F.fill);
                }
                break; // decode that child (it may split again: frames from `depth` upward are free)
            }
            // both children done
            cm = (u32)OG_UNI(F.cm) | (mid_first ? cm << (B0 >> 1) : cm);
            depth--;
        }
    }
}

// quant_band celt.cpp:1526 (mono band or one side of a stereo band); N > 1
OG_DEV u32 band_mono(Rc &rc, const PvqTab &T, int band, int spread, int tf_change, u32 &seed, i32 &remaining_bits, int x, int N, i32 b,
                     int B, int low, int LM, int low_out, i32 gain, int low_scratch, i32 fill) {
    int N0 = N, N_B, B0 = B, time_divide = 0, recombine = 0;
    const int longBlocks = B0 == 1;
    N_B = (int)udiv((u32)N, (u32)B);
    if (tf_change > 0) recombine = tf_change;
    if (low_scratch >= 0 && low >= 0 && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
        OG_SYNC();
        OG_FOR_LANES(j, N) S.v[low_scratch + j] = S.v[low + j];
        OG_SYNC();
        low = low_scratch;
    }
    for (int k = 0; k < recombine; k++) {
        if (low >= 0) haar1(low, N >> k, 1 << k);
        // bit_interleave_table celt.cpp:1560 applied to both nibbles: pairs of bits are OR-ed
        int lo = fill & 0xF, hi = fill >> 4;
        int tl = (lo & 3 ? 1 : 0) | (lo & 12 ? 2 : 0), th = (hi & 3 ? 1 : 0) | (hi & 12 ? 2 : 0);
        fill = tl | th << 2;
    }
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        if (low >= 0) haar1(low, N_B, B);
        fill |= fill << B;
        B <<= 1;
        N_B >>= 1;
        time_divide++;
        tf_change++;
    }
    B0 = B;
    const int N_B0 = N_B;
    if (B0 > 1 && low >= 0) hadamard_reorder(low, N_B >> recombine, B0 << recombine, longBlocks, 0);
    u32 cm = partition_tree(rc, T, band, spread, seed, remaining_bits, x, N, b, B, low, LM, gain, fill);
    if (B0 > 1) hadamard_reorder(x, N_B >> recombine, B0 << recombine, longBlocks, 1);
    N_B = N_B0;
    B = B0;
    for (int k = 0; k < time_divide; k++) {
        B >>= 1;
        N_B <<= 1;
        cm |= cm >> B;
        haar1(x, N_B, B);
    }
    for (int k = 0; k < recombine; k++) {
        // bit_deinterleave_table celt.cpp:1606: duplicate every bit of the low nibble
        u32 c4 = cm & 0xF;
        cm = ((c4 & 1) * 0x03) | ((c4 >> 1 & 1) * 0x0C) | ((c4 >> 2 & 1) * 0x30) | ((c4 >> 3 & 1) * 0xC0);
        haar1(x, N0 >> k, 1 << k);
    }
    B <<= recombine;
    if (low_out >= 0) {
        i32 n = tr16(celt_sqrt(shl32(N0, 22)));
        OG_SYNC();
        OG_FOR_LANES(j, N0) S.v[low_out + j] = (i16)mul16_q15(n, S.v[x + j]);
        OG_SYNC();
    }
    return cm & ((1u << B) - 1);
}

// quant_all_bands celt.cpp:1754 with quant_band_stereo :1628 and quant_band_n1 :1357 folded into the loop.
// N_ch = 120 << LM is the per-channel stride of X.
OG_DEV void decode_all_bands(Rc &rc, int start, int end, int C, int N_ch, int shortBlocks, int spread, int dual_stereo,
                             int intensity, i32 total_bits, i32 balance, int LM, int codedBands, u32 &seed_io, int disable_inv) {
    const int M = 1 << LM, B = shortBlocks ? M : 1;
    const int norm_offset = M * rom_eband[start];
    const int norm = V_NORM, norm2 = V_NORM + M * rom_eband[NBANDS - 1] - norm_offset;
    int low_scratch = V_X + M * rom_eband[NBANDS - 1];
    int lowband_offset = 0, update_lowband = 1;
    u32 seed = seed_io;
    PvqTab T;
    T.load();
    for (int i = start; i < end; i++) {
        const int last = i == end - 1;
        const int eb0 = M * rom_eband[i], N = M * rom_eband[i + 1] - eb0;
        const int x = V_X + eb0, y = C == 2 ? V_X + N_ch + eb0 : -1;
        i32 tell = (i32)rc_tell_frac(rc);
        if (i != start) balance -= tell;
        i32 remaining_bits = total_bits - tell - 1, b;
        const i32 pulses_i = OG_UNI(S.pulses_row()[i]);
        if (i <= codedBands - 1) {
            i32 curr_balance = balance / OG_MIN(3, codedBands - i);
            b = OG_MAX(0, OG_MIN(16383, OG_MIN(remaining_bits + 1, pulses_i + curr_balance)));
        } else
            b = 0;
        if ((eb0 - N >= M * rom_eband[start] || i == start + 1) && (update_lowband || lowband_offset == 0))
            lowband_offset = i;
        if (i == start + 1) { // special_hybrid_folding celt.cpp:1743
            int n1 = M * (rom_eband[start + 1] - rom_eband[start]), n2 = M * (rom_eband[start + 2] - rom_eband[start + 1]);
            if (n2 > n1) {
                OG_SYNC();
                OG_FOR_LANES(j, n2 - n1) {
                    S.v[norm + n1 + j] = S.v[norm + 2 * n1 - n2 + j];
                    if (dual_stereo) S.v[norm2 + n1 + j] = S.v[norm2 + 2 * n1 - n2 + j];
                }
                OG_SYNC();
            }
        }
        const int tf_change = OG_UNI(S.tf_res[i]);
        if (last) low_scratch = -1;
        int effective_lowband = -1;
        u32 x_cm, y_cm;
        if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
            effective_lowband = OG_MAX(0, M * rom_eband[lowband_offset] - norm_offset - N);
            int fold_start = lowband_offset;
            while (M * rom_eband[--fold_start] > effective_lowband + norm_offset) {}
            int fold_end = lowband_offset - 1;
            while (++fold_end < i && M * rom_eband[fold_end] < effective_lowband + norm_offset + N) {}
            x_cm = y_cm = 0;
            int fold_i = fold_start;
            do {
                x_cm |= (u32)OG_UNI(S.cmask_row()[fold_i * C + 0]);
                y_cm |= (u32)OG_UNI(S.cmask_row()[fold_i * C + C - 1]);
            } while (++fold_i < fold_end);
        } else
            x_cm = y_cm = (1u << B) - 1;
        if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            OG_SYNC();
            OG_FOR_LANES(j, eb0 - norm_offset) S.v[norm + j] = (i16)((S.v[norm + j] + S.v[norm2 + j]) >> 1);
            OG_SYNC();
        }
        const int low1 = effective_lowband != -1 ? norm + effective_lowband : -1;
        const int low2 = effective_lowband != -1 ? norm2 + effective_lowband : -1;
        const int out1 = last ? -1 : norm + eb0 - norm_offset;
        const int out2 = last ? -1 : norm2 + eb0 - norm_offset;

        if (N == 1) { // quant_band_n1 celt.cpp:1357 (mono, stereo and dual stereo all end up here)
            // dual stereo calls it once per channel with its own lowband_out; the bit consumption is identical
            for (int c = 0; c < (y >= 0 ? 2 : 1); c++) {
                int sign = 0;
                if (remaining_bits >= 1 << BITRES) {
                    sign = (int)rc_bits(rc, 1);
                    remaining_bits -= 1 << BITRES;
                }
                S.v[c ? y : x] = (i16)(sign ? -16384 : 16384);
            }
            if (out1 >= 0) S.v[out1] = (i16)(S.v[x] >> 4);
            if (dual_stereo && out2 >= 0) S.v[out2] = (i16)(S.v[y] >> 4);
            x_cm = y_cm = 1;
        } else {
            // ---- set up one or two "mono band" jobs
            const int stereo = (y >= 0) && !dual_stereo;
            Split sc;
            sc.inv = 0; sc.imid = 0; sc.iside = 0; sc.delta = 0; sc.itheta = 0; sc.qalloc = 0;
            i32 bb = b, fill0 = (i32)(x_cm | y_cm), orig_fill = fill0;
            i32 mbits = 0, sbits = 0, rebal0 = 0;
            int n2case = 0, swap_c = 0, sign = 1, mid_first = 1, njobs = 1;
            if (stereo) {
                compute_theta<RomGlobal>(rc, i, intensity, disable_inv, remaining_bits, sc, N, bb, B, B, LM, 1, fill0);
                if (N == 2) {
                    n2case = 1;
                    mbits = bb;
                    sbits = 0;
                    if (sc.itheta != 0 && sc.itheta != 16384) sbits = 1 << BITRES;
                    mbits -= sbits;
                    swap_c = sc.itheta > 8192;
                    remaining_bits -= sc.qalloc + sbits;
                    int sg = 0;
                    if (sbits) sg = (int)rc_bits(rc, 1);
                    sign = 1 - 2 * sg;
                } else {
                    mbits = OG_MAX(0, OG_MIN(bb, (bb - sc.delta) / 2));
                    sbits = bb - mbits;
                    remaining_bits -= sc.qalloc;
                    rebal0 = remaining_bits;
                    mid_first = mbits >= sbits;
                    njobs = 2;
                }
            } else if (dual_stereo)
                njobs = 2;
            u32 cm0 = 0, cm1 = 0;
            for (int jb = 0; jb < njobs; jb++) {
                int jx, jlow, jout, jscr;
                i32 jb_bits, jgain, jfill;
                if (dual_stereo) {
                    jx = jb ? y : x; jlow = jb ? low2 : low1; jout = jb ? out2 : out1; jscr = low_scratch;
                    jb_bits = b / 2; jgain = 32767; jfill = (i32)(jb ? y_cm : x_cm);
                } else if (!stereo) {
                    jx = x; jlow = low1; jout = out1; jscr = low_scratch; jb_bits = b; jgain = 32767; jfill = fill0;
                } else if (n2case) {
                    jx = swap_c ? y : x; jlow = low1; jout = out1; jscr = low_scratch; jb_bits = mbits; jgain = 32767; jfill = orig_fill;
                } else {
                    const int is_mid = (jb == 0) == (mid_first != 0);
                    if (jb == 1) { // rebalance between the two halves (celt.cpp:1711-1724)
                        i32 rebalance = (mid_first ? mbits : sbits) - (rebal0 - remaining_bits);
                        if (mid_first) {
                            if (rebalance > 3 << BITRES && sc.itheta != 0) sbits += rebalance - (3 << BITRES);
                        } else {
                            if (rebalance > 3 << BITRES && sc.itheta != 16384) mbits += rebalance - (3 << BITRES);
                        }
                    }
                    if (is_mid) {
                        jx = x; jlow = low1; jout = out1; jscr = low_scratch; jb_bits = mbits; jgain = 32767; jfill = fill0;
                    } else {
                        jx = y; jlow = -1; jout = -1; jscr = -1; jb_bits = sbits; jgain = sc.iside; jfill = fill0 >> B;
                    }
                }
                const u32 cmj = band_mono(rc, T, i, spread, tf_change, seed, remaining_bits, jx, N, jb_bits, B, jlow, LM, jout, jgain,
                                          jscr, jfill);
                if (jb == 0) cm0 = cmj; else cm1 = cmj;
            }
            if (stereo) {
                if (n2case) { // N == 2: the side is the mid rotated by 90 degrees (celt.cpp:1659-1697)
                    const int x2 = swap_c ? y : x;
                    OG_SYNC();
                    const i32 a0 = S.v[x2], a1 = S.v[x2 + 1];
                    const i32 b0 = tr16(-sign * a1), b1 = tr16(sign * a0);
                    i32 X0 = swap_c ? b0 : a0, X1 = swap_c ? b1 : a1, Y0 = swap_c ? a0 : b0, Y1 = swap_c ? a1 : b1;
                    X0 = tr16(mul16_q15(sc.imid, X0));
                    X1 = tr16(mul16_q15(sc.imid, X1));
                    Y0 = tr16(mul16_q15(sc.iside, Y0));
                    Y1 = tr16(mul16_q15(sc.iside, Y1));
                    OG_SYNC();
                    S.v[x] = (i16)sub16(X0, Y0);
                    S.v[y] = (i16)add16(X0, Y0);
                    S.v[x + 1] = (i16)sub16(X1, Y1);
                    S.v[y + 1] = (i16)add16(X1, Y1);
                    OG_SYNC();
                } else
                    stereo_merge(x, y, sc.imid, N);
                if (sc.inv) {
                    OG_SYNC();
                    OG_FOR_LANES(j, N) S.v[y + j] = (i16)(-S.v[y + j]);
                    OG_SYNC();
                }
                x_cm = y_cm = cm0 | cm1;
            } else if (dual_stereo) { // collapse masks are per channel
                x_cm = cm0;
                y_cm = cm1;
            } else
                x_cm = y_cm = cm0;
        }
        S.cmask_row()[i * C + 0] = (u8)x_cm;
        S.cmask_row()[i * C + C - 1] = (u8)y_cm;
        balance += pulses_i + tell;
        update_lowband = b > (N << BITRES);
    }
    seed_io = seed;
}

// anti_collapse celt.cpp:1010 (transient frames only).  The noise for a collapsed short block k of a
// band goes to X[(j<<LM)+k]; the LCG is stepped once per written sample, in (band, channel, k, j) order.
OG_DEV void anti_collapse(int LM, int C, int size, int start, int end, u32 seed) {
    for (int i = start; i < end; i++) {
        int N0 = rom_eband[i + 1] - rom_eband[i];
        int depth = (int)(udiv((u32)(1 + S.pulses_row()[i]), (u32)N0) >> LM);
        i32 thresh32 = celt_exp2(-shl16(depth, 10 - BITRES)) >> 1;
        i32 thresh = tr16(mul16x32_q15(16384, OG_MIN(32767, thresh32)));
        i32 t = N0 << LM;
        int shift = ilog2(t) >> 1;
        t = shl32(t, (7 - shift) << 1);
        i32 sqrt_1 = rsqrt_norm(t);
        for (int c = 0; c < C; c++) {
            i32 prev1 = S.logE1_row()[c * NBANDS + i], prev2 = S.logE2_row()[c * NBANDS + i];
            if (C == 1) {
                prev1 = OG_MAX(prev1, (i32)S.logE1_row()[NBANDS + i]);
                prev2 = OG_MAX(prev2, (i32)S.logE2_row()[NBANDS + i]);
            }
            i32 Ediff = (i32)S.bandE_row()[c * NBANDS + i] - OG_MIN(prev1, prev2);
            Ediff = OG_MAX(0, Ediff);
            i32 r;
            if (Ediff < 16384) {
                i32 r32 = celt_exp2(-tr16(Ediff)) >> 1;
                r = tr16(2 * OG_MIN(16383, r32));
            } else
                r = 0;
            if (LM == 3) r = tr16(mul16_q14(23170, OG_MIN(23169, r)));
            r = tr16(OG_MIN(thresh, r) >> 1);
            r = tr16(mul16_q15(sqrt_1, r) >> shift);
            int x = V_X + c * size + (rom_eband[i] << LM);
            int renorm = 0;
            u32 mask = S.cmask_row()[i * C + c];
            for (int k = 0; k < 1 << LM; k++) {
                if (!(mask & (1u << k))) {
                    OG_SYNC();
                    OG_FOR_LANES(j, N0) {
                        u32 sj = lcg_skip(seed, (u32)j + 1);
                        S.v[x + (j << LM) + k] = (i16)((sj & 0x8000) ? r : -r);
                    }
                    seed = lcg_skip(seed, (u32)N0);
                    renorm = 1;
                }
            }
            if (renorm) renormalise(x, N0 << LM, 32767);
        }
    }
}

} // namespace og
