// og_celt_split.hpp -- the split CELT path: entropy decoding with ONE FRAME PER LANE, then vector reconstruction with
// one frame per wave, then de-emphasis with one (frame, channel) per lane.
//
// Why: everything the range decoder touches is a serial dependency chain over wave-uniform values.  Run one frame
// per wave it occupies a whole 64-lane SIMD for scalar work (measured on the single-kernel path: ~140 k vector +
// ~43 k scalar instructions per frame, half of them walking the PVQ codebook).  None of that work depends on the
// decoded spectrum: the bits a CELT frame reads are fully determined by the bit budget bookkeeping, never by the
// pulse vectors, collapse masks or the noise seed (src/celt.cpp:1382-1741: `fill`, `cm` and `seed` only steer the
// folding).  So the frame splits cleanly:
//
//   parse  (k_celt_parse, one frame per LANE, 64 frames per wave): header, energies, allocation, the band loop's
//          budget logic, split angles and PVQ codeword indices -- and every other wave-uniform quantity of the band
//          loop that does not depend on decoded data (folding offsets, gains, scale factors).  Output: a ParseRec
//          per frame in HBM: a header, a word stream in decode order (4 words per band, 1 per split / leaf) and an
//          array of PVQ leaves (index, position, N, K, blocks, gain).
//   recon  (k_celt_recon, one frame per WAVE): all PVQ leaves of the frame at once, one leaf per lane (index ->
//          pulses -> scaled, de-rotated coefficients + collapse mask: serial per leaf, independent across leaves);
//          then the band loop's vector half (folding / noise fill, Haar / Hadamard, stereo merge, collapse-mask
//          bookkeeping, anti-collapse) interpreting the word stream; then the synthesis half (og_celt.hpp).
//   post   (k_celt_post, one (frame, channel) per lane): the de-emphasis IIR (rounding => serial) and int16 PCM.
//
// All halves restate the same reference functions as og_celt_bands.hpp (file:line cited there); the single-kernel
// path remains for frames whose CELT part follows SILK data in the same range coder (hybrid) and for the SILK-only
// transition frame (Q4).
#pragma once
#include <stddef.h>
#include "og_celt.hpp"

#undef OG_SYNC
#define OG_SYNC() OG_LSYNC()

namespace og {

// ---- the record ------------------------------------------------------------------------------------
constexpr int REC_BAND_WORDS = 4;
constexpr int REC_MAX_LEAVES = NBANDS * 2 * 16;                   // <= 16 leaves per band and channel (4 split levels)
constexpr int REC_MAX_WORDS = NBANDS * (REC_BAND_WORDS + 2 * 33) + 3 * NBANDS + 1; // band words + per job: header + <= 16 leaves x 2 words (+ up to 3 words of padding before a band's header)

enum { // ParseRec.flags
    RF_SILENCE = 1, RF_TRANSIENT = 2, RF_LM_SHIFT = 2 /* 2 bits */, RF_STEREO = 16, RF_SPREAD_SHIFT = 5 /* 2 bits */,
    RF_DUAL = 128, RF_ANTI_COLLAPSE = 256, RF_RC_ERROR = 512, RF_TELL_OVERFLOW = 1024,
    RF_SKIP = 2048,    // descriptor rejected before any state change (decode_frame_wave's BAD_ARG), or (hybrid) the
                       // single-kernel path already reported the frame's error: nothing to do, result untouched
    RF_BAD_CELT = 4096 // celt_decode_frame's early CELT_BAD_ARG: bookkeeping only
};
// Band words.  W0: flags below; W1: eff_low | x << 11 | N << 22 (positions relative to their arena rows);
// W2: imid | iside << 16 (stereo split gains, Q15); W3: lowband_out scale sqrt(N) (Q?) in the low 16 bits.
enum {
    BW_SIGN0 = 1, BW_SIGN1 = 2, BW_INV = 4, BW_SIGN = 8, BW_MID_FIRST = 16, BW_SWAP = 32,
    BW_THETA0 = 64, BW_THETA1 = 128,   // the stereo angle is exactly 0 / exactly 16384 (fill mask halves)
    BW_TF_SHIFT = 8 /* tf_change + 4, 3 bits */, BW_FOLD0_SHIFT = 11 /* 5 bits */, BW_FOLD1_SHIFT = 16 /* 5 bits */,
    BW_HAS_LOW = 1 << 21, BW_DUAL = 1 << 22, BW_DUAL_END = 1 << 23, BW_STEREO = 1 << 24,
    BW_DUAL_PRE = 1 << 25 // dual stereo still on when the band starts (before a switch-off at the intensity band)
};
// Job words (a "job" = one band of one channel, or the mid / side part of a stereo band).  Header: number of leaves
// without pulses that follow | number of PVQ leaves << 5 | index of its first PVQ leaf << 10 | JW_NEED_LOW.  Then,
// per (non-silent) leaf without pulses, in decode order: L0 = off | B-1 | N (LW_* shifts), L1 = x | gain << 11.
enum {
    JW_NPVQ_SHIFT = 5 /* 5 bits */, JW_FIRST_SHIFT = 10 /* 10 bits */, JW_NEED_LOW = 1 << 20,
    LW_OFF_SHIFT = 8 /* 4 bits */, LW_B_SHIFT = 12 /* 4 bits */, LW_N_SHIFT = 16 /* 8 bits */
};

// Hybrid frames: the single-kernel path decodes the SILK half (wave per frame), then hands the live range decoder and
// the SILK PCM over to the split path, which decodes the CELT half (bands 17..20) and mixes the two in k_celt_post.
struct SilkHandoff {
    // SILK output at 48 kHz, interleaved over the packet's channels; first and on a line boundary of the memory system: the
    // synthesis kernel writes it and k_celt_post reads it in 128-byte pieces (at offset 48 of a 3,888-byte record every piece was two lines)
    alignas(128) i16 pcm[1920];
    u32 valid; // 1: SILK half decoded, coder state below is live
    u32 storage, end_offs, end_window;
    i32 nend_bits, nbits_total;
    u32 offs, rng, val, ext;
    i32 rem, error;
};
static_assert(sizeof(SilkHandoff) % 128 == 0 && offsetof(SilkHandoff, pcm) == 0, "handoff alignment");

struct ParseRec {
    i32 ret;       // samples per channel (960) -- or the negative code the frame ends with
    u32 rng_final; // range decoder's rng after the frame
    u32 flags;
    i32 intensity, pf_pitch, pf_gain, pf_tapset, start;
    i32 n_leaves, n_words;
    u32 need_norm; // bands whose folding history is read by a later band
    i32 n_coef;    // coefficients in PVQ leaves (the sum of their N)
    i32 reserved[4];
    i16 bandE[2 * NBANDS]; // final band energies (coarse + fine + finalise)
    i16 pulses[NBANDS];
    u16 band_w[NBANDS];    // where each band's four header words start in words[] (its job words follow them)
    i8 tf_res[NBANDS];
    i8 pad[256 - 64 - 4 * NBANDS - 2 * NBANDS - 2 * NBANDS - NBANDS];
    // A PVQ leaf is 16 bytes -- written by the parse lane with ONE store and fetched by the reconstruction's lane with one load
    // (round 3 kept three arrays of words: three scattered 4-byte stores per leaf from every lane, each into a record of its own).
    struct Leaf {
        u32 idx;  // PVQ codeword index
        u32 geom; // x | N << 11 | K << 19 | (B - 1) << 27   (x: offset into S.v[V_X..])
        u32 aux;  // gain (product of the split gains above the leaf, Q15) | mask offset << 16 (4 bits) | the leaf's job << 20
                  // (2 x band + decode slot: whose collapse mask the leaf's mask goes into, S.job_mask_row())
        u32 pad;
    } leaf[REC_MAX_LEAVES];
    u32 words[(REC_MAX_WORDS + 64) / 64 * 64]; // read in windows of 64 (REC_WORDS_CAP); a band's four header words start on a multiple of four
    // the parse lane's bits per band (32 bits: see ParseLds) from the end of its allocation on -- written once (LaneArr::pulses_rest,
    // 16-byte stores: the pad below is written too), read once per band by the band walk.  (Round 4 measured the array here for the
    // allocation's passes as well: 2.9 KB more HBM traffic per frame; those passes work in LDS, on the words of the two vectors.)
    i32 work_pulses[NBANDS];
    i32 work_pad[32 - NBANDS];
};
static_assert(offsetof(ParseRec, leaf) % 16 == 0 && offsetof(ParseRec, words) % 16 == 0 && offsetof(ParseRec, work_pulses) % 16 == 0 &&
              offsetof(ParseRec, work_pad) == offsetof(ParseRec, work_pulses) + 4 * NBANDS && NBANDS + 3 <= 32, "16-byte stores into the record");
static_assert(sizeof(ParseRec) % 16 == 0, "record alignment");

constexpr int FAST_MAX_LEAVES = 416; // (og_state.hpp: the most a 20 ms frame can have)

// =====================================================================================================
//  parse: one frame per lane
// =====================================================================================================
// Frames per parse wave (= lanes that carry a frame; the [element][lane] arrays below are that wide).  Fewer than the wave's
// 64 lanes means more, smaller waves: less LDS per wave (more of them resident per SIMD) and a shorter divergent union.
#ifndef OG_PL_LANES
#define OG_PL_LANES (OG_NLANES >= 32 ? 32 : OG_NLANES) // measured: 64 / 32 / 16 frames per wave, see DESIGN.md section 6
#endif
// Waves per parse workgroup (they share one copy of the ROM tables, ParseTabLds, 3.3 KB: LDS per wave goes from 13 granules of
// 1280 bytes to 11.5 at two, 10.75 at four).  Measured next to the reconstruction in pipelined steps (opusgpu_set_pipeline), where
// LDS is what the two kernels compete for: 2.74 ms per step at one, 2.73 at two, 2.80 at four (a workgroup's LDS stays
// allocated until its slowest wave is done) -- so one.
#ifndef OG_PL_WAVES
#define OG_PL_WAVES 1
#endif
// a thread's wave within the workgroup and the column of its frame in that wave's [element][column] arrays
#ifdef OG_HOST_EMUL
#define OG_PWAVE 0
#define OG_PCOL OG_LANE
#else
#define OG_PWAVE ((int)(threadIdx.x >> 6))
#define OG_PCOL ((int)(threadIdx.x & 63))
#endif
#define OG_PL_FRAMES (OG_PL_LANES * OG_PL_WAVES) // frames per workgroup
struct ParseLds { // [element][lane]: lanes of a wave touch consecutive addresses, no bank conflicts
    i8 fine_quant[NBANDS][OG_PL_LANES];
    i8 tf_prio[NBANDS][OG_PL_LANES]; // bits 0-3: tf_res (-3 .. 3, two's complement), bit 4: fine_prio
    // Three tenants, one after the other (next to the reconstruction the kernel's LDS is what keeps that kernel's waves out: 16.1 KB
    // per wave of 32 frames in round 2, 14.0 with the caps computed and tf_res / fine_prio in one byte, 11.3 with the energies resting, 8.6 with OG_PARSE_PULSES_REC):
    union {
        // the band energies while the header's energy stages and energy_finalise work on them (coarse energy .. , fine energy, the
        // finalise pass); in between they rest in the frame's record (LaneArr::energies_rest / energies_back: 21 words each way)
        i16 bandE[2 * NBANDS][OG_PL_LANES];
        struct { // from the dynalloc boosts until compute_allocation returns, i.e. before the first band is parsed
            i16 offsets[NBANDS][OG_PL_LANES]; // (the bands' caps are computed where they are used: celt_band_cap)
            // One word per band: the two allocation vectors' entries (bits1 | bits2 << 16) while compute_allocation interpolates
            // between them, then -- written over them band by band by the pass that settles the interpolation -- the band's bits
            // (32 bits: a frame whose budget went negative carries wrapped values here, as the reference does).  When the allocation
            // is done they move to the frame's record (LaneArr::pulses_rest), where the band walk reads one per band.  Round 5: the
            // bits had an array of their own here, a third of the kernel's LDS -- which is what keeps the reconstruction's waves
            // off the CUs the parse kernel runs on.
            u32 bw[NBANDS][OG_PL_LANES];
        } al;
        i32 stack[5][6][OG_PL_LANES]; // split frames of the partition walk: [depth][word][lane]
    } u;
#ifdef OG_PL_PAD /* occupancy experiments only */
    u8 pad_experiment[OG_PL_PAD];
#endif
};
// One per wave.  The union below is private to a wave only because its lanes reconverge between compute_allocation and the band
// walk; two waves of a workgroup do not, so they must not share rows of it.
// OG_PARSE_DYN_LDS (og_parse64.hip): the parse kernel's LDS as DYNAMIC shared memory, sized at the launch.  The compiler derives
// a kernel's occupancy -- and from it the register budget it allocates to -- from the LDS it can see, and with 46 KB per
// workgroup it saw two waves per SIMD and took 219 of their 256 registers, whatever the kernel was told to aim for; a SIMD that
// holds such a wave has registers left for three of the reconstruction's waves, not for five.
#ifdef OG_PARSE_DYN_LDS
extern __shared__ __attribute__((aligned(16))) unsigned char og_dyn_lds[];
#define PLs (reinterpret_cast<ParseLds *>(og_dyn_lds))
#else
OG_LDS ParseLds PLs[OG_PL_WAVES];
#endif
#define PL PLs[OG_PWAVE]

// LDS copy of the entropy-decoding ROM tables (see RomGlobal, og_celt_bands.hpp), loaded once per workgroup
struct ParseTabLds {
    i16 eband[NBANDS + 1], logn[NBANDS];
    u16 pulse_idx[105];
    u32 pulse_v[392]; // size of the PVQ codebook a leaf's index is decoded against, by pulse-cache index (rom_pulse_v)
    u8 pulse_bits[392], band_alloc[231], pulse_caps[168], log2_frac[24], eprob[336];
};
#ifdef OG_PARSE_DYN_LDS
#define PT (*reinterpret_cast<ParseTabLds *>(og_dyn_lds + sizeof(ParseLds) * OG_PL_WAVES))
#define OG_PARSE_LDS_BYTES (sizeof(ParseLds) * OG_PL_WAVES + sizeof(ParseTabLds))
#else
OG_LDS ParseTabLds PT;
#define OG_PARSE_LDS_BYTES 0
#endif
struct RomLds {
    static OG_MEMBER i32 eband(int i) { return PT.eband[i]; }
    static OG_MEMBER i32 logn(int i) { return PT.logn[i]; }
    static OG_MEMBER i32 pulse_idx(int i) { return PT.pulse_idx[i]; }
    static OG_MEMBER i32 pulse_bits(int i) { return PT.pulse_bits[i]; }
    static OG_MEMBER i32 band_alloc(int i) { return PT.band_alloc[i]; }
    static OG_MEMBER i32 pulse_caps(int i) { return PT.pulse_caps[i]; }
    static OG_MEMBER i32 log2_frac(int i) { return PT.log2_frac[i]; }
    static OG_MEMBER i32 eprob(int i) { return PT.eprob[i]; }
};
// cooperative load by the whole workgroup (call before any lane leaves the kernel); ends with a barrier
OG_DEV void parse_tables_load() {
    OG_FOR_LANES(i, NBANDS + 1) PT.eband[i] = rom_eband[i];
    OG_FOR_LANES(i, NBANDS) PT.logn[i] = rom_logn[i];
    OG_FOR_LANES(i, 105) PT.pulse_idx[i] = rom_pulse_idx[i];
    OG_FOR_LANES(i, 392) PT.pulse_bits[i] = rom_pulse_bits[i];
    OG_FOR_LANES(i, 392) PT.pulse_v[i] = rom_pulse_v[i];
    OG_FOR_LANES(i, 231) PT.band_alloc[i] = rom_band_alloc[i];
    OG_FOR_LANES(i, 168) PT.pulse_caps[i] = rom_pulse_caps[i];
    OG_FOR_LANES(i, 24) PT.log2_frac[i] = rom_log2_frac[i];
    OG_FOR_LANES(i, 336) PT.eprob[i] = rom_eprob[i];
    OG_FULL_SYNC();
}

// tf_res and fine_prio of a band share a byte of the lane's column: what the shared header code sees are these two views of it
struct TfResView {
    i8 *p;
    OG_MEMBER operator int() const { return (int)(i8)((u8)*p << 4) >> 4; }
    OG_MEMBER void operator=(int v) const { *p = (i8)((*p & 0xF0) | (v & 15)); }
};
struct FinePrioView {
    i8 *p;
    OG_MEMBER operator int() const { return (*p >> 4) & 1; }
    OG_MEMBER void operator=(int v) const { *p = (i8)((*p & ~0x10) | ((v & 1) << 4)); }
};
struct LaneArr {
    typedef RomLds Rom;
    i32 *pl;   // the bits-per-band array once compute_allocation is done: in the frame's record (ParseRec::work_pulses)
    i16 *rest; // where the band energies rest while the allocation scratch / the partition stack have their LDS: the record's bandE
    i16 *pk;   // the record's 16-bit copy of the bits per band (ParseRec::pulses: what the reconstruction's anti-collapse reads)
    // (pairs of bands per 32-bit access; every load is requested before the first is used)
    OG_MEMBER void energies_rest() const {
        for (int i = 0; i < 2 * NBANDS; i += 2)
            *reinterpret_cast<u32 *>(&rest[i]) = (u32)(u16)PL.u.bandE[i][OG_PCOL] | (u32)(u16)PL.u.bandE[i + 1][OG_PCOL] << 16;
        for (int i = 0; i < NBANDS; i++) PL.u.al.offsets[i][OG_PCOL] = 0;
    }
    OG_MEMBER void energies_back() const {
        u32 w[NBANDS];
        for (int i = 0; i < NBANDS; i++) w[i] = *reinterpret_cast<const u32 *>(&rest[2 * i]);
        for (int i = 0; i < NBANDS; i++) {
            PL.u.bandE[2 * i][OG_PCOL] = (i16)(w[i] & 0xffff);
            PL.u.bandE[2 * i + 1][OG_PCOL] = (i16)(w[i] >> 16);
        }
    }
    typedef u16 __attribute__((may_alias)) u16a;
    typedef i32 __attribute__((may_alias)) i32a;
    OG_MEMBER i32 &pulses(int i) const { return pl[i]; }
    OG_MEMBER i32a &alloc_bits(int i) const { return *reinterpret_cast<i32a *>(&PL.u.al.bw[i][OG_PCOL]); }
    // the allocation is done: bands start .. end - 1 to the record (zero outside), four words per store
    OG_MEMBER void pulses_rest(int start, int end) const {
        i32 v[24];
        for (int i = 0; i < 24; i++) v[i] = (i >= start && i < end) ? (i32)PL.u.al.bw[i < NBANDS ? i : 0][OG_PCOL] : 0;
#ifdef OG_HOST_EMUL
        for (int i = 0; i < NBANDS; i++) pl[i] = v[i];
#else
        typedef i32 i32x4 __attribute__((ext_vector_type(4)));
        for (int i = 0; i < 24; i += 4) *reinterpret_cast<i32x4 *>(&pl[i]) = i32x4{v[i], v[i + 1], v[i + 2], v[i + 3]};
#endif
        for (int i = 0; i < NBANDS; i++) pk[i] = (i16)v[i];
    }
    OG_MEMBER i8 &fine_quant(int i) const { return PL.fine_quant[i][OG_PCOL]; }
    OG_MEMBER FinePrioView fine_prio(int i) const { return FinePrioView{&PL.tf_prio[i][OG_PCOL]}; }
    OG_MEMBER TfResView tf_res(int i) const { return TfResView{&PL.tf_prio[i][OG_PCOL]}; }
    OG_MEMBER i16 &offsets(int i) const { return PL.u.al.offsets[i][OG_PCOL]; }
    OG_MEMBER u16a &bits1(int i) const { return reinterpret_cast<u16a *>(&PL.u.al.bw[i][OG_PCOL])[0]; }
    OG_MEMBER u16a &bits2(int i) const { return reinterpret_cast<u16a *>(&PL.u.al.bw[i][OG_PCOL])[1]; }
    OG_MEMBER i16 &bandE(int i) const { return PL.u.bandE[i][OG_PCOL]; }
};

OG_DEV u32 pvq_u_rom(int a, int b) { // U(a,b) from the ROM table (lane-private lookups)
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return rom_pvq_u[lo * ROM_PVQ_COLS + hi];
}

struct RecWriter {
    ParseRec *rec;
    int nw, nl, ncoef = 0;
    int job = 0; // the job whose leaves are being written (2 x band + decode slot)
    OG_MEMBER void word(u32 w) {
        if (nw < REC_MAX_WORDS) rec->words[nw] = w;
        nw++;
    }
    OG_MEMBER int reserve() { return nw++; } // a slot to be filled in later by patch()
    OG_MEMBER void patch(int at, u32 w) {
        if (at < REC_MAX_WORDS) rec->words[at] = w;
    }
    OG_MEMBER void leaf(int x, int N, int K, int B, i32 gain, int off, u32 idx) {
        if (nl < REC_MAX_LEAVES) {
            const u32 geom = (u32)x | (u32)N << 11 | (u32)K << 19 | (u32)(B - 1) << 27;
            const u32 aux = (u32)(gain & 0xffff) | (u32)off << 16 | (u32)job << 20;
#ifdef OG_HOST_EMUL
            rec->leaf[nl].idx = idx; rec->leaf[nl].geom = geom; rec->leaf[nl].aux = aux; rec->leaf[nl].pad = 0;
#else
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4 *>(&rec->leaf[nl]) = u32x4{idx, geom, aux, 0u};
#endif
        }
        nl++;
        ncoef += N;
    }
    // a band's four header words: on a multiple of four (up to three words skipped), one 16-byte store
    OG_MEMBER int band_begin() { return nw = (nw + 3) & ~3; }
    OG_MEMBER void words4(u32 w0, u32 w1, u32 w2, u32 w3) {
        if (nw + 4 <= REC_MAX_WORDS) {
#ifdef OG_HOST_EMUL
            rec->words[nw] = w0; rec->words[nw + 1] = w1; rec->words[nw + 2] = w2; rec->words[nw + 3] = w3;
#else
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4 *>(&rec->words[nw]) = u32x4{w0, w1, w2, w3};
#endif
        }
        nw += 4;
    }
};

// quant_partition celt.cpp:1382, range-decoder half: split decisions, angles, pulse counts and PVQ indices.  The
// partition tree itself does not reach the record: the reconstruction only needs its LEAVES in decode order, each with
// what the tree implies for it -- position, size, gain, and how the band's fill / collapse masks map onto the leaf:
//   fill(leaf) = silent ? 0 : (fill(job) >> off) & ((1 << B) - 1)        cm(job) |= cm(leaf) << off
// (`off` sums B0 >> 1 over the splits whose side branch leads to the leaf; a split with angle 0 silences its side
// branch, one with angle 16384 its mid branch: compute_theta's fill masks, celt.cpp:1320-1353.)
// In the word stream a job is one header word (JW_*: how many leaves without pulses follow, which PVQ leaves are its
// own, whether it needs its folding source at all) followed by two words per non-silent leaf without pulses; leaves
// with pulses only exist in the leaf arrays.  Returns 1 when the job needs the folding source.
// `silent`: the whole job's fill mask is known to be empty (the mid of a stereo band split at angle 16384, the side of one
// split at angle 0 -- every band from the intensity band on: celt.cpp:1320-1353 clear that half of the mask), so none of
// its leaves without pulses is ever filled and none is recorded.
OG_DEV int parse_tree(RcLane &rc, RecWriter &out, int band, i32 &remaining_bits, int x, int N, i32 b, int B, int LM, i32 gain,
                      int has_low, int silent) {
    int depth = 0, off = 0, n_fill = 0;
    const int jpos = out.reserve(), first_pvq = out.nl;
    for (;;) {
        OG_MARK(41);
        for (;;) { // descend
            if (!(LM != -1 && b > pulse_cache_max<RomLds>(band, LM) + 12 && N > 2)) break;
            const int B0 = B;
            Split sc;
            i32 fill = 0;
            N >>= 1;
            LM -= 1;
            B = (B + 1) >> 1;
            compute_theta<RomLds>(rc, band, 0, 0, remaining_bits, sc, N, b, B, B0, LM, 0, fill);
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
            const int mid_first = mbits >= sbits;
            const int off_side = off + (B0 >> 1), silent_mid = silent | (itheta == 16384), silent_side = silent | (itheta == 0);
            i32 *F = &PL.u.stack[depth][0][OG_PCOL];
            F[0 * OG_PL_LANES] = x | N << 11 | (LM + 1) << 19 | B << 22 | mid_first << 27 | 1 << 28;
            F[1 * OG_PL_LANES] = mbits;
            F[2 * OG_PL_LANES] = sbits;
            F[3 * OG_PL_LANES] = remaining_bits;
            // what the second child needs: its mask offset and whether it is silent
            F[4 * OG_PL_LANES] = itheta | (mid_first ? off_side : off) << 16 | (mid_first ? silent_side : silent_mid) << 24;
            const i32 gain_mid = tr16(mul16_p15(gain, sc.imid)), gain_side = tr16(mul16_p15(gain, sc.iside));
            F[5 * OG_PL_LANES] = (gain_mid & 0xffff) | gain_side << 16;
            depth++;
            if (mid_first) {
                b = mbits;
                gain = gain_mid;
                silent = silent_mid;
            } else {
                x += N;
                b = sbits;
                gain = gain_side;
                off = off_side;
                silent = silent_side;
            }
        }
        { // leaf: pulse count from the remaining budget, then the codeword index (celt.cpp:1463-1480)
            OG_MARK(42);
            int q = bits2pulses<RomLds>(band, LM, b), curr_bits = pulses2bits<RomLds>(band, LM, q);
            remaining_bits -= curr_bits;
            while (remaining_bits < 0 && q > 0) {
                remaining_bits += curr_bits;
                q--;
                curr_bits = pulses2bits<RomLds>(band, LM, q);
                remaining_bits -= curr_bits;
            }
            const int K = q ? get_pulses(q) : 0;
            OG_MARK(43);
            if (K) // V(N, K) = U(N, K) + U(N, K + 1) (celt.cpp:2622), found next to the cache entry that gave q
                out.leaf(x, N, K, B, gain, off, rc_uint(rc, PT.pulse_v[pulse_cache<RomLds>(band, LM) + q]));
            else if (!silent) { // (a silent leaf stays zero, as the spectrum was initialised: nothing to record)
                out.word((u32)off << LW_OFF_SHIFT | (u32)(B - 1) << LW_B_SHIFT | (u32)N << LW_N_SHIFT);
                out.word((u32)x | (u32)(gain & 0xffff) << 11);
                n_fill++;
            }
        }
        OG_MARK(44);
        for (;;) { // back to the parents
            if (depth == 0) {
                OG_MARK(40);
                const int need_low = has_low && n_fill > 0;
                out.patch(jpos, (u32)n_fill | (u32)(out.nl - first_pvq) << JW_NPVQ_SHIFT | (u32)first_pvq << JW_FIRST_SHIFT |
                                    (need_low ? JW_NEED_LOW : 0));
                return need_low;
            }
            i32 *F = &PL.u.stack[depth - 1][0][OG_PCOL];
            const i32 w0 = F[0];
            const int mid_first = (w0 >> 27) & 1, stage = (w0 >> 28) & 3;
            if (stage == 1) {
                i32 mbits = F[1 * OG_PL_LANES], sbits = F[2 * OG_PL_LANES];
                const int itheta = F[4 * OG_PL_LANES] & 0x7fff;
                const i32 rebalance = (mid_first ? mbits : sbits) - (F[3 * OG_PL_LANES] - remaining_bits);
                if (mid_first) {
                    if (rebalance > 3 << BITRES && itheta != 0) sbits += rebalance - (3 << BITRES);
                } else {
                    if (rebalance > 3 << BITRES && itheta != 16384) mbits += rebalance - (3 << BITRES);
                }
                F[0] = (w0 & ~(3 << 28)) | 2 << 28;
                N = (w0 >> 11) & 255;
                LM = ((w0 >> 19) & 7) - 1;
                B = (w0 >> 22) & 31;
                x = (w0 & 2047) + (mid_first ? N : 0);
                b = mid_first ? sbits : mbits;
                gain = mid_first ? F[5 * OG_PL_LANES] >> 16 : (i32)(i16)F[5 * OG_PL_LANES];
                off = (F[4 * OG_PL_LANES] >> 16) & 15;
                silent = (F[4 * OG_PL_LANES] >> 24) & 1;
                break;
            }
            depth--;
        }
    }
}

// quant_all_bands celt.cpp:1754: the range-decoder half, plus everything else about a band that is known without the
// decoded spectrum (folding source and mask range, stereo gains, the folding-history scale).
// Returns the set of bands (bit i = band i) whose folding history some later band actually reads.
OG_DEV u32 parse_all_bands(RcLane &rc, RecWriter &out, int start, int end, int C, int N_ch, int shortBlocks, int spread,
                           int dual_stereo, int intensity, i32 total_bits, i32 balance, int LM, int codedBands, int disable_inv) {
    const LaneArr a{out.rec->work_pulses, nullptr, nullptr}; // (pulses and tf_res only: the band energies rest in the record while the bands are parsed)
    const int M = 1 << LM, B = shortBlocks ? M : 1;
    const int norm_offset = M * RomLds::eband(start);
    int lowband_offset = 0, update_lowband = 1;
    u32 need_norm = 0;
    // the bands' bits from the record (LaneArr::pulses_rest), FOUR bands per 16-byte load, requested four bands ahead: one load per
    // band went to HBM every time -- the record's line does not survive in the L2 from one band to the next (21 read requests and
    // 2.7 KB of traffic per frame, round 5's counters)
#ifdef OG_HOST_EMUL
    i32 p4[4] = {0, 0, 0, 0}, n4[4] = {0, 0, 0, 0};
    auto fetch4 = [&](int b, i32 *o) { for (int k = 0; k < 4; k++) o[k] = b + k < NBANDS ? a.pulses(b + k) : 0; };
    fetch4(start & ~3, n4);
#else
    typedef i32 i32x4p __attribute__((ext_vector_type(4)));
    i32x4p p4 = {0, 0, 0, 0}, n4 = *reinterpret_cast<const i32x4p *>(&a.pulses(start & ~3)); // (work_pulses is padded to 32 words)
#endif
    for (int i = start; i < end; i++) {
        if (i == start || (i & 3) == 0) {
#ifdef OG_HOST_EMUL
            for (int k = 0; k < 4; k++) p4[k] = n4[k];
            fetch4((i & ~3) + 4, n4);
#else
            p4 = n4;
            n4 = *reinterpret_cast<const i32x4p *>(&a.pulses((i & ~3) + 4));
#endif
        }
        const i32 pulses_i = (i & 3) == 0 ? p4[0] : (i & 3) == 1 ? p4[1] : (i & 3) == 2 ? p4[2] : p4[3];
        const int eb0 = M * RomLds::eband(i), N = M * RomLds::eband(i + 1) - eb0;
        const int x = eb0, y = C == 2 ? N_ch + eb0 : -1;
        out.rec->band_w[i] = (u16)OG_MIN(out.band_begin(), REC_MAX_WORDS);
        const i32 tell = (i32)rc_tell_frac(rc);
        if (i != start) balance -= tell;
        i32 remaining_bits = total_bits - tell - 1, b;
        if (i <= codedBands - 1) {
            const i32 curr_balance = balance / OG_MIN(3, codedBands - i);
            b = OG_MAX(0, OG_MIN(16383, OG_MIN(remaining_bits + 1, pulses_i + curr_balance)));
        } else
            b = 0;
        const int tf_change = a.tf_res(i);
        // ---- folding source (celt.cpp:1812-1850): offsets into the folding history and the bands whose collapse
        //      masks feed this band's fill mask
        if ((eb0 - N >= M * RomLds::eband(start) || i == start + 1) && (update_lowband || lowband_offset == 0)) lowband_offset = i;
        u32 w0 = (u32)(tf_change + 4) << BW_TF_SHIFT, w1 = (u32)eb0 << 11 | (u32)N << 22;
        int has_low = 0;
        u32 fold_bands = 0; // the bands the folding source overlaps
        if (lowband_offset != 0 && (spread != 3 || B > 1 || tf_change < 0)) {
            const int effective_lowband = OG_MAX(0, M * RomLds::eband(lowband_offset) - norm_offset - N);
            int fold_start = lowband_offset;
            while (M * RomLds::eband(--fold_start) > effective_lowband + norm_offset) {}
            int fold_end = lowband_offset - 1;
            while (++fold_end < i && M * RomLds::eband(fold_end) < effective_lowband + norm_offset + N) {}
            w0 |= BW_HAS_LOW | (u32)fold_start << BW_FOLD0_SHIFT | (u32)fold_end << BW_FOLD1_SHIFT;
            w1 |= (u32)effective_lowband;
            has_low = 1;
            fold_bands = (1u << fold_end) - (1u << fold_start);
        }
        if (dual_stereo) w0 |= BW_DUAL_PRE;
        if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            w0 |= BW_DUAL_END;
        }
        if (dual_stereo) w0 |= BW_DUAL;
        u32 w2 = 0;
        if (N == 1) { // quant_band_n1 celt.cpp:1357
            for (int c = 0; c < (y >= 0 ? 2 : 1); c++) {
                if (remaining_bits >= 1 << BITRES) {
                    if (rc_bits(rc, 1)) w0 |= c ? BW_SIGN1 : BW_SIGN0;
                    remaining_bits -= 1 << BITRES;
                }
            }
            out.words4(w0, w1, 0, 0);
        } else {
            const int stereo = (y >= 0) && !dual_stereo;
            Split sc;
            sc.inv = 0; sc.imid = 0; sc.iside = 0; sc.delta = 0; sc.itheta = 0; sc.qalloc = 0;
            i32 bb = b, fill_unused = 0, mbits = 0, sbits = 0, rebal0 = 0;
            int n2case = 0, swap_c = 0, mid_first = 1, njobs = 1;
            if (stereo) { // quant_band_stereo celt.cpp:1628
                compute_theta<RomLds>(rc, i, intensity, disable_inv, remaining_bits, sc, N, bb, B, B, LM, 1, fill_unused);
                w0 |= BW_STEREO;
                if (sc.itheta == 0) w0 |= BW_THETA0;
                if (sc.itheta == 16384) w0 |= BW_THETA1;
                if (sc.itheta > 8192) w0 |= BW_SWAP;
                if (sc.inv) w0 |= BW_INV;
                w2 = (u32)(sc.imid & 0xffff) | (u32)sc.iside << 16;
                if (N == 2) {
                    n2case = 1;
                    mbits = bb;
                    sbits = 0;
                    if (sc.itheta != 0 && sc.itheta != 16384) sbits = 1 << BITRES;
                    mbits -= sbits;
                    swap_c = sc.itheta > 8192;
                    remaining_bits -= sc.qalloc + sbits;
                    if (sbits && rc_bits(rc, 1)) w0 |= BW_SIGN;
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
            if (mid_first) w0 |= BW_MID_FIRST;
            out.words4(w0, w1, w2, (u32)(u16)tr16(celt_sqrt(shl32(N, 22)))); // (w3: scale of the folding history, celt.cpp:1617)
            for (int jb = 0; jb < njobs; jb++) {
                int jx, jlow = has_low, jsilent = 0;
                i32 jbits, jgain = 32767;
                if (dual_stereo) {
                    jx = jb ? y : x;
                    jbits = b / 2;
                } else if (!stereo) {
                    jx = x;
                    jbits = b;
                } else if (n2case) {
                    jx = swap_c ? y : x;
                    jbits = mbits;
                } else {
                    const int is_mid = (jb == 0) == (mid_first != 0);
                    if (jb == 1) { // rebalance between the two halves (celt.cpp:1711-1724)
                        const i32 rebalance = (mid_first ? mbits : sbits) - (rebal0 - remaining_bits);
                        if (mid_first) {
                            if (rebalance > 3 << BITRES && sc.itheta != 0) sbits += rebalance - (3 << BITRES);
                        } else {
                            if (rebalance > 3 << BITRES && sc.itheta != 16384) mbits += rebalance - (3 << BITRES);
                        }
                    }
                    jx = is_mid ? x : y;
                    jbits = is_mid ? mbits : sbits;
                    jsilent = is_mid ? sc.itheta == 16384 : sc.itheta == 0;
                    if (!is_mid) {
                        jgain = sc.iside;
                        jlow = 0; // the side never folds (celt.cpp:1709)
                    }
                }
                // quant_band celt.cpp:1526: only the block count reaches the partition walk's decisions
                int Bj = B, N_B = (int)udiv((u32)N, (u32)B), tfc = tf_change;
                const int recombine = tfc > 0 ? tfc : 0;
                Bj >>= recombine;
                N_B <<= recombine;
                while ((N_B & 1) == 0 && tfc < 0) {
                    Bj <<= 1;
                    N_B >>= 1;
                    tfc++;
                }
                out.job = 2 * i + jb;
                if (parse_tree(rc, out, i, remaining_bits, jx, N, jbits, Bj, LM, jgain, jlow, jsilent)) need_norm |= fold_bands;
            }
        }
        balance += pulses_i + tell;
        update_lowband = b > (N << BITRES);
    }
    return need_norm;
}

// One CELT-only frame, lane-private.  `payload`/`len`: the frame's bytes; `ch`: channels coded in the packet,
// CC: decoder channels.  Mirrors decode_frame_wave + celt_decode_frame up to (not including) every vector operation.
// `handoff` (hybrid frames): resume the range decoder where the SILK half left it and start at band 17.
// The stream's band energies (CeltState::bandE) are carried from frame to frame HERE, not by the reconstruction: they are the only
// stream state this half reads, so the parse of a stream's next frame depends on nothing but the parse of this one and may run
// while this frame is still being reconstructed (opusgpu_set_pipeline, og_api.hip).
OG_DEV void celt_parse_lane(StreamState *st, const u8 *payload, int len, int ch, ParseRec *rec, const SilkHandoff *handoff) {
    const LaneArr a{rec->work_pulses, rec->bandE, rec->pulses};
    const int CC = st->channels, C = ch, LM = 3, frame_size = 960, start = handoff ? 17 : 0, end = NBANDS;
    rec->start = start;
    rec->n_leaves = 0;
    rec->n_words = 0;
    if (len < 0 || len > 1275 || (handoff && !handoff->valid)) {
        rec->ret = BAD_ARG;
        rec->flags = RF_SKIP;
        return;
    }
    RcLane rc;
    rc_lane_attach(rc, payload, (u32)len);
    if (handoff) {
        rc.storage = handoff->storage; rc.end_offs = handoff->end_offs; rc.end_window = handoff->end_window;
        rc.nend_bits = handoff->nend_bits; rc.nbits_total = handoff->nbits_total; rc.offs = handoff->offs; rc.rng = handoff->rng;
        rc.val = handoff->val; rc.ext = handoff->ext; rc.rem = handoff->rem; rc.error = handoff->error;
        rc_lane_resume(rc);
    } else
        rc_init(rc, (u32)len);
    if (rc.storage <= 1) { // celt_decode_frame's early exit (celt.cpp:2225)
        rec->ret = CELT_BAD_ARG;
        rec->flags = RF_BAD_CELT;
        rec->rng_final = rc.rng;
        return;
    }
    const int disable_inv = CC == 1;
    for (int i = 0; i < 2 * NBANDS; i++) a.bandE(i) = st->celt.bandE[i];
    if (C == 1)
        for (int i = 0; i < NBANDS; i++) a.bandE(i) = (i16)OG_MAX((i32)a.bandE(i), (i32)a.bandE(NBANDS + i));
    for (int i = 0; i < NBANDS; i++) { // (the dynalloc offsets are cleared where the energies make room for them: energies_rest;
                                       // the bits per band outside start .. end where they leave the allocation scratch: pulses_rest)
        a.fine_quant(i) = 0;
        a.fine_prio(i) = 0;
    }
    CeltHeader h;
    OG_MARK(20);
    celt_parse_header(a, rc, start, end, C, LM, h);
    a.energies_rest(); // the partition walk's stack takes their place
    RecWriter out;
    out.rec = rec;
    out.nw = 0;
    out.nl = 0;
    const int M = 1 << LM, N = M * 120;
    // tf_res and pulses are needed by the reconstruction (pulses_rest wrote those; they change meaning nowhere after the header)
    for (int i = 0; i < NBANDS; i++) rec->tf_res[i] = a.tf_res(i);
    OG_MARK(26);
    rec->need_norm = parse_all_bands(rc, out, start, end, C, N, h.transient ? M : 0, h.spread, h.dual_stereo, h.intensity,
                                     (i32)rc.storage * (8 << BITRES) - h.anti_collapse_rsv, h.balance, LM, h.codedBands, disable_inv);
    OG_MARK(27);
    int anti_collapse_on = 0;
    if (h.anti_collapse_rsv > 0) anti_collapse_on = (int)rc_bits(rc, 1);
    a.energies_back();
    energy_finalise(a, rc, start, end, (i32)rc.storage * 8 - rc_tell(rc), C);
    for (int i = 0; i < 2 * NBANDS; i++) rec->bandE[i] = a.bandE(i);
    u32 flags = (u32)LM << RF_LM_SHIFT | (u32)h.spread << RF_SPREAD_SHIFT;
    if (h.silence) flags |= RF_SILENCE;
    if (h.transient) flags |= RF_TRANSIENT;
    if (C == 2) flags |= RF_STEREO;
    if (h.dual_stereo) flags |= RF_DUAL;
    if (anti_collapse_on) flags |= RF_ANTI_COLLAPSE;
    if (rc.error || out.nw > REC_MAX_WORDS || out.nl > REC_MAX_LEAVES) flags |= RF_RC_ERROR;
    if (rc_tell(rc) > 8 * (i32)rc.storage) flags |= RF_TELL_OVERFLOW;
    rec->flags = flags;
    rec->ret = frame_size;
    rec->rng_final = rc.rng;
    rec->intensity = h.intensity;
    rec->pf_pitch = h.pf_pitch;
    rec->pf_gain = h.pf_gain;
    rec->pf_tapset = h.pf_tapset;
    rec->n_leaves = OG_MIN(out.nl, REC_MAX_LEAVES);
    rec->n_coef = out.ncoef;
    rec->n_words = OG_MIN(out.nw, REC_MAX_WORDS);
    // the energies the next frame predicts from, as celt_synthesis leaves them (celt.cpp:2404-2436): -28 dB in a silent frame,
    // a mono frame's in both channels, zero outside start .. end.  Two bands per store.
    for (int i = 0; i < 2 * NBANDS; i += 2) {
        i32 e[2];
        for (int k = 0; k < 2; k++) {
            const int band = i + k >= NBANDS ? i + k - NBANDS : i + k;
            e[k] = h.silence ? -28 * 1024 : (i32)a.bandE(C == 1 ? band : i + k);
            if (band < start || band >= end) e[k] = 0;
        }
        *reinterpret_cast<u32 *>(&st->celt.bandE[i]) = (u32)(u16)e[0] | (u32)(u16)e[1] << 16;
    }
}

// =====================================================================================================
//  recon: one frame per wave
// =====================================================================================================
// One PVQ leaf, lane-private (alg_unquant celt.cpp:782): codeword index -> signed pulse vector (cwrsi :2545),
// scaled to the leaf's gain (normalise_residual :745), spreading rotation undone (exp_rotation :707, dir = -1),
// collapse mask (extract_collapse_mask :760).  Everything is a serial chain per leaf, so the frame's leaves run
// one per lane; the result is written in place at S.v[pos .. pos+n).  Returns the collapse mask.
#ifdef OG_HOST_EMUL
OG_DEV void rotate1_lane(i16 *xv, int x, int len, int stride, i32 c, i32 s) { // exp_rotation1 celt.cpp:684
    // The reference sweeps i = 0 .. len-stride-1 forward, then len-2*stride-1 .. 0 backward, over pairs (i, i+stride).
    // Pairs with different i mod stride never touch the same element, so each residue class ("chain") can be walked
    // on its own, carrying the element both consecutive steps share in a register: one LDS read and one write per step.
    const i32 ms = tr16(-s);
    for (int r = 0; r < stride; r++) {
        if (r < len - stride) { // forward along the chain r, r+stride, ...
            int i = r;
            i32 x1 = xv[x + i];
            for (; i < len - stride; i += stride) {
                const i32 x2 = xv[x + i + stride];
                xv[x + i] = (i16)pshr32(mul16(c, x1) + mul16(ms, x2), 15);
                x1 = tr16(pshr32(mul16(c, x2) + mul16(s, x1), 15));
            }
            xv[x + i] = (i16)x1;
        }
        const int last = len - 2 * stride - 1;
        if (last >= r) { // backward: from the chain's highest start index <= last down to r
            int i = last - (last - r) % stride;
            i32 x2 = xv[x + i + stride];
            for (; i >= 0; i -= stride) {
                const i32 x1 = xv[x + i];
                xv[x + i + stride] = (i16)pshr32(mul16(c, x2) + mul16(s, x1), 15);
                x2 = tr16(pshr32(mul16(c, x1) + mul16(ms, x2), 15));
            }
            xv[x + r] = (i16)x2;
        }
    }
}
#else
// The same chain walk for the GPU, where a wave runs it for 64 leaves in lock-step and pays for the longest: (a) a step's two
// outputs are each one v_dot2_i32_i16 -- c x1 + s x2 + 16384 with the pair (x1, x2) packed in one register -- and a shift;
// (b) four steps at a time, their four new elements requested together before the first result is stored (a store to the
// spectrum keeps the compiler from moving the next element's read above it, so step by step every element costs an LDS round
// trip); (c) the backward sweep starts where the forward one counted to (no remainder to divide out).
OG_DEV i32 rot_dot2(u32 pair, u32 coef, i32 half) {
    i32 r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(pair), "v"(coef), "v"(half));
    return r >> 15;
}
OG_DEV u32 rot_pack(i32 lo, i32 hi) { return __builtin_amdgcn_perm((u32)hi, (u32)lo, 0x05040100u); } // low halves of both
// one chain r, r + stride, r + 2 stride, .. of exp_rotation1 (celt.cpp:684): forward, then backward
OG_DEV void rotate_chain(i16 *const p0, int r, int len, int stride, u32 k_a, u32 k_b) {
    const i32 half = 16384;
    // forward along the chain: pairs (i, i + stride) while i < len - stride
    int i = r, steps = 0;
    i32 x1 = p0[i];
    for (; i + 3 * stride < len - stride; i += 4 * stride, steps += 4) {
        const i32 e1 = p0[i + stride], e2 = p0[i + 2 * stride], e3 = p0[i + 3 * stride], e4 = p0[i + 4 * stride];
        u32 pk = rot_pack(x1, e1);
        const i32 o0 = rot_dot2(pk, k_a, half);
        x1 = rot_dot2(pk, k_b, half);
        pk = rot_pack(x1, e2);
        const i32 o1 = rot_dot2(pk, k_a, half);
        x1 = rot_dot2(pk, k_b, half);
        pk = rot_pack(x1, e3);
        const i32 o2 = rot_dot2(pk, k_a, half);
        x1 = rot_dot2(pk, k_b, half);
        pk = rot_pack(x1, e4);
        const i32 o3 = rot_dot2(pk, k_a, half);
        x1 = rot_dot2(pk, k_b, half);
        p0[i] = (i16)o0;
        p0[i + stride] = (i16)o1;
        p0[i + 2 * stride] = (i16)o2;
        p0[i + 3 * stride] = (i16)o3;
    }
    for (; i < len - stride; i += stride, steps++) {
        const u32 pk = rot_pack(x1, p0[i + stride]);
        p0[i] = (i16)rot_dot2(pk, k_a, half);
        x1 = rot_dot2(pk, k_b, half);
    }
    p0[i] = (i16)x1;
    // backward: pairs (i, i + stride) from the chain's highest i <= len - 2 stride - 1 down to r -- one pair fewer than forward
    if (steps >= 2) {
        i -= 2 * stride; // (forward ended on the chain's last element, r + steps * stride)
        i32 x2 = p0[i + stride];
        for (; i - 3 * stride >= 0; i -= 4 * stride) {
            const i32 e1 = p0[i], e2 = p0[i - stride], e3 = p0[i - 2 * stride], e4 = p0[i - 3 * stride];
            u32 pk = rot_pack(e1, x2); // (x1, x2) = (element i, carried): second output goes to i + stride, first is carried down
            const i32 o0 = rot_dot2(pk, k_b, half);
            x2 = rot_dot2(pk, k_a, half);
            pk = rot_pack(e2, x2);
            const i32 o1 = rot_dot2(pk, k_b, half);
            x2 = rot_dot2(pk, k_a, half);
            pk = rot_pack(e3, x2);
            const i32 o2 = rot_dot2(pk, k_b, half);
            x2 = rot_dot2(pk, k_a, half);
            pk = rot_pack(e4, x2);
            const i32 o3 = rot_dot2(pk, k_b, half);
            x2 = rot_dot2(pk, k_a, half);
            p0[i + stride] = (i16)o0;
            p0[i] = (i16)o1;
            p0[i - stride] = (i16)o2;
            p0[i - 2 * stride] = (i16)o3;
        }
        for (; i >= 0; i -= stride) {
            const u32 pk = rot_pack(p0[i], x2);
            p0[i + stride] = (i16)rot_dot2(pk, k_b, half);
            x2 = rot_dot2(pk, k_a, half);
        }
        p0[r] = (i16)x2;
    }
}
OG_DEV void rotate1_lane(i16 *xv, int x, int len, int stride, i32 c, i32 s) { // exp_rotation1 celt.cpp:684
    const u32 k_a = rot_pack(c, -s), k_b = rot_pack(s, c); // first output: c x1 - s x2; second (carried on): s x1 + c x2
    for (int r = 0; r < stride; r++) {
        if (r >= len - stride) break; // (the chains are in order: no later one has a pair either)
        rotate_chain(xv + x, r, len, stride, k_a, k_b);
    }
}

// A leaf's spreading rotation, put off to the wave pass below (on == false: the leaf has none).
struct RotJob {
    int x, blen, logB, stride2; // first coefficient, block length, log2 of the block count, the wide stride (0: only stride 1)
    i32 c, s;
    bool on;
};
// The rotations of the (up to 64) leaves the lanes of a wave have just decoded, by the WHOLE wave.  One leaf per lane costs the wave
// its largest rotated leaf's 4 N serial steps with eight lanes busy (a frame of the bench payloads rotates 8 of its 48 leaves:
// 103 steps for the largest on average).  But exp_rotation (celt.cpp:707) is B independent blocks, and its wide-stride sweep
// is `stride2` independent chains per block (rotate1_lane): here every (leaf, block, chain) of the wide sweeps gets a lane of its
// own (6 steps for the longest chain instead of 50), then every (leaf, block) one for the stride-1 sweep, which is serial (47
// steps).  How an item finds its leaf: the leaves' item counts are prefix-summed over the wave; a leaf's lane marks the first
// of its items in a 64-byte row of LDS with its own number, a prefix maximum over that row names every item's leaf, and the
// leaf's job comes over the lane crossbar.  All 64 lanes call this together.
OG_DEV void pvq_rotate_wave(i16 *xv, const RotJob &j, u8 *marker) {
    if (!__any(j.on)) return;
    const int lane = OG_LANE;
    const int w_geo = j.x | j.blen << 16, w_par = j.logB | j.stride2 << 8;
    const int w_cs = (int)((u32)(u16)j.c | (u32)(u16)j.s << 16);
    for (int pass = 0; pass < 2; pass++) { // the wide stride first (exp_rotation with dir = -1)
        int chains = 0;
        if (j.on) chains = pass == 0 ? (j.stride2 ? OG_MAX(0, OG_MIN(j.stride2, j.blen - j.stride2)) : 0) : (j.blen >= 2);
        const int cnt = chains << j.logB; // items: chain r of block b is item r << logB | b
        const int incl = wave_scan_add(cnt), excl = incl - cnt;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        for (int base = 0; base < total; base += 64) {
            marker[lane] = 0;
            OG_SYNC();
            if (cnt > 0 && excl < base + 64 && incl > base) marker[OG_MAX(excl - base, 0)] = (u8)(lane + 1);
            OG_SYNC();
            const int leaf = wave_scan_max((int)marker[lane]) - 1; // (>= 0: item `base` belongs to some leaf)
            const int item = base + lane;
            const bool work = item < total;
            const int src = leaf < 0 ? lane : leaf;
            const int g = __shfl(w_geo, src), q = __shfl(w_par, src), cs = __shfl(w_cs, src), first = __shfl(excl, src);
            if (work) {
                const int logB = q & 255, sub = item - first, b = sub & ((1 << logB) - 1), r = sub >> logB;
                const int blen = g >> 16;
                const i32 c = (i32)(i16)(cs & 0xffff), sn = (i32)(i16)(cs >> 16);
                i16 *const p0 = xv + (g & 0xffff) + b * blen;
                const i32 cc = pass == 0 ? sn : c, ss = pass == 0 ? c : sn; // exp_rotation1(.., stride2, s, c), then (.., 1, c, s)
                rotate_chain(p0, r, blen, pass == 0 ? q >> 8 : 1, rot_pack(cc, -ss), rot_pack(ss, cc));
            }
            OG_SYNC();
        }
    }
}
#endif

// U(a, b) for the leaf pass.  64 lanes walking 64 different leaves ask for 64 unrelated entries per step: from global
// memory that is one cache line per lane and the texture path serialises them (measured: a third of the walk at best, with
// the dense table evicted from L1 by the streaming traffic all the time).  Here rows 0..3 are closed forms and rows 4..14
// sit in LDS, stored by ROW with every column (rom_pvq_rr / rom_pvq_rb, 2.8 KB over the folding-history, pulse and scratch rows,
// none of which is in use during the leaf pass): U(r, c) = rr[rb[r] + c] for r = 4 .. 14 and any c.  Round 2 stored columns (one base per dimension n,
// fetched a step ahead); what the walk spends its time on since zero runs are skipped is the SEARCH for the next pulse's
// dimension at a fixed number of pulses k, i.e. along rows k and k + 1: with rows, a probe is two independent reads off two
// bases that change only when k does (a column base per probe made it two dependent round trips), a pulse's size candidates
// (rows 4..7 at column n) need no base at all, and the two entries of a step with n <= k are neighbours in row n.
#ifdef OG_RECON_TIGHT
// (og_state.hpp: rows 4 - 8 behind X, rows 9 - 11 and 12 - 14 in the two 320-byte tops of the spectrum that no band reaches -- a row's
// base is an offset from the table's first word, negative for those)
constexpr int PVQ_MAIN_LEN = ROM_PVQ_RB9, PVQ_TOP0_OFF = (X_TOP0 - V_NORM) / 2, PVQ_TOP1_OFF = (X_TOP1 - V_NORM) / 2;
static_assert((ROM_PVQ_RB12 - ROM_PVQ_RB9) * 2 <= 160 && (ROM_PVQ_RR_LEN - ROM_PVQ_RB12) * 2 + 32 <= 160, "the short rows (and the rotation marker) fit the tops");
#else
constexpr int PVQ_MAIN_LEN = ROM_PVQ_RR_LEN;
#endif
struct PvqLds {
    u32 rr[PVQ_MAIN_LEN];
    i16 rb[16];
    OG_MEMBER u32 at(int i) const { return reinterpret_cast<const u32 *>(this)[i]; } // entry i of the table (row base + column)
};
OG_DEV int pvq_lds_index(int t) { // where entry t of rom_pvq_rr lies, as an index from the table's first word
#ifdef OG_RECON_TIGHT
    return t < ROM_PVQ_RB9 ? t : t < ROM_PVQ_RB12 ? t - ROM_PVQ_RB9 + PVQ_TOP0_OFF : t - ROM_PVQ_RB12 + PVQ_TOP1_OFF;
#else
    return t;
#endif
}
#ifdef OG_RECON_TIGHT
static_assert(sizeof(PvqLds) <= (V_JOBM - V_NORM) * 2, "the PVQ table's long rows end before the jobs' collapse masks");
#else
static_assert(sizeof(PvqLds) <= (V_TOTAL - V_NORM) * 2, "the PVQ table overlays the folding-history, pulse and scratch rows");
#endif
OG_DEV PvqLds &pvq_lds() { return *reinterpret_cast<PvqLds *>(&S.v[V_NORM]); }
OG_DEV void pvq_tab_load() { // (the caller synchronises)
    u32 *const dst = reinterpret_cast<u32 *>(&pvq_lds());
#ifdef OG_HOST_EMUL
    OG_FOR_LANES(t, ROM_PVQ_RR_LEN) dst[pvq_lds_index(t)] = rom_pvq_rr[t];
    OG_FOR_LANES(t, 16) pvq_lds().rb[t] = (i16)pvq_lds_index(rom_pvq_rb[t]);
#else
    // every load requested before the first store waits for its data (a load - wait - store loop pays the L2's latency per pass)
    constexpr int NRR = (ROM_PVQ_RR_LEN + OG_NLANES - 1) / OG_NLANES;
    u32 rr[NRR];
#pragma unroll
    for (int k = 0; k < NRR; k++) rr[k] = rom_pvq_rr[OG_MIN(OG_LANE + k * OG_NLANES, ROM_PVQ_RR_LEN - 1)];
    const u16 rb = rom_pvq_rb[OG_LANE & 15];
#pragma unroll
    for (int k = 0; k < NRR; k++)
        if (OG_LANE + k * OG_NLANES < ROM_PVQ_RR_LEN) dst[pvq_lds_index(OG_LANE + k * OG_NLANES)] = rr[k];
    if (OG_LANE < 16) pvq_lds().rb[OG_LANE] = (i16)pvq_lds_index((int)rb);
#endif
}
// U(r, h) for a row r <= 3 (<= h), given U(2, h) and U(3, h); written without branches on purpose: the lanes of a wave
// ask for different rows, and as control flow every row would cost the wave a pass of its own
OG_DEV u32 pvq_row_sel(int r, u32 v2, u32 v3) {
    u32 v = (u32)(r >= 1);
    v = r == 2 ? v2 : v;
    return r == 3 ? v3 : v;
}
// U(3, h) = 2 h (h - 1) + 1 and the integer root the k = 2 zero run needs.  On the GPU: one 24-bit multiply-add (the compiler's own
// form of the expression is two masks and a full 32-bit multiply), and the bare v_sqrt_f32 -- one ulp, where the precise sqrtf is a
// dozen instructions of rounding fix-ups: its argument is below 2^15 here (tq <= 176^2), where neighbouring integers' roots are
// 0.0028 apart at least and a float's ulp is 2^-16, so the truncated result is the root's floor, or one less when the root is an
// integer -- which the caller's upward correction covers.
#ifdef OG_HOST_EMUL
OG_DEV u32 pvq_u3(u32 h) { return 2u * h * (h - 1u) + 1u; }
OG_DEV int pvq_isqrt_near(u32 tq) { return (int)__builtin_sqrtf((float)tq); }
#else
OG_DEV u32 pvq_u3(u32 h) { // h < 2^11
    u32 r;
    const u32 a = h << 1, b = h - 1u;
    asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
OG_DEV int pvq_isqrt_near(u32 tq) { return (int)__builtin_amdgcn_sqrtf((float)tq); }
#endif
OG_DEV int pvq_row_base(const PvqLds &T, int r) { return (int)T.rb[r < 4 ? 4 : (r > 14 ? 14 : r)]; } // (rows outside 4..14 are not table rows)

#ifndef OG_SKIP_RATIO
// A leaf's zero runs are skipped while it has more than this many dimensions per pulse left.  Measured in round 2 (k_celt_recon_fb
// alone / pipelined step): no skip 1.869 / 2.525 ms, ratio 1 (whenever n > k) 1.825 / 2.49, 2: 1.891, 3: 1.899, 4: 1.898 -- the wave's
// walk is 34 steps long on average without, 10 with (tools/leaf_balance.py).
#define OG_SKIP_RATIO 1
#endif
// `xv`: the spectrum arena of the leaf's frame (the calling wave's own working set -- or another wave's when the leaves of the
// workgroup's frames are pooled, og_recon.hip); `T`: the table copy to walk.
#ifdef OG_HOST_EMUL
struct RotJob;
#endif
// `defer`: the leaf's rotation is not done here but described there, for pvq_rotate_wave (GPU; the caller cleared defer->on)
OG_DEV u32 pvq_leaf_lane(i16 *xv, const PvqLds &T, int n, int k, u32 i, int pos, int B, i32 gain, int spread, RotJob *defer = nullptr) {
    const int N = n, K = k, x = pos;
    const int logB = ilog2(B), blen = N >> logB; // B is a power of two
    i32 yy = 0;
    // cwrsi celt.cpp:2545.  The reference has two code paths (k >= n: "lots of pulses", k < n: "lots of dimensions")
    // that differ only in how they walk its triangular table; with U(a, b) available for any pair both are
    //   s = (i >= U(n, k+1));  i -= s ? U(n, k+1) : 0;  k' = max { k' <= k : U(n, k') <= i };  value = +-(k - k');  i -= U(n, k')
    // The lanes of a wave decode different leaves and the wave waits for its longest one -- a leaf of many dimensions and
    // few pulses: its runs of zeros are skipped in one go (below), so a step of such a leaf places a pulse.  Steps with n <= k
    // take the general form below it.  The spectrum was cleared before the leaf pass: zeros are not stored.
    OG_MARK(56);
    int b0 = pvq_row_base(T, k), b1 = pvq_row_base(T, k + 1); // where rows k and k + 1 start (while they are table rows)
    while (n > 2) {
        if (k == 0) break; // every pulse is placed: what is left of the leaf stays zero
        u32 h = (u32)n, v2 = 2u * h - 1u, v3 = pvq_u3(h); // U(2, n), U(3, n)
        const bool sparse = n > k;
        u32 p0, p1;         // U(n, k), U(n, k + 1)
        int bn = 0;         // (n <= k) where row n starts
        bool tab = false;   // (n <= k) row n is a table row (n == 3: closed form)
        if (sparse) {
            const u32 c0 = T.at(k >= 4 ? b0 + n : 0), c1 = T.at(k >= 3 ? b1 + n : 0);
            p0 = k >= 4 ? c0 : pvq_row_sel(k, v2, v3);
            p1 = k >= 3 ? c1 : pvq_row_sel(k + 1, v2, v3);
            // A sparse leaf (many dimensions, few pulses) is mostly runs of zeros, and the wave waits for its longest leaf: the run is
            // skipped in one go.  With V(a) = U(a, k) + U(a, k + 1) the dimensions n, n-1, .., a+1 all decode to zero exactly when
            //     V(n) - V(a) <= 2 i < V(n) + V(a)          (one comparison: V(a) >= m, see below)
            // (the zero steps subtract U(n, k), U(n-1, k), ..: their sum down to a+1 is (V(n) - V(a)) / 2 by the recurrence
            // U(t, k+1) = U(t-1, k+1) + U(t, k) + U(t-1, k); the other bound is the one that keeps every step's sign test false);
            // V grows with a, so the smallest such a is found by bisection along rows k and k + 1.  Then i -= (V(n) - V(a)) / 2 and
            // the walk goes on at dimension a -- with a pulse, unless the search range ended there.
            // (tools/pvq_zero_run.py checks the identity against the step-by-step walk.)  Everything fits 32 bits: V(n) is the
            // size of a legal codebook, i < V(n), and with t = V(n) - i the two bounds in one read
            //     V(a) >= m,   m = i >= t ? i - t + 1 : t - i          (= d >= 0 ? d + 1 : -d for d = 2 i - V(n))
            if (k <= 13 && n > OG_SKIP_RATIO * k && n > 3) {
                const u32 Vn = p0 + p1, t = Vn - i, m = i >= t ? i - t + 1u : t - i;
                const int lo0 = k + 1 > 2 ? k + 1 : 2;
                int a;
                u32 t0 = p0, t1 = p1; // U(k, a), U(k + 1, a) of the dimension a the run ends at: the step below needs no second look
                if (k <= 2) { // V(a, 1) = 2 a and V(a, 2) = 2 a^2: solved, not searched (m <= V(n) <= 2 * 176^2)
                    const u32 tq = (m + 1u) >> 1;
                    int r = (int)tq;
                    if (k == 2) {
                        r = pvq_isqrt_near(tq);  // the root's floor, or one less (tq <= 176^2): its ceiling after
                        r += (u32)(r * r) < tq; // the correction
                    }
                    a = r > lo0 ? r : lo0;
                    const u32 u2 = 2u * (u32)a - 1u;
                    t0 = k == 1 ? 1u : u2;
                    t1 = k == 1 ? u2 : pvq_u3((u32)a);
                } else {
                    int lo = lo0, hi = n;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const u32 m0 = T.at(k >= 4 ? b0 + mid : 0), m1 = T.at(b1 + mid);
                        const u32 a0 = k >= 4 ? m0 : pvq_u3((u32)mid); // (row 3 in closed form)
                        if (a0 + m1 >= m) {
                            hi = mid;
                            t0 = a0;
                            t1 = m1;
                        } else
                            lo = mid + 1;
                    }
                    a = lo;
                }
                if (a < n) { // (a > k: the step below is still one with more dimensions than pulses)
                    i -= (Vn - (t0 + t1)) >> 1;
                    pos += n - a;
                    n = a;
                    if (n <= 2) break;
                    h = (u32)n;
                    v2 = 2u * h - 1u;
                    v3 = pvq_u3(h);
                    p0 = t0;
                    p1 = t1;
                }
            }
        } else { // n <= k: everything this step reads lies in row n, whose columns are all in LDS (n == 3: U(3, c) = 2 c (c - 1) + 1)
            const u32 hk = (u32)k;
            bn = pvq_row_base(T, n);
            tab = n >= 4;
            const u32 a0 = T.at(tab ? bn + k : 0), a1 = T.at(tab ? bn + k + 1 : 0);
            p0 = tab ? a0 : pvq_u3(hk);
            p1 = tab ? a1 : pvq_u3(hk + 1u);
        }
        const int s = -(int)(i >= p1);
        i -= p1 & (u32)s;
        if (p0 <= i && s == 0) {
            i -= p0;
        } else { // a pulse: the largest k' < k with U(n, k') <= i (U(n, 0) = 0 <= i; U(n, k) > i here)
            u32 plo = 0;
            int kk = 0;
            if (sparse) { // rows k' < k < n at column n
                if (k <= 8) {
                    // all candidates at once (rows 1..3 computed, rows 4..7 at fixed bases); U(., n) grows with the row: the last one
                    // that passes is k' (rows from k on are not looked at: a row ends where its entries leave 32 bits, and only
                    // U(k, n) and the entries below it are known to exist)
                    const u32 c4 = T.at(ROM_PVQ_RB4 + n), c5 = T.at(ROM_PVQ_RB5 + n), c6 = T.at(ROM_PVQ_RB6 + n), c7 = T.at(ROM_PVQ_RB7 + n);
                    const u32 cand[7] = {1u, v2, v3, c4, c5, c6, c7};
#pragma unroll
                    for (int r = 1; r <= 7; r++) {
                        const bool ok = r < k && cand[r - 1] <= i;
                        kk = ok ? r : kk;
                        plo = ok ? cand[r - 1] : plo;
                    }
                } else {
                    int lo = 0, hi = k - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        const u32 tm = T.at(mid >= 4 ? pvq_row_base(T, mid) + n : 0);
                        const u32 pm = mid >= 4 ? tm : pvq_row_sel(mid, v2, v3);
                        if (pm <= i) {
                            lo = mid;
                            plo = pm;
                        } else
                            hi = mid - 1;
                    }
                    kk = lo;
                }
            } else { // along row n
                int lo = 0, hi = k - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1; // >= 1
                    const u32 tm = T.at(tab ? bn + mid : 0);
                    const u32 pm = tab ? tm : pvq_u3((u32)mid);
                    if (pm <= i) {
                        lo = mid;
                        plo = pm;
                    } else
                        hi = mid - 1;
                }
                kk = lo;
            }
            const int val = (k - kk + s) ^ s;
            k = kk;
            i -= plo;
            xv[pos] = (i16)val;
            yy += val * val;
            b0 = pvq_row_base(T, k); // (rows k and k + 1, for the steps with more dimensions than pulses)
            b1 = pvq_row_base(T, k + 1);
        }
        pos++;
        n--;
    }
    {
        const u32 p = 2 * (u32)k + 1;
        int s = -(int)(i >= p);
        i -= p & (u32)s;
        const int k0 = k;
        k = (int)((i + 1) >> 1);
        if (k) i -= 2 * (u32)k - 1;
        int val = (k0 - k + s) ^ s;
        xv[pos++] = (i16)val;
        yy += val * val;
        s = -(int)i;
        val = (k + s) ^ s;
        xv[pos] = (i16)val;
        yy += val * val;
    }
    // collapse mask from the pulses
    OG_MARK(57);
    u32 cm = 1;
    if (B > 1) {
        cm = 0;
        for (int b = 0, j = 0; b < B; b++) {
            u32 any = 0;
            for (int e = 0; e < blen; e++, j++) any |= (u32)(u16)xv[x + j];
            cm |= (u32)(any != 0) << b;
        }
    }
    // scale the pulses in place
    OG_MARK(58);
    const int kk = ilog2(yy) >> 1;
    const i32 t = vshr32(yy, 2 * (kk - 7));
    const i32 g = tr16(mul16_p15(rsqrt_norm(t), gain));
    for (int j = 0; j < N; j++) xv[x + j] = (i16)pshr32(mul16(g, xv[x + j]), kk + 1);
    OG_MARK(59);
    if (2 * K < N && spread != 0) {
        const int factor = spread == 1 ? 15 : (spread == 2 ? 10 : 5);
        const i32 rg = tr16(mul32_q31(mul16(32767, N), celt_rcp(N + factor * K))); // celt_div celt.h:367
        const i32 theta = tr16(mul16_q15(rg, rg) >> 1);
        const i32 c = cos_norm(theta), s = cos_norm(sub16(32767, theta));
        int stride2 = 0;
        if (N >= 8 * B) {
            stride2 = 1;
            while ((stride2 * stride2 + stride2) * B + (B >> 2) < N) stride2++;
        }
#if !defined(OG_HOST_EMUL)
        if (defer) {
            defer->x = x;
            defer->blen = blen;
            defer->logB = logB;
            defer->stride2 = stride2;
            defer->c = c;
            defer->s = s;
            defer->on = true;
            return cm;
        }
#endif
        for (int blk2 = 0; blk2 < B; blk2++) {
            if (stride2) rotate1_lane(xv, x + blk2 * blen, blen, stride2, s, c);
            rotate1_lane(xv, x + blk2 * blen, blen, 1, c, s);
        }
    }
    return cm;
}

// The collapse mask of a PVQ leaf goes, pre-shifted, into its JOB's word (S.job_mask_row(): 2 x band + decode slot; cleared by
// recon_begin) -- a job's mask is the OR of its leaves' (cm(job) |= cm(leaf) << off, see parse_tree).  Round 5: a row of one mask
// per LEAF (416 x u16) was a tenth of the reconstruction kernel's LDS; the lanes of a round's leaves OR into the row together.
OG_DEV void job_mask_or(int job, u32 m) {
#ifdef OG_HOST_EMUL
    S.job_mask_row()[job] |= m;
#else
    __hip_atomic_fetch_or(&S.job_mask_row()[job], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// The record's word stream is consumed strictly in order: a 64-word window in LDS, refilled by one coalesced load.
constexpr int REC_WORDS_CAP = (REC_MAX_WORDS + 64) / 64 * 64; // size of ParseRec::words
struct RecCur { // read positions in the record: next word, first word of the window in LDS (-64: none), next PVQ leaf
    const u32 *words;
    int w, base, leaf;
};
OG_DEV u32 rec_word(RecCur &cur) {
    if ((unsigned)(cur.w - cur.base) >= 64u) { // the window moves to the word wanted (the sequential walk: every 64 words)
        OG_SYNC();
        OG_FOR_LANES(l, 64) S.word_window()[l] = cur.words[OG_MIN(cur.w + l, REC_WORDS_CAP - 1)];
        OG_SYNC();
        cur.base = cur.w;
    }
    const u32 w = (u32)OG_UNI(S.word_window()[cur.w - cur.base]);
    cur.w++;
    return w;
}

// Four consecutive words (a band's header): when they lie inside the current window -- 15 times out of 16 -- the four LDS
// reads have no refill check between them and issue together (one latency instead of four).
OG_DEV void rec_word4(RecCur &cur, u32 &w0, u32 &w1, u32 &w2, u32 &w3) {
    const int at = cur.w - cur.base;
    if (at >= 0 && at <= 60) {
        const u32 *win = S.word_window();
        const u32 a = win[at], b = win[at + 1], c = win[at + 2], d = win[at + 3];
        w0 = (u32)OG_UNI(a);
        w1 = (u32)OG_UNI(b);
        w2 = (u32)OG_UNI(c);
        w3 = (u32)OG_UNI(d);
        cur.w += 4;
        return;
    }
    w0 = rec_word(cur);
    w1 = rec_word(cur);
    w2 = rec_word(cur);
    w3 = rec_word(cur);
}

// The noise generator jumped ahead (lcg_skip) by n = lane + 1, lane + 65, lane + 129 steps: s -> a s + c.  Computed
// once per frame; every noise / dither pass then costs one multiply-add per coefficient.
struct LcgTab {
#if !defined(OG_HOST_EMUL)
    // The jumps by 1 .. 192 steps are constants (rom_lcg_jump, tools/gen_rom_tables.py): a noise sample reads its pair from there.
    // (Rounds 2 - 3 kept the wave's six values in this object; the compiler put the object in scratch memory and turned at()'s
    // selects into indexed loads from it -- two trips to memory per sample where this is one, and the kernel's only scratch
    // traffic: 2 KB per frame written and read back through HBM.)
    OG_MEMBER void init() {}
    OG_MEMBER u32 at(u32 seed, int j) const { // seed advanced by (j + 1) steps, 0 <= j < 192
        const u32 a = rom_lcg_jump[2 * j], c = rom_lcg_jump[2 * j + 1];
        return a * seed + c;
    }
#else
    u32 a[3], c[3];
    OG_MEMBER void init() {
        for (int k = 0; k < 3; k++) {
            u32 n = (u32)(OG_LANE + 64 * k + 1), ra = 1u, rc = 0u, ba = 1664525u, bc = 1013904223u;
            while (n) {
                if (n & 1u) {
                    ra = ba * ra;
                    rc = ba * rc + bc;
                }
                bc = ba * bc + bc;
                ba = ba * ba;
                n >>= 1;
            }
            a[k] = ra;
            c[k] = rc;
        }
    }
    // seed advanced by (j + 1) steps, j = lane + 64 k   [host emulation: one lane, j arbitrary]
    OG_MEMBER u32 at(u32 seed, int j) const {
#ifdef OG_HOST_EMUL
        return lcg_skip(seed, (u32)j + 1);
#else
        const int k = j >> 6;
        return (k == 0 ? a[0] : k == 1 ? a[1] : a[2]) * seed + (k == 0 ? c[0] : k == 1 ? c[1] : c[2]);
#endif
    }
#endif
};

// anti_collapse (celt.cpp:1010) for the reconstruction kernel of 20 ms frames.  The shared form (og_celt_bands.hpp) derives a band's
// noise amplitude r -- a division, an exp2, a reciprocal square root -- in every lane alike, up to 42 times one after the other (the
// values come from LDS rows: vector work, not scalar), and steps the noise generator with its squaring loop per lane: a frame with
// anti-collapse (one in sixteen of the bench payloads) cost the wave 60 % more than one without.  Here lane (channel, band) derives
// its own r, one pass for all of them, into a scratch row; the fills read it back and jump the generator by the frame's table.
OG_DEV void anti_collapse_pm(const LcgTab &lcg, int LM, int C, int size, int start, int end, u32 seed) {
    i16 *const rrow = &S.v[V_TMP]; // (the band loop's scratch row: free by now)
    OG_SYNC();
    OG_FOR_LANES(l, C * NBANDS) {
        const int c = l >= NBANDS ? 1 : 0, i = l - c * NBANDS;
        i32 r = 0;
        if (i >= start && i < end) {
            const int N0 = rom_eband[i + 1] - rom_eband[i];
            const int depth = (int)(udiv((u32)(1 + S.pulses_row()[i]), (u32)N0) >> LM);
            const i32 thresh32 = celt_exp2(-shl16(depth, 10 - BITRES)) >> 1;
            const i32 thresh = tr16(mul16x32_q15(16384, OG_MIN(32767, thresh32)));
            i32 t = N0 << LM;
            const int shift = ilog2(t) >> 1;
            t = shl32(t, (7 - shift) << 1);
            const i32 sqrt_1 = rsqrt_norm(t);
            i32 prev1 = S.logE1_row()[c * NBANDS + i], prev2 = S.logE2_row()[c * NBANDS + i];
            if (C == 1) {
                prev1 = OG_MAX(prev1, (i32)S.logE1_row()[NBANDS + i]);
                prev2 = OG_MAX(prev2, (i32)S.logE2_row()[NBANDS + i]);
            }
            i32 Ediff = (i32)S.bandE_row()[c * NBANDS + i] - OG_MIN(prev1, prev2);
            Ediff = OG_MAX(0, Ediff);
            if (Ediff < 16384) {
                const i32 r32 = celt_exp2(-tr16(Ediff)) >> 1;
                r = tr16(2 * OG_MIN(16383, r32));
            }
            if (LM == 3) r = tr16(mul16_q14(23170, OG_MIN(23169, r)));
            r = tr16(OG_MIN(thresh, r) >> 1);
            r = tr16(mul16_q15(sqrt_1, r) >> shift);
        }
        rrow[l] = (i16)r;
    }
    OG_SYNC();
    for (int i = start; i < end; i++) {
        const int N0 = rom_eband[i + 1] - rom_eband[i];
        for (int c = 0; c < C; c++) {
            const i32 r = (i32)OG_UNI(rrow[c * NBANDS + i]);
            const int x = V_X + c * size + (rom_eband[i] << LM);
            int renorm = 0;
            const u32 mask = (u32)OG_UNI(S.cmask_row()[i * C + c]);
            for (int k = 0; k < 1 << LM; k++) {
                if (!(mask & (1u << k))) {
                    OG_SYNC();
                    OG_FOR_LANES(j, N0) S.v[x + (j << LM) + k] = (i16)((lcg.at(seed, j) & 0x8000) ? r : -r);
                    seed = lcg_skip(seed, (u32)N0);
                    renorm = 1;
                }
            }
            if (renorm) renormalise(x, N0 << LM, 32767);
        }
    }
}

// The leaves of one job (quant_partition celt.cpp:1382 flattened by the parse kernel), vector half.  The leaves with
// pulses are complete already (pvq_leaf_lane) and only contribute their collapse masks, which the leaf pass ORed, pre-shifted,
// into the job's word of S.job_mask_row() (`job`: 2 x band + decode slot).  A leaf without pulses is zeroed, noise-filled or folded from the lower band
// (celt.cpp:1481-1520).  `jw`: the job's header word.  Returns the job's collapse mask.
OG_DEV u32 recon_job_leaves(RecCur &cur, const LcgTab &lcg, u32 jw, u32 &seed_io, int x_job, int low_job, i32 fill_job, int job) {
    const int n_fill = (int)(jw & 31), n_pvq = (int)(jw >> JW_NPVQ_SHIFT) & 31;
    u32 cm_job = n_pvq ? (u32)OG_UNI(S.job_mask_row()[job]) : 0u;
    for (int f = 0; f < n_fill; f++) {
        OG_MARK(7);
        const u32 w = rec_word(cur), w1 = rec_word(cur);
        const int off = (int)(w >> LW_OFF_SHIFT) & 15, B = ((int)(w >> LW_B_SHIFT) & 15) + 1, N = (int)(w >> LW_N_SHIFT) & 255;
        const int x = V_X + (int)(w1 & 2047);
        const i32 gain = (i32)((w1 >> 11) & 0xffff);
        const u32 cm_mask = (u32)((1ull << B) - 1);
        const i32 fill = (i32)((u32)(fill_job >> off) & cm_mask);
        OG_STAT(10, fill == 0);                     // fill leaves left zero
        OG_STAT(11, fill != 0 && low_job < 0);      // ... noise
        OG_STAT(12, fill != 0 && low_job >= 0);     // ... folded
        if (fill) { // (no fill: the leaf stays zero, as the spectrum was initialised)
            const u32 seed = seed_io;
            u32 cm;
            OG_SYNC();
            if (low_job < 0) { // noise
                OG_FOR_LANES(j, N) S.v[x + j] = (i16)((i32)lcg.at(seed, j) >> 20);
                cm = cm_mask;
            } else { // folded spectrum, +-1/256 dither
                const int low = low_job + (x - x_job);
                OG_FOR_LANES(j, N) S.v[x + j] = (i16)(S.v[low + j] + ((lcg.at(seed, j) & 0x8000) ? 4 : -4));
                cm = (u32)fill;
            }
            seed_io = lcg_skip(seed, (u32)N);
            renormalise(x, N, gain);
            cm_job |= cm << off;
        }
        OG_MARK(6);
    }
    return cm_job;
}

// Haar / Hadamard helpers with power-of-two strides taken as shifts (no integer division in the lane loops)
OG_DEV void haar1_p2(int x, int N0, int log_stride) { // haar1 celt.cpp:1202, stride = 1 << log_stride
    N0 >>= 1;
    const int stride = 1 << log_stride;
    OG_SYNC();
    OG_FOR_LANES(id, N0 << log_stride) {
        const int j = id >> log_stride, i = id & (stride - 1);
        const int a = x + stride * 2 * j + i, b = a + stride;
        const i32 t1 = mul16(23170, S.v[a]), t2 = mul16(23170, S.v[b]);
        S.v[a] = (i16)pshr32(t1 + t2, 15);
        S.v[b] = (i16)pshr32(t1 - t2, 15);
    }
    OG_SYNC();
}
// (de)interleave_hadamard celt.cpp:1162 / :1183; stride = 1 << log_stride.  Lanes enumerate the interleaved index.
OG_DEV void hadamard_p2(int x, int N0, int log_stride, int hadamard, int dir) {
    const int stride = 1 << log_stride, N = N0 << log_stride;
    OG_SYNC();
    OG_FOR_LANES(inter, N) {
        const int j = inter >> log_stride, i = inter & (stride - 1);
        const int blocked = (hadamard ? ordery(stride, i) : i) * N0 + j;
        if (dir == 0)
            S.v[V_TMP + blocked] = S.v[x + inter];
        else
            S.v[V_TMP + inter] = S.v[x + blocked];
    }
    OG_SYNC();
    OG_FOR_LANES(id, N) S.v[x + id] = S.v[V_TMP + id];
    OG_SYNC();
}

// quant_band celt.cpp:1526, vector half; N > 1.  `scale`: sqrt(N) for the folding history (from the record).
OG_DEV u32 recon_band_mono(RecCur &cur, u32 jw, const LcgTab &lcg, int tf_change, u32 &seed, int x, int N, int B, int low, int low_out,
                           i32 scale, int low_scratch, i32 fill, int job) {
    const int N0 = N, longBlocks = B == 1;
    int logB = ilog2(B), time_divide = 0, recombine = 0;
    int N_B = N >> logB;
    OG_STAT(1, 1);                                  // jobs
    OG_STAT(2, (jw & JW_NEED_LOW) && low >= 0);     // jobs that prepare a folding source
    OG_STAT(3, (int)(jw & 31));                     // fill leaves
    OG_STAT(4, (int)(jw >> JW_NPVQ_SHIFT) & 31);    // PVQ leaves
    OG_STAT(5, tf_change != 0);                     // jobs with a tf change
    OG_STAT(6, B > 1);                              // jobs in short-block frames
    if (!(jw & JW_NEED_LOW)) low = -1; // no leaf of this job folds: skip the whole preparation of the folding source
    if (tf_change > 0) recombine = tf_change;
    if (low_scratch >= 0 && low >= 0 && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B > 1)) {
        OG_SYNC();
        OG_FOR_LANES(j, N) S.v[low_scratch + j] = S.v[low + j];
        OG_SYNC();
        low = low_scratch;
    }
    for (int k = 0; k < recombine; k++) {
        if (low >= 0) haar1_p2(low, N >> k, k);
        int lo = fill & 0xF, hi = fill >> 4; // bit_interleave_table celt.cpp:1560
        int tl = (lo & 3 ? 1 : 0) | (lo & 12 ? 2 : 0), th = (hi & 3 ? 1 : 0) | (hi & 12 ? 2 : 0);
        fill = tl | th << 2;
    }
    logB -= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        if (low >= 0) haar1_p2(low, N_B, logB);
        fill |= fill << (1 << logB);
        logB++;
        N_B >>= 1;
        time_divide++;
        tf_change++;
    }
    const int logB0 = logB, N_B0 = N_B, B0 = 1 << logB0;
    if (B0 > 1 && low >= 0) hadamard_p2(low, N_B >> recombine, logB0 + recombine, longBlocks, 0);
    OG_MARK(6);
    u32 cm = recon_job_leaves(cur, lcg, jw, seed, x, low, fill, job);
    OG_MARK(8);
    if (B0 > 1) hadamard_p2(x, N_B >> recombine, logB0 + recombine, longBlocks, 1);
    OG_STAT(7, B0 > 1);                             // jobs that undo a Hadamard interleave on x
    OG_STAT(8, time_divide + recombine);            // Haar passes on x
    OG_STAT(9, low_out >= 0);                       // jobs that write folding history
    N_B = N_B0;
    for (int k = 0; k < time_divide; k++) {
        logB--;
        N_B <<= 1;
        cm |= cm >> (1 << logB);
        haar1_p2(x, N_B, logB);
    }
    for (int k = 0; k < recombine; k++) {
        u32 c4 = cm & 0xF; // bit_deinterleave_table celt.cpp:1606
        cm = ((c4 & 1) * 0x03) | ((c4 >> 1 & 1) * 0x0C) | ((c4 >> 2 & 1) * 0x30) | ((c4 >> 3 & 1) * 0xC0);
        haar1_p2(x, N0 >> k, k);
    }
    logB += recombine;
    OG_MARK(9);
    if (low_out >= 0) {
        OG_SYNC();
        OG_FOR_LANES(j, N0) S.v[low_out + j] = (i16)mul16_q15(scale, S.v[x + j]);
        OG_SYNC();
    }
    return cm & ((1u << (1 << logB)) - 1);
}

// quant_all_bands celt.cpp:1754, vector half: an interpreter of the record's word stream
OG_DEV void recon_all_bands(const u32 *words, u32 need_norm, const LcgTab &lcg, int start, int end, int C, int N_ch, int shortBlocks, int LM,
                            u32 &seed_io) {
    const int M = 1 << LM, B = shortBlocks ? M : 1;
    OG_STAT(0, 1);                 // frames
    OG_STAT(19, shortBlocks != 0); // transient frames
    const int norm_offset = M * rom_eband[start];
    const int norm = V_NORM, norm2 = V_NORM + M * rom_eband[NBANDS - 1] - norm_offset;
    // The reference borrows the last band's spectrum slot as scratch; here that slot already holds the band's
    // decoded pulses, so the scratch row lives in the (otherwise unused) pulse row.
    int low_scratch = V_IY;
    RecCur cur;
    cur.words = words;
    cur.w = 0;
    cur.base = -64;
    cur.leaf = 0;
    u32 seed = seed_io;
    for (int i = start; i < end; i++) {
        OG_MARK(3);
        const int last = i == end - 1;
        u32 w0, w1, w2, w3;
        cur.w = (cur.w + 3) & ~3; // (a band's header starts on a multiple of four: RecWriter::band_begin)
        rec_word4(cur, w0, w1, w2, w3);
        const int eb0 = (int)(w1 >> 11) & 2047, N = (int)(w1 >> 22) & 255;
        const int x = V_X + eb0, y = C == 2 ? V_X + N_ch + eb0 : -1;
        const int dual_stereo = (w0 & BW_DUAL) != 0;
        if (i == start + 1) { // special_hybrid_folding celt.cpp:1743
            int n1 = M * (rom_eband[start + 1] - rom_eband[start]), n2 = M * (rom_eband[start + 2] - rom_eband[start + 1]);
            if (n2 > n1) {
                OG_SYNC();
                OG_FOR_LANES(j, n2 - n1) {
                    S.v[norm + n1 + j] = S.v[norm + 2 * n1 - n2 + j];
                    if (w0 & BW_DUAL_PRE) S.v[norm2 + n1 + j] = S.v[norm2 + 2 * n1 - n2 + j];
                }
                OG_SYNC();
            }
        }
        const int tf_change = (int)((w0 >> BW_TF_SHIFT) & 7) - 4;
        OG_STAT(13, 1);                                             // bands
        OG_STAT(14, N == 1);                                        // N == 1 bands
        OG_STAT(15, (w0 & BW_STEREO) && N > 2);                     // bands that end in a stereo merge
        OG_STAT(16, (w0 & BW_STEREO) && N == 2);                    // N == 2 stereo bands
        OG_STAT(17, dual_stereo);                                   // dual-stereo bands
        OG_STAT(18, (w0 & BW_HAS_LOW) != 0);                        // bands with a folding source available
        if (last) low_scratch = -1;
        u32 x_cm, y_cm;
        if (w0 & BW_HAS_LOW) {
            const int fold_end = (int)(w0 >> BW_FOLD1_SHIFT) & 31;
            int fold_i = (int)(w0 >> BW_FOLD0_SHIFT) & 31;
            x_cm = y_cm = 0;
            do {
                x_cm |= (u32)OG_UNI(S.cmask_row()[fold_i * C + 0]);
                y_cm |= (u32)OG_UNI(S.cmask_row()[fold_i * C + C - 1]);
            } while (++fold_i < fold_end);
        } else
            x_cm = y_cm = (1u << B) - 1;
        if (w0 & BW_DUAL_END) {
            OG_SYNC();
            OG_FOR_LANES(j, eb0 - norm_offset) S.v[norm + j] = (i16)((S.v[norm + j] + S.v[norm2 + j]) >> 1);
            OG_SYNC();
        }
        const int eff = (w0 & BW_HAS_LOW) ? (int)(w1 & 2047) : -1;
        const int low1 = eff >= 0 ? norm + eff : -1, low2 = eff >= 0 ? norm2 + eff : -1;
        // the folding history of a band nobody folds from is not computed at all
        const int want_out = !last && ((need_norm >> i) & 1u);
        const int out1 = want_out ? norm + eb0 - norm_offset : -1, out2 = want_out ? norm2 + eb0 - norm_offset : -1;

        if (N == 1) { // quant_band_n1 celt.cpp:1357
            OG_MARK(11);
            OG_SYNC();
            S.v[x] = (i16)((w0 & BW_SIGN0) ? -16384 : 16384);
            if (y >= 0) S.v[y] = (i16)((w0 & BW_SIGN1) ? -16384 : 16384);
            OG_SYNC();
            if (out1 >= 0) S.v[out1] = (i16)(S.v[x] >> 4);
            if (dual_stereo && out2 >= 0) S.v[out2] = (i16)(S.v[y] >> 4);
            OG_SYNC();
            x_cm = y_cm = 1;
        } else {
            OG_MARK(4);
            const int stereo = (w0 & BW_STEREO) != 0, mid_first = (w0 & BW_MID_FIRST) != 0, swap_c = (w0 & BW_SWAP) != 0;
            const i32 imid = (i32)(i16)(w2 & 0xffff), iside = (i32)(w2 >> 16), scale = (i32)(i16)(w3 & 0xffff);
            i32 fill0 = (i32)(x_cm | y_cm);
            const i32 orig_fill = fill0;
            int n2case = 0, njobs = 1;
            if (stereo) {
                if (w0 & BW_THETA0) fill0 &= (1 << B) - 1;
                if (w0 & BW_THETA1) fill0 &= ((1 << B) - 1) << B;
                if (N == 2)
                    n2case = 1;
                else
                    njobs = 2;
            } else if (dual_stereo)
                njobs = 2;
            u32 cm0 = 0, cm1 = 0;
#pragma nounroll
            for (int jb = 0; jb < njobs; jb++) {
                int jx, jlow, jout, jscr;
                i32 jfill;
                if (dual_stereo) {
                    jx = jb ? y : x; jlow = jb ? low2 : low1; jout = jb ? out2 : out1; jscr = low_scratch;
                    jfill = (i32)(jb ? y_cm : x_cm);
                } else if (!stereo) {
                    jx = x; jlow = low1; jout = out1; jscr = low_scratch; jfill = fill0;
                } else if (n2case) {
                    jx = swap_c ? y : x; jlow = low1; jout = out1; jscr = low_scratch; jfill = orig_fill;
                } else if ((jb == 0) == (mid_first != 0)) {
                    jx = x; jlow = low1; jout = out1; jscr = low_scratch; jfill = fill0;
                } else {
                    jx = y; jlow = -1; jout = -1; jscr = -1; jfill = fill0 >> B;
                }
                OG_MARK(5);
                const u32 jw = rec_word(cur);
                const u32 cmj = recon_band_mono(cur, jw, lcg, tf_change, seed, jx, N, B, jlow, jout, scale, jscr, jfill, 2 * i + jb);
                if (jb == 0) cm0 = cmj; else cm1 = cmj;
            }
            OG_MARK(10);
            if (stereo) {
                if (n2case) { // N == 2: the side is the mid rotated by 90 degrees (celt.cpp:1659-1697)
                    const int x2 = swap_c ? y : x, sign = (w0 & BW_SIGN) ? -1 : 1;
                    OG_SYNC();
                    const i32 a0 = S.v[x2], a1 = S.v[x2 + 1];
                    const i32 b0 = tr16(-sign * a1), b1 = tr16(sign * a0);
                    i32 X0 = swap_c ? b0 : a0, X1 = swap_c ? b1 : a1, Y0 = swap_c ? a0 : b0, Y1 = swap_c ? a1 : b1;
                    X0 = tr16(mul16_q15(imid, X0));
                    X1 = tr16(mul16_q15(imid, X1));
                    Y0 = tr16(mul16_q15(iside, Y0));
                    Y1 = tr16(mul16_q15(iside, Y1));
                    OG_SYNC();
                    S.v[x] = (i16)sub16(X0, Y0);
                    S.v[y] = (i16)add16(X0, Y0);
                    S.v[x + 1] = (i16)sub16(X1, Y1);
                    S.v[y + 1] = (i16)add16(X1, Y1);
                    OG_SYNC();
                } else
                    stereo_merge(x, y, imid, N);
                if (w0 & BW_INV) {
                    OG_SYNC();
                    OG_FOR_LANES(j, N) S.v[y + j] = (i16)(-S.v[y + j]);
                    OG_SYNC();
                }
                x_cm = y_cm = cm0 | cm1;
            } else if (dual_stereo) {
                x_cm = cm0;
                y_cm = cm1;
            } else
                x_cm = y_cm = cm0;
        }
        S.cmask_row()[i * C + 0] = (u8)x_cm;
        S.cmask_row()[i * C + C - 1] = (u8)y_cm;
    }
    seed_io = seed;
}

// =====================================================================================================
//  recon, phase-major band loop (20 ms frames from band 0: every CELT-only frame of the BASELINE workloads)
// =====================================================================================================
// recon_all_bands above walks the bands one after another because the reference does; but once the leaf pass has run,
// what a band still needs is local to it -- undoing its time-frequency change (Haar / Hadamard), its stereo merge --
// except for two things that look back: a leaf WITHOUT pulses is filled from earlier bands (their collapse masks, their
// spectrum as the folding source, the noise seed), and anti-collapse needs every band's mask.  Measured on the bench
// payloads 3 of a frame's 42 jobs have such a leaf; the per-band walk nevertheless paid ~20 dependent LDS round trips and
// ~700 scalar instructions per band for control (half of k_celt_recon's time, 15 k SALU instructions per frame).  Here:
//   B  one LANE per job (band x decode slot) reads the job's words, derives its time-frequency steps and -- for jobs
//      whose leaves all carry pulses -- its collapse mask;
//   C  those jobs' time-frequency changes are undone for the whole spectrum at once, a lane per group of 8 coefficients
//      (every band of a 20 ms frame is a multiple of 8 wide and 16-byte aligned): the interleave as a gather, Haar steps
//      of stride 1 / 2 / 4 in registers;
//   D  the jobs that do fill a leaf run one after another in decode order through the same code as the band walk
//      (recon_band_mono); their folding source is made on demand from the spectrum (the band walk's `norm` rows are
//      exactly scale(band) * X of earlier bands, taken before the stereo merge -- which is why the merge waits for E);
//   E  all stereo merges: partial sums per group of 8, one lane per band for the gains, one apply pass;
//   F  collapse masks per band (anti-collapse reads them).
// Same arithmetic per coefficient as the band walk (src/celt.cpp:1526-1741, :1113-1213), reordered only where the
// reference's order carries no dependency.
struct PmLds { // overlays the folding-history rows S.v[V_NORM ..], which this path never materialises
    u32 jdesc[2 * NBANDS];  // per (band, channel): JD_*
    u32 jaux[2 * NBANDS];   // per (band, decode slot): word position of the job header | channel << 16 | exists << 17 | fills << 18 | has pulses << 19
    u32 bw0[NBANDS], bw1[NBANDS], bw2[NBANDS];
    i32 scale[NBANDS];
    u32 mpar[NBANDS][2];    // stereo merge of the band: mode | kl << 8 | kr << 16, lgain | rgain << 16 (the mid gain is in bw2)
    u16 jcm[2 * NBANDS];    // per (band, channel): the job's collapse mask
    u8 binband[100], binoff[100]; // 5 ms bin (= group of 8 coefficients; 100 of them are coded) -> band, group index within the band
};
#ifdef OG_RECON_TIGHT
constexpr int V_PART = V_IY; // the stereo merges' partial sums take the two scratch rows (the fill jobs are done by then)
static_assert(sizeof(PmLds) <= sizeof(i16) * (V_IY - V_NORM) && 800 <= sizeof(i16) * (V_WIN - V_IY), "the phase-major tables");
#else
constexpr int V_PART = V_NORM + 600;
static_assert(sizeof(PmLds) <= sizeof(i16) * 600 && 600 + 400 <= 1248, "the phase-major tables overlay the folding-history rows");
#endif
OG_DEV PmLds &PM() { return *reinterpret_cast<PmLds *>(&S.v[V_NORM]); }
typedef i32 PmPart[2]; // per group of 8 coefficients: sum y*x, sum y*y
OG_DEV PmPart *pm_part() { return reinterpret_cast<PmPart *>(&S.v[V_PART]); }

enum { // PmLds::jdesc
    JD_VALID = 1, JD_FILL = 2,          // the job exists / has a leaf without pulses (phase D does everything for it)
    JD_PERM_SHIFT = 2 /* 3 bits: log2 of the interleave stride, 0 = none */, JD_HAD = 1 << 5,
    JD_STEP_SHIFT = 8 /* 3 Haar steps x 4 bits: 0 none, else log2(stride) + 1 */
};

struct V8 { i32 v[8]; };
OG_DEV V8 ld8(int pos) { // eight consecutive coefficients, pos a multiple of 8
    V8 r;
#ifdef OG_HOST_EMUL
    for (int k = 0; k < 8; k++) r.v[k] = S.v[pos + k];
#else
    const og_v4i p = *reinterpret_cast<const og_v4i *>(&S.v[pos]);
    r.v[0] = (i32)(i16)p.x; r.v[1] = p.x >> 16; r.v[2] = (i32)(i16)p.y; r.v[3] = p.y >> 16;
    r.v[4] = (i32)(i16)p.z; r.v[5] = p.z >> 16; r.v[6] = (i32)(i16)p.w; r.v[7] = p.w >> 16;
#endif
    return r;
}
OG_DEV void st8(int pos, const V8 &r) {
#ifdef OG_HOST_EMUL
    for (int k = 0; k < 8; k++) S.v[pos + k] = (i16)r.v[k];
#else
    og_v4i p;
    p.x = (i32)(((u32)r.v[0] & 0xffffu) | (u32)r.v[1] << 16); p.y = (i32)(((u32)r.v[2] & 0xffffu) | (u32)r.v[3] << 16);
    p.z = (i32)(((u32)r.v[4] & 0xffffu) | (u32)r.v[5] << 16); p.w = (i32)(((u32)r.v[6] & 0xffffu) | (u32)r.v[7] << 16);
    *reinterpret_cast<og_v4i *>(&S.v[pos]) = p;
#endif
}
OG_DEV void haar_pair(i32 &a, i32 &b) { // one butterfly of haar1 celt.cpp:1202
    const i32 t1 = mul16(23170, a), t2 = mul16(23170, b);
    a = tr16(pshr32(t1 + t2, 15));
    b = tr16(pshr32(t1 - t2, 15));
}
template <int S_>
OG_DEV void haar8(V8 &r) { // the Haar step of stride S_ (1, 2, 4) inside one group of 8
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (!(i & S_)) haar_pair(r.v[i], r.v[i + S_]);
}

struct PmGrp { int band, job, x, gj, N; u32 jd; };
constexpr int PM_GROUPS = 100; // coded groups of 8 per channel (eband5ms[21] = 100)
OG_DEV PmGrp pm_group(int g) { // group g of the coded spectrum: 0..99 first channel, 100..199 second
    const PmLds &P = PM();
    PmGrp r;
    const int ch = g >= PM_GROUPS, bin = g - PM_GROUPS * ch;
    r.band = P.binband[bin];
    r.gj = P.binoff[bin];
    r.job = 2 * r.band + ch;
    r.jd = P.jdesc[r.job];
    r.x = V_X + 960 * ch + 8 * (bin - r.gj);
    r.N = (int)(P.bw1[r.band] >> 22) & 255;
    return r;
}

// B: one lane per (band, decode slot).  Returns (wave-uniform) which time-frequency passes some job needs: bit 0 the
// interleave, bit 1 + 4 k + c the Haar stride 1 << c at step k; fill_lo / fill_hi: the jobs phase D has to run.
// (`start`: the frame's first band -- 17 for the CELT layer of a hybrid frame; the bands below it have no words in the record)
OG_DEV u32 pm_setup_jobs(const ParseRec *rec, int C, int B, u32 &fill_lo, u32 &fill_hi, int &dual_end, int start = 0) {
    PmLds &P = PM();
    OG_SYNC();
    OG_FOR_LANES(bin, PM_GROUPS) { // (tables: the search they replace was up to 21 dependent loads per lane)
        P.binband[bin] = rom_bin2band[bin];
        P.binoff[bin] = rom_binoff[bin];
    }
    OG_FOR_LANES(i, 2 * NBANDS) {
        P.jdesc[i] = 0;
        P.jcm[i] = 0;
    }
    OG_SYNC();
    u32 tfm = 0, flo = 0, fhi = 0, de = 0;
    const int logBf = ilog2(B);
    OG_FOR_LANES(l, 2 * NBANDS) {
        const int band = l >> 1, jb = l & 1, coded = band >= start;
        const u32 *wp = rec->words + (coded ? rec->band_w[band] : 0);
        const u32 w0 = coded ? wp[0] : 0u, w1 = coded ? wp[1] : 0u, w2 = coded ? wp[2] : 0u, w3 = coded ? wp[3] : 0u, jw0 = coded ? wp[4] : 0u;
        const int N = (int)(w1 >> 22) & 255;
        const int stereo = (w0 & BW_STEREO) != 0, dual = (w0 & BW_DUAL) != 0, mid_first = (w0 & BW_MID_FIRST) != 0;
        const int njobs = !coded ? 0 : (stereo || dual) ? 2 : 1;
        if (jb == 0) {
            P.bw0[band] = w0;
            P.bw1[band] = w1;
            P.bw2[band] = w2;
            P.scale[band] = (i32)(i16)(w3 & 0xffff);
            if (w0 & BW_DUAL_END) de |= 1u << band;
        }
        const int exists = jb < njobs;
        const int ch = dual ? jb : stereo ? (((jb == 0) == mid_first) ? 0 : 1) : 0;
        const int jpos = (coded ? rec->band_w[band] : 0) + 4 + (jb ? 1 + 2 * (int)(jw0 & 31) : 0);
        u32 jw = jw0;
        if (jb && exists) jw = rec->words[OG_MIN(jpos, REC_WORDS_CAP - 1)];
        const int n_fill = (int)(jw & 31), n_pvq = (int)(jw >> JW_NPVQ_SHIFT) & 31;
        P.jaux[l] = (u32)jpos | (u32)ch << 16 | (u32)exists << 17 | (u32)(exists && n_fill > 0) << 18 | (u32)(n_pvq > 0) << 19;
        if (exists) {
            // the job's time-frequency bookkeeping (quant_band celt.cpp:1548-1580), as in recon_band_mono
            int tf_change = (int)((w0 >> BW_TF_SHIFT) & 7) - 4;
            int recombine = tf_change > 0 ? tf_change : 0, time_divide = 0;
            int logB = logBf - recombine, N_B = (N >> logBf) << recombine;
            while ((N_B & 1) == 0 && tf_change < 0) {
                logB++;
                N_B >>= 1;
                time_divide++;
                tf_change++;
            }
            const int logB0 = logB;
            u32 jd = JD_VALID;
            if (logB0 > 0) jd |= (u32)(logB0 + recombine) << JD_PERM_SHIFT | (B == 1 ? JD_HAD : 0);
            int step = 0;
            for (int k = 0; k < time_divide; k++, step++) jd |= (u32)(logB0 - 1 - k + 1) << (JD_STEP_SHIFT + 4 * step);
            for (int k = 0; k < recombine; k++, step++) jd |= (u32)(k + 1) << (JD_STEP_SHIFT + 4 * step);
            if (n_fill > 0) {
                jd |= JD_FILL;
                if (l < 32) flo |= 1u << l; else fhi |= 1u << (l - 32);
            } else {
                if (logB0 > 0) tfm |= 1u;
                for (int k = 0; k < 3; k++) {
                    const int c = (int)(jd >> (JD_STEP_SHIFT + 4 * k)) & 15;
                    if (c) tfm |= 1u << (1 + 4 * k + (c - 1));
                }
                // the collapse mask of a job whose leaves all carry pulses: the leaves' masks, then what the way back
                // up does to a mask (celt.cpp:1596-1611)
                u32 cm = n_pvq ? S.job_mask_row()[l] : 0u;
                for (int k = 0; k < time_divide; k++) {
                    logB--;
                    cm |= cm >> (1 << logB);
                }
                for (int k = 0; k < recombine; k++) {
                    const u32 c4 = cm & 0xF; // bit_deinterleave_table celt.cpp:1606
                    cm = ((c4 & 1) * 0x03) | ((c4 >> 1 & 1) * 0x0C) | ((c4 >> 2 & 1) * 0x30) | ((c4 >> 3 & 1) * 0xC0);
                }
                logB += recombine;
                P.jcm[2 * band + ch] = (u16)(cm & ((1u << (1 << logB)) - 1));
            }
            P.jdesc[2 * band + ch] = jd;
        }
    }
    tfm = wave_or(tfm);
    fill_lo = wave_or(flo);
    fill_hi = wave_or(fhi);
    de = wave_or(de);
    dual_end = de ? ilog2((i32)de) : NBANDS + 1; // (at most one band ends dual stereo)
    OG_SYNC();
    return tfm;
}

// C: undo the time-frequency changes of every job without fill leaves, a lane per group of 8 coefficients.
OG_DEV void pm_tf_undo(u32 tfm) {
    constexpr int NG = (2 * PM_GROUPS + OG_NLANES - 1) / OG_NLANES;
    if (tfm & 1u) { // interleave_hadamard celt.cpp:1183 as a gather: all groups read, then all write
        V8 hold[NG];
        u8 mark[NG];
#pragma unroll
        for (int it = 0; it < NG; it++) {
            const int g = OG_LANE + it * OG_NLANES;
            mark[it] = 0;
            if (g < 2 * PM_GROUPS) {
                const PmGrp q = pm_group(g);
                const int ls = (int)(q.jd >> JD_PERM_SHIFT) & 7;
                if ((q.jd & JD_VALID) && !(q.jd & JD_FILL) && ls) {
                    const int stride = 1 << ls, n0 = q.N >> ls, had = (q.jd & JD_HAD) != 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int inter = 8 * q.gj + k, j = inter >> ls, i = inter & (stride - 1);
                        hold[it].v[k] = S.v[q.x + (had ? ordery(stride, i) : i) * n0 + j];
                    }
                    mark[it] = 1;
                }
            }
        }
        OG_SYNC();
#pragma unroll
        for (int it = 0; it < NG; it++) {
            const int g = OG_LANE + it * OG_NLANES;
            if (mark[it]) {
                const PmGrp q = pm_group(g);
                st8(q.x + 8 * q.gj, hold[it]);
            }
        }
        OG_SYNC();
    }
    if (tfm & (1u << (1 + 3))) { // a first Haar step of stride 8 (short blocks divided once more): pairs of groups
#pragma unroll
        for (int it = 0; it < NG; it++) {
            const int g = OG_LANE + it * OG_NLANES;
            if (g < 2 * PM_GROUPS) {
                const PmGrp q = pm_group(g);
                if ((q.jd & JD_VALID) && !(q.jd & JD_FILL) && ((q.jd >> JD_STEP_SHIFT) & 15) == 4 && !(q.gj & 1)) {
                    V8 a = ld8(q.x + 8 * q.gj), b = ld8(q.x + 8 * q.gj + 8);
#pragma unroll
                    for (int k = 0; k < 8; k++) haar_pair(a.v[k], b.v[k]);
                    st8(q.x + 8 * q.gj, a);
                    st8(q.x + 8 * q.gj + 8, b);
                }
            }
        }
        OG_SYNC();
    }
    if (tfm & 0x0eeeu) { // Haar steps of stride 1 / 2 / 4: inside a group, in registers, up to three in a row
#pragma unroll
        for (int it = 0; it < NG; it++) {
            const int g = OG_LANE + it * OG_NLANES;
            if (g < 2 * PM_GROUPS) {
                const PmGrp q = pm_group(g);
                const u32 steps = (q.jd & JD_VALID) && !(q.jd & JD_FILL) ? (q.jd >> JD_STEP_SHIFT) & 0xfffu : 0u;
                if (steps & 0x777u) { // (a step of stride 8 has code 4: bit 3 of its nibble only)
                    V8 r = ld8(q.x + 8 * q.gj);
                    for (int k = 0; k < 3; k++) {
                        const int c = (int)(steps >> (4 * k)) & 15;
                        if (c == 1) haar8<1>(r);
                        else if (c == 2) haar8<2>(r);
                        else if (c == 3) haar8<4>(r);
                    }
                    st8(q.x + 8 * q.gj, r);
                }
            }
        }
        OG_SYNC();
    }
}

// the folding source of job (band i, channel ch) made on demand: `n` entries from position p0 of the folding history as the
// band walk would hold it when band i starts (lowband_out of the earlier bands, celt.cpp:1617; the two channels' histories
// averaged once dual stereo has ended, celt.cpp:1856-1860)
// (p0 counts from the frame's first band, like the folding history of the band walk: norm_offset celt.cpp:1787.  `dup`: the
// second band of a frame that starts above band 0 is wider than the first, and the band walk fills the hole behind the first
// band's history with a copy of its end -- special_hybrid_folding celt.cpp:1743: entries from n1 on repeat the n2 - n1 before n1)
OG_DEV void pm_make_lowband(int dst, int p0, int n, int i, int use_y, int dual_end, int norm_offset = 0, int dup_n1 = 0, int dup_back = 0) {
    const PmLds &P = PM();
    OG_SYNC();
    OG_FOR_LANES(j, n) {
        int r = p0 + j;
        const bool copied = dup_back && r >= dup_n1;
        if (copied) r -= dup_back;
        const int p = norm_offset + r, sb = P.binband[p >> 3];
        const i32 sc = P.scale[sb];
        i32 v;
        // (the copy is made before dual stereo is switched off at this very band, and that averages only the histories of the
        // bands before it, celt.cpp:1856-1860: the copied entries stay the first channel's)
        if (i >= dual_end && sb < dual_end && !copied)
            v = ((i32)(i16)mul16_q15(sc, S.v[V_X + p]) + (i32)(i16)mul16_q15(sc, S.v[V_X + 960 + p])) >> 1;
        else
            v = mul16_q15(sc, S.v[V_X + (use_y ? 960 : 0) + p]);
        S.v[dst + j] = (i16)v;
    }
    OG_SYNC();
}

// collapse mask of band b, channel c (what the band walk keeps in S.cmask)
OG_DEV u32 pm_band_cm(int b, int c) {
    const PmLds &P = PM();
    const u32 w0 = (u32)OG_UNI(P.bw0[b]);
    if (w0 & BW_STEREO) return (u32)OG_UNI(P.jcm[2 * b]) | (u32)OG_UNI(P.jcm[2 * b + 1]);
    return (u32)OG_UNI(P.jcm[2 * b + ((w0 & BW_DUAL) ? c : 0)]);
}

// D: the jobs with leaves without pulses, in decode order
OG_DEV void pm_fill_jobs(const u32 *words, const LcgTab &lcg, u32 fill_lo, u32 fill_hi, int C, int B, int dual_end, u32 &seed,
                         int start = 0) {
    const int norm_offset = 8 * rom_eband[start];
    const int dup_n1 = 8 * (rom_eband[start + 1] - rom_eband[start]), dup_n2 = 8 * (rom_eband[start + 2] - rom_eband[start + 1]);
    PmLds &P = PM();
    RecCur cur;
    cur.words = words;
    cur.w = 0;
    cur.base = -64;
    cur.leaf = 0;
    for (int l = 0; l < 2 * NBANDS; l++) {
        if (!((l < 32 ? fill_lo >> l : fill_hi >> (l - 32)) & 1u)) continue;
        OG_MARK(5);
        OG_STAT(20, 1);                             // fill jobs
        const int i = l >> 1, jb = l & 1;
        const u32 aux = (u32)OG_UNI(P.jaux[l]), w0 = (u32)OG_UNI(P.bw0[i]), w1 = (u32)OG_UNI(P.bw1[i]);
        const int ch = (int)(aux >> 16) & 1;
        const int eb0 = (int)(w1 >> 11) & 2047, N = (int)(w1 >> 22) & 255;
        const int tf_change = (int)((w0 >> BW_TF_SHIFT) & 7) - 4;
        const int stereo = (w0 & BW_STEREO) != 0, dual = (w0 & BW_DUAL) != 0;
        u32 x_cm, y_cm;
        if (w0 & BW_HAS_LOW) {
            const int fold_end = (int)(w0 >> BW_FOLD1_SHIFT) & 31;
            int fold_i = (int)(w0 >> BW_FOLD0_SHIFT) & 31;
            x_cm = y_cm = 0;
            do {
                x_cm |= pm_band_cm(fold_i, 0);
                y_cm |= pm_band_cm(fold_i, C - 1);
            } while (++fold_i < fold_end);
        } else
            x_cm = y_cm = (1u << B) - 1;
        i32 jfill;
        int want_low;
        if (dual) {
            jfill = (i32)(jb ? y_cm : x_cm);
            want_low = 1;
        } else {
            i32 fill0 = (i32)(x_cm | y_cm);
            if (stereo) {
                if (w0 & BW_THETA0) fill0 &= (1 << B) - 1;
                if (w0 & BW_THETA1) fill0 &= ((1 << B) - 1) << B;
            }
            jfill = ch ? fill0 >> B : fill0; // (channel 1 of a stereo band is the side)
            want_low = !ch;                  // the side never folds (celt.cpp:1709)
        }
        if (jfill == 0 && !(OG_UNI(P.jaux[l]) >> 19 & 1)) { // nothing to fill with and no pulses: the job's spectrum stays zero
            OG_STAT(24, 1);
            continue; // (its mask P.jcm is zero from the set-up, the seed does not move: celt.cpp:1481-1520 under `if (fill)`)
        }
        cur.w = (int)(aux & 0xffff);
        const u32 jw = rec_word(cur);
        int low = -1;
        if ((w0 & BW_HAS_LOW) && want_low && (jw & JW_NEED_LOW)) {
            const int dup = (i == start + 1 && dup_n2 > dup_n1) ? dup_n2 - dup_n1 : 0;
            pm_make_lowband(V_IY, (int)(w1 & 2047), N, i, dual && ch, dual_end, norm_offset, dup_n1, dup);
            low = V_IY;
        }
        const u32 cm = recon_band_mono(cur, jw, lcg, tf_change, seed, V_X + 960 * ch + eb0, N, B, low, -1, 0, -1, jfill, l);
        if (OG_LANE == 0) P.jcm[2 * i + ch] = (u16)cm;
        OG_SYNC();
    }
}

// E: every stereo merge of the frame (stereo_merge celt.cpp:1113, the sign flip of celt.cpp:1731)
OG_DEV void pm_stereo_merge(int C) {
    PmLds &P = PM();
    if (C != 2) return;
    OG_SYNC();
    OG_FOR_LANES(g, PM_GROUPS) { // partial sums of a group of 8
        const int band = P.binband[g];
        if (P.bw0[band] & BW_STEREO) {
            const V8 a = ld8(V_X + 8 * g), b = ld8(V_X + 960 + 8 * g);
            i32 xp = 0, side = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                xp += mul16(b.v[k], a.v[k]);
                side += mul16(b.v[k], b.v[k]);
            }
            pm_part()[g][0] = xp;
            pm_part()[g][1] = side;
        }
    }
    OG_SYNC();
    OG_FOR_LANES(band, NBANDS) { // the band's two gains
        const u32 w0 = P.bw0[band];
        if (w0 & BW_STEREO) {
            const int g0 = rom_eband[band], g1 = rom_eband[band + 1];
            i32 xp = 0, side = 0;
            for (int g = g0; g < g1; g++) {
                xp += pm_part()[g][0];
                side += pm_part()[g][1];
            }
            const i32 mid = (i32)(i16)(P.bw2[band] & 0xffff);
            xp = mul16x32_q15(mid, xp);
            const i32 mid2 = tr16(mid >> 1);
            const i32 El = mul16(mid2, mid2) + side - 2 * xp, Er = mul16(mid2, mid2) + side + 2 * xp;
            i32 mode = 1, lgain = 0, rgain = 0;
            int kl = 0, kr = 0;
            if (Er < 161061 || El < 161061) // QCONST32(6e-4f, 28): the right channel becomes a copy of the left
                mode = 2;
            else {
                kl = ilog2(El) >> 1;
                kr = ilog2(Er) >> 1;
                lgain = rsqrt_norm(vshr32(El, (kl - 7) << 1));
                rgain = rsqrt_norm(vshr32(Er, (kr - 7) << 1));
                if (kl < 7) kl = 7;
                if (kr < 7) kr = 7;
            }
            if (w0 & BW_INV) mode |= 4;
            P.mpar[band][0] = (u32)(mode | kl << 8 | kr << 16);
            P.mpar[band][1] = ((u32)lgain & 0xffffu) | (u32)rgain << 16;
        } else
            P.mpar[band][0] = 0u;
    }
    OG_SYNC();
    OG_FOR_LANES(g, PM_GROUPS) {
        const int band = P.binband[g];
        const i32 m0 = (i32)P.mpar[band][0];
        if (m0 & 3) {
            const u32 m1 = P.mpar[band][1];
            const i32 lgain = (i32)(i16)(m1 & 0xffff), rgain = (i32)m1 >> 16, mid = (i32)(i16)(P.bw2[band] & 0xffff);
            const int kl = (m0 >> 8) & 255, kr = (m0 >> 16) & 255;
            V8 a = ld8(V_X + 8 * g), b = ld8(V_X + 960 + 8 * g);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                i32 xo, yo;
                if (m0 & 2) {
                    xo = a.v[k];
                    yo = a.v[k];
                } else {
                    const i32 l = tr16(mul16_p15(mid, a.v[k])), r = b.v[k];
                    xo = tr16(pshr32(mul16(lgain, sub16(l, r)), kl + 1));
                    yo = tr16(pshr32(mul16(rgain, add16(l, r)), kr + 1));
                }
                a.v[k] = xo;
                b.v[k] = (m0 & 4) ? tr16(-yo) : yo;
            }
            if (!(m0 & 2)) st8(V_X + 8 * g, a);
            st8(V_X + 960 + 8 * g, b);
        }
    }
    OG_SYNC();
}

OG_DEV void recon_all_bands_pm(const ParseRec *rec, const LcgTab &lcg, int C, int shortBlocks, u32 &seed_io, int start = 0) {
    const int B = shortBlocks ? 8 : 1;
    u32 fill_lo, fill_hi, seed = seed_io;
    int dual_end;
    OG_MARK(3);
    const u32 tfm = pm_setup_jobs(rec, C, B, fill_lo, fill_hi, dual_end, start);
    OG_STAT(0, 1);                                  // frames
    OG_STAT(19, shortBlocks != 0);                  // transient frames
    OG_STAT(21, tfm != 0);                          // frames with a time-frequency change to undo in the parallel pass
    OG_STAT(22, (tfm & 1u) != 0);                   // ... with an interleave among them
    OG_STAT(23, (fill_lo | fill_hi) != 0);          // frames with fill jobs
    OG_MARK(8);
    if (tfm) pm_tf_undo(tfm);
    if (fill_lo | fill_hi) pm_fill_jobs(rec->words, lcg, fill_lo, fill_hi, C, B, dual_end, seed, start);
    OG_MARK(10);
    pm_stereo_merge(C);
    OG_MARK(4);
    OG_FOR_LANES(t, NBANDS * C) { // F: the bands' collapse masks where anti-collapse looks for them
        const PmLds &P = PM();
        const int b = t / C, c = t - b * C;
        const u32 w0 = P.bw0[b];
        const u32 cm = (w0 & BW_STEREO) ? (u32)P.jcm[2 * b] | (u32)P.jcm[2 * b + 1] : (u32)P.jcm[2 * b + ((w0 & BW_DUAL) ? c : 0)];
        S.cmask_row()[t] = (u8)cm;
    }
    OG_SYNC();
    seed_io = seed;
}

// One CELT-only frame, vector half + synthesis + stream bookkeeping (decode_frame_wave's CELT branch).
// Returns the frame's result code (wave-uniform).  The comb-filtered output goes to the stream's history ring; the
// last, strictly serial step -- de-emphasis to int16 PCM -- is celt_post_lane's, one (frame, channel) per lane.
// Which reconstruction kernel takes a frame: 20 ms frames whose record is complete -- CELT-only ones and the CELT half of
// hybrid ones (bands 17 - 20) -- go to the kernel with the 8 KB working set (og_recon.hip, phase-major band loop only),
// everything else -- the 2.5 ms transition frame, records that overflowed -- to the general one.
enum { RECON_ALL = 0, RECON_FAST_ONLY = 1, RECON_REST_ONLY = 2, RECON_NOT_MINE = -1000 };
// What a reconstruction kernel reports per frame, read by the de-emphasis kernel (k_celt_post), which passes `ret` on to the
// caller's result array: the frame's result code and where in the stream's history ring its first sample went.  (The ring
// position is taken from here, not from the stream state: in pipelined steps the next step's reconstruction may already have
// advanced it when this step's de-emphasis runs.)
struct ReconOut {
    i32 ret, pos;
};

// Everything the reconstruction reads of the record's header and of the stream's scalars.  The kernel of 20 ms frames
// (og_recon.hip) fills it from two batched loads at its start -- a dozen dependent round trips to HBM one after the other, each
// followed by its wait, were 17 % of a wave's lifetime in the section profile (profiles/r02/a_celt_recon_sections_5: "outside") --
// the general kernel and the host emulation by plain loads (recon_hdr_load).
struct ReconHdr {
    i32 ret;
    u32 rng_final, flags;
    i32 pf_pitch, pf_gain, pf_tapset, start, n_leaves, n_words, n_coef;
    u32 need_norm;
    i32 channels, prev_mode, frames_decoded;
    u32 rng;
    i32 ring_pos, st_pf_period, st_pf_period_old, st_pf_gain, st_pf_gain_old, st_pf_tapset, st_pf_tapset_old;
};
OG_DEV void recon_hdr_load(const StreamState *st, const ParseRec *rec, ReconHdr &h) {
    h.ret = OG_UNI(rec->ret); h.rng_final = (u32)OG_UNI(rec->rng_final); h.flags = (u32)OG_UNI(rec->flags);
    h.pf_pitch = OG_UNI(rec->pf_pitch); h.pf_gain = OG_UNI(rec->pf_gain); h.pf_tapset = OG_UNI(rec->pf_tapset);
    h.start = OG_UNI(rec->start); h.n_leaves = OG_UNI(rec->n_leaves); h.n_words = OG_UNI(rec->n_words);
    h.need_norm = (u32)OG_UNI(rec->need_norm); h.n_coef = OG_UNI(rec->n_coef);
    h.channels = OG_UNI(st->channels); h.prev_mode = OG_UNI(st->prev_mode); h.frames_decoded = OG_UNI(st->frames_decoded);
    const CeltState *cs = &st->celt;
    h.rng = (u32)OG_UNI(cs->rng); h.ring_pos = OG_UNI(cs->ring_pos);
    h.st_pf_period = OG_UNI(cs->pf_period); h.st_pf_period_old = OG_UNI(cs->pf_period_old);
    h.st_pf_gain = OG_UNI(cs->pf_gain); h.st_pf_gain_old = OG_UNI(cs->pf_gain_old);
    h.st_pf_tapset = OG_UNI(cs->pf_tapset); h.st_pf_tapset_old = OG_UNI(cs->pf_tapset_old);
}
#ifndef OG_HOST_EMUL
// The same in ONE vector load (per-lane addresses): lanes 0-15 the record's first 16 words, 16-19 the stream's first four,
// 20-31 the twelve words of CeltState from `deemph` on; then lane reads.  (Layout asserted below.)
static_assert(offsetof(ParseRec, ret) == 0 && offsetof(ParseRec, rng_final) == 4 && offsetof(ParseRec, flags) == 8 && offsetof(ParseRec, pf_pitch) == 16 &&
              offsetof(ParseRec, pf_gain) == 20 && offsetof(ParseRec, pf_tapset) == 24 && offsetof(ParseRec, start) == 28 &&
              offsetof(ParseRec, n_leaves) == 32 && offsetof(ParseRec, n_words) == 36 && offsetof(ParseRec, need_norm) == 40 &&
              offsetof(ParseRec, n_coef) == 44, "record header words");
static_assert(offsetof(StreamState, channels) == 0 && offsetof(StreamState, prev_mode) == 4, "stream header words");
static_assert(offsetof(CeltState, rng) == offsetof(CeltState, deemph) + 8 && offsetof(CeltState, ring_pos) == offsetof(CeltState, deemph) + 12 &&
              offsetof(CeltState, pf_period) == offsetof(CeltState, deemph) + 16 && offsetof(CeltState, pf_tapset_old) == offsetof(CeltState, deemph) + 36,
              "stream scalar words");
OG_DEV i32 recon_hdr_fetch(const StreamState *st, const ParseRec *rec) { // the lane's word of the batch
    const int l = OG_LANE;
    const i32 *p = l < 16 ? reinterpret_cast<const i32 *>(rec) + l
                 : l < 20 ? reinterpret_cast<const i32 *>(st) + (l - 16)
                          : reinterpret_cast<const i32 *>(&st->celt.deemph[0]) + ((l < 32 ? l : 31) - 20);
    return *p;
}
OG_DEV void recon_hdr_unpack(i32 w, ReconHdr &h) {
#define OG_HW(lane) __builtin_amdgcn_readlane(w, lane)
    h.ret = OG_HW(0); h.rng_final = (u32)OG_HW(1); h.flags = (u32)OG_HW(2); h.pf_pitch = OG_HW(4); h.pf_gain = OG_HW(5); h.pf_tapset = OG_HW(6);
    h.start = OG_HW(7); h.n_leaves = OG_HW(8); h.n_words = OG_HW(9); h.need_norm = (u32)OG_HW(10); h.n_coef = OG_HW(11);
    h.channels = OG_HW(16); h.prev_mode = OG_HW(17); h.frames_decoded = OG_HW(18);
    h.rng = (u32)OG_HW(22); h.ring_pos = OG_HW(23); h.st_pf_period = OG_HW(24); h.st_pf_period_old = OG_HW(25); h.st_pf_gain = OG_HW(26);
    h.st_pf_gain_old = OG_HW(27); h.st_pf_tapset = OG_HW(28); h.st_pf_tapset_old = OG_HW(29);
#undef OG_HW
}
#endif

// The reconstruction of a frame in three stages, so that the middle one -- the PVQ leaves -- can be done for several frames of a
// workgroup at once (og_recon.hip); celt_recon_wave below strings them together for one frame.
//   recon_begin    is the frame this kernel's?  stream reset on a mode change, the spectrum cleared
//   (leaf pass)    every PVQ leaf: index -> pulses -> scaled, de-rotated coefficients + collapse mask (pvq_leaf_lane)
//   recon_finish   band loop, anti-collapse, synthesis, stream bookkeeping; returns the frame's result code
struct ReconCtx {
    ReconHdr h;
    u32 flags, rng_final;
    int ret, mode, C;
    int mode_after = -1; // what the frame leaves as prev_mode when it is not `mode` (desc_mode_after)
    bool leaves; // the frame has a leaf pass and a synthesis (its record is not a BAD_CELT one)
    bool fast;
    bool was_reset = false; // the stream's CELT state was reset at this frame (mode change)
    bool booked = false;    // recon_bookkeeping has run already (og_recon.hip: right behind recon_begin)
    // what recon_finish stages late, fetched early by the caller (per lane: entry `lane` of the record's band energies and
    // pulses, of the stream's two energy histories as they were BEFORE a reset), or not (pre == false: read there)
    bool pre = false;
    i32 pre_bandE, pre_logE1, pre_logE2, pre_pulses;
};
OG_DEV bool recon_fast_eligible(const ReconHdr &h) {
    if (h.flags & (RF_SKIP | RF_BAD_CELT)) return false;
    return ((h.flags >> RF_LM_SHIFT) & 3) == 3 && h.n_words < REC_MAX_WORDS && h.n_leaves <= FAST_MAX_LEAVES;
}
// (rx.h filled by the caller.)  Returns false when the frame is not for this kernel (rx.ret then holds what celt_recon_wave
// returns for it)
OG_DEV bool recon_begin(StreamState *st, const ParseRec *rec, int mode, int ch, int role, ReconCtx &rx) {
    rx.flags = rx.h.flags;
    rx.ret = rx.h.ret;
    rx.mode = mode;
    rx.C = ch;
    rx.leaves = false;
    if (rx.flags & RF_SKIP) {
        if (role == RECON_FAST_ONLY) rx.ret = (int)RECON_NOT_MINE;
        return false;
    }
    rx.fast = recon_fast_eligible(rx.h);
    if ((role == RECON_FAST_ONLY && !rx.fast) || (role == RECON_REST_ONLY && rx.fast)) {
        rx.ret = (int)RECON_NOT_MINE;
        return false;
    }
    const int prev_mode = rx.h.prev_mode;
    if (mode != prev_mode && prev_mode > 0) {
        celt_reset_state(&st->celt); // (and the copies of what it clears)
        rx.was_reset = true;
        rx.h.rng = 0;
        rx.h.st_pf_period = rx.h.st_pf_period_old = rx.h.st_pf_tapset = rx.h.st_pf_tapset_old = 0;
        rx.h.st_pf_gain = rx.h.st_pf_gain_old = 0;
        OG_SYNC();
    }
    rx.rng_final = rx.h.rng_final;
    if (!(rx.flags & RF_BAD_CELT)) {
        rx.leaves = true;
        OG_MARK(1);
        OG_SYNC();
#ifndef OG_RECON_TIGHT
        CeltState *cs = &st->celt;
        OG_FOR_LANES(i, 2 * NBANDS) {
            S.bandE_row()[i] = rec->bandE[i];
            S.logE1_row()[i] = cs->logE1[i];
            S.logE2_row()[i] = cs->logE2[i];
            S.cmask_row()[i] = 0;
        }
        OG_FOR_LANES(i, NBANDS) {
            S.pulses_row()[i] = rec->pulses[i];
            S.tf_res[i] = rec->tf_res[i];
        }
#endif
        OG_FOR_LANES(i, 2 * NBANDS) S.job_mask_row()[i] = 0;
#ifdef OG_HOST_EMUL
        OG_FOR_LANES(i, 2 * 960) S.v[V_X + i] = 0;
#else
        OG_FOR_LANES(i, 2 * 960 / 8) *reinterpret_cast<og_v4i *>(&S.v[V_X + 8 * i]) = og_v4i{0, 0, 0, 0}; // 16 bytes per lane and store
#endif
    }
    return true;
}

// the frame's own leaves, one per lane of its own wave (the general kernel, the host emulation, one frame per workgroup)
// `pre`: the caller fetched leaf `lane`'s three words already (g0, aux0, idx0)
OG_DEV void recon_leaves_own(const ParseRec *rec, const ReconCtx &rx, bool pre = false, u32 g0 = 0, u32 aux0 = 0, u32 idx0 = 0) {
    const int n_leaves = rx.h.n_leaves, spread = (int)(rx.flags >> RF_SPREAD_SHIFT) & 3;
#if defined(OG_HOST_EMUL) && defined(OG_STATS)
    { // the wave pays for its longest leaf: what does that leaf look like?
        int max_n = 0, k_at_max = 0, sum_n = 0;
        for (int t = 0; t < n_leaves; t++) {
            const u32 g = rec->leaf[t].geom;
            const int n = (int)(g >> 11) & 255, k = (int)(g >> 19) & 255;
            sum_n += n;
            if (n > max_n) { max_n = n; k_at_max = k; }
        }
        OG_STAT(40, max_n); OG_STAT(41, k_at_max); OG_STAT(42, sum_n); OG_STAT(44, n_leaves);
        OG_STAT(45, max_n >= 96); OG_STAT(46, max_n >= 144);
    }
#endif
    OG_MARK(2);
#ifdef OG_HOST_EMUL
    OG_FOR_LANES(t, n_leaves) {
        const bool first = pre && t < OG_NLANES;
        const u32 g = first ? g0 : rec->leaf[t].geom;
        const u32 aux = first ? aux0 : rec->leaf[t].aux;
        const u32 idx = first ? idx0 : rec->leaf[t].idx;
        job_mask_or((int)(aux >> 20) & 63, (pvq_leaf_lane(S.v, pvq_lds(), (int)(g >> 11) & 255, (int)(g >> 19) & 255, idx, V_X + (int)(g & 2047),
                                                          (int)(g >> 27) + 1, (i32)(aux & 0xffff), spread)
                                            << ((aux >> 16) & 15)) & 0xffffu);
    }
    OG_SYNC();
#else
    // rounds of 64 leaves: index -> pulses -> scaled, one leaf per lane; then the round's rotations by the whole wave
    for (int t0 = 0; t0 < n_leaves; t0 += OG_NLANES) {
        const int t = t0 + OG_LANE;
        RotJob job;
        job.x = job.blen = job.logB = job.stride2 = 0;
        job.c = job.s = 0;
        job.on = false;
        if (t < n_leaves) {
            const bool first = pre && t0 == 0;
            const u32 g = first ? g0 : rec->leaf[t].geom;
            const u32 aux = first ? aux0 : rec->leaf[t].aux;
            const u32 idx = first ? idx0 : rec->leaf[t].idx;
            job_mask_or((int)(aux >> 20) & 63, (pvq_leaf_lane(S.v, pvq_lds(), (int)(g >> 11) & 255, (int)(g >> 19) & 255, idx, V_X + (int)(g & 2047),
                                                              (int)(g >> 27) + 1, (i32)(aux & 0xffff), spread, &job)
                                                << ((aux >> 16) & 15)) & 0xffffu);
        }
        OG_SYNC();
        OG_MARK(28);
        pvq_rotate_wave(S.v, job, S.rot_marker());
    }
    OG_SYNC();
#endif
}

// What the frame leaves in the stream's header words, and the frame's result code -- all known once recon_begin has run (nothing
// in between reads these words).  The 20 ms kernel calls this THERE: carried to the end of the frame the three values were two
// spilled registers at its 80.
OG_DEV int recon_result(const ReconCtx &rx) { return (rx.leaves && (rx.flags & RF_TELL_OVERFLOW)) ? INTERNAL_ERROR : rx.ret; }
OG_DEV void recon_bookkeeping(StreamState *st, const ReconCtx &rx) {
    if (OG_LANE == 0) {
        st->prev_mode = rx.mode_after >= 0 ? rx.mode_after : rx.mode;
        st->frames_decoded = rx.h.frames_decoded + 1;
        st->range_final = rx.rng_final;
    }
}
OG_DEV int recon_finish(StreamState *st, const ParseRec *rec, const ReconCtx &rx) {
    const u32 flags = rx.flags;
    const int mode = rx.mode, C = rx.C, CC = rx.h.channels;
    int result = rx.ret;
    if (rx.leaves) {
        const int LM = (int)(flags >> RF_LM_SHIFT) & 3, M = 1 << LM, N = M * 120;
        const int transient = (flags & RF_TRANSIENT) != 0, silence = (flags & RF_SILENCE) != 0;
        const int start = rx.h.start, end = NBANDS;
        CeltState *cs = &st->celt;
        LcgTab lcg;
        lcg.init();
#ifndef OG_RECON_TIGHT
        // the phase-major band loop takes every 20 ms frame whose record did not overflow (hybrid: from band 17)
        const bool pm = rx.fast;
        if (!pm) { // the band walk starts from an empty folding history (the PVQ table that was there is no longer needed)
            OG_FOR_LANES(i, 1248) S.v[V_NORM + i] = 0;
            OG_SYNC();
        }
#endif
        u32 seed = rx.h.rng;
#ifdef OG_RECON_TIGHT
        recon_all_bands_pm(rec, lcg, C, transient ? M : 0, seed, start);
        // what anti-collapse and the synthesis read besides the spectrum, staged only now (og_state.hpp, V_LATE: the rows
        // were the band loop's scratch until here; the bands' collapse masks are there already)
        if (rx.pre) {
            if (OG_LANE < 2 * NBANDS) {
                S.bandE_row()[OG_LANE] = (i16)rx.pre_bandE;
                S.logE1_row()[OG_LANE] = (i16)(rx.was_reset ? -28 * 1024 : rx.pre_logE1);
                S.logE2_row()[OG_LANE] = (i16)(rx.was_reset ? -28 * 1024 : rx.pre_logE2);
            }
            if (OG_LANE < NBANDS) S.pulses_row()[OG_LANE] = rx.pre_pulses;
        } else {
            OG_FOR_LANES(i, 2 * NBANDS) {
                S.bandE_row()[i] = rec->bandE[i];
                S.logE1_row()[i] = cs->logE1[i];
                S.logE2_row()[i] = cs->logE2[i];
            }
            OG_FOR_LANES(i, NBANDS) S.pulses_row()[i] = rec->pulses[i];
        }
        OG_SYNC();
#else
        if (pm)
            recon_all_bands_pm(rec, lcg, C, transient ? M : 0, seed, start);
        else
            recon_all_bands(rec->words, rx.h.need_norm, lcg, start, end, C, N, transient ? M : 0, LM, seed);
#endif
        OG_MARK(12);
        if (flags & RF_ANTI_COLLAPSE) anti_collapse_pm(lcg, LM, C, N, start, end, seed);
        if (silence) {
            OG_SYNC();
            OG_FOR_LANES(i, C * NBANDS) S.bandE_row()[i] = (i16)(-28 * 1024);
        }
        OG_TAP(1);
        CeltSynth sp;
        sp.N = N; sp.LM = LM; sp.C = C; sp.CC = CC; sp.start = start; sp.end = end; sp.silence = silence; sp.transient = transient;
        sp.pf_pitch = rx.h.pf_pitch; sp.pf_tapset = rx.h.pf_tapset; sp.pf_gain = rx.h.pf_gain;
        sp.have_state = 1;
        sp.st_pf_period = rx.h.st_pf_period; sp.st_pf_period_old = rx.h.st_pf_period_old; sp.st_pf_gain = rx.h.st_pf_gain;
        sp.st_pf_gain_old = rx.h.st_pf_gain_old; sp.st_pf_tapset = rx.h.st_pf_tapset; sp.st_pf_tapset_old = rx.h.st_pf_tapset_old;
        sp.st_ring_pos = rx.h.ring_pos;
        sp.rng_final = rx.rng_final; sp.rc_error = (flags & RF_RC_ERROR) != 0; sp.inline_deemph = 0;
        sp.loss = nullptr; sp.lost = 0; sp.energies_kept_by_parse = 1;
        OG_MARK(13);
        celt_synthesis(cs, sp);
        OG_MARK(17);
        if (flags & RF_TELL_OVERFLOW) result = INTERNAL_ERROR;
    }
    if (!rx.booked) recon_bookkeeping(st, rx);
    return result; // de-emphasis and PCM: celt_post_lane (k_celt_post), from the history ring
}

OG_DEV int celt_recon_wave(StreamState *st, const ParseRec *rec, int mode, int ch, int role = RECON_ALL, int mode_after = -1) {
    ReconCtx rx;
    rx.mode_after = mode_after;
    recon_hdr_load(st, rec, rx.h);
    if (!recon_begin(st, rec, mode, ch, role, rx)) return rx.ret;
    if (rx.leaves) {
        pvq_tab_load();
        OG_SYNC();
        recon_leaves_own(rec, rx);
    }
    return recon_finish(st, rec, rx);
}

// Third step of the split path for (frame, channel c): runs whenever the frame was synthesised; PCM only on success.
// `silk` (hybrid frames): the SILK half's PCM, added with saturation over the first 960 * ch interleaved entries
// (opus_decode_frame src/opus_decoder.cpp:271-273, Q3).
OG_DEV void celt_post(StreamState *st, const ParseRec *rec, int result, int c, i16 *pcm, const i16 *silk, int ch) {
    if (rec->flags & (RF_SKIP | RF_BAD_CELT)) return;
    celt_post_lane(&st->celt, c, st->channels, 960, (st->celt.ring_pos - 960) & RING_MASK, result >= 0 ? pcm : nullptr, silk, 960 * ch);
}

} // namespace og

#undef OG_SYNC
#define OG_SYNC() OG_FULL_SYNC()
