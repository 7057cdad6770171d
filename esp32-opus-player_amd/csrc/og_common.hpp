// og_common.hpp -- shared definitions of the MI355X Opus decode kernels (gfx950, wave64).
//
// Execution model: ONE 20 ms FRAME PER WAVEFRONT.  A workgroup is exactly one wave (64 lanes); its
// frame's packet bytes, normalised band vectors, folding history and synthesis buffer are staged
// in LDS (struct FrameLds).  Entropy decoding is inherently serial: those parts are written as
// plain scalar code that every lane executes identically (the compiler keeps wave-uniform values
// in SGPRs / scalar ALU where it can); vector parts are written with OG_FOR_LANES so that the 64
// lanes split the band / butterfly / sample loops.
//
// The same headers compile in a TEST-ONLY host emulation (OG_HOST_EMUL, used by tests/emul) where
// a "wave" has a single lane; lane loops then run sequentially.  That build exists so the kernel
// source can be fuzzed and run under ASan/UBSan on a CPU; it is not linked into the shipped
// library and is not a fallback: the C-ABI fails with OPUSGPU_ERR_NO_DEVICE when HIP is unusable.
#pragma once
#include <stdint.h>

typedef int8_t i8;
typedef int16_t i16;
typedef int32_t i32;
typedef int64_t i64;
typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#ifdef OG_HOST_EMUL
#include <string.h>
#define OG_DEV static inline
#define OG_DEVN static
#define OG_MEMBER inline
#define OG_LANE 0
#define OG_NLANES 1
#define OG_SYNC() ((void)0)
#define OG_FULL_SYNC() ((void)0)
#define OG_LSYNC() ((void)0)
#define OG_LDS static
#define OPUS_ROM static const
#define OG_CLZ(x) __builtin_clz(x)
extern "C" void og_emul_tap(int id); // stage taps for parity tests (host emulation only)
#define OG_TAP(id) og_emul_tap(id)
#else
#include <hip/hip_runtime.h>
#define OG_DEV static __device__ __forceinline__
#define OG_DEVN static __device__ __noinline__
#define OG_MEMBER __device__ __forceinline__
// A workgroup is one wave, except in the translation units that say otherwise before including this header (og_recon.hip:
// several frames per workgroup, one per wave): there OG_LANE is the lane within the wave and OG_WAVE the wave's index.
#ifndef OG_LANE
#define OG_LANE ((int)threadIdx.x)
#endif
#ifndef OG_WAVE
#define OG_WAVE 0
#endif
#define OG_NLANES 64
#define OG_FULL_SYNC() __syncthreads()
// LDS-only ordering between lanes of ONE wave: the LDS unit serves a wave's instructions in order, so only the
// compiler has to be kept from moving LDS accesses across the point (no s_barrier, no s_waitcnt).
#define OG_LSYNC() __syncthreads()
// Every workgroup of this library is ONE wave, so a sync point only has to order that wave's own memory operations:
// LDS instructions of a wave execute in order and LLVM's AMDGPU memory model needs no code for wavefront-scope fences.
// OG_WAVE_SYNC: the compiler may not move memory accesses across the point; no s_waitcnt, no s_barrier.
#define OG_WAVE_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
// (Measured in round 2 with every OG_SYNC wave-scoped: GPU parity suite green, every kernel time unchanged to three digits --
// the waits are forced by data dependencies anyway.  The full barrier is what OG_SYNC is.)
#define OG_SYNC() OG_FULL_SYNC()
#define OG_LDS __shared__
#define OPUS_ROM static __device__ const
#define OG_CLZ(x) __clz(x)
#define OG_TAP(id) ((void)0)
#endif

#define OG_FOR_LANES(i, n) for (int i = OG_LANE; i < (n); i += OG_NLANES)

// Section timers for profiling builds (-DOG_PROF, never in the shipped library): OG_MARK(id) closes the running section
// and opens section `id`; per-wave cycle totals accumulate in LDS and are added to g_prof[] when the kernel ends.
#if defined(OG_PROF) && !defined(OG_HOST_EMUL)
__device__ unsigned long long g_prof[64];
__shared__ unsigned int s_prof[64];
__shared__ unsigned long long s_prof_t;
__shared__ int s_prof_cur;
#define OG_PROF_INIT()                                                  \
    do {                                                                \
        if (threadIdx.x < 32) s_prof[threadIdx.x] = s_prof[threadIdx.x + 32] = 0; /* (kernels with 32 active lanes) */ \
        if (threadIdx.x == 0) {                                         \
            s_prof_cur = 0;                                             \
            s_prof_t = __builtin_amdgcn_s_memtime();                    \
        }                                                               \
        __syncthreads();                                                \
    } while (0)
#define OG_MARK(id)                                                     \
    do {                                                                \
        if (threadIdx.x == 0) {                                         \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
            s_prof[s_prof_cur] += (unsigned int)(t_ - s_prof_t);        \
            s_prof_cur = (id);                                          \
            s_prof_t = t_;                                              \
        }                                                               \
    } while (0)
#define OG_PROF_FLUSH()                                                 \
    do {                                                                \
        OG_MARK(0);                                                     \
        __syncthreads();                                                \
        if (threadIdx.x < 32) {                                         \
            atomicAdd(&g_prof[threadIdx.x], (unsigned long long)s_prof[threadIdx.x]); \
            atomicAdd(&g_prof[threadIdx.x + 32], (unsigned long long)s_prof[threadIdx.x + 32]); \
        }                                                               \
    } while (0)
#else
#define OG_PROF_INIT() ((void)0)
#define OG_MARK(id) ((void)0)
#define OG_PROF_FLUSH() ((void)0)
#endif

// Event counters of the host emulation (tests/emul, -DOG_STATS): how often the band loop takes each of its paths on a
// given workload.  Nothing in any other build.
#if defined(OG_HOST_EMUL) && defined(OG_STATS)
extern long long og_stats[64];
#define OG_STAT(id, n) (og_stats[id] += (n))
#else
#define OG_STAT(id, n) ((void)0)
#endif

#define OG_MIN(a, b) ((a) < (b) ? (a) : (b))
#define OG_MAX(a, b) ((a) > (b) ? (a) : (b))

namespace og {

// ---- error / mode codes (values follow the reference's opus_decoder.h / celt.h) -----------------
enum { OK = 0, BAD_ARG = -1, BUFFER_TOO_SMALL = -2, INTERNAL_ERROR = -3, INVALID_PACKET = -4,
       CELT_BAD_ARG = -18 }; // ERR_OPUS_CELT_BAD_ARG (src/opus_decoder.h:55): celt_decode_with_ec's refusals, src/celt.cpp:2211-2225
enum { MODE_SILK = 1000, MODE_HYBRID = 1001, MODE_CELT = 1002 };
enum { BW_NB = 1101, BW_MB = 1102, BW_WB = 1103, BW_SWB = 1104, BW_FB = 1105 };

// ---- fixed-point primitives; each names the reference macro it reproduces (src/celt.h) ----------
OG_DEV i32 mul16(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }                 // MULT16_16 :338
OG_DEV i32 mul16_q15(i32 a, i32 b) { return mul16(a, b) >> 15; }                     // :355
OG_DEV i32 mul16_q14(i32 a, i32 b) { return mul16(a, b) >> 14; }                     // :354
OG_DEV i32 mul16_p15(i32 a, i32 b) { return (16384 + mul16(a, b)) >> 15; }           // :359
#ifdef OG_HOST_EMUL
OG_DEV i32 mul16x32_q15(i32 a, i32 b) { return (i32)(((i64)(i16)a * (i64)b) >> 15); } // MULT16_32_Q15 :263
#else
// The 64-bit product in one instruction.  Measured on gfx950 (profiles/r03/a_valu_issue_rates.txt): v_mad_i64_i32 issues at the
// rate of a 24-bit multiply -- 32-bit multiplies are not quarter rate here -- so product + v_alignbit_b32 is two instructions
// against four.  Written as inline assembly because the compiler, given the C expression above in these kernels, expands the
// sign-extended operands into mul_lo / mul_hi / mad_u64 sequences instead.
OG_DEV i32 mul16x32_q15(i32 a, i32 b) { // MULT16_32_Q15 :263
    i64 p;
    u64 carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"((i32)(i16)a), "v"(b));
    return (i32)(p >> 15);
}
#endif
OG_DEV i32 mul32_q31(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 31); }        // :266
OG_DEV i32 shl32(i32 a, int s) { return (i32)((u32)a << s); }                        // :292
OG_DEV i32 pshr32(i32 a, int s) { return (a + ((1 << s) >> 1)) >> s; }               // :295
OG_DEV i32 vshr32(i32 a, int s) { return s > 0 ? a >> s : shl32(a, -s); }            // :297
OG_DEV i32 addw(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }                     // ADD32_ovflw :326
OG_DEV i32 subw(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }                     // SUB32_ovflw :328
OG_DEV i32 negw(i32 a) { return (i32)(0u - (u32)a); }                                // NEG32_ovflw :331
OG_DEV i32 clampsym(i32 x, i32 a) { return x > a ? a : (x < -a ? -a : x); }          // SATURATE :303
OG_DEV i32 sat16(i32 x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }    // SAT16 :401
OG_DEV i32 add16(i32 a, i32 b) { return (i32)(i16)((i16)a + (i16)b); }               // ADD16 :317
OG_DEV i32 sub16(i32 a, i32 b) { return (i32)(i16)a - (i32)(i16)b; }                 // SUB16 :319
OG_DEV i32 shl16(i32 a, int s) { return (i32)(i16)((u16)a << s); }                   // SHL16 :288
OG_DEV i32 tr16(i32 a) { return (i32)(i16)a; }                                       // EXTRACT16 :281
OG_DEV i32 frac_mul16(i32 a, i32 b) { return (16384 + (i32)(i16)a * (i32)(i16)b) >> 15; } // :378
OG_DEV int ilog(u32 x) { return x ? 32 - OG_CLZ(x) : 0; }                            // EC_ILOG :250
OG_DEV int ilog2(i32 x) { return ilog((u32)x) - 1; }                                 // celt_ilog2 :469
OG_DEV u32 udiv(u32 n, u32 d) { return n / d; }                                      // celt_udiv :405

static constexpr i32 SIG_SAT = 300000000; // celt.h:234

// ---- SILK flavour (src/silk.h) -------------------------------------------------------------------
#ifdef OG_HOST_EMUL
OG_DEV i32 smulwb(i32 a, i32 b) { return (i32)(((i64)a * (i64)(i16)b) >> 16); }      // silk_SMULWB :447
#else
// (a * (i16)b) >> 16 = the HIGH word of a * (b << 16): one v_mul_hi_i32 (32-bit multiplies issue like 24-bit ones on gfx950,
// profiles/r03/a_valu_issue_rates.txt) plus a shift that leaves the loop wherever b does not change -- where round 3 had two
// 24-bit multiplies, two shifts / masks and an add
OG_DEV i32 smulwb(i32 a, i32 b) {                                                    // silk_SMULWB :447
    return __mulhi(a, (i32)((u32)(i32)(i16)b << 16));
}
#endif
OG_DEV i32 smlawb(i32 acc, i32 a, i32 b) { return addw(acc, smulwb(a, b)); }         // silk_SMLAWB :450
OG_DEV i32 smulww(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 16); }           // silk_SMULWW :474
OG_DEV i32 smulbb(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }                // silk_SMULBB :459
OG_DEV i32 smlabb(i32 acc, i32 a, i32 b) { return addw(acc, smulbb(a, b)); }         // silk_SMLABB :462
OG_DEV i32 smmul(i32 a, i32 b) { return (i32)(((i64)a * (i64)b) >> 32); }            // silk_SMMUL :512
OG_DEV i32 rshift_round(i32 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; } // :156
OG_DEV i64 rshift_round64(i64 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; }
OG_DEV i32 add_sat32(i32 a, i32 b) {                                                 // silk_ADD_SAT32 :480
    i64 s = (i64)a + b;
    return s > 2147483647LL ? 2147483647 : (s < -2147483648LL ? (i32)(-2147483647 - 1) : (i32)s);
}
OG_DEV i32 sub_sat32(i32 a, i32 b) {
    i64 s = (i64)a - b;
    return s > 2147483647LL ? 2147483647 : (s < -2147483648LL ? (i32)(-2147483647 - 1) : (i32)s);
}
OG_DEV i32 limit32(i32 a, i32 l1, i32 l2) {                                          // silk_LIMIT :427
    return l1 > l2 ? (a > l1 ? l1 : (a < l2 ? l2 : a)) : (a > l2 ? l2 : (a < l1 ? l1 : a));
}
OG_DEV i32 lshift_sat32(i32 a, int s) {                                              // silk_LSHIFT_SAT32 :139
    return shl32(limit32(a, (i32)(-2147483647 - 1) >> s, 2147483647 >> s), s);
}
OG_DEV int clz32(i32 x) { return x ? OG_CLZ((u32)x) : 32; }                          // silk_CLZ32 :492

// ---- wave-level helpers ---------------------------------------------------------------------------
// Sum / OR over the 64 lanes, result in every lane.  (Host emulation: one lane, identity.)
// Must be called with all 64 lanes active.  Four DPP row rotations (ror 8/4/2/1) leave the sum of each
// 16-lane row in all of its lanes; the four row sums are then combined on the scalar unit, so the result
// is a wave-uniform (SGPR) value.
#ifndef OG_HOST_EMUL
#define OG_DPP_ROR(v, n) __builtin_amdgcn_update_dpp(0, (int)(v), 0x120 + (n), 0xf, 0xf, false)
#define OG_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#else
#define OG_UNI(x) ((int)(x))
#endif
OG_DEV i32 wave_sum(i32 v) {
#ifndef OG_HOST_EMUL
    v += OG_DPP_ROR(v, 8);
    v += OG_DPP_ROR(v, 4);
    v += OG_DPP_ROR(v, 2);
    v += OG_DPP_ROR(v, 1);
    v = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
        __builtin_amdgcn_readlane(v, 48);
#endif
    return v;
}
OG_DEV u32 wave_or(u32 v) {
#ifndef OG_HOST_EMUL
    v |= (u32)OG_DPP_ROR(v, 8);
    v |= (u32)OG_DPP_ROR(v, 4);
    v |= (u32)OG_DPP_ROR(v, 2);
    v |= (u32)OG_DPP_ROR(v, 1);
    v = (u32)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
              __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
#endif
    return v;
}

#ifndef OG_HOST_EMUL
// Inclusive prefix sum / prefix maximum over the 64 lanes (all active; values >= 0 for the maximum): Hillis-Steele inside the
// 16-lane rows with DPP row shifts (a lane without a source reads 0), then the rows' last lanes carried on with the two row
// broadcasts (lane 15 of a row into the next row; lane 31 into the upper half).
#define OG_DPP_SHR0(v, n) __builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (n), 0xf, 0xf, true)
OG_DEV i32 wave_scan_add(i32 v) {
    v += OG_DPP_SHR0(v, 1);
    v += OG_DPP_SHR0(v, 2);
    v += OG_DPP_SHR0(v, 4);
    v += OG_DPP_SHR0(v, 8);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}
OG_DEV i32 wave_scan_max(i32 v) {
    i32 t;
    t = OG_DPP_SHR0(v, 1);
    v = t > v ? t : v;
    t = OG_DPP_SHR0(v, 2);
    v = t > v ? t : v;
    t = OG_DPP_SHR0(v, 4);
    v = t > v ? t : v;
    t = OG_DPP_SHR0(v, 8);
    v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v = t > v ? t : v;
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    v = t > v ? t : v;
    return v;
}
#endif

} // namespace og
