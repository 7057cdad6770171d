// og_recon.hip -- the reconstruction kernel of 20 ms frames (CELT-only ones, and the CELT layer of hybrid ones) with a 6.2 KB LDS working
// set per frame: five granules of LDS and 80 registers, six waves per SIMD.
//
// A translation unit of its own because the working set is ONE __shared__ object (FrameLds, og_state.hpp) whose layout
// is chosen at compile time: here OG_RECON_TIGHT selects the layout without the folding-history rows, the packet buffer and
// the entropy-decoding arrays, with the synthesis buffer starting inside X.  Only the phase-major band loop and the
// synthesis run from this layout; frames it does not take (recon_fast_eligible, og_celt_split.hpp) are left to the general
// kernel k_celt_recon in og_api.hip, which is launched right behind this one and skips the frames done here.
// Why: measured on the 10 KB layout, k_celt_recon's time goes with 1 / (waves per SIMD) -- 2.69 ms at three, 2.09 ms at four; and
// in pipelined steps the kernel shares its CUs' LDS with the parse kernel's workgroups (og_state.hpp).
//
// One frame per workgroup of one wave, the frame's PVQ leaves one per lane.  (Round 3 measured the alternatives and they lost: the
// leaves of two / four frames pooled in one workgroup and dealt out by cost, and a kernel of its own for the leaves: DESIGN.md 6c.)
#define OG_RECON_TIGHT 1
#include <hip/hip_runtime.h>
#include "og_celt_split.hpp"

using namespace og;

#ifndef OG_FAST_WAVES
#define OG_FAST_WAVES 6 // (80 registers; the working set is five LDS granules: og_state.hpp)
#endif


// `hybrid`: the step's hybrid frames come through the split path too (their SILK half by k_silk_parse / k_silk_synth): this
// kernel then also takes their CELT half
// (measured, round 5: capping this kernel at five / four waves per SIMD -- room for the parse kernel's 168-register waves -- costs the
// CELT step 6 % / 15 %, mixed pages 2 % / 9 %; the SILK synthesis of narrowband frames is the kernel such a cap pays for, og_silk_nb.hip.
// A twin of this kernel held to four waves per SIMD for the steps in which it runs next to the SILK synthesis: hybrid-256k 8.94 -> 8.89 ms,
// inside the spread of the runs -- not kept)
__global__ void __launch_bounds__(64, OG_FAST_WAVES) k_celt_recon_fb(const FrameDesc *__restrict__ descs, StreamState *st,
                                                                           const ParseRec *recs, ReconOut *rout, int n, int n_streams,
                                                                           int hybrid, u32 *started) {
    // `started` (steps queued as a window, opusgpu_decode_steps_device): every 64th workgroup counts itself in when it starts --
    // the de-emphasis of the step before is held until the first round of this launch has its places
    if (started && threadIdx.x == 0 && (blockIdx.x & 63) == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int f = (int)blockIdx.x;
    // the wave's frame, if it has one this kernel takes
    StreamState *sp = nullptr;
    const ParseRec *rec = nullptr;
    ReconCtx rx;
    rx.leaves = false;
    rx.flags = 0;
    bool mine = false;
    int pos = 0;
    u32 lg = 0, la = 0, li = 0;
    OG_PROF_INIT();
    if (f < n) {
        // Round trip 1, everything that hangs off the frame's index: its descriptor, the record's header words, the lane's leaf
        // and band-energy entries.  Round trip 2, what hangs off the stream's index: the stream's scalars and energy histories.
        // (The loads are written before anything waits on one of them: the wave pays two memory latencies here, not fifteen.)
        rec = &recs[f];
        const FrameDesc d = descs[f];
        const int lane = OG_LANE;
        const i32 w_rec = reinterpret_cast<const i32 *>(rec)[lane < 16 ? lane : 15];
        {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 lf = *reinterpret_cast<const u32x4 *>(&rec->leaf[lane]);
            li = lf[0];
            lg = lf[1];
            la = lf[2];
        }
        rx.pre_bandE = rec->bandE[lane < 2 * NBANDS ? lane : 0];
        rx.pre_pulses = rec->pulses[lane < NBANDS ? lane : 0];
        const int mode = desc_mode(d.flags);
        bool ok = !(d.stream < 0 || d.stream >= n_streams || !(mode == MODE_CELT || (mode == MODE_HYBRID && hybrid)) || desc_rfc(d.flags));
        if (ok) {
            sp = &st[d.stream];
            const i32 w_st = recon_hdr_fetch(sp, rec); // (lanes 0-15 re-read the record's words: one instruction for all)
            rx.pre_logE1 = sp->celt.logE1[lane < 2 * NBANDS ? lane : 0];
            rx.pre_logE2 = sp->celt.logE2[lane < 2 * NBANDS ? lane : 0];
            rx.pre = true;
            recon_hdr_unpack(lane < 16 ? w_rec : w_st, rx.h);
            if (mode == MODE_HYBRID && (rx.h.flags & RF_SKIP)) ok = false; // the single-kernel path already reported this frame
        }
        if (ok) {
            pos = rx.h.ring_pos; // where the frame's first sample goes
            mine = recon_begin(sp, rec, mode, desc_channels(d.flags), RECON_FAST_ONLY, rx);
            if (mine) { // (the frame's result and what it leaves in the stream's header words: known here, see recon_bookkeeping)
                if (OG_LANE == 0) rout[f] = ReconOut{recon_result(rx), pos};
                recon_bookkeeping(sp, rx);
                rx.booked = true;
            }
        }
    }
    if (mine && rx.leaves) {
        pvq_tab_load();
        OG_SYNC();
        recon_leaves_own(rec, rx, true, lg, la, li);
    }
    if (mine) recon_finish(sp, rec, rx);
    OG_PROF_FLUSH();
}

extern "C" void og_launch_celt_recon_fb(hipStream_t s, const void *descs, void *streams, const void *recs, void *rout, int n,
                                        int n_streams, int hybrid, unsigned *started) {
    hipLaunchKernelGGL(k_celt_recon_fb, dim3(n), dim3(64), 0, s, (const FrameDesc *)descs, (StreamState *)streams,
                       (const ParseRec *)recs, (ReconOut *)rout, n, n_streams, hybrid, started);
}
extern "C" int og_celt_recon_fb_signals(int n) { return (n + 63) / 64; }

#ifdef OG_PROF
// profiling builds only: this kernel's section counters (OG_MARK) -- the other translation unit has its own copy
extern "C" int og_recon_fb_prof(unsigned long long *out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
