// og_parse_kernel.hpp -- the CELT parse kernel's body, compiled twice: as k_celt_parse with 32 frames per wave (og_api.hip: in-order
// steps, where the kernel's latency is what the step pays) and as k_celt_parse64 with 64 (og_parse64.hip: pipelined steps, where the
// kernel runs next to the arithmetic kernels and its ISSUE SLOTS are what the step pays -- a wave's instruction stream is nearly the
// same for 64 frames as for 32: 3.45 k -> 2.0 k vector instructions per frame, 20 % -> 33 % of the lanes active).
// Define OG_PARSE_KERNEL_NAME (and OG_PL_LANES, before og_celt_split.hpp) before including this.
#pragma once
#include "og_celt_split.hpp"

using namespace og;

// Split CELT path, first half: ONE FRAME PER LANE.  Lane l < OG_PL_LANES (= 32) of wave w of workgroup g parses frame OG_PL_FRAMES g + OG_PL_LANES w + l (range decoder,
// energies, allocation, band budget logic, PVQ indices) into recs[frame]; no vector work, no cross-lane traffic.
// `which`: PARSE_ALL, or one of the two launches of a pipelined step (opusgpu_set_pipeline): PARSE_CELT_ONLY runs ahead on the
// library's own stream, PARSE_HYBRID_ONLY behind the step's k_silk_parse (it resumes the range decoder that kernel hands off).
enum { PARSE_ALL = 0, PARSE_CELT_ONLY = 1, PARSE_HYBRID_ONLY = 2 };
#ifndef OG_PARSE_WAVES_PER_SIMD
#define OG_PARSE_WAVES_PER_SIMD 2 // (the register budget the compiler works to: 512 / this)
#endif
__global__ void __attribute__((amdgpu_waves_per_eu(OG_PARSE_WAVES_PER_SIMD, 8))) __launch_bounds__(64 * OG_PL_WAVES) OG_PARSE_KERNEL_NAME(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena,
                                                      StreamState *st, ParseRec *recs, int n, int n_streams,
                                                      const SilkHandoff *handoff, int which, int groups, u32 *started) {
    // `groups`: a workgroup parses that many groups of OG_PL_FRAMES frames one after the other (1: the grid covers the step once)
    // `started` (steps queued as a window, opusgpu_decode_steps_device): every workgroup counts itself in when it starts -- the
    // reconstruction of the step before is held (a stream memory wait) until this launch's workgroups have their places
    if (started && threadIdx.x == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool lane_on = (int)(threadIdx.x & 63) < OG_PL_LANES;
    const int f0 = (int)blockIdx.x * groups * OG_PL_FRAMES + OG_PWAVE * OG_PL_LANES + OG_PCOL;
    if (which != PARSE_ALL) { // a launch that finds none of its frames among the workgroup's leaves without loading the tables
        bool mine = false;
        for (int g = 0; g < groups; g++) {
            const int f = f0 + g * OG_PL_FRAMES;
            if (lane_on && f < n) {
                const int m0 = desc_mode(descs[f].flags);
                mine |= which == PARSE_CELT_ONLY ? m0 == MODE_CELT : m0 == MODE_HYBRID;
            }
        }
        if (!__syncthreads_or(mine)) return;
    }
    parse_tables_load();
    if (!lane_on) return;
    for (int g = 0; g < groups; g++) {
        const int f = f0 + g * OG_PL_FRAMES;
        if (f >= n) break;
        const FrameDesc d = descs[f];
        const int mode = desc_mode(d.flags);
        if (d.stream < 0 || d.stream >= n_streams || !(mode == MODE_CELT || (mode == MODE_HYBRID && handoff)) || desc_rfc(d.flags)) continue;
        if ((which == PARSE_CELT_ONLY && mode != MODE_CELT) || (which == PARSE_HYBRID_ONLY && mode != MODE_HYBRID)) continue;
#ifdef OG_PROF_PARSE // profiling builds: time the sections of the parse kernel instead of the recon kernel (full batches only)
        OG_PROF_INIT();
#endif
        celt_parse_lane(&st[d.stream], arena + d.offset, d.len, desc_channels(d.flags), &recs[f], mode == MODE_HYBRID ? &handoff[f] : nullptr);
#ifdef OG_PROF_PARSE
        OG_PROF_FLUSH();
#endif
    }
}

