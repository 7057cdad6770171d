// og_silk_synth_kernel.hpp -- the SILK synthesis kernel's body (SILK-only and hybrid frames on the split path, arithmetic half: one
// frame per wave, no CELT code -- decode_frame_wave<false>), compiled twice with the tight layout of its working set (SilkLds,
// og_silk.hpp: OG_SILK_TIGHT): as k_silk_synth with buffers for 20 ms at 16 kHz (og_silk_synth.hip) and as k_silk_synth_nb with
// buffers for 20 ms at 8 kHz (og_silk_nb.hip: narrowband SILK-only frames).  Translation units of their own because the working
// set is ONE __shared__ object whose layout is chosen at compile time; k_decode_step (og_api.hip) and k_decode_rfc keep the full one.
// Define OG_SSYNTH_KERNEL_NAME, OG_SSYNTH_LAUNCHER, OG_SSYNTH_PROF and OG_SSYNTH_NB_ONLY (1: the kernel of narrowband SILK-only frames;
// 0: every other frame -- and those too unless the launch says they are taken elsewhere) before including this.
#pragma once
#include "og_decode.hpp"

using namespace og;

#ifndef OG_SILK_WAVES
#define OG_SILK_WAVES 2
#endif
#ifdef OG_SSYNTH_MAX_WAVES // at most this many of the kernel's waves per SIMD (the register allocation is raised to the count that says so)
#define OG_SSYNTH_OCC __attribute__((amdgpu_waves_per_eu(2, OG_SSYNTH_MAX_WAVES)))
#else
#define OG_SSYNTH_OCC
#endif
__global__ void OG_SSYNTH_OCC __launch_bounds__(64, OG_SILK_WAVES) OG_SSYNTH_KERNEL_NAME(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena,
                                                                            StreamState *st, i16 *pcm, i32 *result, int n, int n_streams,
                                                                            int pcm_stride, SilkHandoff *handoff, const SilkRec *srecs, int nb_elsewhere) {
    const int f = (int)blockIdx.x;
    if (f >= n) return;
    const FrameDesc d = descs[f];
    int ret;
    u32 prefetched = 0;
    if (d.stream < 0 || d.stream >= n_streams) {
        if (OG_SSYNTH_NB_ONLY) return; // (reported by the kernel of the other frames)
        ret = BAD_ARG;
    } else if (desc_mode(d.flags) == MODE_CELT || desc_rfc(d.flags) ||
               (OG_SSYNTH_NB_ONLY ? !desc_silk_nb_only(d.flags) : (nb_elsewhere && desc_silk_nb_only(d.flags)))) {
        return;
    } else {
#ifdef OG_PROF_SSYNTH // profiling builds: time the sections of the SILK synthesis kernel
        OG_PROF_INIT();
#endif
        // The frame's record (2 KB) and the stream's SILK state (1.7 KB) are read below in a dozen dependent steps, each behind a
        // wave-level sync that keeps the compiler from asking early: every one of them paid a trip to HBM.  One load per lane --
        // lane l touches the l-th 64 bytes of the state, lane 32 + l of the record -- brings all of it to the L2 now; its value is
        // never used (the empty asm at the end keeps the register), the later reads find their lines on the way or there.
        {
            const int l = (int)threadIdx.x;
            const char *p = l < 28 ? reinterpret_cast<const char *>(&st[d.stream].silk) + 64 * l
                          : l < 32 ? reinterpret_cast<const char *>(&handoff[f]) + 16 * (l - 28)
                                   : reinterpret_cast<const char *>(&srecs[f]) + 64 * (l - 32);
            prefetched = *reinterpret_cast<const volatile u32 *>(p);
        }
        ret = decode_frame_wave<false>(&st[d.stream], arena + d.offset, d.len, desc_mode(d.flags), desc_bandwidth(d.flags),
                                       desc_channels(d.flags), pcm + (size_t)f * pcm_stride, &handoff[f], &srecs[f]);
#ifdef OG_PROF_SSYNTH
        OG_PROF_FLUSH();
#endif
        asm volatile("" ::"v"(prefetched));
        if (ret == CONTINUE_SPLIT || ret == CONTINUE_Q4) return;
    }
    if (threadIdx.x == 0) result[f] = ret;
}

extern "C" void OG_SSYNTH_LAUNCHER(hipStream_t s, const void *descs, const void *arena, void *streams, void *pcm, void *result, int n,
                                   int n_streams, int pcm_stride, void *handoff, const void *srecs, int nb_elsewhere) {
    hipLaunchKernelGGL(OG_SSYNTH_KERNEL_NAME, dim3(n), dim3(64), 0, s, (const FrameDesc *)descs, (const u8 *)arena, (StreamState *)streams,
                       (i16 *)pcm, (i32 *)result, n, n_streams, pcm_stride, (SilkHandoff *)handoff, (const SilkRec *)srecs, nb_elsewhere);
}
#ifdef OG_PROF
// profiling builds only: this kernel's section counters (OG_MARK) -- every translation unit has its own copy
extern "C" int OG_SSYNTH_PROF(unsigned long long *out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
