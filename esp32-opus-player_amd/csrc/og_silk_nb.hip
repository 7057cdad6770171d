// og_silk_nb.hip -- k_silk_synth_nb: the SILK synthesis kernel (k_silk_synth, og_api.hip) for NARROWBAND SILK-only frames.
//
// A translation unit of its own because the kernel's working set is ONE __shared__ object (SilkLds, og_silk.hpp) whose buffers are
// sized at compile time: here for a 20 ms frame at 8 kHz (OG_SILK_LDS_FRAME = 160), 6.7 KB instead of the 10.2 KB that a frame at
// 16 kHz needs -- six LDS granules instead of eight, so five of these frames fit a SIMD (what the kernel's 84 registers allow)
// where four of the wideband kernel's do.  The synthesis waits on its frames' serial chains (LPC recurrence, all-pass
// up-sampler): frames in flight are what it is short of (one granule MORE cost the SILK-NB step 5 %).  The loops over
// a subframe's samples are three per lane instead of five, too.
// Same code otherwise (decode_frame_wave<false>).  A step with SILK-only frames launches this kernel and then k_silk_synth, which
// leaves the narrowband SILK-only frames alone (`nb_elsewhere`); either kernel's workgroups that find a frame of the other's return
// at once.
#include <hip/hip_runtime.h>
#define OG_SILK_LDS_FRAME 160
#include "og_decode.hpp"

using namespace og;

#ifndef OG_SILK_NB_WAVES
#define OG_SILK_NB_WAVES 2
#endif
__global__ void __launch_bounds__(64, OG_SILK_NB_WAVES) k_silk_synth_nb(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena,
                                                                         StreamState *st, i16 *pcm, i32 *result, int n, int n_streams,
                                                                         int pcm_stride, SilkHandoff *handoff, const SilkRec *srecs) {
    const int f = (int)blockIdx.x;
    if (f >= n) return;
    const FrameDesc d = descs[f];
    // (a stream index out of range is k_silk_synth's to report)
    if (d.stream < 0 || d.stream >= n_streams || desc_rfc(d.flags) || !desc_silk_nb_only(d.flags)) return;
    u32 prefetched;
    { // (as in k_silk_synth: the stream's SILK state and the frame's record on their way to the L2 before the first dependent read)
        const int l = (int)threadIdx.x;
        const char *p = l < 28 ? reinterpret_cast<const char *>(&st[d.stream].silk) + 64 * l
                      : l < 32 ? reinterpret_cast<const char *>(&handoff[f]) + 16 * (l - 28)
                               : reinterpret_cast<const char *>(&srecs[f]) + 64 * (l - 32);
        prefetched = *reinterpret_cast<const volatile u32 *>(p);
    }
    const int ret = decode_frame_wave<false>(&st[d.stream], arena + d.offset, d.len, desc_mode(d.flags), desc_bandwidth(d.flags),
                                             desc_channels(d.flags), pcm + (size_t)f * pcm_stride, &handoff[f], &srecs[f]);
    asm volatile("" ::"v"(prefetched));
    if (ret == CONTINUE_SPLIT || ret == CONTINUE_Q4) return;
    if (threadIdx.x == 0) result[f] = ret;
}

extern "C" void og_launch_silk_synth_nb(hipStream_t s, const void *descs, const void *arena, void *streams, void *pcm, void *result, int n,
                                        int n_streams, int pcm_stride, void *handoff, const void *srecs) {
    hipLaunchKernelGGL(k_silk_synth_nb, dim3(n), dim3(64), 0, s, (const FrameDesc *)descs, (const u8 *)arena, (StreamState *)streams,
                       (i16 *)pcm, (i32 *)result, n, n_streams, pcm_stride, (SilkHandoff *)handoff, (const SilkRec *)srecs);
}
