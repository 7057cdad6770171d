// og_rfc.hip -- RFC mode (opt-in): one Opus frame per wave at the duration its TOC names, and the loss path
// (lost packets, DTX frames) -- decode_frame_rfc / conceal_frame_rfc, og_decode.hpp.
//
// A translation unit and a kernel of its own so that nothing of this mode is compiled into the reference-mode kernels of
// og_api.hip: their register allocation, LDS footprint and timings stay what the profiles under profiles/ were taken from.  The
// mode exists for completeness, not speed: entropy decoding is wave-uniform, the synthesis of loss-aware SILK frames runs one
// lane per channel.
#include <hip/hip_runtime.h>
#define OG_NO_SPLIT_LDS // the record window of the split path's reconstruction: 256 bytes this kernel does not need -- with them it is
                        // 8 bytes over the 20,480 that eight workgroups per CU (two waves per SIMD, what its 256 registers allow) have
#define OG_SILK_TABLES_LDS // SILK's entropy tables in LDS (og_silk.hpp SILK_TAB): the wave-uniform decoder reads one per symbol
#include "og_decode.hpp"

using namespace og;

static_assert(offsetof(StreamState, loss) + offsetof(LossState, silk) + sizeof(SilkLossChannel) + offsetof(SilkLossChannel, cng_synth_state) + 6 * 64 <= sizeof(StreamState),
              "the prefetch of k_decode_rfc stays inside the stream's record");
#ifndef OG_RFC_WAVES
#define OG_RFC_WAVES 2 // (294 registers unbounded = one wave per SIMD; bounded to 256: 107 -> 65 ms per step of the rfc_mixed workload; 3: no gain)
#endif
__global__ void __launch_bounds__(64, OG_RFC_WAVES) k_decode_rfc(const FrameDesc *__restrict__ descs, const u8 *__restrict__ arena, StreamState *st,
                                                      i16 *pcm, i32 *result, int n, int n_streams, int pcm_stride) {
    const int f = (int)blockIdx.x;
    if (f >= n) return;
    const FrameDesc d = descs[f];
    int ret;
    OG_PROF_INIT();
    u32 prefetched = 0;
    if (d.stream < 0 || d.stream >= n_streams || !desc_rfc(d.flags))
        ret = BAD_ARG; // (in RFC mode every descriptor carries the mode bit: opusgpu_packet_to_frames_mode)
    else {
        // (as in k_silk_synth: one load per lane now brings to the L2 what the frame reads of the stream's state in many dependent
        // steps later -- the SILK state, the scalars and energies behind the CELT history ring, the head of the loss state)
        {
            const int l = (int)threadIdx.x;
            const StreamState *sp = &st[d.stream];
            const char *p = l < 28 ? reinterpret_cast<const char *>(&sp->silk) + 64 * l
                          : l < 40 ? reinterpret_cast<const char *>(&sp->celt.tail[0][0]) + 64 * (l - 28)
                          : l < 52 ? reinterpret_cast<const char *>(&sp->loss.silk[(l - 40) / 6].cng_synth_state[0]) + 64 * ((l - 40) % 6)
                                   : reinterpret_cast<const char *>(sp); // (six lines per channel end inside the record: static_assert below)
            prefetched = *reinterpret_cast<const volatile u32 *>(p);
        }
        ret = decode_frame_rfc(&st[d.stream], arena + d.offset, d.len, desc_mode(d.flags), desc_bandwidth(d.flags),
                               desc_channels(d.flags), pcm + (size_t)f * pcm_stride, desc_frame_size(d.flags), desc_fec(d.flags));
    }
    asm volatile("" ::"v"(prefetched));
    if (threadIdx.x == 0) result[f] = ret;
    OG_PROF_FLUSH();
}
#ifdef OG_PROF
// profiling builds only: this kernel's section counters (OG_MARK) -- every translation unit has its own copy
extern "C" int og_rfc_prof(unsigned long long *out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" void og_launch_decode_rfc(hipStream_t s, const void *descs, const void *arena, void *streams, void *pcm, void *result, int n,
                                     int n_streams, int pcm_stride) {
    hipLaunchKernelGGL(k_decode_rfc, dim3(n), dim3(64), 0, s, (const FrameDesc *)descs, (const u8 *)arena, (StreamState *)streams,
                       (i16 *)pcm, (i32 *)result, n, n_streams, pcm_stride);
}
