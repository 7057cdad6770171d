// og_recon.hip -- the reconstruction kernel of 20 ms CELT-only frames with an 8 KB LDS working set (five waves per SIMD).
//
// A translation unit of its own because the working set is ONE __shared__ object (FrameLds S, og_state.hpp) whose layout
// is chosen at compile time: here OG_RECON_TIGHT selects the layout without the folding-history rows, the packet buffer and
// the entropy-decoding arrays, with the synthesis buffer starting inside X.  Only the phase-major band loop and the
// synthesis run from this layout; frames it does not take (recon_fast_eligible, og_celt_split.hpp) are left to the general
// kernel k_celt_recon in og_api.hip, which is launched right behind this one and skips the frames done here.
// Why: measured on the 10 KB layout, k_celt_recon's time goes with 1 / (waves per SIMD) -- 2.69 ms at three, 2.09 ms at four.
#define OG_RECON_TIGHT 1
#include <hip/hip_runtime.h>
#include "og_celt_split.hpp"

using namespace og;

#ifndef OG_FAST_WAVES
#define OG_FAST_WAVES 5
#endif
// `hybrid`: the step's hybrid frames come through the split path too (their SILK half by k_silk_parse / k_silk_synth): this
// kernel then also takes their CELT half
__global__ void __launch_bounds__(64, OG_FAST_WAVES) k_celt_recon_fb(const FrameDesc *__restrict__ descs, StreamState *st,
                                                                      const ParseRec *recs, ReconOut *rout, int n, int n_streams,
                                                                      int hybrid) {
    const int f = (int)blockIdx.x;
    if (f >= n) return;
    const FrameDesc d = descs[f];
    const int mode = desc_mode(d.flags);
    if (d.stream < 0 || d.stream >= n_streams || !(mode == MODE_CELT || (mode == MODE_HYBRID && hybrid)) || desc_rfc(d.flags)) return;
    if (mode == MODE_HYBRID && (recs[f].flags & RF_SKIP)) return; // the single-kernel path already reported this frame
    OG_PROF_INIT();
    const int pos = OG_UNI(st[d.stream].celt.ring_pos); // where the frame's first sample goes
    const int ret = celt_recon_wave(&st[d.stream], &recs[f], mode, desc_channels(d.flags), RECON_FAST_ONLY);
    if (ret != RECON_NOT_MINE && threadIdx.x == 0) rout[f] = ReconOut{ret, pos};
    OG_PROF_FLUSH();
}

extern "C" void og_launch_celt_recon_fb(hipStream_t s, const void *descs, void *streams, const void *recs, void *rout, int n,
                                        int n_streams, int hybrid) {
    hipLaunchKernelGGL(k_celt_recon_fb, dim3(n), dim3(64), 0, s, (const FrameDesc *)descs, (StreamState *)streams,
                       (const ParseRec *)recs, (ReconOut *)rout, n, n_streams, hybrid);
}

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
