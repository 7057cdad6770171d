// og_parse64.hip -- k_celt_parse64: the CELT parse kernel with 64 frames per wave (og_parse_kernel.hpp says why there are two).
// A translation unit of its own because the frames per wave size the kernel's LDS arrays at compile time.
#include <hip/hip_runtime.h>
#define OG_PL_LANES 64
#ifndef OG_PL64_WAVES
#define OG_PL64_WAVES 2 // two such waves per workgroup share one copy of the ROM tables (3.3 KB): next to the reconstruction it is LDS x residency that counts
#endif
#define OG_PL_WAVES OG_PL64_WAVES
#define OG_PARSE_KERNEL_NAME k_celt_parse64
#define OG_PARSE_DYN_LDS 1
#ifndef OG_PARSE_WAVES_PER_SIMD
#define OG_PARSE_WAVES_PER_SIMD 3 // the register budget: 168
#endif
#include "og_parse_kernel.hpp"

extern "C" void og_launch_celt_parse64(hipStream_t s, int grid, const void *descs, const void *arena, void *streams, void *recs, int n,
                                       int n_streams, const void *handoff, int which, int groups, unsigned *started) {
    hipLaunchKernelGGL(k_celt_parse64, dim3(grid), dim3(64 * OG_PL_WAVES), OG_PARSE_LDS_BYTES, s, (const FrameDesc *)descs, (const u8 *)arena, (StreamState *)streams,
                       (ParseRec *)recs, n, n_streams, (const SilkHandoff *)handoff, which, groups, started);
}
extern "C" int og_celt_parse64_frames(void) { return OG_PL_FRAMES; }
