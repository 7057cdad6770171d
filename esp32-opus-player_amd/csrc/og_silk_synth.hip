// og_silk_synth.hip -- k_silk_synth: the SILK synthesis kernel of the split path (SILK-only frames at 12 and 16 kHz, the SILK layer of
// hybrid frames; the narrowband SILK-only frames too when k_silk_synth_nb is not launched), with the tight layout of its working
// set: 8,936 bytes, seven LDS granules.  og_silk_synth_kernel.hpp says why this is a translation unit of its own.
#include <hip/hip_runtime.h>
#define OG_SILK_TIGHT 1
#define OG_SSYNTH_KERNEL_NAME k_silk_synth
#define OG_SSYNTH_LAUNCHER og_launch_silk_synth
#define OG_SSYNTH_PROF og_ssynth_prof
#define OG_SSYNTH_NB_ONLY 0
// (no cap on its waves per SIMD: seven granules of LDS admit 17 workgroups per CU, fewer than its 76 registers would; capped at five: the same times)
#include "og_silk_synth_kernel.hpp"
