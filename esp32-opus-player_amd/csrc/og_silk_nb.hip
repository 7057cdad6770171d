// og_silk_nb.hip -- k_silk_synth_nb: the SILK synthesis kernel (k_silk_synth, og_silk_synth.hip) for NARROWBAND SILK-only frames.
//
// The kernel's working set is sized at compile time: here for a 20 ms frame at 8 kHz (OG_SILK_LDS_FRAME = 160), 6,056 bytes where a
// frame at 16 kHz needs 8,936 -- five LDS granules instead of seven, so five of these frames fit a SIMD (what the kernel's registers
// allow) with room left for the step's other kernels, where four of the wideband kernel's do.  The synthesis waits on its frames'
// serial chains (LPC recurrence, all-pass up-sampler): frames in flight are what it is short of (one granule MORE cost
// the SILK-NB step 5 %).  The loops over a subframe's samples are three per lane instead of five, too.
// Same code otherwise (og_silk_synth_kernel.hpp).  A step with SILK-only frames launches this kernel and then k_silk_synth, which
// leaves the narrowband SILK-only frames alone (`nb_elsewhere`); either kernel's workgroups that find a frame of the other's return
// at once.
#include <hip/hip_runtime.h>
#define OG_SILK_TIGHT 1
#define OG_SILK_LDS_FRAME 160
#define OG_SSYNTH_KERNEL_NAME k_silk_synth_nb
#define OG_SSYNTH_LAUNCHER og_launch_silk_synth_nb
#define OG_SSYNTH_PROF og_ssynth_nb_prof
#define OG_SSYNTH_NB_ONLY 1
// Five waves per SIMD, not the six its 75 registers would allow: the step's entropy kernels (82 and 100 registers a wave) need room
// on the same SIMDs, and the step is as long as their chain -- measured (SILK-NB step, pipelined): 0.94 ms at six, 0.84 at five, 0.90 at four
#ifndef OG_SSYNTH_MAX_WAVES
#define OG_SSYNTH_MAX_WAVES 5
#endif
#include "og_silk_synth_kernel.hpp"
