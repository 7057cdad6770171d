// og_debug.hpp -- every debugging / measurement switch of the library in one place, read from the environment ONCE per process.
//
// None of them is needed in production: the defaults are what the measurements in DESIGN.md settled on.  They exist for A/B
// runs of the same binary (tools/kstats.sh) and for the robustness test of pipelined steps (tools/launch_jitter.py).
//
//   variable                  field            default  meaning
//   OPUSGPU_SPLIT             split            1        0: every frame through the single kernel k_decode_step (round 1's design)
//   OPUSGPU_SPLIT_HYBRID      split_hybrid     1        0: SILK-only and hybrid frames stay on the single kernel
//   OPUSGPU_FAST_RECON        fast_recon       1        0: every CELT frame through the general reconstruction kernel
//   OPUSGPU_SILK_PIPELINE     silk_pipeline    1        0: steps declared SILK-only run in order even with pipelining on (A/B measurements)
//   OPUSGPU_HYBRID_PIPELINE   hybrid_pipeline  1        0: the same for steps declared hybrid (or SILK-only + hybrid)
//   OPUSGPU_PARSE_WIDE        parse_wide       1        0: pipelined steps parse with 32 frames per wave like in-order steps (k_celt_parse instead of k_celt_parse64);
//                                                       2: in-order steps use the wide kernels too, k_silk_parse64 included (counter passes: --pmc serialises kernels, so only in-order steps can be counted)
//   OPUSGPU_HYBRID_RECON_ASIDE hybrid_recon_aside 1     0: pipelined steps with hybrid but no CELT-only frames reconstruct the CELT layer behind the SILK synthesis, on the step's stream, not next to it; 2: steps with CELT-only frames next to it too (measurements)
//   OPUSGPU_SILK_PARAMS_ASIDE silk_params_aside 1       0: pipelined SILK / hybrid steps keep the parameter half (k_silk_params) on the entropy chain's stream
//   OPUSGPU_SILK_NB_KERNEL    silk_nb_kernel   1        0: narrowband SILK-only frames stay in k_silk_synth (no k_silk_synth_nb launch)
//   OPUSGPU_HALVES            halves           1        0: an in-order step runs as ONE chain of kernels, not two
//   OPUSGPU_PARSE_GROUPS      parse_groups     1        groups of frames per workgroup of the early parse, one after the other (1 .. 8)
//   OPUSGPU_PARSE_PRIORITY    parse_priority   1        0: the early parse's stream gets the LOWEST priority instead of the highest
//   OPUSGPU_HOST_PARTS        host_parts       16       slices a large opusgpu_decode_packets call's PCM leaves in (1, 2, 4, 8, 16)
//   OPUSGPU_HOST_SLICES       host_slices      1        0: the round-2 flow -- every part its own in-order step (A/B measurements)
//   OPUSGPU_HOST_TIMING       host_timing      0        1: wall time of the phases of opusgpu_decode_packets on stderr (one batch, waits between phases); 2: of the flow as it is
//   OPUSGPU_PAGES_TIMING      pages_timing     0        1: wall time of the phases of opusgpu_pages_demux on stderr
//   OPUSGPU_LAUNCH_DELAY_US   launch_delay_us  0        the host sleeps this long before every kernel launch of a decode step
//                                                       (robustness of the placement of pipelined steps against launch jitter)
#pragma once
#include <stdlib.h>

struct og_debug_knobs {
    int split = 1, split_hybrid = 1, fast_recon = 1, hybrid_recon_aside = 1, silk_params_aside = 1, silk_nb_kernel = 1, halves = 1, silk_pipeline = 1, hybrid_pipeline = 1, parse_wide = 1, parse_groups = 1, parse_priority = 1, host_parts = 16, host_slices = 1, host_timing = 0,
        pages_timing = 0, launch_delay_us = 0;
};
inline const og_debug_knobs &og_debug() {
    static const og_debug_knobs k = [] {
        og_debug_knobs v;
        auto flag = [](const char *name, int &field) {
            if (const char *e = getenv(name)) field = e[0] != '0' && e[0] != '\0';
        };
        auto number = [](const char *name, int &field, int lo, int hi) {
            if (const char *e = getenv(name)) {
                const int x = atoi(e);
                if (x >= lo && x <= hi) field = x;
            }
        };
        flag("OPUSGPU_SPLIT", v.split);
        flag("OPUSGPU_SPLIT_HYBRID", v.split_hybrid);
        flag("OPUSGPU_FAST_RECON", v.fast_recon);
        number("OPUSGPU_HYBRID_RECON_ASIDE", v.hybrid_recon_aside, 0, 2);
        flag("OPUSGPU_SILK_PARAMS_ASIDE", v.silk_params_aside);
        flag("OPUSGPU_SILK_NB_KERNEL", v.silk_nb_kernel);
        flag("OPUSGPU_HALVES", v.halves);
        flag("OPUSGPU_SILK_PIPELINE", v.silk_pipeline);
        flag("OPUSGPU_HYBRID_PIPELINE", v.hybrid_pipeline);
        number("OPUSGPU_PARSE_WIDE", v.parse_wide, 0, 2);
        number("OPUSGPU_PARSE_GROUPS", v.parse_groups, 1, 8);
        flag("OPUSGPU_PARSE_PRIORITY", v.parse_priority);
        if (const char *e = getenv("OPUSGPU_HOST_PARTS")) {
            const int x = atoi(e);
            if (x == 1 || x == 2 || x == 4 || x == 8 || x == 16) v.host_parts = x;
        }
        flag("OPUSGPU_HOST_SLICES", v.host_slices);
        if (const char *e = getenv("OPUSGPU_HOST_TIMING")) v.host_timing = atoi(e) == 2 ? 2 : 1;
        v.pages_timing = getenv("OPUSGPU_PAGES_TIMING") != nullptr;
        number("OPUSGPU_LAUNCH_DELAY_US", v.launch_delay_us, 0, 100000);
        return v;
    }();
    return k;
}
