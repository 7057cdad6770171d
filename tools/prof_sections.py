#!/usr/bin/env python3
"""Section profile of k_celt_recon from a -DOG_PROF build (OPUSGPU_LIB=<that .so>): share of wave cycles per OG_MARK id.
usage (GPU box): OPUSGPU_LIB=$PWD/build_exp/lib_prof.so python3 tools/prof_sections.py [streams] [steps]"""
import ctypes
import importlib.util
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(here, "..", "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pkg)

NAMES = {0: "outside", 1: "stage record", 2: "leaf pass", 3: "band prologue", 4: "stereo setup", 5: "band_mono pre",
         6: "tree walk", 7: "leaf (fill)", 8: "band_mono post", 9: "lowband out", 10: "stereo merge", 11: "N==1 band",
         12: "anti-collapse", 13: "synth prologue", 14: "imdct", 15: "comb filter", 16: "ring write", 17: "epilogue",
         20: "parse: init", 21: "parse: flags", 22: "parse: coarse energy", 23: "parse: tf/spread/dynalloc", 24: "parse: allocation",
         25: "parse: fine energy", 26: "parse: bands", 27: "parse: finalise", 28: "leaf: rotations by the wave",
         56: "leaf: index walk (cwrsi)", 57: "leaf: collapse mask", 58: "leaf: scale", 59: "leaf: rotation",
         60: "silk core: subframe setup (gains, re-whitening)", 61: "silk core: excitation", 62: "silk core: LTP prediction",
         63: "silk core: LPC recurrence + output scaling",
         40: "parse: stereo theta + band words", 41: "parse: tree descend (split theta)", 42: "parse: leaf bits2pulses", 43: "parse: leaf index (rc_uint)",
         44: "parse: tree ascend",
         45: "silk parse: LBRR skip-decode (indices; its pulses under 46-49)", 46: "silk parse: pulses per block", 47: "silk parse: shell tree",
         48: "silk parse: LSBs", 49: "silk parse: signs",
         50: "silk parse: init + flags + stereo", 51: "silk parse: indices", 52: "silk parse: rate level", 53: "silk parse: parameters",
         54: "silk parse: epilogue",
         30: "silk: flags + stereo pred + indices", 31: "silk: pulses", 32: "silk: parameters (NLSF->LPC, gains)", 33: "silk: bookkeeping",
         34: "silk: stage + core (LTP/LPC)", 35: "silk: outBuf + MS->LR", 36: "silk: up2 (serial)", 37: "silk: FIR to 48k", 38: "silk: epilogue"}

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
toc, plen = {"celt": (pkg.TOC_CELT_FB_STEREO, 160), "silk": (pkg.TOC_SILK_NB_STEREO, 40),
             "hybrid": (pkg.TOC_HYBRID_FB_STEREO, 120)}[sys.argv[3] if len(sys.argv) > 3 else "celt"]
ctx = pkg.Context(0)
ctx.streams_alloc(n, 2)
lib = pkg.load_lib()
lib.opusgpu_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
pay = pkg.lcg_payloads(n, steps, plen)
for s in range(steps):
    pkts = [bytes([toc]) + pay[s, i].tobytes() for i in range(n)]
    if s == 1:
        lib.opusgpu_debug_prof(buf, 1)  # drop the first (cold) step
    ctx.decode_packets(list(range(n)), pkts)
lib.opusgpu_debug_prof(buf, 0)
tot = sum(buf)
print("section                 cycles/frame   share")
for i in range(64):
    if buf[i]:
        print("%2d %-20s %10.0f   %5.1f%%" % (i, NAMES.get(i, "?"), buf[i] / (n * (steps - 1)), 100.0 * buf[i] / tot))
print("total cycles/frame %.0f" % (tot / (n * (steps - 1))))
