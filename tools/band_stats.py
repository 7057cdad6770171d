#!/usr/bin/env python3
"""How often the CELT band loop takes each of its paths on the bench payloads (host emulation built with -DOG_STATS):
   g++ -std=c++17 -O2 -fPIC -shared -fwrapv $(cat esp32-opus-player_amd/csrc/BUILD_FLAGS) -DOG_STATS -DOG_HOST_EMUL \\
       -Iesp32-opus-player_amd/csrc tests/emul/og_emul.cpp -o /tmp/libog_emul_stats.so
usage: python3 tools/band_stats.py [/tmp/libog_emul_stats.so] [streams] [frames]   (CPU only)"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(here, "..", "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pkg)
NAMES = {0: "frames", 19: "transient frames", 13: "bands", 14: "N == 1 bands", 15: "bands ending in a stereo merge",
         16: "N == 2 stereo bands", 17: "dual-stereo bands", 18: "bands with a folding source available", 1: "jobs",
         2: "jobs preparing a folding source", 5: "jobs with tf change", 6: "jobs in short-block frames",
         7: "jobs undoing a Hadamard interleave", 8: "Haar passes on x", 9: "jobs writing folding history",
         4: "PVQ leaves", 3: "fill leaves", 40: "longest PVQ leaf: coefficients", 41: "longest PVQ leaf: pulses",
         42: "coefficients in PVQ leaves", 44: "PVQ leaves (leaf pass)",
         45: "frames whose longest leaf has >= 96 coefficients", 46: "frames whose longest leaf has >= 144 coefficients",
         50: "comb filter calls that filter", 51: "... with lag 15", 52: "... with lag 1020..1022", 53: "... with lag 1022", 20: "fill jobs (phase D)", 24: "... skipped: empty mask, no pulses", 21: "frames with tf work in the parallel pass", 22: "... incl. an interleave", 23: "frames with fill jobs", 10: "fill leaves left zero", 11: "fill leaves: noise", 12: "fill leaves: folded"}
lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else "/tmp/libog_emul_stats.so")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib.emu_state_size.restype = C.c_int
lib.emu_stats.restype = C.POINTER(C.c_longlong)
lib.emu_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
pay = pkg.lcg_payloads(n, frames, 160)
out = np.zeros(960 * 2, dtype=np.int16)
for s in range(n):
    st = C.create_string_buffer(lib.emu_state_size())
    lib.emu_stream_init(st, 2)
    for f in range(frames):
        r = lib.emu_decode_frame(st, pay[f, s].tobytes(), 160, 1002, 1105, 2, out.ctypes.data)
        assert r == 960, r
st = lib.emu_stats()
fr = st[0]
print(f"{fr} CELT-FB stereo frames, 160-byte LCG payloads; per frame:")
for k in (19, 21, 22, 23, 20, 24, 13, 14, 15, 16, 17, 18, 1, 2, 5, 6, 7, 8, 9, 4, 3, 10, 11, 12, 44, 42, 40, 41, 45, 46, 50, 51, 52, 53):
    print(f"  {NAMES[k]:42s} {st[k] / fr:8.2f}")
