#!/usr/bin/env python3
"""Fuzz the WHOLE host stack above the C ABI under AddressSanitizer + UBSan on the CPU: opusfile.h entry points ->
csrc/og_container.hpp -> csrc/og_compat.cpp (opus_multistream_decode, channel mapping) -> a test double of the C ABI whose
decode is the oracle (tests/emul/og_stack_test.cpp).  Files come from the container fuzz's generator, with extra weight
on packets the glue must survive: many short frames (Q6), every frame-count code, odd channel counts in OpusHead.
    make -C tests/emul stack_asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:abort_on_error=1 \\
        python3 tools/fuzz_stack_asan.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_container_asan as gen  # noqa: E402

rng = gen.rng
# more of the packet shapes that matter for the decoder glue: code 3 with 2.5 / 5 / 10 ms configurations
gen.TOCS += [(28 << 3) | 4 | 3, (29 << 3) | 3, (30 << 3) | 4 | 3, (16 << 3) | 4 | 3, 0xFF, 0x0F, 0x7F]

lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emul", "libstack_asan.so"))
lib.st_open.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
lib.st_read.argtypes = [C.c_void_p, C.c_int]
lib.st_close.restype = None
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
guard = 4096
buf = np.full(2048 * 2 + guard, 0x7A7A, dtype=np.int16)
opened = reads = samples = 0
for it in range(N):
    data = gen.make_file()
    if not lib.st_open(data, len(data), rng.choice([-1, 0])):
        continue
    opened += 1
    for _ in range(200):
        size = rng.choice([2048, 2048, 960, 100, 2])
        buf[:] = 0x7A7A
        r = lib.st_read(buf.ctypes.data, size)
        # at most _buf_size / 2 samples per channel, i.e. _buf_size values, may be written
        assert (buf[size:] == 0x7A7A).all(), "op_read_stereo wrote past the caller's buffer"
        if r <= 0:
            break
        assert r <= size // 2
        reads += 1
        samples += r
lib.st_close()
print(f"{N} files, {opened} opened, {reads} successful reads, {samples} samples per channel delivered; reached the end")
