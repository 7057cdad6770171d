#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (opusgpu_decode_packets): packets in host memory -> PCM in host
memory, i.e. host framing + H2D of descriptors/arena + the decode kernels + D2H of 3.84 KB PCM per frame.
usage (GPU box): python3 tools/host_path_rate.py [streams] [steps]"""
import importlib.util
import os
import sys
import time

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(here, "..", "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pkg)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = pkg.Context(0)
ctx.streams_alloc(n, 2)
pay = pkg.lcg_payloads(n, steps + 1, 160)
ids = list(range(n))
pk = [[bytes([pkg.TOC_CELT_FB_STEREO]) + pay[s, i].tobytes() for i in range(n)] for s in range(steps + 1)]
ctx.decode_packets(ids, pk[0])  # warm-up (allocations)
t0 = time.perf_counter()
for s in range(1, steps + 1):
    pcm, res = ctx.decode_packets(ids, pk[s])
dt = time.perf_counter() - t0
assert (np.asarray(res) == 960).all()
print("host-buffer path, a list of bytes objects (decode_packets): %d streams x %d steps in %.3f s = %.0f frames/s (%.1f ms/step), PCM D2H %.1f MB/step"
      % (n, steps, dt, n * steps / dt, dt / steps * 1e3, n * 3840 / 1e6))
# the same packets in one array (decode_packets_arena): no per-packet Python work
ctx.streams_alloc(n, 2)
arenas = [np.concatenate([np.full((n, 1), pkg.TOC_CELT_FB_STEREO, dtype=np.uint8), pay[s]], axis=1).reshape(-1) for s in range(steps + 1)]
offs = np.arange(n, dtype=np.int64) * 161
lens = np.full(n, 161, dtype=np.int32)
idv = np.arange(n, dtype=np.int32)
out, res0 = ctx.decode_packets_arena(idv, arenas[0], offs, lens)
t0 = time.perf_counter()
for s in range(1, steps + 1):
    out, res2 = ctx.decode_packets_arena(idv, arenas[s], offs, lens, pcm=out)
dt = time.perf_counter() - t0
assert (res2 == 960).all() and np.array_equal(out, pcm), "the two entries decode the same packets differently"
print("host-buffer path, packets in one array (decode_packets_arena): %d streams x %d steps in %.3f s = %.0f frames/s (%.1f ms/step)"
      % (n, steps, dt, n * steps / dt, dt / steps * 1e3))
