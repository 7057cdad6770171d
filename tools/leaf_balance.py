#!/usr/bin/env python3
"""How well the PVQ leaf pass of the reconstruction kernel is balanced (CPU, host emulation): the leaves of a frame are decoded
one per lane in rounds of 64, a round costs its longest leaf.  Prints, over `frames` frames of the headline workload, the
summed round maxima in record order against the same leaves sorted by cost, for a cost model of n + W * k steps.
usage: python3 tools/leaf_balance.py [streams [frames [W]]]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(ROOT, "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
lib = C.CDLL(os.path.join(ROOT, "tests", "emul", "libog_emul.so"))
lib.emu_state_size.restype = C.c_int
lib.emu_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
lib.emu_last_leaf_geom.argtypes = [C.c_void_p, C.c_int]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
pay = pkg.lcg_payloads(S, F, 160)
out = np.zeros((960, 2), dtype=np.int16)
geom = np.zeros(512, dtype=np.uint32)
tot_now = tot_sorted = tot_sum = 0.0
nl = []
hist_n = np.zeros(256, int)
for s in range(S):
    st = C.create_string_buffer(lib.emu_state_size()); lib.emu_stream_init(st, 2)
    for f in range(F):
        r = lib.emu_decode_frame(st, pay[f, s].tobytes(), 160, 1002, 1105, 2, out.ctypes.data)
        assert r == 960
        n_l = lib.emu_last_leaf_geom(geom.ctypes.data, 512)
        g = geom[:n_l]
        n, k = (g >> 11) & 255, (g >> 19) & 255
        cost = n + W * k
        nl.append(n_l)
        np.add.at(hist_n, n, 1)
        rounds = lambda c: sum(c[i:i + 64].max() for i in range(0, len(c), 64))
        tot_now += rounds(cost); tot_sorted += rounds(np.sort(cost)[::-1]); tot_sum += cost.sum() / 64
print(f"{S * F} frames: leaves per frame mean {np.mean(nl):.0f} max {max(nl)}; cost model n + {W} k")
print(f"  summed round maxima, record order {tot_now / (S * F):.0f} per frame; sorted by cost {tot_sorted / (S * F):.0f}; perfect balance {tot_sum / (S * F):.0f}")
print("  leaf sizes n (count):", {int(i): int(c) for i, c in enumerate(hist_n) if c})
# the walk itself: the wave's loop runs as long as its longest leaf's (tools/pvq_zero_run.py restates the kernel's walk)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pvq_zero_run as Z
lib.emu_last_leaf_idx.argtypes = [C.c_void_p, C.c_int]
idx = np.zeros(512, dtype=np.uint32)
res = {r: [] for r in (0, 1, 2, 4)}
for s in range(min(S, 24)):
    st = C.create_string_buffer(lib.emu_state_size()); lib.emu_stream_init(st, 2)
    for f in range(F):
        lib.emu_decode_frame(st, pay[f, s].tobytes(), 160, 1002, 1105, 2, out.ctypes.data)
        n_l = lib.emu_last_leaf_geom(geom.ctypes.data, 512); lib.emu_last_leaf_idx(idx.ctypes.data, 512)
        for r in res:
            its = [Z.walk_steps(int((g >> 11) & 255), int((g >> 19) & 255), int(ix), r) for g, ix in zip(geom[:n_l], idx[:n_l])]
            res[r].append((max(t[0] for t in its), max(t[1] for t in its)))
for r in res:
    a = np.array(res[r])
    print(f"  index walk, zero runs skipped while n > {r} k (0: never): the frame's longest leaf takes {a[:, 0].mean():.1f} steps, "
          f"{a[:, 1].mean():.1f} of them with a bisection")
