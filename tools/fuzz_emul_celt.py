#!/usr/bin/env python3
"""CPU parity fuzz of the split CELT path (parse per lane / phase-major reconstruction / post) in host emulation against
the oracle: CELT-only frames of every bandwidth, mono and stereo packets in mono and stereo decoders, payloads from a few
bytes (most leaves without pulses: folding, noise fill, the serial fill jobs) to 1275 (every leaf with pulses), state
carried over `frames` frames per stream.     python3 tools/fuzz_emul_celt.py [streams [frames [seed [hybrid]]]]
With a fourth argument "hybrid": hybrid SWB / FB frames (their CELT layer, bands 17 - 20, runs the same phase-major band loop
from band 17: the folding history counts from there and the second band folds from a stretched copy of the first) mixed with
CELT-only ones in the same streams (regular emulation library only: the tight-layout one has no SILK half)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
o = oracle_py.load()
lib = C.CDLL(os.environ.get("OG_EMUL_LIB", os.path.join(ROOT, "tests", "emul", "libog_emul.so")))
lib.emu_state_size.restype = C.c_int
lib.emu_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
STREAMS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 11)
HYBRID = len(sys.argv) > 4 and sys.argv[4] == "hybrid"
out = np.zeros((960, 2), dtype=np.int16)
n = bad = left = 0
for s in range(STREAMS):
    channels = int(rng.integers(1, 3))
    d = o.decoder(channels); d.init()
    st = C.create_string_buffer(lib.emu_state_size()); lib.emu_stream_init(st, channels)
    base_len = int(rng.choice([6, 12, 20, 30, 45, 60, 80, 120, 160, 250, 400, 800, 1275]))
    for f in range(FRAMES):
        cfg = int(rng.choice([13, 15, 13, 15, 31, 27])) if HYBRID else int(rng.choice([19, 23, 27, 31]))
        stereo = (channels == 2) if rng.random() < 0.85 else bool(rng.integers(2))
        toc = (cfg << 3) | (4 if stereo else 0)
        L = max(2, min(1275, int(base_len * rng.uniform(0.6, 1.4))))
        body = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
        ref, r = d.decode(bytes([toc]) + body)
        if toc & 0x80:
            m, bw = 1002, 1102 + ((toc >> 5) & 3)
            bw = 1101 if bw == 1102 else bw
        else:
            m, bw = 1001, (1105 if toc & 0x10 else 1104)
        out[:] = 0
        r2 = lib.emu_decode_frame(st, body, L, m, bw, 2 if stereo else 1, out.ctypes.data)
        n += 1
        if r2 == -999:  # (the tight-layout library: a frame its kernel leaves to the general one -- the stream ends here)
            left += 1
            break
        if r != r2 or (r > 0 and not np.array_equal(out.reshape(-1)[:960 * channels], ref[:960].reshape(-1)[:960 * channels])):
            bad += 1
            if bad <= 5:
                print("MISMATCH stream", s, "frame", f, "toc", hex(toc), "len", L, "channels", channels, r, r2)
print(f"{n} frames, {bad} mismatches" + (f" ({left} frames left to the general kernel)" if left else ""))
sys.exit(1 if bad else 0)
