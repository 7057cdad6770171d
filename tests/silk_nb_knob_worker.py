"""Child process of tests/test_gpu_pipeline.py::test_narrowband_frames_in_either_synthesis_kernel: with OPUSGPU_SILK_NB_KERNEL as the parent
set it (og_debug.hpp; read once per process), decode 6 steps of 4,096 SILK-NB stereo streams and 6 steps that mix SILK-NB, SILK-WB and
hybrid streams -- in order and pipelined -- and compare every sample with the oracle.  With the switch at 0 the narrowband frames
run in k_silk_synth (buffers sized for 16 kHz) instead of k_silk_synth_nb: the path nothing else exercises.  (GPU box.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest
import oracle_py

pkg = conftest.load_pkg()
oracle = oracle_py.load()
n, frames, L = 4096, 6, 70
rng = np.random.default_rng(4242)
ctx = pkg.Context(0)
for mixed in (False, True):
    tocs = np.full(n, 0x0C, dtype=np.uint8) if not mixed else rng.choice(np.array([0x0C, 0x08, 0x4C, 0x7C, 0x2C], dtype=np.uint8), size=n)
    pkts = [[bytes([int(tocs[s])]) + rng.integers(0, 256, size=L, dtype=np.uint8).tobytes() for s in range(n)] for _ in range(frames)]
    decs = [oracle.decoder(2) for _ in range(n)]
    ref = np.zeros((frames, n, 960, 2), dtype=np.int16)
    for s in range(n):
        decs[s].init()
        for f in range(frames):
            buf, r = decs[s].decode(pkts[f][s])
            assert r == 960
            ref[f, s] = buf[:960]
    for pipelined in (False, True):
        ctx.streams_alloc(n, 2)
        ctx.set_pipeline(pipelined)
        for f in range(frames):
            pcm, res = ctx.decode_packets(np.arange(n), pkts[f])
            assert (np.asarray(res) == 960).all(), (mixed, pipelined, f)
            got = np.asarray(pcm)[:, :960]
            for s in np.nonzero((got != ref[f]).any(axis=(1, 2)))[0]:
                # (a mono SILK-only packet in a stereo decoder defines only the first 960 entries, Q3)
                if not (tocs[s] & 4) and not (tocs[s] & 0x80) and (tocs[s] & 0x60) != 0x60:
                    assert np.array_equal(got[s].reshape(-1)[:960], ref[f, s].reshape(-1)[:960]), (mixed, pipelined, f, int(s))
                else:
                    raise AssertionError((mixed, pipelined, f, int(s), hex(int(tocs[s]))))
        ctx.set_pipeline(False)
ctx.close()
print("silk nb knob worker ok")
