"""Child process of tests/test_gpu_pipeline.py::test_placement_knobs_change_no_result: with whatever og_debug.hpp switches the
parent put into the environment, decode 8 steps of 8,192 CELT-FB streams with pipelining on -- as one window and one call per
step -- and compare every sample with the oracle.  (GPU box.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest
import oracle_py

pkg = conftest.load_pkg()
oracle = oracle_py.load()
n, frames, L = 8192, 8, 160
toc = pkg.TOC_CELT_FB_STEREO
pay = pkg.lcg_payloads(n, frames, L, seed_base=0x9191)
ref, ok = oracle.batch_decode_threads(2, toc, pay)
assert ok == n * frames
ctx = pkg.Context(0)
for window in (True, False):
    ctx.streams_alloc(n, 2)
    ctx.set_pipeline(True)
    tabs, outs = [], []
    for f in range(frames):
        arena, descs = pkg.build_step(toc, pay[f])
        a, d = ctx.dev_alloc(arena.nbytes + 16), ctx.dev_alloc(descs.nbytes)
        ctx.h2d(a, arena)
        ctx.h2d(d, descs)
        tabs.append((d, a))
        outs.append((ctx.dev_alloc(n * 960 * 2 * 2), ctx.dev_alloc(4 * n)))
    if window:
        ctx.decode_steps_device([n] * frames, [t[0] for t in tabs], [t[1] for t in tabs], [o[0] for o in outs], [o[1] for o in outs],
                                modes=pkg.HAS_CELT)
    else:
        for f in range(frames):
            ctx.decode_step_device(n, tabs[f][0], tabs[f][1], outs[f][0], outs[f][1], modes=pkg.HAS_CELT)
    ctx.synchronize()
    got = np.zeros((n, 960, 2), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    for f in range(frames):
        ctx.d2h(got, outs[f][0])
        ctx.d2h(res, outs[f][1])
        assert (res == 960).all(), (window, f)
        assert np.array_equal(got, ref[:, f]), (window, f)
    ctx.set_pipeline(False)
    for t in tabs + outs:
        ctx.dev_free(t[0])
        ctx.dev_free(t[1])
ctx.close()
print("knob worker ok")
