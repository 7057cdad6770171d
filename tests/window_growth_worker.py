"""Child process of tests/test_gpu_pipeline.py::test_windows_that_grow_on_a_fresh_context: on a context whose record slots have
never been sized, (1) a window whose steps grow (every later step larger than anything its slot has held), (2) a second, larger
window on the same context, (3) a window with a refused step in the middle (null pointer: refused before anything is queued, the
context stays usable), (4) SILK-NB steps of growing size with the parse kernels' pipelining on -- every decoded sample against the
oracle.  A hang here (the parent's timeout) is the failure mode this guards against: growing a slot frees the old buffer, and
hipFree waits for every stream, including one that waits for a kernel the host has yet to launch.  (GPU box.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest
import oracle_py

pkg = conftest.load_pkg()
oracle = oracle_py.load()


def run_steps(ctx, toc, L, sizes, modes, window, seed, bad_step=None):
    n, frames = max(sizes), len(sizes)
    pay = np.random.default_rng(seed).integers(0, 256, (frames, n, L), dtype=np.uint8)
    ctx.streams_alloc(n, 2)
    ctx.set_pipeline(True)
    tabs, frees = [], []
    for f in range(frames):
        arena, descs = pkg.build_step(toc, pay[f, :sizes[f]])
        a, d = ctx.dev_alloc(arena.nbytes + 16), ctx.dev_alloc(descs.nbytes)
        ctx.h2d(a, arena)
        ctx.h2d(d, descs)
        p, r = ctx.dev_alloc(sizes[f] * 960 * 2 * 2), ctx.dev_alloc(4 * sizes[f])
        tabs.append((d, a, p, r))
        frees += [a, d, p, r]
    if bad_step is not None:
        ptrs = [t[2] for t in tabs]
        ptrs[bad_step] = 0
        try:
            ctx.decode_steps_device(sizes, [t[0] for t in tabs], [t[1] for t in tabs], ptrs, [t[3] for t in tabs], modes=modes)
            raise SystemExit("a window with a null PCM pointer was accepted")
        except pkg.OpusGpuError as e:
            assert e.code == -1, e.code
        ctx.synchronize()  # (nothing was queued: this returns)
    if window:
        ctx.decode_steps_device(sizes, [t[0] for t in tabs], [t[1] for t in tabs], [t[2] for t in tabs], [t[3] for t in tabs], modes=modes)
    else:
        for f in range(frames):
            ctx.decode_step_device(sizes[f], tabs[f][0], tabs[f][1], tabs[f][2], tabs[f][3], modes=modes)
    ctx.synchronize()
    decs = {}
    for f in range(frames):
        got = np.zeros((sizes[f], 960, 2), dtype=np.int16)
        res = np.zeros(sizes[f], dtype=np.int32)
        ctx.d2h(got, tabs[f][2])
        ctx.d2h(res, tabs[f][3])
        assert (res == 960).all(), f
        for s in sorted(set(list(range(min(sizes[f], 24))) + [sizes[f] - 1])):
            if s not in decs:
                decs[s] = oracle.decoder(2)
                decs[s].init()
                for g in range(f):
                    if s < sizes[g]:
                        decs[s].decode(bytes([toc]) + pay[g, s].tobytes())
            out, r = decs[s].decode(bytes([toc]) + pay[f, s].tobytes())
            assert r == 960 and np.array_equal(out[:960], got[s]), (f, s)
    ctx.set_pipeline(False)
    for p in frees:
        ctx.dev_free(p)


ctx = pkg.Context(0)
run_steps(ctx, pkg.TOC_CELT_FB_STEREO, 160, [64, 300, 1500, 33, 2500, 4096, 1, 6000], pkg.HAS_CELT, True, 1)
print("growing window ok", flush=True)
run_steps(ctx, pkg.TOC_CELT_FB_STEREO, 160, [9000, 100, 12000, 20000], pkg.HAS_CELT, True, 2)
print("second, larger window ok", flush=True)
run_steps(ctx, pkg.TOC_CELT_FB_STEREO, 160, [500, 30000, 700], pkg.HAS_CELT, True, 3, bad_step=1)
print("refused window ok", flush=True)
ctx.close()
ctx = pkg.Context(0)
run_steps(ctx, pkg.TOC_SILK_NB_STEREO, 40, [64, 700, 90, 3000, 5000, 100, 9000], pkg.HAS_SILK, False, 4)
print("growing SILK steps ok", flush=True)
run_steps(ctx, pkg.TOC_HYBRID_FB_STEREO, 120, [100, 2000, 50, 7000, 12000], pkg.HAS_HYBRID, False, 5)
print("growing hybrid steps ok", flush=True)
ctx.close()
print("window growth worker ok")
