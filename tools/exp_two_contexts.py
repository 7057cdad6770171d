#!/usr/bin/env python3
"""Experiment: do the kernels of two independent contexts (own HIP streams) overlap productively on one GPU?
Decodes the same total number of CELT-FB stereo frames per step three ways and prints ms per step (wall clock around
K asynchronous steps, synchronised at both ends):
  A  one context, n streams
  B  two contexts, n/2 streams each, steps issued alternately (ctx0 step f, ctx1 step f, ...): their kernels may overlap
  C  two contexts, n/2 streams each, ctx0 runs all its steps and is synchronised before ctx1 starts (no overlap possible)
usage (GPU box): python3 tools/exp_two_contexts.py [n] [steps]"""
import importlib.util, os, sys, time
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(here, "..", "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
L, toc, warm = 160, pkg.TOC_CELT_FB_STEREO, 2


class Job:
    def __init__(self, m, seed_off):
        self.m = m
        self.ctx = pkg.Context(0)
        self.ctx.streams_alloc(m, 2)
        pay = pkg.lcg_payloads(m, steps + warm, L)
        self.d_pcm = self.ctx.dev_alloc(m * 960 * 2 * 2)
        self.d_res = self.ctx.dev_alloc(4 * m)
        self.d_desc, self.d_arena = [], []
        for f in range(steps + warm):  # every step's input resident before timing
            arena, descs = pkg.build_step(toc, pay[f])
            a, d = self.ctx.dev_alloc(len(arena) + 16), self.ctx.dev_alloc(16 * m)
            self.ctx.h2d(a, arena); self.ctx.h2d(d, descs)
            self.d_arena.append(a); self.d_desc.append(d)
        self.ctx.synchronize()

    def step(self, f):
        self.ctx.decode_step_device(self.m, self.d_desc[f], self.d_arena[f], self.d_pcm, self.d_res)

    def check(self):
        res = np.zeros(self.m, dtype=np.int32)
        self.ctx.synchronize(); self.ctx.d2h(res, self.d_res)
        assert (res == 960).all()


def timed(fn):
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3 / steps


one = Job(n, 0)
for f in range(warm): one.step(f)
one.ctx.synchronize()
def run_a():
    for f in range(warm, warm + steps): one.step(f)
    one.ctx.synchronize()
a = timed(run_a); one.check()
print("A one context, %d streams:                 %.3f ms/step" % (n, a), flush=True)
del one

j0, j1 = Job(n // 2, 0), Job(n // 2, 1)
for f in range(warm): j0.step(f); j1.step(f)
j0.ctx.synchronize(); j1.ctx.synchronize()
half = warm + steps // 2
def run_b():
    for f in range(warm, half): j0.step(f); j1.step(f)
    j0.ctx.synchronize(); j1.ctx.synchronize()
def run_c():
    for f in range(half, warm + steps): j0.step(f)
    j0.ctx.synchronize()
    for f in range(half, warm + steps): j1.step(f)
    j1.ctx.synchronize()
b = timed(run_b) * 2  # each of B and C covers steps/2 steps of the full batch
c = timed(run_c) * 2
j0.check(); j1.check()
print("B two contexts, %d streams each, interleaved: %.3f ms/step" % (n // 2, b))
print("C two contexts, %d streams each, serialised:  %.3f ms/step" % (n // 2, c))
