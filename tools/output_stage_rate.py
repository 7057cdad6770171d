#!/usr/bin/env python3
"""Throughput of the output-stage kernel (opusgpu_output_stage_device): one decode step's PCM (blocks of 960 stereo
samples) -> I2S words, resident in HBM, HIP-event time per launch.  Algorithmic bytes: 4 in + 4 out per output word.
usage (GPU box): python3 tools/output_stage_rate.py [blocks] [launches]"""
import importlib.util
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(here, "..", "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pkg)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = pkg.Context(0)
pcm = np.random.default_rng(1).integers(-32768, 32768, size=n * 1920, dtype=np.int16)
d_pcm, d_out = ctx.dev_alloc(pcm.nbytes), ctx.dev_alloc(4 * n * 960)
ctx.h2d(d_pcm, pcm)
for name, kw in (("16-bit stereo", {}), ("16-bit stereo, force mono, volume 40", dict(force_mono=True, volume=40))):
    ctx.output_stage_device(n, 960, d_pcm, 1920, d_out, 960, valid_all=960, **kw)
    ctx.synchronize()
    ev = [ctx.event() for _ in range(launches + 1)]
    ctx.event_record(ev[0])
    for i in range(launches):
        ctx.output_stage_device(n, 960, d_pcm, 1920, d_out, 960, valid_all=960, **kw)
        ctx.event_record(ev[i + 1])
    ctx.synchronize()
    t = float(np.median([ctx.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(launches)])) / 1e3
    nbytes = 8 * n * 960
    print(f"{name}: {n} blocks, {nbytes / 1e6:.0f} MB moved: {t * 1e3:.3f} ms per launch (median of {launches}) = {nbytes / t / 1e9:.0f} GB/s "
          f"= {nbytes / t / 8e12 * 100:.1f} % of 8 TB/s", flush=True)
