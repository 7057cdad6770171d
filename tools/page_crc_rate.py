#!/usr/bin/env python3
"""Throughput of the GPU page-checksum kernel (opusgpu_pages_crc_device): synthetic Ogg pages of the mixed workload
(10 packets each, SILK-NB / hybrid / CELT sizes), resident in HBM, HIP-event time per launch.
usage (GPU box): python3 tools/page_crc_rate.py [pages] [launches] [align]
align = 64: experiment -- pages placed so that each one ENDS on a 64-byte boundary of the blob (the kernel's chunks are
counted back from the page's end), which tells how much of the time goes to the loads being unaligned"""
import importlib.util
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("opusgpu_pkg", os.path.join(here, "..", "esp32-opus-player_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pkg)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mats = []
for m, (toc, L) in enumerate(((0x0C, 40), (0x7C, 120), (0xFC, 160))):
    k = n // 3 + (1 if m < n % 3 else 0)
    pay = pkg.lcg_payloads(k, 10, L, seed_base=77 + m)
    mats.append(pkg.build_pages(toc, pay, np.arange(k, dtype=np.uint32)))
blob = np.concatenate([x.reshape(-1) for x in mats] + [np.zeros(16, np.uint8)])
lens = np.concatenate([np.full(x.shape[0], x.shape[1], dtype=np.int32) for x in mats])
offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
page_bytes = int(lens.sum())
align = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if align:
    slot = (lens.astype(np.int64) + 2 * align - 1) // align * align  # room for the page and for moving its end to a boundary
    slot_at = np.concatenate([[0], np.cumsum(slot)[:-1]])
    new_offs = slot_at + slot - lens  # slot ends are multiples of `align`
    moved = np.zeros(int(slot.sum()) + 16, dtype=np.uint8)
    for L in np.unique(lens):
        sel = np.nonzero(lens == L)[0]
        moved[(new_offs[sel][:, None] + np.arange(L)[None, :])] = blob[(offs[sel][:, None] + np.arange(L)[None, :])]
    blob, offs = moved, new_offs
ctx = pkg.Context(0)
d_blob, d_offs, d_lens, d_st = ctx.dev_alloc(blob.size), ctx.dev_alloc(8 * len(lens)), ctx.dev_alloc(4 * len(lens)), ctx.dev_alloc(4 * len(lens))
ctx.h2d(d_blob, blob)
# a wave takes as long as its longest page: once with the three page sizes shuffled (every wave sees all of them), once
# with the pages of equal length next to each other
for name, order in (("sizes shuffled", np.random.default_rng(1).permutation(len(lens))), ("grouped by size", np.arange(len(lens)))):
    ctx.h2d(d_offs, offs[order].copy())
    ctx.h2d(d_lens, lens[order].copy())
    ctx.pages_crc_device(len(lens), d_blob, d_offs, d_lens, d_st)  # warm-up (tables, code)
    ctx.synchronize()
    ev = [ctx.event() for _ in range(launches + 1)]
    ctx.event_record(ev[0])
    for i in range(launches):
        ctx.pages_crc_device(len(lens), d_blob, d_offs, d_lens, d_st)
        ctx.event_record(ev[i + 1])
    ctx.synchronize()
    ms = [ctx.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(launches)]
    st = np.zeros(len(lens), dtype=np.int32)
    ctx.d2h(st, d_st)
    assert (st == 1).all(), "a synthetic page failed its checksum"
    t = float(np.median(ms)) / 1e3
    print(f"{name}: {len(lens)} pages, {page_bytes / 1e6:.0f} MB: {t * 1e3:.3f} ms per launch (median of {launches}) = {page_bytes / t / 1e9:.0f} GB/s "
          f"= {page_bytes / t / 8e12 * 100:.1f} % of 8 TB/s, {len(lens) / t / 1e6:.0f} M pages/s", flush=True)
