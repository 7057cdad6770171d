#!/usr/bin/env python3
"""Fuzz the Ogg page demux (csrc/og_pages.cpp, untrusted input) under AddressSanitizer + UBSan: valid pages with random
bit flips, truncations, over-long and zero lengths, random garbage, with and without CRC verification and mode grouping.
    g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -Iinclude -Iesp32-opus-player_amd/csrc \\
        esp32-opus-player_amd/csrc/og_pages.cpp -pthread -o /tmp/libpages_asan.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tools/fuzz_pages_asan.py"""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ogg_util  # noqa: E402

lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else "/tmp/libpages_asan.so")
vp = C.c_void_p
lib.opusgpu_pages_demux.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
lib.opusgpu_page_batch_steps.argtypes = [vp]
lib.opusgpu_page_batch_step.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
lib.opusgpu_page_batch_arena.argtypes = [vp, C.POINTER(C.c_size_t)]
lib.opusgpu_page_batch_arena.restype = vp
lib.opusgpu_page_batch_free.argtypes = [vp]
lib.opusgpu_page_batch_free.restype = None
rng = random.Random(99)
TOCS = [0x0C, 0x7C, 0xFC, 0xFD, 0xFE, 0xFF, 0x08, 0x4B, 0x00, 0xF8]


def packet():
    n = rng.choice([0, 1, 2, 3, 10, 50, 160, 254, 255, 256, 600])
    return bytes([rng.choice(TOCS)]) + bytes(rng.getrandbits(8) for _ in range(n)) if rng.random() < 0.95 else b""


calls = pages_seen = accepted = 0
for it in range(412):
    pages = []
    big = it >= 400  # a few large batches: the slot numbering runs by page ranges in parallel from 256 pages per thread on
    for _ in range(rng.randrange(1500, 2500) if big else rng.randrange(1, 40)):
        pg = bytearray(ogg_util.page(rng.getrandbits(32), rng.getrandbits(16), rng.getrandbits(40), [packet() for _ in range(rng.randrange(0, 12))],
                                     bos=rng.random() < 0.1, eos=rng.random() < 0.1, continued=rng.random() < 0.1))
        how = rng.randrange(8)
        if how == 0 and pg:
            for _ in range(rng.randrange(1, 6)):
                pg[rng.randrange(len(pg))] ^= 1 << rng.randrange(8)
        elif how == 1:
            pg = pg[:rng.randrange(0, len(pg) + 1)]
        elif how == 2:
            pg = bytearray(rng.getrandbits(8) for _ in range(rng.randrange(0, 300)))
        elif how == 3 and len(pg) > 27:
            pg[26] = rng.randrange(256)  # lie about the segment count
        pages.append(bytes(pg))
    blob = np.frombuffer(b"".join(pages) + bytes(8), dtype=np.uint8).copy()
    lens = np.array([len(p) for p in pages], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ptrs = (np.uint64(blob.ctypes.data) + offs).astype(np.uint64)
    ids = np.array([rng.choice([-1, 0, 1, 2, 3, 7, 1000000]) for _ in pages], dtype=np.int32)
    if big:
        ids = np.array([rng.randrange(-1, 400) * (1 if it % 2 else 7919) for _ in pages], dtype=np.int32)
    info = np.zeros(len(pages) * 8, dtype=np.int32)
    for flags in (0, 1, 2, 3):
        h = vp()
        rc = lib.opusgpu_pages_demux(len(pages), ptrs.ctypes.data, lens.ctypes.data, ids.ctypes.data, flags, rng.choice([2, 5, 8]) if big else rng.choice([0, 1, 3]),
                                     info.ctypes.data, C.byref(h))
        assert rc == 0, rc
        nbytes = C.c_size_t()
        arena = lib.opusgpu_page_batch_arena(h, C.byref(nbytes))
        for k in range(lib.opusgpu_page_batch_steps(h)):
            d, sp = vp(), vp()
            n = lib.opusgpu_page_batch_step(h, k, C.byref(d), C.byref(sp))
            descs = np.frombuffer((C.c_uint8 * (16 * n)).from_address(d.value), dtype=np.int32).reshape(n, 4)
            # every descriptor must lie inside the arena and name one of the caller's streams
            assert (descs[:, 1] >= 0).all() and (descs[:, 1] + descs[:, 2] <= nbytes.value).all() and (descs[:, 0] >= 0).all()
            assert len(set(descs[:, 0].tolist())) == n
        lib.opusgpu_page_batch_free(h)
        calls += 1
    pages_seen += len(pages)
    accepted += int((info.reshape(-1, 8)[:, 0] > 0).sum())
print(f"{calls} demux calls, {pages_seen} pages ({accepted} accepted in the last flag setting of each round), no sanitizer report")
