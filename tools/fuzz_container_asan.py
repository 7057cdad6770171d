#!/usr/bin/env python3
"""Fuzz the single-file Ogg Opus reader (csrc/og_container.hpp: page sync, CRC, lacing, continued packets, headers, granule
bookkeeping; untrusted input) under AddressSanitizer + UBSan, with a stub decode callback so that only container logic
runs.  Files are built page by page with structural mutations and, mostly, recomputed CRCs (so the damage gets past the
CRC check), then bit-flipped / truncated.
    make -C tests/emul container_asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:abort_on_error=1 \
        python3 tools/fuzz_container_asan.py
(UBSan only prints by default: with halt_on_error the process dies at the first report, so reaching the last line means none.)"""
import ctypes as C
import os
import random
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ogg_util  # noqa: E402

rng = random.Random(2024)
TOCS = [0xFC, 0xFD, 0xFF, 0x0C, 0x7C, 0x08, 0x00, 0x4B, 0xF8, 0xE3]


def raw_page(serial, seqno, granule, lacing, body, flags, fix_crc=True):
    hdr = bytearray(b"OggS\x00" + bytes([flags & 255]) + struct.pack("<qIII", granule, serial & 0xFFFFFFFF, seqno & 0xFFFFFFFF, 0)
                    + bytes([len(lacing)]) + bytes(lacing))
    if fix_crc:
        hdr[22:26] = struct.pack("<I", ogg_util.ogg_crc(bytes(hdr) + bytes(body)))
    else:
        hdr[22:26] = struct.pack("<I", rng.getrandbits(32))
    return bytes(hdr) + bytes(body)


def make_file():
    serial = rng.getrandbits(32)
    head = bytearray(ogg_util.opus_head(channels=rng.choice([1, 2, 2, 2, 0, 3, 255]), pre_skip=rng.choice([0, 312, 3840, 65535]),
                                        family=rng.choice([0, 0, 0, 1, 255])))
    if rng.random() < 0.1:
        head[8] = rng.choice([0, 2, 15, 16, 255])  # version
    if rng.random() < 0.1:
        head = head[:rng.randrange(0, len(head))]
    pages = [ogg_util.page(serial, 0, 0, [bytes(head)], bos=rng.random() < 0.95)]
    if rng.random() < 0.9:
        pages.append(ogg_util.page(serial, 1, 0, [ogg_util.opus_tags() if rng.random() < 0.9 else bytes(rng.getrandbits(8) for _ in range(20))]))
    gp, seq = 0, 2
    pending = b""  # tail of a packet that continues on the next page
    for _ in range(rng.randrange(0, 12)):
        lacing, body = [], bytearray()
        cont = bool(pending)
        if pending:
            body += pending
            k = len(pending)
            while k >= 255 and len(lacing) < 255:
                lacing.append(255); k -= 255
            if len(lacing) < 255:
                lacing.append(k)
            pending = b""
        for _ in range(rng.randrange(0, 14)):
            if len(lacing) >= 250:
                break
            n = rng.choice([0, 1, 2, 10, 100, 161, 254, 255, 256, 510, 700])
            pkt = bytes([rng.choice(TOCS)]) + bytes(rng.getrandbits(8) for _ in range(n))
            if rng.random() < 0.1 and len(pkt) > 300:  # span: head here, tail on the next page
                cut = 255 * rng.randrange(1, len(pkt) // 255 + 1)
                cut = min(cut, len(pkt) - 1) // 255 * 255
                if cut > 0 and len(lacing) + cut // 255 <= 255:
                    body += pkt[:cut]
                    lacing += [255] * (cut // 255)
                    pending = pkt[cut:]
                    break
            k = len(pkt)
            body += pkt
            while k >= 255:
                lacing.append(255); k -= 255
            lacing.append(k)
            gp += rng.choice([960, 960, 960, 1920, 0, 120])
        lacing = lacing[:255]
        flags = (1 if cont else 0) | (4 if rng.random() < 0.1 else 0) | (rng.getrandbits(8) if rng.random() < 0.03 else 0)
        granule = rng.choice([gp, gp, gp, -1, 0, gp - 5000, 2 ** 62, -(2 ** 63)])
        how = rng.random()
        if how < 0.05:
            lacing = [rng.randrange(256) for _ in range(rng.randrange(0, 256))]  # table that does not match the body
        pages.append(raw_page(serial if rng.random() < 0.95 else rng.getrandbits(32), seq if rng.random() < 0.9 else rng.getrandbits(32),
                              granule, lacing, body, flags, fix_crc=rng.random() < 0.9))
        seq += 1
    rng.random() < 0.1 and rng.shuffle(pages)
    data = bytearray(b"".join(pages))
    m = rng.random()
    if m < 0.25 and data:
        for _ in range(rng.randrange(1, 8)):
            data[rng.randrange(len(data))] ^= 1 << rng.randrange(8)
    elif m < 0.4:
        data = data[:rng.randrange(0, len(data) + 1)]
    elif m < 0.5:
        at = rng.randrange(0, len(data) + 1)
        data[at:at] = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 400)))
    return bytes(data)


def main():
    lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "emul", "libct_asan.so"))
    lib.ct_open.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    lib.ct_read_stereo.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros(2048 * 2, dtype=np.int16)
    opened = reads = samples = 0
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    for it in range(N):
        data = make_file()
        if lib.ct_open(data, len(data), rng.choice([-1, 0])) != 0:
            continue
        opened += 1
        for _ in range(300):
            r = lib.ct_read_stereo(buf.ctypes.data, rng.choice([2048, 2048, 960, 100, 2]))
            if r <= 0:
                break
            assert r <= 1024
            reads += 1
            samples += r
    print(f"{N} files, {opened} opened, {reads} successful reads, {samples} samples per channel delivered; reached the end (run with UBSAN_OPTIONS=halt_on_error=1 for that to mean: no report)")


if __name__ == "__main__":
    main()
