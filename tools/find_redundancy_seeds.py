#!/usr/bin/env python3
"""Finds hybrid packets whose redundancy flag is set (RFC 6716 section 4.5.1) among random payloads: the flag is one range-coded
bit of probability 2^-12, so random data almost never sets it and the RFC-mode parity fuzz would not reach hybrid redundancy.
A packet's first SILK frame is coded independently of the stream's history, so where the flag sits -- and what it decodes to --
depends on the payload alone: the packets found here carry redundancy wherever in a stream they are decoded.
Writes tests/golden/rfc_hybrid_redundancy_seeds.json: {"toc", "len", "seed"} per packet, payload = default_rng(seed) bytes.
    python3 tools/find_redundancy_seeds.py [packets wanted per (toc, kind)]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
o = oracle_py.load()
o.lib.oc_decoder_last_redundancy.argtypes = [C.c_void_p]
want = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out, seed = [], 0
for toc in (0x60, 0x64, 0x68, 0x6C, 0x70, 0x74, 0x78, 0x7C):  # hybrid SWB / FB x 10 / 20 ms x mono / stereo
    channels = 2 if toc & 4 else 1
    d = o.decoder(channels)
    d.set_rfc(True)
    got = {1: 0, 2: 0, 3: 0}  # (2: flagged, then cancelled by the size check -- the hybrid frame's CELT layer conceals)
    while min(got[1], got[3]) < want:
        seed += 1
        n = 40 + seed % 160
        pay = np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()
        d.init()
        _, r = d.decode(bytes([toc]) + pay)
        k = o.lib.oc_decoder_last_redundancy(d.h)
        if r > 0 and k and got[k] < want:
            got[k] += 1
            out.append({"toc": toc, "len": n, "seed": seed, "kind": {1: "silk_to_celt", 2: "cancelled", 3: "celt_to_silk"}[k]})
    print(hex(toc), got, "after", seed, "payloads", flush=True)
path = os.path.join(ROOT, "tests", "golden", "rfc_hybrid_redundancy_seeds.json")
json.dump(out, open(path, "w"), indent=0)
print(len(out), "packets ->", path)
