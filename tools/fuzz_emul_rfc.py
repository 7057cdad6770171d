#!/usr/bin/env python3
"""CPU parity fuzz of RFC mode: the kernel source in host emulation (decode_frame_rfc, og_decode.hpp) against the oracle's RFC
mode (oc_decoder_set_rfc) -- all 32 TOC configurations, mono and stereo packets in mono and stereo decoders, frame-count codes
0..3, configuration switches inside a stream (incl. hybrid -> SILK-only: the silence-frame fade-out).
    python3 tools/fuzz_emul_rfc.py [streams [packets per stream [seed]]]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
o = oracle_py.load()
lib = C.CDLL(os.environ.get("OG_EMUL_LIB", os.path.join(ROOT, "tests", "emul", "libog_emul.so")))
lib.emu_state_size.restype = C.c_int
lib.emu_decode_frame_rfc.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
o.lib.oc_packet_parse.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


def dur(toc):
    if toc & 0x80:
        return (48000 << ((toc >> 3) & 3)) // 400
    if (toc & 0x60) == 0x60:
        return 960 if toc & 8 else 480
    a = (toc >> 3) & 3
    return 2880 if a == 3 else (48000 << a) // 100


def mode_bw(toc):
    if toc & 0x80:
        bw = 1102 + ((toc >> 5) & 3)
        return 1002, (1101 if bw == 1102 else bw)
    if (toc & 0x60) == 0x60:
        return 1001, (1105 if toc & 0x10 else 1104)
    return 1000, 1101 + ((toc >> 5) & 3)


STREAMS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
PACKETS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
n = bad = frames = 0
for s in range(STREAMS):
    channels = int(rng.integers(1, 3))
    d = o.decoder(channels); d.init(); d.set_rfc(True)
    st = C.create_string_buffer(lib.emu_state_size()); lib.emu_stream_init(st, channels)
    cfg = int(rng.integers(32))
    history, stream_bad = [], False
    for f in range(PACKETS):
        if rng.random() < 0.4:
            cfg = int(rng.integers(32))
        stereo = (channels == 2) if rng.random() < 0.85 else bool(rng.integers(2))
        code = int(rng.choice([0, 0, 0, 1, 2, 3]))
        toc = (cfg << 3) | (4 if stereo else 0) | code
        L = int(rng.choice([3, 8, 20, 40, 80, 120, 160, 300]))
        body = lambda k: rng.integers(0, 256, k, dtype=np.uint8).tobytes()
        if code == 0:
            pkt = bytes([toc]) + body(L)
        elif code == 1:
            pkt = bytes([toc]) + body(2 * L)
        elif code == 2:
            L = min(L, 250)
            pkt = bytes([toc, L]) + body(L + int(rng.integers(2, 120)))
        else:
            cnt = int(rng.integers(1, 5))
            while dur(toc) * cnt > 5760:
                cnt -= 1
            pkt = bytes([toc, cnt]) + body(cnt * L)
        ref, r = d.decode(pkt)
        ref = ref[:max(r, 0)].copy()
        history.append(hex(toc))
        # the host side of the product: frame the packet, then one device call per frame
        size = (C.c_int16 * 48)(); tocb = C.c_uint8(); off = C.c_int()
        cnt = o.lib.oc_packet_parse(pkt, len(pkt), 0, C.byref(tocb), size, C.byref(off), None)
        n += 1
        if cnt < 0 or cnt * dur(toc) > 5760:
            if r >= 0:
                bad += 1; print("framing disagrees", hex(toc), cnt, r)
            continue
        m, bw = mode_bw(toc)
        fs = dur(toc)
        out = np.zeros((cnt * fs, channels), dtype=np.int16)
        at, r2 = off.value, 0
        for k in range(cnt):
            buf = np.zeros((fs, channels), dtype=np.int16)
            rr = lib.emu_decode_frame_rfc(st, pkt[at:at + size[k]], size[k], m, bw, 2 if stereo else 1, buf.ctypes.data, fs)
            frames += 1
            if rr < 0:
                r2 = rr
                break
            out[k * fs:(k + 1) * fs] = buf
            r2 += rr
            at += size[k]
        ok = r == r2
        if ok and r > 0:
            pch = 2 if stereo else 1
            if m == 1000 and pch < channels:  # Q3: only the first fs * pch linear entries of a frame are defined
                for k in range(cnt):
                    a = out[k * fs:(k + 1) * fs].reshape(-1)[:fs * pch]
                    b = ref[k * fs:(k + 1) * fs].reshape(-1)[:fs * pch]
                    ok = ok and np.array_equal(a, b)
            else:
                ok = np.array_equal(out, ref)
        if not ok:
            bad += 1
            if not stream_bad and bad <= 40:
                print("MISMATCH stream", s, "packet", f, "toc", hex(toc), "len", len(pkt), "channels", channels, r, r2, "history", history)
            stream_bad = True
print(f"{n} packets, {frames} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
