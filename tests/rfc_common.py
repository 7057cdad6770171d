"""Shared by the RFC-mode parity tests (tests/test_emul_rfc.py via tools/fuzz_emul_rfc.py, tests/test_gpu_rfc.py): packet
generation over all TOC configurations, framing, and WHICH output entries a packet defines -- the comparison domain."""
import ctypes as C
import json
import os

import numpy as np

MODE_SILK, MODE_HYBRID, MODE_CELT = 1000, 1001, 1002


def dur(toc):
    """samples per frame at 48 kHz named by the TOC (RFC 6716 section 3.1)"""
    if toc & 0x80:
        return (48000 << ((toc >> 3) & 3)) // 400
    if (toc & 0x60) == 0x60:
        return 960 if toc & 8 else 480
    a = (toc >> 3) & 3
    return 2880 if a == 3 else (48000 << a) // 100


def mode_bw(toc):
    if toc & 0x80:
        bw = 1102 + ((toc >> 5) & 3)
        return MODE_CELT, (1101 if bw == 1102 else bw)
    if (toc & 0x60) == 0x60:
        return MODE_HYBRID, (1105 if toc & 0x10 else 1104)
    return MODE_SILK, 1101 + ((toc >> 5) & 3)


def make_packet(rng, cfg, stereo, code, L):
    """One packet of configuration cfg (0..31) with frame-count code 0..3 and random payload; L: bytes per frame (0 or 1: DTX)."""
    toc = (cfg << 3) | (4 if stereo else 0) | code
    body = lambda k: rng.integers(0, 256, k, dtype=np.uint8).tobytes()
    if code == 0:
        return bytes([toc]) + body(L)
    if code == 1:
        return bytes([toc]) + body(2 * L)
    if code == 2:
        L = min(L, 250)
        return bytes([toc, L]) + body(L + int(rng.integers(0, 120)))
    cnt = int(rng.integers(1, 5))
    while dur(toc) * cnt > 5760:
        cnt -= 1
    return bytes([toc, cnt]) + body(cnt * L)


def frame_payloads(oracle, pkt):
    """-> list of the packet's frame payloads (opus_packet_parse_impl through the oracle), or None when the framing fails."""
    lib = oracle.lib
    lib.oc_packet_parse.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    size = (C.c_int16 * 48)()
    tocb, off = C.c_uint8(), C.c_int()
    cnt = lib.oc_packet_parse(pkt, len(pkt), 0, C.byref(tocb), size, C.byref(off), None)
    if cnt < 0 or cnt * dur(pkt[0]) > 5760:
        return None
    out, at = [], off.value
    for k in range(cnt):
        out.append(pkt[at:at + size[k]])
        at += size[k]
    return out


def same_pcm(got, want, fs, frame_lens, prev_mode, toc_mode, pch, channels):
    """Compare a packet's PCM ([frames * fs, channels] each) over the entries the decoder defines.  A frame of at most one
    byte -- and every frame of a lost packet (frame_lens all 0) -- is concealed in the mode of the frame before it; the others
    run in the TOC's mode.  Q3: a SILK-only frame with fewer channels than the decoder defines only its first fs * pch LINEAR
    entries -- of every 20 ms chunk when a longer frame is concealed (the concealment goes 20 ms at a time)."""
    mode = prev_mode
    for k, ln in enumerate(frame_lens):
        concealed = ln <= 1
        if not concealed:
            mode = toc_mode
        a, b = got[k * fs:(k + 1) * fs].reshape(-1), want[k * fs:(k + 1) * fs].reshape(-1)
        if mode == MODE_SILK and pch < channels:
            if concealed and fs > 960:
                keep = np.concatenate([np.arange(c * 960 * channels, c * 960 * channels + 960 * pch) for c in range(fs // 960)])
                a, b = a[keep], b[keep]
            else:
                a, b = a[:fs * pch], b[:fs * pch]
        if not np.array_equal(a, b):
            return False, k
    return True, -1


def conceal_pieces(lost_dur, last_fs):
    """How a concealment of lost_dur samples is cut into device frames (valid duration codes): a frame of the last packet's size
    at a time like opus_decode(NULL), what is left over (30 / 50 ms) as 20 / 40 ms + 10 ms."""
    out, rem = [], lost_dur
    while rem > 0:
        w = min(rem, last_fs)
        rem -= w
        while w > 0:
            piece = next(v for v in (2880, 1920, 960, 480, 240, 120) if v <= w)
            out.append(piece)
            w -= piece
    return out


def fec_plan(last, toc, channels):
    """RFC 6716's opus_decode(decode_fec = 1) for the packet with this TOC after a lost packet; last = (frames, frame duration,
    mode) of the stream's last packet or None.  -> (samples to produce, [concealment frame durations], use the FEC frame?)"""
    lost_dur, last_fs, last_mode = (last[0] * last[1], last[1], last[2]) if last else (960, 120, 0)
    pfs, pmode = dur(toc), mode_bw(toc)[0]
    if lost_dur < pfs or pmode == MODE_CELT or last_mode == MODE_CELT:
        return lost_dur, conceal_pieces(lost_dur, last_fs), False
    return lost_dur, conceal_pieces(lost_dur - pfs, last_fs), True


_REDUNDANT = None


def redundancy_packet(rng, channels_pref=None):
    """A hybrid packet whose redundancy flag is set (tests/golden/rfc_hybrid_redundancy_seeds.json, found by
    tools/find_redundancy_seeds.py: the flag is one range-coded bit of probability 2^-12, random payloads almost never set it).
    -> (packet bytes, kind)"""
    global _REDUNDANT
    if _REDUNDANT is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rfc_hybrid_redundancy_seeds.json")
        _REDUNDANT = json.load(open(path))
    pool = [e for e in _REDUNDANT if channels_pref is None or bool(e["toc"] & 4) == (channels_pref == 2)] or _REDUNDANT
    e = pool[int(rng.integers(len(pool)))]
    pay = np.random.default_rng(e["seed"]).integers(0, 256, e["len"], dtype=np.uint8).tobytes()
    return bytes([e["toc"]]) + pay, e["kind"]
