#!/usr/bin/env python3
"""Regenerates tests/golden/*.json|npz.

Two kinds of fixture:
  survey_kats.json   -- outputs of the REFERENCE itself, recorded while it ran during the survey session
                        (KAT 1 and KAT 2).  Copied verbatim; this script only re-checks that
                        the oracle still reproduces them.  They are the parity pin of the oracle.
  oracle_vectors.*   -- vectors produced by the oracle (NOT by the reference) for regression and for checking the
                        GPU path on machines where the oracle library is not built: per-frame FNV-1a hashes of
                        64 streams x 16 frames per mode, and full PCM of 2 streams x 4 frames per mode.
                        Inputs are the LCG payloads of SURVEY.md section 8d (seed 0x9E3779B9 ^ stream id).
  oracle_sequences.json -- explicit packet sequences (hex) for the quirk cases, with the oracle's return code and PCM hash
                        per call: mono decoders, a mono packet in a stereo decoder (Q3: only the defined half is
                        hashed), multi-frame packets of every frame-count code with room for three frames (Q6), and a
                        mode-switch sequence incl. hybrid -> SILK-only (Q4) and CELT <-> SILK, and frames of 0 / 1 payload bytes
                        (CELT-only and hybrid: the reference's ERR_OPUS_CELT_BAD_ARG, -18).
  rfc_sequences.json -- the same for RFC mode (oracle/oc_opus.h oc_decoder_set_rfc; PARITY-UNPINNED: these vectors freeze what
                        the oracle does today, they do not come from any reference decoder): event sequences for a stereo
                        and a mono decoder over every frame duration and frame-count code -- packets, lost packets
                        (concealed for the last packet's duration), DTX frames, packets preceded by a forward error
                        correction recovery (oc_decode_fec), hybrid packets with redundancy -- return code and PCM hash."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py  # noqa: E402
from conftest import load_pkg  # noqa: E402

MODES = {"celt_fb_stereo": (0xFC, 160), "silk_nb_stereo": (0x0C, 40), "hybrid_fb_stereo": (0x7C, 120)}


CAP = 3  # room for three 20 ms frames per call in the sequences


def _bytes(rng, n):
    return rng.integers(0, 256, n, dtype=np.uint8).tobytes()


def _size(a):  # frame length field of code 2 / VBR code 3 (RFC 6716 section 3.2.1)
    return bytes([a]) if a < 252 else bytes([252 + (a & 3), (a - 252 - (a & 3)) >> 2])


def sequence_packets():
    """name -> (decoder channels, [packets])."""
    rng = np.random.default_rng(20260401)
    seqs = {}
    for name, toc, L in (("mono_celt_fb", 0xF8, 80), ("mono_silk_nb", 0x08, 30), ("mono_hybrid_fb", 0x78, 70)):
        seqs[name] = (1, [bytes([toc]) + _bytes(rng, L) for _ in range(6)])
    # Q3: mono packets in a stereo decoder, all three modes, between stereo packets
    seqs["mono_packets_in_stereo_decoder"] = (2, [bytes([t]) + _bytes(rng, 60) for t in (0xFC, 0xF8, 0x0C, 0x08, 0x0C, 0x7C, 0x78, 0xFC)])
    # Q6: every frame-count code; CELT, SILK and hybrid; a 10 ms configuration that the reference decodes as 20 ms
    mf = []
    for toc in (0xFC, 0x0C, 0x7C):
        mf.append(bytes([toc | 1]) + _bytes(rng, 2 * 50))                                   # code 1: two equal frames
        mf.append(bytes([toc | 2]) + _size(40) + _bytes(rng, 40 + 70))                      # code 2: 40 + 70
        mf.append(bytes([toc | 3, 3]) + _bytes(rng, 3 * 45))                                # code 3 CBR, 3 frames
        mf.append(bytes([toc | 3, 0x80 | 2]) + _size(30) + _bytes(rng, 30 + 55))            # code 3 VBR, 2 frames
        mf.append(bytes([toc | 3, 0x40 | 2, 5]) + _bytes(rng, 2 * 35) + bytes(5))           # code 3 CBR, padding
        mf.append(bytes([toc]) + _bytes(rng, 90))                                           # back to one frame
    mf.append(bytes([(30 << 3) | 4 | 1]) + _bytes(rng, 2 * 40))                             # CELT FB 10 ms x 2 (Q6)
    mf.append(bytes([0xFC | 3, 4]) + _bytes(rng, 4 * 30))                                   # four 20 ms frames: too many for the room
    mf.append(bytes([0xFC | 3, 0]) + _bytes(rng, 10))                                       # code 3 with 0 frames: invalid
    mf.append(bytes([0xFC]) + _bytes(rng, 100))
    seqs["multiframe_packets"] = (2, mf)
    # Q4 and friends: CELT -> hybrid -> SILK-only (the transition frame) -> hybrid -> CELT -> SILK -> CELT, varying bandwidth
    order = (0xFC, 0x7C, 0x0C, 0x0C, 0x6C, 0x2C, 0x7C, 0xFC, 0xDC, 0x4C, 0xFC, 0x7C, 0x4C, 0x0C)
    seqs["mode_switches"] = (2, [bytes([t]) + _bytes(rng, 50 + 7 * i) for i, t in enumerate(order)])
    # frames of 0 / 1 payload bytes: celt_decode_with_ec refuses them with ERR_OPUS_CELT_BAD_ARG = -18 (src/celt.cpp:2225,
    # src/opus_decoder.h:55) in CELT-only and hybrid mode, SILK-only decodes them (the range decoder runs out of bytes); ordinary
    # frames in between so that what such a frame leaves behind (prev_mode, SILK state of a refused hybrid frame) is pinned too
    tiny = []
    for toc in (0xFC, 0x7C, 0x0C):
        tiny += [bytes([toc]) + _bytes(rng, 40), bytes([toc]), bytes([toc]) + _bytes(rng, 1), bytes([toc, 0xFF]), bytes([toc]) + _bytes(rng, 2),
                 bytes([toc]) + _bytes(rng, 45)]
    seqs["tiny_frames"] = (2, tiny)
    # empty packets (len 0): the reference's opus_decode_native runs opus_decode_frame(NULL, 0) in the decoder's LAST mode, 960
    # samples per pass, until frame_size (here CAP x 960) is filled or a pass fails (src/opus_decoder.cpp:290-308): mode 0 before the
    # first packet (SILK runs, then CELT's -18), SILK-only decodes, hybrid advances SILK and ends in -18, CELT-only ends in -18;
    # ordinary packets in between pin what each case leaves behind
    e = [b""]
    for toc, L in ((0x0C, 40), (0x7C, 70), (0xFC, 80), (0x4C, 50), (0x08, 30), (0x6C, 60)):
        e += [bytes([toc]) + _bytes(rng, L), b"", bytes([toc]) + _bytes(rng, L)]
    e += [b"", b""]
    seqs["empty_packets"] = (2, e)
    seqs["empty_packets_mono"] = (1, [b"", bytes([0x08]) + _bytes(rng, 30), b"", bytes([0x0C]) + _bytes(rng, 40), b"", bytes([0x78]) + _bytes(rng, 60), b"",
                                      bytes([0x08]) + _bytes(rng, 35)])
    return seqs


def sequences(o):
    out = {"generator": "oracle (oracle/liboc_oracle.so), NOT the reference", "frame_capacity": CAP, "sequences": {}}
    for name, (channels, packets) in sequence_packets().items():
        d = o.decoder(channels)
        d.init()
        calls = []
        toc = 0x7C  # (an empty packet decodes in the mode of the last one; before the first: like hybrid)
        for p in packets:
            pcm, r = d.decode_cap(p, CAP)
            toc = p[0] if p else toc
            silk_mono_in_stereo = channels == 2 and not (toc & 0x80) and (toc & 0x60) != 0x60 and not (toc & 4)
            h = None
            if r > 0 and not silk_mono_in_stereo:
                h = oracle_py.fnv1a_u16(pcm[:r])
            calls.append({"packet": p.hex(), "ret": int(r), "fnv1a_u16": h})
        out["sequences"][name] = {"channels": channels, "calls": calls}
    return out


def rfc_events(channels):
    """[(kind, packet or None)]: kind in packet / lost / fec (fec: recover the packet lost before this one from it, then decode it)"""
    from rfc_common import make_packet, redundancy_packet
    rng = np.random.default_rng(20261004 + channels)
    stereo = channels == 2
    ev = []
    walk = [31, 31, 29, 23, 19, 17, 16, 31, 15, 15, 13, 12, 14, 15, 1, 0, 2, 3, 9, 8, 10, 11, 5, 7, 15, 9, 31, 15, 1, 27]
    for k, cfg in enumerate(walk):
        code = (0, 0, 1, 2, 3)[k % 5]
        ev.append(("packet", make_packet(rng, cfg, stereo, code, int(rng.choice([20, 40, 80, 120])))))
        if k % 3 == 2:
            ev.append(("lost", None))
        if k % 7 == 6:
            ev.append(("lost", None))  # (two in a row with the one above, now and then)
        if k % 4 == 1:
            ev.append(("fec", make_packet(rng, cfg if cfg < 16 else 9, stereo, 0, 100)))
        if k % 9 == 4:
            ev.append(("packet", make_packet(rng, cfg, stereo, 0, int(rng.integers(0, 2)))))  # a DTX frame
        if k % 6 == 3:
            ev.append(("packet", redundancy_packet(rng, channels)[0]))
    return ev


def rfc_sequences(o):
    import ctypes as C
    from rfc_common import fec_plan, dur, mode_bw, frame_payloads
    o.lib.oc_decode_fec.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int]
    out = {"generator": "oracle RFC mode (oc_decoder_set_rfc), NOT the reference and NOT pinned by it", "frame_capacity": 6, "sequences": {}}
    for channels in (2, 1):
        d = o.decoder(channels)
        d.init()
        d.set_rfc(True)
        last, calls = None, []
        for kind, p in rfc_events(channels):
            if kind == "lost":
                want = last[0] * last[1] if last else 960
                pcm, r = d.conceal(want)
            elif kind == "fec":
                total, _, _ = fec_plan((last[0], last[1], last[2]) if last else None, p[0], channels)
                buf = np.zeros((5760, channels), dtype=np.int16)
                r = o.lib.oc_decode_fec(d.h, p, len(p), buf.ctypes.data, total)
                calls.append({"kind": "fec", "packet": p.hex(), "ret": int(r), "fnv1a_u16": oracle_py.fnv1a_u16(buf[:r]) if r > 0 else None})
                pcm, r = d.decode(p)
                kind = "packet"
            else:
                pcm, r = d.decode(p)
            if kind == "packet" and frame_payloads(o, p) is not None:
                last = (len(frame_payloads(o, p)), dur(p[0]), mode_bw(p[0])[0])
            calls.append({"kind": kind, "packet": p.hex() if p else None, "ret": int(r),
                          "fnv1a_u16": oracle_py.fnv1a_u16(pcm[:r]) if r > 0 else None})
        out["sequences"]["stereo" if channels == 2 else "mono"] = {"channels": channels, "calls": calls}
    return out


def main():
    o = oracle_py.load()
    pkg = load_pkg()
    hashes, pcm = {}, {}
    for name, (toc, L) in MODES.items():
        pay = pkg.lcg_payloads(64, 16, L)
        ref, ok = o.batch_decode(2, toc, pay)
        assert ok == 64 * 16
        hashes[name] = {"toc": toc, "payload_len": L, "streams": 64, "frames": 16,
                        "fnv1a_u16": [[oracle_py.fnv1a_u16(ref[s, f]) for f in range(16)] for s in range(64)]}
        pcm[name] = ref[:2, :4].copy()
    json.dump({"generator": "oracle (oracle/liboc_oracle.so), NOT the reference", "payloads": "lcg_payloads(seed 0x9E3779B9 ^ stream)",
               "modes": hashes}, open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors_pcm.npz"), **pcm)
    json.dump(sequences(o), open(os.path.join(HERE, "oracle_sequences.json"), "w"), indent=0)
    json.dump(rfc_sequences(o), open(os.path.join(HERE, "rfc_sequences.json"), "w"), indent=0)
    print("wrote oracle_vectors.json / oracle_vectors_pcm.npz / oracle_sequences.json / rfc_sequences.json")


if __name__ == "__main__":
    main()
