#!/usr/bin/env python3
"""CPU parity fuzz of RFC mode: the kernel source in host emulation (decode_frame_rfc, og_decode.hpp) against the oracle's RFC
mode (oc_decoder_set_rfc) -- all 32 TOC configurations, mono and stereo packets in mono and stereo decoders, frame-count codes
0..3, configuration switches inside a stream (incl. hybrid -> SILK-only: the silence-frame fade-out), and the loss path: lost
packets (concealed for the duration of the stream's last packet, like a caller of opus_decode(NULL) would ask) and DTX frames
(at most one payload byte), and forward error correction: a third of the losses are not concealed when they happen but
recovered from the NEXT packet's LBRR data (opus_decode(decode_fec = 1)), which is then decoded normally.  Redundant CELT frames (RFC 6716 section 4.5.1): SILK-only frames with
random payloads carry one almost always; hybrid packets with the flag set come from tests/golden/rfc_hybrid_redundancy_seeds.json.
    python3 tools/fuzz_emul_rfc.py [streams [packets per stream [seed [loss probability]]]]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py
from rfc_common import dur, mode_bw, make_packet, frame_payloads, same_pcm, fec_plan, redundancy_packet
o = oracle_py.load()
lib = C.CDLL(os.environ.get("OG_EMUL_LIB", os.path.join(ROOT, "tests", "emul", "libog_emul.so")))
lib.emu_state_size.restype = C.c_int
lib.emu_decode_frame_rfc.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
lib.emu_decode_frame_rfc_fec.argtypes = lib.emu_decode_frame_rfc.argtypes
o.lib.oc_decode_fec.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int]

STREAMS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
PACKETS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
P_LOSS = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
n = bad = frames = lost = fec_calls = fec_used = redundant = 0


def run_frames(st, channels, payloads, fs, m, bw, pch):
    """The host side of the product in miniature: one device call per frame -> (samples or error, PCM [frames * fs, channels])."""
    global frames
    out = np.zeros((len(payloads) * fs, channels), dtype=np.int16)
    total = 0
    for k, pay in enumerate(payloads):
        buf = np.zeros((fs, channels), dtype=np.int16)
        rr = lib.emu_decode_frame_rfc(st, pay, len(pay), m, bw, pch, buf.ctypes.data, fs)
        frames += 1
        if rr < 0:
            return rr, out
        out[k * fs:(k + 1) * fs] = buf
        total += rr
    return total, out


for s in range(STREAMS):
    channels = int(rng.integers(1, 3))
    d = o.decoder(channels); d.init(); d.set_rfc(True)
    st = C.create_string_buffer(lib.emu_state_size()); lib.emu_stream_init(st, channels)
    cfg = int(rng.integers(32))
    history, stream_bad = [], False
    last = None  # (frame count, frame duration, mode, bandwidth, packet channels) of the last packet framed
    for f in range(PACKETS):
        before = d.prev_mode()
        if rng.random() < P_LOSS:  # a lost packet: conceal what the last packet carried (20 ms when there was none)
            cnt, fs, m, bw, pch = last if last else (1, 960, 1002, 1105, channels)
            pays, pkt, label = [b""] * cnt, None, "lost"
            ref, r = d.conceal(cnt * fs)
            lost += 1
        else:
            if rng.random() < 0.4:
                cfg = int(rng.integers(32))
            stereo = (channels == 2) if rng.random() < 0.85 else bool(rng.integers(2))
            L = int(rng.choice([3, 8, 20, 40, 80, 120, 160, 300])) if rng.random() >= 0.08 else int(rng.integers(0, 2))  # (DTX frames)
            pkt = make_packet(rng, cfg, stereo, int(rng.choice([0, 0, 0, 1, 2, 3])), L)
            label = hex(pkt[0]) + ("/%d" % L if L < 2 else "")
            if rng.random() < 0.06:  # a hybrid packet that carries a redundant CELT frame (random payloads almost never do)
                pkt, kind = redundancy_packet(rng, channels if rng.random() < 0.85 else None)
                stereo, label, L = bool(pkt[0] & 4), hex(pkt[0]) + ":" + kind, 99
                redundant += 1
            if rng.random() < P_LOSS / 2 and frame_payloads(o, pkt) is not None:
                # the packet before this one was lost and is recovered from this one's forward error correction data
                fec_calls += 1
                total, pieces, use = fec_plan((last[0], last[1], last[2]) if last else None, pkt[0], channels)
                before_f = d.prev_mode()
                lfs, lm, lbw, lpch = (last[1], last[2], last[3], last[4]) if last else (960, 1002, 1105, channels)
                ref = np.zeros((5760, channels), dtype=np.int16)
                r = o.lib.oc_decode_fec(d.h, pkt, len(pkt), ref.ctypes.data, total)
                got, at, r2 = np.zeros((total, channels), dtype=np.int16), 0, 0
                spans = []  # (start, duration, payload length, mode the frame is announced in, its channels)
                for w in pieces:
                    buf = np.zeros((w, channels), dtype=np.int16)
                    rr = lib.emu_decode_frame_rfc(st, b"", 0, lm, lbw, lpch, buf.ctypes.data, w)
                    frames += 1
                    assert rr == w, (rr, w)
                    got[at:at + w] = buf
                    spans.append((at, w, 0, lm, lpch))
                    at += w
                if use:
                    fec_used += 1
                    (m, bw), fs, pch = mode_bw(pkt[0]), dur(pkt[0]), 2 if stereo else 1
                    pay = frame_payloads(o, pkt)[0]
                    buf = np.zeros((fs, channels), dtype=np.int16)
                    rr = lib.emu_decode_frame_rfc_fec(st, pay, len(pay), m, bw, pch, buf.ctypes.data, fs)
                    frames += 1
                    got[at:at + fs] = buf
                    spans.append((at, fs, len(pay), m, pch))
                    at += fs
                    r2 = rr if rr < 0 else at
                else:
                    r2 = at
                okf = r == r2
                mode_now = before_f
                for (a0, w, ln, mm, pc) in spans if okf and r > 0 else []:
                    okk, _ = same_pcm(got[a0:a0 + w], ref[a0:a0 + w], w, [ln], mode_now, mm, pc, channels)
                    if ln > 1:
                        mode_now = mm
                    okf = okf and okk
                n += 1
                if not okf:
                    bad += 1
                    if not stream_bad and bad <= 40:
                        print("MISMATCH (fec) stream", s, "packet", f, label, "channels", channels, r, r2, "plan", total, pieces, use, "last", last, "history", history)
                    stream_bad = True
                history.append("fec:" + label)
                before = d.prev_mode()
            ref, r = d.decode(pkt)
            pays = frame_payloads(o, pkt)
            if pays is None:
                n += 1
                if r >= 0:
                    bad += 1; print("framing disagrees", label, r)
                continue
            (m, bw), fs, pch = mode_bw(pkt[0]), dur(pkt[0]), 2 if stereo else 1
            last = (len(pays), fs, m, bw, pch)
        ref = ref[:max(r, 0)].copy()
        history.append(label)
        n += 1
        r2, out = run_frames(st, channels, pays, fs, m, bw, pch)
        ok = r == r2
        if ok and r > 0:
            ok, k = same_pcm(out, ref, fs, [len(p) for p in pays], before, m, pch, channels)
        if not ok:
            bad += 1
            if not stream_bad and bad <= 40:
                print("MISMATCH stream", s, "packet", f, label, "channels", channels, r, r2, "last", last, "history", history)
            stream_bad = True
print(f"{n} packets ({lost} lost, {fec_calls} recovered with decode_fec: {fec_used} from LBRR data; {redundant} hybrid packets with redundancy), {frames} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
