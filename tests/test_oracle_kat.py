"""The oracle is pinned by the outputs of the REFERENCE recorded in SURVEY.md appendix B (KAT 1, KAT 2).

The reference itself cannot be built in this image (it needs <Arduino.h>; writing a stand-in header is not
allowed), and it has no tests of its own, so these recorded hashes / samples are the parity pin."""
import numpy as np

from oracle_py import fnv1a_u16

TOCS = (0xFC, 0x0C, 0x7C)  # CELT-FB, SILK-NB, Hybrid-FB; 20 ms stereo


def _lcg_bytes(state, n):
    out = bytearray()
    for _ in range(n):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        out.append(state >> 24)
    return state, bytes(out)


def test_kat1_continuous_lcg_reset_between_modes(oracle):
    """One decoder, global LCG seed 12345 running across the three modes, 50 packets of 161 bytes per mode,
    OPUS_RESET_STATE (NOT a full reset, Q5) before SILK and hybrid."""
    expect = [(0x818F2314, (198, 372)), (0x9B451028, (4300, 5469)), (0x729184E0, (-1227, -791))]
    d = oracle.decoder(2)
    lcg = 12345
    for m, toc in enumerate(TOCS):
        if m > 0:
            d.reset()
        h = 0
        for f in range(50):
            lcg, body = _lcg_bytes(lcg, 160)
            out, r = d.decode(bytes([toc]) + body)
            assert r == 960
            h ^= (fnv1a_u16(out[:960]) + f) & 0xFFFFFFFF
        assert h == expect[m][0], (m, hex(h))
        flat = out[:960].reshape(-1)
        assert (int(flat[100]), int(flat[101])) == expect[m][1]


def test_kat2_fresh_state_per_mode(oracle):
    """Fresh decoder state per mode, per-stream LCG seed 999, 5 packets, FNV-1a chained over all frames."""
    expect = [0x165E980F, 0xD19FF868, 0xCFEE645B]
    d = oracle.decoder(2)
    for m, toc in enumerate(TOCS):
        d.init()
        x, h = 999, 2166136261
        for f in range(5):
            x, body = _lcg_bytes(x, 160)
            out, r = d.decode(bytes([toc]) + body)
            assert r == 960
            h = fnv1a_u16(out[:960], h)
        assert h == expect[m], (m, hex(h))


# celt_decode_with_ec refuses a frame of <= 1 byte with ERR_OPUS_CELT_BAD_ARG = -18 (reference src/celt.cpp:2225, enum
# src/opus_decoder.h:55), opus_decode_frame returns `celt_ret < 0 ? celt_ret : audiosize` (src/opus_decoder.cpp:277) and
# opus_decode_native passes it up (:333-337).  SILK-only frames never reach CELT (the 2.5 ms transition frame's result is ignored,
# :265-268).  Derived by hand from those lines, not read off the oracle.
TINY_KAT = [(bytes([0xFC]), -18), (bytes([0xFC, 0xFF]), -18), (bytes([0xFC, 0x00]), -18), (bytes([0x7C]), -18), (bytes([0x7C, 0xFF]), -18),
            (bytes([0x7C, 0x00]), -18), (bytes([0x0C]), 960), (bytes([0x0C, 0xFF]), 960), (bytes([0xF8]), -18), (bytes([0x78, 0x55]), -18),
            (bytes([0x08]), 960)]


def test_tiny_celt_and_hybrid_frames_return_the_reference_code(oracle):
    for channels in (2, 1):
        for pkt, want in TINY_KAT:
            d = oracle.decoder(channels)
            d.init()
            assert d.decode(pkt)[1] == want, (channels, pkt.hex())
            assert d.decode(bytes([pkt[0]]) + bytes(range(40)))[1] == 960  # the decoder goes on after a refused frame


def test_threaded_batch_decode_matches_single_thread(pkg, oracle):
    """The threaded helper the full-size GPU tests rely on returns exactly what one thread returns (streams are
    independent; each thread owns a stream range)."""
    for toc, L in ((pkg.TOC_CELT_FB_STEREO, 160), (pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120)):
        pay = pkg.lcg_payloads(37, 3, L)
        one, ok1 = oracle.batch_decode(2, toc, pay)
        many, okn = oracle.batch_decode_threads(2, toc, pay, threads=5)
        assert ok1 == okn == 37 * 3
        assert (one == many).all()


def test_rfc_batch_matches_one_decoder_per_stream(oracle):
    """oc_batch_decode_rfc (bench.py's rfc workload: checker and CPU baseline) against the same calls made one by one: every
    configuration, lost packets concealed, lost packets recovered from the next packet's forward error correction data"""
    import ctypes as C
    from rfc_common import dur, make_packet
    rng = np.random.default_rng(5)
    n, F = 64, 6
    pk = [[make_packet(rng, s % 32, True, 0, int(rng.choice([20, 60, 120]))) for s in range(n)] for _ in range(F + 1)]
    ops = (rng.random((F, n)) < 0.25).astype(np.uint8)
    ops[0] = 0
    ops[(ops == 1) & (rng.random((F, n)) < 0.5)] = 2
    arena, offs, lens = bytearray(), np.zeros((F, n), dtype=np.int64), np.zeros((F, n), dtype=np.int32)
    for f in range(F):
        for s in range(n):
            p = pk[f + 1][s] if ops[f, s] == 2 else (b"" if ops[f, s] == 1 else pk[f][s])
            offs[f, s], lens[f, s] = len(arena), len(p)
            arena += p
    pcm, rets = oracle.batch_decode_rfc(2, np.frombuffer(bytes(arena), dtype=np.uint8), offs, lens, ops, threads=3)
    oracle.lib.oc_decode_fec.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int]
    for s in range(n):
        d = oracle.decoder(2)
        d.init()
        d.set_rfc(True)
        last = 960
        for f in range(F):
            if ops[f, s] == 1:
                out, r = d.conceal(last)
            elif ops[f, s] == 2:
                r = oracle.lib.oc_decode_fec(d.h, pk[f + 1][s], len(pk[f + 1][s]), d.buf.ctypes.data, last)
                out = d.buf
            else:
                out, r = d.decode(pk[f][s])
                last = r if r > 0 else last
            assert rets[s, f] == r, (s, f, int(ops[f, s]))
        if r > 0:
            assert r == dur(pk[0][s][0]) and np.array_equal(pcm[s, :r], out[:r]), s
