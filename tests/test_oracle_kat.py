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


def test_threaded_batch_decode_matches_single_thread(pkg, oracle):
    """The threaded helper the full-size GPU tests rely on returns exactly what one thread returns (streams are
    independent; each thread owns a stream range)."""
    for toc, L in ((pkg.TOC_CELT_FB_STEREO, 160), (pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120)):
        pay = pkg.lcg_payloads(37, 3, L)
        one, ok1 = oracle.batch_decode(2, toc, pay)
        many, okn = oracle.batch_decode_threads(2, toc, pay, threads=5)
        assert ok1 == okn == 37 * 3
        assert (one == many).all()
