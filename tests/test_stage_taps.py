"""Stage-level parity: the kernel source in host emulation against the oracle, not only at the PCM but at
the intermediate arrays of the CELT path -- the decoded normalised spectrum X, the band energies, the IMDCT output
before the comb filter and the comb filter's output -- frame by frame, state carried across frames.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    lib = C.CDLL(os.path.join(ROOT, "tests", "emul", "libog_emul.so"))
    lib.emu_state_size.restype = C.c_int
    lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
    lib.emu_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.emu_tap_X.restype = C.POINTER(C.c_int16)
    lib.emu_tap_bandE.restype = C.POINTER(C.c_int16)
    lib.emu_tap_syn_pre.restype = C.POINTER(C.c_int32)
    lib.emu_tap_syn_pre.argtypes = [C.c_int]
    lib.emu_tap_syn_post.restype = C.POINTER(C.c_int32)
    lib.emu_tap_syn_post.argtypes = [C.c_int]
    return lib


def _oracle_taps(oracle, d, what, c, dtype, count):
    buf = np.zeros(count, dtype=dtype)
    oracle.lib.oc_taps_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    n = oracle.lib.oc_taps_copy(d.h, what, c, buf.ctypes.data)
    assert n == buf.nbytes, (what, n, buf.nbytes)
    return buf


@pytest.mark.parametrize("L", [160, 60, 400])
def test_celt_stage_taps_match(pkg, oracle, emu, L):
    n, frames = 24, 4
    pay = pkg.lcg_payloads(n, frames, L, seed_base=0x51A6E + L)
    out = np.zeros((960, 2), dtype=np.int16)
    oracle.lib.oc_taps_enable.argtypes = [C.c_void_p]
    for s in range(n):
        d = oracle.decoder(2)
        d.init()
        assert oracle.lib.oc_taps_enable(d.h)
        st = C.create_string_buffer(emu.emu_state_size())
        emu.emu_stream_init(st, 2)
        for f in range(frames):
            pkt = bytes([pkg.TOC_CELT_FB_STEREO]) + pay[f, s].tobytes()
            ref, r = d.decode(pkt)
            assert r == 960
            assert emu.emu_decode_frame(st, pkt[1:], L, 1002, 1105, 2, out.ctypes.data) == 960
            assert (out == ref[:960]).all(), (s, f)
            where = (L, s, f)
            x = np.ctypeslib.as_array(emu.emu_tap_X(), shape=(1920,))
            assert np.count_nonzero(x) > 100, ("the spectrum tap carries no signal", where)  # not a vacuous comparison
            assert (x == _oracle_taps(oracle, d, 0, 0, np.int16, 1920)).all(), ("X", where)
            e = np.ctypeslib.as_array(emu.emu_tap_bandE(), shape=(42,))
            assert (e == _oracle_taps(oracle, d, 1, 0, np.int16, 42)).all(), ("bandE", where)
            for c in range(2):
                pre = np.ctypeslib.as_array(emu.emu_tap_syn_pre(c), shape=(1080,))
                assert np.count_nonzero(pre) > 500, ("the IMDCT tap carries no signal", c, where)
                # (the transform defines 60 words of overlap tail + 960 outputs; the 60 words behind them are whatever the
                # emulation's working set held before -- other tests of this process decode other frames first)
                assert (pre[:1020] == _oracle_taps(oracle, d, 2, c, np.int32, 1080)[:1020]).all(), ("IMDCT output", c, where)
                post = np.ctypeslib.as_array(emu.emu_tap_syn_post(c), shape=(1080,))[:960]
                assert (post == _oracle_taps(oracle, d, 3, c, np.int32, 960)).all(), ("comb filter output", c, where)


@pytest.mark.parametrize("toc, L, channels", [(0x0C, 40, 2), (0x2C, 50, 2), (0x4C, 70, 2), (0x08, 30, 1), (0x48, 60, 1),
                                              (0x7C, 120, 2), (0x6C, 100, 2), (0x78, 90, 1)])
def test_silk_stage_taps_match(pkg, oracle, emu, toc, L, channels):
    """SILK stage values of the kernel source (host emulation) against the oracle's, per coded channel and frame: signal
    type, gains, pitch lags, both sets of LPC coefficients, LTP taps and scale, and the synthesis core's output at the
    internal rate (before stereo un-mixing and resampling).  SILK-only NB / MB / WB and hybrid, stereo and mono."""
    emu.emu_tap_silk.argtypes = [C.c_int, C.c_int, C.c_void_p]
    oracle.lib.oc_silk_taps_copy.argtypes = [C.c_int, C.c_int, C.c_void_p]
    oracle.lib.oc_silk_taps_enable.argtypes = [C.c_int]
    oracle.lib.oc_silk_taps_enable(1)

    def both(what, ch, dtype, count):
        a, b = np.zeros(count, dtype=dtype), np.zeros(count, dtype=dtype)
        assert emu.emu_tap_silk(what, ch, a.ctypes.data) >= 0
        assert oracle.lib.oc_silk_taps_copy(what, ch, b.ctypes.data) >= 0
        return a, b

    m = 1000 if (toc & 0x60) != 0x60 else 1001
    bw = 1101 + ((toc >> 5) & 3) if m == 1000 else (1105 if toc & 0x10 else 1104)
    n, frames = 16, 5
    pay = pkg.lcg_payloads(n, frames, L, seed_base=0x51 + toc)
    out = np.zeros((960, channels), dtype=np.int16)
    voiced = coded_frames = 0
    try:
        for s in range(n):
            d = oracle.decoder(channels)
            d.init()
            st = C.create_string_buffer(emu.emu_state_size())
            emu.emu_stream_init(st, channels)
            for f in range(frames):
                pkt = bytes([toc]) + pay[f, s].tobytes()
                ref, r = d.decode(pkt)
                assert r == 960
                assert emu.emu_decode_frame(st, pkt[1:], L, m, bw, 2 if toc & 4 else 1, out.ctypes.data) == 960
                assert (out == ref[:960]).all(), (s, f)
                for ch in range(2 if toc & 4 else 1):
                    where = (hex(toc), s, f, ch)
                    sa, sb = both(0, ch, np.int32, 6)
                    assert bool(sa[0]) == bool(sb[0]), ("coded", where)
                    if not sb[0]:
                        continue
                    coded_frames += 1
                    flen, order = int(sb[3]), int(sb[4])
                    assert (sa[1], sa[2], sa[5]) == (sb[1], sb[2], sb[5]), ("signal type / offset type / LTP scale", where, sa, sb)
                    voiced += int(sb[1] == 2)
                    a, b = both(1, ch, np.int32, 8)
                    assert (a == b).all(), ("pitch lags / gains", where, a, b)
                    a, b = both(2, ch, np.int16, 32)
                    assert (a.reshape(2, 16)[:, :order] == b.reshape(2, 16)[:, :order]).all(), ("LPC coefficients", where)
                    a, b = both(3, ch, np.int16, 20)
                    assert (a == b).all(), ("LTP coefficients", where)
                    a, b = both(4, ch, np.int16, 320)
                    assert np.count_nonzero(b[:flen]) > flen // 4, ("the core output tap carries no signal", where)
                    assert (a[:flen] == b[:flen]).all(), ("core output", where)
    finally:
        oracle.lib.oc_silk_taps_enable(0)
    assert coded_frames >= n * frames and voiced > 0  # not a vacuous comparison: coded channels, some of them voiced
