"""The one frame whose PCM and range-decoder state can be DERIVED BY HAND from the reference's source lines: a CELT-only frame whose
payload is all 0xFF -- the silence flag.  (Known answers read off the oracle pin nothing: oracle and kernels share an author, DESIGN.md
section 5; tests/test_return_codes.py does the same for the packet layer's return values.)

The derivation, for a payload of n >= 2 bytes 0xFF behind a CELT-only 20 ms TOC:
  * ec_dec_init (src/celt.cpp:2627-2640): rng = 128, rem = 0xFF, val = 127 - (0xFF >> 1) = 0; ec_dec_normalize (:2648-2663) runs
    three times (rng 2^7 -> 2^15 -> 2^23 -> 2^31), each time sym = (0xFF << 8 | next byte) >> 1 with a low byte of 0xFF (next byte
    0xFF) or 0x80 (past the end: ec_read_byte gives 0, :2642), so val grows by (255 & ~sym) = 0x00 resp. 0x7F per pass: val < 2^16.
    nbits_total = 9 + 24 = 33, ec_tell = 33 - 32 = 1.
  * celt_decode_with_ec (:2241-2252): tell == 1 -> silence = ec_dec_bit_logp(15): s = 2^31 >> 15 = 2^16 > val -> 1, rng = 2^16, one
    more normalisation: rng = 2^24.  The frame then pretends every bit is read (tell = 8 n), so no other symbol is decoded: every
    later read is guarded by the bits left (:2257, :2271, :2284, unquant_coarse_energy's budget, tf_decode, :2299, :2316, :2330;
    the allocation starts from bits = -1) -- the decoder's range stays 2^24.
  * (:2373-2376, denormalise_bands :958-961) silence: band energies -28 dB, bound = start = end = 0: the spectrum is all zeros; an
    IMDCT of zeros onto a fresh decoder's zero overlap, a comb filter with all gains 0 and a de-emphasis with zero memory leave
    zeros: the PCM is 960 x channels zeros, and the state is what it was (except the energies), so the next such frame gives
    zeros again.
  * celt_decode_with_ec stores that range (:2436, s_celtDec->rng = s_ec.rng: what celt_decoder_ctl(OPUS_GET_FINAL_RANGE) reports,
    :2513-2517); opus_decode_frame returns 960.  The Opus decoder's own OPUS_GET_FINAL_RANGE is another matter: it returns
    OpusDecoder::rangeFinal (src/opus_decoder.cpp:375-380), a field that is declared (:58), cleared with the struct (:90) and NEVER
    ASSIGNED -- opus_decode_frame ends at :276 without RFC 6716's `rangeFinal = dec.rng ^ redundant_rng` -- so that ctl reports 0
    after any packet.  Both are asserted below: the range decoder's state (oracle tap, opusgpu_stream_state_get) is 2^24, the ctl
    of include/opus_decoder.h is 0.
"""
import numpy as np
import pytest

FF = lambda n: b"\xff" * n
# CELT-only 20 ms TOCs: NB, WB, SWB, FB x mono / stereo (configs 19, 23, 27, 31)
TOCS = [0x98, 0x9C, 0xB8, 0xBC, 0xD8, 0xDC, 0xF8, 0xFC]
FINAL_RANGE = 1 << 24


def test_oracle_silence_frame_known_answer(oracle):
    for toc in TOCS:
        for channels in (2, 1):
            for n in (2, 3, 4, 7, 40):
                d = oracle.decoder(channels)
                d.init()
                for _ in range(3):
                    d.buf[:] = 0x5A5A
                    r = oracle.lib.oc_decode(d.h, bytes([toc]) + FF(n), 1 + n, d.buf.ctypes.data, 960)
                    assert r == 960, (hex(toc), channels, n, r)
                    assert not d.buf[:960].any(), (hex(toc), channels, n)
                    assert oracle.lib.oc_decoder_final_range(d.h) == FINAL_RANGE, (hex(toc), channels, n)
                    assert oracle.lib.oc_decoder_ctl_final_range(d.h) == 0
    # two such frames in a code-1 packet: 1920 samples of zeros
    d = oracle.decoder(2)
    d.init()
    d.buf[:] = 0x5A5A
    assert oracle.lib.oc_decode(d.h, b"\xfd" + FF(16), 17, d.buf.ctypes.data, 1920) == 1920 and not d.buf[:1920].any()


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [2, 1])
def test_gpu_silence_frame_known_answer(pkg, gpu_ctx, channels):
    """through opusgpu_decode_packets (host framing, k_celt_parse, k_celt_recon_fb, k_celt_post) and, for the final range,
    opusgpu_stream_state_get"""
    import ctypes as C
    rows = [(toc, n) for toc in TOCS for n in (2, 3, 4, 7, 40)]
    gpu_ctx.streams_alloc(len(rows) + 1, channels)
    for _ in range(3):
        pcm, res = gpu_ctx.decode_packets(np.arange(len(rows) + 1), [bytes([toc]) + FF(n) for toc, n in rows] + [b"\xfd" + FF(16)], frame_capacity=2)
        pcm = np.asarray(pcm)
        assert (np.asarray(res)[:-1] == 960).all() and res[-1] == 1920, res
        assert not pcm[:-1, :960].any() and not pcm[-1, :1920].any()
        head = (C.c_int32 * 4)()
        for s in range(len(rows) + 1):
            assert gpu_ctx.lib.opusgpu_stream_state_get(gpu_ctx.h, s, head, 16) == 0
            assert (head[3] & 0xFFFFFFFF) == FINAL_RANGE, (s, hex(head[3] & 0xFFFFFFFF))


@pytest.mark.gpu
def test_opus_decoder_h_silence_frame_known_answer(tmp_path):
    """through opus_decode / opus_multistream_decode of include/opus_decoder.h (the GPU behind it); OPUS_GET_FINAL_RANGE there is the
    reference's: 0"""
    import compat_util
    steps = []
    for toc in TOCS:
        steps += [("R",)] + [("D", 960, bytes([toc]) + FF(n)) for n in (2, 7, 40)] + [("F",)]
    steps += [("R",), ("D", 1920, b"\xfd" + FF(16)), ("F",)]
    got = compat_util.run(tmp_path, steps)
    for s, g in zip(steps, got):
        if s[0] == "D":
            assert g[0] == g[1] == s[1], (s, g[:2])
            assert not g[2].any(), s
        elif s[0] == "F":
            assert g == (0, 0), [hex(x) for x in g]  # (never assigned in the reference: see the module's docstring)


@pytest.mark.gpu
def test_gpu_final_range_matches_the_oracle_in_every_mode(pkg, gpu_ctx, oracle):
    """OPUS_GET_FINAL_RANGE (src/opus_decoder.cpp:375-380: the range decoder's last range, what conformance tools compare) after
    every packet of random streams in each mode, GPU against the oracle"""
    import ctypes as C
    rng = np.random.default_rng(77)
    for toc in (0xFC, 0xF8, 0x0C, 0x08, 0x2C, 0x4C, 0x6C, 0x7C, 0x78):
        for channels in (2, 1):
            ns = 24
            gpu_ctx.streams_alloc(ns, channels)
            decs = [oracle.decoder(channels) for _ in range(ns)]
            for d in decs:
                d.init()
            head = (C.c_int32 * 4)()
            for k in range(6):
                pkts = [bytes([toc]) + rng.integers(0, 256, size=int(rng.integers(2, 120)), dtype=np.uint8).tobytes() for _ in range(ns)]
                _, res = gpu_ctx.decode_packets(np.arange(ns), pkts)
                for s in range(ns):
                    r = oracle.lib.oc_decode(decs[s].h, pkts[s], len(pkts[s]), decs[s].buf.ctypes.data, 960)
                    assert r == res[s], (hex(toc), channels, k, s, r, int(res[s]))
                    assert gpu_ctx.lib.opusgpu_stream_state_get(gpu_ctx.h, s, head, 16) == 0
                    assert (head[3] & 0xFFFFFFFF) == oracle.lib.oc_decoder_final_range(decs[s].h), (hex(toc), channels, k, s)


@pytest.mark.gpu
def test_opus_decoder_h_ctls_the_reference_answers_oddly(tmp_path, oracle):
    """Two ctls of include/opus_decoder.h whose answers follow from reading the reference's lines, not from RFC 6716's decoder:
    OPUS_GET_FINAL_RANGE is 0 after ANY packet (rangeFinal is never assigned, src/opus_decoder.cpp:58, :90, :375-380), in both
    decoders; OPUS_GET_PITCH (:399-407) is OPUS_UNIMPLEMENTED after a CELT-only frame (the pointer goes to celt_decoder_ctl as the
    request number), else the SILK decoder's last exported lag at 48 kHz (src/silk.cpp:1764-1769) -- which OPUS_RESET_STATE does not
    clear -- and OPUS_UNIMPLEMENTED through the multistream ctl (not among the requests it passes on, :945-1026).  The pitch values
    against the oracle driven the same way; the rest are known answers."""
    import compat_util
    rng = np.random.default_rng(99)
    P = lambda toc, n: bytes([toc]) + rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    steps = [("P",)]                                                   # before any packet: no CELT frame yet -> a value (0)
    for toc in (0x0C, 0x4C, 0x7C, 0x0C, 0x0C, 0x2C, 0x7C):             # SILK-only NB / WB, hybrid: some frames voiced
        for _ in range(4):
            steps += [("D", 960, P(toc, 60)), ("P",), ("F",)]
    steps += [("D", 960, P(0xFC, 60)), ("P",), ("F",)]                 # after a CELT-only frame: OPUS_UNIMPLEMENTED
    steps += [("D", 960, P(0x4C, 60)), ("P",), ("R",), ("P",)]         # a reset keeps the last exported lag
    steps += [("D", 960, P(0x0C, 50)), ("P",)]
    got = compat_util.run(tmp_path, steps)
    d = oracle.decoder(2)
    d.init()
    import ctypes as C
    seen = set()
    for s, g in zip(steps, got):
        if s[0] == "D":
            r = oracle.lib.oc_decode(d.h, s[2], len(s[2]), d.buf.ctypes.data, s[1])
            assert g[0] == g[1] == r
        elif s[0] == "R":
            d.reset()
        elif s[0] == "F":
            assert g == (0, 0)
        elif s[0] == "P":
            v = C.c_int32(-777)
            r = oracle.lib.oc_decoder_ctl_pitch(d.h, C.byref(v))
            assert g[0] == r and g[2] == -5, (g, r)
            if r == 0:
                assert g[1] == v.value, (g, v.value)
                seen.add(v.value)
            else:
                assert r == -5 and g[1] == -777  # (the value is left alone)
    assert len(seen) > 2  # (the walk saw voiced frames: the comparison is not one of zeros)
