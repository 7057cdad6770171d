"""Every `return` of the reference's packet layer, as known answers DERIVED BY HAND from its source lines -- not read off the
oracle, which shares an author with the kernels: opus_packet_parse_impl (src/opus_decoder.cpp:559-680),
opus_decode_native (:280-348), opus_decode (:350), opus_multistream_decode_native (:826-913), and the two frame-level refusals
that pass up through them (celt_decode_with_ec, src/celt.cpp:2211-2225).  The same table is put to the oracle, to the host
framing of the C ABI (opusgpu_packet_to_frames), to opusgpu_decode_packets on the GPU and to include/opus_decoder.h on the GPU.

Notation: a packet is TOC | rest.  TOC 0xFC = CELT-only FB 20 ms stereo, 0x0C = SILK-only NB 20 ms stereo, 0x7C = hybrid FB 20 ms
stereo, 0x1C = SILK-only NB 60 ms stereo (2880 samples by the TOC), 0xE4 = CELT-only FB 2.5 ms stereo; TOC | 1, | 2, | 3 = frame
count codes 1 - 3.  The reference decodes EVERY frame as 960 samples whatever the TOC says (:161, Q6), so a packet that is
accepted returns 960 x its frame count -- unless a frame fails, whose code then comes back (:333-337)."""
import numpy as np
import pytest

P = lambda n, k=7: bytes((k * i + 3) & 255 for i in range(n))  # n payload bytes

# (name, packet, frame_size, return value, frame count by opus_packet_parse_impl or its error, the line that decides)
KAT = [
    # ---- opus_packet_parse_impl
    ("code 0", b"\xfc" + P(50), 960, 960, 1, ":582 count = 1; :665 last_size 50 <= 1275"),
    ("code 0, 1275 payload bytes", b"\xfc" + P(1275), 960, 960, 1, ":665 last_size == 1275 passes"),
    ("code 0, 1276 payload bytes", b"\xfc" + P(1276), 960, -4, -4, ":665 last_size > 1275"),
    ("code 1, odd payload", b"\xfd" + P(3), 1920, -4, -4, ":590 len & 1"),
    ("code 1", b"\xfd" + P(100), 1920, 1920, 2, ":591 two frames of 50"),
    ("code 1, SILK", b"\x0d" + P(60), 1920, 1920, 2, ":591"),
    ("code 2, no size byte", b"\xfe", 1920, -4, -4, "parse_size :524 len < 1 -> size -1; :603"),
    ("code 2, size beyond the packet", b"\xfe\x0a" + P(5), 1920, -4, -4, ":603 size[0] 10 > len 5"),
    ("code 2, first byte of a two-byte size only", b"\xfe\xfc", 1920, -4, -4, "parse_size :530 len < 2; :603"),
    ("code 2, two-byte size beyond the packet", b"\xfe\xfc\x01" + P(255), 1920, -4, -4, ":603 size 4 * 1 + 252 = 256 > 255"),
    ("code 2, SILK, second frame empty", b"\x0e\xfc\x01" + P(256), 1920, 1920, 2, ":605 last_size 0: SILK decodes an empty frame"),
    ("code 2, CELT, second frame empty", b"\xfe\xfc\x01" + P(256), 1920, -18, 2, "frame 2 has 0 bytes: celt.cpp:2225 via :335"),
    ("code 2", b"\xfe\x28" + P(40 + 70), 1920, 1920, 2, ":605 40 + 70"),
    ("code 3, no count byte", b"\xff", 960, -4, -4, ":609 len < 1"),
    ("code 3, zero frames", b"\xff\x00" + P(10), 960, -4, -4, ":613 count <= 0"),
    ("code 3, 7 x 20 ms", b"\xff\x07" + P(70), 5760, -4, -4, ":613 960 * 7 > 5760"),
    ("code 3, 6 x 20 ms", b"\xff\x06" + P(60), 5760, 5760, 6, ":613 960 * 6 == 5760 passes; :641 CBR 10 each"),
    ("code 3, 60 ms x 3", b"\x1f\x03" + P(30), 5760, -4, -4, ":613 2880 * 3 > 5760"),
    ("code 3, padding flag, no length byte", b"\xff\x41", 960, -4, -4, ":619 len <= 0"),
    ("code 3, padding longer than the packet", b"\xff\x41\x05" + P(3), 960, -4, -4, ":626 len < 0"),
    ("code 3, padding", b"\xff\x41\x02" + P(10) + b"\0\0", 960, 960, 1, ":623 pad 2; :641 10 / 1"),
    ("code 3, padding of 255 continues", b"\xff\x41\xff\x01" + P(10) + bytes(255), 960, 960, 1, ":622 254 + 1"),
    ("code 3 VBR, size beyond the packet", b"\xff\x82\x14" + P(10), 1920, -4, -4, ":635 size[0] 20 > len"),
    ("code 3 VBR", b"\xff\x82\x05" + P(15), 1920, 1920, 2, ":637 last_size 16 - (1 + 5) = 10"),
    ("code 3 CBR, not divisible", b"\xff\x02" + P(5), 1920, -4, -4, ":642 2 * 2 != 5"),
    ("code 3 CBR", b"\xff\x03" + P(3 * 45), 2880, 2880, 3, ":641"),
    # ---- opus_decode_native / opus_decode
    ("room for less than the TOC's duration", b"\xfc" + P(50), 959, -2, 1, ":323 1 * 960 > 959"),
    ("two frames, room for one", b"\xfd" + P(80), 960, -2, 2, ":323 2 * 960 > 960"),
    ("60 ms TOC, room for 20 ms", b"\x1c" + P(40), 960, -2, 1, ":323 the check uses the TOC's 2880 ..."),
    ("60 ms TOC, room for 60 ms", b"\x1c" + P(40), 2880, 960, 1, "... :161 the decode is 960 all the same (Q6)"),
    ("frame_size 0", b"\xfc" + P(50), 0, -1, 1, ":351"),
    ("frame_size negative", b"\xfc" + P(50), -960, -1, 1, ":351"),
    # ---- the frame-level refusals that pass up (:333-337)
    ("CELT-only, TOC only", b"\xfc", 960, -18, 1, "celt.cpp:2225 storage 0 <= 1"),
    ("CELT-only, one payload byte", b"\xfc\x55", 960, -18, 1, "celt.cpp:2225 storage 1 <= 1"),
    ("CELT-only, two payload bytes", b"\xfc\x55\xaa", 960, 960, 1, "storage 2 passes"),
    ("hybrid, TOC only", b"\x7c", 960, -18, 1, "celt.cpp:2225 after the SILK half"),
    ("SILK-only, TOC only", b"\x0c", 960, 960, 1, ":259 the CELT branch is not taken"),
    ("code 1, CELT, two frames of one byte", b"\xfd\x01\x02", 1920, -18, 2, "frame 1: celt.cpp:2225"),
]


def test_packet_framing_known_answers(pkg):
    """opusgpu_packet_to_frames = opus_packet_parse_impl + the TOC helpers (host code)"""
    for name, pkt, fs, ret, count, why in KAT:
        got = pkg.packet_to_frames(pkt)
        assert (got if isinstance(got, int) else len(got)) == count, (name, why)
    assert pkg.packet_to_frames(b"") == -4  # :572 len == 0


def test_oracle_return_code_known_answers(oracle):
    for name, pkt, fs, ret, count, why in KAT:
        for channels in (2, 1):
            d = oracle.decoder(channels)
            d.init()
            r = oracle.lib.oc_decode(d.h, pkt, len(pkt), d.buf.ctypes.data, fs)
            assert r == ret, (name, channels, r, ret, why)
    d = oracle.decoder(2)
    assert oracle.lib.oc_decode(d.h, b"\xfc" + P(10), -1, d.buf.ctypes.data, 960) == -1  # :309 len < 0


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [2, 1])
def test_gpu_return_code_known_answers(pkg, gpu_ctx, channels):
    """opusgpu_decode_packets, one fresh stream per row (frame_size = frame_capacity x 960: the rows that can be said so)"""
    rows = [r for r in KAT if r[2] > 0 and r[2] % 960 == 0]
    for cap in sorted({r[2] // 960 for r in rows}):
        part = [r for r in rows if r[2] == 960 * cap]
        gpu_ctx.streams_alloc(len(part), channels)
        _, res = gpu_ctx.decode_packets(np.arange(len(part)), [r[1] for r in part], frame_capacity=cap)
        for r, got in zip(part, res):
            assert got == r[3], (r[0], int(got), r[3], r[5])
    # a negative length is refused per packet (:309)
    gpu_ctx.streams_alloc(1, channels)
    import ctypes as C
    buf = C.create_string_buffer(b"\xfc" + P(10))
    ptrs = np.array([C.addressof(buf)], dtype=np.uint64)
    pcm, res = np.zeros((1, 960, channels), dtype=np.int16), np.zeros(1, dtype=np.int32)
    gpu_ctx.decode_packets_raw(np.zeros(1, dtype=np.int32), ptrs, np.array([-1], dtype=np.int32), pcm, res)
    assert res[0] == -1


@pytest.mark.gpu
def test_opus_decoder_h_return_code_known_answers(tmp_path, oracle):
    """the same table through opus_decode AND opus_multistream_decode (include/opus_decoder.h over the GPU), a reset between rows;
    opus_multistream_decode_native's own returns on top: frame_size <= 0 -> -1 (:836), len < 0 -> -1 (:848), frame_size capped at
    5760 (:841), the validation's -4 / -2 (:854-860) equal to opus_decode_native's"""
    import compat_util
    steps, want = [], {}
    for name, pkt, fs, ret, count, why in KAT:
        steps += [("R",), ("D", fs, pkt)]
        want[len(steps) - 1] = (name, ret, why)
    steps += [("R",), ("D", 20000, b"\xff\x06" + P(60)), ("R",), ("D", 960, b"\xfc" + P(20)), ("N", 960), ("N", 0)]
    got = compat_util.run(tmp_path, steps)
    for k, (name, ret, why) in want.items():
        ra, rb, pcm = got[k]
        assert ra == rb == ret, (name, ra, rb, ret, why)
    assert got[-5][0] == got[-5][1] == 5760   # frame_size 20000: six frames fit whatever the cap
    assert got[-2] == (-1, -1) and got[-1] == (-1, -1)
    # and the PCM of the rows that decode, against the oracle driven the same way (OPUS_RESET_STATE is the reference's PARTIAL
    # reset, Q5: what it keeps of the CELT state carries from row to row)
    d = oracle.decoder(2)
    d.init()
    for k, s in enumerate(steps):
        if s[0] == "R":
            d.reset()
        elif s[0] == "D":
            fs, pkt = s[1], s[2]
            r = oracle.lib.oc_decode(d.h, pkt, len(pkt), d.buf.ctypes.data, min(fs, 5760))
            assert got[k][0] == r, (k, got[k][0], r)
            toc = pkt[0]
            if r > 0 and not (not toc & 0x80 and (toc & 0x60) != 0x60 and not toc & 4):  # (Q3: mono SILK-only in a stereo decoder)
                n = min(r, fs)
                assert np.array_equal(got[k][2], d.buf[:n]), (k, want.get(k))


@pytest.mark.gpu
def test_opus_decoder_h_lower_level_entry_points(tmp_path):
    """The entry points the reference's header declares below opus_decode / opus_multistream_decode (src/opus_decoder.h:182,
    :202-207): opus_packet_parse_impl's count, payload offset and packet offset -- known answers from :559-680 --,
    opus_decode_native with self_delimited = 1 (OPUS_UNIMPLEMENTED here: one elementary stream) and with frame_size 0 (:323: a packet
    that parses fails the size check), opus_multistream_decode_native with a copy function of the caller's."""
    import compat_util
    rows = [
        # packet, count, payload offset, packet offset (:676-677: pad + what the frames take = the whole packet), ret at frame_size 0
        (b"\xfc" + P(50), 1, 1, 51, -2),
        (b"\xfd" + P(100), 2, 1, 101, -2),
        (b"\xfe\x28" + P(40 + 70), 2, 2, 112, -2),
        (b"\xff\x41\x02" + P(10) + b"\0\0", 1, 3, 15, -2),
        (b"\xff\x82\x05" + P(15), 2, 3, 18, -2),
        (b"\xfe\x0a" + P(5), -4, -1, -1, -4),
    ]
    steps = []
    for pkt, *_ in rows:
        steps += [("R",), ("D", 5760, pkt), ("X",)]
    got = compat_util.run(tmp_path, steps)
    for k, (pkt, count, poff, koff, r0) in enumerate(rows):
        x = got[3 * k + 2]
        assert x[0] == count, (k, x)
        if count > 0:
            assert x[1] == poff and x[2] == koff, (k, x)
        assert x[3] == -5 and x[4] == r0, (k, x)
        assert x[5] == (960 * count if count > 0 else count), (k, x)  # (and the program compared the PCM: -9998 / -9999 on a difference)


@pytest.mark.gpu
def test_opus_decoder_h_argument_checks(tmp_path):
    """opus_multistream_decoder_create / _init (src/opus_decoder.cpp:757-821), validate_layout (:688-697), the get_size functions
    (:66-71, :740-750) and opus_decoder_init (:82-88): known answers from those lines -- and the two cases this library answers
    differently on purpose (output rates other than 48 kHz, more than one elementary stream: OPUS_UNIMPLEMENTED)."""
    import compat_util
    (v,) = compat_util.run(tmp_path, [("V",)])
    assert v[:6] == (-1, -1, -1, -1, -1, -1), v    # :805-810 five times, then :774 (validate_layout)
    assert v[6:8] == (0, 0), v                      # a muted channel (255) and a mono stream are layouts
    assert v[8] == -1, v                            # :86: 44,100 Hz is not a rate
    assert v[9:11] == (-5, -5), v                   # (this library's two: INTEGRATION.md)
    assert v[11:15] == (0, 0, 0, 1), v              # :744
    assert v[15:18] == (0, 0, 1), v                 # :70
    assert v[18] == -1, v                           # :87: channels
