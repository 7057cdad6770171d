"""Empty packets in reference mode (data == NULL / len == 0): the reference conceals nothing, but its branch for them is live
(src/opus_decoder.cpp:290-308) -- passes of opus_decode_frame(st, NULL, 0), a frame of NO bytes in the decoder's last mode,
960 samples each, until frame_size is filled or a pass fails.

The known answers below are derived BY HAND from the reference's lines, not read off the oracle:
  * opus_decode (:351): frame_size <= 0 -> OPUS_BAD_ARG (-1);
  * opus_decode_native (:290): an empty packet with frame_size % (48000 / 400 = 120) != 0 -> -1;
  * opus_decode_frame (:154-278) with len 0: mode SILK-only (1000) -> silk_Decode runs off a coder that reads zeros
    (ec_dec_init with storage 0), the CELT branch is `else` (:259-268: zeros), return audiosize = 960 (:277);
    mode hybrid (1001) -> SILK runs, then celt_decode_with_ec: s_ec.storage (0) <= 1 -> ERR_OPUS_CELT_BAD_ARG = -18
    (src/celt.cpp:2225), st->prev_mode is set all the same (:276) and -18 comes back (:277), which :300 passes up;
    mode CELT-only (1002) -> -18 the same way, no SILK;
    mode 0 (no packet since opus_decoder_init :82 or OPUS_RESET_STATE :382, both clear st->mode) -> `s_mode != MODE_CELT_ONLY`
    (:175) is TRUE, so SILK runs (at 16 kHz, the `else` of :187), then `s_mode != MODE_SILK_ONLY` (:249) is true too: CELT, -18;
  * the loop (:296-305) adds 960 per pass until pcm_count >= frame_size: 1920 -> 1920, 2880 -> 2880.
And one derived property that needs no second decoder's word: ec_dec_init(buf, 0) never reads buf, so opus_decode_frame(NULL, 0)
IS opus_decode_frame(data, 0) -- an empty packet equals a TOC-only packet (one byte, no payload) with the last packet's TOC."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from test_gpu_pipeline import compare, desc_flags, make_walk, run_queued

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODY = bytes((37 * i + 11) & 255 for i in range(60))
SILK, HYB, CELT = bytes([0x0C]) + BODY[:40], bytes([0x7C]) + BODY, bytes([0xFC]) + BODY

# (packets decoded first, frame_size of the empty packet's call, the hand-derived return value)
EMPTY_KAT = [
    ([], 960, -18), ([], 1920, -18),            # fresh decoder: mode 0 -> SILK, then CELT's refusal
    ([SILK], 960, 960), ([SILK], 1920, 1920), ([SILK], 2880, 2880), ([SILK, SILK, SILK], 960, 960),
    ([SILK], 961, -1), ([SILK], 100, -1), ([SILK], 1081, -1), ([SILK], 0, -1), ([SILK], -960, -1),
    ([HYB], 960, -18), ([HYB], 1920, -18), ([HYB], 7, -1),
    ([CELT], 960, -18), ([CELT], 2880, -18), ([CELT], 50, -1),
    ([CELT, SILK], 960, 960), ([SILK, CELT], 960, -18), ([SILK, HYB], 960, -18), ([HYB, SILK], 1920, 1920),
    ([bytes([0x08]) + BODY[:30]], 960, 960),    # SILK-only mono (TOC 0x08)
    ([bytes([0x4C]) + BODY[:50]], 1920, 1920),  # SILK-only WB stereo
]


def _oc_decode(oracle, d, pkt, frame_size):
    r = oracle.lib.oc_decode(d.h, pkt if pkt is not None else None, len(pkt) if pkt is not None else 0, d.buf.ctypes.data, frame_size)
    return d.buf, r


@pytest.mark.parametrize("channels", [2, 1])
def test_oracle_empty_packet_known_answers(oracle, channels):
    for before, fs, want in EMPTY_KAT:
        d = oracle.decoder(channels)
        d.init()
        for p in before:
            assert d.decode(p)[1] == 960
        assert _oc_decode(oracle, d, b"", fs)[1] == want, (len(before), fs, want)
        d.init()
        for p in before:
            d.decode(p)
        assert _oc_decode(oracle, d, None, fs)[1] == want  # data == NULL
    # OPUS_RESET_STATE clears the mode (:382-390): SILK-only, reset, empty -> the mode-0 case
    d = oracle.decoder(channels)
    d.init()
    d.decode(SILK)
    d.reset()
    assert _oc_decode(oracle, d, b"", 960)[1] == -18


def test_oracle_empty_packet_is_a_toc_only_packet(oracle):
    """the derived property of the header: per pass, an empty packet == a packet of the last TOC and no payload -- PCM, return
    value and everything decoded afterwards"""
    rng = np.random.default_rng(77)
    for trial in range(60):
        channels = int(rng.integers(1, 3))
        a, b = oracle.decoder(channels), oracle.decoder(channels)
        a.init(), b.init()
        toc = None
        for f in range(10):
            if toc is not None and rng.random() < 0.35:
                passes = int(rng.integers(1, 4))
                pa, ra = _oc_decode(oracle, a, b"", 960 * passes)
                pa = pa.copy()
                total = 0
                for k in range(passes):
                    pb, rb = b.decode_cap(bytes([toc]), 1)
                    if rb < 0:
                        assert ra == rb
                        break
                    q3 = channels == 2 and not toc & 4 and not toc & 0x80 and (toc & 0x60) != 0x60  # only 960 entries defined
                    n = 960 if q3 else 960 * channels
                    assert np.array_equal(pa[total:total + 960].reshape(-1)[:n], pb[:960].reshape(-1)[:n]), (trial, f, k)
                    total += 960
                else:
                    assert ra == total
            else:
                cfg = int(rng.choice([1, 5, 9, 13, 15, 19, 31]))
                toc = cfg << 3 | (4 if rng.random() < 0.7 else 0)
                pkt = bytes([toc]) + rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8).tobytes()
                (pa, ra), (pb, rb) = a.decode_cap(pkt, 1), b.decode_cap(pkt, 1)
                q3 = channels == 2 and not toc & 4 and not toc & 0x80 and (toc & 0x60) != 0x60
                n = 960 if q3 else 960 * channels
                assert ra == rb and (ra < 0 or np.array_equal(pa[:960].reshape(-1)[:n], pb[:960].reshape(-1)[:n])), (trial, f)


def _mode_bw(toc):
    if toc & 0x80:
        bw = 1102 + ((toc >> 5) & 3)
        return 1002, (1101 if bw == 1102 else bw)
    if (toc & 0x60) == 0x60:
        return 1001, (1105 if toc & 0x10 else 1104)
    return 1000, 1101 + ((toc >> 5) & 3)


@pytest.mark.parametrize("entry", ["split", "single", "shadowed"])
def test_emulated_kernels_empty_packets(oracle, entry):
    """The kernel source in host emulation, driven frame by frame the way the host drives the device for an empty packet: a frame
    of len 0 with the last accepted packet's mode / bandwidth / channels, or mode 0 (descriptor bit 11) before the first packet --
    the split path, the single-kernel path and the pipelined SILK / hybrid path with its entropy-side copy of the past."""
    emul = os.path.join(ROOT, "tests", "emul", "libog_emul.so")
    subprocess.check_call(["make", "-C", os.path.dirname(emul), "-s"])
    emu = C.CDLL(emul)
    emu.emu_state_size.restype = C.c_int
    emu.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
    sig = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    emu.emu_decode_frame.argtypes = emu.emu_decode_frame_single.argtypes = sig
    emu.emu_decode_frame_shadowed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint] + sig[1:]
    emu.emu_shadow_vs_state.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng({"split": 5, "single": 6, "shadowed": 7}[entry])
    st = C.create_string_buffer(emu.emu_state_size())
    out = np.zeros((960, 2), dtype=np.int16)
    empties = fresh = 0
    for stream in range(80):
        channels = int(rng.integers(1, 3))
        d = oracle.decoder(channels)
        d.init()
        emu.emu_stream_init(st, channels)
        shadow, epoch = C.create_string_buffer(128), 1
        last = None  # (mode, bw, ch) of the last accepted packet

        def run(body, m, bw, ch):
            nonlocal epoch
            out[:] = 0
            if entry == "single":
                return emu.emu_decode_frame_single(st, body, len(body), m, bw, ch, out.ctypes.data)
            if entry == "shadowed" and m != 1002:
                r = emu.emu_decode_frame_shadowed(st, shadow, epoch, body, len(body), m, bw, ch, out.ctypes.data)
                if r > 0:
                    assert emu.emu_shadow_vs_state(st, shadow) == 0
                return r
            epoch += 1
            return emu.emu_decode_frame(st, body, len(body), m, bw, ch, out.ctypes.data)

        for f in range(12):
            if rng.random() < (0.5 if f == 0 else 0.3):
                passes = int(rng.integers(1, 3))
                ref, r = _oc_decode(oracle, d, b"", 960 * passes)
                ref = ref.copy()
                m, bw, ch = last if last else (0, 1104, channels)
                empties += 1
                fresh += last is None
                got = 0
                for k in range(passes):
                    r2 = run(b"", m, bw, ch)
                    if r2 < 0:
                        got = r2
                        break
                    ncmp = 960 * ch if (m == 1000 and ch < channels) else 960 * channels
                    assert np.array_equal(out.reshape(-1)[:ncmp], ref[got:got + 960].reshape(-1)[:ncmp]), (stream, f, k)
                    got += r2
                assert got == r, (stream, f, m, got, r)
                if entry == "shadowed" and last is None:  # the mode-0 frame leaves prev_mode 0 in the copy as well
                    assert emu.emu_shadow_vs_state(st, shadow) in (0,), (stream, f)
            else:
                cfg = int(rng.choice([1, 5, 9, 13, 15, 19, 23, 31]))
                stereo = rng.random() < 0.7
                toc = cfg << 3 | (4 if stereo else 0)
                body = rng.integers(0, 256, int(rng.choice([0, 1, 2, 30, 80, 200])), dtype=np.uint8).tobytes()
                ref, r = d.decode(bytes([toc]) + body)
                m, bw = _mode_bw(toc)
                ch = 2 if stereo else 1
                last = (m, bw, ch)  # accepted by opus_decode_native whatever its frame returns (:327-331)
                r2 = run(body, m, bw, ch)
                assert r == r2, (stream, f, hex(toc), r, r2)
                if r > 0:
                    ncmp = 960 * ch if (m == 1000 and ch < channels) else 960 * channels
                    assert np.array_equal(out.reshape(-1)[:ncmp], ref[:960].reshape(-1)[:ncmp]), (stream, f, hex(toc))
    assert empties > 150 and fresh > 20


def test_empty_packet_descriptors(pkg):
    """opusgpu_empty_packet_to_frames (host code): one descriptor per pass, the last packet's flags or the mode-0 flags"""
    fl = pkg.packet_to_frames(SILK)[0][2]
    assert pkg.empty_packet_to_frames(fl, 2, 960) == [(0, 0, fl)]
    assert pkg.empty_packet_to_frames(fl, 2, 1920) == [(0, 0, fl)] * 2
    assert pkg.empty_packet_to_frames(fl, 2, 1080) == [(0, 0, fl)] * 2  # whole passes, as the reference makes them
    none2, none1 = 1 | 3 << 2 | 32 | 1 << 11, 1 | 3 << 2 | 1 << 11      # hybrid, the decoder's channels, OPUSGPU_DESC_NO_MODE
    assert pkg.empty_packet_to_frames(-1, 2, 960) == [(0, 0, none2)] and pkg.empty_packet_to_frames(-1, 1, 960) == [(0, 0, none1)]
    for bad in (0, -960, 961, 100, 960 * 49):
        assert pkg.empty_packet_to_frames(fl, 2, bad) == -1
    assert pkg.packet_to_frames(b"") == -4  # opus_packet_parse_impl's own answer to len == 0 (src/opus_decoder.cpp:567)


# ---- on the GPU --------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("channels", [2, 1])
def test_gpu_empty_packet_known_answers(pkg, gpu_ctx, channels):
    """the hand-derived table through opusgpu_decode_packets (frame_size = frame_capacity x 960), one stream per row; NULL and
    zero-length packets alike"""
    rows = [(b, fs, w) for b, fs, w in EMPTY_KAT if fs > 0 and fs % 960 == 0]
    for as_null in (False, True):
        gpu_ctx.streams_alloc(len(rows), channels)
        for k in range(max(len(b) for b, _, _ in rows)):
            ids = [i for i, (b, _, _) in enumerate(rows) if len(b) > k]
            _, res = gpu_ctx.decode_packets(np.array(ids), [rows[i][0][k] for i in ids])
            assert (res == 960).all()
        for cap in (1, 2, 3):
            ids = [i for i, (_, fs, _) in enumerate(rows) if fs == 960 * cap]
            _, res = gpu_ctx.decode_packets(np.array(ids), [None if as_null else b""] * len(ids), frame_capacity=cap)
            assert list(res) == [rows[i][2] for i in ids], (cap, list(res))
    # after OPUS_RESET_STATE the stream is back in mode 0
    gpu_ctx.streams_alloc(1, channels)
    gpu_ctx.decode_packets([0], [SILK])
    gpu_ctx.streams_reset(0, 1, full=False)
    assert gpu_ctx.decode_packets([0], [b""])[1][0] == -18


@pytest.mark.gpu
@pytest.mark.parametrize("channels,cap", [(2, 1), (2, 3), (1, 2)])
def test_gpu_empty_packets_in_random_walks(pkg, oracle, gpu_ctx, channels, cap):
    """opusgpu_decode_packets: every mode and bandwidth with switches, mono and stereo packets, payloads of 0 - 1275 bytes, a
    fifth of the packets empty (also before a stream's first packet) -- every return value and every sample against the oracle,
    which runs the reference's loop (oracle/oc_packet.c oc_decode)"""
    rng = np.random.default_rng(900 + 10 * channels + cap)
    n, frames = 1500, 10
    arena, offs, plen, lens, toc = make_walk(rng, n, frames, channels)
    plen = np.where(rng.random((frames, n)) < 0.2, 0, plen)
    ref, rets = oracle.batch_decode_var(channels, arena, offs, plen.astype(np.int32), cap_frames=cap)
    gpu_ctx.streams_alloc(n, channels)
    ids = np.arange(n, dtype=np.int32)
    seen = {"pcm": 0, "refused": 0}
    for f in range(frames):
        pcm, res = gpu_ctx.decode_packets_arena(ids, arena, offs[f], plen[f], frame_capacity=cap)
        assert np.array_equal(res, rets[:, f]), (f, np.nonzero(res != rets[:, f])[0][:8])
        for i in np.nonzero(res > 0)[0]:
            r = int(res[i])
            if plen[f, i] == 0:
                seen["pcm"] += 1
            # (Q3: a mono SILK-only packet in a stereo decoder defines 960 of a frame's 1920 entries: compare what is defined)
            last = toc[:f + 1, i][plen[:f + 1, i] > 0]
            t = int(last[-1]) if len(last) else 0
            q3 = channels == 2 and not t & 4 and not t & 0x80 and (t & 0x60) != 0x60
            a, b = pcm[i, :r].reshape(r // 960, -1), ref[i, f, :r].reshape(r // 960, -1)
            w = 960 if q3 else 960 * channels
            assert np.array_equal(a[:, :w], b[:, :w]), (f, i, r)
        seen["refused"] += int(((res == -18) & (plen[f] == 0)).sum())
    assert seen["pcm"] > 300 and seen["refused"] > 300


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [2, 1])
def test_gpu_empty_packets_on_the_device_path(pkg, oracle, gpu_ctx, channels):
    """opusgpu_decode_step_device with the descriptors opusgpu_empty_packet_to_frames makes: len 0 and the flags of the stream's
    last accepted packet, or the mode-0 flags (OPUSGPU_DESC_NO_MODE) before its first -- in order, pipelined step by step, and
    as one window; all against the oracle"""
    rng = np.random.default_rng(950 + channels)
    n, frames = 2048, 12
    arena, offs, plen, lens, toc = make_walk(rng, n, frames, channels)
    empty = rng.random((frames, n)) < 0.2
    empty[0] |= rng.random(n) < 0.3
    plen = np.where(empty, 0, plen)
    ref, rets = oracle.batch_decode_var(channels, arena, offs, plen.astype(np.int32))
    flags, _ = desc_flags(toc)
    none = 1 | 3 << 2 | (32 if channels == 2 else 0) | 1 << 11
    assert pkg.empty_packet_to_frames(-1, channels, 960)[0][2] == none
    cur = np.full(n, none, dtype=np.int32)
    eff_toc = toc.copy()
    last_toc = np.full(n, 0x7C if channels == 2 else 0x78, dtype=np.uint8)  # (for compare()'s Q3 rule only)
    for f in range(frames):
        cur = np.where(empty[f], cur, flags[f])
        flags[f] = cur
        last_toc = np.where(empty[f], last_toc, toc[f])
        eff_toc[f] = last_toc
    lens = np.where(empty, 0, lens)
    for kw in (dict(pipeline=False), dict(pipeline=True), dict(pipeline=True, window=True)):
        pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, flags=flags, **kw)
        assert compare(pcm, res, ref, rets, eff_toc, channels) == 0, kw
    assert (rets[empty.T] == 960).sum() > 500 and (rets[empty.T] == -18).sum() > 500
