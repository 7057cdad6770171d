"""GPU parity for the SILK and hybrid paths and for mode / bandwidth / channel switching (C ABI vs oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_run(pkg, ctx, toc, pay, channels=2):
    """Decode pay[frames, streams, L] through the device-resident path; returns int16 [streams, frames, 960, ch]."""
    frames, n, L = pay.shape
    ctx.streams_alloc(n, channels)
    d_desc, d_arena = ctx.dev_alloc(16 * n), ctx.dev_alloc(n * (L + 1) + 16)
    d_pcm, d_res = ctx.dev_alloc(n * 960 * channels * 2), ctx.dev_alloc(4 * n)
    out = np.zeros((n, frames, 960, channels), dtype=np.int16)
    tmp = np.zeros((n, 960, channels), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    for f in range(frames):
        arena, descs = pkg.build_step(toc, pay[f])
        ctx.h2d(d_arena, arena)
        ctx.h2d(d_desc, descs)
        ctx.decode_step_device(n, d_desc, d_arena, d_pcm, d_res)
        ctx.synchronize()
        ctx.d2h(tmp, d_pcm)
        ctx.d2h(res, d_res)
        assert (res == 960).all()
        out[:, f] = tmp
    for p in (d_desc, d_arena, d_pcm, d_res):
        ctx.dev_free(p)
    return out


def test_silk_nb_stereo_every_sample(pkg, oracle, gpu_ctx):
    """BASELINE config 3 in miniature: SILK-only NB stereo, every PCM sample compared."""
    pay = pkg.lcg_payloads(1024, 8, 40)
    ref, ok = oracle.batch_decode(2, pkg.TOC_SILK_NB_STEREO, pay)
    assert ok == 1024 * 8
    got = _device_run(pkg, gpu_ctx, pkg.TOC_SILK_NB_STEREO, pay)
    assert np.array_equal(got, ref)


def test_hybrid_fb_stereo(pkg, oracle, gpu_ctx):
    pay = pkg.lcg_payloads(768, 6, 120)
    ref, ok = oracle.batch_decode(2, pkg.TOC_HYBRID_FB_STEREO, pay)
    assert ok == 768 * 6
    got = _device_run(pkg, gpu_ctx, pkg.TOC_HYBRID_FB_STEREO, pay)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("toc,L", [(0x2C, 60), (0x4C, 80), (0x48, 70), (0x6C, 100), (0x08, 30)])
def test_other_silk_and_hybrid_configs(pkg, oracle, gpu_ctx, toc, L):
    """SILK MB / WB (stereo, mono), hybrid SWB, SILK NB mono."""
    ch = 2 if toc & 4 else 1
    pay = pkg.lcg_payloads(128, 5, L, seed_base=0x1234 + toc)
    ref, ok = oracle.batch_decode(ch, toc, pay)
    assert ok == 128 * 5
    got = _device_run(pkg, gpu_ctx, toc, pay, channels=ch)
    assert np.array_equal(got, ref)


def test_mode_bandwidth_and_channel_switching(pkg, oracle, gpu_ctx):
    """Random per-frame mode switches (incl. the reference's hybrid->SILK quirk Q4) through the host-buffer path."""
    rng = np.random.default_rng(11)
    cfgs = {0: [1, 5, 9], 1: [13, 15], 2: [19, 23, 27, 31]}
    for channels in (1, 2):
        n, frames = 160, 12
        pk = []
        for s in range(n):
            row, mode = [], int(rng.integers(3))
            for f in range(frames):
                if rng.random() < 0.3:
                    mode = int(rng.integers(3))
                stereo = (channels == 2) if rng.random() < 0.9 else bool(rng.integers(2))
                toc = (int(rng.choice(cfgs[mode])) << 3) | (4 if stereo else 0)
                L = int(rng.choice([40, 120, 160, 7, 300]))
                row.append(bytes([toc]) + rng.integers(0, 256, L, dtype=np.uint8).tobytes())
            pk.append(row)
        ref, rets = oracle.decode_streams(channels, pk)
        ctx = gpu_ctx
        ctx.streams_alloc(n, channels)
        for f in range(frames):
            pkts = [pk[s][f] for s in range(n)]
            pcm, res = ctx.decode_packets(np.arange(n), pkts)
            assert (res == rets[:, f]).all()
            for s in range(n):
                if res[s] <= 0:
                    continue
                toc = pkts[s][0]
                pch = 2 if toc & 4 else 1
                silk_only = not (toc & 0x80) and (toc & 0x60) != 0x60
                # a mono SILK-only packet in a stereo decoder: the reference writes 960 entries only (Q3)
                ncmp = 960 * pch if (silk_only and pch < channels) else 960 * channels
                assert np.array_equal(pcm[s].reshape(-1)[:ncmp], ref[s, f].reshape(-1)[:ncmp]), (s, f, hex(toc))


def test_reset_state_semantics(pkg, oracle, gpu_ctx):
    """OPUS_RESET_STATE keeps CELT history (Q5): same packets after reset differ from a fresh stream, and match the oracle."""
    n, L = 32, 160
    pay = pkg.lcg_payloads(n, 6, L, seed_base=77)
    ctx = gpu_ctx
    ctx.streams_alloc(n, 2)
    decs = [oracle.decoder(2) for _ in range(n)]
    for f in range(6):
        if f == 3:
            ctx.streams_reset(0, n, full=False)
            for d in decs:
                d.reset()
        pkts = [bytes([pkg.TOC_CELT_FB_STEREO]) + pay[f, s].tobytes() for s in range(n)]
        pcm, res = ctx.decode_packets(np.arange(n), pkts)
        assert (res == 960).all()
        for s in range(n):
            out, r = decs[s].decode(pkts[s])
            assert r == 960 and np.array_equal(pcm[s], out[:960])


def test_multiframe_packets_and_errors(pkg, oracle, gpu_ctx):
    """Code 1/2/3 packets are decoded frame after frame; malformed packets return the reference's error codes."""
    rng = np.random.default_rng(5)
    ctx = gpu_ctx
    n = 24
    ctx.streams_alloc(n, 2)
    decs = [oracle.decoder(2) for _ in range(n)]
    for d in decs:
        d.init()
    pkts = []
    for s in range(n):
        body = rng.integers(0, 256, 2 * 60, dtype=np.uint8).tobytes()
        kind = s % 4
        if kind == 0:
            pkts.append(bytes([0xFC | 1]) + body)                       # code 1: two CBR frames
        elif kind == 1:
            pkts.append(bytes([0xFC | 2, 50]) + body[:110])             # code 2: two VBR frames (50 + 60)
        elif kind == 2:
            pkts.append(bytes([0xFC | 3, 0x02]) + body)                 # code 3, 2 CBR frames
        else:
            pkts.append(bytes([0xFC | 1]) + body[:119])                 # odd payload for code 1 -> invalid
    pcm, res = ctx.decode_packets(np.arange(n), pkts, frame_capacity=2)
    for s in range(n):
        out, r = decs[s].decode(pkts[s])
        assert res[s] == r, (s, res[s], r)
        if r > 0:
            assert np.array_equal(pcm[s, :r], out[:r])


def test_tiny_frames(pkg, oracle, gpu_ctx):
    """Frames of 0 .. 3 payload bytes (the reference has no special case for them: the range decoder simply runs out of
    bytes), in every mode, stereo and mono packets, interleaved with ordinary frames so that state carries across."""
    rng = np.random.default_rng(11)
    for channels in (2, 1):
        tocs = [0xFC, 0x0C, 0x7C, 0x4C, 0x2C] if channels == 2 else [0xF8, 0x08, 0x78, 0xFC, 0x0C]
        n, frames = 120, 6
        pk = []
        for s in range(n):
            toc = tocs[s % len(tocs)]  # mode fixed per stream (no Q4 transitions here; they have their own test)
            row = []
            for f in range(frames):
                L = int(rng.choice([0, 1, 2, 3, 0, 1, 50]))
                kind = rng.integers(4)
                body = bytes(L) if kind == 0 else (b"\xff" * L if kind == 1 else rng.integers(0, 256, L, dtype=np.uint8).tobytes())
                row.append(bytes([toc]) + body)
            pk.append(row)
        ref, rets = oracle.decode_streams(channels, pk)
        gpu_ctx.streams_alloc(n, channels)
        for f in range(frames):
            pcm, res = gpu_ctx.decode_packets(np.arange(n), [pk[s][f] for s in range(n)], frame_capacity=1)
            assert (res == rets[:, f]).all(), (channels, f, res[res != rets[:, f]][:5], rets[:, f][res != rets[:, f]][:5])
            for s in range(n):
                if res[s] <= 0:
                    continue
                toc = pk[s][f][0]
                stereo_pkt = bool(toc & 4)
                silk_only = not (toc & 0x80) and (toc & 0x60) != 0x60
                # Q3: a mono SILK-only packet in a stereo decoder defines only the first 960 interleaved entries
                ncmp = 960 if (silk_only and not stereo_pkt and channels == 2) else 960 * channels
                assert (pcm[s].reshape(-1)[:ncmp] == ref[s, f].reshape(-1)[:ncmp]).all(), (channels, s, f, hex(toc), len(pk[s][f]))


def test_tiny_celt_and_hybrid_frames_return_the_reference_code(pkg, gpu_ctx):
    """Hand-derived known answers (tests/test_oracle_kat.py::TINY_KAT, reference src/celt.cpp:2225 / src/opus_decoder.h:55):
    CELT-only and hybrid frames of 0 / 1 bytes come back as -18 (ERR_OPUS_CELT_BAD_ARG), SILK-only ones decode; both through the
    split kernels (steps of many frames) and the single-frame call, in order and pipelined."""
    from test_oracle_kat import TINY_KAT
    for channels in (2, 1):
        n = len(TINY_KAT)
        gpu_ctx.streams_alloc(n, channels)
        _, res = gpu_ctx.decode_packets(np.arange(n), [p for p, _ in TINY_KAT], frame_capacity=1)
        assert res.tolist() == [w for _, w in TINY_KAT], (channels, res.tolist())
        _, res = gpu_ctx.decode_packets(np.arange(n), [bytes([p[0]]) + bytes(range(40)) for p, _ in TINY_KAT], frame_capacity=1)
        assert (res == 960).all()
        for s, (p, w) in enumerate(TINY_KAT):  # one packet per call
            _, res = gpu_ctx.decode_packets([s], [p], frame_capacity=1)
            assert res[0] == w, (channels, p.hex(), res[0])


def test_more_short_frames_than_room_is_refused(pkg, gpu_ctx):
    """include/opusgpu.h, opusgpu_decode_packets: four 2.5 ms CELT frames are 480 samples by the TOC, which passes the
    reference's size check against one 960-sample frame of room -- and the reference then writes 4 x 960 samples (Q6).  The
    library returns OPUSGPU_BUFFER_TOO_SMALL and writes nothing; the stream's state is untouched (the next frame decodes as
    if the packet had not been there)."""
    rng = np.random.default_rng(21)
    n = 8
    gpu_ctx.streams_alloc(n, 2)
    first = [bytes([0xFC]) + rng.integers(0, 256, 100, dtype=np.uint8).tobytes() for _ in range(n)]
    follow = [bytes([0xFC]) + rng.integers(0, 256, 100, dtype=np.uint8).tobytes() for _ in range(n)]
    cfg_celt_fb_2_5ms = 28
    short4 = bytes([cfg_celt_fb_2_5ms << 3 | 4 | 3, 4]) + rng.integers(0, 256, 4 * 20, dtype=np.uint8).tobytes()  # code 3, CBR, 4 frames
    assert len(pkg.packet_to_frames(short4)) == 4
    pcm0, res0 = gpu_ctx.decode_packets(np.arange(n), first)
    assert (res0 == 960).all()
    pcm1, res1 = gpu_ctx.decode_packets(np.arange(n), [short4] * n, frame_capacity=1)
    assert (res1 == -2).all() and not pcm1.any()
    pcm2, res2 = gpu_ctx.decode_packets(np.arange(n), follow)
    # the same two good packets on fresh streams, without the refused one in between
    gpu_ctx.streams_alloc(n, 2)
    gpu_ctx.decode_packets(np.arange(n), first)
    pcm3, res3 = gpu_ctx.decode_packets(np.arange(n), follow)
    assert (res2 == 960).all() and (res3 == 960).all() and (pcm2 == pcm3).all()


def test_packets_in_one_array_decode_like_a_list_of_packets(pkg, oracle, gpu_ctx):
    """Context.decode_packets_arena (pointer table made by numpy) against Context.decode_packets (one buffer per packet):
    ragged packets of all three modes, multi-frame packets, an empty and a malformed packet, a reused PCM array."""
    rng = np.random.default_rng(17)
    n, cap = 500, 3
    tocs = [pkg.TOC_SILK_NB_STEREO, pkg.TOC_HYBRID_FB_STEREO, pkg.TOC_CELT_FB_STEREO]
    packets = []
    for i in range(n):
        toc = tocs[i % 3]
        L = int(rng.integers(1, 300))
        body = rng.integers(0, 256, size=L, dtype=np.uint8).tobytes()
        kind = i % 11
        if kind == 0:
            packets.append(bytes([toc | 1]) + body + body)          # two equal-size frames
        elif kind == 1:
            packets.append(b"")                                      # no data: the reference's empty-packet branch (tests/test_empty_packets.py)
        elif kind == 2:
            packets.append(bytes([toc | 3, 0]))                      # code 3 with zero frames: invalid
        else:
            packets.append(bytes([toc]) + body)
    lens = np.array([len(p) for p in packets], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
    arena = np.frombuffer(b"".join(packets) + b"\0", dtype=np.uint8)
    ids = np.arange(n, dtype=np.int32)
    gpu_ctx.streams_alloc(n, 2)
    want = [gpu_ctx.decode_packets(ids, packets, frame_capacity=cap) for _ in range(2)]
    gpu_ctx.streams_alloc(n, 2)
    out = None
    for rnd in range(2):
        out, res = gpu_ctx.decode_packets_arena(ids, arena, offs, lens, frame_capacity=cap, pcm=out)
        assert np.array_equal(res, want[rnd][1])
        ok = res > 0
        assert ok.sum() > 300 and (res < 0).sum() >= 70 and (res == 1920).sum() >= 30
        for i in np.nonzero(ok)[0]:
            assert np.array_equal(out[i, :res[i]], want[rnd][0][i, :res[i]]), (rnd, i)
    with pytest.raises(ValueError):
        gpu_ctx.decode_packets_arena(ids, arena, offs + 10**9, lens)
    # ... and both against the oracle: the two entry points share decode_packets_impl, so a framing or mode-mask slip in it
    # would cancel out in the comparison above
    dec = [oracle.decoder(2) for _ in range(n)]
    for d in dec:
        d.init()
    for rnd in range(2):
        for i in range(n):
            o, r = dec[i].decode_cap(packets[i], cap)
            assert want[rnd][1][i] == r, (rnd, i, want[rnd][1][i], r)
            if r > 0:
                assert np.array_equal(want[rnd][0][i, :r], o[:r]), (rnd, i)


def test_steps_of_a_multi_frame_call_launch_the_kernels_of_their_own_modes(pkg, oracle, gpu_ctx):
    """opusgpu_decode_packets with room for several frames runs one step per frame index; a step's mode mask (which kernels
    are launched at all) has to come from that step's own table.  Packets whose frame counts differ by mode make the steps'
    mode sets differ from any prefix of the call's frames in packet order: [2-frame CELT, 1-frame SILK, 2-frame hybrid],
    and batches whose first packets do not hold every mode."""
    rng = np.random.default_rng(23)

    def body(L):
        return rng.integers(0, 256, size=L, dtype=np.uint8).tobytes()

    C, S, H = pkg.TOC_CELT_FB_STEREO, pkg.TOC_SILK_NB_STEREO, pkg.TOC_HYBRID_FB_STEREO
    batches = [
        [bytes([C | 1]) + body(2 * 70), bytes([S]) + body(40), bytes([H | 1]) + body(2 * 90)],
        [bytes([C | 1]) + body(2 * 70), bytes([S]) + body(40)],                       # step 0 = [C, S], all[0..2) = [C, C]
        [bytes([S | 1]) + body(2 * 30), bytes([C]) + body(100)],                      # step 0 = [S, C], all[0..2) = [S, S]
        [bytes([H | 3, 3]) + body(3 * 60), bytes([H | 3, 3]) + body(3 * 60), bytes([S]) + body(33), bytes([C]) + body(77)],
        [bytes([C | 3, 3]) + body(3 * 50)] * 5 + [bytes([S | 1]) + body(2 * 25)] + [bytes([H]) + body(64)],
    ]
    for bi, packets in enumerate(batches):
        n = len(packets)
        gpu_ctx.streams_alloc(n, 2)
        dec = [oracle.decoder(2) for _ in range(n)]
        for d in dec:
            d.init()
        for rnd in range(2):  # (a second round: the streams' state advanced for every packet of the first)
            pcm, res = gpu_ctx.decode_packets(np.arange(n), packets, frame_capacity=3)
            for i in range(n):
                o, r = dec[i].decode_cap(packets[i], 3)
                assert res[i] == r, (bi, rnd, i, int(res[i]), r)
                assert r > 0 and np.array_equal(pcm[i, :r], o[:r]), (bi, rnd, i)


@pytest.mark.parametrize("registered", [False, True])
def test_large_regular_call_in_slices(pkg, oracle, gpu_ctx, registered):
    """The host-buffer path's large regular call (>= 4096 packets of one frame each): one framing pass, the entropy kernels once,
    the arithmetic kernels in slices with each slice's PCM leaving behind it -- into a landing zone and on into the caller's
    pageable array, or straight into the caller's array when that is page-locked (Context.host_register).  Three modes in one call
    (ragged lengths), three calls in a row, every sample of every packet against the oracle; then an irregular call of the same
    size (one two-frame packet, one empty packet) takes the general flow and must match as well."""
    rng = np.random.default_rng(23)
    n, rounds = 6000, 3
    tocs = [pkg.TOC_SILK_NB_STEREO, pkg.TOC_HYBRID_FB_STEREO, pkg.TOC_CELT_FB_STEREO]
    calls = []
    for r in range(rounds + 1):
        pk = [bytes([tocs[i % 3]]) + rng.integers(0, 256, size=int(rng.integers(8, 200)), dtype=np.uint8).tobytes() for i in range(n)]
        if r == rounds:
            pk[7] = bytes([pk[7][0] | 1]) + pk[7][1:41] + pk[7][1:41]
            pk[11] = b""
        calls.append(pk)
    dec = [oracle.decoder(2) for _ in range(n)]
    for d in dec:
        d.init()
    gpu_ctx.streams_alloc(n, 2)
    ids = np.arange(n, dtype=np.int32)
    raw = np.zeros(n * 2 * 960 * 2 + 4096, dtype=np.int16)
    out2 = raw[(-raw.ctypes.data) % 4096 // 2:][:n * 2 * 960 * 2].reshape(n, 2 * 960, 2)
    if registered:
        gpu_ctx.host_register(out2)
    try:
        for r, pk in enumerate(calls):
            cap = 2 if r == rounds else 1
            lens = np.array([len(p) for p in pk], dtype=np.int32)
            offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
            arena = np.frombuffer(b"".join(pk) + b"\0", dtype=np.uint8)
            out = out2.reshape(-1)[:n * cap * 960 * 2].reshape(n, cap * 960, 2)
            out[...] = 0x5555
            _, res = gpu_ctx.decode_packets_arena(ids, arena, offs, lens, frame_capacity=cap, pcm=out)
            for i in range(n):
                o, want = dec[i].decode_cap(pk[i], cap)
                assert res[i] == want, (r, i, int(res[i]), want)
                if want > 0:
                    assert np.array_equal(out[i, :want], o[:want]), (r, i)
            if r == rounds:
                assert res[7] == 1920 and res[11] < 0
    finally:
        if registered:
            gpu_ctx.host_unregister(out2)
