"""GPU parity of PIPELINED decode steps (opusgpu_set_pipeline, include/opusgpu.h): step k+1's CELT parse runs on the library's
own stream next to step k's reconstruction.  All steps of a run are queued back to back with no synchronisation in between
(so the overlap really happens), every PCM sample and return code is compared with the oracle, which decodes packet after
packet as the reference does (src/opus_decoder.cpp:931 -> :280 -> :154)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = np.array([1, 5, 9, 13, 15, 19, 23, 27, 31])  # the 20 ms configuration of every mode / bandwidth
LENS = np.array([0, 1, 2, 3, 5, 8, 13, 20, 30, 40, 60, 80, 100, 120, 160, 200, 320, 500, 800, 1275])


def make_walk(rng, n, frames, channels, configs=CONFIGS, p_home=0.8):
    """Random per-stream sequences of 20 ms packets: arena bytes, per-frame offsets / packet lengths / TOCs."""
    home = rng.choice(configs, n)
    cfg = np.where(rng.random((frames, n)) < p_home, home[None, :], rng.choice(configs, (frames, n)))
    stereo = rng.random((frames, n)) < (0.85 if channels == 2 else 0.15)
    toc = (cfg << 3 | np.where(stereo, 4, 0)).astype(np.uint8)
    lens = rng.choice(LENS, (frames, n), p=np.r_[np.full(4, 0.02), np.full(15, 0.06), 0.02])
    plen = (lens + 1).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(plen.reshape(-1))[:-1]]).reshape(frames, n)
    arena = rng.integers(0, 256, int(plen.sum()) + 16, dtype=np.uint8)
    arena[offs.reshape(-1)] = toc.reshape(-1)
    return arena, offs, plen, lens, toc


def desc_flags(toc):
    mode = np.where(toc & 0x80, 2, np.where((toc & 0x60) == 0x60, 1, 0)).astype(np.int32)
    bw_c = ((toc >> 5) & 3).astype(np.int32)
    bw = np.where(mode == 2, np.where(bw_c == 0, 0, bw_c + 1), np.where(mode == 1, np.where(toc & 0x10, 4, 3), bw_c))
    return (mode | bw << 2 | np.where(toc & 4, 32, 0)).astype(np.int32), mode


def run_queued(pkg, ctx, channels, arena, offs, lens, toc, pipeline, sync_every=0, reset_at=None, modes=0, one_pcm=False, streams=None,
               window=False, flags=None):
    """All steps queued back to back (tables of every step resident before the first call).  Returns PCM [frames, n, 960*ch]
    and result codes [frames, n].  window: the steps go in ONE opusgpu_decode_steps_device call (cut where a reset or a
    synchronisation point is asked for) instead of one call per step."""
    frames, n = offs.shape
    if flags is None:  # (given: descriptors that are not a packet's own -- the frames of empty packets, tests/test_empty_packets.py)
        flags, _ = desc_flags(toc)
    ctx.streams_alloc(n, channels)
    ctx.set_pipeline(pipeline)
    frame_bytes = n * 960 * channels * 2
    d_arena = ctx.dev_alloc(arena.size)
    d_desc = [ctx.dev_alloc(16 * n) for _ in range(frames)]
    d_pcm = [ctx.dev_alloc(frame_bytes) for _ in range(frames)]
    d_res = [ctx.dev_alloc(4 * n) for _ in range(frames)]
    ctx.h2d(d_arena, arena)
    descs = np.zeros(n, dtype=pkg.DESC_DTYPE)
    descs["stream"] = np.arange(n, dtype=np.int32) if streams is None else streams
    for f in range(frames):
        descs["offset"] = (offs[f] + 1).astype(np.int32)
        descs["len"] = lens[f].astype(np.int32)
        descs["flags"] = flags[f]
        ctx.h2d(d_desc[f], descs)  # (synchronous: complete in device memory before any step is queued)
    if window:
        cuts = sorted({0, frames} | ({reset_at} if reset_at is not None else set()) |
                      (set(range(sync_every, frames, sync_every)) if sync_every else set()))
        for a, b in zip(cuts[:-1], cuts[1:]):
            if reset_at is not None and a == reset_at:
                ctx.streams_reset(0, n, full=False)
            ctx.decode_steps_device([n] * (b - a), d_desc[a:b], [d_arena] * (b - a), d_pcm[a:b], d_res[a:b], modes=modes)
            if sync_every and b % sync_every == 0:
                ctx.synchronize()
    else:
        for f in range(frames):
            if reset_at is not None and f == reset_at:
                ctx.streams_reset(0, n, full=False)
            ctx.decode_step_device(n, d_desc[f], d_arena, d_pcm[f], d_res[f], modes=modes)
            if sync_every and (f + 1) % sync_every == 0:
                ctx.synchronize()
    ctx.synchronize()
    pcm = np.zeros((frames, n, 960 * channels), dtype=np.int16)
    res = np.zeros((frames, n), dtype=np.int32)
    for f in range(frames):
        ctx.d2h(pcm[f], d_pcm[f])
        ctx.d2h(res[f], d_res[f])
    for p in [d_arena] + d_desc + d_pcm + d_res:
        ctx.dev_free(p)
    ctx.set_pipeline(False)
    return pcm, res


def compare(pcm, res, ref, rets, toc, channels):
    """ref [n, frames, 960, ch], rets [n, frames] from the oracle.  Returns the number of (frame, stream) blocks that differ."""
    frames, n = res.shape
    _, mode = desc_flags(toc)
    bad = 0
    for f in range(frames):
        bad += int((res[f] != rets[:, f]).sum())
        ok = rets[:, f] == 960
        half = ok & (mode[f] == 0) & ((toc[f] & 4) == 0) & (channels == 2)  # Q3: only 960 interleaved entries are defined
        full = ok & ~half
        a, b = pcm[f], ref[:, f].reshape(n, -1)
        bad += int((a[full] != b[full]).any(axis=1).sum())
        if half.any():
            bad += int((a[half][:, :960] != b[half][:, :960]).any(axis=1).sum())
    return bad


@pytest.mark.parametrize("channels", [2, 1])
def test_pipelined_steps_random_mode_walks(pkg, oracle, gpu_ctx, channels):
    """Every mode / bandwidth, switches between frames (incl. the hybrid -> SILK-only transition frames, Q4, whose full-kernel
    pass writes the band energies a following CELT parse reads), payloads of 0 .. 1275 bytes."""
    rng = np.random.default_rng(4100 + channels)
    n, frames = 3072, 14
    arena, offs, plen, lens, toc = make_walk(rng, n, frames, channels)
    ref, rets = oracle.batch_decode_var(channels, arena, offs, plen.astype(np.int32))
    pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=True)
    assert compare(pcm, res, ref, rets, toc, channels) == 0
    pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=True, window=True)
    assert compare(pcm, res, ref, rets, toc, channels) == 0
    # and so is the in-order flow through the same harness
    pcm0, res0 = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=False)
    assert compare(pcm0, res0, ref, rets, toc, channels) == 0


def test_pipelined_celt_only_and_hybrid(pkg, oracle, gpu_ctx):
    """The two workloads the overlap is for: CELT-only FB (the whole parse runs ahead) and hybrid FB (its parse stays behind the
    step's SILK parse); long enough for the double-buffered records to be reused several times."""
    for toc_byte, L in ((pkg.TOC_CELT_FB_STEREO, 160), (pkg.TOC_HYBRID_FB_STEREO, 120)):
        n, frames = 4096, 10
        pay = pkg.lcg_payloads(n, frames, L, seed_base=0x5150 + toc_byte)
        ref, ok = oracle.batch_decode(2, toc_byte, pay)
        assert ok == n * frames
        plen = np.full((frames, n), L + 1, dtype=np.int64)
        offs = np.arange(frames * n, dtype=np.int64).reshape(frames, n) * (L + 1)
        arena = np.zeros(frames * n * (L + 1) + 16, dtype=np.uint8)
        blk = arena[: frames * n * (L + 1)].reshape(frames, n, L + 1)
        blk[:, :, 0] = toc_byte
        blk[:, :, 1:] = pay
        toc = np.full((frames, n), toc_byte, dtype=np.uint8)
        for modes in (0, pkg.toc_modes(toc_byte)):  # not known / named by the caller (CELT-only: the reconstruction runs ahead too)
            for window in (False, True):  # (a window of CELT-only steps: the kernels wait on each other's start counts)
                pcm, res = run_queued(pkg, gpu_ctx, 2, arena, offs, plen - 1, toc, pipeline=True, modes=modes, window=window)
                assert (res == 960).all()
                assert np.array_equal(pcm.reshape(frames, n, 960, 2).transpose(1, 0, 2, 3), ref)
        pcm, res = run_queued(pkg, gpu_ctx, 2, arena, offs, plen - 1, toc, pipeline=False, modes=pkg.toc_modes(toc_byte))
        assert (res == 960).all()
        assert np.array_equal(pcm.reshape(frames, n, 960, 2).transpose(1, 0, 2, 3), ref)


def test_mode_mask_is_checked(pkg, oracle, gpu_ctx):
    """opusgpu_decode_step_device_modes: frames of a mode the caller ruled out come back as OPUSGPU_BAD_ARG, the others decode
    as ever; so does a stream index out of range when the kernels that usually report it were not launched."""
    rng = np.random.default_rng(4300)
    n, frames, channels = 512, 4, 2
    arena, offs, plen, lens, toc = make_walk(rng, n, frames, channels, configs=np.array([31, 31, 31, 27, 15, 1]), p_home=1.0)
    _, mode = desc_flags(toc)
    streams = np.arange(n, dtype=np.int32)
    streams[7] = n + 5  # out of range
    streams[9] = -1
    for pipeline in (False, True):
        pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=pipeline, modes=pkg.HAS_CELT, streams=streams)
        pcm0, res0 = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=False, streams=streams)
        celt = mode == 2
        celt[:, [7, 9]] = False
        assert (res[~celt] == -1).all()                      # hybrid / SILK frames and the two bad stream indices
        assert (res0[:, [7, 9]] == -1).all()
        assert np.array_equal(res[celt], res0[celt])
        ok = celt & (res0 == 960)
        assert ok.sum() > 500 and np.array_equal(pcm[ok], pcm0[ok])


def test_arena_alignment_is_checked(pkg, gpu_ctx):
    """opusgpu.h: d_arena is 16-byte aligned (the parse lanes fetch packets as aligned 16-byte pieces); a call with an arena that is
    not comes back with OPUSGPU_BAD_ARG before anything is queued, for a single step and for a window."""
    import ctypes as C
    n = 64
    gpu_ctx.streams_alloc(n, 2)
    d_arena, d_desc, d_pcm, d_res = gpu_ctx.dev_alloc(4096), gpu_ctx.dev_alloc(16 * n), gpu_ctx.dev_alloc(n * 960 * 2 * 2), gpu_ctx.dev_alloc(4 * n)
    base = d_arena.value if isinstance(d_arena, C.c_void_p) else int(d_arena)
    for skew in (4, 8, 12):
        with pytest.raises(pkg.OpusGpuError) as e:
            gpu_ctx.decode_step_device(n, d_desc, C.c_void_p(base + skew), d_pcm, d_res)
        assert e.value.code == -1
        with pytest.raises(pkg.OpusGpuError) as e:
            gpu_ctx.decode_steps_device([n, n], [d_desc, d_desc], [d_arena, C.c_void_p(base + skew)], [d_pcm, d_pcm], [d_res, d_res])
        assert e.value.code == -1
    gpu_ctx.synchronize()


def test_pipeline_with_synchronisation_points_and_reset(pkg, oracle, gpu_ctx):
    """Synchronising between some steps and resetting the streams in the middle (OPUS_RESET_STATE keeps the band energies, Q5)
    changes nothing; neither does switching the option off and on between runs."""
    rng = np.random.default_rng(4200)
    n, frames, channels = 1024, 10, 2
    arena, offs, plen, lens, toc = make_walk(rng, n, frames, channels, configs=np.array([27, 31, 15, 31, 31]))
    decs = [oracle.decoder(channels) for _ in range(64)]
    pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=True, sync_every=3, reset_at=5)
    pcm0, res0 = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=False, reset_at=5)
    assert np.array_equal(res, res0)
    ok = res0 == 960
    assert np.array_equal(pcm[ok], pcm0[ok])
    pcm1, res1 = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=True, sync_every=4, reset_at=5, window=True)
    assert np.array_equal(res1, res0) and np.array_equal(pcm1[ok], pcm0[ok])
    for s, d in enumerate(decs):  # and a sample of the streams against the oracle, reset included
        d.init()
        for f in range(frames):
            if f == 5:
                d.reset()
            o = int(offs[f, s])
            out, r = d.decode(arena[o:o + int(plen[f, s])].tobytes())
            assert r == res[f, s]
            if r == 960:
                assert np.array_equal(out[:960].reshape(-1), pcm[f, s])


def test_host_path_and_device_steps_mix_with_pipelining_on(pkg, oracle, gpu_ctx):
    """opusgpu_decode_packets queues its own uploads, so it runs in order even with pipelining on -- also right behind pipelined
    device steps of the same streams, and before more of them."""
    n, frames, L = 2048, 9, 160
    toc_byte = pkg.TOC_CELT_FB_STEREO
    pay = pkg.lcg_payloads(n, frames, L, seed_base=0x7171)
    ref, ok = oracle.batch_decode(2, toc_byte, pay)
    assert ok == n * frames
    ctx = gpu_ctx
    ctx.streams_alloc(n, 2)
    ctx.set_pipeline(True)
    try:
        d_pcm, d_res = ctx.dev_alloc(n * 960 * 2 * 2), ctx.dev_alloc(4 * n)
        tabs = []
        for f in range(frames):
            arena, descs = pkg.build_step(toc_byte, pay[f])
            a, d = ctx.dev_alloc(arena.nbytes + 16), ctx.dev_alloc(descs.nbytes)
            ctx.h2d(a, arena)
            ctx.h2d(d, descs)
            tabs.append((a, d))
        got = np.zeros((n, 960, 2), dtype=np.int16)
        res = np.zeros(n, dtype=np.int32)
        for f in range(frames):
            if f % 3 == 2:  # every third frame through the host-buffer path, no synchronisation before it
                pkts = [bytes([toc_byte]) + pay[f, s].tobytes() for s in range(n)]
                pcm, r = ctx.decode_packets(np.arange(n), pkts)
                assert (r == 960).all() and np.array_equal(pcm.reshape(n, 960, 2), ref[:, f])
            else:
                ctx.decode_step_device(n, tabs[f][1], tabs[f][0], d_pcm, d_res, modes=pkg.HAS_CELT)
                if f % 3 == 1:  # (the frame before a host-path call is read back; the other one is overwritten unseen)
                    ctx.synchronize()
                    ctx.d2h(got, d_pcm)
                    ctx.d2h(res, d_res)
                    assert (res == 960).all() and np.array_equal(got, ref[:, f])
        ctx.synchronize()
        for p in [d_pcm, d_res] + [x for t in tabs for x in t]:
            ctx.dev_free(p)
    finally:
        ctx.set_pipeline(False)


def test_placement_knobs_change_no_result(pkg, oracle):
    """What decides which kernel of a pipelined step meets which on the CUs -- the early parse's stream priority and workgroup
    size, how late the host's launches arrive (og_debug.hpp: OPUSGPU_PARSE_PRIORITY, OPUSGPU_PARSE_GROUPS,
    OPUSGPU_LAUNCH_DELAY_US) -- changes no byte.  The switches are read once per process: each setting runs in a child process
    (tests/pipeline_knob_worker.py), which checks a window of CELT-only steps and the same steps one call each against the oracle."""
    import os
    import subprocess
    import sys
    settings = [{"OPUSGPU_PARSE_PRIORITY": "0"}, {"OPUSGPU_PARSE_GROUPS": "1"}, {"OPUSGPU_PARSE_GROUPS": "4"},
                {"OPUSGPU_LAUNCH_DELAY_US": "250"}]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_knob_worker.py")
    for env in settings:
        out = subprocess.run([sys.executable, worker], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "knob worker ok" in out.stdout, (env, out.stdout[-400:], out.stderr[-400:])


def test_narrowband_frames_in_either_synthesis_kernel():
    """Narrowband SILK-only frames run in k_silk_synth_nb (og_silk_nb.hip) -- or, with OPUSGPU_SILK_NB_KERNEL=0, in k_silk_synth like
    every other SILK frame: both against the oracle, each setting in a child process (tests/silk_nb_knob_worker.py)."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "silk_nb_knob_worker.py")
    for env in ({"OPUSGPU_SILK_NB_KERNEL": "0"}, {"OPUSGPU_SILK_NB_KERNEL": "1"}):
        out = subprocess.run([sys.executable, worker], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "silk nb knob worker ok" in out.stdout, (env, out.stdout[-400:], out.stderr[-600:])


def test_windows_that_grow_on_a_fresh_context():
    """Buffer growth under queued windows (tests/window_growth_worker.py): every other test here shares a context whose record
    slots the full-size tests have already grown.  In a child process with a timeout, so that a host deadlock fails the test."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "window_growth_worker.py")
    out = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "window growth worker ok" in out.stdout, (out.stdout[-600:], out.stderr[-600:])


def test_window_with_steps_of_different_sizes(pkg, oracle, gpu_ctx):
    """opusgpu_decode_steps_device: the steps of a window need not have the same number of frames (the wait on the next step's
    parse counts THAT step's workgroups); streams 0 .. n_k - 1 take part in step k."""
    sizes = [4096, 100, 4096, 33, 2500, 1, 4096, 4096]
    n, frames, L = max(sizes), len(sizes), 160
    toc_byte = pkg.TOC_CELT_FB_STEREO
    rng = np.random.default_rng(515)
    pay = rng.integers(0, 256, (frames, n, L), dtype=np.uint8)
    ctx = gpu_ctx
    ctx.streams_alloc(n, 2)
    ctx.set_pipeline(True)
    frees = []
    try:
        tabs = []
        for f in range(frames):
            arena, descs = pkg.build_step(toc_byte, pay[f, :sizes[f]])
            a, d = ctx.dev_alloc(arena.nbytes + 16), ctx.dev_alloc(descs.nbytes)
            ctx.h2d(a, arena)
            ctx.h2d(d, descs)
            p, r = ctx.dev_alloc(sizes[f] * 960 * 2 * 2), ctx.dev_alloc(4 * sizes[f])
            tabs.append((d, a, p, r))
            frees += [a, d, p, r]
        ctx.decode_steps_device(sizes, [t[0] for t in tabs], [t[1] for t in tabs], [t[2] for t in tabs], [t[3] for t in tabs],
                                modes=pkg.HAS_CELT)
        ctx.synchronize()
        decs = {}
        for f in range(frames):  # the oracle: stream s decodes the frames of the steps it takes part in, in order
            got = np.zeros((sizes[f], 960, 2), dtype=np.int16)
            res = np.zeros(sizes[f], dtype=np.int32)
            ctx.d2h(got, tabs[f][2])
            ctx.d2h(res, tabs[f][3])
            assert (res == 960).all(), f
            for s in list(range(min(sizes[f], 40))) + ([sizes[f] - 1] if sizes[f] > 40 else []):
                if s not in decs:
                    decs[s] = (oracle.decoder(2), [])
                    decs[s][0].init()
                    for g in range(f):  # (catch up on the earlier steps this stream took part in)
                        if s < sizes[g]:
                            decs[s][0].decode(bytes([toc_byte]) + pay[g, s].tobytes())
                out, r = decs[s][0].decode(bytes([toc_byte]) + pay[f, s].tobytes())
                assert r == 960 and np.array_equal(out[:960], got[s]), (f, s)
    finally:
        ctx.set_pipeline(False)
        for p in frees:
            ctx.dev_free(p)


@pytest.mark.parametrize("n", [1, 33, 1000])
def test_pipelined_steps_of_odd_sizes(pkg, oracle, gpu_ctx, n):
    """Batch sizes that fill neither a parse wave's 32 lanes nor a parse workgroup's two groups."""
    frames, L = 6, 160
    toc_byte = pkg.TOC_CELT_FB_STEREO
    pay = pkg.lcg_payloads(n, frames, L, seed_base=0x3300 + n)
    ref, ok = oracle.batch_decode(2, toc_byte, pay)
    assert ok == n * frames
    plen = np.full((frames, n), L + 1, dtype=np.int64)
    offs = np.arange(frames * n, dtype=np.int64).reshape(frames, n) * (L + 1)
    arena = np.zeros(frames * n * (L + 1) + 16, dtype=np.uint8)
    blk = arena[: frames * n * (L + 1)].reshape(frames, n, L + 1)
    blk[:, :, 0] = toc_byte
    blk[:, :, 1:] = pay
    toc = np.full((frames, n), toc_byte, dtype=np.uint8)
    for modes in (0, pkg.HAS_CELT):
        for window in (False, True):
            pcm, res = run_queued(pkg, gpu_ctx, 2, arena, offs, plen - 1, toc, pipeline=True, modes=modes, window=window)
            assert (res == 960).all()
            assert np.array_equal(pcm.reshape(frames, n, 960, 2).transpose(1, 0, 2, 3), ref)


SILK_CONFIGS = np.array([1, 5, 9])  # SILK-only NB / MB / WB, 20 ms


@pytest.mark.parametrize("channels,configs,mask", [(2, SILK_CONFIGS, 1), (1, SILK_CONFIGS, 1), (2, np.array([1, 5, 9, 13, 15]), 3),
                                                   (1, np.array([13, 15, 9]), 3), (2, np.array([13, 15]), 2)])
def test_pipelined_silk_only_steps_random_walks(pkg, oracle, gpu_ctx, channels, configs, mask):
    """Steps the caller declares SILK-only (mode mask 1) with pipelining on: the parse of step k + 1 runs on the library's stream
    next to step k's synthesis, from the copy of the entropy half's past that the parse kernel keeps (SilkShadow).  Random walks
    over NB / MB / WB, mono and stereo packets in mono and stereo decoders (the side channel appears, vanishes and restarts), payloads
    of 0 .. 1275 bytes (error frames leave the past alone); queued back to back, one call per step and as a window, with
    synchronisation points and a reset in the middle (which ends the copies' epoch)."""
    rng = np.random.default_rng(5200 + channels + 10 * mask)
    n, frames = 3072, 16
    # (mask 3 / 2: hybrid frames among them or alone -- their CELT parse runs ahead with the SILK parse; SILK-only frames right
    # behind hybrid ones are Q4 transition frames)
    arena, offs, plen, lens, toc = make_walk(rng, n, frames, channels, configs=configs, p_home=0.7)
    ref, rets = oracle.batch_decode_var(channels, arena, offs, plen.astype(np.int32))
    for window in (False, True):
        pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=True, modes=mask, window=window)
        assert compare(pcm, res, ref, rets, toc, channels) == 0
    pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs, lens, toc, pipeline=True, modes=mask, sync_every=3)
    assert compare(pcm, res, ref, rets, toc, channels) == 0
    # a reset between steps 7 and 8: the oracle's decoders are reset there too
    decs = [oracle.decoder(channels) for _ in range(256)]
    sub = slice(0, 256)
    want = np.zeros((256, frames, 960, channels), dtype=np.int16)
    wret = np.zeros((256, frames), dtype=np.int32)
    for s_, d in enumerate(decs):
        d.init()
        for f in range(frames):
            if f == 8:
                d.reset()
            o, r = d.decode(arena[offs[f, s_]:offs[f, s_] + plen[f, s_]].tobytes())
            wret[s_, f] = r
            if r > 0:
                want[s_, f] = o[:960]
    pcm, res = run_queued(pkg, gpu_ctx, channels, arena, offs[:, sub], lens[:, sub], toc[:, sub], pipeline=True, modes=mask, reset_at=8)
    assert compare(pcm, res, want, wret, toc[:, sub], channels) == 0


def test_pipelined_silk_only_steps_between_steps_of_other_kinds(pkg, oracle, gpu_ctx):
    """Runs of declared SILK-only steps between in-order steps (undeclared, any mode) and pipelined CELT-only steps over the SAME
    streams: every switch of kind starts from an idle device and a new epoch, so the parse re-reads the state the other kinds left
    (prev_mode = CELT makes the next SILK frame start from a fresh SILK decoder; a hybrid frame before a SILK-only one makes it a Q4
    transition frame).  Streams also sit out some SILK-only steps (descriptor tables of a part of the streams)."""
    rng = np.random.default_rng(5300)
    channels, n = 2, 2048
    plan = [0, 1, 1, 1, 4, 4, 1, 1, 0, 1, 1, 2, 1, 1, 1, 4, 1, 1]  # per step: mode mask the caller declares (0 = undeclared, any mode)
    frames = len(plan)
    pick = {1: SILK_CONFIGS, 2: np.array([13, 15]), 4: np.array([19, 23, 27, 31]), 0: CONFIGS}
    cfg = np.stack([rng.choice(pick[m], n) for m in plan])
    stereo = rng.random((frames, n)) < 0.85
    toc = (cfg << 3 | np.where(stereo, 4, 0)).astype(np.uint8)
    lens = rng.choice(LENS, (frames, n), p=np.r_[np.full(4, 0.02), np.full(15, 0.06), 0.02])
    plen = (lens + 1).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(plen.reshape(-1))[:-1]]).reshape(frames, n)
    arena = rng.integers(0, 256, int(plen.sum()) + 16, dtype=np.uint8)
    arena[offs.reshape(-1)] = toc.reshape(-1)
    # which streams take part in a step: all, except in some SILK-only steps where a random third sits out
    part = np.ones((frames, n), dtype=bool)
    for f, m in enumerate(plan):
        if m == 1 and f % 2 == 0:
            part[f] = rng.random(n) < 0.67
    # the oracle: every stream decodes the packets of the steps it takes part in
    want = np.zeros((frames, n, 960 * channels), dtype=np.int16)
    wret = np.full((frames, n), -12345, dtype=np.int32)
    for s_ in range(n):
        d = oracle.decoder(channels)
        d.init()
        for f in range(frames):
            if part[f, s_]:
                o, r = d.decode(arena[offs[f, s_]:offs[f, s_] + plen[f, s_]].tobytes())
                wret[f, s_] = r
                if r > 0:
                    want[f, s_] = o[:960].reshape(-1)
    ctx = gpu_ctx
    flags, mode = desc_flags(toc)
    ctx.streams_alloc(n, channels)
    ctx.set_pipeline(True)
    d_arena = ctx.dev_alloc(arena.size)
    ctx.h2d(d_arena, arena)
    bufs = []
    try:
        tables = []
        for f in range(frames):
            ids = np.nonzero(part[f])[0].astype(np.int32)
            descs = np.zeros(len(ids), dtype=pkg.DESC_DTYPE)
            descs["stream"], descs["offset"], descs["len"], descs["flags"] = ids, (offs[f, ids] + 1).astype(np.int32), lens[f, ids].astype(np.int32), flags[f, ids]
            dd, dp, dr = ctx.dev_alloc(16 * len(ids)), ctx.dev_alloc(len(ids) * 960 * channels * 2), ctx.dev_alloc(4 * len(ids))
            ctx.h2d(dd, descs)
            bufs += [dd, dp, dr]
            tables.append((ids, dd, dp, dr))
        for f, (ids, dd, dp, dr) in enumerate(tables):  # queued back to back, no synchronisation by the caller
            ctx.decode_step_device(len(ids), dd, d_arena, dp, dr, modes=plan[f])
        ctx.synchronize()
        for f, (ids, dd, dp, dr) in enumerate(tables):
            pcm = np.zeros((len(ids), 960 * channels), dtype=np.int16)
            res = np.zeros(len(ids), dtype=np.int32)
            ctx.d2h(pcm, dp)
            ctx.d2h(res, dr)
            assert np.array_equal(res, wret[f, ids]), (f, plan[f], int((res != wret[f, ids]).sum()))
            ok = res == 960
            half = ok & (mode[f, ids] == 0) & ((toc[f, ids] & 4) == 0)  # Q3
            full = ok & ~half
            assert np.array_equal(pcm[full], want[f, ids][full]), (f, plan[f])
            assert np.array_equal(pcm[half][:, :960], want[f, ids][half][:, :960]), (f, plan[f])
    finally:
        ctx.set_pipeline(False)
        for p in [d_arena] + bufs:
            ctx.dev_free(p)


def test_keeps_mode_steps_of_both_kinds_on_the_same_celt_streams(pkg, oracle, gpu_ctx):
    """OPUSGPU_STEP_KEEPS_MODE lets a step of ANY mix run as a pipelined SILK / hybrid step (its CELT-only frames reconstructed on
    the step's own stream) and lets declared CELT-only steps (reconstruction on the library's stream) follow without a drain --
    which is only safe while the two are about different streams.  Here they are not: every stream keeps its mode (the caller's
    word is true), but mixed steps over ALL streams alternate with declared CELT-only steps over the SAME CELT-only streams and
    declared SILK / hybrid steps over the others, everything queued back to back.  Every sample against the oracle."""
    rng = np.random.default_rng(4242)
    n = 3 * 1536
    tocs = np.array([pkg.TOC_SILK_NB_STEREO, pkg.TOC_HYBRID_FB_STEREO, pkg.TOC_CELT_FB_STEREO], dtype=np.uint8)[np.arange(n) % 3]
    L = np.array([40, 120, 160])[np.arange(n) % 3]
    celt, rest, every = np.nonzero(np.arange(n) % 3 == 2)[0], np.nonzero(np.arange(n) % 3 != 2)[0], np.arange(n)
    K = pkg.STEP_KEEPS_MODE
    plan = [(every, 7 | K), (celt, 4 | K), (every, 7 | K), (celt, 4 | K), (celt, 4 | K), (rest, 3 | K), (every, 7 | K), (rest, 3 | K),
            (celt, 4 | K), (every, 7 | K), (every, 7 | K), (celt, 4 | K)]
    # per step: one packet for every stream of the step
    flags, _ = desc_flags(tocs)
    steps, per_stream = [], [[] for _ in range(n)]
    for ids, modes in plan:
        lens = L[ids]
        offs = np.concatenate([[0], np.cumsum(lens + 1)[:-1]])
        arena = rng.integers(0, 256, int((lens + 1).sum()) + 16, dtype=np.uint8)
        arena[offs] = tocs[ids]
        for j, s in enumerate(ids):
            per_stream[s].append(arena[offs[j]:offs[j] + lens[j] + 1].tobytes())
        descs = np.zeros(len(ids), dtype=pkg.DESC_DTYPE)
        descs["stream"], descs["offset"], descs["len"], descs["flags"] = ids, offs + 1, lens, flags[ids]
        steps.append((ids, modes, arena, descs))
    ctx = gpu_ctx
    ctx.streams_alloc(n, 2)
    ctx.set_pipeline(True)
    bufs = []
    for ids, modes, arena, descs in steps:
        d_arena, d_desc = ctx.dev_alloc(arena.size), ctx.dev_alloc(16 * len(ids))
        d_pcm, d_res = ctx.dev_alloc(len(ids) * 960 * 2 * 2), ctx.dev_alloc(4 * len(ids))
        ctx.h2d(d_arena, arena)
        ctx.h2d(d_desc, descs)
        bufs.append((d_arena, d_desc, d_pcm, d_res))
    for (ids, modes, _, _), (d_arena, d_desc, d_pcm, d_res) in zip(steps, bufs):
        ctx.decode_step_device(len(ids), d_desc, d_arena, d_pcm, d_res, modes=modes)
    ctx.synchronize()
    ref, rets = oracle.decode_streams(2, per_stream)
    assert (rets[:, :min(len(p) for p in per_stream)] == 960).all()
    seen = np.zeros(n, dtype=np.int64)
    for (ids, _, _, _), (d_arena, d_desc, d_pcm, d_res) in zip(steps, bufs):
        pcm, res = np.zeros((len(ids), 960, 2), dtype=np.int16), np.zeros(len(ids), dtype=np.int32)
        ctx.d2h(pcm, d_pcm)
        ctx.d2h(res, d_res)
        assert (res == 960).all()
        want = ref[ids, seen[ids]]
        bad = np.nonzero((pcm != want).any(axis=(1, 2)))[0]
        assert len(bad) == 0, (len(bad), ids[bad][:8])
        seen[ids] += 1
        for p in (d_arena, d_desc, d_pcm, d_res):
            ctx.dev_free(p)
    ctx.set_pipeline(False)
