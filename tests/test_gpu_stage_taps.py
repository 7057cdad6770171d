"""Stage-level parity ON THE GPU (tests/test_stage_taps.py does this for the host emulation of the kernel source):
what every kernel of the split path leaves for the next one is compared with the oracle's taps, so that a difference can be
pinned on a kernel instead of showing up only in the PCM --
  k_celt_parse   -> the parse record: header flags, post-filter parameters, band energies, pulses, tf_res, the coder's final rng
  k_celt_recon*  -> the stream state: the comb-filtered synthesis output of the frame (history ring), the IMDCT overlap tail,
                    the energy histories (i.e. band reconstruction incl. the DPP / phase-major passes, IMDCT, comb filter)
  k_celt_post    -> the PCM (de-emphasis)
  k_silk_parse   -> the SILK record: signal type, gains, pitch lags, both LPC sets, LTP taps and scale
  k_silk_synth   -> the channel state: the synthesis core's output at the internal rate (the DPP-row recurrence), LPC state;
                    then the PCM (stereo un-mixing, resampler)
through opusgpu_debug_stage_taps (include/opusgpu.h), frame by frame with state carried over."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_taps(oracle, d, what, c, dtype, count):
    buf = np.zeros(count, dtype=dtype)
    oracle.lib.oc_taps_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    n = oracle.lib.oc_taps_copy(d.h, what, c, buf.ctypes.data)
    assert n == buf.nbytes, (what, n, buf.nbytes)
    return buf


def _step(pkg, ctx, toc, pay_f, bufs):
    arena, descs = pkg.build_step(toc, pay_f)
    n = len(descs)
    ctx.h2d(bufs["arena"], np.concatenate([arena, np.zeros(16, dtype=np.uint8)]))
    ctx.h2d(bufs["desc"], descs)
    ctx.decode_step_device(n, bufs["desc"], bufs["arena"], bufs["pcm"], bufs["res"])
    ctx.synchronize()
    pcm = np.zeros((n, 960, ctx.channels), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    ctx.d2h(pcm, bufs["pcm"])
    ctx.d2h(res, bufs["res"])
    return pcm, res


def _bufs(ctx, n, L, channels):
    return {"arena": ctx.dev_alloc(n * (L + 1) + 16), "desc": ctx.dev_alloc(16 * n), "pcm": ctx.dev_alloc(n * 960 * channels * 2),
            "res": ctx.dev_alloc(4 * n)}


@pytest.mark.parametrize("L", [160, 60, 400])
def test_celt_stages_on_the_gpu(pkg, oracle, gpu_ctx, L):
    ctx, n, frames = gpu_ctx, 96, 4
    ctx.streams_alloc(n, 2)
    pay = pkg.lcg_payloads(n, frames, L, seed_base=0x7A95 + L)
    bufs = _bufs(ctx, n, L, 2)
    oracle.lib.oc_taps_enable.argtypes = [C.c_void_p]
    decs = []
    for s in range(n):
        d = oracle.decoder(2)
        d.init()
        assert oracle.lib.oc_taps_enable(d.h)
        decs.append(d)
    transients = 0
    try:
        for f in range(frames):
            pcm, res = _step(pkg, ctx, pkg.TOC_CELT_FB_STEREO, pay[f], bufs)
            for s in range(n):
                ref, r = decs[s].decode(bytes([pkg.TOC_CELT_FB_STEREO]) + pay[f, s].tobytes())
                assert r == 960 and res[s] == 960
                where = (L, f, s)
                t = ctx.debug_stage_taps(s)
                h = _oracle_taps(oracle, decs[s], 4, 0, np.int32, 75)
                # ---- k_celt_parse: the record
                assert t.celt_valid and t.celt_ret == 960, where
                got = (t.transient, t.silence, t.intensity, t.dual_stereo, t.spread, t.lm, t.pf_pitch, t.pf_gain, t.pf_tapset, t.anti_collapse_on)
                want = (h[0], h[1], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10])
                assert got == tuple(int(v) for v in want), ("parse record header", where, got, want)
                assert t.celt_rng_final == np.uint32(h[11]), ("range coder state after the frame", where)
                assert (np.array(t.pulses) == h[12:33]).all(), ("pulses", where)
                assert (np.array(t.tf_res) == h[54:75]).all(), ("tf_res", where)
                assert (np.array(t.bandE) == _oracle_taps(oracle, decs[s], 1, 0, np.int16, 42)).all(), ("band energies", where)
                transients += int(h[0])
                # ---- k_celt_recon_fb / k_celt_recon: the state behind the reconstruction
                for c in range(2):
                    post = _oracle_taps(oracle, decs[s], 3, c, np.int32, 960)
                    assert np.count_nonzero(post) > 500, ("the synthesis tap carries no signal", c, where)
                    assert (np.array(t.syn_post[c]) == post).all(), ("synthesis output after the comb filter", c, where)
                    pre = _oracle_taps(oracle, decs[s], 2, c, np.int32, 1080)
                    assert (np.array(t.overlap_tail[c]) == pre[960:1020]).all(), ("IMDCT overlap tail", c, where)
                assert (np.array(t.state_bandE) == _oracle_taps(oracle, decs[s], 1, 0, np.int16, 42)).all(), ("energies in the state", where)
                assert t.state_rng == np.uint32(h[11]), where
                # ---- k_celt_post: the PCM
                assert (pcm[s] == ref[:960]).all(), ("PCM", where)
        assert transients > 0  # both block layouts were seen
    finally:
        for b in bufs.values():
            ctx.dev_free(b)


@pytest.mark.parametrize("toc, L, channels", [(0x0C, 40, 2), (0x4C, 70, 2), (0x48, 60, 1), (0x7C, 120, 2), (0x78, 90, 1)])
def test_silk_stages_on_the_gpu(pkg, oracle, gpu_ctx, toc, L, channels):
    """SILK-only NB / WB and hybrid, stereo and mono: k_silk_parse's record and k_silk_synth's core output against the oracle's
    taps per coded channel; for hybrid also k_celt_parse's record of the CELT layer."""
    ctx, n, frames = gpu_ctx, 64, 5
    ctx.streams_alloc(n, channels)
    pay = pkg.lcg_payloads(n, frames, L, seed_base=0x511C + toc)
    bufs = _bufs(ctx, n, L, channels)
    oracle.lib.oc_silk_taps_copy.argtypes = [C.c_int, C.c_int, C.c_void_p]
    oracle.lib.oc_silk_taps_enable.argtypes = [C.c_int]
    oracle.lib.oc_taps_enable.argtypes = [C.c_void_p]
    oracle.lib.oc_silk_taps_enable(1)

    def otap(what, ch, dtype, count):
        b = np.zeros(count, dtype=dtype)
        assert oracle.lib.oc_silk_taps_copy(what, ch, b.ctypes.data) >= 0
        return b

    hybrid = (toc & 0x60) == 0x60
    decs = []
    for s in range(n):
        d = oracle.decoder(channels)
        d.init()
        assert oracle.lib.oc_taps_enable(d.h)
        decs.append(d)
    voiced = coded = loud = 0
    try:
        for f in range(frames):
            pcm, res = _step(pkg, ctx, toc, pay[f], bufs)
            for s in range(n):
                ref, r = decs[s].decode(bytes([toc]) + pay[f, s].tobytes())  # (the SILK taps are those of this call)
                assert r == 960 and res[s] == 960
                t = ctx.debug_stage_taps(s)
                assert t.silk_valid and t.silk_ret == 0
                for ch in range(2 if toc & 4 else 1):
                    where = (hex(toc), f, s, ch)
                    sb = otap(0, ch, np.int32, 6)
                    if not sb[0]:  # side channel not coded in this frame
                        assert ch == 1 and t.decode_only_middle == 1, where
                        continue
                    coded += 1
                    flen, order = int(sb[3]), int(sb[4])
                    k = t.silk_ch[ch]
                    # ---- k_silk_parse (+ silk_params_lane): the record
                    assert (k.signalType, k.quantOffsetType, k.LTP_scale_Q14) == (sb[1], sb[2], sb[5]), ("signal type / offset type / LTP scale", where)
                    voiced += int(sb[1] == 2)
                    b = otap(1, ch, np.int32, 8)
                    assert (np.array(k.pitchL) == b[:4]).all() and (np.array(k.Gains_Q16) == b[4:]).all(), ("pitch lags / gains", where)
                    b = otap(2, ch, np.int16, 32).reshape(2, 16)
                    assert (np.array(k.PredCoef_Q12).reshape(2, 16)[:, :order] == b[:, :order]).all(), ("LPC coefficients", where)
                    assert (np.array(k.LTPCoef_Q14) == otap(3, ch, np.int16, 20)).all(), ("LTP coefficients", where)
                    # ---- k_silk_synth: the core's output at the internal rate (for 20 ms frames the output history is the frame)
                    xq = otap(4, ch, np.int16, 320)
                    loud += int(np.count_nonzero(xq[:flen]) > flen // 4)
                    assert t.silk_fs_kHz[ch] * 20 == flen, where
                    assert (np.array(t.silk_out[ch])[:flen] == xq[:flen]).all(), ("synthesis core output", where)
                if hybrid:  # the CELT layer's record (bands 17..20)
                    h = _oracle_taps(oracle, decs[s], 4, 0, np.int32, 75)
                    assert t.celt_valid and t.celt_ret == 960
                    assert (t.transient, t.silence, t.spread, t.lm) == (h[0], h[1], h[5], h[6]), ("hybrid: CELT header", hex(toc), f, s)
                    assert (np.array(t.pulses)[17:] == h[12 + 17:33]).all(), ("hybrid: pulses", hex(toc), f, s)
                    assert t.celt_rng_final == np.uint32(h[11])
                # ---- stereo un-mixing, resampler (and for hybrid k_celt_recon + k_celt_post's mix): the PCM
                assert (pcm[s] == ref[:960]).all(), ("PCM", hex(toc), f, s)
        assert coded > n * frames // 2 and voiced > 0
        assert loud > coded // 2  # the core-output comparison is not a comparison of silences
    finally:
        oracle.lib.oc_silk_taps_enable(0)
        for b in bufs.values():
            ctx.dev_free(b)
