"""RFC mode on the GPU (include/opusgpu.h OPUSGPU_MODE_RFC) against the oracle's RFC mode
(oc_decoder_set_rfc -- itself parity-UNPINNED: the reference cannot decode these durations and no libopus exists here):
all 32 TOC configurations x frame-count codes 0..3, mono and stereo packets in mono and stereo decoders, every packet at the
duration its TOC names (2.5 ... 60 ms frames, up to 120 ms per packet), state carried over a sequence of packets with
configuration switches (incl. hybrid -> SILK-only: the silence-frame fade-out).  Through the C ABI (opusgpu_decode_packets)."""
import numpy as np
import pytest

import ctypes as C

from rfc_common import dur, make_packet, mode_bw, frame_payloads, same_pcm, fec_plan, redundancy_packet

pytestmark = pytest.mark.gpu


def _run(pkg, oracle, ctx, channels, plan, seed, p_loss=0.0, p_dtx=0.0, p_fec=0.0):
    """plan["pick"](stream, step, rng) -> (cfg, code); every stream decodes plan["steps"] packets, one per step.  p_loss: the
    share of packets that are lost (empty: concealed for as long as the stream's last packet was); p_dtx: the share of packets
    whose frames carry 0 or 1 bytes; p_fec: the share of packets before which a packet was lost and is recovered from the
    packet's forward error correction data (decode_packets_fec), after which the packet is decoded normally.
    -> packets compared."""
    rng = np.random.default_rng(seed)
    n, steps = plan["streams"], plan["steps"]
    ctx.set_mode(True)
    try:
        ctx.streams_alloc(n, channels)
        decs = []
        for s in range(n):
            d = oracle.decoder(channels)
            d.init()
            d.set_rfc(True)
            decs.append(d)
        last = [None] * n  # (frame count, frame duration, packet channels, mode) of the stream's last packet that framed
        checked = 0
        oracle.lib.oc_decode_fec.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int]
        for f in range(steps):
            pk = []
            for s in range(n):
                if rng.random() < p_loss:
                    pk.append(b"")
                    continue
                cfg, code = plan["pick"](s, f, rng)
                stereo = (channels == 2) if rng.random() < 0.85 else bool(rng.integers(2))
                L = int(rng.integers(0, 2)) if rng.random() < p_dtx else int(rng.choice([3, 8, 20, 40, 80, 120, 160, 300]))
                if rng.random() < plan.get("p_redundant", 0.0):  # a hybrid packet with its redundancy flag set
                    pk.append(redundancy_packet(rng, channels if rng.random() < 0.85 else None)[0])
                else:
                    pk.append(make_packet(rng, cfg, stereo, code, L))
            who = [s for s in range(n) if len(pk[s]) and rng.random() < p_fec and frame_payloads(oracle, pk[s]) is not None]
            if who:  # the packet before pk[s] was lost: recover it from pk[s]
                pcm, res = ctx.decode_packets_fec(np.array(who), [pk[s] for s in who], frame_capacity=6)
                for j, s in enumerate(who):
                    toc = pk[s][0]
                    total, pieces, use = fec_plan((last[s][0], last[s][1], last[s][3]) if last[s] else None, toc, channels)
                    before = decs[s].prev_mode()
                    ref = np.zeros((5760, channels), dtype=np.int16)
                    r = oracle.lib.oc_decode_fec(decs[s].h, pk[s], len(pk[s]), ref.ctypes.data, total)
                    assert res[j] == r, (f, s, "fec", hex(toc), int(res[j]), r)
                    if r <= 0:
                        continue
                    lpch = last[s][2] if last[s] else channels
                    at, mode_now = 0, before
                    for w in pieces:  # concealed in the mode of the frame before, over the last packet's channels
                        ok, _ = same_pcm(pcm[j][at:at + w], ref[at:at + w], w, [0], mode_now, mode_now, lpch, channels)
                        assert ok, (f, s, "fec: concealment piece at", at)
                        at += w
                    if use:
                        fs, pch, m = dur(toc), (2 if toc & 4 else 1), mode_bw(toc)[0]
                        ln = len(frame_payloads(oracle, pk[s])[0])
                        ok, _ = same_pcm(pcm[j][at:at + fs], ref[at:at + fs], fs, [ln], mode_now, m, pch, channels)
                        assert ok, (f, s, "fec frame", hex(toc))
                    checked += 1
            pcm, res = ctx.decode_packets(np.arange(n), pk, frame_capacity=6)
            for s in range(n):
                before = decs[s].prev_mode()
                if len(pk[s]) == 0:
                    cnt, fs, pch = last[s][:3] if last[s] else (1, 960, channels)
                    ref, r = decs[s].conceal(cnt * fs)
                    lens, toc_mode, label = [0] * cnt, before, "lost"
                else:
                    ref, r = decs[s].decode(pk[s])
                    pays = frame_payloads(oracle, pk[s])
                    toc = pk[s][0]
                    fs, pch, toc_mode, label = dur(toc), (2 if toc & 4 else 1), mode_bw(toc)[0], hex(toc)
                    if pays is not None:
                        last[s] = (len(pays), fs, pch, toc_mode)
                        lens = [len(p) for p in pays]
                assert res[s] == r, (f, s, label, int(res[s]), r)
                if r <= 0:
                    continue
                ok, k = same_pcm(pcm[s][:r], ref[:r], fs, lens, before, toc_mode, pch, channels)
                assert ok, (f, s, label, "frame", k)
                checked += 1
        return checked
    finally:
        ctx.set_mode(False)


@pytest.mark.parametrize("channels", [2, 1])
def test_rfc_all_configs_and_codes(pkg, oracle, gpu_ctx, channels):
    """stream s decodes configuration s % 32 with frame-count code (s // 32) % 4, four packets in a row"""
    plan = {"streams": 256, "steps": 4, "pick": lambda s, f, rng: (s % 32, (s // 32) % 4)}
    assert _run(pkg, oracle, gpu_ctx, channels, plan, 11 + channels) > 256 * 3


@pytest.mark.parametrize("channels", [2, 1])
def test_rfc_configuration_switches(pkg, oracle, gpu_ctx, channels):
    """every stream walks its own random sequence of configurations and codes (mode, bandwidth and duration switches,
    hybrid -> SILK-only among them); redundant CELT frames (RFC 6716 section 4.5.1): SILK-only frames with random payloads
    carry one almost always, hybrid packets with the flag set are mixed in from tests/golden/rfc_hybrid_redundancy_seeds.json"""
    state = {}

    def pick(s, f, rng):
        if s not in state or rng.random() < 0.4:
            state[s] = int(rng.integers(32))
        return state[s], int(rng.choice([0, 0, 0, 1, 2, 3]))

    plan = {"streams": 384, "steps": 8, "pick": pick, "p_redundant": 0.06}
    assert _run(pkg, oracle, gpu_ctx, channels, plan, 77 + channels) > 384 * 5


@pytest.mark.parametrize("channels", [2, 1])
def test_rfc_loss_path(pkg, oracle, gpu_ctx, channels):
    """lost packets (len == 0) and DTX frames conceal on the GPU, bit-exact to the oracle's RFC mode -- SILK PLC +
    comfort noise + the glue to the next decoded frame, CELT's noise-based concealment, hybrid both; random configuration
    walks so that losses follow (and are followed by) every mode, bandwidth and frame duration; several losses in a row."""
    state = {}

    def pick(s, f, rng):
        if s not in state or rng.random() < 0.3:
            state[s] = int(rng.integers(32))
        return state[s], int(rng.choice([0, 0, 0, 1, 2, 3]))

    plan = {"streams": 512, "steps": 10, "pick": pick, "p_redundant": 0.05}
    assert _run(pkg, oracle, gpu_ctx, channels, plan, 501 + channels, p_loss=0.3, p_dtx=0.06) > 512 * 7


@pytest.mark.parametrize("channels", [2, 1])
def test_rfc_forward_error_correction(pkg, oracle, gpu_ctx, channels):
    """opus_decode(decode_fec = 1): a lost packet recovered from the NEXT packet's LBRR frames (SILK-only and hybrid; per channel
    and internal frame a concealment where the packet carries no copy; hybrid's CELT layer concealed), plain concealment
    around CELT-only packets; mixed with ordinary losses and DTX frames"""
    state = {}

    def pick(s, f, rng):
        if s not in state or rng.random() < 0.25:
            state[s] = int(rng.integers(32)) if rng.random() < 0.3 else int(rng.integers(16))  # (mostly SILK-only / hybrid)
        return state[s], int(rng.choice([0, 0, 0, 1, 2, 3]))

    plan = {"streams": 384, "steps": 10, "pick": pick}
    assert _run(pkg, oracle, gpu_ctx, channels, plan, 801 + channels, p_loss=0.15, p_dtx=0.04, p_fec=0.3) > 384 * 9


def test_rfc_loss_before_any_packet_and_after_reset(pkg, oracle, gpu_ctx):
    """a loss with nothing decoded yet is 20 ms of zeros; after a stream reset the same"""
    ctx = gpu_ctx
    ctx.set_mode(True)
    try:
        ctx.streams_alloc(4, 2)
        pcm, res = ctx.decode_packets(np.arange(4), [b""] * 4, frame_capacity=6)
        assert (res == 960).all() and not pcm.any()
        rng = np.random.default_rng(9)
        pk = [make_packet(rng, 31, True, 0, 80) for _ in range(4)]
        pcm, res = ctx.decode_packets(np.arange(4), pk, frame_capacity=6)
        assert (res == 960).all() and pcm.any()
        pcm, res = ctx.decode_packets(np.arange(4), [None] * 4, frame_capacity=6)
        assert (res == 960).all() and pcm[:, :960].any()  # concealed from the CELT state
        ctx.streams_reset(0, 4, True)
        pcm, res = ctx.decode_packets(np.arange(4), [b""] * 4, frame_capacity=6)
        assert (res == 960).all() and not pcm.any()
    finally:
        ctx.set_mode(False)


def test_reference_mode_conceals_nothing(pkg, gpu_ctx):
    """Reference mode has no concealment (Q8) -- an empty packet there is NOT a lost packet but what the reference's own
    empty-packet branch makes of it (src/opus_decoder.cpp:290-308; tests/test_empty_packets.py has the known answers): a frame of
    no bytes in the stream's last mode.  Fresh streams are in mode 0: SILK runs, CELT refuses (src/celt.cpp:2225)."""
    ctx = gpu_ctx
    ctx.streams_alloc(2, 2)
    pcm, res = ctx.decode_packets(np.arange(2), [b"", None])
    assert (res == -18).all()  # ERR_OPUS_CELT_BAD_ARG


def test_reference_mode_unchanged_after_rfc(pkg, oracle, gpu_ctx):
    """switching the mode off again gives the reference-exact decode (every frame 960 samples)"""
    ctx = gpu_ctx
    ctx.set_mode(True)
    ctx.set_mode(False)
    n = 64
    ctx.streams_alloc(n, 2)
    pay = pkg.lcg_payloads(n, 2, 60)
    ref, ok = oracle.batch_decode(2, 0xE4, pay)  # CELT FB 2.5 ms TOC: decodes as 20 ms in reference mode (Q6)
    for f in range(2):
        pk = [bytes([0xE4]) + pay[f, s].tobytes() for s in range(n)]
        pcm, res = ctx.decode_packets(np.arange(n), pk)
        assert (res == 960).all()
        assert (pcm == ref[:, f]).all()


@pytest.mark.parametrize("channels", [2, 1])
def test_rfc_celt_loss_bursts_pitch_then_noise(pkg, oracle, gpu_ctx, channels):
    """CELT-only streams of every bandwidth and frame duration (configurations 16 .. 31) losing half of their packets: bursts of one
    to ten lost frames -- the first five of a burst extrapolated from the pitch period (og_plc.hpp: the search and the LPC analysis at
    the burst's first frame, period and filter kept for the rest), the later ones noise at the decaying band energies -- and the
    first decoded frame behind a burst blending into the concealment's overlap tail.  Every sample against the oracle's RFC mode."""
    plan = {"streams": 640, "steps": 14, "pick": lambda s, f, rng: (16 + s % 16, int(rng.choice([0, 0, 1, 3])))}
    assert _run(pkg, oracle, gpu_ctx, channels, plan, 901 + channels, p_loss=0.5, p_dtx=0.03) > 640 * 9
