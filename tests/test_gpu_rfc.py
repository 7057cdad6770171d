"""RFC mode on the GPU (include/opusgpu.h OPUSGPU_MODE_RFC) against the oracle's RFC mode
(oc_decoder_set_rfc -- itself parity-UNPINNED: the reference cannot decode these durations and no libopus exists here):
all 32 TOC configurations x frame-count codes 0..3, mono and stereo packets in mono and stereo decoders, every packet at the
duration its TOC names (2.5 ... 60 ms frames, up to 120 ms per packet), state carried over a sequence of packets with
configuration switches (incl. hybrid -> SILK-only: the silence-frame fade-out).  Through the C ABI (opusgpu_decode_packets)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dur(toc):
    if toc & 0x80:
        return (48000 << ((toc >> 3) & 3)) // 400
    if (toc & 0x60) == 0x60:
        return 960 if toc & 8 else 480
    a = (toc >> 3) & 3
    return 2880 if a == 3 else (48000 << a) // 100


def make_packet(rng, cfg, stereo, code, L):
    toc = (cfg << 3) | (4 if stereo else 0) | code
    body = lambda k: rng.integers(0, 256, k, dtype=np.uint8).tobytes()
    if code == 0:
        return bytes([toc]) + body(L)
    if code == 1:
        return bytes([toc]) + body(2 * L)
    if code == 2:
        L = min(L, 250)
        return bytes([toc, L]) + body(L + int(rng.integers(2, 120)))
    cnt = int(rng.integers(1, 5))
    while dur(toc) * cnt > 5760:
        cnt -= 1
    return bytes([toc, cnt]) + body(cnt * L)


def _run(pkg, oracle, ctx, channels, plan, seed):
    """plan(stream, step, rng) -> (cfg, code); every stream decodes len-of-plan packets, one per step."""
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
        checked = 0
        for f in range(steps):
            pk = []
            for s in range(n):
                cfg, code = plan["pick"](s, f, rng)
                stereo = (channels == 2) if rng.random() < 0.85 else bool(rng.integers(2))
                pk.append(make_packet(rng, cfg, stereo, code, int(rng.choice([3, 8, 20, 40, 80, 120, 160, 300]))))
            pcm, res = ctx.decode_packets(np.arange(n), pk, frame_capacity=6)
            for s in range(n):
                ref, r = decs[s].decode(pk[s])
                assert res[s] == r, (f, s, hex(pk[s][0]), int(res[s]), r)
                if r <= 0:
                    continue
                toc = pk[s][0]
                fs, pch = dur(toc), (2 if toc & 4 else 1)
                got, want = pcm[s][:r], ref[:r]
                if not (toc & 0x80) and (toc & 0x60) != 0x60 and pch < channels:
                    # Q3: a mono SILK-only packet in a stereo decoder defines only the first fs * pch linear entries of a frame
                    for k in range(r // fs):
                        a = got[k * fs:(k + 1) * fs].reshape(-1)[:fs * pch]
                        b = want[k * fs:(k + 1) * fs].reshape(-1)[:fs * pch]
                        assert np.array_equal(a, b), (f, s, hex(toc), k)
                else:
                    assert np.array_equal(got, want), (f, s, hex(toc), int(np.argmax((got != want).any(axis=1))))
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
    hybrid -> SILK-only among them)"""
    state = {}

    def pick(s, f, rng):
        if s not in state or rng.random() < 0.4:
            state[s] = int(rng.integers(32))
        return state[s], int(rng.choice([0, 0, 0, 1, 2, 3]))

    plan = {"streams": 384, "steps": 8, "pick": pick}
    assert _run(pkg, oracle, gpu_ctx, channels, plan, 77 + channels) > 384 * 5


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
