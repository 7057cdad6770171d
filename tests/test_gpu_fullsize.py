"""GPU parity at BASELINE.json's full batch size (65,536 streams per step), through size-independent properties:

  * placement independence -- the second half of the batch replays the payloads of the first half: stream i and
    stream i + n/2 must produce identical PCM and identical final coder state, whatever wave / lane / CU decodes them;
  * sampled oracle comparison -- a few hundred streams spread over the batch are compared with the CPU oracle, every
    sample, every step;
  * every frame of every step reports 960 samples.
One test per mode (CELT-only = the bench workload, SILK-only, hybrid)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 65536
STEPS = 3


def _run(pkg, oracle, ctx, toc, L, sample_every):
    half = N // 2
    pay_half = pkg.lcg_payloads(half, STEPS, L)
    pay = np.concatenate([pay_half, pay_half], axis=1)  # [steps, N, L]
    ctx.streams_alloc(N, 2)
    d_desc = ctx.dev_alloc(16 * N)
    d_arena = ctx.dev_alloc(N * (L + 1) + 16)
    d_pcm = ctx.dev_alloc(N * 960 * 2 * 2)
    d_res = ctx.dev_alloc(4 * N)
    out = np.zeros((N, 960, 2), dtype=np.int16)
    res = np.zeros(N, dtype=np.int32)
    picks = np.arange(0, half, sample_every)
    pk = [[bytes([toc]) + pay_half[f, s].tobytes() for f in range(STEPS)] for s in picks]
    ref, rets = oracle.decode_streams(2, pk)
    assert (rets == 960).all()
    for f in range(STEPS):
        arena, descs = pkg.build_step(toc, pay[f])
        ctx.h2d(d_arena, arena)
        ctx.h2d(d_desc, descs)
        ctx.decode_step_device(N, d_desc, d_arena, d_pcm, d_res)
        ctx.synchronize()
        ctx.d2h(out, d_pcm)
        ctx.d2h(res, d_res)
        assert (res == 960).all(), f"step {f}: {(res != 960).sum()} frames failed"
        twin = np.nonzero((out[:half] != out[half:]).reshape(half, -1).any(axis=1))[0]
        assert twin.size == 0, f"step {f}: {twin.size} twin streams differ, first {twin[:5]}"
        bad = np.nonzero((out[picks] != ref[:, f]).reshape(len(picks), -1).any(axis=1))[0]
        assert bad.size == 0, f"step {f}: {bad.size} sampled streams differ from the oracle, first {picks[bad[:5]]}"
    for p in (d_desc, d_arena, d_pcm, d_res):
        ctx.dev_free(p)


def test_fullsize_celt(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_CELT_FB_STEREO, 160, 127)


def test_fullsize_silk(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_SILK_NB_STEREO, 40, 127)


def test_fullsize_hybrid(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_HYBRID_FB_STEREO, 120, 127)
