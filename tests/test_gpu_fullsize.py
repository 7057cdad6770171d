"""GPU parity at BASELINE.json's full batch sizes, every PCM sample of every stream of every
step compared with the CPU oracle:

  * C2  65,536 CELT-FB stereo streams (the bench workload),
  * C3  65,536 SILK-NB stereo streams,
  *     65,536 hybrid FB stereo streams,
  * C4  262,144 hybrid FB stereo streams,

plus: every frame of every step reports 960 samples.  The oracle side runs on the host threads this process is granted.
The bench's own configuration -- C2 with pipelined steps (opusgpu_set_pipeline), the mode mask naming CELT-only frames, all
steps queued back to back so that the three kernels of neighbouring steps really overlap -- has a test of its own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEPS = 3


def _run(pkg, oracle, ctx, toc, L, n, steps):
    pay = pkg.lcg_payloads(n, steps, L)  # [steps, n, L], one LCG stream per decoder stream
    ref, ok = oracle.batch_decode_threads(2, toc, pay)  # [n, steps, 960, 2]
    assert ok == n * steps
    ctx.streams_alloc(n, 2)
    d_desc = ctx.dev_alloc(16 * n)
    d_arena = ctx.dev_alloc(n * (L + 1) + 16)
    d_pcm = ctx.dev_alloc(n * 960 * 2 * 2)
    d_res = ctx.dev_alloc(4 * n)
    out = np.zeros((n, 960, 2), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    try:
        for f in range(steps):
            arena, descs = pkg.build_step(toc, pay[f])
            ctx.h2d(d_arena, arena)
            ctx.h2d(d_desc, descs)
            ctx.decode_step_device(n, d_desc, d_arena, d_pcm, d_res)
            ctx.synchronize()
            ctx.d2h(out, d_pcm)
            ctx.d2h(res, d_res)
            assert (res == 960).all(), f"step {f}: {(res != 960).sum()} frames failed"
            bad = np.nonzero((out != ref[:, f]).reshape(n, -1).any(axis=1))[0]
            assert bad.size == 0, f"step {f}: {bad.size} of {n} streams differ from the oracle, first {bad[:5]}"
    finally:
        for p in (d_desc, d_arena, d_pcm, d_res):
            ctx.dev_free(p)


def test_fullsize_celt(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_CELT_FB_STEREO, 160, 65536, STEPS)


def test_fullsize_silk(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_SILK_NB_STEREO, 40, 65536, STEPS)


def test_fullsize_hybrid(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_HYBRID_FB_STEREO, 120, 65536, STEPS)


def test_fullsize_hybrid_c4(pkg, oracle, gpu_ctx):
    _run(pkg, oracle, gpu_ctx, pkg.TOC_HYBRID_FB_STEREO, 120, 262144, 2)


@pytest.mark.parametrize("which", ["celt", "silk_nb", "hybrid"])
def test_fullsize_pipelined_queued(pkg, oracle, gpu_ctx, which):
    """What `python bench.py` times: 65,536 CELT-FB / SILK-NB / hybrid-FB streams, pipelined steps with the mode mask (CELT-only: parse,
    reconstruction and de-emphasis of neighbouring steps overlap; SILK-only and hybrid: the parse kernels of step k + 1 next to the
    synthesis of step k), no synchronisation between the steps (every step's tables resident before the first call), queued one
    call per step and, a second time on fresh streams, as one window.  Every sample of every step against the oracle."""
    toc, L = {"celt": (pkg.TOC_CELT_FB_STEREO, 160), "silk_nb": (pkg.TOC_SILK_NB_STEREO, 40), "hybrid": (pkg.TOC_HYBRID_FB_STEREO, 120)}[which]
    ctx, n, steps = gpu_ctx, 65536, 8
    pay = pkg.lcg_payloads(n, steps, L, seed_base=0x0C2B1A5)
    ref, ok = oracle.batch_decode_threads(2, toc, pay)
    assert ok == n * steps
    ctx.streams_alloc(n, 2)
    d_desc = [ctx.dev_alloc(16 * n) for _ in range(steps)]
    d_arena = [ctx.dev_alloc(n * (L + 1) + 16) for _ in range(steps)]
    d_pcm = [ctx.dev_alloc(n * 960 * 2 * 2) for _ in range(steps)]
    d_res = [ctx.dev_alloc(4 * n) for _ in range(steps)]
    out = np.zeros((n, 960, 2), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    try:
        for f in range(steps):
            arena, descs = pkg.build_step(toc, pay[f])
            ctx.h2d(d_arena[f], arena)
            ctx.h2d(d_desc[f], descs)
        ctx.set_pipeline(True)
        for window in (False, True):
            if window:
                ctx.streams_reset(0, n)
                ctx.decode_steps_device([n] * steps, d_desc, d_arena, d_pcm, d_res, modes=pkg.toc_modes(toc))
            else:
                for f in range(steps):
                    ctx.decode_step_device(n, d_desc[f], d_arena[f], d_pcm[f], d_res[f], modes=pkg.toc_modes(toc))
            ctx.synchronize()
            for f in range(steps):
                ctx.d2h(out, d_pcm[f])
                ctx.d2h(res, d_res[f])
                assert (res == 960).all(), f"step {f}: {(res != 960).sum()} frames failed"
                bad = np.nonzero((out != ref[:, f]).reshape(n, -1).any(axis=1))[0]
                assert bad.size == 0, f"window {window} step {f}: {bad.size} of {n} streams differ from the oracle, first {bad[:5]}"
    finally:
        ctx.set_pipeline(False)
        for p in d_desc + d_arena + d_pcm + d_res:
            ctx.dev_free(p)
