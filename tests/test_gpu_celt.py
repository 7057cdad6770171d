"""GPU parity: HIP CELT path vs the CPU oracle, through the C ABI (include/opusgpu.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _streams(pkg, toc, n_streams, n_frames, payload_len, seed_base=0x9E3779B9):
    pay = pkg.lcg_payloads(n_streams, n_frames, payload_len, seed_base)
    return pay, [[bytes([toc]) + pay[f, s].tobytes() for f in range(n_frames)] for s in range(n_streams)]


def test_celt_fb_stereo_device_path_bit_exact(pkg, oracle, gpu_ctx):
    n, frames, L = 512, 6, 160
    pay, pk = _streams(pkg, pkg.TOC_CELT_FB_STEREO, n, frames, L)
    ref, rets = oracle.decode_streams(2, pk)
    assert (rets == 960).all()
    ctx = gpu_ctx
    ctx.streams_alloc(n, 2)
    d_desc = ctx.dev_alloc(16 * n)
    d_arena = ctx.dev_alloc(n * (L + 1) + 16)
    d_pcm = ctx.dev_alloc(n * 960 * 2 * 2)
    d_res = ctx.dev_alloc(4 * n)
    out = np.zeros((n, 960, 2), dtype=np.int16)
    res = np.zeros(n, dtype=np.int32)
    for f in range(frames):
        arena, descs = pkg.build_step(pkg.TOC_CELT_FB_STEREO, pay[f])
        ctx.h2d(d_arena, arena)
        ctx.h2d(d_desc, descs)
        ctx.decode_step_device(n, d_desc, d_arena, d_pcm, d_res)
        ctx.synchronize()
        ctx.d2h(out, d_pcm)
        ctx.d2h(res, d_res)
        assert (res == 960).all(), res[res != 960][:8]
        bad = np.nonzero((out != ref[:, f]).reshape(n, -1).any(axis=1))[0]
        assert bad.size == 0, f"frame {f}: {bad.size} streams differ, first {bad[:5]}"
    for p in (d_desc, d_arena, d_pcm, d_res):
        ctx.dev_free(p)


def test_celt_host_path_mixed_lengths_and_mono(pkg, oracle, gpu_ctx):
    rng = np.random.default_rng(7)
    for channels in (2, 1):
        n, frames = 96, 5
        pk = []
        for s in range(n):
            row = []
            for f in range(frames):
                L = int(rng.choice([160, 160, 40, 2, 700, 1274]))
                stereo = (channels == 2) if rng.random() < 0.85 else bool(rng.integers(2))
                cfg = int(rng.choice([19, 23, 27, 31]))
                toc = (cfg << 3) | (4 if stereo else 0)
                kind = rng.integers(12)
                body = bytes(L) if kind == 0 else (b"\xff" * L if kind == 1 else rng.integers(0, 256, L, dtype=np.uint8).tobytes())
                row.append(bytes([toc]) + body)
            pk.append(row)
        ref, rets = oracle.decode_streams(channels, pk)
        ctx = gpu_ctx
        ctx.streams_alloc(n, channels)
        for f in range(frames):
            pcm, res = ctx.decode_packets(np.arange(n), [pk[s][f] for s in range(n)], frame_capacity=1)
            assert (res == rets[:, f]).all()
            ok = res > 0
            assert (pcm[ok] == ref[ok, f]).all()


def test_device_path_reports_bad_descriptors(pkg, oracle, gpu_ctx):
    """include/opusgpu.h, opusgpu_decode_step_device: a stream index out of range or a frame length outside 0..1275 comes
    back as OPUSGPU_BAD_ARG in that frame's result; the other frames of the step are decoded as if it were not there."""
    ctx = gpu_ctx
    for toc, L in ((pkg.TOC_CELT_FB_STEREO, 160), (pkg.TOC_SILK_NB_STEREO, 40), (pkg.TOC_HYBRID_FB_STEREO, 120)):
        n = 200
        pay = pkg.lcg_payloads(n, 2, L, seed_base=0xBAD + L)
        ref, ok = oracle.batch_decode(2, toc, pay)
        assert ok == n * 2
        ctx.streams_alloc(n, 2)
        d_desc, d_arena = ctx.dev_alloc(16 * n), ctx.dev_alloc(n * (L + 1) + 16)
        d_pcm, d_res = ctx.dev_alloc(n * 960 * 2 * 2), ctx.dev_alloc(4 * n)
        out = np.zeros((n, 960, 2), dtype=np.int16)
        res = np.zeros(n, dtype=np.int32)
        bad = {5: ("stream", -1), 17: ("stream", n), 40: ("stream", 1 << 30), 63: ("len", 1276), 64: ("len", -1), 130: ("len", 70000)}
        for f in range(2):
            arena, descs = pkg.build_step(toc, pay[f])
            descs = descs.copy()
            if f == 0:  # the first step carries the broken descriptors; their streams simply skip that frame
                for slot, (field, value) in bad.items():
                    descs[field][slot] = value
            ctx.h2d(d_arena, arena)
            ctx.h2d(d_desc, descs)
            ctx.decode_step_device(n, d_desc, d_arena, d_pcm, d_res)
            ctx.synchronize()
            ctx.d2h(out, d_pcm)
            ctx.d2h(res, d_res)
            good = np.ones(n, dtype=bool)
            if f == 0:
                good[list(bad)] = False
                assert (res[~good] == -1).all(), (hex(toc), res[~good])
                assert (res[good] == 960).all()
                assert (out[good] == ref[good, 0]).all()
        for p in (d_desc, d_arena, d_pcm, d_res):
            ctx.dev_free(p)


def test_optional_kernels_in_child_processes(pkg):
    """Code paths that og_debug.hpp's switches select (read once per process): the general reconstruction kernel for every frame
    (OPUSGPU_FAST_RECON=0), the single-kernel path for every frame (OPUSGPU_SPLIT=0: round 1's design, and what Q4 frames run on).  Each runs
    tests/pipeline_knob_worker.py -- 8 steps of 8,192 CELT-FB streams, pipelined, as a window and call by call, every sample
    against the oracle -- in a child process."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_knob_worker.py")
    for env in ({"OPUSGPU_FAST_RECON": "0"}, {"OPUSGPU_SPLIT": "0"}):
        out = subprocess.run([sys.executable, worker], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "knob worker ok" in out.stdout, (env, out.stdout[-400:], out.stderr[-400:])
