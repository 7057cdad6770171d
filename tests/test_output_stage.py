"""Output stage of the player (src/main.cpp playChunk / playSample / Gain -> opusgpu_output_stage_device).
The oracle (oracle/oc_output.c) is pinned by known answers worked out by hand from the reference's source -- main.cpp
needs <Arduino.h> and has no vectors of its own, so this row's parity is pinned no further than that.  CPU: the kernel's
per-word source in host emulation against the oracle.  GPU: the kernel through the C ABI against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = [(bits, ch, mono) for bits in (16, 8) for ch in (1, 2) for mono in (0, 1)]


def _oracle_lib():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboc_oracle.so"))
    lib.oc_output_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.oc_output_stage.restype = C.c_long
    return lib


def oracle_block(lib, blk, valid, bits, channels, mono, vol):
    blk = np.ascontiguousarray(blk, dtype=np.int16)
    out = np.zeros(2 * max(valid, 1), dtype=np.uint32)
    n = lib.oc_output_stage(blk.ctypes.data, valid, bits, channels, mono, vol, out.ctypes.data)
    return out[:max(n, 0)], n


def word(left, right):
    return ((right & 0xffff) << 16) | (left & 0xffff)


def test_known_answers_from_the_reference_source():
    lib = _oracle_lib()
    # 16-bit stereo, unity volume: each sample halved (main.cpp:236), (s * 64) >> 6 = s (:142), right in the high half (:145)
    out, n = oracle_block(lib, [1000, -1000, 32767, -32768, 1, -1], 3, 16, 2, 0, 64)
    assert n == 3 and list(out) == [word(500, -500), word(16383, -16384), word(0, -1)]  # -1 >> 1 = -1 (arithmetic shift)
    # volume 32 = half, 0 = silence, 128 = double
    assert list(oracle_block(lib, [1000, -1001], 1, 16, 2, 0, 32)[0]) == [word(250, -251)]  # (-501 * 32) >> 6 = -250.5 -> -251
    assert list(oracle_block(lib, [1000, -1001], 1, 16, 2, 0, 0)[0]) == [0]
    assert list(oracle_block(lib, [1000, -1001], 1, 16, 2, 0, 128)[0]) == [word(1000, -1002)]
    # above 16 bits the halves wrap: 16383 * 255 = 4177665, >> 6 = 65276 = 0xfefc in the low half; the right channel's bit 16 is shifted out
    assert list(oracle_block(lib, [32767, 32767], 1, 16, 2, 0, 255)[0]) == [0xfefcfefc]
    # force mono: (l + r) / 2 with C division (towards zero), then both channels (:213)
    assert list(oracle_block(lib, [3, -6, -32768, -32768], 2, 16, 2, 1, 64)[0]) == [word(-1, -1), word(-16384, -16384)]  # -3/2 = -1; -1 >> 1 = -1
    # 16-bit mono: the sample on both channels (:196-197); m_validSamples counts samples
    assert list(oracle_block(lib, [200, -200], 2, 16, 1, 0, 64)[0]) == [word(100, 100), word(-100, -100)]
    # 8-bit stereo: low byte left, high byte right, (x - 128) << 8 (:231-234): 0x80 -> 0, 0xff -> 32512 >> 1 = 16256, 0x00 -> -16384
    assert list(oracle_block(lib, [np.int16(-128), 0x0080], 2, 8, 2, 0, 64)[0]) == [word(0, 16256), word(0, -16384)]  # 0xff80, 0x0080
    # 8-bit stereo, force mono: (x + y) / 2 in unsigned bytes (:179): (0x80 + 0xff) / 2 = 0xbf -> (191 - 128) << 8 >> 1 = 8064
    assert list(oracle_block(lib, [np.int16(-128)], 1, 8, 2, 1, 64)[0]) == [word(8064, 8064)]
    # 8-bit mono: every word plays twice, low byte first (:156-165)
    assert list(oracle_block(lib, [0x7f81], 1, 8, 1, 0, 64)[0]) == [word(128, 128), word(-128, -128)]  # 0x81 -> 256 >> 1; 0x7f -> -256 >> 1
    # other bit depths play nothing (:222-223)
    assert oracle_block(lib, [1, 2], 1, 24, 2, 0, 64)[1] == -1


def test_kernel_source_in_host_emulation_matches_the_oracle():
    lib = _oracle_lib()
    emu = C.CDLL(os.path.join(ROOT, "tests", "emul", "libog_emul.so"))
    emu.emu_output_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(7)
    edge = np.array([0, 1, -1, 2, -2, 3, -3, 127, 128, 255, 256, -255, -256, -257, 32767, -32768, 32766, -32767, 16384, -16384], dtype=np.int16)
    for bits, ch, mono in MODES:
        for vol in (0, 1, 21, 63, 64, 65, 127, 128, 200, 255):
            valid = int(rng.integers(0, 400))
            blk = rng.integers(-32768, 32768, size=2 * valid + 2, dtype=np.int16)
            k = min(len(edge), len(blk))
            blk[:k] = rng.permutation(edge)[:k]
            want, n = oracle_block(lib, blk, valid, bits, ch, mono, vol)
            got = np.zeros(2 * valid + 2, dtype=np.uint32)
            m = emu.emu_output_block(blk.ctypes.data, valid, vol, mono, bits, ch, got.ctypes.data)
            assert m == n and np.array_equal(got[:m], want), (bits, ch, mono, vol)
    # every pair of 16-bit samples that matters for the mono average's rounding and the wrap of the packed halves
    blk = np.array([[a, b] for a in edge for b in edge], dtype=np.int16).reshape(-1)
    for mono in (0, 1):
        for vol in (64, 255):
            want, n = oracle_block(lib, blk, len(blk) // 2, 16, 2, mono, vol)
            got = np.zeros(len(blk), dtype=np.uint32)
            assert emu.emu_output_block(blk.ctypes.data, len(blk) // 2, vol, mono, 16, 2, got.ctypes.data) == n
            assert np.array_equal(got[:n], want)
    # settings the reference's setters refuse: nothing is played
    assert emu.emu_output_block(blk.ctypes.data, 4, 64, 0, 24, 2, got.ctypes.data) == 0
    assert emu.emu_output_block(blk.ctypes.data, 4, 64, 0, 16, 3, got.ctypes.data) == 0


@pytest.mark.gpu
def test_output_stage_on_the_gpu(pkg, gpu_ctx):
    """One setting for all blocks (every mode; aligned layout = the 16-byte path, and an odd layout = the word path),
    valid counts taken from a result array with failed frames, then per-block settings."""
    lib = _oracle_lib()
    ctx = gpu_ctx
    rng = np.random.default_rng(3)
    n, cap = 300, 960
    SENT = 0xDEADBEEF

    def run(pcm, pcm_stride, valid, cfgs, scalar, i2s_stride, shift):
        """pcm: int16 [n * pcm_stride]; valid: int32 [n] or an int; cfgs: OUTPUT_CFG_DTYPE [n] or None; shift: elements
        by which both device buffers are moved off their 16-byte alignment"""
        d_pcm = ctx.dev_alloc(pcm.nbytes + 64)
        d_out = ctx.dev_alloc(4 * n * i2s_stride + 64)
        d_valid = ctx.dev_alloc(4 * n)
        d_cfg = ctx.dev_alloc(4 * n)
        try:
            p_pcm, p_out = C.c_void_p(d_pcm.value + 2 * shift), C.c_void_p(d_out.value + 4 * shift)
            ctx.h2d(p_pcm, pcm)
            out = np.full(n * i2s_stride, SENT, dtype=np.uint32)
            ctx.h2d(p_out, out)
            per_block_valid = not np.isscalar(valid)
            if per_block_valid:
                ctx.h2d(d_valid, np.ascontiguousarray(valid, dtype=np.int32))
            if cfgs is not None:
                ctx.h2d(d_cfg, cfgs)
            ctx.output_stage_device(n, cap, p_pcm, pcm_stride, p_out, i2s_stride, d_valid=d_valid if per_block_valid else None,
                                    valid_all=0 if per_block_valid else int(valid), d_cfgs=d_cfg if cfgs is not None else None, **scalar)
            ctx.synchronize()
            ctx.d2h(out, p_out)
            return out.reshape(n, i2s_stride)
        finally:
            for p in (d_pcm, d_out, d_valid, d_cfg):
                ctx.dev_free(p)

    def check(out, pcm, pcm_stride, valid, cfg_of):
        for b in range(n):
            bits, ch, mono, vol = cfg_of(b)
            v = int(valid if np.isscalar(valid) else valid[b])
            v = min(max(v, 0), cap)
            want, cnt = oracle_block(lib, pcm[b * pcm_stride:(b + 1) * pcm_stride], v, bits, ch, mono, vol)
            cnt = max(cnt, 0) if ch in (1, 2) else 0
            assert np.array_equal(out[b, :cnt], want[:cnt]), (b, bits, ch, mono, vol, v)
            assert (out[b, cnt:] == SENT).all(), (b, "words past the block's count were written")

    for stride, i2s_stride, shift in ((2 * cap, 2 * cap, 0), (2 * cap + 2, 2 * cap + 1, 1)):
        pcm = rng.integers(-32768, 32768, size=n * stride, dtype=np.int16)
        for bits, ch, mono in MODES:
            for vol, valid in ((64, cap), (200, 957), (17, 2), (64, 0)):
                out = run(pcm, stride, valid, None, dict(volume=vol, force_mono=bool(mono), bits=bits, channels=ch), i2s_stride, shift)
                check(out, pcm, stride, valid, lambda b: (bits, ch, mono, vol))
        # valid counts as a decode step reports them: 960, error codes, short blocks
        results = rng.choice(np.array([960, 960, 960, -4, -2, 0, 1, 3, 5, 480, 959, 5000], dtype=np.int32), size=n)
        out = run(pcm, stride, results, None, dict(volume=90, force_mono=False, bits=16, channels=2), i2s_stride, shift)
        check(out, pcm, stride, results, lambda b: (16, 2, 0, 90))
        # per-block settings, including ones the setters refuse
        cfgs = np.zeros(n, dtype=pkg.OUTPUT_CFG_DTYPE)
        cfgs["volume"] = rng.integers(0, 256, n)
        cfgs["force_mono"] = rng.integers(0, 2, n)
        cfgs["bits"] = rng.choice([16, 16, 16, 8, 8, 24], n)
        cfgs["channels"] = rng.choice([2, 2, 1, 3], n)
        out = run(pcm, stride, results, cfgs, {}, i2s_stride, shift)
        check(out, pcm, stride, results, lambda b: (int(cfgs["bits"][b]), int(cfgs["channels"][b]), int(cfgs["force_mono"][b]), int(cfgs["volume"][b])))
    # argument errors
    with pytest.raises(pkg.OpusGpuError):
        ctx.output_stage_device(1, 960, C.c_void_p(256), 1920, C.c_void_p(256), 960, valid_all=960, bits=24)
    with pytest.raises(pkg.OpusGpuError):
        ctx.output_stage_device(1, 960, C.c_void_p(256), 1920, C.c_void_p(256), 960, valid_all=961)
