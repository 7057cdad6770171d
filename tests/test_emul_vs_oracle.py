"""CPU test of the KERNEL SOURCE: the HIP headers compiled in the test-only one-lane host emulation
(tests/emul) are fuzzed against the oracle.  Catches arithmetic / control-flow slips in the device code
without a GPU (cross-lane synchronisation is only exercised by the -m gpu tests)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul", "libog_emul.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.dirname(EMUL), "-s"])
    lib = C.CDLL(EMUL)
    lib.emu_state_size.restype = C.c_int
    lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
    lib.emu_stream_reset.argtypes = [C.c_void_p]
    lib.emu_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return lib


def _mode_bw(toc):
    if toc & 0x80:
        bw = 1102 + ((toc >> 5) & 3)
        return 1002, (1101 if bw == 1102 else bw)
    if (toc & 0x60) == 0x60:
        return 1001, (1105 if toc & 0x10 else 1104)
    return 1000, 1101 + ((toc >> 5) & 3)


CFGS = {0: [1, 5, 9], 1: [13, 15], 2: [19, 23, 27, 31]}


@pytest.mark.parametrize("seed,modes,switch", [(1, [2], False), (2, [0], False), (3, [1], False), (4, [0, 1, 2], True)])
def test_emulated_kernels_match_oracle(emu, oracle, seed, modes, switch):
    rng = np.random.default_rng(seed)
    st = C.create_string_buffer(emu.emu_state_size())
    out = np.zeros((960, 2), dtype=np.int16)
    for stream in range(40):
        channels = int(rng.integers(1, 3))
        d = oracle.decoder(channels)
        d.init()
        emu.emu_stream_init(st, channels)
        mode = int(rng.choice(modes))
        for f in range(8):
            if switch and rng.random() < 0.3:
                mode = int(rng.choice(modes))
            stereo = (channels == 2) if (not switch or rng.random() < 0.9) else bool(rng.integers(2))
            toc = (int(rng.choice(CFGS[mode])) << 3) | (4 if stereo else 0)
            L = int(rng.choice([40, 120, 160, 3, 333]))
            kind = int(rng.integers(12))
            body = bytes(L) if kind == 0 else (b"\xff" * L if kind == 1 else rng.integers(0, 256, L, dtype=np.uint8).tobytes())
            ref, r = d.decode(bytes([toc]) + body)
            m, bw = _mode_bw(toc)
            out[:] = 0
            r2 = emu.emu_decode_frame(st, body, L, m, bw, 2 if stereo else 1, out.ctypes.data)
            assert r == r2, (stream, f, hex(toc), r, r2)
            if r > 0:
                pch = 2 if stereo else 1
                ncmp = 960 * pch if (m == 1000 and pch < channels) else 960 * channels  # Q3: see test_gpu_modes
                assert np.array_equal(out.reshape(-1)[:ncmp], ref[:960].reshape(-1)[:ncmp]), (stream, f, hex(toc))


@pytest.mark.parametrize("seed,modes", [(11, [0]), (12, [0, 1]), (13, [0, 1, 2])])
def test_entropy_half_past_kept_by_the_parse_matches_the_state(emu, oracle, seed, modes):
    """Pipelined SILK-only steps (og_api.hip): the parse kernel keeps its own copy of what the entropy half needs of the frames
    before (SilkShadow) and computes what the synthesis WILL write to the state.  Here, frame by frame in emulation: random walks
    over SILK NB / MB / WB and hybrid configurations, mono and stereo packets in mono and stereo decoders, empty-ish and error
    frames -- after every frame that decoded, the copy must equal the state field by field, and PCM and return codes the oracle's.
    A CELT frame in between goes the ordinary way and ends the copy's epoch, as the host does for every step of another kind."""
    emu.emu_decode_frame_shadowed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    emu.emu_shadow_vs_state.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)
    st = C.create_string_buffer(emu.emu_state_size())
    out = np.zeros((960, 2), dtype=np.int16)
    shadowed = 0
    for stream in range(60):
        channels = int(rng.integers(1, 3))
        d = oracle.decoder(channels)
        d.init()
        emu.emu_stream_init(st, channels)
        shadow = C.create_string_buffer(128)
        epoch = 1
        mode = int(rng.choice(modes))
        for f in range(14):
            if rng.random() < 0.3:
                mode = int(rng.choice(modes))
            stereo = (channels == 2) if rng.random() < 0.8 else bool(rng.integers(2))
            toc = (int(rng.choice(CFGS[mode])) << 3) | (4 if stereo else 0)
            L = int(rng.choice([40, 120, 160, 3, 1, 333, 1275]))
            kind = int(rng.integers(12))
            body = bytes(L) if kind == 0 else (b"\xff" * L if kind == 1 else rng.integers(0, 256, L, dtype=np.uint8).tobytes())
            ref, r = d.decode(bytes([toc]) + body)
            m, bw = _mode_bw(toc)
            out[:] = 0
            if m == 1002:
                r2 = emu.emu_decode_frame(st, body, L, m, bw, 2 if stereo else 1, out.ctypes.data)
                epoch += 1
            else:
                r2 = emu.emu_decode_frame_shadowed(st, shadow, epoch, body, L, m, bw, 2 if stereo else 1, out.ctypes.data)
                if r2 > 0:
                    assert emu.emu_shadow_vs_state(st, shadow) == 0, (stream, f, hex(toc), emu.emu_shadow_vs_state(st, shadow))
                    shadowed += 1
            assert r == r2, (stream, f, hex(toc), r, r2)
            if r > 0:
                pch = 2 if stereo else 1
                ncmp = 960 * pch if (m == 1000 and pch < channels) else 960 * channels
                assert np.array_equal(out.reshape(-1)[:ncmp], ref[:960].reshape(-1)[:ncmp]), (stream, f, hex(toc))
    assert shadowed > 300


def test_silk_working_set_sized_for_narrowband(oracle):
    """og_silk_nb.hip builds the SILK synthesis with OG_SILK_LDS_FRAME = 160 (buffers for 20 ms at 8 kHz).  The same sizing in host
    emulation, array bounds trapped (-fsanitize=bounds): narrowband SILK-only streams, mono and stereo, against the oracle.
    (The GPU kernel's tight LAYOUT of those buffers is GPU-only code: tests/test_gpu_pipeline.py, test_gpu_modes.py.)"""
    lib_path = os.path.join(os.path.dirname(EMUL), "libog_emul_nb.so")
    subprocess.check_call(["make", "-C", os.path.dirname(EMUL), "-s", "libog_emul_nb.so"])
    lib = C.CDLL(lib_path)
    lib.emu_state_size.restype = C.c_int
    lib.emu_stream_init.argtypes = [C.c_void_p, C.c_int]
    lib.emu_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(160)
    out = np.zeros((960, 2), dtype=np.int16)
    for s in range(60):
        channels = int(rng.integers(1, 3))
        d = oracle.decoder(channels)
        d.init()
        st = C.create_string_buffer(lib.emu_state_size())
        lib.emu_stream_init(st, channels)
        for f in range(8):
            stereo = (channels == 2) if rng.random() < 0.8 else bool(rng.integers(2))
            toc = (1 << 3) | (4 if stereo else 0)  # SILK-only NB 20 ms
            L = int(rng.choice([0, 1, 2, 7, 20, 40, 80, 160, 400]))
            body = rng.integers(0, 256, L, dtype=np.uint8).tobytes()
            ref, r = d.decode(bytes([toc]) + body)
            out[:] = 0
            r2 = lib.emu_decode_frame(st, body, L, 1000, 1101, 2 if stereo else 1, out.ctypes.data)
            assert r == r2 == 960, (s, f, r, r2)
            k = 960 * channels if (stereo or channels == 1) else 960  # (a mono packet in a stereo decoder defines 960 entries, Q3)
            assert np.array_equal(ref.reshape(-1)[:k], out.reshape(-1)[:k]), (s, f)
