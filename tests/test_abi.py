"""The C-ABI shared library loads, exports every symbol include/opusgpu.h declares, and fails LOUDLY when no
GPU is usable (there is no CPU fallback).  Host-only entry points are checked against the oracle's parser."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_lib()
    hdr = open(os.path.join(ROOT, "include", "opusgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(opusgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in opusgpu.h but not exported"
    assert sorted(pkg.EXPORTS) == declared
    assert lib.opusgpu_version() >= 100
    assert lib.opusgpu_stream_state_bytes() > 16384


def test_library_exports_the_reference_surface(pkg):
    """Every function include/opus_decoder.h and include/opusfile.h declare (the reference's own C++ prototypes,
    src/opus_decoder.h:165-218, src/opusfile.h:144-156) is defined in libopusgpu.so."""
    import subprocess
    pkg.load_lib()
    syms = subprocess.run(["nm", "-DC", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r"\b([a-z_0-9]+)\(", syms))
    want = set()
    for h in ("opus_decoder.h", "opusfile.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        want |= set(re.findall(r"^\s*(?:[A-Za-z_0-9]+[ \*]+)+((?:opus|op)_[a-z_0-9]+)\s*\(", text, flags=re.M))
    assert {"opus_decoder_get_nb_samples", "opus_multistream_decode", "op_read_stereo", "opus_init_decoder"} <= want
    missing = sorted(want - have)
    assert not missing, f"declared but not exported: {missing}"


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = pkg.load_lib().opusgpu_ctx_create(0, C.byref(h))
    assert rc == pkg.OPUSGPU_ERR_NO_DEVICE and not h
    with pytest.raises(pkg.OpusGpuError):
        pkg.Context(0)


def test_packet_framing_matches_oracle(pkg, oracle):
    lib = oracle.lib
    lib.oc_packet_parse.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    checked = 0
    for _ in range(3000):
        L = int(rng.integers(1, 400))
        pkt = bytearray(rng.integers(0, 256, L, dtype=np.uint8).tobytes())
        if rng.random() < 0.5 and L > 2:           # bias towards plausible code-3 headers
            pkt[0] = (pkt[0] & 0xFC) | 3
            pkt[1] = (pkt[1] & 0xC0) | int(rng.integers(0, 8))
        pkt = bytes(pkt)
        size = (C.c_int16 * 48)()
        toc = C.c_uint8()
        off = C.c_int()
        n_ref = lib.oc_packet_parse(pkt, L, 0, C.byref(toc), size, C.byref(off), None)
        got = pkg.packet_to_frames(pkt, stream=7)
        if n_ref < 0:
            assert got == n_ref
            continue
        assert len(got) == n_ref
        o = off.value
        for k, (offset, ln, flags) in enumerate(got):
            assert (offset, ln) == (o, size[k])
            o += size[k]
        checked += 1
    assert checked > 500
    assert pkg.packet_to_frames(b"") < 0
