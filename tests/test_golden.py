"""Committed golden vectors: the oracle still reproduces them (CPU), and the GPU path matches them (-m gpu)."""
import json
import os

import numpy as np
import pytest

from oracle_py import fnv1a_u16

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    meta = json.load(open(os.path.join(HERE, "oracle_vectors.json")))["modes"]
    pcm = np.load(os.path.join(HERE, "oracle_vectors_pcm.npz"))
    return meta, pcm


def test_survey_kats_are_what_the_oracle_tests_pin():
    k = json.load(open(os.path.join(HERE, "survey_kats.json")))
    assert k["kat1"]["hashes"] == ["818f2314", "9b451028", "729184e0"]
    assert k["kat2"]["hashes"] == ["165e980f", "d19ff868", "cfee645b"]


def test_oracle_reproduces_committed_vectors(pkg, oracle):
    meta, pcm = _load()
    for name, m in meta.items():
        pay = pkg.lcg_payloads(m["streams"], m["frames"], m["payload_len"])
        ref, ok = oracle.batch_decode(2, m["toc"], pay)
        assert ok == m["streams"] * m["frames"]
        got = [[fnv1a_u16(ref[s, f]) for f in range(m["frames"])] for s in range(m["streams"])]
        assert got == m["fnv1a_u16"], name
        assert np.array_equal(ref[:2, :pcm[name].shape[1]], pcm[name])


@pytest.mark.gpu
def test_gpu_matches_committed_vectors(pkg, gpu_ctx):
    meta, pcm = _load()
    for name, m in meta.items():
        n, frames = m["streams"], m["frames"]
        pay = pkg.lcg_payloads(n, frames, m["payload_len"])
        gpu_ctx.streams_alloc(n, 2)
        for f in range(frames):
            pk = [bytes([m["toc"]]) + pay[f, s].tobytes() for s in range(n)]
            out, res = gpu_ctx.decode_packets(np.arange(n), pk)
            assert (res == 960).all()
            assert [fnv1a_u16(out[s]) for s in range(n)] == [m["fnv1a_u16"][s][f] for s in range(n)], (name, f)
            if f < pcm[name].shape[1]:
                assert np.array_equal(out[:2], pcm[name][:, f])


def _sequences():
    return json.load(open(os.path.join(HERE, "oracle_sequences.json")))


def test_oracle_reproduces_committed_sequences(oracle):
    """Quirk sequences (mono decoders, Q3, multi-frame packets / Q6, mode switches / Q4): return code and PCM hash per call."""
    seq = _sequences()
    cap = seq["frame_capacity"]
    for name, q in seq["sequences"].items():
        d = oracle.decoder(q["channels"])
        d.init()
        for i, c in enumerate(q["calls"]):
            pcm, r = d.decode_cap(bytes.fromhex(c["packet"]), cap)
            assert r == c["ret"], (name, i)
            if c["fnv1a_u16"] is not None:
                assert fnv1a_u16(pcm[:r]) == c["fnv1a_u16"], (name, i)
    rets = [c["ret"] for c in seq["sequences"]["multiframe_packets"]["calls"]]
    assert 1920 in rets and 2880 in rets and -2 in rets and -4 in rets  # the cases the sequence is there for


@pytest.mark.gpu
def test_gpu_matches_committed_sequences(pkg, gpu_ctx):
    seq = _sequences()
    cap = seq["frame_capacity"]
    for name, q in seq["sequences"].items():
        gpu_ctx.streams_alloc(1, q["channels"])
        for i, c in enumerate(q["calls"]):
            out, res = gpu_ctx.decode_packets([0], [bytes.fromhex(c["packet"])], frame_capacity=cap)
            assert res[0] == c["ret"], (name, i, res[0], c["ret"])
            if c["fnv1a_u16"] is not None:
                assert fnv1a_u16(out[0, :c["ret"]]) == c["fnv1a_u16"], (name, i)


def _rfc_sequences():
    return json.load(open(os.path.join(HERE, "rfc_sequences.json")))


def test_oracle_reproduces_committed_rfc_sequences(oracle):
    """RFC mode (parity-unpinned: the vectors freeze the oracle's behaviour, no reference decoder made them): every frame
    duration and frame-count code, lost packets, DTX frames, forward error correction, redundancy -- code and PCM hash per call."""
    import ctypes as C
    oracle.lib.oc_decode_fec.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int]
    seq = _rfc_sequences()
    kinds = set()
    for name, q in seq["sequences"].items():
        d = oracle.decoder(q["channels"])
        d.init()
        d.set_rfc(True)
        prev = 960
        for i, c in enumerate(q["calls"]):
            kinds.add(c["kind"])
            if c["kind"] == "lost":
                pcm, r = d.conceal(c["ret"])
            elif c["kind"] == "fec":
                pcm = np.zeros((5760, q["channels"]), dtype=np.int16)
                p = bytes.fromhex(c["packet"])
                r = oracle.lib.oc_decode_fec(d.h, p, len(p), pcm.ctypes.data, c["ret"])
            else:
                pcm, r = d.decode(bytes.fromhex(c["packet"]))
            assert r == c["ret"], (name, i, c["kind"])
            if c["fnv1a_u16"] is not None:
                assert fnv1a_u16(pcm[:r]) == c["fnv1a_u16"], (name, i, c["kind"])
    assert kinds == {"packet", "lost", "fec"}
    rets = {c["ret"] for q in seq["sequences"].values() for c in q["calls"]}
    assert {120, 240, 480, 960, 1920, 2880} <= rets  # every frame duration is in there


@pytest.mark.gpu
def test_gpu_matches_committed_rfc_sequences(pkg, gpu_ctx):
    seq = _rfc_sequences()
    gpu_ctx.set_mode(True)
    try:
        for name, q in seq["sequences"].items():
            gpu_ctx.streams_alloc(1, q["channels"])
            for i, c in enumerate(q["calls"]):
                if c["kind"] == "lost":
                    out, res = gpu_ctx.decode_packets([0], [b""], frame_capacity=seq["frame_capacity"])
                elif c["kind"] == "fec":
                    out, res = gpu_ctx.decode_packets_fec([0], [bytes.fromhex(c["packet"])], frame_capacity=seq["frame_capacity"])
                else:
                    out, res = gpu_ctx.decode_packets([0], [bytes.fromhex(c["packet"])], frame_capacity=seq["frame_capacity"])
                assert res[0] == c["ret"], (name, i, c["kind"], int(res[0]), c["ret"])
                if c["fnv1a_u16"] is not None:
                    assert fnv1a_u16(out[0, :c["ret"]]) == c["fnv1a_u16"], (name, i, c["kind"])
    finally:
        gpu_ctx.set_mode(False)
