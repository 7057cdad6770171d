"""Runs a script of calls through include/opus_decoder.h (tests/player/compat_main.cpp, linked against libopusgpu.so: the GPU)."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tmp_path, steps):
    """steps: ('D', frame_size, packet bytes) | ('N', frame_size): the last packet with len = -1 | ('R',) | ('Q',) | ('F',) | ('P',) | ('X',) | ('V',).  -> one entry per step: for 'D' (ret of opus_decode, ret of
    opus_multistream_decode, PCM int16 [min(ret, frame_size), 2] or None), for 'Q' the 8 int32 of the ctl queries and packet
    helpers, for 'F' the two decoders' final range, for 'P' (ret of the single-stream OPUS_GET_PITCH, its value, ret of the multistream one), for 'R' None.  The program itself checks that both entry points agree on the PCM and that no call writes past
    frame_size samples (guard region)."""
    src = os.path.join(ROOT, "tests", "player", "compat_main.cpp")
    exe = str(tmp_path / "compat")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src,
                           "-L", os.path.join(ROOT, "esp32-opus-player_amd"), "-lopusgpu",
                           "-Wl,-rpath," + os.path.join(ROOT, "esp32-opus-player_amd"), "-o", exe])
    script = b""
    for s in steps:
        script += s[0].encode() + (struct.pack("<iI", s[1], len(s[2])) + s[2] if s[0] == "D" else struct.pack("<i", s[1]) if s[0] == "N" else b"")
    (tmp_path / "script.bin").write_bytes(script)
    log = subprocess.check_output([exe, str(tmp_path / "script.bin"), str(tmp_path / "out.bin")], text=True)
    assert "guards=intact" in log, log
    got = (tmp_path / "out.bin").read_bytes()
    at, out = 0, []
    for s in steps:
        if s[0] == "D":
            ra, rb = struct.unpack_from("<ii", got, at)
            at += 8
            pcm = None
            if ra > 0:
                n = min(ra, s[1])
                pcm = np.frombuffer(got, dtype=np.int16, count=2 * n, offset=at).reshape(n, 2)
                at += 4 * n
            out.append((ra, rb, pcm))
        elif s[0] == "N":
            out.append(struct.unpack_from("<ii", got, at))
            at += 8
        elif s[0] == "Q":
            out.append(struct.unpack_from("<8i", got, at))
            at += 32
        elif s[0] == "F":
            out.append(struct.unpack_from("<2I", got, at))
            at += 8
        elif s[0] == "P":
            out.append(struct.unpack_from("<3i", got, at))
            at += 12
        elif s[0] == "X":
            out.append(struct.unpack_from("<6i", got, at))
            at += 24
        elif s[0] == "V":
            out.append(struct.unpack_from("<19i", got, at))
            at += 76
        else:
            out.append(None)
    assert at == len(got)
    return out
