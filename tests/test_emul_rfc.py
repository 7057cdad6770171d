"""CPU: RFC mode of the kernel source (host emulation of decode_frame_rfc) against the oracle's RFC mode, all 32 TOC
configurations x codes 0..3 with configuration switches (tools/fuzz_emul_rfc.py at a small size); and the oracle's RFC
mode against what is defined independently of it: the sample count of every packet."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_rfc_mode_matches_oracle():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emul"), "-s"])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emul_rfc.py"), "250", "8", "3"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert " 0 mismatches" in p.stdout


def test_oracle_rfc_mode_returns_true_durations(oracle):
    rng = np.random.default_rng(4)
    for channels in (1, 2):
        d = oracle.decoder(channels)
        d.init()
        d.set_rfc(True)
        for cfg in range(32):
            toc = (cfg << 3) | (4 if channels == 2 else 0)
            if toc & 0x80:
                want = (48000 << ((toc >> 3) & 3)) // 400
            elif (toc & 0x60) == 0x60:
                want = 960 if toc & 8 else 480
            else:
                want = [480, 960, 1920, 2880][(toc >> 3) & 3]
            for count in (1, 2):
                if want * count > 5760:
                    continue
                pkt = bytes([toc | 3, count]) + rng.integers(0, 256, 60 * count, dtype=np.uint8).tobytes()
                _, r = d.decode(pkt)
                assert r == want * count, (hex(toc), count, r)
        d.set_rfc(False)
        _, r = d.decode(bytes([0xE0]) + bytes(60))  # reference mode again: a 2.5 ms TOC decodes as 20 ms (Q6)
        assert r == 960
