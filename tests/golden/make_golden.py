#!/usr/bin/env python3
"""Regenerates tests/golden/*.json|npz.

Two kinds of fixture:
  survey_kats.json   -- outputs of the REFERENCE itself, recorded while it ran during the survey session
                        (KAT 1 and KAT 2).  Copied verbatim; this script only re-checks that
                        the oracle still reproduces them.  They are the parity pin of the oracle.
  oracle_vectors.*   -- vectors produced by the oracle (NOT by the reference) for regression and for checking the
                        GPU path on machines where the oracle library is not built: per-frame FNV-1a hashes of
                        48 streams x 6 frames per mode, and full PCM of 2 streams x 2 frames per mode.
Inputs are the LCG payloads of SURVEY.md section 8d (seed 0x9E3779B9 ^ stream id)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py  # noqa: E402
from conftest import load_pkg  # noqa: E402

MODES = {"celt_fb_stereo": (0xFC, 160), "silk_nb_stereo": (0x0C, 40), "hybrid_fb_stereo": (0x7C, 120)}


def main():
    o = oracle_py.load()
    pkg = load_pkg()
    hashes, pcm = {}, {}
    for name, (toc, L) in MODES.items():
        pay = pkg.lcg_payloads(48, 6, L)
        ref, ok = o.batch_decode(2, toc, pay)
        assert ok == 48 * 6
        hashes[name] = {"toc": toc, "payload_len": L, "streams": 48, "frames": 6,
                        "fnv1a_u16": [[oracle_py.fnv1a_u16(ref[s, f]) for f in range(6)] for s in range(48)]}
        pcm[name] = ref[:2, :2].copy()
    json.dump({"generator": "oracle (oracle/liboc_oracle.so), NOT the reference", "payloads": "lcg_payloads(seed 0x9E3779B9 ^ stream)",
               "modes": hashes}, open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors_pcm.npz"), **pcm)
    print("wrote oracle_vectors.json / oracle_vectors_pcm.npz")


if __name__ == "__main__":
    main()
