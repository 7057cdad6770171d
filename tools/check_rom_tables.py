#!/usr/bin/env python3
"""Container-only check: every table in the generated rom_tables.h that has a counterpart in the reference
sources is value-for-value identical to the reference's literal (read as TEXT from /root/reference/src).
Run by tests/test_rom_tables.py when /root/reference exists; skipped on the GPU box."""
import os
import re
import sys

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

PAIRS = {  # ours -> reference name
    "rom_band_alloc": "band_allocation", "rom_eband": "eband5ms", "rom_logn": "logN400",
    "rom_pulse_idx": "cache_index50", "rom_pulse_bits": "cache_bits50", "rom_pulse_caps": "cache_caps50",
    "rom_log2_frac": "LOG2_FRAC_TABLE", "rom_emeans": "eMeans", "rom_eprob": "e_prob_model",
    "rom_mdct_trig": "mdct_twiddles960", "rom_fft_tw": "fft_twiddles48000_960", "rom_win120": "window120",
    "rom_bitrev480": "fft_bitrev480", "rom_bitrev240": "fft_bitrev240", "rom_bitrev120": "fft_bitrev120",
    "rom_bitrev60": "fft_bitrev60",
}


def parse_arrays(text, macros=None):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    pat = r"\b(?:static\s+)?(?:OPUS_ROM|const)\s+(?:unsigned\s+char|signed\s+char|[A-Za-z_0-9]+)\s+([A-Za-z_0-9]+)\s*((?:\[[^\]]*\])+)\s*(?:PROGMEM)?\s*=\s*\{"
    for m in re.finditer(pat, text):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"{": 1, "}": -1}.get(text[j], 0)
            j += 1
        items = [t.strip() for t in text[i:j - 1].replace("{", " ").replace("}", " ").split(",") if t.strip()]
        try:
            vals = []
            for t in items:
                for k, v in (macros or {}).items():
                    t = re.sub(r"\b%s\b" % k, str(v), t)
                vals.append(int(eval(re.sub(r"(\d)[uUlL]+\b", r"\1", t), {}, {})))
            out[m.group(1)] = vals
        except Exception:
            pass
    return out


def main():
    if not os.path.isdir(REF):
        print("reference not present: nothing to check")
        return 0
    ours = parse_arrays(open(os.path.join(os.path.dirname(HERE), "oracle", "rom_tables.h")).read())
    macros = {"OFFSET_VL_Q10": 32, "OFFSET_VH_Q10": 100, "OFFSET_UVL_Q10": 100, "OFFSET_UVH_Q10": 240}
    ref = parse_arrays(open(os.path.join(REF, "celt.cpp")).read())
    ref.update(parse_arrays(open(os.path.join(REF, "silk.cpp")).read(), macros))
    from silk_rom_data import SILK_TABLES
    bad = 0
    # SILK tables keep the reference's order of appearance: match by content
    ref_by_content = {tuple(v): k for k, v in ref.items()}
    for name in SILK_TABLES:
        if tuple(ours[name]) not in ref_by_content:
            print("MISMATCH (no reference table with these values):", name)
            bad += 1
    for mine, theirs in PAIRS.items():
        if ours[mine] != ref[theirs]:
            print("MISMATCH", mine, theirs)
            bad += 1
    # PVQ U(n,k): reference stores ragged rows (celt.cpp:75-183, row offsets :651)
    rows, data, U = ref["row_idx"], ref["CELT_PVQ_U_DATA"], ours["rom_pvq_u"]
    for n in range(15):
        end = (rows[n + 1] + n + 1) if n < 14 else len(data)
        k = n
        while rows[n] + k < end:
            if U[n * 177 + k] != data[rows[n] + k]:
                print("MISMATCH pvq_u", n, k)
                bad += 1
                break
            k += 1
    print("rom tables checked:", len(PAIRS) + len(SILK_TABLES) + 1, "mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
