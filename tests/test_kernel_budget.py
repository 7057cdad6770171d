"""Register budgets of the shipped kernels, read from the code objects inside libopusgpu.so (no GPU needed).

A kernel's occupancy is decided at compile time -- waves per SIMD = 512 / its vector registers (in eights) -- and is easy to lose
without noticing: in round 4 a run-time `if` kept a call to the one-lane SILK synthesis alive in k_silk_synth, whose register
count is then the CALLEE's (147 instead of 90): three waves per SIMD instead of four, SILK-NB 1.05 -> 1.27 ms, every test green.
The bounds below are the occupancy steps the measurements in DESIGN.md were taken at, not the exact counts."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "esp32-opus-player_amd", "libopusgpu.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernel -> most vector registers it may use (the next occupancy step would be lost above it), scratch bytes allowed
BUDGET = {
    "k_silk_synth": (80, 0),       # six waves per SIMD by registers (its seven granules of LDS allow 17 workgroups per CU)
    "k_celt_recon_fb": (80, 0),    # six waves per SIMD (launch bound), five granules of LDS (6.2 KB, round 5); no scratch (round 4: its noise generator's table had been there)
    "k_silk_synth_nb": (80, 0),    # the synthesis of narrowband SILK-only frames: 75 registers (its allocation is raised to five waves per SIMD on purpose), five granules of LDS
    "k_silk_parse": (84, 0),       # round 5, without the parameter half: 82 registers, no spills (round 4: 128 and a few spills), six waves per SIMD
    "k_silk_parse64": (84, 0),     # the same with 64 frames per wave (pipelined steps, large batches)
    "k_silk_params": (128, 0),     # one (frame, channel) per lane: 100 registers, four waves per SIMD
    "k_celt_parse64": (168, 64),   # one wave per SIMD next to the reconstruction's: the fewer registers, the more of those fit (its LDS is dynamic)
    "k_celt_parse": (256, 64),
    "k_celt_post": (128, 0),
    "k_decode_rfc": (256, 1024),   # two waves per SIMD (launch bound; it spills)
}


def _kernel_metadata():
    have = all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))
    if not have or not os.path.exists(LIB):
        pytest.skip("LLVM tools or the library are not there")
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
        data = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        at = [m.start() for m in re.finditer(re.escape(magic), data)]
        for j, lo in enumerate(at):  # one bundle per translation unit
            part = os.path.join(d, f"b{j}.bin")
            open(part, "wb").write(data[lo:at[j + 1] if j + 1 < len(at) else len(data)])
            co = os.path.join(d, f"b{j}.co")
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                   f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk)
                if name and vg:
                    out[name.group(1)] = (int(vg.group(1)), int(priv.group(1)) if priv else 0, int(lds.group(1)) if lds else 0)
    return out


def test_kernels_keep_their_register_budgets():
    meta = _kernel_metadata()
    seen = {}
    for mangled, (vgpr, scratch, lds) in meta.items():
        for k in BUDGET:
            if re.search(r"\d+" + k + r"(P|E|v|$)", mangled) or mangled == k:
                seen[k] = (vgpr, scratch, lds)
    missing = [k for k in BUDGET if k not in seen]
    assert not missing, f"kernels not found in the library's code objects: {missing} (have {sorted(meta)[:6]}...)"
    over = {k: seen[k] for k in BUDGET if seen[k][0] > BUDGET[k][0] or seen[k][1] > BUDGET[k][1]}
    assert not over, f"over budget (vgpr, scratch bytes, lds bytes): {over}; budgets {({k: BUDGET[k] for k in over})}"
    # LDS steps the occupancy figures in DESIGN.md rest on (granules of 1,280 bytes per workgroup)
    assert seen["k_silk_synth"][2] <= 8960 and seen["k_silk_synth_nb"][2] <= 6400 and seen["k_celt_recon_fb"][2] <= 6400
    assert seen["k_silk_parse"][2] <= 8192 and seen["k_silk_params"][2] <= 12800  # (7.8 KB: the table blob and the pulse decoder's block rows, 64 columns)
    assert seen["k_celt_parse"][2] <= 11520  # the in-order parse kernel: nine granules (its 64-frame twin sizes its LDS at the launch)
