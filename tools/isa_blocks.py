#!/usr/bin/env python3
"""usage: tools/isa_blocks.py file.s function-substring  -- basic blocks of one function of a gfx950 assembly listing with their
instruction mix (vector / scalar / LDS / vector-memory / scalar-memory), so that a kernel's instruction budget can be read
without a GPU.  Cycle estimate: 2 for the plain two-operand VALU class measured at 2 cycles (profiles/r03/a_valu_issue_rates.txt), 4 otherwise."""
import re, sys
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32",
        "v_add_f32", "v_mul_f32"}
src, fn = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_].*:", l) and fn in l and not l.startswith("."))
blocks, cur = [], None
for l in lines[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end") or l.startswith("\t.section"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = dict(name=m.group(1), v=0, s=0, ds=0, vm=0, sm=0, cyc=0, br=[], first="")
        blocks.append(cur)
        continue
    if cur is None:
        cur = dict(name="entry", v=0, s=0, ds=0, vm=0, sm=0, cyc=0, br=[], first="")
        blocks.append(cur)
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."):
        continue
    op = t[0]
    if op.startswith("v_"):
        cur["v"] += 1
        sdwa_dpp = "_sdwa" in op or "_dpp" in op or " s" in l.split(";")[0] or "vcc" in l
        cur["cyc"] += 2 if (op.replace("_e32", "") in FAST and not sdwa_dpp) else 4
    elif op.startswith("ds_"):
        cur["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur["vm"] += 1
    elif op.startswith(("s_load", "s_buffer_load")):
        cur["sm"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["br"].append(t[1])
tot = dict(v=0, s=0, ds=0, vm=0, sm=0)
idx = {b["name"]: i for i, b in enumerate(blocks)}
for i, b in enumerate(blocks):
    back = [x for x in b["br"] if x in idx and idx[x] <= i]
    print("%-12s v %4d (~%5d cyc)  s %4d  ds %3d  vmem %3d  smem %2d  %s%s" % (b["name"], b["v"], b["cyc"], b["s"], b["ds"], b["vm"], b["sm"],
          "-> " + ",".join(b["br"]) if b["br"] else "", "   <== LOOP back to " + ",".join(back) if back else ""))
    for k in tot:
        tot[k] += b[k]
print("total (static):", tot)
