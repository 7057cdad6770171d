#!/usr/bin/env python3
"""usage: tools/kernel_meta.py <lib.so> [kernel name part ...] -- VGPRs, scratch bytes and static LDS bytes of the kernels in a built
library's code objects (what tests/test_kernel_budget.py asserts on, for any build: build_ab/lib_<variant>.so)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_kernel_budget as t
t.LIB = os.path.abspath(sys.argv[1])
for name, v in sorted(t._kernel_metadata().items()):
    if len(sys.argv) < 3 or any(k in name for k in sys.argv[2:]):
        print("%-28s vgpr %3d  scratch %4d  lds %6d" % (name.split("k_")[-1][:28] if "k_" in name else name[:28], *v))
