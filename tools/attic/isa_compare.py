#!/usr/bin/env python
"""Per-kernel instruction count / MFMA count / VGPRs / scratch of two `tools/isa_dump.sh` outputs, side by side.
    python tools/isa_compare.py before.s after.s"""
import re
import sys


def kernels(path):
    lines = open(path).read().split("\n")
    out, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = [0, 0, None, None]
            continue
        if cur is None:
            continue
        st = ln.strip()
        if st.startswith(".amdhsa_next_free_vgpr"):
            out[cur][2] = int(st.split()[1])
        elif st.startswith(".amdhsa_private_segment_fixed_size"):
            out[cur][3] = int(st.split()[1])
        elif st.startswith(".end_amdhsa_kernel"):
            cur = None
        elif ln.startswith("\t") and st and not st.startswith((".", ";")):
            out[cur][0] += 1
            out[cur][1] += st.startswith("v_mfma")
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
for k in sorted(set(a) | set(b)):
    x, y = a.get(k), b.get(k)
    print("%-72s %-26s -> %-26s %s" % (k[:72], x, y, "" if x == y else "<<< differs"))
