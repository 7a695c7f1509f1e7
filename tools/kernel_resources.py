#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel
(VGPR, AGPR, scratch bytes/lane, occupancy waves/SIMD, SGPR, LDS, VGPR spills).
usage: kernel_resources.py <remarks.txt> [name-filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b in blocks:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("SGPRs"),
                 g(r"LDS Size \[bytes/block\]"), g("VGPR Spill")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("kernel".ljust(84), "VGPR AGPR scratch occ SGPR LDS vspill")
for n, r in zip(names, rows):
    if flt in n:
        n = n.replace("void ", "")
        print(n[:84].ljust(84), *["%4d" % v for v in r[1:]])
