"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one line per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:64]
    print("%-64s V=%4s S=%4s scratch=%5s occ=%s lds=%s" % (
        dn, g("VGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
