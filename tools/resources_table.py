"""profiles/rNN_kernel_resources.txt from the log of a resource build:

    MCRAT_RESOURCE_LOG=/tmp/res.log python -m mcrat_amd.build --force
    python tools/resources_table.py /tmp/res.log "round 4's final build" > profiles/r04_kernel_resources.txt

One line per kernel (every template instantiation): registers, scratch bytes per lane, occupancy, LDS, spilled registers."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
what = sys.argv[2] if len(sys.argv) > 2 else "this build"
blocks = re.split(r"remark: Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print("kernel resources per instantiation (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the product build's flags, %s); occupancy in waves per SIMD" % what)
print("rank_loop_kernel<DIMS, GEOM, STOKES, RESIDENT, THREADS, FUSE, CSH, QUEUE>: GEOM 0 Cartesian, 1 spherical, 2 cylindrical, 3 polar; tau_direct_* / tau_table_*: TAU_CALCULATION")
print("%-118s %5s %5s %8s %4s %7s %6s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS", "sgprSp", "vgprSp"))
seen = set()
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    d = re.sub(r"^void ", "", d)
    # the argument list goes, the template arguments stay
    depth, cut = 0, len(d)
    for i, ch in enumerate(d):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    d = d[:cut]
    if "<" in d:
        d = re.sub(r"^mcrat::", "", d)
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    if d in seen:
        continue
    seen.add(d)
    print("%-118s %5s %5s %8s %4s %7s %6s %6s" % (d[:118], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                                               g(r"LDS Size \[bytes/block\]"), g("SGPRs Spill"), g("VGPRs Spill")))
