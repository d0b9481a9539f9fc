"""Static instruction mix per LOOP of one kernel from a `hipcc -S --cuda-device-only` dump: for every loop header the blocks the assembler's
comments attribute to it ("in Loop: Header=BBn_m"), with instruction counts by kind -- f64 arithmetic, other vector ALU, v_readlane / v_writelane
(scalar registers spilled to vector-register lanes), scalar ALU / branches, LDS, vector memory.
usage: isa_loops.py k.s mangled_prefix [header-label-to-list-block-by-block]"""
import re
import sys
from collections import Counter, OrderedDict


def kind(op):
    if "f64" in op:
        return "f64"
    if "readlane" in op or "writelane" in op:
        return "lane"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global", "flat", "buffer", "scratch")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    txt = open(sys.argv[1]).read()
    m = re.search(r"^(%s\w*):" % re.escape(sys.argv[2]), txt, re.M)
    body = txt[m.start():txt.find(".Lfunc_end", m.start())]
    blocks, cur = [], None
    for l in body.split("\n")[1:]:
        ls = l.strip()
        mm = re.match(r"^(\.LBB\d+_\d+):(.*)", ls) or re.match(r"^; (%bb\.\d+):(.*)", ls)
        if mm:
            cur = [mm.group(1), mm.group(2), []]
            blocks.append(cur)
            continue
        if cur is None or not ls or ls.startswith((";", ".", "//")) or ls.endswith(":"):
            continue
        cur[2].append(ls)
    loops = OrderedDict()
    for b in blocks:
        hm = re.search(r"Header=(BB\d+_\d+)", b[1])
        key = hm.group(1) if hm else (b[0][2:] if "Loop Header" in b[1] else None)
        if key is None:
            continue
        c = loops.setdefault(key, Counter())
        for ins in b[2]:
            c[kind(ins.split()[0])] += 1
            c["all"] += 1
        c["blocks"] += 1
    for k, c in loops.items():
        if c["all"] >= 60:
            print("%-12s %5d instr in %3d blocks  %s" % (k, c["all"], c["blocks"], {x: c[x] for x in ("f64", "valu", "lane", "salu", "lds", "vmem") if c[x]}))
    if len(sys.argv) > 3:
        for b in blocks:
            hm = re.search(r"Header=(BB\d+_\d+)", b[1])
            if (hm and hm.group(1) == sys.argv[3]) or b[0] == ".L" + sys.argv[3]:
                c = Counter(kind(i.split()[0]) for i in b[2])
                br = [i.split()[-1] for i in b[2] if i.startswith(("s_cbranch", "s_branch"))]
                print("  %-11s %4d %-58s -> %s" % (b[0], len(b[2]), dict(c), " ".join(br)))


if __name__ == "__main__":
    main()
