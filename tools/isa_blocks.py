"""Per-basic-block instruction mix of one kernel from a `hipcc -S --cuda-device-only` dump.
usage: isa_blocks.py k.s mangled_prefix [min_instructions]"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
m = re.search(r"^(%s\w*):" % re.escape(sys.argv[2]), txt, re.M)
start = m.start()
end = txt.find(".Lfunc_end", start)
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
blocks, cur, n, kinds = [], "entry", 0, Counter()
for l in txt[start:end].split("\n")[1:]:
    ls = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", ls):
        blocks.append((cur, n, kinds))
        cur, n, kinds = ls[:-1], 0, Counter()
        continue
    if not ls or ls.startswith((";", ".", "//")) or ls.endswith(":"):
        continue
    op = ls.split()[0]
    n += 1
    if "f64" in op:
        kinds["f64"] += 1
    elif op.startswith(("v_mul_hi", "v_mul_lo", "v_mad_u64", "v_mad_i64")):
        kinds["imul"] += 1
    elif op.startswith("v_"):
        kinds["valu"] += 1
    elif op.startswith("s_"):
        kinds["salu"] += 1
    elif op.startswith(("global", "flat", "buffer", "scratch")):
        kinds["vmem"] += 1
    elif op.startswith("ds_"):
        kinds["lds"] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        kinds["->" + ls.split()[-1]] += 1
blocks.append((cur, n, kinds))
tot = 0
for b in blocks:
    tot += b[1]
    if b[1] >= minn:
        print("%-12s %5d  %s" % (b[0], b[1], dict(b[2])))
print("total", tot, "in", len(blocks), "blocks")
