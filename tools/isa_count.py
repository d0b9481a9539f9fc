"""Instruction mix of selected kernels from a `hipcc -S --cuda-device-only` dump.
usage: isa_count.py k.s substring [substring...]"""
import re
import subprocess
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
pat = re.compile(r"^(_Z[A-Za-z0-9_]+):\s*(?:;.*)?$", re.M)
marks = [(m.start(), m.group(1)) for m in pat.finditer(txt)]
for k, (pos, name) in enumerate(marks):
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0]
    if not any(w in dn for w in sys.argv[2:]):
        continue
    end = txt.find("s_endpgm", pos)
    nxt = marks[k + 1][0] if k + 1 < len(marks) else len(txt)
    body = txt[pos:nxt]
    lines = [l.strip() for l in body.split("\n")[1:]]
    lines = [l for l in lines if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
    c = Counter(l.split()[0] for l in lines)
    fam = Counter()
    for op, v in c.items():
        if op.startswith("v_") and "f64" in op:
            fam["v_f64"] += v
        elif op.startswith("v_"):
            fam["v_other"] += v
        elif op.startswith("s_"):
            fam["salu"] += v
        elif op.startswith(("global_", "flat_", "scratch_", "buffer_")):
            fam["vmem"] += v
        elif op.startswith("ds_"):
            fam["lds"] += v
        else:
            fam["other"] += v
    print(dn.strip(), "instructions:", len(lines), dict(fam))
    print("   top:", c.most_common(12))
