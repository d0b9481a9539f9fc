"""Average rocprofv3 --pmc counters per kernel name from the counter_collection CSVs under a directory."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void mcrat::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "step_kernel" not in k and "event_kernel" not in k:
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print("    %-28s n=%4d  mean=%.6g  min=%.6g  max=%.6g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
