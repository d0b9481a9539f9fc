"""Busy time of overlapping kernel launches from a rocprofv3 --kernel-trace CSV (run on the GPU box, before the per-dispatch trace is deleted).

With several rank pools on their own HIP streams (bench.py --pools 3, the headline's launch shape) the rank_loop_kernel launches of one hydro frame
overlap: none of them has "the" duration of the step.  What can be read off the trace is the UNION of their intervals -- the time during which at least
one of them was running -- per timed step, next to their count and individual durations.
    python tools/union_busy.py <dir with *kernel_trace.csv> <kernel name substring> <timed steps> <warm-up steps> <launches per step> [out.json]"""
import csv
import glob
import json
import os
import sys

root, want, steps, warm, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
iv = []
for f in files:
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
iv.sort()
# the timed region's launches are the last steps * per_step of the main measurement; bench.py runs other sections afterwards (roofline pass, extras), so take the
# launches [warm * per_step, (warm + steps) * per_step) in start order
sel = iv[warm * per_step:(warm + steps) * per_step]
union = 0
cur_s, cur_e = None, None
for s, e in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
dur = [e - s for s, e in sel]
span = (max(e for _, e in sel) - min(s for s, _ in sel)) if sel else 0
out = {"kernel": want, "launches_in_trace": len(iv), "launches_counted": len(sel), "steps": steps, "launches_per_step": per_step,
       "union_busy_ms_per_step": union / 1e6 / max(steps, 1), "span_ms_per_step": span / 1e6 / max(steps, 1),
       "mean_launch_ms": (sum(dur) / len(dur) / 1e6) if dur else None, "max_launch_ms": (max(dur) / 1e6) if dur else None,
       "sum_of_launches_ms_per_step": sum(dur) / 1e6 / max(steps, 1)}
print(json.dumps(out))
if len(sys.argv) > 6:
    json.dump(out, open(sys.argv[6], "w"), indent=1)
