"""The dispatches of one kernel from a `rocprofv3 --kernel-trace` directory, one line each: start, duration, workgroups -- so that the ONE launch
that holds a bench's timed frames (the frame queue) can be read off beside the warm-up and the one-frame launches of the same run.
usage: queue_dispatches.py <trace dir> <kernel name substring> <out.csv>"""
import csv
import glob
import os
import sys

src, want, out = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r.get("Workgroup_Size", 1))
            grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), grid // max(1, wg), r["Kernel_Name"].split("(")[0]))
rows.sort()
t0 = rows[0][0] if rows else 0
with open(out, "w") as f:
    f.write("start_us,duration_us,workgroups,kernel\n")
    for a, b, g, k in rows:
        f.write("%.1f,%.1f,%d,%s\n" % ((a - t0) / 1e3, (b - a) / 1e3, g, k))
print("%d dispatches of %s; the longest: %.1f us with %d workgroups" % (len(rows), want, max((b - a) / 1e3 for a, b, g, k in rows) if rows else 0,
                                                                      max(rows, key=lambda r: r[1] - r[0])[2] if rows else 0))
