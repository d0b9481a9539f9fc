#!/bin/bash
# FETCH_SIZE against known byte counts for the loop's access widths (tools/fetch_calibration.hip): through gpurun; result in gpurun_out/fetchcal/summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fetchcal; rm -rf $out; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o /tmp/fetch_calibration || exit 1
/tmp/fetch_calibration > $out/known.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p1 -- /tmp/fetch_calibration > $out/p1.out 2> $out/p1.err; echo "pmc exit=$?"
python3 - <<'PY' | tee gpurun_out/fetchcal/summary.txt
import csv, glob, re
from collections import defaultdict
known = {}
for l in open("gpurun_out/fetchcal/known.txt"):
    m = re.match(r"known_bytes (\w+) requested=(\d+) lines=(\d+)", l)
    if m:
        known[m.group(1)] = (int(m.group(2)), int(m.group(3)))
acc = defaultdict(list)
for f in glob.glob("gpurun_out/fetchcal/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
print("FETCH_SIZE (rocprofv3, in KB as it reports it) against known bytes; factor = known bytes / (FETCH_SIZE x 1024)")
for k in ("stream8", "stream16", "gather16", "gather128"):
    v = acc.get(k, [])
    if not v:
        continue
    mean = sum(v) / len(v)
    req, lines = known[k]
    extra = (16 << 20) * 4 if k.startswith("gather") else 0
    print("%-10s launches=%d FETCH_SIZE=%.0f KB = %.4g B | requested %.4g B: factor %.3f | 128-B lines touched %.4g B: factor %.3f | (indices streamed beside: %.3g B, included in the known figures)"
          % (k, len(v), mean, mean * 1024, req + extra, (req + extra) / (mean * 1024), lines + extra, (lines + extra) / (mean * 1024), extra))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
