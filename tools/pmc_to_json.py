"""Turn the output of tools/profile_round.sh (merged back under gpurun_out/round) into the files committed under
profiles/:  <tag>_bench.json, <tag>_kernel_stats.csv (default command), <tag>_list_kernel_stats.csv,
<tag>_rank_loop_kernel_pmc.json, <tag>_step_kernel_pmc.json, <tag>_pmc_summary.txt.
usage: pmc_to_json.py gpurun_out/round profiles r01"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def newest(files):
    """gpurun merges successive runs into the same local directory (file names carry the pid): keep the latest per directory"""
    by_dir = {}
    for f in files:
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in newest(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void mcrat::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def stats_file(sub):
    f = newest(glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True))
    return f[0] if f else None


lines = []
out = {}
for sub, want, bytes_note in (("pmc_ranks", "rank_loop_kernel", "ranks"), ("pmc_list", "step_kernel", "list")):
    acc = counters(sub)
    for k in sorted(acc):
        lines.append("[%s] %s" % (sub, k))
        for c, v in sorted(acc[k].items()):
            lines.append("    %-24s n=%4d mean=%.6g min=%.6g max=%.6g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
    ks = [k for k in acc if want in k and "FETCH_SIZE" in acc[k]]
    if not ks:
        continue
    # the variant that does the work: the one with the largest mean FETCH_SIZE (step_kernel<.., true> is the forced pass)
    k = max(ks, key=lambda q: sum(acc[q]["FETCH_SIZE"]) / len(acc[q]["FETCH_SIZE"]) * len(acc[q]["FETCH_SIZE"]))
    fetch_kb = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
    write_kb = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
    # profiles/r04_fetch_calibration.txt (tools/fetch_calibration.hip): on gfx950 FETCH_SIZE x 1024 is half the bytes of the 128-B lines a kernel
    # touches, for 8-B and 16-B coalesced streams and for 16-B and 128-B gathers alike -- one factor for every kernel
    rd = 2 * fetch_kb * 1024
    corr = ("gfx950: FETCH_SIZE tallies the 128-B fabric requests at 64 B; read bytes = 2 x FETCH_SIZE x 1024 for every access width of this kernel "
            "(calibrated against known byte counts: profiles/r04_fetch_calibration.txt); WRITE_SIZE as counted")
    wr = write_kb * 1024
    out[want] = {"kernel": k, "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb, "launches_counted": len(acc[k]["FETCH_SIZE"]),
                 "correction": corr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
                 "command": "tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes, --kernel-trace)"}
    if want == "step_kernel":
        out[want]["algorithmic_bytes_per_launch"] = 110000000
    json.dump(out[want], open(os.path.join(dst, "%s_%s_pmc.json" % (tag, want)), "w"), indent=1)
open(os.path.join(dst, "%s_pmc_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
for sub, name in (("kt_default", "%s_kernel_stats.csv" % tag), ("kt_oneframe", "%s_kernel_stats_one_frame_per_launch.csv" % tag), ("kt_list", "%s_list_kernel_stats.csv" % tag),
                  ("kt_ingest", "%s_ingest_kernel_stats.csv" % tag), ("kt_cfg3", "%s_cfg3_kernel_stats.csv" % tag),
                  ("kt_cfg5", "%s_cfg5_kernel_stats.csv" % tag), ("kt_pools3", "%s_kernel_stats_pools3.csv" % tag)):
    f = stats_file(sub)
    if f:
        shutil.copy(f, os.path.join(dst, name))
t = os.path.join(src, "ingest_timings.txt")
if os.path.exists(t) and os.path.getsize(t) > 0:
    open(os.path.join(dst, "%s_ingest_timings.txt" % tag), "w").write(
        "".join(ln for ln in open(t) if not ln.startswith(("W2", "E2", "I2")) and "amdgpu.ids" not in ln))
b = os.path.join(src, "bench.json")
if os.path.exists(b) and os.path.getsize(b) > 0:
    shutil.copy(b, os.path.join(dst, "%s_bench.json" % tag))
u = os.path.join(src, "pools3_union.json")
if os.path.exists(u):
    shutil.copy(u, os.path.join(dst, "%s_pools3_union.json" % tag))
u = os.path.join(src, "queue_dispatches.csv")
if os.path.exists(u):
    shutil.copy(u, os.path.join(dst, "%s_queue_dispatches.csv" % tag))
for name in ("bench_cfg3", "bench_cfg5", "kt_cfg5", "kt_default", "kt_pools3", "kt_oneframe"):
    b = os.path.join(src, name + ".json")
    if os.path.exists(b) and os.path.getsize(b) > 0:
        shutil.copy(b, os.path.join(dst, "%s_%s.json" % (tag, name.replace("kt_cfg5", "bench_cfg5_traced").replace("kt_default", "bench_traced").replace("kt_pools3", "bench_traced_pools3").replace("kt_oneframe", "bench_traced_one_frame_per_launch"))))
print(json.dumps({k: v["traffic_bytes_per_launch"] for k, v in out.items()}))
