"""Condenses a tools/profile.sh output directory into summary.txt (kernel
statistics + per-kernel counter totals for the dispatches of k_render)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel statistics (rocprofv3 --kernel-trace --stats) ==")
    with open(path) as fh:
        for row in csv.DictReader(fh):
            lines.append("  {Name:60.60s} calls={Calls:>4} total_ns={TotalDurationNs:>14} avg_ns={AverageNs:>14} pct={Percentage}".format(**row))
for sub in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(sub):
        continue
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"][:40]
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add(row["Dispatch_Id"])
    lines.append(f"== {os.path.basename(sub)} (sum over dispatches) ==")
    for k, cs in tot.items():
        if "k_render" not in k and "k_reduce" not in k:
            continue
        lines.append(f"  {k} dispatches={len(calls[k])}")
        for c, v in sorted(cs.items()):
            lines.append(f"      {c:28s} {v:.6g}   per dispatch {v / max(1, len(calls[k])):.6g}")
text = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(text + "\n")
print(text)
