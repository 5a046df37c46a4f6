#!/usr/bin/env python3
"""Mean per launch of every counter of a rocprofv3 --pmc pass, per kernel:
    python profiles/summarize_sq.py <tag> <config> <name> [kernel-substring]
reads gpurun_out/prof_<tag>_sq/<config>_<name>_counter_collection.csv, writes profiles/<tag>_<config>_<name>_summary.csv"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg, name = sys.argv[1:4]
want = sys.argv[4] if len(sys.argv) > 4 else "k_"
path = os.path.join(ROOT, "gpurun_out", "prof_%s_sq" % tag, "%s_%s_counter_collection.csv" % (cfg, name))
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if want in r["Kernel_Name"]:
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = os.path.join(ROOT, "profiles", "%s_%s_%s_summary.csv" % (tag, cfg, name))
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "launches", "mean_per_launch"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), "%.1f" % (sum(v) / len(v))])
        print("%-60s %-24s %6d %14.1f" % (k[:60], c, len(v), sum(v) / len(v)))
