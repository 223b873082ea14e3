#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel over all dispatches of each pass directory."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
passes = defaultdict(lambda: defaultdict(set))     # a counter collected in several passes is averaged over them
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    pass_name = os.path.relpath(f, root).split(os.sep)[0]
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            for short in ("k_closest_hit", "k_shade", "k_raygen", "k_accumulate", "k_resolve", "k_stats"):
                if short in k:
                    k = short + ("<counters>" if short == "k_closest_hit" and "k_closest_hit<true" in k else "")
                    break
            else:
                continue
            out[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]].add((pass_name, row["Dispatch_Id"]))
            passes[k][row["Counter_Name"]].add(pass_name)
res = {}
for k, d in out.items():
    res[k] = {c: v / len(passes[k][c]) for c, v in sorted(d.items())}
    res[k]["_dispatches"] = max(len(s) // len(passes[k][c]) for c, s in calls[k].items())
print(json.dumps(res, indent=1))
