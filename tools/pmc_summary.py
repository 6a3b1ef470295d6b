#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass: per kernel, the number of launches, their total duration and every counter's
total / per-launch value.  Usage: pmc_summary.py <counter_collection.csv> [<kernel_trace.csv>]  -> JSON on stdout."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if not rows:
        print(json.dumps({"error": "empty counter file"}))
        return
    kn = "Kernel_Name" if "Kernel_Name" in rows[0] else "Kernel-Name"
    disp = defaultdict(dict)       # kernel -> dispatch id -> (start, end)
    ctr = defaultdict(lambda: defaultdict(float))
    for r in rows:
        k = short(r[kn])
        d = r.get("Dispatch_Id") or r.get("Dispatch-Id")
        try:
            disp[k][d] = (int(r.get("Start_Timestamp", 0) or 0), int(r.get("End_Timestamp", 0) or 0))
        except ValueError:
            disp[k][d] = (0, 0)
        ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = {k: sum(e - s for s, e in v.values()) / 1e6 for k, v in disp.items()}
    if len(sys.argv) > 2 and not any(dur.values()):
        # durations from the kernel trace of the same run
        for r in csv.DictReader(open(sys.argv[2])):
            k = short(r["Kernel_Name"])
            dur[k] = dur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = []
    for k in sorted(dur, key=lambda k: -dur[k])[:8]:
        n = len(disp.get(k, {})) or 1
        out.append({"kernel": k, "launches": n, "total_ms": dur[k],
                    "counters": {c: {"total": v, "per_launch": v / n} for c, v in ctr.get(k, {}).items()}})
    print(json.dumps({"kernels": out}, indent=1))


if __name__ == "__main__":
    main()
