#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV into a timeline of the LAST step of a bench run: per kernel launch its start
offset, duration and the idle gap before it; totals of busy / idle time.  Usage: trace_timeline.py <kernel_trace.csv> [r0]
(r0: name fragment of the kernel that starts a step, default "r0_kernel")."""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    mark = sys.argv[2] if len(sys.argv) > 2 else "r0_kernel"
    starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
    lo = starts[-1] if starts else 0
    # a step begins a few launches before r0 (memset, idx kernel); take everything from the launch that follows a >2 ms gap
    seg = rows[lo:]
    t0 = int(seg[0]["Start_Timestamp"])
    out, busy, prev_end = [], 0, t0
    # concurrent streams: busy time = union of the intervals
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(0, s - prev_end)
        if gap > 20e6:                      # the step is over: what follows is the teardown of the bench process
            break
        out.append({"t_ms": round((s - t0) / 1e6, 3), "dur_ms": round((e - s) / 1e6, 3), "gap_ms": round(gap / 1e6, 3), "kernel": short(r["Kernel_Name"]),
                    "queue": r.get("Queue_Id"), "stream": r.get("Stream_Id"), "grid": r.get("Grid_Size_X") or r.get("Grid_Size")})
        if e > prev_end:
            busy += e - max(s, prev_end)
            prev_end = e
    total = prev_end - t0
    agg = {}
    for o in out:
        a = agg.setdefault(o["kernel"], [0, 0.0])
        a[0] += 1; a[1] += o["dur_ms"]
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]
    print(json.dumps({"launches": len(out), "span_ms": total / 1e6, "busy_ms": busy / 1e6, "idle_frac": 1 - busy / total if total else 0,
                      "by_kernel": [{"kernel": k, "launches": v[0], "total_ms": round(v[1], 3)} for k, v in top],
                      "timeline": [o for o in out if o["dur_ms"] > 0.3 or o["gap_ms"] > 0.3]}, indent=1))


if __name__ == "__main__":
    main()
