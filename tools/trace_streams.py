#!/usr/bin/env python3
"""How much do the batches of bench.py's c4_stream section overlap on the device?  Input: a rocprofv3 --kernel-trace CSV of
`bench.py --sessions 1024 --steps 1 --warmup 0 --no-cpu-baseline --only c4_stream`.  The section's window = the last `win_ms`
of GPU activity (default 1200).  Per queue: launches, busy time; union busy time, pairwise overlap, and a coarse timeline
(10 ms buckets: which queues had a kernel running).  Usage: trace_streams.py <kernel_trace.csv> [win_ms]"""
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1200e6
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(r["e"] for r in rows)
    seg = [r for r in rows if r["s"] >= end - win]
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
    skey = "Stream_Id" if "Stream_Id" in rows[0] else None
    key = lambda r: (r.get(qkey), r.get(skey), r.get("Thread_Id"))
    groups = {}
    for r in seg:
        groups.setdefault(key(r), []).append(r)
    t0 = min(r["s"] for r in seg)

    def union(iv):
        iv = sorted(iv)
        tot, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        return tot + (cur_e - cur_s if cur_e is not None else 0)
    out = {"columns": list(rows[0].keys()), "window_ms": (end - t0) / 1e6, "queues": []}
    allv = []
    for k, rs in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        iv = [(r["s"], r["e"]) for r in rs]
        allv += iv
        out["queues"].append({"queue/stream/thread": k, "launches": len(rs), "busy_ms": union(iv) / 1e6,
                              "first_ms": (min(s for s, _ in iv) - t0) / 1e6, "last_ms": (max(e for _, e in iv) - t0) / 1e6})
    out["union_busy_ms"] = union(allv) / 1e6
    out["sum_busy_ms"] = sum(q["busy_ms"] for q in out["queues"])
    out["overlapped_ms"] = out["sum_busy_ms"] - out["union_busy_ms"]
    out["idle_ms"] = out["window_ms"] - out["union_busy_ms"]
    # coarse timeline
    nb = int((end - t0) / 10e6) + 1
    big = sorted(groups.items(), key=lambda kv: -len(kv[1]))[:6]
    lanes = []
    for k, rs in big:
        lane = [0.0] * nb
        for r in rs:
            b0, b1 = int((r["s"] - t0) / 10e6), int((r["e"] - t0) / 10e6)
            for b in range(b0, b1 + 1):
                lo, hi = max(r["s"], t0 + b * 10e6), min(r["e"], t0 + (b + 1) * 10e6)
                lane[b] += max(0, hi - lo) / 10e6
        lanes.append("".join(" .:-=+*#%@"[min(9, int(x * 9.99))] for x in lane))
    out["timeline_10ms_buckets"] = lanes
    # the longest kernels of the window
    top = sorted(seg, key=lambda r: r["s"] - r["e"])[:12]
    out["longest"] = [{"ms": (r["e"] - r["s"]) / 1e6, "at_ms": (r["s"] - t0) / 1e6, "q": key(r)[0], "kernel": r["Kernel_Name"][:70]} for r in top]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
