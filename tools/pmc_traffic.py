#!/usr/bin/env python3
"""Condenses the three rocprofv3 --pmc passes of tools/profile_round.sh (pmc_fetch.json, pmc_write.json, pmc_sq.json) into the
traffic / issue summary bench.py's roofline line refers to (profiles/<round>/pmc_traffic.json).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads -> x2;
WRITE_SIZE as reported; both counters are in KB (1024 B).
Usage: pmc_traffic.py <dir with pmc_*.json> <sessions> > pmc_traffic.json"""
import json
import os
import sys


def main():
    d, sessions = sys.argv[1], int(sys.argv[2])
    load = lambda n: {k["kernel"]: k for k in json.load(open(os.path.join(d, n)))["kernels"]}
    fe, wr, sq = load("pmc_fetch.json"), load("pmc_write.json"), load("pmc_sq.json")
    import subprocess
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:                                    # noqa: BLE001 — the GPU box has no .git: the caller passes it in MPE_COMMIT
        commit = os.environ.get("MPE_COMMIT")
    out = {"sessions": sessions, "host": os.uname().nodename, "commit": commit,
           "command": "tools/profile_round.sh (bench.py --no-cpu-baseline --no-configs --steps 1 --warmup 0 under rocprofv3 --kernel-trace "
                      "--pmc <counters>, one pass per counter group)",
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; "
                         "WRITE_SIZE uncorrected; KB = 1024 B", "kernels": {}}
    for k in fe:
        if k not in wr or k not in sq:
            continue
        n = fe[k]["launches"]
        avg_ms = sq[k]["total_ms"] / sq[k]["launches"]
        f_kb, w_kb = fe[k]["counters"]["FETCH_SIZE"]["per_launch"], wr[k]["counters"]["WRITE_SIZE"]["per_launch"]
        hbm = (2 * f_kb + w_kb) * 1024
        c = {name: v["per_launch"] for name, v in sq[k]["counters"].items()}
        clock = c["GRBM_GUI_ACTIVE"] / 8 / (avg_ms * 1e-3) / 1e9
        ipc = c["SQ_INSTS_VALU"] / (1024 * clock * 1e9 * avg_ms * 1e-3)             # wave instructions per SIMD cycle (256 CU x 4 SIMD)
        out["kernels"][k] = {"launches": n, "avg_ms": avg_ms, "FETCH_SIZE_KB_per_launch_raw": f_kb, "WRITE_SIZE_KB_per_launch_raw": w_kb,
                             "hbm_bytes_per_launch": hbm, "hbm_GB_per_s": hbm / (avg_ms * 1e-3) / 1e9, "sq_per_launch": c,
                             "effective_clock_GHz": clock, "note_clock": "GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / kernel seconds",
                             "valu_insts_per_simd_cycle": ipc, "valu_issue_busy": ipc * 4,
                             "note_busy": "a wave64 v_mad_u64_u32 occupies the 16-lane multiplier 4 cycles (profiles/r01_valu_rate.json)"}
    dom = next((k for k in out["kernels"] if "pair_modexp_kernel<mpe::Cfg<2048, 29, 18, 4>" in k), None)
    if dom:
        out["hbm_bytes_per_launch"] = out["kernels"][dom]["hbm_bytes_per_launch"]
        out["dominant_kernel"] = dom
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
