#!/usr/bin/env python3
"""Per-wave trace of the ladder kernel (profiling build: `tools/ab.sh wavetrace -DMPE_WAVE_TRACE`, loaded through MPE_LIB_PATH).

Every (wave, trip) of pair_modexp_kernel leaves {grid, block, batch, trip, limbs per lane, HW_ID, XCC_ID, s_memtime begin / end,
s_memrealtime begin / end}.  This tool runs the launches round 5 could not explain — a FRESH launch of <= 1 024 ladder waves against a
padded one and against the tail of a full grid — and one lone 1 024-session GG20 batch, and condenses the records per launch:

  * placement: how many SIMDs hold 1 / 2 / 3+ of the launch's waves (HW_ID: SIMD bits 5:4, CU 11:8, SH 12, SE 15:13; XCC_ID 3:0);
  * duration of a trip by class (the wave alone on its SIMD / sharing it), in ms of the 100 MHz counter;
  * the clock a wave saw: d(s_memtime) / d(s_memrealtime) x 100 MHz (if s_memtime counts shader cycles) — printed raw.

    MPE_LIB_PATH=$PWD/tools/ab/wavetrace.so python tools/trace_waves.py > gpurun_out/wave_trace.jsonl
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import fixtures as F  # noqa: E402
from multi_party_ecdsa_amd import _native as N  # noqa: E402
from multi_party_ecdsa_amd import engine as E  # noqa: E402

CAP = 1 << 18


class Trace:
    def __init__(self, dev):
        self.bufs = {}
        for bits in (2048, 1024):
            arm = getattr(N.lib, f"mpe_wave_trace_arm_{bits}", None)
            if arm is None:
                raise SystemExit("this library has no wave trace: build it with tools/ab.sh wavetrace -DMPE_WAVE_TRACE and set MPE_LIB_PATH")
            arm.argtypes = [C.c_void_p, C.c_uint]
            cnt = getattr(N.lib, f"mpe_wave_trace_count_{bits}")
            cnt.argtypes = [C.POINTER(C.c_uint)]
            self.bufs[bits] = (torch.zeros((CAP, 8), dtype=torch.int64, device=dev), arm, cnt)
        self.arm()

    def arm(self):
        torch.cuda.synchronize()
        for bits, (buf, arm, _) in self.bufs.items():
            assert arm(C.c_void_p(buf.data_ptr()), CAP) == 0

    def take(self):
        torch.cuda.synchronize()
        out = []
        for bits, (buf, _, cnt) in self.bufs.items():
            n = C.c_uint(0)
            assert cnt(C.byref(n)) == 0
            k = min(n.value, CAP)
            if k:
                out.append(buf[:k].cpu().numpy().view(np.uint64))
        self.arm()
        return np.concatenate(out) if out else np.zeros((0, 8), dtype=np.uint64)


def condense(recs, label, extra=None):
    """one JSON line per launch (records clustered by (grid, batch, limbs, bits, exp_words) and by time)"""
    lines = []
    if not len(recs):
        return lines
    key = np.stack([recs[:, 0] >> np.uint64(32), recs[:, 1] >> np.uint64(32), (recs[:, 1] >> np.uint64(8)) & np.uint64(0xFF),
                    recs[:, 7] >> np.uint64(32), recs[:, 7] & np.uint64(0xFFFFFFFF), recs[:, 1] & np.uint64(3)], axis=1)
    order = np.argsort(recs[:, 5], kind="stable")
    recs, key = recs[order], key[order]
    # a launch = a maximal run of records with the same key whose trips overlap in time
    launches = []
    for i in range(len(recs)):
        k = tuple(int(x) for x in key[i])
        placed = False
        for L in reversed(launches[-8:]):
            if L["key"] == k and int(recs[i, 5]) <= L["end"] :
                L["rows"].append(i); L["end"] = max(L["end"], int(recs[i, 6])); placed = True
                break
        if not placed:
            launches.append({"key": k, "rows": [i], "end": int(recs[i, 6])})
    for L in launches:
        r = recs[L["rows"]]
        grid, batch, limbs, bits, exp_words, flags = L["key"]
        hw = (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        xcc = ((r[:, 2] >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
        simd = (hw >> 4) & 3
        cu = (hw >> 8) & 15
        sh = (hw >> 12) & 1
        se = (hw >> 13) & 7
        trip = ((r[:, 1] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
        simd_key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
        dur = (r[:, 6] - r[:, 5]).astype(np.float64) / 1e5         # ms of the 100 MHz counter
        dm = (r[:, 4] - r[:, 3]).astype(np.float64)
        ratio = dm / np.maximum(1.0, (r[:, 6] - r[:, 5]).astype(np.float64))
        t0 = int(r[:, 5].min())
        out = {"label": label, "grid": grid, "batch": batch, "limbs_per_lane": limbs, "bits": bits, "exp_words": exp_words, "half": (flags >> 1) & 1,
               "slide": flags & 1, "wave_trips": int(len(r)), "launch_ms": round((int(r[:, 6].max()) - t0) / 1e5, 3),
               "memtime_per_realtime_tick": [round(float(np.percentile(ratio, q)), 3) for q in (5, 50, 95)]}
        for tp in sorted(set(trip.tolist())):
            sel = trip == tp
            sk = simd_key[sel]
            uniq, cnt = np.unique(sk, return_counts=True)
            per = dict(zip(uniq.tolist(), cnt.tolist()))
            co = np.array([per[k] for k in sk.tolist()])
            d = dur[sel]
            ent = {"waves": int(sel.sum()), "simds_used": int(len(uniq)), "simds_with_1": int((cnt == 1).sum()), "simds_with_2": int((cnt == 2).sum()),
                   "simds_with_3plus": int((cnt >= 3).sum()), "cus_used": int(len(np.unique(sk // 4))), "xccs_used": int(len(np.unique(xcc[sel]))),
                   "start_spread_ms": round(float((r[sel, 5].max() - r[sel, 5].min())) / 1e5, 3)}
            for c in (1, 2, 3):
                m = co == c if c < 3 else co >= 3
                if m.any():
                    ent[f"trip_ms_waves_{c}_per_simd"] = {"n": int(m.sum()), "p5": round(float(np.percentile(d[m], 5)), 3), "p50": round(float(np.median(d[m])), 3),
                                                           "p95": round(float(np.percentile(d[m], 95)), 3), "max": round(float(d[m].max()), 3)}
            out[f"trip{tp}"] = ent
        if extra:
            out.update(extra)
        lines.append(out)
    return lines


def emit(lines):
    for ln in lines:
        print(json.dumps(ln), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="16384x2,12288x2,8192x2,4096x2,49152x2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sessions", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--raw", default="", help="also save the raw records (npz) under this path")
    a = ap.parse_args()
    ctx = E.Context(0)
    dev = ctx.device
    tr = Trace(dev)
    keys = F.load_keys()
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    raw = {}
    env = {k: v for k, v in os.environ.items() if k.startswith("MPE_") and k != "MPE_LIB_PATH"}
    for case in [c for c in a.cases.split(",") if c]:
        B, nk = (int(x) for x in case.split("x"))
        pk = E.PaillierKeys(ctx, N=[k.N for k in keys[:nk]])
        m = bench.rand_words(g, dev, B, 64, 8)
        rr = bench.rand_words(g, dev, B, 64, 63)
        idx = (torch.arange(B, device=dev, dtype=torch.int32) % nk).contiguous()
        c = torch.empty((B, 128), dtype=torch.int32, device=dev)
        pk.encrypt_device(m, rr, idx, c)                      # warm-up (tables allocated), not traced
        tr.take()
        import time
        time.sleep(0.5)                                        # "after an idle moment": the first repetition of round 5's file
        for rep in range(a.reps):
            pk.encrypt_device(m, rr, idx, c)
        recs = tr.take()
        raw[f"c2_{case}"] = recs
        emit(condense(recs, f"public r^N, {case}, {a.reps} back-to-back launches after 0.5 s idle", {"env": env}))
    if a.sessions:
        import hashlib
        import gg20_fixture as G
        T, NP, SIGNERS = 1, 3, [0, 1]
        lk = G.make_local_keys(keys, T, NP, SIGNERS)
        gk = E.Gg20Keys(ctx, T, NP, SIGNERS, lk["arrays"])
        seed = hashlib.sha256(b"trace_waves").digest()
        B = a.sessions
        msg = bench.rand_words(g, dev, B, 8, 8)
        nonces, _ = E.gg20_sample_nonces(ctx, gk, B, seed, 0, msg=msg)
        E.gg20_sign(ctx, gk, nonces, B)
        tr.take()
        for step in range(a.steps):
            E.gg20_sample_nonces(ctx, gk, B, seed, step + 1, out=nonces)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = E.gg20_sign(ctx, gk, nonces, B)
            e1.record()
            recs = tr.take()
            raw[f"gg20_{B}_step{step}"] = recs
            ok = bool((out[3] == 0).all().item())
            emit(condense(recs, f"gg20 t=1 n=3, {B} sessions, step {step}", {"env": env, "step_ms": round(e0.elapsed_time(e1), 2), "all_signed": ok}))
    if a.raw:
        np.savez_compressed(a.raw, **raw)


if __name__ == "__main__":
    main()
