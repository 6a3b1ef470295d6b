"""BASELINE config 2's public-key encryption (r^N mod N^2: the 'Paillier-2048 modexp' of the metric) alone, with the launch records of
the profiler: batch sizes x number of keys, so that a slow launch can be told from a slow box.  One JSON line per case."""
import argparse
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
from multi_party_ecdsa_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="65536x16,65536x2,32768x16,65536x1")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    ctx = E.Context(0)
    dev = ctx.device
    keys = F.load_keys()
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    for case in a.cases.split(","):
        B, nk = (int(x) for x in case.split("x"))
        ks = keys[:nk]
        pk = E.PaillierKeys(ctx, N=[k.N for k in ks])
        m = bench.rand_words(g, dev, B, 64, 8)
        rr = bench.rand_words(g, dev, B, 64, 63)
        idx = (torch.arange(B, device=dev, dtype=torch.int32) % nk).contiguous()
        c = torch.empty((B, 128), dtype=torch.int32, device=dev)
        pk.encrypt_device(m, rr, idx, c)
        torch.cuda.synchronize()
        ctx.prof_enable(True)
        for _ in range(a.reps):
            pk.encrypt_device(m, rr, idx, c)
        torch.cuda.synchronize()
        recs = ctx.prof_collect()
        ctx.prof_enable(False)
        heavy = [r for r in recs if r["kind"] in (0, 3, 6) and r["bits"] == 4096]
        ms = [round(r["ms"], 3) for r in heavy]
        kern = float(np.mean(ms)) * 1e-3
        sl = float(np.mean([bench.slid(r) for r in heavy]))
        print(json.dumps({"case": case, "grid": os.environ.get("MPE_GRID", "default"), "no_sliding": bool(os.environ.get("MPE_NO_SLIDING")),
                          "launch_ms": ms, "kinds": sorted({r["kind"] for r in heavy}), "modexp4096_2048_per_s": round(B / kern, 1),
                          "executed_frac": round(B * bench.pair_modexp_macs(64, 64, sliding=sl) / kern / bench.PEAK_MAC_PER_S, 4),
                          "sliding_share": sl, "launch": ctx.launch_info()}), flush=True)
        pk.close() if hasattr(pk, "close") else None


if __name__ == "__main__":
    main()
