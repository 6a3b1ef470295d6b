#!/usr/bin/env python3
"""Experiment behind the pipelined engine: a stream of 1 024-session GG20 batches handled as `lanes` concurrent
super-batches of `group` batches each, every lane on ONE HIP stream of its own context, all lanes fed by ONE host thread
(mpe_gg20_sign only enqueues).  Prints one JSON line: signatures/s, ms per super-batch.
  python tools/exp_superbatch.py --lanes 2 --group 4 --reps 6          (MPE_NO_PAR=1 in the environment: no forked streams)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=6, help="super-batches per lane in the timed region")
    ap.add_argument("--share-keys", type=int, default=1)
    a = ap.parse_args()
    import torch
    import fixtures as F
    import gg20_fixture as G
    import bench as Bn
    from multi_party_ecdsa_amd import engine as E
    dev = torch.device("cuda", 0)
    t, n, signers = 1, 3, [0, 1]
    S = 2
    lk = G.make_local_keys(F.load_keys(), t, n, signers)
    B = a.group * a.batch
    lanes = []
    gk0 = None
    for w in range(a.lanes):
        ctx = E.Context(0)
        if a.share_keys and gk0 is not None:
            gk = gk0
        else:
            gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
            gk0 = gk0 or gk
        gen = torch.Generator(device=dev)
        gen.manual_seed(77 + w)
        lanes.append(dict(ctx=ctx, gk=gk, stream=torch.cuda.Stream(device=dev), nonces=Bn.make_device_nonces(gen, dev, B, S, S, n)))
    torch.cuda.synchronize()
    outs = []

    def submit(w):
        wk = lanes[w]
        with torch.cuda.stream(wk["stream"]):
            outs.append(E.gg20_sign(wk["ctx"], wk["gk"], wk["nonces"], B))
    for w in range(a.lanes):
        submit(w)
    torch.cuda.synchronize()
    outs.clear()
    t0 = time.perf_counter()
    for r in range(a.reps):
        for w in range(a.lanes):
            submit(w)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(bool((o[3] == 0).all().item()) for o in outs)
    total = a.reps * a.lanes * B
    print(json.dumps({"lanes": a.lanes, "group": a.group, "sessions_per_superbatch": B, "no_par": bool(os.environ.get("MPE_NO_PAR")),
                      "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "signatures_per_s": round(total / dt, 1),
                      "ms_per_superbatch_sustained": round(dt / (a.reps * a.lanes) * 1e3, 2), "host_enqueue_s": round(t_enq, 3), "seconds": round(dt, 3),
                      "latency_bound_ms": round(a.lanes * dt / (a.reps * a.lanes) * 1e3, 1), "all_signed": ok}))


if __name__ == "__main__":
    main()
