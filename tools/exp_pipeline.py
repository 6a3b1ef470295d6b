#!/usr/bin/env python3
"""Throughput / latency of the pipelined engine (mpe_gg20_pipeline_*) on a stream of 1 024-session batches sampled on the device.
  python tools/exp_pipeline.py --lanes 2 --group 4 --batches 48"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--batches", type=int, default=48)
    ap.add_argument("--explicit", action="store_true", help="submit caller-sampled arrays instead of a seed")
    a = ap.parse_args()
    import numpy as np, torch
    import fixtures as F, gg20_fixture as G, bench as Bn
    from multi_party_ecdsa_amd import engine as E
    ctx = E.Context(0)
    dev = ctx.device
    lk = G.make_local_keys(F.load_keys(), 1, 3, [0, 1])
    gk = E.Gg20Keys(ctx, 1, 3, [0, 1], lk["arrays"])
    pipe = E.Gg20Pipeline(ctx, gk, a.batch, group=a.group, lanes=a.lanes)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    msg = Bn.rand_words(gen, dev, a.batch, 8, 8)
    nonces = Bn.make_device_nonces(gen, dev, a.batch, 2, 2, 3) if a.explicit else None
    seed = bytes(range(32))

    def run(n, c0):
        ts = [pipe.submit(nonces) if a.explicit else pipe.submit_seeded(seed, c0 + i, msg) for i in range(n)]
        pipe.flush()
        return ts
    for t in run(a.lanes * a.group, 0):
        pipe.wait(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = run(a.batches, 1000)
    t_enq = time.perf_counter() - t0
    outs = [pipe.wait(t) for t in ts]
    dt = time.perf_counter() - t0
    lat = sorted(pipe.latency_ms(t) for t in ts)
    pas = sorted(pipe.pass_ms(t) for t in ts)
    ok = all(bool((o[3] == 0).all().item()) for o in outs)
    print(json.dumps({"lanes": a.lanes, "group": a.group, "batch": a.batch, "batches": a.batches, "seeded": not a.explicit,
                      "merge_r1": not os.environ.get("MPE_NO_MERGE_R1"),
                      "signatures_per_s": round(a.batches * a.batch / dt, 1), "host_enqueue_s": round(t_enq, 3), "seconds": round(dt, 3),
                      "latency_ms_p50": round(lat[len(lat) // 2], 1), "latency_ms_max": round(lat[-1], 1), "pass_ms_p50": round(pas[len(pas) // 2], 1), "pass_ms_max": round(pas[-1], 1), "all_signed": ok,
                      "sampler_failures": pipe.sampler_failures()}))
    pipe.close()


if __name__ == "__main__":
    main()
