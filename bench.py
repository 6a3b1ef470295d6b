"""bench.py — headline benchmark (BASELINE.json config 2 core): batched Paillier-2048 modexp on MI355X.

One step = one pass of the hot path over one batch: 65 536 x (r^N mod N^2) — the modular
exponentiation with a 4096-bit modulus and a 2048-bit exponent that dominates
Paillier::encrypt_with_chosen_randomness (reference: src/utilities/mta/mod.rs:68-75) — over 16 keys,
inputs resident in HBM.  N>1: one process per GPU (torch.distributed / RCCL only for the barrier and
the max-reduce of the timing); every rank processes its own 65 536 items ("weak" scaling, no
data-path collective: independent units).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BATCH = 65536
BITS, EXP_BITS = 4096, 2048
K32 = BITS // 32
# algorithmic work: MAC(k) = 2k^2+k 32x32->64 multiply-accumulates per modular
# multiplication, modexp(k,E) = (E + ceil(E/4) + 16) * MAC(k), k = 128 limbs, E = 2048
ALG_MAC_PER_MODEXP = (EXP_BITS + EXP_BITS // 4 + 16) * (2 * K32 * K32 + K32)
# gfx950 v_mad_u64_u32 peak: 16 lanes/clk/SIMD (measured, profiles/r01_valu_rate.json) x 4 SIMD x 256 CU x 2.4 GHz
PEAK_MAC_PER_S = 16 * 4 * 256 * 2.4e9


def cpu_baseline(keys, sample, threads):
    """The GMP oracle (mpz_powm — the reference's own engine) on the host cores: `sample` items of the
    same workload split over `threads` threads (ctypes releases the GIL)."""
    import fixtures as F
    import orc
    r = np.random.default_rng(7)
    mods = F.words([k.NN for k in keys], K32)
    base = r.integers(0, 2**32, size=(sample, K32), dtype=np.uint32)
    exps = F.words([keys[i % len(keys)].N for i in range(sample)], EXP_BITS // 32)
    idx = (np.arange(sample) % len(keys)).astype(np.int32)
    chunks = np.array_split(np.arange(sample), threads)

    def run(ix):
        if len(ix):
            orc.modexp(mods, np.ascontiguousarray(base[ix]), np.ascontiguousarray(exps[ix]), idx[ix])
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, chunks))
    return sample / (time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import fixtures as F
    from multi_party_ecdsa_amd import engine as E
    keys = F.load_keys()
    ctx = E.Context(local_rank)
    dev = ctx.device
    ms = E.ModSet(ctx, BITS, [k.NN for k in keys])
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    # synthetic inputs: r uniform below 2^2048 (zero-extended to the 4096-bit interface), exponent = N of the key
    base = torch.zeros((BATCH, K32), dtype=torch.int32, device=dev)
    base[:, :64] = torch.randint(-2**31, 2**31 - 1, (BATCH, 64), dtype=torch.int32, device=dev, generator=g)
    n_words = torch.from_numpy(F.words([k.N for k in keys], EXP_BITS // 32).view(np.int32)).to(dev)
    idx = (torch.arange(BATCH, device=dev, dtype=torch.int32) % len(keys)).contiguous()
    exps = n_words[idx.long()].contiguous()
    out = torch.empty_like(base)

    def step():
        E.modexp_device(ctx, ms, base, exps, out, idx)

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev]            # HIP events on the launch stream: kernel duration
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # parity spot-check outside the timed region (the full parity suite is tests/ -m gpu)
        import orc
        hb = base[:8].cpu().numpy().view(np.uint32)
        ho = out[:8].cpu().numpy().view(np.uint32)
        want = orc.modexp(F.words([k.NN for k in keys], K32), np.ascontiguousarray(hb),
                          F.words([keys[i % len(keys)].N for i in range(8)], 64), list(range(8)))
        parity_ok = bool(np.array_equal(np.ascontiguousarray(ho), want))
        value = BATCH * world * args.steps / elapsed
        avg_kernel_s = float(np.mean(kern_ms)) * 1e-3
        achieved = BATCH * ALG_MAC_PER_MODEXP / avg_kernel_s
        res = {
            "metric": "Paillier-2048 modexp/s (4096-bit modulus, 2048-bit exponent; BASELINE config 2 core)",
            "value": value, "unit": "modexp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (radix 2^29) / u64 accumulators", "data": "synthetic",
            "config": {"workload": "65536 x r^N mod N^2 per GPU, 16 Paillier-2048 keys (tests/golden/keys16.json)",
                       "batch_per_gpu": BATCH, "modulus_bits": BITS, "exponent_bits": EXP_BITS,
                       "parallelism": f"session-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "valu-int (v_mad_u64_u32)", "achieved": achieved / 1e12, "peak": PEAK_MAC_PER_S / 1e12,
                         "unit": "TMAC/s (algorithmic 32x32+64 MACs, SURVEY.md 8d)", "frac": achieved / PEAK_MAC_PER_S,
                         "traffic": None, "kernel": "mpe::modexp_kernel<Cfg4096>", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "alg_mac_per_launch": BATCH * ALG_MAC_PER_MODEXP,
                         "alg_bytes_per_launch": BATCH * (512 + 256 + 512)},
            "parity_spot_check": parity_ok, "launch": ctx.launch_info(),
        }
        if not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 64)
            sample = 64 * threads                       # ~64 modexp per thread, ~0.7 s each thread at 10 ms/op
            v = cpu_baseline(keys, sample, threads)
            res["cpu_baseline"] = {"value": v, "unit": "modexp/s", "cores": threads, "kind": "port",
                                   "sample": f"{sample} items of the same workload (GMP mpz_powm oracle, {threads} threads)"}
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
