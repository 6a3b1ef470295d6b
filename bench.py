"""bench.py — headline benchmark, BASELINE.json config 2: batched Paillier-2048 encrypt + decrypt on MI355X.

One step = one pass of the hot path over one batch: 65 536 `Paillier::encrypt_with_chosen_randomness`
(reference src/utilities/mta/mod.rs:68-75) followed by 65 536 `Paillier::decrypt` (mta/mod.rs:165) of
those ciphertexts, 16 keys, inputs resident in HBM.  `value` counts Paillier operations (encrypts +
decrypts) per second; the dominant kernel is the 4096-bit / 2048-bit-exponent modexp inside encrypt
("Paillier-2048 modexp" of BASELINE.json's metric) and the roofline object is about that kernel.

N>1: one process per GPU; torch.distributed (RCCL) is used only for the barriers and the max-reduce
of the elapsed time.  Every rank processes its own 65 536 + 65 536 operations ("weak" scaling, no
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
# algorithmic work: MAC(k) = 2k^2+k 32x32->64 multiply-accumulates per modular
# multiplication on k 32-bit limbs; modexp(k,E) = (E + ceil(E/4) + 16) * MAC(k)


def mac(k):
    return 2 * k * k + k


def modexp_macs(k, e_bits):
    return (e_bits + (e_bits + 3) // 4 + 16) * mac(k)


ALG_MAC_MODEXP_4096_2048 = modexp_macs(128, 2048)          # "Paillier-2048 modexp" unit, 8.47e7
ALG_MAC_ENCRYPT = ALG_MAC_MODEXP_4096_2048 + 2 * mac(128)  # + (1+mN) product and the final mulmod
ALG_MAC_DECRYPT = 2 * modexp_macs(64, 1024) + 6 * mac(64)  # CRT halves + L/h/CRT multiplications
# gfx950 v_mad_u64_u32 peak: 16 lanes/clk/SIMD (measured, profiles/r01_valu_rate.json) x 4 SIMD x 256 CU x 2.4 GHz
PEAK_MAC_PER_S = 16 * 4 * 256 * 2.4e9


def cpu_baseline(keys, sample, threads):
    """The GMP oracle (mpz_powm — the reference's own engine) on the host cores: `sample` encrypts and
    `sample` decrypts of the same workload split over `threads` threads (ctypes releases the GIL)."""
    import fixtures as F
    import orc
    r = np.random.default_rng(7)
    nk = len(keys)
    N = F.words([k.N for k in keys], 64)
    p, q = F.words([k.p for k in keys], 32), F.words([k.q for k in keys], 32)
    m = np.zeros((sample, 64), dtype=np.uint32)
    m[:, :8] = r.integers(0, 2**32, size=(sample, 8), dtype=np.uint32)
    rr = np.zeros((sample, 64), dtype=np.uint32)
    rr[:, :63] = r.integers(0, 2**32, size=(sample, 63), dtype=np.uint32)
    idx = (np.arange(sample) % nk).astype(np.int32)
    chunks = [c for c in np.array_split(np.arange(sample), threads) if len(c)]

    def run(ix):
        c = orc.paillier_encrypt(N, np.ascontiguousarray(m[ix]), np.ascontiguousarray(rr[ix]), idx[ix])
        back = orc.paillier_decrypt(p, q, c, idx[ix])
        assert np.array_equal(back, m[ix])
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, chunks))
    return 2 * sample / (time.time() - t0)


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
    pk = E.PaillierKeys(ctx, p=[k.p for k in keys], q=[k.q for k in keys])
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    # synthetic inputs: half the plaintexts 256-bit (the k_i case), half ~2016-bit (beta')
    m = torch.zeros((BATCH, 64), dtype=torch.int32, device=dev)
    m[:, :8] = torch.randint(-2**31, 2**31 - 1, (BATCH, 8), dtype=torch.int32, device=dev, generator=g)
    m[BATCH // 2:, :63] = torch.randint(-2**31, 2**31 - 1, (BATCH // 2, 63), dtype=torch.int32, device=dev, generator=g)
    rr = torch.zeros((BATCH, 64), dtype=torch.int32, device=dev)
    rr[:, :63] = torch.randint(-2**31, 2**31 - 1, (BATCH, 63), dtype=torch.int32, device=dev, generator=g)
    idx = (torch.arange(BATCH, device=dev, dtype=torch.int32) % len(keys)).contiguous()
    c = torch.empty((BATCH, 128), dtype=torch.int32, device=dev)
    back = torch.empty((BATCH, 64), dtype=torch.int32, device=dev)

    def step():
        pk.encrypt_device(m, rr, idx, c)
        pk.decrypt_device(c, idx, back)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.prof_enable(True)                              # HIP events around every heavy-kernel launch, on the launch stream
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    recs = ctx.prof_collect()
    ctx.prof_enable(False)
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        parity_ok = bool(torch.equal(back, m))         # round trip over the whole batch (full parity: tests/ -m gpu)
        import orc
        want = orc.paillier_encrypt(F.words([k.N for k in keys], 64), np.ascontiguousarray(m[:8].cpu().numpy().view(np.uint32)),
                                    np.ascontiguousarray(rr[:8].cpu().numpy().view(np.uint32)), list(range(8)))
        parity_ok = parity_ok and bool(np.array_equal(np.ascontiguousarray(c[:8].cpu().numpy().view(np.uint32)), want))
        dom = [r["ms"] for r in recs if r["kind"] == 0 and r["bits"] == 4096 and r["exp_words"] == 64]
        dec = [r["ms"] for r in recs if r["kind"] == 0 and r["bits"] == 2048]
        avg_dom_s = float(np.mean(dom)) * 1e-3
        achieved = BATCH * ALG_MAC_MODEXP_4096_2048 / avg_dom_s
        heavy_ms = sum(r["ms"] for r in recs) / args.steps
        res = {
            "metric": "Paillier-2048 ops/s per GPU-job (encrypt+decrypt, BASELINE config 2); modexp/s in roofline",
            "value": 2 * BATCH * world * args.steps / elapsed, "unit": "paillier_ops/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (radix 2^29) / u64 accumulators", "data": "synthetic",
            "config": {"workload": "65536 Paillier-2048 encrypt_with_chosen_randomness + 65536 decrypt per GPU, 16 keys "
                                   "(tests/golden/keys16.json)", "batch_per_gpu": BATCH,
                       "parallelism": f"session-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "valu-int (v_mad_u64_u32 issue rate; HBM traffic is negligible)",
                         "achieved": achieved / 1e12, "peak": PEAK_MAC_PER_S / 1e12,
                         "unit": "TMAC/s (algorithmic 32x32+64 MACs, SURVEY.md 8d)", "frac": achieved / PEAK_MAC_PER_S,
                         "traffic": None, "kernel": "mpe::modexp_kernel<Cfg<4096,29,18,8>> (r^N mod N^2 in encrypt)",
                         "avg_kernel_ms": avg_dom_s * 1e3, "launches_timed": len(dom),
                         "modexp4096_per_s": BATCH / avg_dom_s,
                         "alg_mac_per_launch": BATCH * ALG_MAC_MODEXP_4096_2048,
                         "alg_bytes_per_launch": BATCH * (256 + 256 + 512)},
            "breakdown": {"encrypt_modexp4096_ms": float(np.mean(dom)), "decrypt_modexp2048_ms": float(np.mean(dec)),
                          "heavy_kernels_ms_per_step": heavy_ms,
                          "whole_step_alg_TMAC_per_s": BATCH * (ALG_MAC_ENCRYPT + ALG_MAC_DECRYPT) * args.steps / elapsed / 1e12,
                          "encrypt_per_s": BATCH / (float(np.mean(dom)) * 1e-3)},
            "parity_spot_check": parity_ok, "launch": ctx.launch_info(),
        }
        if not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 64)
            sample = 48 * threads
            v = cpu_baseline(keys, sample, threads)
            res["cpu_baseline"] = {"value": v, "unit": "paillier_ops/s", "cores": threads, "kind": "port",
                                   "sample": f"{sample} encrypts + {sample} decrypts of the same workload "
                                             f"(GMP oracle, {threads} threads)"}
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
