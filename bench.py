"""bench.py — headline benchmark: GG20 (t=1, n=3) threshold-ECDSA signatures per second on MI355X, with the
Paillier-2048 numbers of BASELINE.json's metric and every other BASELINE config in the same JSON line.

One step = one pass of the hot path over one batch: B concurrent signing sessions (BASELINE config 4 shape:
t=1, n=3, signers {1,2}, one LocalKey fixture shared by all sessions, distinct nonces and messages per
session), Round0..Round7 of both parties computed in lock-step on the GPU through `mpe_gg20_sign` — faithful
work (every range proof verified for both MessageB::b calls, every party verifies every PDL proof, exactly as
src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign/rounds.rs:151-175,546-558).  All inputs (keys,
nonces, messages) are resident in HBM before the timed region.  `value` = signatures / second.

`roofline`: the dominant kernel is the exponentiation modulo N^2 (`pair_modexp_kernel<Cfg2048>`); all its launches of the
timed region are timed with HIP events on the launch stream (`mpe_prof_*`).  `frac` = the 32x32+64 MACs the EXECUTED
algorithm needs (N-adic pairs on ideal 32-bit limbs: 2 MAC(64) per squaring, 2.5 per multiplication, the kernel's own
window counts) per second / the measured v_mad_u64_u32 issue peak — a hardware-utilisation figure <= 1.  SURVEY.md 8(d)'s
unit (textbook CIOS on the 4096-bit integers) is reported beside it as `alg_unit_*`; it exceeds the peak because the
kernel computes the same residues with ~0.46x those MACs.
`roofline_secondary[]` / `whole_step`: the same accounting for every other profiled kernel and for the whole step.
`openssl_verified`: every signature of the timed batch under OpenSSL's ECDSA_do_verify (after the timed region; tests/ossl.py).
`configs`: c2 (65 536 Paillier encrypt + decrypt, 16 keys), c3 (262 144 EC scalar multiplications and PDL-with-slack
prove + verify, prefix checked against the oracle and OpenSSL, 1 % corrupted proofs all rejected), c3b (65 536 BobProof generate +
BobProofExt verify), c4_literal_1024 (config 4 at its literal size), c5_share_t2n5_8192 (one GPU's share of config 5), measured
right after the timed region.
`cpu_baseline`: the GMP oracle (oracle/gg20_oracle.c — the reference's formulas over the reference's own bignum engine)
on the host cores for a bounded sample of the same sessions (16 per thread), plus its single-thread rate; the GPU
signatures are checked against it bit for bit.

N>1 (one process per GPU; RCCL).  `python bench.py --gpus N` with no launcher around it spawns its N ranks itself (re-exec under
torch.distributed.run, 127.0.0.1, a free port); under the driver's own torch.distributed.run line it reads the environment.
The line then carries n_gpus, per_rank{signatures_per_s[], min, max}, an rccl{} block (all-reduce of ones == N, version) and —
session mode — the node's Paillier ops/s.  --share-device: every rank on cuda:0 over gloo (the same code path on a 1-GPU box).
  --mode session (default): sessions sharded across ranks, no data-path collective (independent units);
  --mode party: the parties of a session live on different GPUs (party p of session block s on rank (s + p) % N) and
  every round's messages travel through one all-gather (BASELINE config 5); same per-GPU work.
At N > 1 the line of the timed region is assembled before the collective sections that follow it (per-rank oracle parity, `mode_b{}`, config 2 on
every GPU); if one of those wedges, a per-rank watchdog (MPE_BENCH_POST_TIMEOUT_S, default 420 s) makes rank 0 print that line with
`post_timing_sections{completed: false}` and every rank leave with status 0.
--dump-launches adds the timed region's profiled launches (kind, modulus bits, exponent words, batch, ms) — what tools/ab_*.sh compare.
Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

# (c4_stream_1024 runs in THIS process through mpe_gg20_pipeline with the runtime's default hardware queues — round 4's child process
#  with GPU_MAX_HW_QUEUES=16 is gone)
import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_MAC_PER_S = 16 * 4 * 256 * 2.4e9     # gfx950 v_mad_u64_u32: 16 lanes/clk/SIMD (profiles/r01_valu_rate.json)
LIMB_INFLATION = 4 * 72 * 71 / (2 * (2 * 64 * 64 + 64))   # executed 29-bit MACs of a squaring modulo N^2 (2 passes x 2 streams x 72 limbs x 71 steps) / its ideal 2 MAC(64)
EXP_BITS = {8: 256, 24: 768, 25: 769, 32: 1024, 64: 2048, 72: 2304, 80: 2560, 81: 2561, 88: 2816, 89: 2817}


def mac(k):
    return 2 * k * k + k                     # 32x32->64 MACs per modular multiplication, k 32-bit limbs


def modexp_macs(k, e_bits):
    return (e_bits + (e_bits + 3) // 4 + 16) * mac(k)


def window_bits(exp_words):
    return 4 if exp_words <= 8 else (5 if exp_words < 48 else 6)       # the library's rule (mpe_lib.hip)


_SLIDING = {}
SLIDING_EXPONENTS = None      # the PUBLIC exponents the timed launches really raise to (the signers' Paillier moduli N); set by main()


def sliding_counts(bits, wb):
    """(squarings, window multiplications) of the kernel's left-to-right sliding-window schedule (mpe_pairexp.h slide_window)
    for the public exponents of the run — the fixture's own moduli N when main() has set them (SLIDING_EXPONENTS: an exact
    count, averaged over the keys in use), else the expectation over 32 seeded `bits`-bit exponents with the top bit set"""
    if (bits, wb) not in _SLIDING:
        import random
        rnd = random.Random(2048)
        sq_t = mul_t = 0
        exps = [e for e in (SLIDING_EXPONENTS or []) if e.bit_length() > bits - 32]
        if not exps:
            exps = [rnd.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(32)]
        for e in exps:
            i, first, sq, mul = e.bit_length() - 1, True, 0, 0
            while i >= 0:
                if not (e >> i) & 1:
                    i -= 1
                    sq += 1
                    continue
                lo = max(i - wb + 1, 0)
                while not (e >> lo) & 1:
                    lo += 1
                if not first:
                    sq += i - lo + 1
                    mul += 1
                first = False
                i = lo - 1
            sq_t += sq
            mul_t += mul
        _SLIDING[(bits, wb)] = (sq_t / len(exps), mul_t / len(exps))
    return _SLIDING[(bits, wb)]


def slid(x):
    """share of a launch that ran the sliding-window schedule: counted by the kernel itself per (wave, trip) (mpe_prof_rec.
    sliding_frac) — "kind 6" only says the launch was ALLOWED to slide; waves that straddle a key boundary and two-base ladders
    whose first window would dip below the second exponent keep the fixed windows"""
    if x["kind"] != 6:
        return 0.0
    f = x.get("sliding_frac", -1.0)
    return 1.0 if f is None or f < 0 else min(1.0, float(f))


def pair_modexp_macs(k, exp_words, exp2_words=0, sliding=False):
    """32x32->64 MACs of ONE exponentiation modulo a square N^2 in the N-adic pair arithmetic the kernel runs
    (mpe_pairexp.h), counted on k 32-bit limbs of N (the ideal radix): a squaring is 2 half-size Montgomery passes
    (2 MAC(k)), a multiplication 2.5.  Fixed windows as the kernel chooses them; sliding=True: the schedule it runs for the
    PUBLIC exponent N (odd powers only in the table: one squaring + 2^(wb-1) - 1 multiplications; expected window count)."""
    wb = window_bits(exp_words)
    f = float(sliding)                              # True / False, or the measured share of the launch on the sliding schedule
    tot = 0.0
    for share, sl in ((f, True), (1.0 - f, False)):
        if share <= 0:
            continue
        if sl:
            sq, win = sliding_counts(32 * exp_words, wb)
            sq += 1                                     # x^2 for the table of odd powers
            mul = (1 << (wb - 1)) - 1 + win + 2         # x^3 .. x^(2^wb - 1), the windows, conversion in and out
        else:
            nwin = (32 * exp_words + wb - 1) // wb
            sq = (nwin - 1) * wb
            mul = (1 << wb) + nwin + 2                  # table (with the conversion in), one per window, conversion out
        if exp2_words:
            mul += 16 + 8 * exp2_words + 1
        tot += share * (2 * sq + 2.5 * mul) * mac(k)
    return tot


def plain_modexp_macs(k, exp_words, exp2_words=0):
    """MACs of one fixed-window Montgomery exponentiation on k ideal 32-bit limbs with the kernel's own window rule
    (modexp_kernel, and the pair kernel's `half` mode): one MAC(k) per squaring / multiplication"""
    wb = window_bits(exp_words)
    nwin = (32 * exp_words + wb - 1) // wb
    mul = (1 << wb) + nwin + 2
    if exp2_words:
        mul += 16 + 8 * exp2_words + 1
    return ((nwin - 1) * wb + mul) * mac(k)


def fixed_base_macs(k, exp_words, fb_wb):
    """fixed-base ladder: one Montgomery multiplication per table window, plus the conversion out"""
    return ((32 * exp_words + fb_wb - 1) // fb_wb + 1) * mac(k)


def secondary_rooflines(recs, elapsed):
    """executed-MAC rate of every heavy kernel besides the dominant one, from the same HIP-event records: which share of the
    step each takes and how far from the v_mad_u64_u32 peak it runs (same accounting as `roofline`: ideal 32-bit limbs)"""
    groups = {
        "pair_modexp_kernel<Cfg<1024,29,18,2>> (key holder's CRT halves modulo p^2 | q^2)":
            ([x for x in recs if x["kind"] in (3, 6) and x["bits"] == 2048], lambda x: pair_modexp_macs(32, x["exp_words"], x.get("exp2_words", 0), sliding=slid(x))),
        "pair_modexp_kernel<Cfg<1024,...>> half mode (x^(q mod p-1) modulo p)":
            ([x for x in recs if x["kind"] == 4 and x["bits"] == 1024], lambda x: plain_modexp_macs(32, x["exp_words"], x.get("exp2_words", 0))),
        "pair_modexp_kernel<Cfg<2048,...>> half mode (modulo N)":
            ([x for x in recs if x["kind"] == 4 and x["bits"] == 2048], lambda x: plain_modexp_macs(64, x["exp_words"], x.get("exp2_words", 0))),
        "modexp_kernel<Cfg<2048,29,18,4>> (variable base modulo N~)":
            ([x for x in recs if x["kind"] == 0 and x["bits"] == 2048], lambda x: plain_modexp_macs(64, x["exp_words"], x.get("exp2_words", 0))),
        "fb_modexp_kernel<Cfg<2048,29,18,4>> (h1^x, h2^x modulo N~ from window tables)":
            ([x for x in recs if x["kind"] == 5], lambda x: fixed_base_macs(64, x["exp_words"], x.get("exp2_words", 0) or 13)),
        "modmul_kernel (2048 / 4096 bit)":
            ([x for x in recs if x["kind"] == 1], lambda x: mac(x["bits"] // 32)),
    }
    out = []
    for name, (rs, macs) in groups.items():
        if not rs:
            continue
        t = sum(x["ms"] for x in rs) * 1e-3
        m = sum(x["batch"] * macs(x) for x in rs)
        out.append({"kernel": name, "launches": len(rs), "seconds": t, "time_share_of_step": t / elapsed, "executed_TMAC_per_s": m / t / 1e12 if t else None,
                    "frac": m / t / PEAK_MAC_PER_S if t else None})
    return out


def dominant_roofline(recs, elapsed):
    """the same accounting as the headline's `roofline` for the launches modulo N^2 of another shape (c4_literal_1024, c5_share_t2n5_8192)"""
    dom = [x for x in recs if x["kind"] in (3, 6) and x["bits"] == 4096]
    if not dom:
        return None
    t = sum(x["ms"] for x in dom) * 1e-3
    m = sum(x["batch"] * pair_modexp_macs(64, x["exp_words"], x.get("exp2_words", 0), sliding=slid(x)) for x in dom)
    return {"kernel": "pair_modexp_kernel<Cfg<2048,...>> (all launches modulo N^2)", "launches": len(dom), "items": int(sum(x["batch"] for x in dom)),
            "seconds": t, "time_share_of_step": t / elapsed, "executed_TMAC_per_s": m / t / 1e12, "frac": m / t / PEAK_MAC_PER_S}


def executed_macs(recs):
    """32x32+64 MACs (ideal 32-bit limbs) of every profiled heavy launch: the whole-step companion of `roofline`"""
    tot = 0.0
    for x in recs:
        k, b, ew, e2 = x["kind"], x["bits"], x["exp_words"], x.get("exp2_words", 0)
        if k in (3, 6):
            tot += x["batch"] * pair_modexp_macs(b // 64, ew, e2, sliding=slid(x))
        elif k in (0, 4):
            tot += x["batch"] * plain_modexp_macs(b // 32, ew, e2)
        elif k == 5:
            tot += x["batch"] * fixed_base_macs(b // 32, ew, e2 or 13)
        elif k == 1:
            tot += x["batch"] * mac(b // 32)
    return tot


def whole_step(recs, wall_s):
    m = executed_macs(recs)
    return {"executed_TMAC_per_s": m / wall_s / 1e12, "frac": m / wall_s / PEAK_MAC_PER_S, "heavy_kernel_seconds": sum(x["ms"] for x in recs) * 1e-3,
            "wall_seconds": wall_s, "note": "all modexp / modmul launches of the pass (executed MACs on ideal 32-bit limbs) / wall time / peak; "
                                            "EC, hashing, packing and idle gaps count as time only"}


def sig_macs(S, n):
    """algorithmic MACs per signature, faithful path"""
    b2048 = n * 6400 + 2 * (S - 1) * n * 3842 + 2 * (S - 1) * 2048 + (S - 1) * 6400 + S * (S - 1) * 3843
    b4096 = 2048 + n * 2048 + 2 * (S - 1) * (n * 2304 + 2304) + (S - 1) * 2816 + S * (S - 1) * 3074
    return 1.25 * (b2048 * mac(64) + b4096 * mac(128)) * S


def rand_words(gen, dev, rows, width, full):
    """device int32 [rows, width]: `full` random words, the rest zero (value < 2^(32*full))"""
    t = torch.zeros((rows, width), dtype=torch.int32, device=dev)
    t[:, :full] = torch.randint(-2**31, 2**31 - 1, (rows, full), dtype=torch.int32, device=dev, generator=gen)
    return t


def _scalar(gen, dev, rows):
    t = rand_words(gen, dev, rows, 8, 8)
    t[:, 7] &= 0x3FFFFFFF                      # < 2^254 < q, nonzero with overwhelming probability
    t[:, 0] |= 1
    return t


def make_device_nonces(gen, dev, B, L, S, n):
    """Synthetic nonces inside the reference's sampling ranges (party_i.rs:559-563,574; mta/mod.rs:57,97-98;
    range_proofs.rs:48-51; zk_pdl_with_slack/mod.rs:73-77): uniform below a power of two under each bound.
    Leading dimensions [B][L] (L local parties)."""
    P = L * (S - 1)
    sc = lambda rows: _scalar(gen, dev, rows)
    z = dict(k=sc(B * L), gamma=sc(B * L), blind=rand_words(gen, dev, B * L, 8, 8), r_a=rand_words(gen, dev, B * L, 64, 63),
             al_alpha=rand_words(gen, dev, B * L * n, 24, 23), al_beta=rand_words(gen, dev, B * L * n, 64, 63),
             al_gamma=rand_words(gen, dev, B * L * n, 88, 87), al_rho=rand_words(gen, dev, B * L * n, 72, 71),
             mb_beta_tag=rand_words(gen, dev, B * P * 2, 64, 63), mb_r=rand_words(gen, dev, B * P * 2, 64, 63),
             mb_nonce_b=sc(B * P * 2), mb_nonce_bt=sc(B * P * 2), l=sc(B * L), ped_s1=sc(B * L), ped_s2=sc(B * L),
             pdl_alpha=rand_words(gen, dev, B * P, 24, 23), pdl_beta=rand_words(gen, dev, B * P, 64, 63),
             pdl_rho=rand_words(gen, dev, B * P, 72, 71), pdl_gamma=rand_words(gen, dev, B * P, 88, 87),
             heg_s1=sc(B * L), heg_s2=sc(B * L), msg=rand_words(gen, dev, B, 8, 8))
    return z


def _host(nonces):
    return {f: np.ascontiguousarray(v.cpu().numpy().view(np.uint32)) for f, v in nonces.items()}


def host_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()
        if a != "max":
            quota = float(a) / float(b)
    except (OSError, ValueError):
        pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline_gg20(lk, host_nonces, sample, threads):
    """the oracle on `threads` host threads over `sample` sessions (ctypes releases the GIL); returns sig/s and the signatures"""
    import gg20_fixture as G
    chunks = [c for c in np.array_split(np.arange(sample), threads) if len(c)]
    outs = {}

    def run(ix):
        outs[int(ix[0])] = G.oracle_sign(lk, host_nonces, sample, first=int(ix[0]), count=len(ix))
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, chunks))
    dt = time.time() - t0
    r = np.zeros((sample, 8), dtype=np.uint32); s = np.zeros((sample, 8), dtype=np.uint32)
    recid = np.zeros(sample, dtype=np.int32); status = np.zeros(sample, dtype=np.int32)
    for first, (rr, ss, rc, _, stt) in outs.items():
        cnt = [len(c) for c in chunks if int(c[0]) == first][0]
        sl = slice(first, first + cnt)
        r[sl], s[sl], recid[sl], status[sl] = rr[sl], ss[sl], rc[sl], stt[sl]
    return sample / dt, r, s, recid, status


def paillier_config2(ctx, E, keys, F, steps=1, oracle_threads=0, oracle_items=0):
    """BASELINE config 2: 65 536 encrypt + 65 536 decrypt, 16 keys.  Encryption is timed twice: by the key holder
    (p, q known: the p^2 | q^2 path) and by a peer that only has N (the plain exponentiation r^N mod N^2 — the
    'Paillier-2048 modexp/s' of the metric, with its kernel time from the HIP-event records)."""
    B = 65536
    dev = ctx.device
    sk = E.PaillierKeys(ctx, p=[k.p for k in keys], q=[k.q for k in keys])
    pk = E.PaillierKeys(ctx, N=[k.N for k in keys])
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    m = rand_words(g, dev, B, 64, 8)
    m[B // 2:, :63] = torch.randint(-2**31, 2**31 - 1, (B // 2, 63), dtype=torch.int32, device=dev, generator=g)
    rr = rand_words(g, dev, B, 64, 63)
    idx = (torch.arange(B, device=dev, dtype=torch.int32) % len(keys)).contiguous()
    c = torch.empty((B, 128), dtype=torch.int32, device=dev)
    c2 = torch.empty((B, 128), dtype=torch.int32, device=dev)
    back = torch.empty((B, 64), dtype=torch.int32, device=dev)
    sk.encrypt_device(m, rr, idx, c); pk.encrypt_device(m, rr, idx, c2); sk.decrypt_device(c, idx, back)   # warm-up

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    t_enc = timed(lambda: sk.encrypt_device(m, rr, idx, c))
    t_dec = timed(lambda: sk.decrypt_device(c, idx, back))
    ctx.prof_enable(True)
    t_pub = timed(lambda: pk.encrypt_device(m, rr, idx, c2))
    recs = ctx.prof_collect()
    ctx.prof_enable(False)
    kern = float(np.mean([r["ms"] for r in recs if r["kind"] in (0, 3, 6) and r["bits"] == 4096])) * 1e-3
    c2_recs = [r for r in recs if r["kind"] == 6 and r["bits"] == 4096]
    c2_sliding = float(np.mean([slid(r) for r in c2_recs])) if c2_recs else 0.0          # the share the kernel itself counted
    cpu = {}
    if oracle_threads:
        # the reference CPU path beside it: the GMP oracle (reference formulas over mpz_powm) on a bounded prefix, bit-exact check included
        import orc
        per = max(1, (oracle_items + oracle_threads - 1) // oracle_threads) if oracle_items else 24   # tests ask for all 65 536
        n_cpu = min(B, per * oracle_threads)
        oracle_threads = (n_cpu + per - 1) // per
        Nw, pw, qw = F.words([k.N for k in keys], 64), F.words([k.p for k in keys], 32), F.words([k.q for k in keys], 32)
        hm, hr, hi = (np.ascontiguousarray(t_[:n_cpu].cpu().numpy().view(np.uint32)) for t_ in (m, rr, idx))
        hi = hi.view(np.int32).reshape(-1)
        outs = {}

        def run(j):
            sl = slice(j * per, (j + 1) * per)
            cc = orc.paillier_encrypt(Nw, hm[sl], hr[sl], list(hi[sl]))
            outs[j] = (cc, orc.paillier_decrypt(pw, qw, cc, list(hi[sl])))
        t0 = time.time()
        with ThreadPoolExecutor(oracle_threads) as ex:
            list(ex.map(run, range(oracle_threads)))
        dt = time.time() - t0
        gc, gb = c[:n_cpu].cpu().numpy().view(np.uint32), back[:n_cpu].cpu().numpy().view(np.uint32)
        same = all(np.array_equal(outs[j][0], gc[j * per:(j + 1) * per]) and np.array_equal(outs[j][1], gb[j * per:(j + 1) * per])
                   for j in range(oracle_threads))
        cpu = {"oracle_ops_per_s": 2 * n_cpu / dt, "oracle_threads": oracle_threads, "oracle_sample": n_cpu, "parity_prefix": n_cpu,
               "parity_vs_oracle_on_prefix": bool(same)}
    return {**cpu, "ops_per_s": 2 * B / (t_enc + t_dec), "batch": B,
            "roundtrip_ok": bool(torch.equal(back, m)), "holder_equals_public_ciphertext": bool(torch.equal(c, c2)),
            "encrypt_per_s": B / t_enc, "decrypt_per_s": B / t_dec, "encrypt_public_key_per_s": B / t_pub,
            "modexp4096_2048_per_s": B / kern,
            "modexp4096_executed_TMAC_per_s": B * pair_modexp_macs(64, 64, sliding=c2_sliding) / kern / 1e12,
            "modexp4096_executed_frac": B * pair_modexp_macs(64, 64, sliding=c2_sliding) / kern / PEAK_MAC_PER_S, "modexp4096_sliding_windows": c2_sliding > 0, "modexp4096_sliding_share_of_waves": c2_sliding,
            "modexp4096_alg_unit_TMAC_per_s": B * modexp_macs(128, 2048) / kern / 1e12}


def config3(ctx, E, keys, F, B=262144, prefix=4096, threads=None, oracle=True):
    """BASELINE config 3: B secp256k1 scalar multiplications (fixed base x G, variable base x P) and
    B PDLwSlackProof::prove + verify over K = 16 (ek, N~, h1, h2) tuples, witnesses x uniform < 2^224 < q, nonces in the
    reference's ranges; a `prefix` of the proofs compared with the oracle bit for bit, 100 % of the honest proofs
    accepted, 1 % deliberately corrupted -> all rejected (and only those)."""
    import orc
    dev = ctx.device
    K = len(keys)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    pk = E.PaillierKeys(ctx, p=[k.p for k in keys], q=[k.q for k in keys])          # the prover owns the key
    pub = E.PaillierKeys(ctx, N=[k.N for k in keys])
    stm = E.Statements(ctx, [k.Nt for k in keys], [k.h1 for k in keys], [k.h2 for k in keys])
    x = rand_words(g, dev, B, 8, 7)
    rr = rand_words(g, dev, B, 64, 63)
    nonces = dict(alpha=rand_words(g, dev, B, 24, 23), beta=rand_words(g, dev, B, 64, 63), rho=rand_words(g, dev, B, 72, 71),
                  gamma=rand_words(g, dev, B, 88, 87))
    kidx = (torch.arange(B, device=dev, dtype=torch.int32) % K).contiguous()
    sidx = ((torch.arange(B, device=dev, dtype=torch.int32) * 7 + 3) % K).contiguous()
    kb = rand_words(g, dev, B, 8, 7)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out
    E.ec_mul_base(ctx, kb[:1024])                                                    # warm-up
    t_fix, Gp = timed(lambda: E.ec_mul_base(ctx, kb))
    t_var, Qp = timed(lambda: E.ec_mul(ctx, x, Gp))
    xm = torch.zeros((B, 64), dtype=torch.int32, device=dev)
    xm[:, :8] = x
    c = pk.encrypt_device(xm, rr, kidx)
    small = {f: v[:256] for f, v in nonces.items()}
    E.pdl_verify(ctx, pub, stm, c[:256], Qp[:256], Gp[:256], E.pdl_prove(ctx, pk, stm, c[:256], Qp[:256], Gp[:256], x[:256], rr[:256], small,
                                                                         kidx[:256], sidx[:256]), kidx[:256], sidx[:256])
    ctx.prof_enable(True)
    t_prove, pr = timed(lambda: E.pdl_prove(ctx, pk, stm, c, Qp, Gp, x, rr, nonces, kidx, sidx))
    rec_prove = ctx.prof_collect(4096)
    ctx.prof_enable(True)                                                            # (re-arming clears the records)
    t_ver, ok = timed(lambda: E.pdl_verify(ctx, pub, stm, c, Qp, Gp, pr, kidx, sidx))
    rec_ver = ctx.prof_collect(4096)
    ctx.prof_enable(False)
    accepted = int(ok.sum())
    bad = {k: v.clone() for k, v in pr.items()}
    fields = ["z", "u2", "u3", "s1", "s2", "s3"]
    nbad = 0
    for q_, f in enumerate(fields):                                                   # 1 %: every 100th proof, a different field each time
        sl = slice(q_ * 100 + 7, B, 600)
        bad[f][sl, 1] ^= 0x10
        nbad += len(range(*sl.indices(B)))
    okb = E.pdl_verify(ctx, pub, stm, c, Qp, Gp, bad, kidx, sidx).cpu().numpy()
    mask = np.zeros(B, dtype=bool)
    for q_ in range(len(fields)):
        mask[q_ * 100 + 7:B:600] = True
    out = {"instances": B, "keys": K, "ec_fixed_per_s": B / t_fix, "ec_var_per_s": B / t_var, "pdl_prove_per_s": B / t_prove,
           "pdl_verify_per_s": B / t_ver, "accepted": accepted, "accept_rate": accepted / B,
           "corrupted": int(nbad), "corrupted_1pct_all_rejected": bool((okb[mask] == 0).all() and (okb[~mask] == 1).all()),
           "pdl_prove_whole_step": whole_step(rec_prove, t_prove), "pdl_verify_whole_step": whole_step(rec_ver, t_ver)}
    if oracle:
        n = min(prefix, B)
        threads = threads or min(host_cores()[0], 64)
        h = lambda t: np.ascontiguousarray(t[:n].cpu().numpy().view(np.uint32))
        tabs = dict(N=F.words([k.N for k in keys], 64), Nt=F.words([k.Nt for k in keys], 64), h1=F.words([k.h1 for k in keys], 64),
                    h2=F.words([k.h2 for k in keys], 64))
        hin = dict(c=h(c), Q=h(Qp), G=h(Gp), x=h(x), r=h(rr), **{f: h(v) for f, v in nonces.items()})
        ki, si = kidx[:n].cpu().numpy(), sidx[:n].cpu().numpy()
        got = {f: h(v) for f, v in pr.items()}
        chunks = [c_ for c_ in np.array_split(np.arange(n), threads) if len(c_)]

        def run(ix):
            sl = slice(int(ix[0]), int(ix[-1]) + 1)
            want = orc.pdl_prove(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], ki[sl], si[sl], hin["c"][sl], hin["Q"][sl], hin["G"][sl],
                                 hin["x"][sl], hin["r"][sl], hin["alpha"][sl], hin["beta"][sl], hin["rho"][sl], hin["gamma"][sl])
            good = all(np.array_equal(got[f][sl], want[f]) for f in want)
            okv = orc.pdl_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], ki[sl], si[sl], hin["c"][sl], hin["Q"][sl], hin["G"][sl], want)
            return good and bool(okv.all())
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(run, chunks))
        out.update({"parity_prefix": n, f"parity_prefix_{n}": bool(all(res)), "oracle_prove_verify_per_s": n / (time.time() - t0),
                    "oracle_threads": threads})
        # the EC results of the prefix against the oracle too
        out["ec_parity_prefix"] = bool(np.array_equal(hin["G"], orc.ec_mul_base(h(kb))) and np.array_equal(hin["Q"], orc.ec_mul(hin["x"], hin["G"])))
        import ossl
        out["ec_parity_prefix_openssl"] = bool(np.array_equal(hin["G"], ossl.ec_mul(h(kb), threads=threads)) and
                                               np.array_equal(hin["Q"], ossl.ec_mul(hin["x"], hin["G"], threads=threads)))
    return out


def openssl_verify_all(y_words, msg, r, s, threads):
    """every signature of a batch under the wallet's public key with OpenSSL's ECDSA_do_verify (tests/ossl.py): the independent
    third-party check the reference runs with libsecp256k1 (gg_2020/test.rs:711-748).  After the timed region, on host copies."""
    import ossl
    u = lambda a: np.ascontiguousarray(a if isinstance(a, np.ndarray) else a.cpu().numpy()).view(np.uint32)
    t0 = time.time()
    ok = ossl.ecdsa_verify(u(y_words), u(msg), u(r), u(s), threads=threads)
    return {"openssl_verified": int(ok.sum()), "of": int(ok.shape[0]), "openssl": ossl.version(), "seconds": round(time.time() - t0, 2)}


def bob_section(ctx, E, keys, F, B=65536, prefix=1024, oracle=True, threads=None):
    """The Bob-side MtA range proof as a measured workload (`BobProof::generate` / `BobProofExt::verify`,
    range_proofs.rs:321-534 — not called by the GG20 state machine, so no other section times it): B instances over K = 16
    (ek, N~, h1, h2) tuples with check = true (the `u = alpha G` / `s1 G = e X + u` extension included), nonces in the reference's
    ranges (:231-237); a prefix compared with the oracle bit for bit, every honest proof accepted, 1 % corrupted all rejected."""
    import orc
    dev = ctx.device
    K = len(keys)
    g = torch.Generator(device=dev)
    g.manual_seed(24)
    pub = E.PaillierKeys(ctx, N=[k.N for k in keys])                                  # Bob only has Alice's public key
    stm = E.Statements(ctx, [k.Nt for k in keys], [k.h1 for k in keys], [k.h2 for k in keys])
    kidx = (torch.arange(B, device=dev, dtype=torch.int32) % K).contiguous()
    sidx = ((torch.arange(B, device=dev, dtype=torch.int32) * 5 + 1) % K).contiguous()
    a = rand_words(g, dev, B, 64, 7)
    b = rand_words(g, dev, B, 8, 7)
    beta_prim = rand_words(g, dev, B, 64, 63)
    r = rand_words(g, dev, B, 64, 63)
    a_enc = pub.encrypt_device(a, rand_words(g, dev, B, 64, 63), kidx)
    mta = pub.add_device(pub.mul_device(a_enc, b, kidx), pub.encrypt_device(beta_prim, r, kidx), kidx)     # c_a^b * Enc(beta'; r), mta/mod.rs:133-145
    nonces = dict(alpha=rand_words(g, dev, B, 24, 23), beta=rand_words(g, dev, B, 64, 63), gamma=rand_words(g, dev, B, 80, 79),
                  rho=rand_words(g, dev, B, 72, 71), rho_prim=rand_words(g, dev, B, 88, 87), sigma=rand_words(g, dev, B, 72, 71),
                  tau=rand_words(g, dev, B, 88, 87))
    X = E.ec_mul_base(ctx, b)
    sm = lambda d_, n_: {f: v[:n_] for f, v in d_.items()}
    pr0, u0 = E.bob_generate(ctx, pub, stm, a_enc[:256], mta[:256], b[:256], beta_prim[:256], r[:256], sm(nonces, 256), True, kidx[:256], sidx[:256])
    E.bob_verify(ctx, pub, stm, a_enc[:256], mta[:256], pr0, X[:256], u0, kidx[:256], sidx[:256])            # warm-up

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out
    ctx.prof_enable(True)
    t_gen, (pr, u) = timed(lambda: E.bob_generate(ctx, pub, stm, a_enc, mta, b, beta_prim, r, nonces, True, kidx, sidx))
    rec_gen = ctx.prof_collect(4096)
    ctx.prof_enable(True)
    t_ver, ok = timed(lambda: E.bob_verify(ctx, pub, stm, a_enc, mta, pr, X, u, kidx, sidx))
    rec_ver = ctx.prof_collect(4096)
    ctx.prof_enable(False)
    accepted = int(ok.sum())
    bad = {k_: v.clone() for k_, v in pr.items()}
    fields = ["t", "z", "e", "s", "s1", "s2", "t1", "t2"]
    mask = np.zeros(B, dtype=bool)
    for q_, f in enumerate(fields):                                                   # 1 %: every 100th proof, a different field each time
        bad[f][q_ * 100 + 3:B:800, 1] ^= 0x20
        mask[q_ * 100 + 3:B:800] = True
    okb = E.bob_verify(ctx, pub, stm, a_enc, mta, bad, X, u, kidx, sidx).cpu().numpy()
    out = {"instances": B, "keys": K, "check": True, "bob_generate_per_s": B / t_gen, "bob_verify_ext_per_s": B / t_ver, "accepted": accepted,
           "accept_rate": accepted / B, "corrupted": int(mask.sum()), "corrupted_1pct_all_rejected": bool((okb[mask] == 0).all() and (okb[~mask] == 1).all()),
           "generate_whole_step": whole_step(rec_gen, t_gen), "verify_whole_step": whole_step(rec_ver, t_ver)}
    if oracle:
        n_ = min(prefix, B)
        threads = threads or min(host_cores()[0], 64)
        h = lambda t_: np.ascontiguousarray(t_[:n_].cpu().numpy().view(np.uint32))
        tabs = dict(N=F.words([k.N for k in keys], 64), Nt=F.words([k.Nt for k in keys], 64), h1=F.words([k.h1 for k in keys], 64),
                    h2=F.words([k.h2 for k in keys], 64))
        hin = dict(a_enc=h(a_enc), mta=h(mta), b=h(b), bp=h(beta_prim), r=h(r), X=h(X), **{f: h(v) for f, v in nonces.items()})
        ki, si = kidx[:n_].cpu().numpy(), sidx[:n_].cpu().numpy()
        got, gu = {f: h(v) for f, v in pr.items()}, h(u)
        chunks = [c_ for c_ in np.array_split(np.arange(n_), threads) if len(c_)]

        def run(ix):
            sl = slice(int(ix[0]), int(ix[-1]) + 1)
            want, wu = orc.bob_generate(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], ki[sl], si[sl], hin["a_enc"][sl], hin["mta"][sl], hin["b"][sl],
                                        hin["bp"][sl], hin["r"][sl], hin["alpha"][sl], hin["beta"][sl], hin["gamma"][sl], hin["rho"][sl],
                                        hin["rho_prim"][sl], hin["sigma"][sl], hin["tau"][sl], True)
            good = all(np.array_equal(got[f][sl], want[f]) for f in want) and np.array_equal(gu[sl], wu)
            okv = orc.bob_verify(tabs["N"], tabs["Nt"], tabs["h1"], tabs["h2"], ki[sl], si[sl], hin["a_enc"][sl], hin["mta"][sl], want, hin["X"][sl], wu)
            return good and bool(np.all(okv))
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(run, chunks))
        out.update({"parity_prefix": n_, "parity_vs_oracle_on_prefix": bool(all(res)), "oracle_generate_verify_per_s": n_ / (time.time() - t0),
                    "oracle_threads": threads})
    return out


def gg20_config(ctx, E, G, keys, t, n, B, steps, gen, parity_sample=0, threads=None, openssl=False):
    """one more GG20 shape on this GPU: sessions/s over `steps` passes of B sessions, every pass on values sampled on the device inside
    the timed region (a fresh batch counter per pass); optional parity sample vs the oracle, which expands the last pass's seed itself"""
    dev = ctx.device
    signers = list(range(t + 1))
    lk = G.make_local_keys(keys, t, n, signers)
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
    seed = hashlib.sha256(b"bench.py gg20_config %d %d %d" % (t, n, B)).digest()
    msg = rand_words(gen, dev, B, 8, 8)
    nonces, _ = E.gg20_sample_nonces(ctx, gk, B, seed, 0, msg=msg)
    out = E.gg20_sign(ctx, gk, nonces, B)
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for c_ in range(steps):
        _, fail = E.gg20_sample_nonces(ctx, gk, B, seed, 1 + c_, out=nonces)
        out = E.gg20_sign(ctx, gk, nonces, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    recs = ctx.prof_collect(16384)
    ctx.prof_enable(False)
    r, s, recid, status = [o.cpu().numpy() for o in out]
    res = {"sessions": B, "t": t, "n": n, "signatures_per_s": B / dt, "ms_per_batch": dt * 1e3, "all_sessions_signed": bool((status == 0).all()),
           "nonces": "sampled on the device inside the timed region, a fresh batch counter per pass", "sampler_rejection_loops_given_up": int(fail.item()),
           "whole_step": whole_step(recs, dt * steps), "roofline_dominant": dominant_roofline(recs, dt * steps),
           "roofline_secondary": secondary_rooflines(recs, dt * steps)}
    if parity_sample:
        threads = threads or min(host_cores()[0], 64)
        hm = np.ascontiguousarray(msg[:parity_sample].cpu().numpy().view(np.uint32))
        hn, wf = G.oracle_sample_nonces(lk, parity_sample, seed, steps, msg=hm)
        v, wr, ws, wrecid, wstatus = cpu_baseline_gg20(lk, hn, parity_sample, min(threads, parity_sample))
        res["parity_sample"] = parity_sample
        res["parity_vs_oracle_on_sample"] = bool(wf == 0 and (wstatus == 0).all() and np.array_equal(r[:parity_sample].view(np.uint32), wr) and
                                                 np.array_equal(s[:parity_sample].view(np.uint32), ws) and np.array_equal(recid[:parity_sample], wrecid))
        res["oracle_signatures_per_s"] = v
    if openssl:
        res["openssl"] = openssl_verify_all(lk["arrays"]["y"][0], msg, r, s, threads or min(host_cores()[0], 64))
    gk.close()
    return res


def c4_pipeline(ctx, E, G, keys, batches=96, B=1024, lanes=2, group=4, parity_sample=16, threads=None, oracle=True, verify=True, window=None):
    """BASELINE config 4 as a SERVICE sees it: a stream of `batches` successive 1 024-session (t=1, n=3) batches through the pipelined
    engine (mpe_gg20_pipeline_*, csrc/mpe_pipeline.h): `group` batches coalesced per lock-step pass, `lanes` passes in flight on one
    stream each, inputs staged on one more stream — ONE context handle, ONE host thread, the runtime's default hardware queues, no child
    process.  Every batch is signed from values sampled ON THE DEVICE from (seed, batch counter) (submit_seeded).  The client is a
    closed loop: at most 2 x lanes x group batches outstanding (one group running and one queued per lane), a new batch goes in when
    the oldest completes — so `latency_ms` is what a caller of a loaded service sees, not the depth of a flooded queue.
    Afterwards every signature is checked under OpenSSL's ECDSA_do_verify and the first sessions of EVERY batch against the GMP oracle,
    which re-expands the batch's (seed, counter) itself (oracle/sampler_oracle.c).  The reference runs many OfflineStage instances side
    by side the same way (state_machine/sign.rs:667-691; rounds.rs:106,215,323 `is_expensive`)."""
    t, n, signers = 1, 3, [0, 1]
    lk = G.make_local_keys(keys, t, n, signers)
    dev = ctx.device
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
    pipe = E.Gg20Pipeline(ctx, gk, B, group=group, lanes=lanes)
    seed = hashlib.sha256(b"bench.py c4_stream_1024 lanes %d group %d" % (lanes, group)).digest()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1024)
    msgs = [rand_words(gen, dev, B, 8, 8) for _ in range(batches)]
    window = window or 2 * lanes * group
    # warm-up: both staging buffers of every lane, every allocation of the lanes' workspaces
    warm = [pipe.submit_seeded(seed, (1 << 40) + i, msgs[i % batches]) for i in range(window)]
    pipe.flush()
    for tk in warm:
        pipe.wait(tk)
    torch.cuda.synchronize()
    results, tickets = [None] * batches, [None] * batches
    t0 = time.perf_counter()
    head = 0
    for b in range(batches):
        if b - head >= window:                       # closed loop: the oldest outstanding batch completes before the next goes in
            results[head] = pipe.wait(tickets[head])
            head += 1
        tickets[b] = pipe.submit_seeded(seed, b, msgs[b])
    pipe.flush()
    while head < batches:
        results[head] = pipe.wait(tickets[head])
        head += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lat = sorted(pipe.latency_ms(tk) for tk in tickets)
    pas = sorted(pipe.pass_ms(tk) for tk in tickets)
    res = {"batches": batches, "sessions_per_batch": B, "lanes": lanes, "batches_per_pass": group, "in_flight_bound": window,
           "streams": lanes + 1, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES") or "runtime default", "host_threads": 1, "child_process": False,
           "signatures_per_s": batches * B / dt, "seconds": dt, "ms_per_batch_sustained": dt / batches * 1e3,
           "latency_ms": {"p50": lat[len(lat) // 2], "p95": lat[int(len(lat) * 0.95)], "max": lat[-1],
                          "what": "device time from the submit call of a batch to its results, closed loop with the bound above"},
           "pass_ms": {"p50": pas[len(pas) // 2], "max": pas[-1], "what": f"one lock-step pass over {group} x {B} sessions"},
           "nonces": "sampled on the device per batch from (seed, batch counter): no value is used twice",
           "sampler_rejection_loops_given_up": pipe.sampler_failures(),
           "what": f"{batches} successive {B}-session t=1 n=3 batches through mpe_gg20_pipeline: {group} batches per pass, {lanes} passes in flight"}
    threads = threads or min(host_cores()[0], 64)
    signed, verified, parity = True, 0, True
    for b in range(batches):
        r, s_, recid, status = [o.cpu().numpy() for o in results[b]]
        signed = signed and bool((status == 0).all())
        if verify:
            verified += openssl_verify_all(lk["arrays"]["y"][0], msgs[b], r, s_, threads)["openssl_verified"]
        if oracle and parity_sample:
            k = min(parity_sample, B)
            hm = np.ascontiguousarray(msgs[b][:k].cpu().numpy().view(np.uint32))
            z, wf = G.oracle_sample_nonces(lk, k, seed, b, msg=hm)
            _, wr, ws, wrecid, wstatus = cpu_baseline_gg20(lk, z, k, min(threads, k))
            parity = parity and bool(wf == 0 and (wstatus == 0).all() and np.array_equal(r[:k].view(np.uint32), wr) and
                                     np.array_equal(s_[:k].view(np.uint32), ws) and np.array_equal(recid[:k], wrecid))
    res.update(all_sessions_signed=signed)
    if verify:
        res.update(openssl_verified=verified, openssl_of=batches * B)
    if oracle and parity_sample:
        res.update(parity_vs_oracle_on_sample=parity, parity_sample_per_batch=min(parity_sample, B),
                   parity_what="the oracle expands each batch's (seed, counter) itself and signs: (r, s, recid) bit for bit")
    pipe.close()
    gk.close()
    return res


def c4_open_loop(ctx, E, G, keys, capacity_sig_s, loads=(0.5, 0.9), batches=64, B=1024, lanes=2, group=4, deadline_us=250000, threads=None):
    """The stream of 1 024-session batches as an OPEN loop: batches arrive as a Poisson process at `load` x the closed-loop capacity,
    whether or not the service keeps up (a closed loop hides queueing: its client waits).  The pipeline groups by ARRIVAL
    (mpe_gg20_pipeline_set_eager: a part-filled group goes as soon as its lane is idle; set_deadline_us: or when its oldest batch has
    waited that long), driven by this one host thread's poll() — at a low load a batch starts at once and costs a lone batch's latency,
    near capacity the lanes are busy and the groups fill by themselves.  Latency = host time from a batch's ARRIVAL to the moment
    its completion is seen (the loop polls every ~50 us).  Every batch signed from device-sampled values; all of them under OpenSSL."""
    import random
    t, n, signers = 1, 3, [0, 1]
    lk = G.make_local_keys(keys, t, n, signers)
    dev = ctx.device
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(2048)
    msgs = [rand_words(gen, dev, B, 8, 8) for _ in range(batches)]
    out = {"capacity_signatures_per_s": capacity_sig_s, "sessions_per_batch": B, "lanes": lanes, "max_batches_per_pass": group,
           "deadline_us": deadline_us, "eager": True, "arrivals": "Poisson (exponential gaps, seed 7), open loop", "runs": []}
    threads = threads or min(host_cores()[0], 64)
    for load in loads:
        pipe = E.Gg20Pipeline(ctx, gk, B, group=group, lanes=lanes)
        seed = hashlib.sha256(b"bench.py c4 open loop %d" % int(load * 100)).digest()
        warm = [pipe.submit_seeded(seed, (1 << 40) + i, msgs[i % batches]) for i in range(lanes * group)]
        pipe.flush()
        for tk in warm:
            pipe.wait(tk)
        torch.cuda.synchronize()
        pipe.set_eager(True)
        pipe.set_deadline_us(deadline_us)
        rng = random.Random(7)
        lam = load * capacity_sig_s / B                      # batches per second
        arrive, acc = [], 0.0
        for _ in range(batches):
            acc += rng.expovariate(lam)
            arrive.append(acc)
        tickets, done_at, pending = [None] * batches, [None] * batches, []
        nxt = 0
        t0 = time.perf_counter()
        while nxt < batches or pending:
            now = time.perf_counter() - t0
            if nxt < batches and now >= arrive[nxt]:
                tickets[nxt] = pipe.submit_seeded(seed, nxt, msgs[nxt])
                pending.append(nxt)
                nxt += 1
                continue
            pipe.poll()
            still = []
            for b in pending:
                if pipe.done(tickets[b]):
                    done_at[b] = time.perf_counter() - t0
                else:
                    still.append(b)
            pending = still
            if nxt >= batches and pending and not pipe.ticket_rc(tickets[pending[-1]])[0]:
                pipe.flush()                                 # the stream has ended: nothing more will fill the last group
            time.sleep(0.00005)
        total = max(done_at)
        lat = sorted((done_at[b] - arrive[b]) * 1e3 for b in range(batches))
        signed, ver = True, 0
        for b in range(batches):
            r, s_, recid, status = [o.cpu().numpy() for o in pipe.wait(tickets[b])]
            signed = signed and bool((status == 0).all())
            ver += openssl_verify_all(lk["arrays"]["y"][0], msgs[b], r, s_, threads)["openssl_verified"]
        c = pipe.counters()
        out["runs"].append({"load": load, "offered_signatures_per_s": lam * B, "achieved_signatures_per_s": batches * B / total,
                            "latency_ms": {"p50": lat[len(lat) // 2], "p90": lat[int(len(lat) * 0.9)], "p99": lat[min(len(lat) - 1, int(len(lat) * 0.99))], "max": lat[-1]},
                            "passes": c["groups"], "mean_batches_per_pass": batches / max(1, c["groups"]), "passes_started_by_an_idle_lane": c["by_idle"],
                            "passes_started_by_the_deadline": c["by_deadline"], "all_sessions_signed": signed, "openssl_verified": ver,
                            "openssl_of": batches * B})
        pipe.close()
    gk.close()
    return out


def multi_wallet(ctx, E, G, keys, K, B, gen, t=1, n=3):
    """One batch whose sessions belong to K different wallets (key sets), round-robin — SURVEY.md 8d config 4 allows "16
    fixtures round-robin".  The 16 Paillier / N~ fixtures are reused cyclically for the K * n key slots (the per-key state
    and the fixed-base tables are sized and addressed for K * n distinct keys); every wallet has its own Shamir shares."""
    dev = ctx.device
    signers = list(range(t + 1))
    S = len(signers)
    t0 = time.perf_counter()
    lks = [G.make_local_keys(keys[(kk * n) % len(keys):] + keys[:(kk * n) % len(keys)], t, n, signers, seed=f"wallet-{kk}") for kk in range(min(K, 16))]
    arrays = {f: np.concatenate([lks[kk % len(lks)]["arrays"][f] for kk in range(K)]) for f in ("x", "p", "q", "Nt", "h1", "h2", "y", "X")}
    gk = E.Gg20Keys(ctx, t, n, signers, arrays, nkeysets=K)
    torch.cuda.synchronize()
    t_keys = time.perf_counter() - t0
    keyset = (torch.arange(B, device=dev, dtype=torch.int32) % K).contiguous()
    seed = hashlib.sha256(b"bench.py multi_wallet %d %d" % (K, B)).digest()
    msg = rand_words(gen, dev, B, 8, 8)
    nonces, _ = E.gg20_sample_nonces(ctx, gk, B, seed, 0, msg=msg, keyset=keyset)           # warm-up pass on its own values
    out = E.gg20_sign(ctx, gk, nonces, B, keyset=keyset)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, fail = E.gg20_sample_nonces(ctx, gk, B, seed, 1, out=nonces, keyset=keyset)          # the timed pass samples on the device, fresh counter
    out = E.gg20_sign(ctx, gk, nonces, B, keyset=keyset)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    status = out[3].cpu().numpy()
    res = {"wallets": K, "sessions": B, "signatures_per_s": B / dt, "fb_window_bits": gk.fb_window_bits(), "key_setup_s": t_keys,
           "all_sessions_signed": bool((status == 0).all()), "sampler_rejection_loops_given_up": int(fail.item()),
           "nonces": "sampled on the device inside the timed pass, a fresh batch counter"}
    gk.close()
    return res


def mint_distinct_moduli(count, threads, seed=b"bench.py distinct wallets"):
    """`count` pairwise distinct 2048-bit Paillier moduli N = p q and as many distinct (N~, h1, h2), derived from a POOL of fresh 1 024-bit
    primes (mpz_nextprime from a SHA-256 stream, the way tests/golden/make_keys.py mints the fixtures) taken in pairs: P primes give
    P (P - 1) / 2 distinct products, so 320 primes serve 51 040 moduli in ~20 s where 2 x 49 152 primes of their own would take half an
    hour of host time.  Distinct as NUMBERS — which is what the device's per-key state (Montgomery / pair constants, CRT idempotents,
    window tables, key-ordered launches) depends on; two moduli may share a prime, which no kernel can tell.  h1 = r^2, h2 = h1^alpha
    (alpha of 64 bits: signing never uses the relation, only keygen's proof does).  Returns word arrays p, q [count][32], Nt, h1, h2 [count][64]."""
    import concurrent.futures as cf
    import orc
    import fixtures as F
    P = 2
    while P * (P - 1) // 2 < count:
        P += 1
    P += P % 2

    def prime(i):
        start = int.from_bytes(hashlib.sha512(seed + b"|%d" % i).digest() * 2, "big") | (1 << 1023) | (1 << 1022)
        out = np.zeros((1, 32), dtype=np.uint32)
        orc.lib.orc_nextprime(32, orc._p(F.words([start], 32)), orc._p(out))
        return F.ints(out)[0]
    orc.lib.orc_nextprime.restype = None
    with cf.ThreadPoolExecutor(max(1, threads)) as ex:
        pool = list(ex.map(prime, range(2 * P)))
    pa, pb = pool[:P], pool[P:]                                  # Paillier primes | N~ primes
    pairs = [(i, j) for i in range(P) for j in range(i + 1, P)][:count]
    rs = F.Rng("distinct wallets h1")
    ps, qs, nts, h1s, h2s = [], [], [], [], []
    for (i, j) in pairs:
        ps.append(pa[i]); qs.append(pa[j])
        nt = pb[i] * pb[j]
        h1 = pow(rs.below(nt - 2) + 2, 2, nt)
        nts.append(nt); h1s.append(h1); h2s.append(pow(h1, rs.bits(64) | 1, nt))
    assert len({a * b for a, b in zip(ps, qs)}) == count and len(set(nts)) == count
    return F.words(ps, 32), F.words(qs, 32), F.words(nts, 64), F.words(h1s, 64), F.words(h2s, 64), 2 * P


def multi_wallet_distinct(ctx, E, G, keys, K, B, gen, threads, t=1, n=3, parity_sample=16, cache=None):
    """EVERY session its own wallet (the reference takes an arbitrary LocalKey per OfflineStage, state_machine/sign.rs:78;
    keygen/rounds.rs:311-322): B sessions over K wallets whose K n Paillier moduli and K n (N~, h1, h2) are pairwise DISTINCT
    (mint_distinct_moduli); the Shamir material (x_i, X_i, y) cycles through 16 wallets.  Values sampled on the device inside the timed
    pass; an oracle parity sample and OpenSSL on every signature afterwards."""
    dev = ctx.device
    signers = list(range(t + 1))
    t0 = time.perf_counter()
    if cache is not None and cache.get("count", 0) >= K * n:
        pw, qw, ntw, h1w, h2w, nprimes = cache["arrays"]
    else:
        pw, qw, ntw, h1w, h2w, nprimes = mint_distinct_moduli(K * n, threads)
        if cache is not None:
            cache.update(count=K * n, arrays=(pw, qw, ntw, h1w, h2w, nprimes))
    t_mint = time.perf_counter() - t0
    lks = [G.make_local_keys(keys[(kk * n) % len(keys):] + keys[:(kk * n) % len(keys)], t, n, signers, seed=f"wallet-{kk}") for kk in range(16)]
    arrays = {f: np.concatenate([lks[kk % 16]["arrays"][f] for kk in range(K)]) for f in ("x", "y", "X")}
    arrays.update(p=np.ascontiguousarray(pw[:K * n]), q=np.ascontiguousarray(qw[:K * n]), Nt=np.ascontiguousarray(ntw[:K * n]),
                  h1=np.ascontiguousarray(h1w[:K * n]), h2=np.ascontiguousarray(h2w[:K * n]), signers=lks[0]["arrays"]["signers"])
    lk = dict(t=t, n=n, S=len(signers), arrays=arrays, nkeysets=K)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    gk = E.Gg20Keys(ctx, t, n, signers, arrays, nkeysets=K)
    torch.cuda.synchronize()
    t_keys = time.perf_counter() - t0
    key_bytes = free0 - torch.cuda.mem_get_info()[0]
    keyset_h = (np.arange(B, dtype=np.int32) % K).astype(np.int32)
    keyset = torch.from_numpy(keyset_h).to(dev)
    seed = hashlib.sha256(b"bench.py multi_wallet_distinct %d %d" % (K, B)).digest()
    msg = rand_words(gen, dev, B, 8, 8)
    nonces, _ = E.gg20_sample_nonces(ctx, gk, B, seed, 0, msg=msg, keyset=keyset)
    out = E.gg20_sign(ctx, gk, nonces, B, keyset=keyset)
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    _, fail = E.gg20_sample_nonces(ctx, gk, B, seed, 1, out=nonces, keyset=keyset)
    out = E.gg20_sign(ctx, gk, nonces, B, keyset=keyset)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = ctx.prof_collect(16384)
    ctx.prof_enable(False)
    r, s_, recid, status = [o.cpu().numpy() for o in out]
    by_kind = {}
    for x in recs:
        k_ = (x["kind"], x["bits"])
        by_kind.setdefault(k_, [0, 0.0])
        by_kind[k_][0] += 1
        by_kind[k_][1] += x["ms"]
    slide = [slid(x) for x in recs if x["kind"] == 6]
    res = {"wallets": K, "sessions": B, "distinct_paillier_moduli": K * n, "distinct_ntilde": K * n, "primes_minted": nprimes,
           "moduli_from": "pairwise products of a pool of fresh 1024-bit primes (mpz_nextprime): distinct numbers, primes shared between pairs",
           "signatures_per_s": B / dt, "ms_per_batch": dt * 1e3, "fb_window_bits": gk.fb_window_bits(), "key_setup_s": t_keys,
           "key_setup_s_amortised_per_signature": t_keys / B, "key_object_device_bytes": int(key_bytes), "host_mint_s": t_mint,
           "all_sessions_signed": bool((status == 0).all()), "sampler_rejection_loops_given_up": int(fail.item()),
           "nonces": "sampled on the device inside the timed pass (mpe_gg20_sample_nonces with the sessions' key sets)",
           "share_of_sliding_ladder_waves_that_slide": (sum(slide) / len(slide)) if slide else None,
           "whole_step": whole_step(recs, dt),
           "launch_table": [{"kind": k_[0], "bits": k_[1], "launches": v[0], "ms": round(v[1], 2), "share": round(v[1] * 1e-3 / dt, 4)}
                            for k_, v in sorted(by_kind.items(), key=lambda kv: -kv[1][1])]}
    if parity_sample:
        k = min(parity_sample, B)
        hm = np.ascontiguousarray(msg[:k].cpu().numpy().view(np.uint32))
        z, wf = G.oracle_sample_nonces(lk, k, seed, 1, keyset=keyset_h[:k], msg=hm)
        w = G.oracle_sign_ex(lk, z, k, keyset=keyset_h[:k])
        res["parity_sample"] = k
        res["parity_vs_oracle_on_sample"] = bool(wf == 0 and (w["status"] == 0).all() and np.array_equal(r[:k].view(np.uint32), w["r"]) and
                                                 np.array_equal(s_[:k].view(np.uint32), w["s"]) and np.array_equal(recid[:k], w["recid"]))
    ver = 0
    for wl in range(16):                                   # the public key y cycles through 16 Shamir wallets
        sel = np.nonzero(keyset_h % 16 == wl)[0]
        if len(sel):
            ver += openssl_verify_all(lks[wl]["arrays"]["y"][0], msg.cpu().numpy()[sel], r[sel], s_[sel], threads)["openssl_verified"]
    res.update(openssl_verified=ver, openssl_of=B)
    gk.close()
    return res


def lindell_section(ctx, E, keys, F, cpu=True):
    """Lindell'17 two-party signing (PartialSig::compute + Signature::compute_with_recid), 65 536
    independent sessions over 16 Paillier keys; the first sessions are checked against the oracle bit for bit."""
    import lindell_fixture as L
    B, dev = 65536, ctx.device
    small = 128
    fx = L.make(keys, small, seed="bench-lindell")
    rep_ = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (B // small, 1))).view(np.int32)).to(dev)
    d = {k: rep_(fx[k]) for k in ("c_key", "x2", "k1", "k2", "R1", "R2", "msg", "rho", "r")}
    kidx = torch.tensor(fx["kidx"] * (B // small), dtype=torch.int32, device=dev)
    sk = E.PaillierKeys(ctx, p=[k.p for k in keys], q=[k.q for k in keys])
    pk = E.PaillierKeys(ctx, N=[k.N for k in keys])

    def run():
        c3 = E.lindell_partial_sig(ctx, pk, d["c_key"], d["x2"], d["k2"], d["R1"], d["msg"], d["rho"], d["r"], kidx)
        return E.lindell_sign(ctx, sk, c3, d["k1"], d["R2"], kidx)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r, s_, recid = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"signatures_per_s": B / dt, "batch": B}
    if cpu:
        t0 = time.perf_counter()
        _, wr, ws, wrec = L.oracle_run(fx)
        out["cpu_oracle_signatures_per_s_1_thread"] = small / (time.perf_counter() - t0)
        out["parity_vs_oracle_on_sample"] = bool(np.array_equal(r[:small].cpu().numpy().view(np.uint32), wr) and
                                                 np.array_equal(s_[:small].cpu().numpy().view(np.uint32), ws) and
                                                 list(recid[:small].cpu().numpy()) == list(wrec))
    return out


def blame_section(ctx, E, G, keys, F, gen, B=4096, cpu=True, sample=16):
    """identifiable abort (gg_2020/blame.rs) at scale.  B sessions (t=1, n=3, two signers) run on the round
    engine with signer 0 corrupted the way the reference's own tests corrupt it (`mpe_gg20_session_fault_inject`: delta_i, sigma_i
    or s_i doubled, gg_2020/test.rs:282-289,458-465,679-686); the failing check is the reference's (502 / 602 / 701) in every
    session; then the openings every signer publishes go through `mpe_gg20_blame5/6/7`, which must name exactly signer 0 in every
    session.  Timed: the blame calls alone (the openings are protocol traffic).  Phase-6 openings come from the device itself
    (`mpe_gg20_session_blame6_state`, `mpe_paillier_open`), as a party would produce them.  Oracle: the first `sample` sessions."""
    import pyref
    dev = ctx.device
    t, n, signers, S, P1, mask = 1, 3, [0, 1], 2, 1, 0b01
    lk = G.make_local_keys(keys, t, n, signers)
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"])
    dn = make_device_nonces(gen, dev, B, S, S, n)
    hn = _host(dn)
    to_dev = lambda d: {f: torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).to(dev) for f, v in d.items()}
    head = lambda d, cnt: {f: np.ascontiguousarray(v[: cnt * (v.shape[0] // B)]) for f, v in d.items()}
    out = {"sessions": B, "t": t, "n": n, "corrupted_signer": 0}

    def run_faulty(step):
        sess = E.Gg20Session(ctx, gk, B, list(range(S)), dn)
        sess.fault_inject(step, mask)
        slabs, prev = {}, None
        for rnd in range(9):
            o = sess.round(rnd, d_in=prev, msg=dn["msg"] if rnd == 7 else None)
            if o is not None:
                slabs[rnd] = o.cpu().numpy().view(np.uint32)
                prev = o.reshape(-1)
        res = sess.result()
        torch.cuda.synchronize()
        return sess, slabs, res

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, got.cpu().numpy().view(np.uint32)

    def oracle(which, o):
        if not cpu:
            return {}
        t0 = time.perf_counter()
        want = G.oracle_blame(lk, which, head(o, sample), sample)
        return {"oracle_sessions_per_s_1_thread": sample / (time.perf_counter() - t0), "want": want}

    tr = lambda a: np.ascontiguousarray(np.transpose(a, (1, 0, 2)).reshape(B * S, a.shape[2]))          # [S][B][w] -> [B][S] rows
    # phase 5 (blame.rs:116-224)
    sess, slabs, res = run_faulty(5)
    st5 = res["status"].cpu().numpy()
    o5 = G.blame5_opened(lk, hn, slabs, B)
    d5 = to_dev(o5)
    dt, got = timed(lambda: E.gg20_blame5(ctx, gk, B, d5))
    orc5 = oracle("b5", o5)
    out["blame5"] = {"sessions_per_s": B / dt, "failing_check_is_502_everywhere": bool((st5 == 502).all()), "names_exactly_the_corrupted_signer": bool((got == mask).all())}
    sess.close()
    # phase 6 (blame.rs:322-421)
    sess, slabs, res = run_faulty(6)
    st6 = res["status"].cpu().numpy()
    en = _scalar(gen, dev, B * S)
    miu, a1, a2, z = sess.blame6_state(en)
    dtr = lambda x: x.transpose(0, 1).reshape(B * S, -1).contiguous()
    cb = G.blame6_cb(lk, slabs, B)
    d_cb = torch.from_numpy(cb.view(np.int32)).to(dev)
    sk_all = E.PaillierKeys(ctx, p=[k.p for k in lk["keys"]], q=[k.q for k in lk["keys"]])
    kx = torch.tensor([signers[(r_ // P1) % S] for r_ in range(B * S * P1)], dtype=torch.int32, device=dev)
    om, orr = E.paillier_open(ctx, sk_all, d_cb, kx)
    torch.cuda.synchronize()
    opens_agree = bool(torch.equal(om.reshape(-1, 64), miu.transpose(0, 1).reshape(-1, 64)))
    d6 = dict(k=dn["k"], k_rand=dn["r_a"], miu=om, miu_rand=orr, a1=dtr(a1), a2=dtr(a2), z=dtr(z),
              S=torch.from_numpy(tr(slabs[5][:, :, 0:16]).view(np.int32)).to(dev),
              c_a=torch.from_numpy(tr(slabs[0][:, :, n * 256:n * 256 + 128]).view(np.int32)).to(dev), c_b=d_cb, R=res["R"][0].contiguous())
    dt, got = timed(lambda: E.gg20_blame6(ctx, gk, B, d6))
    o6 = {f: v.cpu().numpy().view(np.uint32) for f, v in d6.items()}
    orc6 = oracle("b6", o6)
    out["blame6"] = {"sessions_per_s": B / dt, "failing_check_is_602_everywhere": bool((st6 == 602).all()), "names_exactly_the_corrupted_signer": bool((got == mask).all()),
                     "paillier_open_equals_the_sessions_own_miu": opens_agree}
    got6 = got
    sess.close()
    # phase 7 (blame.rs:434-454)
    sess, slabs, res = run_faulty(7)
    st7 = res["status"].cpu().numpy()
    Rr = res["R"][0].cpu().numpy().view(np.uint32)
    o7 = dict(s=tr(slabs[7]), r=F.words([x % pyref.Q for x in F.ints(np.ascontiguousarray(Rr[:, :8]))], 8), R_dash=tr(slabs[4][:, :, 450 * (S - 1):450 * (S - 1) + 16]),
              m=hn["msg"].copy(), R=Rr, S=tr(slabs[5][:, :, 0:16]))
    d7 = to_dev(o7)
    dt, got7 = timed(lambda: E.gg20_blame7(ctx, S, B, d7))
    orc7 = oracle("b7", o7)
    out["blame7"] = {"sessions_per_s": B / dt, "failing_check_is_701_everywhere": bool((st7 == 701).all()), "names_exactly_the_corrupted_signer": bool((got7 == mask).all())}
    sess.close()
    if cpu:
        got5 = E.gg20_blame5(ctx, gk, B, d5).cpu().numpy().view(np.uint32)
        for name, o_, g_ in (("blame5", orc5, got5), ("blame6", orc6, got6), ("blame7", orc7, got7)):
            out[name]["oracle_sessions_per_s_1_thread"] = o_["oracle_sessions_per_s_1_thread"]
            out[name]["parity_vs_oracle_on_sample"] = bool(list(o_["want"]) == list(g_[:sample]))
        out["oracle_sample"] = sample
    sk_all.close()
    gk.close()
    return out


def keygen_verify_section(ctx, E, keys, F, B=8192, cpu=True):
    """what every party checks about every other party's keygen messages (party_i.rs:260-438), batched over
    (verifier, prover) pairs: `NiCorrectKeyProof::verify` (11 modular exponentiations with a 2048-bit exponent per proof),
    `CompositeDLogProof::verify`, Feldman `validate_share`.  16 provers' real proofs (the oracle's prove side) tiled to B items, 1 %
    of them corrupted: those and only those are refused.  Oracle: the 16 distinct items on one thread."""
    import keygen_fixture as KF
    import orc
    dev = ctx.device
    K, reps = len(keys), B // len(keys)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    bad_rows = np.arange(7, B, 100)
    out = {"items": B, "corrupted": int(len(bad_rows))}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ok.cpu().numpy()

    def verdict(ok):
        want = np.ones(B, dtype=np.uint8)
        want[bad_rows] = 0
        return bool(np.array_equal(ok.astype(np.uint8), want))

    N, sigma = KF.correct_key_case(keys)
    Nb, sb = np.tile(N, (reps, 1)), np.tile(sigma.reshape(K, 11 * 64), (reps, 1))
    sb[bad_rows, 64 * 3 + 5] ^= 2
    dN, ds = up(Nb), up(sb.reshape(-1, 64))
    dt, ok = timed(lambda: E.correct_key_verify(ctx, dN, ds))
    out["correct_key_verify_per_s"] = B / dt
    out["correct_key_exactly_the_corrupted_refused"] = verdict(ok)
    Nw, gw, nw, x, y = KF.composite_dlog_case(keys)
    tiles = [np.tile(a, (reps, 1)) for a in (Nw, gw, nw, x, y)]
    tiles[4][bad_rows, 70] ^= 1
    dc = [up(a) for a in tiles]
    dt, ok = timed(lambda: E.composite_dlog_verify(ctx, *dc))
    out["composite_dlog_verify_per_s"] = B / dt
    out["composite_dlog_note"] = "every item brings its own modulus: its Montgomery constants and the inversion behind the gcd checks are part of the call"
    out["composite_dlog_exactly_the_corrupted_refused"] = verdict(ok)
    t_, n_, dealers = 2, 5, 16
    commits, shares, index, _ = KF.vss_case(t_, n_, dealers, seed="bench-vss")
    rv = B // (dealers * n_) + 1
    cm, sh, ix = np.tile(commits, (rv, 1))[:B], np.tile(shares, (rv, 1))[:B], np.tile(index, rv)[:B]
    sh = sh.copy()
    sh[bad_rows, 1] ^= 1
    dv = (up(cm), up(sh), torch.from_numpy(np.ascontiguousarray(ix, dtype=np.int32)).to(dev))
    dt, ok = timed(lambda: E.vss_validate_share(ctx, t_ + 1, *dv))
    out["vss_validate_share_per_s"] = B / dt
    out["vss_exactly_the_corrupted_refused"] = verdict(ok)
    if cpu:
        w = np.zeros(K, dtype=np.uint8)
        t0 = time.perf_counter()
        orc.lib.orc_correct_key_verify(K, orc._p(N), orc._p(sigma), orc._p(w))
        out["oracle_correct_key_verify_per_s_1_thread"] = K / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        orc.lib.orc_composite_dlog_verify(K, *[orc._p(a) for a in (Nw, gw, nw, x, y, w)])
        out["oracle_composite_dlog_verify_per_s_1_thread"] = K / (time.perf_counter() - t0)
        wv = np.zeros(dealers * n_, dtype=np.uint8)
        t0 = time.perf_counter()
        orc.lib.orc_vss_validate_share(dealers * n_, t_ + 1, orc._p(commits), orc._p(shares), orc._p(index), orc._p(wv))
        out["oracle_vss_validate_share_per_s_1_thread"] = dealers * n_ / (time.perf_counter() - t0)
        out["oracle_accepts_the_uncorrupted_items"] = bool(w.all() and wv.all())
    return out


def make_comm(ctx, E, rank, world, share, distributed):
    """the RCCL communicator of the round fan-out behind the C-ABI (mpe_comm_create): rank 0's id travels through the process group
    that the launcher set up; ranks that share one device (--share-device, gloo) cannot use RCCL and keep the host-staged gather"""
    if share:
        return None
    if not distributed:
        return E.Comm(ctx, 0, 1)
    import torch.distributed as dist

    def exchange(ident):
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    return E.Comm(ctx, rank, world, exchange_id=exchange)


class GpuRoundEngine:
    """dist.PartySharded engine over mpe_gg20_roundN: one session object per session block, local = the parties this rank
    hosts; the object lives across steps (mpe_gg20_session_rearm) and writes its records into the gather buffer.  Every step of
    the timed region re-arms it with values sampled ON THE DEVICE for exactly its local parties from the engine's own seed and a
    fresh batch counter (a party process owns its seed: no two parties, and no two batches, ever share a stream); `known` holds one
    caller-made set for the parity pass against the oracle."""
    writes_in_place = True

    def __init__(self, ctx, E, gk, Bblk, parties, nonces, seed):
        self.ctx, self.E, self.gk, self.B, self.parties = ctx, E, gk, Bblk, parties
        self.sess = E.Gg20Session(ctx, gk, Bblk, parties, nonces)
        self.known, self.seed, self.counter, self.fresh = nonces, seed, 0, None

    def rearm(self, known=False):
        if known:
            self.sess.rearm(self.known)
            return
        self.counter += 1
        self.fresh, _ = self.E.gg20_sample_nonces(self.ctx, self.gk, self.B, self.seed, self.counter, local=self.parties, out=self.fresh,
                                                  msg=self.known["msg"])
        self.sess.rearm(self.fresh)

    def round(self, rnd, d_in, in_off, msg, out=None):
        return self.sess.round(rnd, d_in=d_in, in_off=in_off, msg=msg, out=out)

    def result(self):
        return self.sess.result()

    def close(self):
        self.sess.close()


class PartyMode:
    """Mode B on this rank: party p of session block s lives on rank (s + p) % world, this rank hosts S
    (block, party) pairs of B sessions each, every round's records travel through ONE all-gather (dist.PartySharded).  The
    session objects live across steps, as a party process would keep them: a step re-arms them and runs the nine rounds.

    Every timed step re-arms every hosted (block, parties) object with FRESHLY SAMPLED values (device-side sampler, the object's own
    seed, a new batch counter): re-using k_i, gamma_i or a Paillier randomness across two signatures leaks the key share
    (include/mpecdsa_hip.h: mpe_gg20_session_rearm), and round 4's bench did exactly that.  The warm-up step and the parity pass
    (`step(known=True)`) use one caller-made set whose full-session view the oracle signs.
    comm: engine.Comm — the per-round all-gathers then go through mpe_comm_all_gather (RCCL behind the C-ABI)."""

    def __init__(self, ctx, E, G, mpe_dist, lk, arrays, T, n, signers, B, dev, world, parity_sessions=0, comm=None):
        self.E, self.G, self.lk, self.B, self.S, self.n = E, G, lk, B, len(signers), n
        self.engines, self.block_nonces, self.block_sample = {}, {}, {}
        S = self.S

        def make_engine(s, parties):
            g2 = torch.Generator(device=dev)
            g2.manual_seed(977 * s + 13 + 7919 * n)            # the nonces of a block are a function of the block, on every rank
            full = make_device_nonces(g2, dev, B, S, S, n)
            if parity_sessions:                                # the whole block's sampled values of the first sessions, for the oracle
                k = min(parity_sessions, B)
                self.block_sample[s] = _host({f: v[: k * (v.shape[0] // B)] for f, v in full.items()})
            per = dict(k=1, gamma=1, blind=1, r_a=1, l=1, ped_s1=1, ped_s2=1, heg_s1=1, heg_s2=1, al_alpha=n, al_beta=n, al_gamma=n,
                       al_rho=n, mb_beta_tag=2 * (S - 1), mb_r=2 * (S - 1), mb_nonce_b=2 * (S - 1), mb_nonce_bt=2 * (S - 1),
                       pdl_alpha=S - 1, pdl_beta=S - 1, pdl_rho=S - 1, pdl_gamma=S - 1)
            mine = {}
            for f, v in full.items():
                if f == "msg":
                    mine[f] = v
                else:
                    w = v.shape[1]
                    mine[f] = v.reshape(B, S, per[f], w)[:, parties].reshape(B * len(parties) * per[f], w).contiguous()
            self.block_nonces[s] = mine
            # a key object per hosted (block, parties): ONLY those parties' x_i, p, q reach it (mpe_gg20_keys_create n_own / h_own),
            # as in the reference's deployment where a process holds one party's LocalKey
            gk_own = E.Gg20Keys(ctx, T, n, signers, arrays, own=[signers[p_] for p_ in parties])
            seed = hashlib.sha256(b"bench.py party mode|block %d|parties %s|n %d" % (s, ",".join(map(str, parties)).encode(), n)).digest()
            self.engines[s] = GpuRoundEngine(ctx, E, gk_own, B, parties, mine, seed)
            self.engines[s].keys = gk_own
            return self.engines[s]
        self.ps = mpe_dist.PartySharded(S, B, lambda rnd: E.gg20_msg_words(S, n, rnd), make_engine, dev, placement="rotated",
                                        colocate=world < S, timing=True, comm=comm)
        self.armed = True
        self.last = None

    def step(self, known=False):
        if not self.armed or not known:
            for _, e_ in self.ps.engines.values():
                e_.rearm(known=known)
        self.armed = False
        res = self.ps.run({s: self.block_nonces[s]["msg"] for s in self.ps.engines})
        self.last = res
        first = res[sorted(res)[0]]
        return first["r"][0], first["s"][0], first["recid"][0], torch.cat([r_["status"].reshape(-1) for r_ in res.values()])

    def parity(self, threads):
        """the first sessions of this rank's first hosted block: the signature each hosted party of the block ended with against
        the GMP oracle's for the same sampled values (bit for bit) -> (ok, sessions compared)"""
        if not self.block_sample or self.last is None:
            return None, 0
        self.step(known=True)                                  # one untimed pass on the caller-made values the oracle can sign too
        torch.cuda.synchronize()
        s0 = sorted(self.block_sample)[0]
        host = self.block_sample[s0]
        k = host["msg"].shape[0]
        _, wr, ws, wrecid, wstatus = cpu_baseline_gg20(self.lk, host, k, max(1, min(threads, k)))
        res = self.last[s0]
        ok = bool((wstatus == 0).all())
        for li in range(res["r"].shape[0]):
            ok = ok and np.array_equal(res["r"][li, :k].cpu().numpy().view(np.uint32), wr)
            ok = ok and np.array_equal(res["s"][li, :k].cpu().numpy().view(np.uint32), ws)
            ok = ok and np.array_equal(res["recid"][li, :k].cpu().numpy(), wrecid)
        return ok, k

    def close(self):
        for e_ in self.engines.values():
            e_.close()
            e_.keys.close()


class PostTimingWatchdog:
    """N > 1 only.  Armed when the timed region has ended; if the collective sections after it have not finished by the deadline,
    rank 0 prints the line of the timed region (with `post_timing_sections` saying what happened) and EVERY rank leaves with status 0
    at the same deadline, so the launcher returns instead of waiting on ranks stuck in a collective."""

    def __init__(self, rank, seconds, line_fn=None):
        import threading
        self.rank, self.seconds, self.line_fn = rank, seconds, line_fn
        self.where = "start"
        self._done = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def at(self, where):
        self.where = where

    def disarm(self):
        self._done.set()

    def _run(self):
        if self._done.wait(self.seconds):
            return
        try:
            if self.line_fn is not None:
                res = self.line_fn()
                res["post_timing_sections"] = {"completed": False, "gave_up_after_s": self.seconds, "stuck_in": self.where,
                                               "note": "the timed region had ended and its numbers are final; the sections after it "
                                                       "(per-rank oracle parity / party-sharded pass / config 2 on every GPU) did not finish"}
                sys.stdout.write(json.dumps(res) + "\n")
                sys.stdout.flush()
        finally:
            os._exit(0)


def respawn_under_torchrun(n, argv):
    """`python bench.py --gpus N` with no torch.distributed environment: this process becomes the launcher of N ranks, one
    per GPU (the same command line the driver would use), and exits with their status; rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + argv     # "--": our own flags (--n, --t) are not the launcher's
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sessions", type=int, default=65536, help="concurrent signing sessions per GPU per step")
    ap.add_argument("--t", type=int, default=1, help="threshold (t+1 signers); BASELINE config 5 is --t 2 --n 5")
    ap.add_argument("--n", type=int, default=3, help="parties of the key")
    ap.add_argument("--chunk", type=int, default=0, help="sessions per internal pass of mpe_gg20_sign (0 = library default)")
    ap.add_argument("--dedup", action="store_true", help="evaluate identical checks once (same outputs; not the faithful path)")
    ap.add_argument("--mode", choices=["session", "party"], default="session", help="multi-GPU layout (SURVEY.md 8e A / B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the c2 / c3 / c4 / c5 / Lindell sections")
    ap.add_argument("--stream-batches", type=int, default=96, help="c4_stream_1024: batches of 1 024 sessions in the stream")
    ap.add_argument("--stream-lanes", type=int, default=4, help="c4_stream_1024: passes in flight (mpe_gg20_pipeline lanes; 4 since round 6: "
                    "16.85 k signatures/s at p50 0.97 s against 16.1 - 16.3 k at 1.02 s with 2, profiles/r06/ab_stream12.jsonl)")
    ap.add_argument("--stream-group", type=int, default=4, help="c4_stream_1024: batches coalesced per pass")
    ap.add_argument("--dump-launches", action="store_true", help="add the timed region's heavy launches (kind, bits, batch, ms) to the line")
    ap.add_argument("--only", default="", help="comma list of config sections to run after the timed region (default: all)")
    ap.add_argument("--no-mode-b", action="store_true", help="N > 1, session mode: skip the party-sharded (config 5 shape) pass after the timed region")
    ap.add_argument("--mode-b-sessions", type=int, default=0, help="sessions per block of that pass (0 = min(8192, --sessions))")
    ap.add_argument("--mode-b-steps", type=int, default=2)
    ap.add_argument("--share-device", action="store_true",
                    help="all ranks use cuda:0 and talk through gloo (host-staged): exercises the N>1 code path on a 1-GPU box; "
                         "the ranks time-share one GPU, so `value` says nothing about a node")
    ap.add_argument("--wallet-cases", default="16384x16384,4096x4096,4096x16384",
                    help="c4_every_session_its_own_wallet: comma list of WALLETSxSESSIONS (every wallet has moduli of its own)")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE config 5 at its literal shape as the TIMED region: 65 536 concurrent t=2 n=5 sessions over the node "
                         "(65536 / N per GPU), party-sharded, one RCCL all-gather per round (= --t 2 --n 5 --mode party --sessions 65536/N)")
    args = ap.parse_args()
    if args.config5:
        args.t, args.n, args.mode = 2, 5, "party"
        if args.sessions == 65536:                              # (an explicit --sessions keeps a smoke run small)
            args.sessions = max(1, 65536 // max(1, args.gpus))

    if "RANK" not in os.environ and args.gpus > 1:
        # no launcher around us: become it (N ranks, one per GPU, RCCL over xGMI)
        sys.exit(respawn_under_torchrun(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting n_gpus = {world}", file=sys.stderr)
    distributed = world > 1 or "RANK" in os.environ          # under torch.distributed.run even one rank goes through RCCL
    share = args.share_device or bool(os.environ.get("MPE_BENCH_SHARE_DEVICE"))
    if not share and distributed and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); "
                         "--share-device runs all ranks on cuda:0 over gloo")
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    rccl = None
    if distributed:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # the collective library really is there and really spans the ranks: all-reduce of ones == world
            ones = torch.ones(1, dtype=torch.float32, device=torch.device("cuda", local_rank))
            dist.all_reduce(ones)
            torch.cuda.synchronize()
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
            rccl = {"backend": dist.get_backend(), "all_reduce_of_ones": float(ones.item()), "ok": float(ones.item()) == world,
                    "version": ver}

    import fixtures as F
    import gg20_fixture as G
    from multi_party_ecdsa_amd import dist as mpe_dist
    from multi_party_ecdsa_amd import engine as E
    keys = F.load_keys()
    T, N_PARTIES = args.t, args.n
    ctx = E.Context(local_rank)
    dev = ctx.device
    SIGNERS = list(range(T + 1))                               # parties 1..t+1 sign
    B, S, n = args.sessions, len(SIGNERS), N_PARTIES
    lk = G.make_local_keys(keys, T, N_PARTIES, SIGNERS)
    global SLIDING_EXPONENTS
    SLIDING_EXPONENTS = [keys[i].N for i in SIGNERS]            # every launch on sliding windows raises to a signer's modulus N
    # the public key tables come from rank 0 once (LocalKey's public part is identical for everybody)
    pub = {f: torch.from_numpy(np.ascontiguousarray(lk["arrays"][f]).view(np.int32)) for f in ("Nt", "h1", "h2", "y", "X")}
    coll_dev = torch.device("cpu") if share else dev           # where the (tiny) control collectives run: gloo works on host tensors
    pub = mpe_dist.broadcast_tables(pub, coll_dev)
    arrays = dict(lk["arrays"])
    for f in pub:
        arrays[f] = np.ascontiguousarray(pub[f].cpu().numpy().view(np.uint32))
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242 + rank)
    extra = {}
    gk = None
    sampler = None
    if args.mode == "session":
        gk = E.Gg20Keys(ctx, T, N_PARTIES, SIGNERS, arrays)      # every signer local: the Simulation harness of the reference
        # Every value the reference draws from OsRng is drawn ON THE DEVICE, INSIDE the timed step, from (seed, step counter) with the
        # reference's distributions (mpe_gg20_sample_nonces: sample_below by rejection, from_modulo with its gcd, Scalar::random):
        # no step signs with the nonces of another, and what a host hands over per step is 32 bytes + the messages.
        seed = hashlib.sha256(b"bench.py headline|rank %d" % rank).digest()
        msg = rand_words(gen, dev, B, 8, 8)
        nonces, fail0 = E.gg20_sample_nonces(ctx, gk, B, seed, 0, msg=msg)
        torch.cuda.synchronize()
        sampler = {"seed": seed, "counter": 0, "events": [], "fail": fail0}

        def step():
            sampler["counter"] += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _, sampler["fail"] = E.gg20_sample_nonces(ctx, gk, B, seed, sampler["counter"], out=nonces)
            e1.record()
            sampler["events"].append((e0, e1))
            return E.gg20_sign(ctx, gk, nonces, B, dedup_verify=args.dedup, chunk=args.chunk)
    else:
        # party p of session block s on rank (s + p) % world: this rank hosts S (block, party) pairs of B sessions each —
        # the same per-GPU work as B whole sessions; the messages of every round travel through one all-gather
        comm = make_comm(ctx, E, rank, world, share, distributed)
        pm = PartyMode(ctx, E, G, mpe_dist, lk, arrays, T, N_PARTIES, SIGNERS, B, dev, world, parity_sessions=0 if args.no_cpu_baseline else 8, comm=comm)
        engines = pm.engines
        gather_test = pm.ps.layout_self_test()                 # the gather layout on the real backend, before anything is timed
        ps_holder = {"ps": pm.ps}
        step = pm.step

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    if sampler is not None:
        sampler["events"] = []
    if args.mode == "party":
        ps_holder["ps"].comm_seconds()                     # drop the warm-up's share
    comm_total = 0.0
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.mode == "party":
        comm_total = ps_holder["ps"].comm_seconds()        # HIP events around every all-gather of the timed region
    recs = ctx.prof_collect(16384)
    ctx.prof_enable(False)
    own_elapsed = elapsed
    elapsed = mpe_dist.max_over_ranks(elapsed, coll_dev)     # the job ends when its slowest rank does
    per_rank = None
    if distributed:
        mine = torch.tensor([B * args.steps / own_elapsed, float(bool((out[3] == 0).all().item()))], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rates = [float(t_[0]) for t_ in every]
        per_rank = {"signatures_per_s": rates, "min": min(rates), "max": max(rates), "all_ranks_signed": all(float(t_[1]) == 1.0 for t_ in every)}
    usable_cores, _quota = host_cores()
    rank_threads = max(1, min(16, usable_cores // max(1, world if not share else world)))      # the ranks share the host's cores
    gather_layout = None
    if args.mode == "party":
        ps = ps_holder["ps"]
        gather_layout = gather_test
        extra = {"bytes_all_gathered_per_round": {str(k): int(v) for k, v in ps.bytes_per_round.items()},
                 "rccl_time_share": comm_total / (elapsed if elapsed > 0 else 1.0), "placement": ps.placement,
                 "pairs_per_rank": ps.per_rank, "gather_mode": ps.gather_mode,
                 "nonces": "every timed step re-arms every hosted (block, parties) object with values sampled on the device from its own seed and a fresh batch counter",
                 "fan_out": "mpe_comm_all_gather (ncclAllGather behind the C-ABI)" if ps.comm is not None else "torch.distributed (host-staged under gloo)"}

    def core_line():
        """The Mode-A line proper: everything the timed region determines (rank 0)."""
        all_signed = bool((out_host[3] == 0).all())
        # roofline of the dominant kernel: every launch modulo N^2 (4096 bit) of the timed region
        dom = [x for x in recs if x["kind"] in (0, 3, 6) and x["bits"] == 4096]
        dom_s = sum(x["ms"] for x in dom) * 1e-3
        pair = bool(dom) and all(x["kind"] in (3, 6) for x in dom)
        # a two-base launch (mpe_modexp2 pattern) does the algorithmic work of both exponentiations
        def rec_macs(x, k):
            m = modexp_macs(k, EXP_BITS.get(x["exp_words"], 32 * x["exp_words"]))
            if x.get("exp2_words"):
                m += modexp_macs(k, EXP_BITS.get(x["exp2_words"], 32 * x["exp2_words"]))
            return x["batch"] * m
        dom_macs = sum(rec_macs(x, 128) for x in dom)
        # the MACs the executed algorithm needs (N-adic pairs: half-size passes) — what the hardware is asked to do
        exe_macs = sum(x["batch"] * pair_modexp_macs(64, x["exp_words"], x.get("exp2_words", 0), sliding=slid(x)) for x in dom) if pair else dom_macs
        n_sliding = sum(1 for x in dom if x["kind"] == 6)
        sec = [x for x in recs if x["kind"] in (0, 3, 6) and x["bits"] == 2048]
        sec_s = sum(x["ms"] for x in sec) * 1e-3
        heavy_s = sum(x["ms"] for x in recs) * 1e-3
        value = B * world * args.steps / elapsed
        # HBM traffic of the dominant kernel: measured in separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE, WRITE_SIZE; gfx950 correction applied) and committed under profiles/ — not re-measured here
        traffic, traffic_src = None, None
        for rel in ("profiles/r06/pmc_traffic.json", "profiles/r05/pmc_traffic.json", "profiles/r04/pmc_traffic.json", "profiles/r03/pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, rel)) as f:
                    pmc = json.load(f)
                if pmc.get("sessions") == B and not args.dedup and args.mode == "session" and (T, N_PARTIES) == (1, 3):
                    # a figure QUOTED from a committed PMC pass (another box, possibly an earlier kernel build): say so next to it
                    traffic = pmc["hbm_bytes_per_launch"]
                    traffic_src = {"file": rel, "measured_in_this_run": False, "host": pmc.get("host"), "commit": pmc.get("commit"),
                                   "sessions": pmc.get("sessions"),
                                   "kernel_avg_ms_there": pmc.get("kernels", {}).get(pmc.get("dominant_kernel", ""), {}).get("avg_ms")}
                    break
            except OSError:
                pass
        nl = max(1, len(dom))
        res = {
            "metric": f"GG20 signatures/sec (t={T}, n={N_PARTIES}; all parties of each session on the node's GPUs) + Paillier-2048 modexp/s per GPU",
            "value": value, "unit": "signatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (radix 2^29) / u64 accumulators", "data": "synthetic",
            "config": {"workload": f"{B} concurrent GG20 t={T} n={N_PARTIES} signing sessions per GPU, full MtA path (BASELINE config 4 shape), "
                                   f"{'deduplicated checks' if args.dedup else 'faithful work'}, one LocalKey fixture, signers {{1..{T + 1}}}",
                       "sessions_per_gpu": B, "t": T, "n": N_PARTIES, "signers": S,
                       "parallelism": (f"session-sharded x{world}, no data-path collective" if args.mode == "session" else
                                       f"party-sharded x{world}: party p of session block s on rank (s+p)%{world}, one {'gloo (host-staged)' if share else 'RCCL'} all-gather per round"),
                       **extra},
            "roofline": {"bound": "valu-int (v_mad_u64_u32 issue rate; HBM traffic is negligible)",
                         "achieved": exe_macs / dom_s / 1e12 if dom_s else None, "peak": PEAK_MAC_PER_S / 1e12,
                         "unit": "TMAC/s: 32x32+64 MACs the executed algorithm needs on ideal 32-bit limbs (N-adic pairs: 2 MAC(64) per "
                                 "squaring, 2.5 per multiplication modulo N^2, the kernel's own window counts)",
                         "frac": exe_macs / dom_s / PEAK_MAC_PER_S if dom_s else None,
                         "alg_unit_TMAC_per_s": dom_macs / dom_s / 1e12 if dom_s else None,
                         "alg_unit_frac": dom_macs / dom_s / PEAK_MAC_PER_S if dom_s else None,
                         "alg_unit_note": "SURVEY.md 8d's unit (CIOS on 32-bit limbs of the 4096-bit modulus, 4-bit windows): above the peak "
                                          "because the pair arithmetic computes the same residues with ~0.46x those MACs",
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE; separate rocprofv3 --pmc passes)",
                         "traffic_source": traffic_src,
                         # flat, because consumers that keep only roofline's scalars drop the object above: the figure is QUOTED from a
                         # committed PMC pass of this command (rocprofv3 --pmc cannot run inside this process), never measured by this run
                         "traffic_measured_in_this_run": False, "traffic_quoted_from": traffic_src["file"] if traffic_src else None,
                         "traffic_quoted_commit": traffic_src["commit"] if traffic_src else None,
                         "kernel": ("mpe::pair_modexp_kernel<Cfg<2048,29,18,4>>" if pair else "mpe::modexp_kernel<Cfg<4096,29,18,8>>") +
                                   " (all launches modulo N^2 of the timed region)",
                         "launches": len(dom), "launches_on_sliding_windows": n_sliding,
                         "sliding_share_of_waves": (sum(slid(x) * x["batch"] for x in dom) / max(1, sum(x["batch"] for x in dom if x["kind"] == 6))
                                                    if n_sliding else 0.0),
                         "sliding_schedule_priced_on": "the signers' own moduli N (exact left-to-right window counts)",
                         "avg_kernel_ms": dom_s / nl * 1e3,
                         "executed_mac_per_launch": exe_macs / nl, "alg_unit_mac_per_launch": dom_macs / nl,
                         "kernel_time_share_of_step": dom_s / elapsed,
                         # the issue ceiling actually measured for this instruction (tools/ubench/valu_rate.hip): a stream of
                         # v_mad_u64_u32 with VGPR operands sustains 31.2 T lane-ops/s at the kernel's 2 waves per SIMD, not the 39.3 T
                         # of the 4-cycle issue model.  Against it stand the MACs the kernel REALLY executes: 29-bit limbs, 72 limbs x
                         # 71 CIOS steps per pass (LIMB_INFLATION = 1.238 x the ideal count) — multiply instructions only; the ~14 %
                         # cheaper non-MAC instructions of the loops are not priced at the MAC's rate (round 4 did, and got 1.01)
                         "issue_ceiling": ({"measured_T_mac_per_s": 31.2, "source": "profiles/r01_valu_rate.json (mad_u64_u32_vv, 2 waves/SIMD)",
                                            "executed_29bit_T_mac_per_s": exe_macs * LIMB_INFLATION / dom_s / 1e12,
                                            "frac_of_measured": exe_macs * LIMB_INFLATION / dom_s / 31.2e12,
                                            "limb_inflation": LIMB_INFLATION}
                                           if pair and dom_s else None)},
            "breakdown": {"modexp4096_s_per_step": dom_s / args.steps, "modexp2048_s_per_step": sec_s / args.steps,
                          "heavy_kernels_s_per_step": heavy_s / args.steps, "wall_s_per_step": elapsed / args.steps,
                          "alg_unit_mac_per_signature": sig_macs(S, n),
                          "fb_window_bits": (gk if gk is not None else next(iter(engines.values())).keys).fb_window_bits()},
            "roofline_secondary": secondary_rooflines(recs, elapsed),
            "whole_step": whole_step(recs, elapsed),
            "all_sessions_signed": all_signed, "launch": ctx.launch_info(),
            "device": {"name": torch.cuda.get_device_name(local_rank), "host": os.uname().nodename,
                       "note": "boxes of this pool differ by up to ~3 % in signatures/s (profiles/r03/README.md)"},
        }
        if args.dump_launches:
            res["launches_timed_region"] = [{"kind": x["kind"], "bits": x["bits"], "exp_words": x["exp_words"], "exp2_words": x.get("exp2_words", 0),
                                             "batch": x["batch"], "ms": round(x["ms"], 3)} for x in recs]
        if distributed:
            res["per_rank"] = per_rank
            res["all_sessions_signed"] = all_signed and per_rank["all_ranks_signed"]
            res["rccl"] = dict(rccl) if rccl is not None else {"backend": "gloo", "note": "--share-device: every rank on cuda:0, collectives staged "
                                                               "through host memory; the ranks time-share one GPU (a functional run, not a node figure)"}
        return res

    # The timed region is over and `elapsed` is final.  What follows at N > 1 (per-rank oracle parity, the party-sharded pass over the
    # library's own RCCL communicator, config 2 on every GPU) is collective work that has never met more than one GPU before the
    # driver's node: if any of it wedges, every rank gives up at the same deadline and rank 0 still prints the Mode-A line it has —
    # marked as such — instead of the whole scaling run ending in the launcher's timeout with nothing.
    out_host = [o.cpu().numpy() for o in out]
    watchdog = PostTimingWatchdog(rank, float(os.environ.get("MPE_BENCH_POST_TIMEOUT_S", "420")), core_line if rank == 0 else None) if (distributed and world > 1) else None
    # (c) a parity sample against the GMP oracle on EVERY rank at N > 1 (at N = 1 the cpu_baseline leg below does it on 256+ sessions)
    rank_parity = None
    if distributed and not args.no_cpu_baseline:
        if watchdog is not None:
            watchdog.at("per-rank oracle parity")
        if args.mode == "session":
            k = min(B, 16)
            host_n = _host({f: v[: k * (v.shape[0] // B)] for f, v in nonces.items()})
            _, wr, ws, wrecid, wstatus = cpu_baseline_gg20(lk, host_n, k, min(rank_threads, k))
            ok = bool((wstatus == 0).all() and np.array_equal(out[0][:k].cpu().numpy().view(np.uint32), wr) and
                      np.array_equal(out[1][:k].cpu().numpy().view(np.uint32), ws) and np.array_equal(out[2][:k].cpu().numpy(), wrecid))
        else:
            ok, k = pm.parity(rank_threads)
        flags = torch.tensor([1.0 if ok else 0.0, float(k)], dtype=torch.float64, device=coll_dev)
        every_f = [torch.zeros_like(flags) for _ in range(world)]
        dist.all_gather(every_f, flags)
        rank_parity = {"ok_per_rank": [bool(float(t_[0])) for t_ in every_f], "sessions_per_rank": int(k),
                       "all_ok": all(float(t_[0]) == 1.0 for t_ in every_f),
                       "what": "(r, s, recid) of the first sessions of every rank's own batch, bit for bit against the GMP oracle"}
        if per_rank is not None:
            per_rank["parity_vs_oracle"] = rank_parity

    # (b) Mode B in the SAME line: after the session-sharded timed region a short party-sharded pass at BASELINE config 5's
    # per-GPU share (t=2, n=5: S=3 signers; party p of session block s on rank (s+p) % N; one all-gather per round), so that
    # the driver's one command `bench.py --gpus N` exercises the RCCL data path and reports its rate beside Mode A's
    mode_b = None
    want_b = (world > 1 or os.environ.get("MPE_BENCH_FORCE_MODE_B")) and distributed and args.mode == "session" and not args.no_mode_b
    if want_b:
        if watchdog is not None:
            watchdog.at("mode_b (party-sharded pass over mpe_comm_*)")
        try:
            tb, nb, sg_b = 2, 5, [0, 1, 2]
            Bb = args.mode_b_sessions if args.mode_b_sessions else min(8192, B)
            lk_b = G.make_local_keys(keys, tb, nb, sg_b)
            comm_b = make_comm(ctx, E, rank, world, share, distributed)
            pm_b = PartyMode(ctx, E, G, mpe_dist, lk_b, lk_b["arrays"], tb, nb, sg_b, Bb, dev, world,
                             parity_sessions=0 if args.no_cpu_baseline else 4, comm=comm_b)
            gather_layout = pm_b.ps.layout_self_test()
            pm_b.step()                                        # warm-up
            torch.cuda.synchronize()
            pm_b.ps.comm_seconds()
            dist.barrier()
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            for _ in range(args.mode_b_steps):
                out_b = pm_b.step()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            own_b = time.perf_counter() - tb0
            comm_b = pm_b.ps.comm_seconds()
            el_b = mpe_dist.max_over_ranks(own_b, coll_dev)
            ok_b, k_b = (None, 0) if args.no_cpu_baseline else pm_b.parity(rank_threads)
            signed_b = bool((out_b[3] == 0).all().item())
            mine_b = torch.tensor([Bb * args.mode_b_steps / own_b, 1.0 if signed_b else 0.0, 1.0 if (ok_b or ok_b is None) else 0.0, comm_b],
                                  dtype=torch.float64, device=coll_dev)
            every_b = [torch.zeros_like(mine_b) for _ in range(world)]
            dist.all_gather(every_b, mine_b)
            S_b = len(sg_b)
            # a rank hosts S (block, party) pairs of Bb sessions: world blocks of Bb sessions are signed per step by the node
            mode_b = {"workload": f"{Bb} sessions per GPU-share, t={tb} n={nb} (BASELINE config 5's shape), party-sharded: party p of session block s on "
                                  f"rank (s+p)%{world}, {world} blocks, one {'gloo (host-staged)' if share else 'RCCL'} all-gather per round",
                      "signatures_per_s": Bb * world * args.mode_b_steps / el_b, "ms_per_step": el_b / args.mode_b_steps * 1e3,
                      "steps": args.mode_b_steps, "sessions_per_block": Bb, "blocks": world, "signers": S_b,
                      "rccl_time_share": max(float(t_[3]) for t_ in every_b) / el_b,
                      "bytes_all_gathered_per_round": {str(k_): int(v_) for k_, v_ in pm_b.ps.bytes_per_round.items()},
                      "gather_mode": pm_b.ps.gather_mode, "per_rank_signatures_per_s": [float(t_[0]) for t_ in every_b],
                      "all_sessions_signed": all(float(t_[1]) == 1.0 for t_ in every_b),
                      "parity_sample_vs_oracle": None if args.no_cpu_baseline else all(float(t_[2]) == 1.0 for t_ in every_b),
                      "parity_sessions_per_rank": int(k_b),
                      "nonces": "fresh per step: device-side sampler, one seed per hosted (block, parties) object",
                      "fan_out": "mpe_comm_all_gather (ncclAllGather behind the C-ABI)" if pm_b.ps.comm is not None else "torch.distributed (host-staged under gloo)"}
            pm_b.close()
        except Exception as e_b:                               # noqa: BLE001 — the Mode-A line must survive a Mode-B failure, and say so
            mode_b = {"error": repr(e_b)}

    # north_star asks for Paillier ops/s at 1, 2, 4 and 8 GPUs too: at N > 1 every rank runs BASELINE config 2 at the same time
    # (after the timed signing region) and the rates add up; at N = 1 it is the c2 section below
    node_paillier = None
    if distributed and (world > 1 or os.environ.get("MPE_BENCH_FORCE_NODE_PAILLIER")) and args.mode == "session" and not args.no_configs:
        if watchdog is not None:
            watchdog.at("config 2 on every GPU")
        dist.barrier()
        p2 = paillier_config2(ctx, E, keys, F)
        agg = torch.tensor([p2["ops_per_s"], p2["encrypt_per_s"], p2["decrypt_per_s"], p2["modexp4096_2048_per_s"]], dtype=torch.float64, device=coll_dev)
        lo = agg.clone()
        dist.all_reduce(agg)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        node_paillier = {"n_gpus": world, "batch_per_gpu": p2["batch"], "ops_per_s": float(agg[0]), "encrypt_per_s": float(agg[1]),
                         "decrypt_per_s": float(agg[2]), "modexp4096_2048_per_s": float(agg[3]), "slowest_gpu_ops_per_s": float(lo[0]),
                         "roundtrip_ok": bool(p2["roundtrip_ok"]), "note": "sum over ranks of BASELINE config 2 run concurrently on every GPU"}
    if watchdog is not None:
        watchdog.disarm()
    if rank == 0:
        res = core_line()
        r, s, recid, status = out_host
        all_signed = bool((status == 0).all())
        if distributed:
            if mode_b is not None:
                res["mode_b"] = mode_b
            # the layout of the round all-gather checked on the real backend before any party-sharded work (dist.PartySharded.layout_self_test)
            res["rccl"]["all_gather_layout_self_test"] = gather_layout
        # the other configs and the CPU baseline belong to the single-GPU line (rank 0 at N=1 only): at N>1 the other ranks
        # would just wait for them
        single = world == 1 and args.mode == "session"
        usable, quota = host_cores()
        threads = min(usable, 64)
        if single and not args.no_cpu_baseline:
            sample = min(B, max(1024, 16 * threads))     # >= BASELINE config 4's full size: the driver's line alone carries its parity
            host_nonces = _host({f: v[: sample * (v.shape[0] // B)] for f, v in nonces.items()})
            one = min(4, sample)
            t1 = time.time()
            G.oracle_sign(lk, host_nonces, one)
            per_core = one / (time.time() - t1)
            v, wr, ws, wrecid, wstatus = cpu_baseline_gg20(lk, host_nonces, sample, threads)
            parity = bool((wstatus == 0).all() and np.array_equal(r[:sample].view(np.uint32), wr) and
                          np.array_equal(s[:sample].view(np.uint32), ws) and np.array_equal(recid[:sample], wrecid))
            res["cpu_baseline"] = {"value": v, "unit": "signatures/s", "cores": threads, "kind": "port", "per_core_1thread": per_core,
                                   "parallel_speedup": v / per_core, "host_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
                                   "sample": f"the first {sample} sessions of the same batch ({sample // threads} per thread; GMP oracle, "
                                             f"{threads} threads of {os.cpu_count()} host CPUs)"}
            res["parity_vs_oracle_on_sample"] = parity
            res["parity_sample"] = sample
        if sampler is not None:
            samp_s = sum(a_.elapsed_time(b_) for a_, b_ in sampler["events"]) * 1e-3
            res["sampling"] = {"where": "on the device, inside the timed step (mpe_gg20_sample_nonces: ChaCha20 keystream of a 32-byte seed, curv's "
                                        "sample_below / sample_range / from_modulo / Scalar::random rules; a fresh batch counter per step)",
                               "seconds_per_step": samp_s / max(1, len(sampler["events"])), "share_of_step": samp_s / elapsed,
                               "steps_sampled": len(sampler["events"]), "rejection_loops_given_up": int(sampler["fail"].item()),
                               "bytes_sampled_per_step": int(sum(v.numel() * 4 for f_, v in nonces.items() if f_ != "msg"))}
            res["config"]["sampling_share_of_step"] = res["sampling"]["share_of_step"]
            if single and not args.no_cpu_baseline:
                # the oracle expands the SAME (seed, counter) of the last step: identical arrays (oracle/sampler_oracle.c)
                k_ = min(B, 64)
                want_, wf_ = G.oracle_sample_nonces(lk, k_, sampler["seed"], sampler["counter"])
                got_ = _host({f_: v[: k_ * (v.shape[0] // B)] for f_, v in nonces.items() if f_ != "msg"})
                res["sampling"]["oracle_expands_the_seed_to_the_same_values"] = bool(wf_ == 0 and all(np.array_equal(got_[f_], want_[f_]) for f_ in got_))
                res["sampling"]["sessions_compared"] = k_
        if single:
            # A host that samples for itself and hands its values over for every batch (mpe_gg20_sign with caller arrays): the measured cost of moving
            # the step's inputs in (pinned host memory -> HBM) and its signatures out.  Reported beside the line, never part of `value`.
            try:
                pinned = {f: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v) for f, v in nonces.items()}
                outs = [torch.empty((B, 8), dtype=torch.int32, device=dev), torch.empty((B, 8), dtype=torch.int32, device=dev),
                        torch.empty((B,), dtype=torch.int32, device=dev)]
                host_out = [torch.empty(o.shape, dtype=o.dtype, pin_memory=True) for o in outs]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for f, v in pinned.items():
                    nonces[f].copy_(v, non_blocking=True)
                for o, h in zip(outs, host_out):
                    h.copy_(o, non_blocking=True)
                torch.cuda.synchronize()
                io_s = time.perf_counter() - t1
                nin = sum(v.numel() * 4 for v in pinned.values())
                nout = sum(o.numel() * 4 for o in outs)
                step_s = elapsed / args.steps
                res["host_handover"] = {"bytes_in_per_step": nin, "bytes_out_per_step": nout, "seconds": io_s, "GB_per_s": (nin + nout) / io_s / 1e9,
                                        "signatures_per_s_if_not_overlapped": B / (step_s + io_s), "share_of_step": io_s / step_s,
                                        "note": "pinned host memory over PCIe, serial with the step; `value` has its inputs resident in HBM"}
                del pinned, host_out, outs
            except RuntimeError as e_:
                res["host_handover"] = {"error": str(e_)}
            # every signature of the timed batch under OpenSSL (not only "status == 0", which is the device's own verdict)
            res["openssl"] = openssl_verify_all(lk["arrays"]["y"][0], nonces["msg"], r, s, threads)
            res["openssl_verified"] = res["openssl"]["openssl_verified"]
        if single and not args.no_configs and (T, N_PARTIES) == (1, 3):
            gk.close()
            cfg, took = {}, {}

            only = [x for x in args.only.split(",") if x]

            def section(name, fn):
                if only and not any(name.startswith(o) for o in only):
                    return
                t_ = time.perf_counter()
                try:
                    cfg[name] = fn()
                except Exception as e_:                      # a secondary section never takes the headline line down: it reports its failure
                    import traceback
                    cfg[name] = {"error": f"{type(e_).__name__}: {e_}", "where": traceback.format_exc().strip().splitlines()[-3:]}
                took[name] = round(time.perf_counter() - t_, 2)
            section("c2_paillier_65536", lambda: paillier_config2(ctx, E, keys, F, oracle_threads=0 if args.no_cpu_baseline else threads, oracle_items=4096))
            section("c3_ec_pdl_262144", lambda: config3(ctx, E, keys, F, oracle=not args.no_cpu_baseline))
            section("c3b_bob_65536", lambda: bob_section(ctx, E, keys, F, oracle=not args.no_cpu_baseline))
            section("c4_literal_1024", lambda: gg20_config(ctx, E, G, keys, 1, 3, 1024, 4, gen, parity_sample=0 if args.no_cpu_baseline else 128,
                                                           openssl=True))
            def stream_section():
                main_ = c4_pipeline(ctx, E, G, keys, batches=args.stream_batches, lanes=args.stream_lanes, group=args.stream_group,
                                    oracle=not args.no_cpu_baseline, parity_sample=16)
                # the same stream with twice the batches per pass: more throughput for a longer pass
                o_ = c4_pipeline(ctx, E, G, keys, batches=args.stream_batches, lanes=args.stream_lanes, group=2 * args.stream_group,
                                 oracle=False, parity_sample=0)
                keep_ = ("signatures_per_s", "in_flight_bound", "latency_ms", "pass_ms", "all_sessions_signed", "openssl_verified", "openssl_of")
                main_["other_shapes"] = {f"{args.stream_lanes}x{2 * args.stream_group}": {k_: o_[k_] for k_ in keep_}}
                # ... and with only ONE group outstanding per lane: a batch never waits behind a queued pass (about half the latency),
                # the lanes idle while the host turns a completed group around
                l_ = c4_pipeline(ctx, E, G, keys, batches=args.stream_batches, lanes=args.stream_lanes, group=args.stream_group,
                                 oracle=False, parity_sample=0, window=args.stream_lanes * args.stream_group)
                main_["other_shapes"][f"{args.stream_lanes}x{args.stream_group}_one_group_outstanding_per_lane"] = {k_: l_[k_] for k_ in keep_}
                # group = 2: half the pass, half the latency (the review's shape) — with FOUR lanes: 4 x 2 keeps as many sessions in flight as
                # 2 x 4 in passes half as long (profiles/r06/ab_stream12.jsonl: 2 x 2 12.4 - 14.0 k at p50 0.58 - 0.66 s, 4 x 2 15.5 k at 0.50 - 0.54 s)
                g2_ = c4_pipeline(ctx, E, G, keys, batches=args.stream_batches, lanes=4, group=2, oracle=False, parity_sample=0)
                main_["other_shapes"]["4x2"] = {k_: g2_[k_] for k_ in keep_}
                if args.stream_lanes != 2:                   # rounds 4-5's shape, for continuity
                    r5_ = c4_pipeline(ctx, E, G, keys, batches=args.stream_batches, lanes=2, group=args.stream_group, oracle=False, parity_sample=0)
                    main_["other_shapes"][f"2x{args.stream_group}"] = {k_: r5_[k_] for k_ in keep_}
                # the same service under an OPEN loop: Poisson arrivals at 50 % and 90 % of the closed-loop capacity, arrival-driven grouping
                main_["open_loop"] = c4_open_loop(ctx, E, G, keys, main_["signatures_per_s"], lanes=args.stream_lanes, group=args.stream_group)
                main_["open_loop_2_lanes"] = c4_open_loop(ctx, E, G, keys, main_["signatures_per_s"], loads=(0.5,), lanes=2, group=args.stream_group)
                return main_
            section("c4_stream_1024", stream_section)
            section("c5_share_t2n5_8192", lambda: gg20_config(ctx, E, G, keys, 2, 5, 8192, 1, gen, parity_sample=0 if args.no_cpu_baseline else 32,
                                                              openssl=True))
            section("c4_multi_wallet_16384", lambda: [multi_wallet(ctx, E, G, keys, K_, 16384, gen) for K_ in (16, 1024)])
            mint_cache = {}
            section("c4_every_session_its_own_wallet", lambda: [multi_wallet_distinct(ctx, E, G, keys, K_, B_, gen, threads, cache=mint_cache,
                                                                                     parity_sample=0 if args.no_cpu_baseline else 16)
                                                                for K_, B_ in [tuple(int(v_) for v_ in x_.split("x")) for x_ in args.wallet_cases.split(",") if x_]])
            section("lindell17", lambda: lindell_section(ctx, E, keys, F, cpu=not args.no_cpu_baseline))
            section("f2_blame_4096", lambda: blame_section(ctx, E, G, keys, F, gen, cpu=not args.no_cpu_baseline))
            section("f3_keygen_verify_8192", lambda: keygen_verify_section(ctx, E, keys, F, cpu=not args.no_cpu_baseline))
            if "lindell17" in cfg:
                res["lindell17"] = cfg.pop("lindell17")
            res["configs"] = cfg
            if "c2_paillier_65536" in cfg:
                res["paillier"] = cfg["c2_paillier_65536"]
            res["section_seconds"] = took
        if node_paillier is not None:
            res["paillier"] = node_paillier
        # the scalars a reader of the truncated line needs, inside the two objects every consumer keeps
        res["roofline"]["whole_step_frac"] = res["whole_step"]["frac"]
        also = {"whole_step_frac": res["whole_step"]["frac"]}
        cfgs = res.get("configs", {})
        if isinstance(cfgs.get("c4_literal_1024"), dict) and "signatures_per_s" in cfgs["c4_literal_1024"]:
            also["c4_literal_1024_signatures_per_s"] = cfgs["c4_literal_1024"]["signatures_per_s"]
        if isinstance(cfgs.get("c4_stream_1024"), dict) and "signatures_per_s" in cfgs["c4_stream_1024"]:
            also["c4_stream_1024_signatures_per_s"] = cfgs["c4_stream_1024"]["signatures_per_s"]
            also["c4_stream_1024_latency_ms_p50"] = cfgs["c4_stream_1024"]["latency_ms"]["p50"]
        if isinstance(cfgs.get("c5_share_t2n5_8192"), dict) and "signatures_per_s" in cfgs["c5_share_t2n5_8192"]:
            also["c5_share_t2n5_8192_signatures_per_s"] = cfgs["c5_share_t2n5_8192"]["signatures_per_s"]
        if isinstance(res.get("paillier"), dict) and "modexp4096_2048_per_s" in res["paillier"]:
            also["paillier_2048_modexp_per_s"] = res["paillier"]["modexp4096_2048_per_s"]
        res["config"]["also"] = also
        for k_, v_ in also.items():                          # ... and flat inside roofline{}, whose scalars every consumer keeps
            res["roofline"].setdefault(k_, v_)
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
