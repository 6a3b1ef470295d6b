"""Quick on-GPU sanity + timing of mpe_modexp against Python pow (development aid)."""
import random, sys, time, json
import torch
sys.path.insert(0, ".")
from multi_party_ecdsa_amd import engine as E
from multi_party_ecdsa_amd.words import ints_to_words, words_to_ints

def rnd_mod(r, bits):
    return r.getrandbits(bits) | (1 << (bits - 1)) | 1

def main():
    r = random.Random(7)
    ctx = E.Context(0)
    res = {}
    for bits in (2048, 4096):
        for B, ebits, nmod in ((5, 64, 5), (40, 256, 3), (300, 521, 300)):
            mods = [rnd_mod(r, bits) for _ in range(nmod)]
            if nmod >= 3:
                mods[1] = r.getrandbits(bits - 3) | 1          # a short modulus
                mods[2] = (1 << bits) - 1                       # all ones
            ms = E.ModSet(ctx, bits, mods)
            idx = [i % nmod for i in range(B)]
            bases = [r.getrandbits(bits) for _ in range(B)]
            bases[0] = 0; bases[-1] = mods[idx[-1]]
            exps = [r.getrandbits(ebits) for _ in range(B)]
            exps[1 % B] = 0
            got = E.mod_pow(ctx, ms, bases, exps, mod_idx=idx, exp_bits=ebits)
            want = [pow(b, e, mods[i]) for b, e, i in zip(bases, exps, idx)]
            bad = [i for i in range(B) if got[i] != want[i]]
            print(f"modexp bits={bits} B={B} ebits={ebits}: mismatches={len(bad)} {bad[:8]}", flush=True)
            a = [r.getrandbits(bits) for _ in range(B)]; b = [r.getrandbits(bits) for _ in range(B)]
            gm = E.mod_mul(ctx, ms, a, b, mod_idx=idx)
            badm = [i for i in range(B) if gm[i] != (a[i] * b[i]) % mods[idx[i]]]
            print(f"modmul bits={bits} B={B}: mismatches={len(badm)} {badm[:8]}", flush=True)
            res[f"{bits}_{B}"] = (len(bad), len(badm))
    # timing: 4096-bit modulus, 2048-bit exponent (Paillier r^N mod N^2)
    for bits, ebits, B in ((4096, 2048, 16384), (2048, 2048, 32768), (2048, 1024, 32768)):
        mods = [rnd_mod(r, bits) for _ in range(16)]
        ms = E.ModSet(ctx, bits, mods)
        k32 = bits // 32
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        d_base = torch.randint(-2**31, 2**31 - 1, (B, k32), dtype=torch.int32, device="cuda", generator=g)
        d_exp = torch.randint(-2**31, 2**31 - 1, (B, ebits // 32), dtype=torch.int32, device="cuda", generator=g)
        d_idx = (torch.arange(B, device="cuda", dtype=torch.int32) % 16).contiguous()
        d_out = torch.empty_like(d_base)
        E.modexp_device(ctx, ms, d_base, d_exp, d_out, d_idx); torch.cuda.synchronize()
        t0 = time.time()
        E.modexp_device(ctx, ms, d_base, d_exp, d_out, d_idx); torch.cuda.synchronize()
        dt = time.time() - t0
        k = k32
        macs = (ebits + ebits // 4 + 16) * (2 * k * k + k)
        print(f"timing bits={bits} ebits={ebits} B={B}: {dt*1e3:.1f} ms  {B/dt:.0f} modexp/s  "
              f"{B/dt*macs/1e12:.2f} T alg-MAC/s  info={ctx.launch_info()}", flush=True)
        # spot check 4 items
        hb = words_to_ints(d_base[:4].cpu().numpy().view('uint32')); he = words_to_ints(d_exp[:4].cpu().numpy().view('uint32'))
        ho = words_to_ints(d_out[:4].cpu().numpy().view('uint32'))
        print("  spot:", [ho[i] == pow(hb[i], he[i], mods[i % 16]) for i in range(4)], flush=True)

if __name__ == "__main__":
    main()
