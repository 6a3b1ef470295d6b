"""Bit-level model of the N-adic pair multiplication with a DELAYED-FREE quotient digit (Orup 1995, delay 0) for the lane layouts whose lone
waves wait for their own quotient-digit chain (9 and 5 limbs per lane: tools/model/lone_ladder_model.py) — the algorithm of
mpe_pairexp.h `cios1o` / `cios2o`, in pure Python ints, written and checked BEFORE the kernel.

Plain CIOS (what the 18-limb layout keeps):   c += a * b_j;  m = lo(c) * n0inv mod r;  c += m * n;  c >>= W
    chain per step: mad(a0 b_j) -> v_mul_lo -> DPP broadcast -> mad(m n0) -> shift -> add -> next mad            (6 - 9 links)
Here:   N~ = N * N' with N' = -N^-1 mod r, so N~ = -1 (mod r);  Np = (N~ + 1) / r  (K limbs, < N).  Per step
    q = lo(c) mod r  (NO multiplication);   c = (c >> W) + q * Np + b_j * a
    chain per step: v_and / DPP broadcast -> mad(q Np0) -> next                                                 (2 - 4 links)
Exactness: r * S_(i+1) = S_i + q_i N~ + b_i A r, so after n + 1 steps with b_n = 0 the pass returns S = (c_in / r + A B + Q N~ / r) / r^n:
the SAME Montgomery radix R = r^n as the n-step CIOS (n = 71 for 2048-bit moduli, 36 for the 1024-bit halves), and the residues are right
(checked below).  **What the model found before any kernel was written: the VALUES are not.**  Every step adds q * Np with q < r and
Np ~ N N' / r, so S settles at ~ r (Np + A) / ... ~ N N': the results are bounded by N~ = N N' (29 bits more than N), not by 2N — Orup's
method reduces modulo N~.  Consequences for this engine: operands of K limbs no longer fit the 1024-bit halves at 9 limbs per lane
(36 x 29 = 1 044 bits < 1 024 + 29 + 2), R has to grow to r^(n+1) > 4 N~ — other per-key constants than the 18-limb layout's, so launches of
different layouts could not share them — and the exact normalisation at the end of an exponentiation needs one plain CIOS reduction first.
That is the "one more limb of R" of the literature, and it is what tools/model/lone_ladder_model.py prices (+1 limb per lane).  The timing side
(tools/ubench/chain_latency.hip): the chain falls from 51 / 65 to 31 / 43 cycles per step, but a lone wave is bound by ~5.5 cycles per issued
instruction and the instruction count does not fall — at most ~6 ms of a 122 ms lone batch.  The kernel was NOT written.

The pair arithmetic needs the integer M with x0 y0 + M N = u R.  Pass A starts from c = 0, hence q_0 = 0 and M = (Q / r) N' where
Q / r = q_1 + q_2 r + ... — not a digit string any more.  Pass B therefore adds, at step i, the single product  (r - q_(i+1)) * N'  to lane 0's
incoming low column (one more MAC per step with a lane-masked multiplier: the stored digits are r - q) and starts from the per-key constant
K_c2 = -(D2 r) mod N with D2 = N' r (r^(n-1) ... + r + 1) restricted to the digits that exist:  sum_i (r - q_(i+1)) N' r^i = D2 - M.

Checks (run(), tests/test_model_cpu.py): residues modulo N^2, 64-bit column bound; the value bound is REPORTED (it is the finding).
Run: python tools/model/orup_model.py
"""
import random

W = 29
MASK = (1 << W) - 1
RADIX = 1 << W


def to_limbs(x, K):
    return [(x >> (W * i)) & MASK for i in range(K)]


def from_limbs(l):
    return sum(v << (W * i) for i, v in enumerate(l))


def tail(c, L, TPI, stats):
    K = L * TPI
    r = [0] * K
    couts = []
    for t in range(TPI):
        carry = 0
        for i in range(L):
            v = c[t][i] + carry
            stats['maxcol'] = max(stats['maxcol'], v)
            r[t * L + i] = v & MASK
            carry = v >> W
        couts.append(carry)
    assert couts[-1] == 0, "top carry must vanish"
    for t in range(1, TPI):
        cin = couts[t - 1]
        v0 = r[t * L] + (cin & MASK)
        r[t * L] = v0 & MASK
        r[t * L + 1] += (cin >> W) + (v0 >> W)
    stats['maxlimb'] = max(stats['maxlimb'], max(r))
    return r


def cios_orup(c0, streams, npl, L, TPI, stats, steps, keep_q=False, qbar=None, nprime=0):
    """One pass.  c0: K initial columns; streams: [(a_limbs, b_limbs)]; npl: limbs of Np; steps = n + 1 (the multipliers' limbs from n on are
    zero or do not exist).  qbar: pass B — the stored digits r - q_(i+1) of pass A, added times N' to lane 0's incoming low column."""
    K = L * TPI
    c = [[c0[t * L + i] for i in range(L)] for t in range(TPI)]
    qs = []
    for j in range(steps):
        q = c[0][0] & MASK                                         # lane 0's low limb: THE quotient digit, no multiplication
        qs.append(q)
        # shift: every lane keeps the high part of its lowest column, passes the low W bits to the lane below
        pulled = [(c[t + 1][0] & MASK) if t + 1 < TPI else 0 for t in range(TPI)]
        for t in range(TPI):
            c[t][1] += c[t][0] >> W
            c[t] = c[t][1:] + [pulled[t]]
        # products of this step land AFTER the shift
        for a, b in streams:
            bj = b[j] if j < len(b) else 0
            for t in range(TPI):
                for i in range(L):
                    c[t][i] += a[t * L + i] * bj
        for t in range(TPI):
            for i in range(L):
                c[t][i] += q * npl[t * L + i]
        if qbar is not None and j < len(qbar):
            c[0][0] += qbar[j] * nprime
        for t in range(TPI):
            for i in range(L):
                stats['maxcol'] = max(stats['maxcol'], c[t][i])
    r = tail(c, L, TPI, stats)
    return (r, qs) if keep_q else r


def pairmul_orup(X, Y, npl, nprime, kc2, L, TPI, stats, sq, steps):
    K = L * TPI
    x0, x1 = X
    y0, y1 = Y
    u, qs = cios_orup([0] * K, [(x0, y0)], npl, L, TPI, stats, steps, keep_q=True)
    assert qs[0] == 0
    qbar = [RADIX - q for q in qs[1:]]                              # steps - 1 digits, each in [1, r]
    if sq:
        z1 = cios_orup(list(kc2), [(x1, [2 * v for v in x0])], npl, L, TPI, stats, steps, qbar=qbar, nprime=nprime)
    else:
        z1 = cios_orup(list(kc2), [(x0, y1), (x1, y0)], npl, L, TPI, stats, steps, qbar=qbar, nprime=nprime)
    return u, z1


def constants(N, L, TPI, n):
    """n = limbs the multipliers really have (R = r^n); returns (Np limbs, N', K_c2 limbs)"""
    K = L * TPI
    nprime = (-pow(N, -1, RADIX)) % RADIX
    Nt = N * nprime
    assert (Nt + 1) % RADIX == 0
    Np = (Nt + 1) // RADIX
    assert Np < N
    D2 = nprime * RADIX * sum(RADIX ** i for i in range(n))          # sum_i r * N' * r^i over the n stored digits
    kc2 = (-(D2 * RADIX)) % N
    return to_limbs(Np, K), nprime, to_limbs(kc2, K)


def run(bits, L, TPI, n, iters, seed, stress):
    """n: R = r^n; the pass runs n + 1 steps"""
    rnd = random.Random(seed)
    K = L * TPI
    R = 1 << (W * n)
    assert R > 4 << bits and K * W >= bits + 2
    N = rnd.getrandbits(bits) | (1 << (bits - 1)) | 1
    if stress:
        N = (1 << bits) - 1 - 2 * rnd.getrandbits(8)
    NN = N * N
    npl, nprime, kc2 = constants(N, L, TPI, n)
    Rinv = pow(R, -1, NN)
    stats = {'maxcol': 0, 'maxlimb': 0}
    lazy = MASK + (1 << 12) - 1

    def val(P):
        return (from_limbs(P[0]) + from_limbs(P[1]) * N) % NN

    for it in range(iters):
        if stress:
            top = (2 * N - 1).bit_length()
            X = ([lazy if W * i < top - W else 0 for i in range(K)], [lazy if W * i < top - W else 0 for i in range(K)])
            Y = X if it % 2 == 0 else (list(X[1]), list(X[0]))
        else:
            X = (to_limbs(rnd.randrange(2 * N), K), to_limbs(rnd.randrange(2 * N), K))
            Y = (to_limbs(rnd.randrange(2 * N), K), to_limbs(rnd.randrange(2 * N), K))
        for sq in (True, False):
            Yp = X if sq else Y
            Z = pairmul_orup(X, Yp, npl, nprime, kc2, L, TPI, stats, sq, n + 1)
            assert val(Z) == val(X) * val(Yp) * Rinv % NN, "wrong residue"
            stats['maxval_over_N'] = max(stats.get('maxval_over_N', 0), from_limbs(Z[0]) // N, from_limbs(Z[1]) // N)
    # half mode (plain Montgomery on the x0 components): pass A alone
    a, b = rnd.randrange(2 * N), rnd.randrange(2 * N)
    u = cios_orup([0] * K, [(to_limbs(a, K), to_limbs(b, K))], npl, L, TPI, stats, n + 1)
    assert from_limbs(u) % N == a * b * pow(R, -1, N) % N
    assert stats['maxcol'] < (1 << 64), "column overflow"
    stats['nprime'] = nprime
    return stats


CASES = ((2048, 9, 8, 71), (2048, 5, 16, 71), (1024, 5, 8, 36))
OVERFLOWS = (1024, 9, 4, 36)          # 36 limbs x 29 bits cannot hold results of ~ N N': the pass loses its top carry


def overflows(case=OVERFLOWS):
    try:
        run(*case, 2, 5, False)
    except AssertionError as e:
        return "top carry" in str(e)
    return False


if __name__ == '__main__':
    import math
    for bits, L, TPI, n in CASES:
        st = run(bits, L, TPI, n, 3, 5, False)
        print(f"bits={bits} L={L} TPI={TPI} steps={n + 1}: residues right; max column 2^{math.log2(st['maxcol']):.3f}; results up to "
              f"{st['maxval_over_N']} x N = 2^{math.log2(max(1, st['maxval_over_N'])):.1f} N  (N' = 2^{math.log2(st['nprime']):.1f})")
    print("bits=1024 L=9 TPI=4: the K = 36 limbs overflow:", overflows())
