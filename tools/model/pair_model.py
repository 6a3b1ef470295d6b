"""Bit-level model of the N-adic pair multiplication of mpe_pairexp.h (the exact lane algorithm: pass A with the
quotient digits kept, pass B on pre-loaded columns, one or two product streams), in pure Python ints.

Checks, for random operands and for operands with every limb at its lazy maximum, that
  * every 64-bit column accumulator stays below 2^64,
  * the pair returned represents  X * Y * R^-1  (mod N^2),
  * the lazily normalised limbs stay below 2^W + 2^12 and the values below 2N.

Layout as in the kernel: K = TPI*L limbs of W bits; lane t owns limbs [t*L, (t+1)*L).
Run:  python tools/model/pair_model.py
"""
import random

W, L = 29, 18
MASK = (1 << W) - 1


def to_limbs(x, K):
    return [(x >> (W * i)) & MASK for i in range(K)]


def from_limbs(l):
    return sum(v << (W * i) for i, v in enumerate(l))


def cios(c0, streams, n, n0inv, TPI, stats, keep_m=False, steps=None):
    """One CIOS pass.  c0: K initial column values (lane-distributed as limbs), streams: [(a_limbs, b_limbs), ...].
    steps: outer steps = limbs of the multipliers that are read (default K).  With steps = K - 1 the pass divides by
    2^(W (K-1)) instead of 2^(W K): legitimate whenever the multipliers' top limb is zero and 2^(W (K-1)) > 4 N (the
    2048-bit moduli: 71 x 29 = 2059 bits) -- the "71 steps" variant DESIGN.md section 9 lists, studied here before any kernel work."""
    K = L * TPI
    steps = K if steps is None else steps
    c = [[c0[t * L + i] for i in range(L)] for t in range(TPI)]
    ms = []
    for a, b in streams:
        assert all(v == 0 for v in b[steps:]), "multiplier limbs beyond the steps must be zero"
    for j in range(steps):
        for a, b in streams:
            for t in range(TPI):
                for i in range(L):
                    c[t][i] += a[t * L + i] * b[j]
        m = ((c[0][0] & 0xFFFFFFFF) * n0inv) & MASK
        ms.append(m)
        for t in range(TPI):
            for i in range(L):
                c[t][i] += m * n[t * L + i]
                stats['maxcol'] = max(stats['maxcol'], c[t][i])
        assert c[0][0] & MASK == 0
        for t in range(TPI):
            c[t][1] += c[t][0] >> W
            stats['maxcol'] = max(stats['maxcol'], c[t][1])
        pulled = [(c[t + 1][0] & MASK) if t + 1 < TPI else 0 for t in range(TPI)]
        for t in range(TPI):
            c[t] = c[t][1:] + [pulled[t]]
    # tail: local ripple, one cross-lane carry hand-off (lazy result)
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
    return (r, ms) if keep_m else r


def pairmul(X, Y, n, n0inv, kc, TPI, stats, sq, steps=None):
    """X, Y: pairs of limb lists (lazy).  Returns the pair of limb lists of X * Y * R^-1 (mod N^2), R = 2^(W steps)."""
    K = L * TPI
    steps = K if steps is None else steps
    x0, x1 = X
    y0, y1 = Y
    u, ms = cios([0] * K, [(x0, y0)], n, n0inv, TPI, stats, keep_m=True, steps=steps)
    # K_c + (R - 1 - m), limb-wise non-negative: R - 1 is all-ones over `steps` limbs, the digits beyond do not exist
    pre = [kc[i] + (MASK - ms[i] if i < steps else 0) for i in range(K)]
    if sq:
        z1 = cios(pre, [(x1, [2 * v for v in x0])], n, n0inv, TPI, stats, steps=steps)
    else:
        z1 = cios(pre, [(x0, y1), (x1, y0)], n, n0inv, TPI, stats, steps=steps)
    return u, z1


def finish(Z, N):
    """The kernel's output step: the pair means the INTEGER z0 + z1 N with z0 < 2N lazily, so reducing z0 by N carries 1
    into z1; z1 is then reduced modulo N.  Returns the canonical residue z0 + z1 N in [0, N^2)."""
    z0, z1 = from_limbs(Z[0]), from_limbs(Z[1])
    assert z0 < 2 * N + 2 and z1 < 2 * N + 2
    if z0 >= N:
        z0 -= N
        z1 += 1
    while z1 >= N:
        z1 -= N
    return z0 + z1 * N


def run(bits, TPI, iters, seed, stress, steps=None):
    rnd = random.Random(seed)
    K = L * TPI
    steps = K if steps is None else steps
    R = 1 << (W * steps)
    assert R > 4 << bits, "the Montgomery radix must exceed 4 N"
    while True:
        N = rnd.getrandbits(bits) | (1 << (bits - 1)) | 1
        if stress:
            N = (1 << bits) - 1 - 2 * rnd.getrandbits(8)       # every limb of N near its maximum
        if N % 2 == 1:
            break
    NN = N * N
    n = to_limbs(N, K)
    n0inv = (-pow(N, -1, 1 << W)) % (1 << W)
    kc = to_limbs((-(R - 1)) % N, K)
    Rinv = pow(R, -1, NN)
    stats = {'maxcol': 0, 'maxlimb': 0}
    lazy = MASK + (1 << 12) - 1

    def val(P):
        return (from_limbs(P[0]) + from_limbs(P[1]) * N) % NN

    for it in range(iters):
        if stress:
            # lazily normalised operands with EVERY limb at the bound the kernel guarantees (values just under 2N are
            # not reachable with all limbs maximal, so this over-approximates the real worst case)
            top = (2 * N - 1).bit_length()
            X = ([lazy if W * i < top - W else 0 for i in range(K)], [lazy if W * i < top - W else 0 for i in range(K)])
            Y = X if it % 2 == 0 else (list(X[1]), list(X[0]))
        else:
            X = (to_limbs(rnd.randrange(2 * N), K), to_limbs(rnd.randrange(2 * N), K))
            Y = (to_limbs(rnd.randrange(2 * N), K), to_limbs(rnd.randrange(2 * N), K))
        for sq in (True, False):
            Yp = X if sq else Y
            Z = pairmul(X, Yp, n, n0inv, kc, TPI, stats, sq, steps)
            assert val(Z) == val(X) * val(Yp) * Rinv % NN, "wrong residue"
            assert from_limbs(Z[0]) < 2 * N + (1 << (bits - 30)) and from_limbs(Z[1]) < 2 * N + (1 << (bits - 30)), "value bound"
    # the corner the final normalisation has to get right: the base N itself (x0 = N is the lazy form of 0)
    one = (to_limbs(1, K), to_limbs(0, K))
    XN = (to_limbs(N, K), to_limbs(0, K))                                    # the plain pair of the value N
    r2 = (R * R) % NN
    F_N = pairmul(XN, (to_limbs(r2 % N, K), to_limbs(r2 // N, K)), n, n0inv, kc, TPI, stats, False, steps)   # its form
    back = pairmul(F_N, one, n, n0inv, kc, TPI, stats, False, steps)
    assert finish(back, N) == N, "z0 >= N must carry into z1"
    assert finish(pairmul(pairmul(F_N, F_N, n, n0inv, kc, TPI, stats, True, steps), one, n, n0inv, kc, TPI, stats, False, steps), N) == 0
    assert stats['maxcol'] < (1 << 64), "column overflow"
    assert stats['maxlimb'] <= lazy, "lazy limb bound"
    return stats


if __name__ == '__main__':
    import math
    for bits, TPI in ((2048, 4), (1024, 2)):
        for stress in (False, True):
            st = run(bits, TPI, 3 if bits == 2048 else 6, 7, stress)
            print(f"bits={bits} TPI={TPI} {'stress' if stress else 'random'}: max column 2^{math.log2(st['maxcol']):.3f}, "
                  f"max lazy limb 2^{math.log2(st['maxlimb']):.4f}")
    for stress in (False, True):                    # the 71-step variant for 2048-bit moduli (not in the kernel yet)
        st = run(2048, 4, 3, 11, stress, steps=71)
        print(f"bits=2048 TPI=4 steps=71 {'stress' if stress else 'random'}: max column 2^{math.log2(st['maxcol']):.3f}, "
              f"max lazy limb 2^{math.log2(st['maxlimb']):.4f}")
