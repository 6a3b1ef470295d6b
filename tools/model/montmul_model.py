"""Bit-level model of the wave-distributed reduced-radix Montgomery multiplication
used by the HIP modexp kernel.  Pure Python ints; checks the
column-accumulator bounds (everything must fit 64 bits) and the final value.

Layout: K = TPI*L limbs of W bits; thread t owns limbs [t*L, (t+1)*L).
Per outer step j: c[i] += a[i]*b_j ; m = (c_0 * n0inv) mod 2^W (thread 0) ; c[i] += m*n[i];
thread 0 folds c[0]>>W into c[1]; every thread pulls the next thread's c[0] as its new top column.
"""
import random, sys

def to_limbs(x, W, K):
    return [(x >> (W * i)) & ((1 << W) - 1) for i in range(K)]

def from_limbs(l, W):
    return sum(v << (W * i) for i, v in enumerate(l))

def montmul(a, b, n, n0inv, W, L, TPI, stats):
    K = L * TPI
    MASK = (1 << W) - 1
    c = [[0] * L for _ in range(TPI)]            # c[t][i] 64-bit accumulators
    for j in range(K):
        bj = b[j]
        # a*b
        for t in range(TPI):
            for i in range(L):
                c[t][i] += a[t * L + i] * bj
        m = ((c[0][0] & 0xFFFFFFFF) * n0inv) & MASK
        for t in range(TPI):
            for i in range(L):
                c[t][i] += m * n[t * L + i]
                stats['maxcol'] = max(stats['maxcol'], c[t][i])
        assert c[0][0] & MASK == 0
        # long lanes (L > 18 at W = 29): also split the column half-way down the lane
        if 2 * L * (1 << (2 * W)) * 1.02 >= (1 << 64):
            h = L // 2
            for t in range(TPI):
                c[t][h + 1] += c[t][h] >> W
                c[t][h] &= MASK
        # fold + shift: every lane keeps the high part of its lowest column, the low W bits
        # move to the lower neighbour lane as its new top column
        for t in range(TPI):
            c[t][1] += c[t][0] >> W
        pulled = [(c[t + 1][0] & MASK) if t + 1 < TPI else 0 for t in range(TPI)]
        for t in range(TPI):
            c[t] = c[t][1:] + [pulled[t]]
    # local ripple + one cross-thread carry hand-off (lazy result)
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
        v = r[t * L] + (cin & MASK)
        r[t * L] = v & MASK                       # one extra ripple step keeps every limb < 2^W + 2^12
        r[t * L + 1] += (cin >> W) + (v >> W)
    stats['maxlimb'] = max(stats['maxlimb'], max(r))
    return r

def run(bits, W, L, TPI, iters=20, seed=1):
    rnd = random.Random(seed)
    K = L * TPI
    R = 1 << (W * K)
    stats = {'maxcol': 0, 'maxlimb': 0}
    for it in range(iters):
        N = rnd.getrandbits(bits) | (1 << (bits - 1)) | 1
        if it == 0:
            N = (1 << bits) - 1            # worst-case all-ones modulus
        n = to_limbs(N, W, K)
        n0inv = (-pow(N, -1, 1 << W)) % (1 << W)
        x = rnd.randrange(2 * N) if it else 2 * N - 1
        y = rnd.randrange(2 * N) if it else 2 * N - 1
        a = to_limbs(x, W, K); b = to_limbs(y, W, K)
        # chain several multiplications feeding lazy outputs back in
        for rep in range(4):
            r = montmul(a, b, n, n0inv, W, L, TPI, stats)
            val = from_limbs(r, W)
            assert val < 2 * N, "Montgomery bound"
            assert (val * R - from_limbs(a, W) * from_limbs(b, W)) % N == 0
            a, b = r, r
    print(f"bits={bits} W={W} L={L} TPI={TPI} K={K}: maxcol=2^{stats['maxcol'].bit_length()} "
          f"maxlimb=2^{stats['maxlimb'].bit_length()}  OK")
    assert stats['maxcol'] < 1 << 64
    return stats

if __name__ == '__main__':
    run(4096, 27, 19, 8, iters=4)
    run(2048, 27, 19, 4, iters=6)
    run(4096, 29, 18, 8, iters=4)
    run(2048, 29, 18, 4, iters=6)
    run(4096, 29, 36, 4, iters=4)
    run(2048, 29, 36, 2, iters=6)
