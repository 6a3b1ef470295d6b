"""Model of the sliding-window schedule the pair kernel runs for a PUBLIC exponent (mpe_pairexp.h slide_window and the phase
machine around it): the same word-wise scan (highest set bit at or below `from`, a window of at most wb bits that ends in a
set bit), the table of odd powers built from x^2, the ladder with its pending-window bookkeeping (sw_lo, sw_val), and — on a
two-base ladder — the fixed 4-bit windows of the short second exponent interleaved on the same squarings.
Run:  python tools/model/sliding_model.py"""
import random


def clz32(x):
    return 32 - x.bit_length()


def slide_window(words, exp_words, frm, wb):
    """mirror of the device function: returns (lo, val) or (-1, 0)"""
    if frm >= exp_words * 32:
        frm = exp_words * 32 - 1
    if frm < 0:
        return -1, 0
    w = frm >> 5
    x = words[w] & (0xFFFFFFFF >> (31 - (frm & 31)))
    while x == 0:
        w -= 1
        if w < 0:
            return -1, 0
        x = words[w]
    hi = w * 32 + 31 - clz32(x)
    lo = max(hi - wb + 1, 0)
    q, sh = lo >> 5, lo & 31
    two = words[q]
    if q + 1 < exp_words:
        two |= words[q + 1] << 32
    v = (two >> sh) & ((1 << (hi - lo + 1)) - 1)
    tz = (v & -v).bit_length() - 1
    return lo + tz, v >> tz


def ladder(x, e, mod, wb, exp_words, x2=None, e2=0, exp2_words=0):
    """x^e [* x2^e2] mod `mod` with the kernel's operation sequence; returns (value, squarings, multiplications) or None when
    the kernel would fall back to fixed windows (no set bit, or the first window reaches below the second exponent)"""
    words = [(e >> (32 * i)) & 0xFFFFFFFF for i in range(exp_words)]
    dual = x2 is not None
    sw_lo, sw_val = slide_window(words, exp_words, exp_words * 32 - 1, wb)
    if sw_lo < 0 or (dual and sw_lo < 32 * exp2_words):
        return None
    sq = mul = 0
    # table of odd powers: x^2, then x^3, x^5, ... x^(2^wb - 1)
    xx = x * x % mod
    sq += 1
    tab = {1: x % mod}
    k = 1
    while k <= (1 << wb) - 3:
        tab[k + 2] = tab[k] * xx % mod
        k += 2
        mul += 1
    tab2 = None
    if dual:
        tab2 = [1]
        for _ in range(15):
            tab2.append(tab2[-1] * x2 % mod)
        mul += 14
    nwin2 = exp2_words * 8
    cur = tab[sw_val]
    b = sw_lo
    sw_lo, sw_val = slide_window(words, exp_words, b - 1, wb)
    ph = "SQ" if b else "FINAL"
    while ph != "FINAL":
        if ph == "SQ":
            cur = cur * cur % mod
            sq += 1
            b -= 1
        elif ph == "MUL1":
            cur = cur * tab[sw_val] % mod
            mul += 1
            sw_lo, sw_val = slide_window(words, exp_words, b - 1, wb)
        else:
            cur = cur * tab2[(e2 >> b) & 15] % mod
            mul += 1
        m1 = ph == "SQ" and b == sw_lo
        m2 = ph != "MUL2" and dual and (b & 3) == 0 and (b >> 2) < nwin2
        ph = "MUL1" if m1 else ("MUL2" if m2 else ("FINAL" if b == 0 else "SQ"))
    return cur, sq, mul


def run(iters=40, seed=3):
    rnd = random.Random(seed)
    worst = 0
    for it in range(iters):
        bits = rnd.choice([64, 512, 2047, 2048])
        ew = (bits + 31) // 32
        mod = rnd.getrandbits(2048) | 1
        x, x2 = rnd.getrandbits(2048) % mod, rnd.getrandbits(2048) % mod
        e = rnd.getrandbits(bits) | (1 << (bits - 1))
        if it % 5 == 0:
            e |= 1
        if it % 7 == 0:
            e = (1 << (bits - 1))                      # a single set bit
        if it % 11 == 0:
            e = (1 << bits) - 1                        # all ones
        wb = 6 if ew >= 48 else 5
        got = ladder(x, e, mod, wb, ew)
        assert got is not None and got[0] == pow(x, e, mod), "single-base ladder"
        e2 = rnd.getrandbits(256)
        got2 = ladder(x, e, mod, wb, ew, x2, e2, 8)
        if got2 is None:
            assert bits < 256 + wb + 1 or e < (1 << (256 + wb))
        else:
            assert got2[0] == pow(x, e, mod) * pow(x2, e2, mod) % mod, "two-base ladder"
        worst = max(worst, got[2])
    return worst


if __name__ == "__main__":
    print("sliding-window schedule reproduces pow() for", 40, "random / edge exponents; most multiplications:", run())
