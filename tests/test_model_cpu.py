"""The bit-level Python models of the lane algorithms (tools/model/): column accumulators must stay below 2^64 and the
results must be the right residues, for random operands and with every limb at its lazy maximum."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "model", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pair_multiplication_model_bounds_and_residues():
    pm = _load("pair_model")
    for bits, tpi in ((2048, 4), (1024, 2)):
        for stress in (False, True):
            st = pm.run(bits, tpi, 2, 11, stress)
            assert st["maxcol"] < 1 << 64
            assert st["maxlimb"] < (1 << 29) + (1 << 12)


def test_montmul_model_bounds():
    mm = _load("montmul_model")
    for bits, w, l, tpi in ((4096, 29, 18, 8), (2048, 29, 18, 4)):
        st = mm.run(bits, w, l, tpi, iters=2)
        assert st["maxcol"] < 1 << 64


def test_pair_model_with_71_steps_for_2048_bit_moduli():
    """The variant DESIGN.md section 9 lists as a next step — 71 CIOS steps instead of 72 when the modulus has 2048 bits
    (R = 2^(29 * 71) still exceeds 4 N and the multipliers' 72nd limb is zero) — is correct and within the same bounds in the
    model: right residues modulo N^2, columns below 2^64, lazy limbs below 2^29 + 2^12, the z0 >= N corner.  Not in the kernel."""
    pm = _load("pair_model")
    for stress in (False, True):
        st = pm.run(2048, 4, 2, 5, stress, steps=71)
        assert st["maxcol"] < 1 << 64
        assert st["maxlimb"] < (1 << 29) + (1 << 12)


def test_sliding_window_schedule_of_the_public_exponent():
    """the schedule the pair kernel runs when the exponent is the PUBLIC key N (mpe_pairexp.h slide_window + phase machine):
    word-wise window scan, odd-power table from x^2, pending-window bookkeeping, a short second exponent on fixed 4-bit windows
    riding the same squarings — reproduces pow() for random and edge exponents (single bit, all ones, 64..2048 bits)"""
    sm = _load("sliding_model")
    assert sm.run(iters=60, seed=9) > 0
    # and the averages bench.py prices the launches with: a 2048-bit exponent on 6-bit windows
    import random
    rnd = random.Random(1)
    e = rnd.getrandbits(2048) | (1 << 2047) | 1
    val, sq, mul = sm.ladder(3, e, (1 << 2048) - 159, 6, 64)
    assert val == pow(3, e, (1 << 2048) - 159)
    assert 2040 <= sq <= 2049 and 31 + 270 <= mul <= 31 + 315         # vs 2046 squarings + 62 + 342 multiplications on fixed windows


def test_karatsuba_on_the_product_halves_does_not_pay_in_this_lane_layout():
    """round-4 review item 3a: price one level of Karatsuba on the product halves of pass A / pass B with the real lane layout and build
    it only if the model says >= 6 %.  tools/model/karatsuba_model.py: the 306 MACs per lane it saves are spent on re-laying the halves
    out over the four lanes, re-normalising the 30-bit sums (the middle product would overflow the 64-bit columns), combining 64-bit
    columns with carry pairs and folding the held product into the reduction — and the held product needs 36 more VGPRs at 255 of 256."""
    km = _load("karatsuba_model")
    p = km.price()
    assert abs(p["today"]["total"] - 71 * 740 / 18) < 1e-6                      # the measured loop body: 740 VALU per 18 steps
    assert p["karatsuba_relayout"]["product_mac"] == 3 * 36 * 9 < 71 * 18       # the product half does get cheaper ...
    assert p["pass_gain"] < 0.06 and p["squaring_gain"] < 0.06                  # ... the pass does not: below the bar (in fact a loss)
    assert p["pass_gain_no_relayout"] < p["pass_gain"]                          # leaving two lanes idle is worse still
    assert p["karatsuba_relayout"]["vgpr_columns_held"] - 2 * km.L == 36
