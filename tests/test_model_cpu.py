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


def test_issue_budget_the_multiplier_issue_rate_explains_the_dominant_launch():
    """round-4 review item 3 ("<= 800 ms or a committed model + ISA count showing why not"): the instruction mix counted on the shipped kernel's
    ISA (profiles/r05/isa_census_pair2048.json), priced at the measured issue rates (profiles/r01_valu_rate.json) and the measured clock
    (profiles/r05/pmc_traffic.json), predicts the launch within a few per cent — and the measurement is FASTER than the prediction: nothing is left
    to gain from latency hiding or occupancy, only from executing fewer instructions (tools/model/issue_budget.py)."""
    import json
    ib = _load("issue_budget")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = json.load(open(os.path.join(root, "profiles", "r05", "isa_census_pair2048.json")))
    pmc = json.load(open(os.path.join(root, "profiles", "r05", "pmc_traffic.json")))
    k = pmc["kernels"][pmc["dominant_kernel"]]
    rates = {(r["instr"], r["waves_per_simd"]): r["wall_cycles_per_instr_per_simd_at_2.4GHz"]
             for r in json.load(open(os.path.join(root, "profiles", "r01_valu_rate.json")))["results"]}
    c_mad = rates[("mad_u64_u32_vv", 2)]
    assert 5.0 < c_mad < 5.1 and 2.2 < rates[("add_u32", 2)] < 2.5 and 4.2 < rates[("mul_lo_u32", 2)] < 4.5
    # the hot loops are what the source says they are: 36 MACs per CIOS step in a one-stream pass, 54 in the two-stream pass, ~5 others
    for b in c["pass_one_stream"]:
        assert b["mad"] == 2 * 18 * 18 and 5.0 < (b["valu"] - b["mad"]) / 18 < 5.3
    assert c["pass_two_streams"]["mad"] == 3 * 18 * 18 and (c["pass_two_streams"]["valu"] - 972) / 18 < 6.0
    lad = ib.per_ladder(c)
    items = 655360
    measured_valu = k["sq_per_launch"]["SQ_INSTS_VALU"] / (items / 16)
    assert abs(lad["valu"] / measured_valu - 1) < 0.015                      # the census accounts for the VALU instructions the counters saw
    p = ib.predict(c, items, k["effective_clock_GHz"], c_mad=c_mad)
    assert p["trips"] == 20.0
    ratio = k["avg_ms"] / (p["seconds"] * 1e3)
    assert 0.95 < ratio < 1.0                                                # measured: 2 % faster than the issue-rate model
    assert p["mad_share_of_issue_time"] > 0.88
    # even with NO instruction other than the multiplies the launch would take >= 0.89 of today's time
    floor = ib.predict(c, items, k["effective_clock_GHz"], c_mad=c_mad, c_other=0.0)["seconds"] * 1e3
    assert floor / k["avg_ms"] > 0.89


def test_lone_ladder_model_prices_the_delayed_quotient_before_anyone_builds_it():
    """round-5 review item 1b (model first): unit times of lone waves over the instruction census of the shipped loops give ONE cadence —
    ~5.5 cycles per issued instruction — for the 18- and the 9-limb layout (they agree within 2 %: issue-bound, the quotient chain hidden);
    only 5 limbs x 16 lanes waits (~16 %).  The bare chains measured by tools/ubench/chain_latency.hip are a third of a step.  So Orup's
    quotient buys nothing at 9 limbs and at best ~15 % at 5, a look-ahead quotient loses everywhere: <= ~6 ms of a 122 ms lone batch, against
    the 27 ms that 95 ms needs"""
    m = _load("lone_ladder_model")
    t = {r["limbs_per_lane"]: r for r in m.table()}
    assert abs(m.cadence(18) / m.cadence(9) - 1) < 0.02 and 5.3 < m.lone_cadence() < 5.7
    assert t[18]["waiting_share"] < 0.01 and t[9]["waiting_share"] < 0.02 and 0.12 < t[5]["waiting_share"] < 0.20
    for r in t.values():
        assert r["bare_chain_cycles"] < 0.45 * r["cycles_per_step"]                  # the chain alone never fills a step
    assert t[9]["orup_gain_at_best"] == 0.0 and 0.10 < t[5]["orup_gain_at_best"] < 0.20
    assert t[9]["lookahead_unit_ms_at_best"] > t[9]["unit_ms"] and t[5]["lookahead_unit_ms_at_best"] > t[5]["unit_ms"]
    assert m.CHAIN["orup"][16] < m.CHAIN["lookahead"][16] < m.CHAIN["shipped"][16]
    assert 4.0 < m.lone_batch_gain_ms() < 8.0 < 122.1 - 95.0
    # the census the model rests on is the shipped build's, when its ISA dump is around (build/ is not in the repository)
    import os
    import re
    isa = os.path.join(ROOT, "build", "mpe_pair2048-hip-amdgcn-amd-amdhsa-gfx950.s")
    if os.path.exists(isa):
        text = open(isa).read()
        for L, lanes in ((18, 4), (9, 8), (5, 16)):
            start = text.index("_ZN3mpe18pair_modexp_kernelINS_3CfgILi2048ELi29ELi%dELi%dEEELb1EEE" % (L, lanes))
            body = text[start:text.index("s_endpgm", start)]
            blocks = [[l for l in b.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))] for b in re.split(r"\n\.LBB\d+_\d+:", body)]
            loops = sorted(len(b) for b in blocks if sum("v_mad_u64_u32" in l for l in b) >= 2 * L * L)[:3]
            want = sorted(round(x * L) for x in m.LAYOUTS[L]["insts"])
            assert all(abs(a - b) <= 0.03 * b for a, b in zip(loops, want)), (L, loops, want)


def test_scheduling_model_static_units_against_the_unit_queue():
    """why the headline moved in round 6 with no change to the arithmetic: one SIMD, two ladder waves, the arbiter's measured preference for the
    older one (profiles/r06/wave_trace*.jsonl).  With STATIC units the older wave finishes its half at 0.6 of the launch and leaves; with the
    unit QUEUE it keeps pulling — 30 of a SIMD's 40 units — and the launch ends 6 % earlier (measured: 1 234 -> 1 144 ms, 7.3 %)"""
    m = _load("sched_model")
    t = {r["units_per_simd"]: r for r in m.table()}
    for u in (4, 40):
        assert abs(t[u]["static_ms"] / t[u]["static_measured_ms"][0] - 1) < 0.05 and abs(t[u]["queue_ms"] / t[u]["queue_measured_ms"] - 1) < 0.06
    assert 0.05 < t[40]["gain"] < 0.08 and t[40]["units_run_by_the_older_wave"] >= 29
    assert 88.0 <= t[3]["static_ms"] <= 96.0
    assert 1.15 < (m.R_O + m.R_Y) / m.R_L < 1.25                                  # two waves deliver ~1.2x the units of one
    assert m.simulate(1, True)[0] == m.LONE and m.simulate(2, True)[0] > m.simulate(1, True)[0]


def test_orup_model_the_residues_are_right_and_the_values_are_29_bits_too_large():
    """the delayed-free quotient digit (q = low limb, no multiplication) written as a bit-level model BEFORE any kernel: the pair residues come
    out right with the SAME Montgomery radix, the columns stay far below 2^64 — and the results are bounded by N N', not by 2N (Orup reduces
    modulo N~): the 9-limb layout of the 1024-bit halves overflows its 36 limbs, the others would need R = r^(n+1) and a plain reduction
    before the final normalisation.  The finding that kept the kernel unwritten this round (tools/model/orup_model.py)."""
    m = _load("orup_model")
    for case in m.CASES:
        st = m.run(*case, 2, 5, False)
        assert st["maxcol"] < (1 << 62)
        assert (1 << 20) < st["maxval_over_N"] < (1 << 29)                      # ~ N' times larger than a Montgomery result
    assert m.overflows()
