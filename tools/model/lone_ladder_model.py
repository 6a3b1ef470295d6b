"""What a LONE ladder wave is bound by, and what a shorter quotient chain (Orup's delayed-free quotient, a look-ahead quotient) could buy —
priced BEFORE building anything (round-5 review, item 1b: "bit-level model + CPU test before the kernel").

Measured inputs
  * unit times of lone waves (s_memrealtime around every (wave, unit) of pair_modexp_kernel, one wave per SIMD: profiles/r06/wave_trace*.jsonl,
    ab_xwide13.jsonl; one unit = a 2048-bit two-base ladder modulo N^2 = 2 044 squarings + 406 multiplications of pairs, 71 CIOS steps per pass):
        18 limbs x 4 lanes  35.3 ms      9 x 8  23.4 ms      5 x 16  23.9 ms      at 2.37 GHz under lone waves
  * instructions per CIOS step in the three pass loops of the shipped build (tools/isa_blocks.py on build/mpe_pair2048-...gfx950.s: pass A with
    the digits stored, pass B with one product stream (squaring), pass B with two (multiplication); v_mad_u64_u32 / all):
        18 x 4   36 / 42.6   36 / 42.2   54 / 61.8         9 x 8   18 / 28.3   18 / 27.8   27 / 37.4         5 x 16   10 / 25.0   10 / 23.8   15 / 29.6
  * the latency of the quotient-digit chain by itself, a single wave running nothing else (tools/ubench/chain_latency.hip ->
    profiles/r06/chain_latency.json): shipped chain 36 / 51 / 65 cycles per step for 4 / 8 / 16 lanes per integer; a 29-bit look-ahead quotient
    38 / 50 (8 / 16 lanes); Orup's quotient 31 / 43; single links: v_mad_u64_u32 8.5, v_mul_lo_u32 8.0, s_nop 1 + DPP move 14.25, 64-bit shift or add 7.5.

Result 1 — a lone wave pays the same ~5.5 cycles for EVERY instruction it issues, multiply-add or not:
        cycles per instruction = unit x clock / (instructions of the unit)  =  5.46 (18 x 4)   5.52 (9 x 8)   6.54 (5 x 16)
  (r01_valu_rate.json has the same figure from the other side: a single resident wave issues a v_mad_u64_u32 every 5.66 cycles and a plain
  v_fma_f32 every 5.4, where a full SIMD needs 5.0 and 3.0.)  The 18- and 9-limb layouts are therefore ISSUE-bound at the lone wave's cadence; the
  quotient chain (36 / 51 cycles of a 240 / 159-cycle step) is hidden behind the other products.  Only 5 x 16 waits: 16 % above the cadence — its
  65-cycle chain has just four independent multiply-adds to hide behind.  (An earlier version of this file read the equal step times of the 9- and
  5-limb layouts, ~160 cycles, as "the chain is 160 cycles long".  The micro-benchmark says the chain is 51 / 65; what is equal is the
  instruction count: 28 against 25 per step.)

Result 2 — what shortening the chain can buy is bounded by the waiting share, and costs instructions:
  * Orup (digit = the accumulator's low limb; tools/model/orup_model.py): minus v_mul_lo_u32 (and the mask where it is not folded into a DPP
    move), plus one more limb per lane (2 multiply-adds per step) and one more step: +0 ... +1 instruction per step.  9 x 8: no gain (it does not
    wait).  5 x 16: at best its 16 % of waiting, 23.9 -> ~20.5 ms — IF the 43-cycle chain then hides, with one more limb's products to hide behind.
  * look-ahead quotient (a 32-bit shadow of the low column scaled by 8; chain v_mul_lo -> hops -> v_mul_hi -> v_lshl_add): +7 instructions per step
    = +38 cycles at the lone cadence for a chain 13 - 15 cycles shorter: a loss everywhere.
  On the lone 1 024-session batch (122 ms) the launches that could run on 5 x 16 are round 5's PDL verification (23.4 + 7.0 ms) and round 4's
  halves (10.9 ms): 41 ms x 16 % = at most ~6 ms.  95 ms is not reachable by shortening the chain; NOT built.
  What WOULD shorten a lone ladder is fewer instructions per step per wave — and the non-multiply overhead is already 10 - 15 of the 25 - 28.
tests/test_model_cpu.py keeps the arithmetic honest.  Run: python tools/model/lone_ladder_model.py
"""
import json

STEPS = 71
SQUARINGS, MULTIPLICATIONS = 2044, 406            # one two-base ladder with 2048- and 256-bit exponents, sliding windows
GHZ = 2.37
LAYOUTS = {  # limbs per lane: lanes per integer, lone unit ms, instructions per step (all, v_mad_u64_u32) in pass A / pass B one stream / pass B two streams
    18: {"lanes": 4, "unit_ms": 35.3, "insts": (767 / 18, 760 / 18, 1113 / 18), "mads": (36, 36, 54)},
    9: {"lanes": 8, "unit_ms": 23.4, "insts": (255 / 9, 250 / 9, 337 / 9), "mads": (18, 18, 27)},
    5: {"lanes": 16, "unit_ms": 23.9, "insts": (125 / 5, 119 / 5, 148 / 5), "mads": (10, 10, 15)},
}
CHAIN = {  # lanes per integer: cycles per step of the bare quotient chain — profiles/r06/chain_latency.json
    "shipped": {4: 36.0, 8: 51.0, 16: 65.0}, "lookahead": {8: 38.25, 16: 50.25}, "orup": {8: 31.25, 16: 43.25}}
LOOKAHEAD_EXTRA_INSTS = 7                          # v_mul_hi, v_mul_lo (n1), shift of the digit, v_alignbit, v_cmp, v_addc, 2 x v_lshl_add - the mask
ORUP_EXTRA_INSTS = {8: 2 - 1, 16: 2 - 2}           # one more limb: + 2 multiply-adds; - v_mul_lo, - the mask (16 lanes; folded into a DPP move at 8)


def insts_per_unit(L, extra_per_step=0.0, steps=STEPS):
    a, b1, b2 = LAYOUTS[L]["insts"]
    return steps * (SQUARINGS * (a + b1 + 2 * extra_per_step) + MULTIPLICATIONS * (a + b2 + 2 * extra_per_step))


def cadence(L):
    """cycles per issued instruction of a lone wave, measured: unit time x clock / instructions of the unit (the loops' instructions only:
    the code between the passes is ~5 % more and is left out on purpose — the same omission for every layout)"""
    return LAYOUTS[L]["unit_ms"] * 1e-3 * GHZ * 1e9 / insts_per_unit(L)


def lone_cadence():
    """the two layouts that do not wait agree within 2 %: that IS the lone wave's cadence"""
    return 0.5 * (cadence(18) + cadence(9))


def table():
    c0 = lone_cadence()
    out = []
    for L, d in LAYOUTS.items():
        a, b1, b2 = d["insts"]
        step = d["unit_ms"] * 1e-3 * GHZ * 1e9 / ((SQUARINGS + MULTIPLICATIONS) * 2 * STEPS)
        waits = max(0.0, cadence(L) / c0 - 1.0)
        row = {"limbs_per_lane": L, "lanes_per_integer": d["lanes"], "unit_ms": d["unit_ms"], "cycles_per_step": round(step, 1),
               "instructions_per_step": round((SQUARINGS * (a + b1) + MULTIPLICATIONS * (a + b2)) / (2.0 * (SQUARINGS + MULTIPLICATIONS)), 1),
               "cycles_per_instruction": round(cadence(L), 2), "waiting_share": round(waits / (1.0 + waits), 3),
               "bare_chain_cycles": CHAIN["shipped"][d["lanes"]]}
        if d["lanes"] in CHAIN["orup"]:
            # Orup: the instruction count moves by ORUP_EXTRA_INSTS per step and one step is added; the waiting share is what it can remove at best
            orup_issue = insts_per_unit(L, ORUP_EXTRA_INSTS[d["lanes"]], STEPS + 1) * c0
            row["orup_unit_ms_at_best"] = round(orup_issue / (GHZ * 1e6), 2)
            row["orup_gain_at_best"] = round(1.0 - min(1.0, row["orup_unit_ms_at_best"] / d["unit_ms"]), 3)
            look_issue = insts_per_unit(L, LOOKAHEAD_EXTRA_INSTS) * c0
            row["lookahead_unit_ms_at_best"] = round(look_issue / (GHZ * 1e6), 2)
        out.append(row)
    return out


def lone_batch_gain_ms(stretches_5_limb_ms=(23.4, 7.0, 4.5 + 6.4)):
    """the lone-ladder stretches of one 1 024-session batch whose launches (<= 4 096 items) could run on the 5-limb layout — round 5's PDL
    verification ladder and its short companion, round 4's halves (profiles/r06/final_lib_timeline_lone_1024.json, ab_xwide13.jsonl) — times
    the most Orup could take off a 5-limb ladder; the 9- and 18-limb stretches do not wait, nothing to gain there"""
    g = next(r for r in table() if r["limbs_per_lane"] == 5)["orup_gain_at_best"]
    return sum(stretches_5_limb_ms) * g


if __name__ == "__main__":
    print(json.dumps({"lone_wave_cycles_per_instruction": round(lone_cadence(), 2), "table": table(),
                      "lone_1024_batch_gain_ms_at_best": round(lone_batch_gain_ms(), 2), "lone_1024_batch_ms": 122.1, "asked_for_ms": 95.0}, indent=1))
