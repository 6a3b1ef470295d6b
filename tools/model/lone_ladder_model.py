"""What a LONE ladder wave is bound by, and what a delayed quotient (Orup) could buy — priced BEFORE building anything
(round-5 review, item 1b: "bit-level model + CPU test before the kernel").

Measured inputs (profiles/r06/wave_trace*.jsonl: s_memrealtime around every (wave, unit) of pair_modexp_kernel, waves alone on their SIMD;
profiles/r01_valu_rate.json: issue cycles of a single resident wave):

  layout (limbs per lane x lanes per integer)   18 x 4      9 x 8       5 x 16
  unit = one 2048-bit two-base ladder mod N^2   35.3 ms     23.4 ms     23.9 ms   (2 450 multiplications mod N^2 = 4 900 passes of 71 steps)
  clock under lone waves                        2.37 GHz    2.37 GHz    2.37 GHz
  (5 x 16: profiles/r06/ab_xwide13.jsonl, with the final scheduler.  An earlier 43.8 ms was a primary running two units in a row.)

Per CIOS step a lane issues  2 L  v_mad_u64_u32 (L products a_i b_j, L products m n_i; one-stream pass) and ~5.1 other VALU instructions
(the quotient digit's v_mul_lo_u32, its DPP broadcast, the 64-bit shift + add of the finished column, the DPP pull of the neighbour's limb).
A single resident wave issues a v_mad_u64_u32 every 5.66 cycles and the cheap class every ~4.2.  So

    issue(L)  = 2 L x 5.66 + 5.1 x 4.2              cycles per step if nothing ever waits
    step(L)   = unit x clock / (4 900 x 71)          cycles per step measured
    exposed   = step - issue                         cycles per step in which the wave waits for its OWN previous result:
                                                     the quotient-digit chain  mad(a0 b_j) -> mul_lo -> DPP -> DPP -> mad(m n) -> shift -> add -> mad

Both small layouts sit at ~160 cycles per step whatever they issue: THE CHAIN IS ~160 CYCLES (8 links of ~20).  Orup's quotient (1995: N~ = N (-N^-1
mod 2^W) has low limb -1, so the digit is the accumulator's low limb itself and the products are added AFTER the shift; tools/model/orup_model.py is
the bit-level model) leaves  v_and_dpp -> DPP -> mad(q Np0)  on the chain — 3 of the 8 links — and costs one more limb per lane (its values are bounded
by N N', 29 bits more than a Montgomery result) and one more step:

    orup(L)   = max(issue(L + 1) , chain(L) x 3 / 8)   with chain(L) = step(L) measured when exposed > 0

(Every pass is priced as a one-stream pass; the two-stream pass B of a multiplication has 54 MACs per step instead of 36 — 1 / 6 of the passes —
which makes the issue-bound share slightly LARGER than stated, the possible gain smaller.)

Result (python tools/model/lone_ladder_model.py):  18 x 4 is issue-bound already (6 % exposed: nothing to gain); 9 x 8 would become issue-bound at
10 limbs per lane (159 -> 134 cycles per step, -16 %); 5 x 16 would drop from 163 to ~89 cycles (-45 %: 23.9 -> 13.1 ms per lone ladder).  On the lone
1 024-session batch: the launches of <= 4 096 items on the 5-limb layout (round 5's PDL verification 23.4 + 7.0 ms, round 4's halves 10.9 ms) would
save ~18 ms; the 1024-bit halves at 9 limbs cannot take it (36 limbs overflow); round 1's 36 ms are issue-bound on 18 limbs.  122 -> ~104 ms: still
not the 95 ms asked for, for a second Montgomery radix with per-key constants of its own, three new pass variants, a plain reduction before every
final normalisation and one more lane layout.  NOT built this round; tests/test_model_cpu.py keeps the arithmetic honest.
"""
import json

STEPS, PASSES = 71, 4900
C_MAD_1WAVE, C_OTHER, OTHERS_PER_STEP = 5.66, 4.2, 5.1
CHAIN_LINKS, CHAIN_LINKS_ORUP = 8, 3
MEASURED = {  # limbs per lane: (lanes per integer, unit ms of a lone wave, clock GHz) — profiles/r06/wave_trace_sched2.jsonl, ab_lone_1024.jsonl
    18: (4, 35.3, 2.37),
    9: (8, 23.4, 2.37),
    5: (16, 23.9, 2.37),
}


def issue_cycles(L):
    return 2 * L * C_MAD_1WAVE + OTHERS_PER_STEP * C_OTHER


def step_cycles(L):
    _, ms, ghz = MEASURED[L]
    return ms * 1e-3 * ghz * 1e9 / (PASSES * STEPS)


def orup_cycles(L):
    """the step after the rewrite: one more limb per lane to issue, 6 of the 8 chain links left"""
    step, issue = step_cycles(L), issue_cycles(L)
    chain = step if step > issue * 1.05 else 0.0            # the chain is what bounds the step only when something is exposed
    return max(issue_cycles(L + 1), chain * CHAIN_LINKS_ORUP / CHAIN_LINKS)


def table():
    out = []
    for L, (tpi, ms, ghz) in MEASURED.items():
        step, issue, orup = step_cycles(L), issue_cycles(L), orup_cycles(L)
        out.append({"limbs_per_lane": L, "lanes_per_integer": tpi, "unit_ms": ms, "cycles_per_step": round(step, 1), "issue_cycles_per_step": round(issue, 1),
                    "exposed_cycles_per_step": round(max(0.0, step - issue), 1), "exposed_share": round(max(0.0, step - issue) / step, 3),
                    "orup_cycles_per_step": round(orup, 1), "orup_unit_ms": round(ms * min(1.0, orup / step), 2),
                    "orup_gain": round(1 - min(1.0, orup / step), 3)})
    return out


def lone_batch_gain_ms(stretches_5_limb_ms=(23.4, 7.0, 4.5 + 6.4)):
    """the lone-ladder stretches of one 1 024-session batch whose launches (<= 4 096 items) could run on the 5-limb layout — round 5's PDL
    verification ladder and its short 4096-bit companion, round 4's halves (profiles/r06/final_lib_timeline_lone_1024.json,
    ab_xwide13.jsonl) — and what the 5-limb gain would take off them; the 9-limb 1024-bit halves of rounds 0 and 2 cannot take the method"""
    g = next(r for r in table() if r["limbs_per_lane"] == 5)["orup_gain"]
    return sum(stretches_5_limb_ms) * g


if __name__ == "__main__":
    print(json.dumps({"table": table(), "lone_1024_batch_gain_ms": round(lone_batch_gain_ms(), 2), "lone_1024_batch_ms": 122.1}, indent=1))
