"""What a LONE ladder wave is bound by, and what a delayed quotient (Orup) could buy — priced BEFORE building anything
(round-5 review, item 1b: "bit-level model + CPU test before the kernel").

Measured inputs (profiles/r06/wave_trace*.jsonl: s_memrealtime around every (wave, unit) of pair_modexp_kernel, waves alone on their SIMD;
profiles/r01_valu_rate.json: issue cycles of a single resident wave):

  layout (limbs per lane x lanes per integer)   18 x 4      9 x 8       5 x 16
  unit = one 2048-bit two-base ladder mod N^2   35.3 ms     23.4 ms     43.8 ms   (2 450 multiplications mod N^2 = 4 900 passes of 71 steps)
  clock under lone waves                        2.37 GHz    2.37 GHz    2.37 GHz

Per CIOS step a lane issues  2 L  v_mad_u64_u32 (L products a_i b_j, L products m n_i; one-stream pass) and ~5.1 other VALU instructions
(the quotient digit's v_mul_lo_u32, its DPP broadcast, the 64-bit shift + add of the finished column, the DPP pull of the neighbour's limb).
A single resident wave issues a v_mad_u64_u32 every 5.66 cycles and the cheap class every ~4.2.  So

    issue(L)  = 2 L x 5.66 + 5.1 x 4.2              cycles per step if nothing ever waits
    step(L)   = unit x clock / (4 900 x 71)          cycles per step measured
    exposed   = step - issue                         cycles per step in which the wave waits for its OWN previous result:
                                                     the quotient-digit chain  mad(a0 b_j) -> mul_lo -> DPP -> DPP -> mad(m n) -> shift -> add -> mad

The delayed quotient (Orup 1995: N' = N (-N^-1 mod 2^W) has low limb -1, so the digit is the accumulator's low limb itself, and one more limb of
R lets digit j be taken BEFORE a_0 b_j is added) removes the first mad and the v_mul_lo_u32 from that chain — 2 of its ~8 links — and costs one
more limb of N' and one more step: K + 1 limbs do not divide over the lanes (72 = 8 x 9; 73 -> 8 x 10 padded), i.e. +1 limb per lane.

    orup(L)   = max(issue(L + 1) , chain(L) x 6 / 8)   with chain(L) = step(L) measured when exposed > 0

(Every pass is priced as a one-stream pass; the two-stream pass B of a multiplication has 54 MACs per step instead of 36 — 1 / 6 of the passes —
which makes the issue-bound share slightly LARGER than stated, the possible gain smaller.)

Result (python tools/model/lone_ladder_model.py):  18 x 4 is issue-bound already (6 % exposed: nothing to gain); 9 x 8 could go from 159 to
~135 cycles per step (-15 %: 23.4 -> 19.8 ms per lone ladder = ~8 ms of a 124 ms lone 1 024-session batch over its lone-ladder stretches —
95 ms is out of reach this way); 5 x 16 is so chain-bound (298 cycles per step for 15 instructions: the broadcast crosses four DPP hops) that even
6 / 8 of its chain loses to 9 x 8 as it is.  A rewrite of every pass of the pair engine (pre-loaded columns of pass B, the per-key constants K_c,
R, R^2, 2^BITS R as pairs — all derived from R — and a tenth limb per lane in the layout that is register-bound today) for <= 6.6 % of the lone
batch and nothing anywhere else: NOT built this round.  tests/test_model_cpu.py keeps the arithmetic honest.
"""
import json

STEPS, PASSES = 71, 4900
C_MAD_1WAVE, C_OTHER, OTHERS_PER_STEP = 5.66, 4.2, 5.1
CHAIN_LINKS, CHAIN_LINKS_ORUP = 8, 6
MEASURED = {  # limbs per lane: (lanes per integer, unit ms of a lone wave, clock GHz) — profiles/r06/wave_trace_sched2.jsonl, ab_lone_1024.jsonl
    18: (4, 35.3, 2.37),
    9: (8, 23.4, 2.37),
    5: (16, 43.8, 2.37),
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


def lone_batch_gain_ms(stretches_ms=(3.8 + 6.8, 6.3, 4.7 + 6.4, 24.7)):
    """the lone-ladder stretches of one 1 024-session batch that run on the 9-limb layout (rounds 0, 2, 4: the 1024-bit halves; round 5: the PDL
    verification's ladder) — profiles/r06/timeline_*.json — and what the 9-limb gain would take off them"""
    g = next(r for r in table() if r["limbs_per_lane"] == 9)["orup_gain"]
    return sum(stretches_ms) * g


if __name__ == "__main__":
    print(json.dumps({"table": table(), "lone_1024_batch_gain_ms": round(lone_batch_gain_ms(), 2), "lone_1024_batch_ms": 124.2}, indent=1))
