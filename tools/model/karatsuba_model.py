"""Instruction-count model of ONE level of Karatsuba on the product halves of the pair kernel's passes (mpe_pairexp.h), priced with the
kernel's real lane layout BEFORE any kernel work (round-4 review, item 3a: "build it only if the model says >= 6 % on
pair_modexp_kernel<2048>").  Everything is counted in VALU wave-instructions per lane for one pass over a K = 72-limb operand pair
(4 lanes x 18 limbs of 29 bits, 71 outer steps, 64-bit column accumulators) — the unit the kernel is bound by (the VALU
port is 88 % busy and v_mad_u64_u32 issues at the same rate as an add).

Today (interleaved CIOS, measured on the emitted ISA by tools/isa_blocks.py: 740 VALU per 18 steps, 648 of them MACs):
    per step and lane:  18 MACs a_i * b_j  +  18 MACs m * n_i  +  5.1 others (quotient digit, its DPP broadcast, the 64-bit shift / add of
    the finished column, the DPP pull of the neighbour's limb)

Karatsuba needs the product DE-INTERLEAVED from the reduction (the three half-size products are formed first, then combined, then
reduced), which changes four things at once:
  1. MACs of the product: 3 x (36 x 36) instead of 72 x 71, spread over the 4 lanes;
  2. the operands of a half-size product live in TWO of the four lanes (lane t owns limbs [18 t, 18 t + 18)): either half of the lanes
     idle during P0 = A0 B0 and P2 = A1 B1 — or the halves are re-laid out over all four lanes (9 limbs per lane) with DPP moves;
  3. the middle product (A0 + A1)(B0 + B1) has 30-bit limbs: a column takes 36 products of 2^60 = 2^65.2 — it no longer fits the
     64-bit accumulator, so the sums must be re-normalised to 29 bits (one more limb, carries across lanes) or the middle product
     split in two accumulation rounds with a fold in between;
  4. the full 143-column product has to be HELD (36 columns of 64 bits per lane = 72 VGPRs) while the reduction runs over it — the
     interleaved CIOS holds 18 columns; the kernel sits at 255 of 256 VGPRs today.
`price()` adds these up.  Run: python tools/model/karatsuba_model.py"""

K, L, TPI, STEPS = 72, 18, 4, 71
OTHER_PER_STEP = 740 / 18 - 36            # measured: 5.1 non-MAC VALU instructions per CIOS step (ISA of the shipped kernel)


def today():
    """VALU instructions per lane for one single-stream pass (product + reduction interleaved)"""
    mac = STEPS * 2 * L
    other = STEPS * OTHER_PER_STEP
    return dict(mac=mac, other=other, total=mac + other)


def karatsuba(relayout=True):
    """one level of Karatsuba on the product half, reduction unchanged in cost (K^2 MACs + its per-step overhead)"""
    h = K // 2                                                   # 36-limb halves
    # 1. product MACs per lane.  Re-laid out over 4 lanes every half-product is 36 steps x 9 MACs; without re-layout P0 and P2 run on
    #    two lanes each (the other two idle: a wave instruction is issued for all 64 lanes anyway) = 36 steps x 18 MACs.
    per_half_product = h * (h // TPI if relayout else L)
    mac_prod = 3 * per_half_product
    # 2. re-layout: A0, A1, B0, B1 and the two sums from (2 lanes x 18) to (4 lanes x 9): one DPP move per limb that changes lane
    #    (half of them), and the three products' 72-column results back into the 4 x 36-column accumulator layout: one move per column
    #    that changes lane (again half), 64-bit columns = 2 moves
    moves = (6 * h // 2 + 3 * 2 * (2 * h) // 2) / TPI if relayout else 0
    # 3. the sums A0 + A1, B0 + B1 (36 adds each, spread over the lanes that hold them) and their re-normalisation to 29-bit limbs
    #    (mask + shift + add per limb, one cross-lane carry per lane): without it the middle product overflows the 64-bit columns
    sums = 2 * (h + 3 * h) / TPI
    # 4. combination: P1 - P0 - P2 on 72 64-bit columns (2 x 72 subtractions of 64-bit values = 2 instructions each on this ISA:
    #    v_sub_co + v_subb_co), then its addition into the middle of the 143-column product (72 64-bit adds: v_lshl_add_u64 has no
    #    carry-in, so v_add_co + v_addc_co)
    combine = (2 * 2 * (2 * h) + 2 * (2 * h)) / TPI
    # 5. the multiplier limbs are still broadcast from LDS once per product step: 3 x 36 steps instead of 71; every step keeps the
    #    loop bookkeeping the compiler emits today around the ds_read (about 1 VALU per step is address / move work)
    steps_prod = 3 * h
    step_other = steps_prod * 1.0
    # 6. reduction over the held product: 71 steps x (18 MACs m n_i + the same 5.1 others as today), plus folding each finished
    #    product column into the running window (one 64-bit add = 2 instructions per step and lane that the interleaved form gets for free)
    red = STEPS * (L + OTHER_PER_STEP + 2)
    total = mac_prod + moves + sums + combine + step_other + red
    return dict(mac=mac_prod + STEPS * L, product_mac=mac_prod, moves=moves, sums=sums, combine=combine, step_other=step_other, reduction=red, total=total,
                vgpr_columns_held=2 * (2 * K // TPI))


def price():
    t, k = today(), karatsuba(True)
    k2 = karatsuba(False)
    # a squaring modulo N^2 = pass A + one single-stream pass B; 264 instructions between the loops (ISA: tails, pre-load, LDS traffic)
    between = 264
    sq_today = 2 * t["total"] + between
    sq_kara = 2 * k["total"] + between
    return dict(today=t, karatsuba_relayout=k, karatsuba_two_lanes_idle=k2,
                pass_gain=1 - k["total"] / t["total"], pass_gain_no_relayout=1 - k2["total"] / t["total"],
                squaring_gain=1 - sq_kara / sq_today,
                verdict="build only if >= 0.06 (round-4 review): the model says %.3f on a pass, %.3f on a squaring — and the held product needs "
                        "%d more VGPRs in a kernel that uses 255 of 256" % (1 - k["total"] / t["total"], 1 - sq_kara / sq_today, k["vgpr_columns_held"] - 2 * L))


if __name__ == "__main__":
    import json
    print(json.dumps(price(), indent=1))
