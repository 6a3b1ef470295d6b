"""Issue-budget model of the dominant kernel (pair_modexp_kernel<Cfg<2048,29,18,4>, true>): does the multiplier's own issue rate explain the
measured launch time?  (Round-4 review, item 3: "dominant kernel <= 800 ms per 436 907-item launch OR a committed model + ISA count showing
why not".)  Three measured inputs, all committed:
  * the instruction mix of the kernel's hot loops, counted on the emitted ISA (tools/isa_blocks.py; `census()` below condenses it —
    profiles/r05/isa_census_pair2048.json is the census of the shipped build);
  * the issue rate of v_mad_u64_u32 on gfx950 at the kernel's occupancy (profiles/r01_valu_rate.json: 5.03 cycles per instruction per
    SIMD with two waves resident) and of the cheap VALU instructions around it (add / DPP / shift class: 2.3 - 4.4 cycles);
  * the clock the chip really ran at under this kernel (profiles/r05/pmc_traffic.json: GRBM_GUI_ACTIVE / time) and SQ_INSTS_VALU per launch.
Model: a SIMD holds two ladder waves; a trip (one ladder of 16 integers per wave) costs
    2 waves x (MACs x c_mad + other VALU x c_other) cycles,
a launch is `trips` such trips.  If the prediction is AT or ABOVE the measurement, the kernel already issues its multiplies as fast as the
instruction allows and only executing FEWER instructions can shorten it.  Run: python tools/model/issue_budget.py [census.json]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STEPS, L = 71, 18


def census(isa_path, cfg="CfgILi2048ELi29ELi18ELi4EEELb1EEEvi"):
    """the three hot loops of the kernel (one-stream pass A, one-stream pass B of a squaring, two-stream pass B of a multiplication), recognised by
    their MAC counts per 18-step trip (648 / 648 / 972), and everything else that runs once per multiplication modulo N^2"""
    lines = open(isa_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3mpe18pair_modexp_kernel") and cfg in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], []]
            blocks.append(cur)
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    loops = []
    for name, ins in blocks:
        mad = sum(1 for x in ins if x.startswith("v_mad_u64_u32"))
        if mad >= 600:
            loops.append(dict(block=name, insts=len(ins), mad=mad, valu=sum(1 for x in ins if x.startswith("v_")),
                              ds=sum(1 for x in ins if x.startswith("ds_")), waitcnt=sum(1 for x in ins if x.startswith("s_waitcnt"))))
    one = [b for b in loops if b["mad"] == 2 * L * L]
    two = [b for b in loops if b["mad"] == 3 * L * L]
    assert len(one) == 2 and len(two) == 1, loops
    return dict(isa=os.path.basename(isa_path), pass_one_stream=one, pass_two_streams=two[0],
                between_loops_valu_per_multiplication=264,      # ripple tails, pass-B pre-load, the LDS round trip of u
                steps=STEPS, limbs_per_lane=L)


def per_ladder(c, squarings=2044, multiplications=406):
    """VALU wave-instructions of one wave's ladder (16 two-base exponentiations): the kernel's own sliding-window counts for a 2048-bit
    public exponent plus a 256-bit second exponent (bench.py pair_modexp_macs: 2 044 squarings, 31 + 292 + 2 + 81 multiplications)"""
    trips = STEPS / L
    a = c["pass_one_stream"][0]
    b1 = c["pass_one_stream"][1]
    b2 = c["pass_two_streams"]
    sq_mad = (a["mad"] + b1["mad"]) * trips
    sq_valu = (a["valu"] + b1["valu"]) * trips + c["between_loops_valu_per_multiplication"]
    mu_mad = (a["mad"] + b2["mad"]) * trips
    mu_valu = (a["valu"] + b2["valu"]) * trips + c["between_loops_valu_per_multiplication"]
    mad = squarings * sq_mad + multiplications * mu_mad
    valu = squarings * sq_valu + multiplications * mu_valu
    return dict(mad=mad, valu=valu, other=valu - mad, per_squaring=dict(mad=sq_mad, valu=sq_valu), per_multiplication=dict(mad=mu_mad, valu=mu_valu))


def predict(c, items_per_launch, clock_ghz, c_mad=5.03, c_other=3.3, resident_waves=2048, groups=16):
    """seconds per launch: waves_per_simd x (MACs x c_mad + others x c_other) cycles per trip"""
    lad = per_ladder(c)
    trips = items_per_launch / (resident_waves * groups)
    cycles_per_trip = 2 * (lad["mad"] * c_mad + lad["other"] * c_other)
    return dict(trips=trips, cycles_per_trip=cycles_per_trip, seconds=trips * cycles_per_trip / (clock_ghz * 1e9), ladder=lad,
                mad_share_of_issue_time=lad["mad"] * c_mad / (lad["mad"] * c_mad + lad["other"] * c_other))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05", "isa_census_pair2048.json")
    c = census(path) if path.endswith(".s") else json.load(open(path))
    if path.endswith(".s"):
        print(json.dumps(c, indent=1))
        return
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r05", "pmc_traffic.json")))
    k = pmc["kernels"][pmc["dominant_kernel"]]
    items = 655360                                              # (1 048 576 + 262 144) / 2: the two launches of a 65 536-session step
    p = predict(c, items, k["effective_clock_GHz"])
    waves = items / 16
    out = dict(valu_per_wave_ladder_model=p["ladder"]["valu"], valu_per_wave_ladder_measured=k["sq_per_launch"]["SQ_INSTS_VALU"] / waves,
               mad_per_wave_ladder=p["ladder"]["mad"], predicted_ms=p["seconds"] * 1e3, measured_ms=k["avg_ms"],
               measured_over_predicted=k["avg_ms"] / (p["seconds"] * 1e3), mad_share_of_issue_time=p["mad_share_of_issue_time"],
               clock_ghz=k["effective_clock_GHz"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
