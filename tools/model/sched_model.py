"""How long does a ladder launch take under the two ways of handing out its units — and why the headline moved after four flat rounds.

Inputs, all from the per-wave trace (profiles/r06/wave_trace.jsonl, wave_trace_sched2.jsonl; units of pair_modexp_kernel<Cfg<2048,29,18,4>>:
16 two-base ladders modulo N^2 per wave):

    LONE  = 35.3 ms   a unit on a SIMD that holds no other ladder wave
    OLDER = 38.4 ms   a unit of the OLDER of two resident waves (the arbiter favours it: 0.92 of its lone speed)
    the YOUNGER wave progresses in what is left: it finishes its unit 64 ms after a common start when the older wave leaves after one unit and
    88 ms after it when the older wave runs two; the mean of the two implied rates is used (pair / lone = 1.20 units per time), and the
    lone rate once the older wave has left

One SIMD is simulated (all 1 024 behave alike): two waves start together at t = 0.
  * STATIC (rounds 1-5): each wave owns half of the SIMD's units.  The older wave finishes its share at 0.6 of the launch and exits; the younger
    one runs the rest alone.
  * QUEUE (round 6, mpe_sched.h SCHED_ALL): whoever finishes a unit pulls the next one until the SIMD's share of the queue is dry.
The model reproduces the measured launches within 5 % and their gain within two points (tests/test_model_cpu.py):

    units per SIMD      static model / measured      queue model / measured      gain model / measured
    3  (1.5 passes)         90 /  88 - 96                 90 /  86                    0 / 2 - 10 %  (static tails depended on who owned them)
    4  (2 passes)          126 / 128                     118 / 112                  6.2 / 12.3 %
    40 (20 passes)       1 257 / 1 234                 1 179 / 1 144                6.2 /  7.3 %    <- the dominant kernel of the 65 536-session step

Run: python tools/model/sched_model.py
"""
import json

LONE, OLDER, YOUNGER_FIRST, THREE_UNITS = 35.3, 38.4, 64.0, 88.0
# progress rate (units per ms) of the younger wave while the older one is running: two measurements — it finishes its unit 64 ms after a
# common start when the older wave leaves after ONE unit, and 88 ms after it when the older wave runs TWO — give 0.0072 and 0.0089; the mean
R_Y = 0.5 * ((1.0 - (YOUNGER_FIRST - OLDER) / LONE) / OLDER + (1.0 - (THREE_UNITS - 2 * OLDER) / LONE) / (2 * OLDER))
R_O = 1.0 / OLDER
R_L = 1.0 / LONE


def simulate(units, queue):
    """time until the SIMD's `units` units are done: two waves from t = 0; returns (ms, units run by the older wave)"""
    if units <= 0:
        return 0.0, 0
    if units == 1:
        return LONE, 1
    t = 0.0
    if queue:
        left = units - 2                    # both waves hold one unit
        own = [1, 1]
    else:
        own = [(units + 1) // 2, units // 2]
        left = 0
    done_o = 0
    prog_o, prog_y = 0.0, 0.0               # progress inside the current unit
    alive_o, alive_y = True, True
    have_o, have_y = own[0], own[1]         # units still to finish, the current one included
    while alive_o or alive_y:
        ro = (R_O if alive_y else R_L) if alive_o else 0.0
        ry = (R_Y if alive_o else R_L) if alive_y else 0.0
        dt_o = (1.0 - prog_o) / ro if alive_o else float("inf")
        dt_y = (1.0 - prog_y) / ry if alive_y else float("inf")
        dt = min(dt_o, dt_y)
        t += dt
        prog_o += ro * dt
        prog_y += ry * dt
        if alive_o and prog_o >= 1.0 - 1e-12:
            prog_o = 0.0
            done_o += 1
            have_o -= 1
            if have_o == 0:
                if queue and left > 0:
                    left -= 1
                    have_o = 1
                else:
                    alive_o = False
        if alive_y and prog_y >= 1.0 - 1e-12:
            prog_y = 0.0
            have_y -= 1
            if have_y == 0:
                if queue and left > 0:
                    left -= 1
                    have_y = 1
                else:
                    alive_y = False
    return t, done_o


def table():
    out = []
    for units, static_meas, queue_meas in ((3, (88.0, 96.0), 86.0), (4, (128.0, 128.0), 112.3), (40, (1234.0, 1234.0), 1144.0)):
        ts, _ = simulate(units, False)
        tq, older = simulate(units, True)
        out.append({"units_per_simd": units, "static_ms": round(ts, 1), "static_measured_ms": static_meas, "queue_ms": round(tq, 1),
                    "queue_measured_ms": queue_meas, "units_run_by_the_older_wave": older, "gain": round(1 - tq / ts, 3)})
    return out


if __name__ == "__main__":
    print(json.dumps({"rates_units_per_ms": {"lone": R_L, "older": R_O, "younger_beside_older": R_Y, "pair": R_O + R_Y,
                                              "pair_over_lone": (R_O + R_Y) / R_L}, "table": table()}, indent=1))
