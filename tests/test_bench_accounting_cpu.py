"""bench.py's roofline arithmetic, checked without a GPU: the MAC models against their closed forms and SURVEY.md's table, and
the committed bench lines (profiles/r02/*.json: fixed windows; profiles/r03/*.json: sliding windows for the public exponent, 71
CIOS steps) against the models — `frac`, `achieved`, `executed_mac_per_launch` and the
per-launch batch sizes must be the numbers the formulas give for the workload the line names."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mpe_bench", os.path.join(ROOT, "bench.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def test_mac_models():
    assert B.mac(64) == 2 * 64 * 64 + 64 and B.mac(128) == 2 * 128 * 128 + 128
    # "Paillier-2048 modexp" = modexp(128, 2048) = 8.474e7 MACs; a (1, 3) signature = 4.17e9
    assert abs(B.modexp_macs(128, 2048) - 8.474e7) / 8.474e7 < 1e-3
    assert abs(B.sig_macs(2, 3) - 4.174e9) / 4.174e9 < 1e-3
    # SURVEY.md 8a-work exponent-bit totals for (S, n) = (2, 3) and (3, 5)
    bits = lambda S, n: (n * 6400 + 2 * (S - 1) * n * 3842 + 2 * (S - 1) * 2048 + (S - 1) * 6400 + S * (S - 1) * 3843,
                         2048 + n * 2048 + 2 * (S - 1) * (n * 2304 + 2304) + (S - 1) * 2816 + S * (S - 1) * 3074)
    assert bits(2, 3) == (60434, 35588) and bits(3, 5) == (152890, 91660)
    # the pair engine: 6-bit windows for a 2048-bit exponent: 342 windows, 2046 squarings, 64 + 342 + 2 multiplications
    assert B.window_bits(64) == 6 and B.window_bits(8) == 4 and B.window_bits(25) == 5
    assert B.pair_modexp_macs(64, 64) == (2 * 2046 + 2.5 * (64 + 342 + 2)) * B.mac(64)
    # a second (256-bit) base on the same ladder adds its 16-entry table, 64 window multiplications and one conversion
    assert B.pair_modexp_macs(64, 64, 8) - B.pair_modexp_macs(64, 64) == 2.5 * (16 + 64 + 1) * B.mac(64)
    # the pair arithmetic needs half the MACs of the textbook exponentiation on the 4096-bit integers ...
    assert 0.49 < B.pair_modexp_macs(64, 64) / B.modexp_macs(128, 2048) < 0.51
    assert B.PEAK_MAC_PER_S == 16 * 4 * 256 * 2.4e9


    # round 3: the public exponent N on sliding windows: ~2 043 squarings + 1, 31 + ~292 + 2 multiplications (expected counts)
    sq, win = B.sliding_counts(2048, 6)
    assert 2040 < sq < 2046 and 285 < win < 300
    assert B.pair_modexp_macs(64, 64, sliding=True) == (2 * (sq + 1) + 2.5 * (31 + win + 2)) * B.mac(64)
    assert 0.955 < B.pair_modexp_macs(64, 64, 8, sliding=True) / B.pair_modexp_macs(64, 64, 8) < 0.965
    assert abs(B.LIMB_INFLATION - 4 * 72 * 71 / (2 * B.mac(64))) < 1e-12 and 1.23 < B.LIMB_INFLATION < 1.24


@pytest.fixture
def sliding_priced_on(request):
    """round 4 prices the sliding schedule on the signers' own moduli N (exact window counts); earlier lines on 32 seeded exponents"""
    def set_for(rnd):
        B._SLIDING.clear()
        if rnd >= "r04":
            import sys
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import fixtures as F
            B.SLIDING_EXPONENTS = [F.load_keys()[i].N for i in (0, 1)]
        else:
            B.SLIDING_EXPONENTS = None
    yield set_for
    B._SLIDING.clear()
    B.SLIDING_EXPONENTS = None


@pytest.mark.parametrize("rnd,name", [("r02", "bench_gg20_default.json"), ("r02", "bench_gg20_driver_flags.json"),
                                      ("r03", "bench_gg20_default.json"), ("r03", "bench_gg20_driver_flags.json"),
                                      ("r04", "bench_gg20_default.json"), ("r04", "bench_gg20_driver_flags.json")])
def test_committed_bench_lines_follow_the_models(rnd, name, sliding_priced_on):
    path = os.path.join(ROOT, "profiles", rnd, name)
    if not os.path.exists(path):
        pytest.skip("no committed bench line")
    sliding_priced_on(rnd)
    b = json.loads([ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")][-1])
    sliding = rnd != "r02"
    rf, cfg = b["roofline"], b["config"]
    assert b["unit"] == "signatures/s" and b["higher_is_better"] and b["scaling"] == "weak" and b["vs_baseline"] is None
    sessions, S, n = cfg["sessions_per_gpu"], cfg["signers"], cfg["n"]
    # value = sessions * steps / time
    assert abs(b["value"] - sessions * b["n_gpus"] / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-6
    # the three launches modulo N^2 of a faithful (S, n) = (2, 3) step: AliceProof::verify for both MessageB::b calls
    # (2 (S-1) n per party, two-base), MessageB's ciphertext (2 (S-1), two-base), PDLwSlackProof::verify (S (S-1) per local party set, two-base)
    per_session = S * (2 * (S - 1) * n + 2 * (S - 1)) + S * S * (S - 1)
    model = sessions * per_session * B.pair_modexp_macs(64, 64, 8, sliding=sliding) / 3
    assert abs(rf["executed_mac_per_launch"] - model) / model < 1e-9
    assert rf["launches"] == 3 * b["steps"]
    achieved = rf["executed_mac_per_launch"] / (rf["avg_kernel_ms"] * 1e-3) / 1e12
    assert abs(achieved - rf["achieved"]) / achieved < 1e-6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] <= 1
    assert abs(rf["peak"] - B.PEAK_MAC_PER_S / 1e12) < 1e-9
    assert rf["alg_unit_frac"] > 1 > rf["frac"]                       # the SURVEY unit over-counts; the executed figure is the utilisation
    # ... and ~0.46x on the launch mix, where the short second exponent of a two-base ladder rides on the long one's squarings
    assert (0.42 if sliding else 0.44) < rf["executed_mac_per_launch"] / rf["alg_unit_mac_per_launch"] < (0.46 if sliding else 0.48)
    if sliding:
        assert rf["launches_on_sliding_windows"] == rf["launches"] and b["openssl_verified"] == sessions
        assert {x["kernel"].split("<")[0] for x in b["roofline_secondary"]} >= {"pair_modexp_kernel", "modexp_kernel", "fb_modexp_kernel"}
        assert 0.45 < b["whole_step"]["frac"] < rf["frac"]
    if "issue_ceiling" in rf:
        ic = rf["issue_ceiling"]
        infl = ic.get("limb_inflation", 1.2558)
        assert abs(ic["kernel_valu_T_lane_ops_per_s"] - rf["achieved"] * infl * 740 / 648) < 1e-6
        assert 0.9 < ic["frac_of_measured"] < 1.15
    # the dominant kernel's time is part of the step
    assert rf["avg_kernel_ms"] * 3 <= b["ms_per_step"] and 0.5 < rf["kernel_time_share_of_step"] < 0.8
    cb = b["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and b["parity_vs_oracle_on_sample"] is True
    c = b["configs"]
    if rnd >= "r04":
        # round 4: the profiler's own count of the waves that slid, the 1 024-session parity sample, the stream section
        assert rf["sliding_share_of_waves"] == 1.0 and b["parity_sample"] >= 1024
        st = c["c4_stream_1024"]
        assert st["all_sessions_signed"] and st["openssl_verified"] == st["openssl_of"] == st["batches"] * 1024 and st["parity_vs_oracle_on_sample"] is True
        assert abs(st["signatures_per_s"] - st["batches"] * 1024 / st["seconds"]) < 1e-6 * st["signatures_per_s"]
        assert rf["traffic_source"] is None or rf["traffic_source"]["measured_in_this_run"] is False
    assert c["c2_paillier_65536"]["roundtrip_ok"] and c["c2_paillier_65536"]["holder_equals_public_ciphertext"]
    assert c["c3_ec_pdl_262144"]["accept_rate"] == 1.0 and c["c3_ec_pdl_262144"]["corrupted_1pct_all_rejected"]
    assert c["c4_literal_1024"]["all_sessions_signed"] and c["c5_share_t2n5_8192"]["all_sessions_signed"]


@pytest.mark.parametrize("rnd", ["r02", "r03", "r04"])
def test_rocprof_artifacts_agree_with_the_bench_line(rnd):
    """profiles/<round>: the rocprofv3 --stats average of the dominant kernel vs the HIP-event average in the bench line (different
    boxes: within 3 %), and the PMC traffic figure the bench line quotes"""
    import csv
    d = os.path.join(ROOT, "profiles", rnd)
    if not os.path.exists(os.path.join(d, "gg20_bench_kernel_stats.csv")):
        pytest.skip("no committed profile")
    rows = list(csv.DictReader(open(os.path.join(d, "gg20_bench_kernel_stats.csv"))))
    dom = [r for r in rows if "pair_modexp_kernel<mpe::Cfg<2048, 29, 18, 4>" in r["Name"]][0]
    assert rows[0] is dom                                               # it IS the kernel that dominates the step
    b = json.loads([ln for ln in open(os.path.join(d, "bench_gg20_default.json")).read().strip().splitlines() if ln.startswith("{")][-1])
    assert abs(float(dom["AverageNs"]) / 1e6 - b["roofline"]["avg_kernel_ms"]) / b["roofline"]["avg_kernel_ms"] < 0.03
    assert 0.6 < float(dom["Percentage"]) / 100 < 0.72
    src = (b["roofline"].get("traffic_source") or {}).get("file")          # round 4: the line names the PMC pass it quotes
    pmc = json.load(open(os.path.join(ROOT, src) if src else os.path.join(d, "pmc_traffic.json")))
    k = [v for n, v in pmc["kernels"].items() if "pair_modexp_kernel" in n and "2048, 29, 18, 4" in n][0]
    assert abs(b["roofline"]["traffic"] - k["hbm_bytes_per_launch"]) / k["hbm_bytes_per_launch"] < 0.01
    assert pmc["sessions"] == b["config"]["sessions_per_gpu"]
    # HBM is not the bound: the kernel moves ~2 % of 8 TB/s
    assert k["hbm_GB_per_s"] < 0.05 * 8000


def test_self_spawn_command_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run: one rank per GPU on
    127.0.0.1, and bench.py's own flags (some are prefixes of the launcher's options: --n, --t) travel behind a `--`"""
    import subprocess
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    rc = B.respawn_under_torchrun(4, ["--gpus", "4", "--mode", "party", "--t", "2", "--n", "5"])
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    sep = cmd.index("--")
    assert cmd[sep + 1].endswith("bench.py") and cmd[sep + 2:] == ["--gpus", "4", "--mode", "party", "--t", "2", "--n", "5"]
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_the_stream_section_needs_no_child_process_and_no_queue_setting():
    """round 4 ran c4_stream_1024 in a child with GPU_MAX_HW_QUEUES=16 and a retry chain; the pipelined engine runs it in this process
    with the runtime's defaults: bench.py neither spawns for it nor touches the variable, and names the closed-loop client's bound"""
    import inspect
    src = inspect.getsource(B)
    assert not hasattr(B, "stream_child") and "--stream-child" not in src
    assert 'environ["GPU_MAX_HW_QUEUES"]' not in src and "env[\"GPU_MAX_HW_QUEUES\"]" not in src
    body = inspect.getsource(B.c4_pipeline)
    assert "Gg20Pipeline" in body and "submit_seeded" in body and "window = window or 2 * lanes * group" in body
    assert "oracle_sample_nonces" in body and "openssl_verify_all" in body        # every batch: OpenSSL on all, the oracle on a sample


def test_post_timing_watchdog_prints_the_timed_line_and_leaves_with_status_0():
    """N > 1: if a collective section after the timed region wedges, rank 0 still prints the Mode-A line (marked) and the rank exits 0
    at the deadline (bench.py PostTimingWatchdog) — run in a child process, because the watchdog ends its process."""
    import subprocess
    import sys
    code = ("import importlib.util, time, sys\n"
            f"spec = importlib.util.spec_from_file_location('mpe_bench', {os.path.join(ROOT, 'bench.py')!r})\n"
            "B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)\n"
            "w = B.PostTimingWatchdog(0, 0.3, lambda: {'metric': 'm', 'value': 1.0})\n"
            "w.at('mode_b')\n"
            "time.sleep(30)\n"
            "sys.exit(7)\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0
    line = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert line["value"] == 1.0 and line["post_timing_sections"]["completed"] is False and line["post_timing_sections"]["stuck_in"] == "mode_b"
    # a disarmed watchdog does nothing; a rank other than 0 prints nothing
    code2 = code.replace("w.at('mode_b')\n", "w.disarm()\n").replace("time.sleep(30)", "time.sleep(1)")
    p2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert p2.returncode == 7 and "{" not in p2.stdout
    code3 = code.replace("PostTimingWatchdog(0, 0.3, lambda: {'metric': 'm', 'value': 1.0})", "PostTimingWatchdog(1, 0.3, None)")
    p3 = subprocess.run([sys.executable, "-c", code3], capture_output=True, text=True, timeout=120)
    assert p3.returncode == 0 and "{" not in p3.stdout


def test_distinct_wallet_moduli_are_distinct_and_well_formed():
    """bench.py's c4_every_session_its_own_wallet: K n pairwise distinct Paillier moduli and N~ from a pool of fresh primes"""
    import bench
    import fixtures as F
    p, q, nt, h1, h2, nprimes = bench.mint_distinct_moduli(21, 4)
    P, Q_, NT, H1, H2 = F.ints(p), F.ints(q), F.ints(nt), F.ints(h1), F.ints(h2)
    assert len({a * b for a, b in zip(P, Q_)}) == 21 and len(set(NT)) == 21 and nprimes == 16
    for a, b, n, x, y in zip(P, Q_, NT, H1, H2):
        assert a.bit_length() == 1024 and b.bit_length() == 1024 and (a * b).bit_length() in (2047, 2048) and a != b
        assert n.bit_length() in (2047, 2048) and 1 < x < n and 1 < y < n
