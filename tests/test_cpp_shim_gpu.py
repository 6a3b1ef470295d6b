"""-m gpu: the reference's own unit tests of the hot path — `alice_zkp` (mta/range_proofs.rs:614-633), `bob_zkp` (:636-709), `test_mta`
(mta/test.rs:6-19: alpha + beta == a b), `test_zk_pdl_with_slack` and its `#[should_panic]` soundness twin
(zk_pdl_with_slack/test.rs:12-129) — re-stated in C++ over include/mpecdsa.hpp (the host layer a maintainer would write over the
C-ABI, with the reference's type and method names) and run on the GPU — and the reference's state-machine tests
(`simulate_signing_t1_n2_s2`, `_t1_n3_s2` for [1,2], [1,3], [2,3], `_t2_n3_s3`; gg_2020/state_machine/sign.rs:667-762): one
`OfflineStage` per party holding only its own secrets, a `Simulation` relaying their messages, `SignManual::new` / `complete` for
every party; every round message and every signature byte-identical to the oracle, every signature accepted by OpenSSL's
ECDSA_do_verify, and the constructor / message-store / pick_output errors of sign.rs:77-101,246-330.  No Python, no torch in that
process: a compiled host of the C-ABI.  Every value is also compared bit for bit with the CPU oracle inside the program."""
import os
import subprocess

import pytest

import cpp_shim

pytestmark = pytest.mark.gpu


def test_reference_unit_tests_over_the_cpp_host_layer(tmp_path, keys):
    exe = cpp_shim.build(str(tmp_path))
    fx = os.path.join(str(tmp_path), "fixture.bin")
    cpp_shim.write_fixture(fx, keys)
    p = subprocess.run([exe, fx], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    for name in ("alice_zkp", "bob_zkp", "test_mta", "test_zk_pdl_with_slack", "test_zk_pdl_with_slack_soundness", "error_mapping", "state_machine_errors", "test_two_party_sign",
                 "simulate_signing_t1_n2_s2 [1, 2]", "simulate_signing_t1_n3_s2 [1, 2]", "simulate_signing_t1_n3_s2 [1, 3]",
                 "simulate_signing_t1_n3_s2 [2, 3]", "simulate_signing_t2_n3_s3 [1, 2, 3]",
                 # party-sharded signing through mpe_comm_* / mpe_gg20_round_exchange on the real RCCL communicator (world 1)
                 "party_sharded_rccl_world1_t1_n3_s2", "party_sharded_rccl_world1_t2_n3_s3"):
        assert f"test {name} ... ok" in p.stdout, name
    assert "all tests passed" in p.stdout
    maps_check = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libmpecdsa_hip.so" in maps_check


def test_party_sharded_over_rccl_between_two_gpus_compiled_host(tmp_path, keys):
    """two processes, one per GPU, no Python in them: rank r hosts party p of session block s when (s + p) % 2 == r, every round's
    records travel through ncclAllGather behind the C-ABI, every hosted pair ends with the oracle's signature.  Needs two devices."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); the world-1 form runs in the test above")
    exe = cpp_shim.build(str(tmp_path))
    fx = os.path.join(str(tmp_path), "fixture.bin")
    cpp_shim.write_fixture(fx, keys)
    idf = os.path.join(str(tmp_path), "rccl.id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, fx, "--sharded", str(r), "2", idf, "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} of 2: party_sharded ... ok" in o, o[-2000:]
