"""The code paths LARGE batches take (one stream, 18 limbs per lane, one-item-per-lane EC round kernels, two-base verifier
ladders), exercised on small parity cases through a context pinned to them (conftest.gpu_ctx_serial): the ordinary GPU tests
run tiny batches and therefore the small-batch variants (forks, 5- and 9-limb lanes, lane-group kernels, split inversions)."""
import numpy as np
import pytest

import fixtures as F
import gg20_fixture as G
import pyref
from test_gg20_gpu import TAMPERS, _run, tamper_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("t,n,signers,B,kw", [(1, 3, [0, 2], 5, {}), (2, 5, [0, 2, 3, 4], 2, {}), (1, 3, [1, 2], 3, {"dedup_verify": True})])
def test_sign_matches_oracle_on_the_large_batch_paths(gpu_ctx_serial, keys, t, n, signers, B, kw):
    lk, nonces, (r, s, recid, status, R) = _run(gpu_ctx_serial, keys, t, n, signers, B, f"serial-{t}-{n}-{signers}", **kw)
    wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
    assert list(status) == [0] * B == list(wstatus)
    assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws) and list(recid) == list(wrecid)
    assert np.array_equal(R.view(np.uint32), wR)
    for b in range(B):
        assert pyref.ecdsa_verify(lk["y"], F.ints(nonces["msg"][b:b + 1])[0], F.ints(wr[b:b + 1])[0], F.ints(ws[b:b + 1])[0])


def test_tamper_matrix_on_the_large_batch_paths(gpu_ctx_serial, keys):
    """the full matrix of tests/test_gg20_gpu.py (one session per case in one batch) on the code paths large batches take"""
    tamper_cases(gpu_ctx_serial, keys, TAMPERS)


@pytest.mark.parametrize("xdiv", ["0", "1000000"])
def test_sign_matches_oracle_on_each_small_batch_lane_layout(keys, xdiv):
    """Tiny batches take the 5-limbs-per-lane layout of the pair engine (16 / 8 lanes per integer); option xwide_div = 0
    (mpe_ctx_set_option) sends them to the 9-limb layout instead, a huge value sends every launch below the
    2x threshold to the 5-limb one: the three layouts share the per-modulus constants and must produce the oracle's bytes."""
    from multi_party_ecdsa_amd import engine as E
    gpu_ctx = E.Context(0, options={"xwide_div": xdiv})
    if True:
        for t, n, signers, B in [(1, 3, [0, 1], 4), (2, 5, [0, 2, 4], 2)]:
            lk, nonces, (r, s, recid, status, R) = _run(gpu_ctx, keys, t, n, signers, B, f"lanes-{xdiv}-{t}-{n}")
            wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
            assert list(status) == [0] * B == list(wstatus)
            assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws) and list(recid) == list(wrecid)
            assert np.array_equal(R.view(np.uint32), wR)


def test_round1_inversion_started_in_round0_gives_the_same_bytes(keys):
    """Lock-step signing of a small batch (mpe_gg20_sign) starts the inversion of the ciphertexts that round 1's verifiers need in round 0,
    behind the encryption, and sends MessageB's DLog proofs in front of the N~ side (mpe_gg20.h round0 / round1).  That happens
    when round 1 merges its ladder launches — above 512 sessions with the shipped thresholds (tests/test_fullsize_gpu.py's config 4, the
    bench), at ANY size with merge_r1_quarters = 0: the small parity cases here, against the oracle and against a context with the
    schedule switched off."""
    from multi_party_ecdsa_amd import engine as E
    ahead = E.Context(0, options={"merge_r1_quarters": 0})
    plain = E.Context(0, options={"merge_r1_quarters": 0, "no_r1_inversion_ahead": 1})
    behind = E.Context(0, options={"merge_r1_quarters": 0, "no_r1_dlog_first": 1})
    # ... and the rest of round 6's small-batch schedule off: no wave priorities, the PDL proofs' beta^N computed in round 4 instead of
    # two rounds ahead, the provers' r^e mod N on the 2048-bit ladder instead of through p | q
    round5 = E.Context(0, options={"merge_r1_quarters": 0, "no_r1_inversion_ahead": 1, "no_prio": 1, "no_pdl_ahead": 1, "no_crt_n": 1})
    assert ahead.get_option("no_r1_inversion_ahead") == 0 and plain.get_option("no_r1_inversion_ahead") == 1
    for t, n, signers, B, kw in [(2, 5, [0, 2, 4], 2, {}), (1, 3, [0, 1], 5, {"chunk": 2})]:     # (a dedup_verify case ran once: same bytes)
        seed = f"ahead-{t}-{n}-{signers}-{sorted(kw)}"
        lk, nonces, (r, s, recid, status, R) = _run(ahead, keys, t, n, signers, B, seed, **kw)
        wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, nonces, B)
        assert list(status) == [0] * B == list(wstatus)
        assert np.array_equal(r.view(np.uint32), wr) and np.array_equal(s.view(np.uint32), ws) and list(recid) == list(wrecid)
        assert np.array_equal(R.view(np.uint32), wR)
        for other in ((plain, behind, round5) if t == 1 else (plain, round5)):
            _, _, (r2, s2, recid2, status2, R2) = _run(other, keys, t, n, signers, B, seed, **kw)
            assert np.array_equal(r, r2) and np.array_equal(s, s2) and np.array_equal(R, R2) and list(status2) == [0] * B
