"""Every recalled byte-level convention of the un-vendored crates is a run-time PROFILE (mpe_encoding / orc_encoding /
pyref.Encoding, same fields).  Here, without a GPU: under every profile of enc_profiles.PROFILES the C oracle and the
independent Python restatement produce the same proofs and the same round messages of a whole signing session, each accepts
what the other produced, and a proof made under one profile is rejected under another (so the switches really reach every
transcript).  The GPU counterpart is tests/test_encodings_gpu.py."""
import ctypes
import os
import subprocess

import pytest

import enc_cases as EC
import enc_profiles as ENCS
import fixtures as F
import gg20_fixture as G
import orc
import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_three_default_profiles_are_the_same(tmp_path):
    """pyref.Encoding(), the oracle's start-up profile and mpe_encoding_default (read through a C program that includes the
    product header — no GPU) agree field by field, and the struct layouts are identical (one ctypes struct serves both)"""
    assert orc.get_encoding() == ENCS.DEFAULT.as_dict()
    src = tmp_path / "enc.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mpecdsa_hip.h"\n'
                   '#include "mpe_oracle.h"\n'
                   '_Static_assert(sizeof(mpe_encoding) == sizeof(orc_encoding), "layout");\n'
                   '_Static_assert(offsetof(mpe_encoding, ck_salt) == offsetof(orc_encoding, ck_salt), "layout");\n'
                   '_Static_assert(offsetof(mpe_encoding, ord_heg) == offsetof(orc_encoding, ord_heg), "layout");\n'
                   '_Static_assert(offsetof(mpe_encoding, ord_cdlog) == offsetof(orc_encoding, ord_cdlog), "layout");\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mpe_encoding), offsetof(mpe_encoding, ck_salt), offsetof(mpe_encoding, ord_dlog),\n'
                   '  offsetof(mpe_encoding, ord_pedersen), offsetof(mpe_encoding, ord_ecddh), offsetof(mpe_encoding, ord_cdlog)); return 0; }\n')
    exe = tmp_path / "enc"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    E = orc.Encoding
    assert got == [ctypes.sizeof(E), E.ck_salt.offset, E.ord_dlog.offset, E.ord_pedersen.offset, E.ord_ecddh.offset, E.ord_cdlog.offset]


@pytest.mark.parametrize("name", list(ENCS.PROFILES))
def test_oracle_equals_python_under_every_profile(keys, name):
    inp = EC.inputs(keys)
    with ENCS.applied(ENCS.PROFILES[name]):
        want, got = EC.python_outputs(inp), EC.oracle_outputs(inp)
        assert got == want
        assert all(EC.oracle_verdicts(inp, want).values()) and all(EC.python_verdicts(inp, got).values())
    assert orc.get_encoding() == ENCS.DEFAULT.as_dict()                    # restored


def test_each_switch_reaches_its_transcripts(keys):
    """a proof made under the defaults is rejected under a profile that changes a convention its transcript uses — and only then"""
    inp = EC.inputs(keys)
    dflt = EC.python_outputs(inp)
    expect_rejected = {
        "compressed": {"dlog", "pedersen", "heg", "ecddh"},               # chain_point: the four curv sigma proofs
        "zero-empty": {"ck"},                                             # BigInt::from(0).to_bytes() inside zk-paillier's digest (i = 0, j = 0)
        "mask-be": {"ck"},
        "reordered": {"dlog", "pedersen", "heg", "ecddh", "cdlog"},
        "all-alt": {"dlog", "pedersen", "heg", "ecddh", "ck", "cdlog"},
    }
    for name, rejected in expect_rejected.items():
        with ENCS.applied(ENCS.PROFILES[name]):
            ov, pv = EC.oracle_verdicts(inp, dflt), EC.python_verdicts(inp, dflt)
        assert ov == pv, name
        assert {k for k, v in ov.items() if not v} == rejected, name
    # the hash commitment only depends on the zero encoding, and only for a zero blinding factor
    with ENCS.applied(ENCS.PROFILES["zero-empty"]):
        alt = EC.python_outputs(inp)["commit"]
    assert alt[0] != dflt["commit"][0] and alt[1] == dflt["commit"][1]


@pytest.mark.parametrize("name", ["all-alt", "reordered"])
def test_a_whole_signing_session_under_an_alternative_profile(keys, name):
    """every round message of every party: the per-party C oracle == the Python restatement, byte for byte, under a profile
    that changes the DLog / Pedersen / HomoELGamal transcripts of rounds 1, 2 and 5; the signature itself does not depend on
    the profile (challenges only enter proofs)"""
    lk = G.make_local_keys(keys, 1, 3, [0, 2])
    nonces = G.make_nonces(lk, 1, seed="enc-session")
    base = G.oracle_sign_ex(lk, nonces, 1)
    with ENCS.applied(ENCS.PROFILES[name]):
        want, sigs, pst = G.py_session(lk, nonces, 0)
        got = G.oracle_sign_ex(lk, nonces, 1)
    for rnd in G.ROUNDS:
        for i in range(lk["S"]):
            assert got["slabs"][rnd][i, 0].tobytes() == want[rnd][i], f"round {rnd} message of party {i}"
    assert all(st == (0, []) for st in pst) and not got["party_status"].any()
    assert (F.ints(got["r"])[0], F.ints(got["s"])[0], int(got["recid"][0])) == sigs[0]
    assert (F.ints(base["r"])[0], F.ints(base["s"])[0]) == sigs[0][:2]
    differ = [rnd for rnd in G.ROUNDS if got["slabs"][rnd].tobytes() != base["slabs"][rnd].tobytes()]
    assert differ == [1, 2, 5], differ                 # MessageB's DLog proofs, the Pedersen proof of T_i, the HomoELGamal proof of S_i


def test_a_message_made_under_another_profile_is_refused_with_the_reference_status(keys):
    """parties that disagree on the conventions cannot sign together: a party on the defaults rejects a peer's round-1
    MessageB whose DLog proofs were made under the other point form — status 201 (mta/mod.rs:170-171 -> InvalidKey), as for
    any bad proof, never an acceptance or a crash"""
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    nonces = G.make_nonces(lk, 1, seed="enc-mixed")
    parties = G.py_parties(lk, nonces, 0)
    m0 = [p.round0() for p in parties]
    m1 = [p.round1(m0) for p in parties]
    with pyref.use_encoding(ENCS.PROFILES["compressed"]):
        other = G.py_parties(lk, nonces, 0)
        o0 = [p.round0() for p in other]
        o1 = [p.round1(o0) for p in other]
    assert o0[1]["c"] == m0[1]["c"] and o1[1][0][0]["c"] == m1[1][0][0]["c"]          # same ciphertexts ...
    assert o1[1][0][0]["b_proof"] != m1[1][0][0]["b_proof"]                            # ... different Schnorr responses
    parties[0].round2([m1[0], o1[1]])
    assert parties[0].status == 201
    parties[1].round2(m1)
    assert parties[1].status == 0
