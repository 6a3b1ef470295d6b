"""The device-side sampler (mpe_sample.h) against its CPU restatement (oracle/sampler_oracle.c, itself checked against an independent
Python restatement in test_sampler_cpu.py): the same seed expands to the same arrays, bit for bit — rejected draws, the gcd loop of
`from_modulo` (src/utilities/mta/range_proofs.rs:538-557) and all — and sessions signed from device-sampled values equal the oracle's."""
import hashlib
import math

import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
import orc
from multi_party_ecdsa_amd import engine as E

pytestmark = pytest.mark.gpu
SEED = bytes(range(32))


def dv(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(ctx.device)


def hv(t):
    return t.cpu().numpy().view(np.uint32)


def test_sample_bits_and_scalars(gpu_ctx):
    ctx = gpu_ctx
    for bits, words in ((256, 8), (1, 1), (7, 2), (33, 2), (2047, 64), (2816, 88)):
        got = hv(E.sample_bits(ctx, 130, SEED, 0x55, bits, words))
        assert np.array_equal(got, orc.sample_bits(130, SEED, 0x55, bits, words)), bits
    got, fail = E.sample_scalar(ctx, 1000, SEED, 0x56)
    want, _ = orc.sample_scalar(1000, SEED, 0x56)
    assert np.array_equal(hv(got), want) and int(fail.item()) == 0


def test_sample_below_matches_the_oracle_rejected_draws_included(gpu_ctx):
    ctx = gpu_ctx
    keys = F.load_keys()
    Q = F.Q
    cases = [(keys[0].N, 64), ((1 << 2047) + 1, 64), ((1 << 2046) + 12345, 64), ((1 << 300) - 1, 10), (Q ** 3, 24), (Q * keys[1].Nt, 72),
             (Q ** 3 * keys[2].Nt, 88), (3, 1), (2, 2), (1, 1), ((1 << 64) + 1, 3)]
    for sid, (u, words) in enumerate(cases):
        bw = (u.bit_length() + 31) // 32
        got, fail = E.sample_below(ctx, 200, SEED, 100 + sid, dv(ctx, F.words([u], bw)), words)
        want, wf = orc.sample_below(200, SEED, 100 + sid, F.words([u], bw), words)
        assert wf == 0 and int(fail.item()) == 0 and np.array_equal(hv(got), want), u.bit_length()
    # per-item bounds through an index, ragged batch
    tab = F.words([k.N for k in keys[:5]], 64)
    idx = np.array([(7 * i) % 5 for i in range(77)], dtype=np.int32)
    got, _ = E.sample_below(ctx, 77, SEED, 200, dv(ctx, tab), 64, d_bound_idx=torch.from_numpy(idx).to(ctx.device))
    want, _ = orc.sample_below(77, SEED, 200, tab, 64, bound_idx=idx)
    assert np.array_equal(hv(got), want)


def test_flags_on_the_device(gpu_ctx):
    ctx = gpu_ctx
    N = F.load_keys()[0].N
    got, _ = E.sample_below(ctx, 64, SEED, 300, dv(ctx, F.words([3], 1)), 1, flags=E.SAMPLE_NONZERO)
    assert np.array_equal(hv(got), orc.sample_below(64, SEED, 300, F.words([3], 1), 1, flags=orc.SAMPLE_NONZERO)[0])
    got, _ = E.sample_below(ctx, 64, SEED, 301, dv(ctx, F.words([N - 2], 64)), 64, flags=E.SAMPLE_PLUS_ONE)
    assert np.array_equal(hv(got), orc.sample_below(64, SEED, 301, F.words([N - 2], 64), 64, flags=orc.SAMPLE_PLUS_ONE)[0])
    # from_modulo where the gcd really refuses candidates: a modulus made of small primes, and an honest key (never refuses)
    smooth = 1
    for p in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47):
        smooth *= p
    smooth = smooth ** 32
    for sid, u in ((302, smooth), (303, N), (304, 3 * 5 * 7 * 11)):
        bw = (u.bit_length() + 31) // 32
        got, fail = E.sample_below(ctx, 150, SEED, sid, dv(ctx, F.words([u], bw)), 64, flags=E.SAMPLE_COPRIME)
        want, wf = orc.sample_below(150, SEED, sid, F.words([u], bw), 64, flags=orc.SAMPLE_COPRIME)
        assert wf == 0 and int(fail.item()) == 0 and np.array_equal(hv(got), want)
        assert all(math.gcd(v, u) == 1 for v in F.ints(hv(got)))
    # an even bound is a failure for every item, as in the oracle
    got, fail = E.sample_below(ctx, 5, SEED, 305, dv(ctx, F.words([1 << 100], 4)), 4, flags=E.SAMPLE_COPRIME)
    assert int(fail.item()) == 5 and not hv(got).any()


@pytest.mark.parametrize("shape", [(1, 3, [0, 1], None, 37), (1, 3, [0, 2], None, 64), (2, 5, [0, 2, 4], [1], 20), (2, 4, [1, 2, 3], [0, 2], 9)])
def test_gg20_nonces_equal_the_oracles(gpu_ctx, shape):
    ctx = gpu_ctx
    t, n, signers, local, B = shape
    keys = F.load_keys()
    lk = G.make_local_keys(keys, t, n, signers)
    own = None if local is None else sorted(signers[i] for i in local)
    gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"], own=own)
    got, fail = E.gg20_sample_nonces(ctx, gk, B, SEED, 1234567, local=local)
    want, wf = G.oracle_sample_nonces(lk, B, SEED, 1234567, local=local)
    assert wf == 0 and int(fail.item()) == 0
    for f in G.NONCE_FIELDS[:-1]:
        assert np.array_equal(hv(got[f]), want[f]), f
    gk.close()


def test_gg20_nonces_with_key_sets(gpu_ctx):
    """sessions of different wallets draw below THEIR moduli (d_keyset)"""
    ctx = gpu_ctx
    keys = F.load_keys()
    t, n, signers, K, B = 1, 3, [0, 1], 3, 10
    lks = [G.make_local_keys(keys[3 * kk:3 * kk + 3], t, n, signers, seed="w%d" % kk) for kk in range(K)]
    arrays = {f: np.concatenate([lk["arrays"][f] for lk in lks]) for f in ("x", "p", "q", "Nt", "h1", "h2", "y", "X")}
    arrays["signers"] = lks[0]["arrays"]["signers"]
    lk = dict(t=t, n=n, S=2, arrays=arrays, nkeysets=K)
    keyset = np.array([b % K for b in range(B)], dtype=np.int32)
    gk = E.Gg20Keys(ctx, t, n, signers, arrays, nkeysets=K)
    got, fail = E.gg20_sample_nonces(ctx, gk, B, SEED, 9, keyset=torch.from_numpy(keyset).to(ctx.device))
    want, wf = G.oracle_sample_nonces(lk, B, SEED, 9, keyset=keyset)
    assert wf == 0 and int(fail.item()) == 0
    for f in G.NONCE_FIELDS[:-1]:
        assert np.array_equal(hv(got[f]), want[f]), f
    gk.close()


def test_sessions_signed_from_device_sampled_nonces(gpu_ctx):
    """seed -> nonces -> signatures entirely on the device == the oracle expanding the same seed and signing on the CPU"""
    import ossl
    ctx = gpu_ctx
    keys = F.load_keys()
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    gk = E.Gg20Keys(ctx, 1, 3, [0, 1], lk["arrays"])
    B = 24
    msg = F.words([int.from_bytes(hashlib.sha256(b"sampled %d" % b).digest(), "big") for b in range(B)], 8)
    nonces, fail = E.gg20_sample_nonces(ctx, gk, B, SEED, 42, msg=dv(ctx, msg))
    r, s, recid, status = E.gg20_sign(ctx, gk, nonces, B)
    ctx.sync()
    z, _ = G.oracle_sample_nonces(lk, B, SEED, 42, msg=msg)
    wr, ws, wrecid, _, wstatus = G.oracle_sign(lk, z, B)
    assert int(fail.item()) == 0 and not status.cpu().numpy().any() and not wstatus.any()
    assert np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws) and np.array_equal(recid.cpu().numpy(), wrecid)
    assert ossl.ecdsa_verify(lk["arrays"]["y"][0], msg, wr, ws).all()
    # the next batch counter is a different batch: no value repeats
    n2, _ = E.gg20_sample_nonces(ctx, gk, B, SEED, 43, msg=dv(ctx, msg))
    for f in ("k", "gamma", "r_a", "mb_r"):
        assert not np.intersect1d(hv(nonces[f])[:, :2].copy().view(np.uint64), hv(n2[f])[:, :2].copy().view(np.uint64)).size
    gk.close()


def test_a_rejection_loop_that_gives_up_is_the_sessions_status(gpu_ctx):
    """curv's sample_below / from_modulo / Scalar::random loop until a candidate fits (range_proofs.rs:538-557); a GPU lane stops after
    sampler_max_attempts (default 128: < 2^-128 per draw) — a deliberate divergence that must never let a session sign on the zeros a
    given-up draw leaves.  With THREE attempts per draw some draws below N give up: the device and the oracle agree on every array, the
    party's k_i is the invalid scalar 2^256 - 1, mpe_gg20_sign answers MPE_GG20_STATUS_BAD_NONCE (91) for exactly those sessions —
    no signature — and every other session of the batch signs and equals the oracle's."""
    keys = F.load_keys()
    lk = G.make_local_keys(keys, 1, 3, [0, 1])
    B = 24
    msg = F.words([int.from_bytes(hashlib.sha256(b"gives up %d" % b).digest(), "big") for b in range(B)], 8)
    ctx = E.Context(0, options={"sampler_max_attempts": 3})
    assert ctx.get_option("sampler_max_attempts") == 3
    gk = E.Gg20Keys(ctx, 1, 3, [0, 1], lk["arrays"])
    # the primitive: a bound that rejects EVERY candidate (sample_below(1) with "non-zero") fails for every item, zeros out
    got, fail = E.sample_below(ctx, 9, SEED, 700, dv(ctx, F.words([1], 1)), 1, flags=E.SAMPLE_NONZERO)
    assert int(fail.item()) == 9 and not hv(got).any()
    try:
        orc.lib.orc_sampler_set_max_attempts(3)
        z, wf = G.oracle_sample_nonces(lk, B, SEED, 4242, msg=msg)
    finally:
        orc.lib.orc_sampler_set_max_attempts(128)
    nonces, fail = E.gg20_sample_nonces(ctx, gk, B, SEED, 4242, msg=dv(ctx, msg))
    assert wf > 0 and int(fail.item()) >= wf            # (the device counts a from_modulo item in both of its passes)
    for f in G.NONCE_FIELDS[:-1]:
        assert np.array_equal(hv(nonces[f]), z[f]), f
    k = F.ints(z["k"])
    bad = sorted({pi // 2 for pi in range(2 * B) if k[pi] == (1 << 256) - 1})
    assert bad and len(bad) < B
    r, s, recid, status = E.gg20_sign(ctx, gk, nonces, B)
    ctx.sync()
    wr, ws, wrecid, _, wstatus = G.oracle_sign(lk, z, B)
    st = status.cpu().numpy()
    assert np.array_equal(st, wstatus) and [b for b in range(B) if st[b]] == bad and all(st[b] == E.N_.GG20_STATUS_BAD_NONCE for b in bad)
    assert np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws) and not hv(r)[bad].any() and not hv(s)[bad].any()
    # a caller's own out-of-range k_i is refused the same way (Scalar::random never returns it)
    host = G.make_nonces(lk, 4, seed="bad k")
    host["gamma"][3] = 0                                                # session 1, party 1: gamma_i = 0
    host["k"][4] = np.array([0xFFFFFFFF] * 8, dtype=np.uint32)          # session 2, party 0: k_i >= q
    r, s, recid, status = E.gg20_sign(ctx, gk, {f: dv(ctx, v) for f, v in host.items()}, 4)
    ctx.sync()
    assert status.cpu().numpy().tolist() == [0, 91, 91, 0] and np.array_equal(status.cpu().numpy(), G.oracle_sign(lk, host, 4)[4])
    gk.close()


def test_options_are_the_only_switch(gpu_ctx):
    """the library reads no environment variable: an unknown key or a value out of range is refused, every key round-trips"""
    ctx = E.Context(0)
    with pytest.raises(E.N_.MpeError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(E.N_.MpeError):
        ctx.set_option("wide_div", 0)
    with pytest.raises(E.N_.MpeError):
        ctx.set_option("grid", "diagonal")
    n = E.N_.lib.mpe_ctx_option_count()
    names = [E.N_.lib.mpe_ctx_option_name(i).decode() for i in range(n)]
    assert {"no_par", "wide_div", "fb_window_bits", "sampler_max_attempts", "grid_mode"} <= set(names)
    for key, val in (("no_par", 1), ("wide_div", 4), ("xwide_div", 0), ("fb_window_bits", 10), ("waves_per_cu", 4), ("no_sliding", 1)):
        ctx.set_option(key, val)
        assert ctx.get_option(key) == val
    ctx.set_option("grid", "full")
    assert ctx.get_option("grid_mode") == 1
