"""The pipelined engine (mpe_gg20_pipeline_*, mpe_pipeline.h): a stream of small batches coalesced into passes that run on a few
concurrent lanes gives, batch by batch, exactly what mpe_gg20_sign and the oracle give — whatever the grouping, partly filled groups
and the order of waiting; the seeded form signs from values sampled on the device (mpe_sample.h) and equals the oracle expanding the
same seed.  Reference shape: many OfflineStage instances side by side, state_machine/sign.rs:667-691."""
import hashlib

import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
from multi_party_ecdsa_amd import engine as E

pytestmark = pytest.mark.gpu
SEED = hashlib.sha256(b"pipeline").digest()


def dv(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(ctx.device)


def hv(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.fixture(scope="module")
def wallet(gpu_ctx):
    lk = G.make_local_keys(F.load_keys(), 1, 3, [0, 1])
    gk = E.Gg20Keys(gpu_ctx, 1, 3, [0, 1], lk["arrays"])
    yield lk, gk
    gk.close()


@pytest.mark.parametrize("group,lanes,nb", [(3, 2, 5), (1, 1, 2)])     # (a third shape, (4, 3, 5), ran until round 5: 10 s, nothing the first does not cover)
def test_stream_of_batches_equals_batch_by_batch_signing(gpu_ctx, wallet, group, lanes, nb):
    ctx = gpu_ctx
    lk, gk = wallet
    B = 6
    host = [G.make_nonces(lk, B, seed="pipe-%d" % b) for b in range(nb)]
    dev = [{f: dv(ctx, v) for f, v in h.items()} for h in host]
    pipe = E.Gg20Pipeline(ctx, gk, B, group=group, lanes=lanes)
    tickets = [pipe.submit(d, want_R=True) for d in dev]
    assert tickets == list(range(1, nb + 1))
    # waiting in reverse order: the last ticket sits in a partly filled group that wait() must flush
    got = {}
    for t in reversed(tickets):
        got[t] = pipe.wait(t, want_R=True)
        assert pipe.done(t)
    assert all(pipe.latency_ms(t) > 0 for t in tickets)
    for b, t in enumerate(tickets):
        r, s, recid, status, R = got[t]
        wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, host[b], B)
        assert not wstatus.any() and not status.cpu().numpy().any()
        assert np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws) and np.array_equal(recid.cpu().numpy(), wrecid) and np.array_equal(hv(R), wR)
    pipe.close()


def test_a_tampered_batch_fails_alone(gpu_ctx, wallet):
    """a batch whose values make a check fail (a Paillier randomness of 0: the ciphertext is 0, decryption and the proofs break) gets
    its statuses; its neighbours in the same pass sign — exactly what mpe_gg20_sign says about the same arrays"""
    ctx = gpu_ctx
    lk, gk = wallet
    B = 4
    host = [G.make_nonces(lk, B, seed="pt-%d" % b) for b in range(3)]
    host[1]["r_a"][2:4] = 0                                         # session 1 of batch 1: both parties encrypt with r = 0
    dev = [{f: dv(ctx, v) for f, v in h.items()} for h in host]
    pipe = E.Gg20Pipeline(ctx, gk, B, group=3, lanes=1)
    tickets = [pipe.submit(d) for d in dev]
    for b, t in enumerate(tickets):
        r, s, recid, status = pipe.wait(t)
        r1, s1, recid1, status1 = E.gg20_sign(ctx, gk, dev[b], B)
        ctx.sync()
        assert np.array_equal(status.cpu().numpy(), status1.cpu().numpy()) and np.array_equal(hv(r), hv(r1)) and np.array_equal(hv(s), hv(s1))
        wstatus = G.oracle_sign(lk, host[b], B)[4]
        assert np.array_equal(status.cpu().numpy(), wstatus)
        assert (status.cpu().numpy() != 0).tolist() == ([False, True, False, False] if b == 1 else [False] * B)
    pipe.close()


def test_seeded_stream_equals_the_oracle_expanding_the_same_seed(gpu_ctx, wallet):
    import ossl
    ctx = gpu_ctx
    lk, gk = wallet
    B, nb = 5, 6
    msgs = [F.words([int.from_bytes(hashlib.sha256(b"m %d %d" % (b, i)).digest(), "big") for i in range(B)], 8) for b in range(nb)]
    pipe = E.Gg20Pipeline(ctx, gk, B, group=2, lanes=2)
    tickets = [pipe.submit_seeded(SEED, 1000 + b, dv(ctx, msgs[b])) for b in range(nb)]
    pipe.flush()
    for b, t in enumerate(tickets):
        r, s, recid, status = pipe.wait(t)
        z, fails = G.oracle_sample_nonces(lk, B, SEED, 1000 + b, msg=msgs[b])
        wr, ws, wrecid, _, wstatus = G.oracle_sign(lk, z, B)
        assert fails == 0 and not wstatus.any() and not status.cpu().numpy().any()
        assert np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws) and np.array_equal(recid.cpu().numpy(), wrecid)
        assert ossl.ecdsa_verify(lk["arrays"]["y"][0], msgs[b], wr, ws).all()
    assert pipe.sampler_failures() == 0
    pipe.close()


def test_pipeline_with_key_sets_and_argument_errors(gpu_ctx):
    ctx = gpu_ctx
    keys = F.load_keys()
    t, n, signers, K, B = 1, 3, [0, 1], 2, 4
    lks = [G.make_local_keys(keys[3 * kk:3 * kk + 3], t, n, signers, seed="pw%d" % kk) for kk in range(K)]
    arrays = {f: np.concatenate([lk["arrays"][f] for lk in lks]) for f in ("x", "p", "q", "Nt", "h1", "h2", "y", "X")}
    arrays["signers"] = lks[0]["arrays"]["signers"]
    lk = dict(t=t, n=n, S=2, arrays=arrays, nkeysets=K)
    gk = E.Gg20Keys(ctx, t, n, signers, arrays, nkeysets=K)
    pipe = E.Gg20Pipeline(ctx, gk, B, group=2, lanes=2)
    msg = F.words([7 + i for i in range(B)], 8)
    with pytest.raises(E.N_.MpeError):
        pipe.submit_seeded(SEED, 1, dv(ctx, msg))                                  # several wallets: every session must name its key set
    with pytest.raises(E.N_.MpeError):
        pipe.submit_seeded(SEED, 1 << 56, dv(ctx, msg), keyset=dv(ctx, np.zeros(B, dtype=np.int32)))
    sets = [np.array([0, 1, 1, 0], dtype=np.int32), np.array([1, 1, 0, 0], dtype=np.int32), np.array([0, 0, 0, 1], dtype=np.int32)]
    tickets = [pipe.submit_seeded(SEED, 50 + b, dv(ctx, msg), keyset=dv(ctx, sets[b])) for b in range(3)]
    for b, tk in enumerate(tickets):
        r, s, recid, status = pipe.wait(tk)
        z, _ = G.oracle_sample_nonces(lk, B, SEED, 50 + b, keyset=sets[b], msg=msg)
        res = G.oracle_sign_ex(lk, z, B, keyset=sets[b])
        assert not res["status"].any() and not status.cpu().numpy().any()
        assert np.array_equal(hv(r), res["r"]) and np.array_equal(hv(s), res["s"])
    with pytest.raises(E.N_.MpeError):
        pipe.wait(99)
    pipe.close()
    with pytest.raises(E.N_.MpeError):
        E.Gg20Pipeline(ctx, gk, B, group=2, lanes=5)
    gk.close()


def test_a_refused_submission_takes_no_slot_and_no_ticket(gpu_ctx, wallet):
    """a group holds seeded or caller-sampled batches, not both: the refused call must leave the open group, the ticket counter and the
    ring as they were — the next accepted batch gets the next ticket and the group still signs (mpe_pipeline.h: every refusal comes
    before a slot is taken)"""
    ctx = gpu_ctx
    lk, gk = wallet
    B = 4
    msg = F.words([11 + i for i in range(B)], 8)
    host = G.make_nonces(lk, B, seed="refused")
    dev = {f: dv(ctx, v) for f, v in host.items()}
    pipe = E.Gg20Pipeline(ctx, gk, B, group=3, lanes=1)
    t1 = pipe.submit_seeded(SEED, 7, dv(ctx, msg))
    for _ in range(3):
        with pytest.raises(E.N_.MpeError):
            pipe.submit(dev)                                          # caller-sampled values into a group that holds a seeded batch
    t2 = pipe.submit_seeded(SEED, 8, dv(ctx, msg))
    assert (t1, t2) == (1, 2)
    for b, t in ((7, t1), (8, t2)):
        r, s, recid, status = pipe.wait(t)
        z, fails = G.oracle_sample_nonces(lk, B, SEED, b, msg=msg)
        wr, ws, wrecid, _, wstatus = G.oracle_sign(lk, z, B)
        assert fails == 0 and not wstatus.any() and not status.cpu().numpy().any()
        assert np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws) and np.array_equal(recid.cpu().numpy(), wrecid)
    # ... and the other way round: a seeded batch into a group that holds caller-sampled values
    t3 = pipe.submit(dev)
    with pytest.raises(E.N_.MpeError):
        pipe.submit_seeded(SEED, 9, dv(ctx, msg))
    assert t3 == 3
    r, s, recid, status = pipe.wait(t3)
    wr, ws, wrecid, _, wstatus = G.oracle_sign(lk, host, B)
    assert not wstatus.any() and not status.cpu().numpy().any() and np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws)
    pipe.close()


def test_a_failed_pass_is_every_tickets_news_and_the_next_group_signs(gpu_ctx, wallet):
    """the pass of a group fails as a whole (fault injection: MPE_E_NOMEM after the inputs were staged): EVERY ticket of that group —
    not only the one whose submit closed it — reports the error from wait / ticket_rc / done, its status array holds
    MPE_GG20_STATUS_PASS_FAILED(rc) for every session and its signature arrays are zero; the submissions themselves succeeded; the
    next group signs and equals the oracle.  (The reference reports per session and never drops an error: gg_2020/mod.rs:23-27,
    rounds.rs:696-713.)"""
    ctx = gpu_ctx
    lk, gk = wallet
    B, group = 4, 3
    msgs = [F.words([int.from_bytes(hashlib.sha256(b"fp %d %d" % (b, i)).digest(), "big") for i in range(B)], 8) for b in range(2 * group)]
    for rc in (E.N_.MPE_E_NOMEM,):
        pipe = E.Gg20Pipeline(ctx, gk, B, group=group, lanes=2)
        pipe.inject_fault(1, rc)
        tickets = [pipe.submit_seeded(SEED, 300 + b, dv(ctx, msgs[b]), want_R=True) for b in range(2 * group)]     # no submit raises
        want_status = E.N_.gg20_status_pass_failed(rc)
        for b, t in enumerate(tickets):
            failed = b < group
            assert pipe.ticket_rc(t) == (True, rc if failed else 0)
            if failed:
                keep = pipe._keep[t]
                with pytest.raises(E.N_.MpeError):
                    pipe.wait(t)
                _, _, r, s, recid, status, R = keep
                assert pipe.done(t)
                assert (status.cpu().numpy() == want_status).all() and not hv(r).any() and not hv(s).any() and not recid.cpu().numpy().any() and not hv(R).any()
            else:
                r, s, recid, status, R = pipe.wait(t, want_R=True)
                z, fails = G.oracle_sample_nonces(lk, B, SEED, 300 + b, msg=msgs[b])
                wr, ws, wrecid, wR, wstatus = G.oracle_sign(lk, z, B)
                assert fails == 0 and not wstatus.any() and not status.cpu().numpy().any()
                assert np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws) and np.array_equal(recid.cpu().numpy(), wrecid) and np.array_equal(hv(R), wR)
        assert pipe.counters()["failed"] == 1 and pipe.counters()["groups"] == 2
        # check=False hands the failed batch's arrays over instead of raising; the other error code travels the same way
        pipe.inject_fault(1, E.N_.MPE_E_HIP)
        t = pipe.submit_seeded(SEED, 399, dv(ctx, msgs[0]))
        r, s, recid, status = pipe.wait(t, check=False)
        assert pipe.counters()["failed"] == 2
        assert (status.cpu().numpy() == E.N_.gg20_status_pass_failed(E.N_.MPE_E_HIP)).all() and not hv(r).any()
        pipe.close()


def test_deadline_and_idle_lane_launch_part_filled_groups(gpu_ctx, wallet):
    """arrival-driven grouping: with `eager` a batch that finds its lane idle starts at once (group of one); with a deadline a
    part-filled group goes when its oldest batch has waited that long — checked inside submit / done / poll of the one host thread.
    Results do not depend on how the batches were grouped."""
    import time
    ctx = gpu_ctx
    lk, gk = wallet
    B = 4
    msgs = [F.words([int.from_bytes(hashlib.sha256(b"dl %d %d" % (b, i)).digest(), "big") for i in range(B)], 8) for b in range(6)]

    def check(pipe, tickets, first):
        for b, t in enumerate(tickets):
            r, s, recid, status = pipe.wait(t)
            z, _ = G.oracle_sample_nonces(lk, B, SEED, first + b, msg=msgs[b])
            wr, ws, wrecid, _, wstatus = G.oracle_sign(lk, z, B)
            assert not wstatus.any() and not status.cpu().numpy().any() and np.array_equal(hv(r), wr) and np.array_equal(hv(s), ws)

    # eager: the first batch finds lane 0 idle and goes alone; the second finds lane 1 idle and goes alone; further batches find both
    # lanes busy and share a group until something makes it go
    pipe = E.Gg20Pipeline(ctx, gk, B, group=4, lanes=2)
    pipe.set_eager(True)
    t = [pipe.submit_seeded(SEED, 500 + b, dv(ctx, msgs[b])) for b in range(5)]
    assert pipe.ticket_rc(t[0])[0] and pipe.ticket_rc(t[1])[0]
    c = pipe.counters()
    assert c["by_idle"] >= 2 and c["groups"] >= 2
    check(pipe, t, 500)
    pipe.close()
    # deadline: nothing goes at submit time (group of 4, 2 batches), poll() before the deadline launches nothing, after it the group goes
    pipe = E.Gg20Pipeline(ctx, gk, B, group=4, lanes=1)
    pipe.set_deadline_us(200_000)
    t = [pipe.submit_seeded(SEED, 600 + b, dv(ctx, msgs[b])) for b in range(2)]
    assert not pipe.ticket_rc(t[0])[0] and not pipe.poll()
    time.sleep(0.25)
    assert pipe.poll() and pipe.ticket_rc(t[0])[0] and pipe.ticket_rc(t[1])[0]
    assert pipe.counters()["by_deadline"] == 1
    # ... and done() alone drives it too: a service that only polls its tickets never strands a batch
    t2 = pipe.submit_seeded(SEED, 602, dv(ctx, msgs[2]))
    time.sleep(0.25)
    pipe.done(t2)
    assert pipe.ticket_rc(t2)[0] and pipe.counters()["by_deadline"] == 2
    check(pipe, t + [t2], 600)
    pipe.close()
