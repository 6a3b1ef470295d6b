"""-m gpu: a party process signs batch after batch on the same objects, and creates / destroys contexts, wallets and sessions
over its life — device memory must reach a steady state in both senses.  The library allocates with hipMalloc (not through
torch), so the check reads the DEVICE's free memory (`torch.cuda.mem_get_info`) around the calls:
 * `mpe_gg20_sign` (all parties local) six times on one context: no growth from call 3 on, identical signatures;
 * one `mpe_gg20_session` re-armed five times (the service loop of bench.py --mode party): no growth after the second batch;
 * closing the key object returns its fixed-base tables (GBs);
 * three whole life cycles (context + keys + session created, used, destroyed): the free memory at the end of cycle 3 equals
   that at the end of cycle 2.  (Cycle 1 is the warm-up: the HIP runtime keeps, per hardware queue and for the life of the
   process, the scratch it sized for the kernels with the largest private segment — the EC round kernels, 3.7 KB per lane —
   about 1.9 GB per queue; that is the runtime's, not an allocation of this library.)"""
import gc

import numpy as np
import pytest
import torch

import gg20_fixture as G

pytestmark = pytest.mark.gpu
SLACK = 8 << 20          # bytes: the runtime may keep a few small blocks


def _dev(ctx, arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(ctx.device)


def _free(dev, ctx=None):
    if ctx is not None:
        ctx.sync()
    torch.cuda.synchronize(dev)
    return torch.cuda.mem_get_info(dev)[0]


def _life_cycle(dev, lk, nonces, B, signers, want, first):
    from multi_party_ecdsa_amd import engine as E
    ctx = E.Context(0)
    gk = E.Gg20Keys(ctx, lk["t"], lk["n"], signers, lk["arrays"])
    dn = {f: _dev(ctx, v) for f, v in nonces.items()}
    free, sigs = [], []
    for it in range(6 if first else 2):
        r, s, recid, status = E.gg20_sign(ctx, gk, dn, B)
        free.append(_free(dev, ctx))
        sigs.append((r.cpu().numpy().copy(), s.cpu().numpy().copy()))
        assert not status.cpu().numpy().any()
        del r, s, recid, status
    assert np.array_equal(sigs[0][0][:4].view(np.uint32), want[0]) and np.array_equal(sigs[0][1][:4].view(np.uint32), want[1])
    assert all(np.array_equal(sigs[0][0], x[0]) and np.array_equal(sigs[0][1], x[1]) for x in sigs[1:])
    if first:
        assert abs(free[2] - free[5]) <= SLACK, [f - free[5] for f in free]               # no growth from call 3 on

    # the service loop: one session object, re-armed for every batch
    sess = E.Gg20Session(ctx, gk, B, list(range(len(signers))), dn)
    msg = _dev(ctx, nonces["msg"])
    free2 = []
    for it in range(5 if first else 2):
        prev = None
        for rnd in range(9):
            out = sess.round(rnd, d_in=prev, msg=msg if rnd == 7 else None)
            if out is not None:
                prev = out.reshape(-1)
        res = sess.result()
        assert not res["status"].cpu().numpy().any()
        assert np.array_equal(res["r"].cpu().numpy()[0], sigs[0][0])
        free2.append(_free(dev, ctx))
        del res, out, prev
        sess.rearm(dn)                                                  # the same sampled values: a memory test, never a deployment pattern
    if first:
        assert abs(free2[1] - free2[4]) <= SLACK, [f - free2[4] for f in free2]
    sess.close()
    before_keys = _free(dev, ctx)
    gk.close()
    tables = _free(dev, ctx) - before_keys
    assert tables > (1 << 30), tables                                  # the fixed-base tables of 2 n bases came back
    ctx.close()
    del sess, gk, dn, msg, ctx
    gc.collect()
    torch.cuda.empty_cache()
    return _free(dev)


def test_memory_reaches_a_steady_state_within_and_across_object_lifetimes(keys):
    torch.cuda.init()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    t, n, signers, B = 1, 3, [0, 2], 48
    lk = G.make_local_keys(keys, t, n, signers)
    nonces = G.make_nonces(lk, B, seed="soak")
    want = G.oracle_sign(lk, nonces, 4)                               # the first sessions against the oracle
    ends = [_life_cycle(dev, lk, nonces, B, signers, want, first=(c == 0)) for c in range(3)]
    print("soak: KB of free memory relative to the end of cycle 3:", [(e - ends[2]) >> 10 for e in ends])
    assert abs(ends[1] - ends[2]) <= SLACK, [(e - ends[2]) >> 10 for e in ends]
