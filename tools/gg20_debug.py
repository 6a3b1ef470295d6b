import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import fixtures as F, gg20_fixture as G
from multi_party_ecdsa_amd import engine as E
keys = F.load_keys()
ctx = E.Context(0)
t, n, signers, B = 1, 3, [0, 1], 2
lk = G.make_local_keys(keys, t, n, signers)
nonces = G.make_nonces(lk, B, seed="dbg")
print("fixtures ready", flush=True)
gk = E.Gg20Keys(ctx, t, n, signers, lk["arrays"]); ctx.sync()
print("keys ready", flush=True)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(ctx.device)
out = E.gg20_sign(ctx, gk, {f: dv(v) for f, v in nonces.items()}, B, want_R=True)
ctx.sync()
r, s, recid, status, R = [o.cpu().numpy() for o in out]
print("status", status, flush=True)
wr, ws, wrecid, wR, wst = G.oracle_sign(lk, nonces, B)
print("oracle status", wst, "match r", np.array_equal(r.view(np.uint32), wr), "s", np.array_equal(s.view(np.uint32), ws), "recid", list(recid) == list(wrecid))
