import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import fixtures as F, gg20_fixture as G
from multi_party_ecdsa_amd import engine as E
ctx = E.Context(0)
lk = G.make_local_keys(F.load_keys(), 1, 3, [0, 1])
gk = E.Gg20Keys(ctx, 1, 3, [0, 1], lk["arrays"])
for B in (1024, 65536):
    out, fail = E.gg20_sample_nonces(ctx, gk, B, bytes(32), 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(3):
        out, fail = E.gg20_sample_nonces(ctx, gk, B, bytes(32), 2 + c, out=out)
    torch.cuda.synchronize()
    print("sample", B, "sessions:", (time.perf_counter() - t0) / 3 * 1e3, "ms; fails", int(fail.item()))
