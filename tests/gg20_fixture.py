"""GG20 signing fixtures: a `LocalKey` set (Shamir shares of the ECDSA key + the Paillier / N~ material of
tests/golden/keys16.json, as the reference's keygen would leave it: keygen/rounds.rs:283-322) and per-session
nonces drawn with the reference's ranges from a seeded stream.  Array layouts = oracle/mpe_oracle.h."""
import ctypes as C

import numpy as np

import fixtures as F
import pyref

Q = pyref.Q


class KeysStruct(C.Structure):
    _fields_ = [("t", C.c_int), ("n", C.c_int), ("S", C.c_int), ("nkeysets", C.c_int), ("signers", C.c_void_p)] + \
               [(f, C.c_void_p) for f in ("x", "p", "q", "N", "Nt", "h1", "h2", "y", "X")]


NONCE_FIELDS = ["k", "gamma", "blind", "r_a", "al_alpha", "al_beta", "al_gamma", "al_rho", "mb_beta_tag", "mb_r",
                "mb_nonce_b", "mb_nonce_bt", "l", "ped_s1", "ped_s2", "pdl_alpha", "pdl_beta", "pdl_rho", "pdl_gamma",
                "heg_s1", "heg_s2", "msg"]


class NoncesStruct(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in NONCE_FIELDS]


def make_local_keys(keys, t, n, signers, seed="keygen"):
    """Shamir-share a secret x with a degree-t polynomial (the sum polynomial of GG20 keygen); party i holds
    x_i = F(i+1), pk_vec[i] = x_i G, y = x G; Paillier/N~ material of party i = keys[i]."""
    r = F.Rng(seed)
    coef = [r.below(Q - 1) + 1 for _ in range(t + 1)]
    xs = [sum(c * pow(i + 1, e, Q) for e, c in enumerate(coef)) % Q for i in range(n)]
    y = pyref.ec_mul(coef[0], pyref.G)
    X = [pyref.ec_mul(x, pyref.G) for x in xs]
    arr = dict(signers=np.array(signers, dtype=np.int32), x=F.words(xs, 8), p=F.words([keys[i].p for i in range(n)], 32),
               q=F.words([keys[i].q for i in range(n)], 32), Nt=F.words([keys[i].Nt for i in range(n)], 64),
               h1=F.words([keys[i].h1 for i in range(n)], 64), h2=F.words([keys[i].h2 for i in range(n)], 64),
               y=F.point_words([y]), X=F.point_words(X))
    return dict(t=t, n=n, S=len(signers), arrays=arr, y=y, x=coef[0], keys=keys[:n])


def keys_struct(lk):
    s = KeysStruct()
    s.t, s.n, s.S, s.nkeysets = lk["t"], lk["n"], lk["S"], lk.get("nkeysets", 1)
    for f, a in lk["arrays"].items():
        setattr(s, f, a.ctypes.data)
    return s


def make_nonces(lk, B, seed="sessions"):
    """Every value the reference samples during signing (party_i.rs:559-563,574,627; mta/mod.rs:57,97-98;
    range_proofs.rs:48-51; zk_pdl_with_slack/mod.rs:73-77; curv sigma proofs), per session."""
    r = F.Rng(seed)
    S, n, keys = lk["S"], lk["n"], lk["keys"]
    signers = list(lk["arrays"]["signers"])
    P = S * (S - 1)
    q3 = Q ** 3
    z = {f: [] for f in NONCE_FIELDS}
    sc = lambda: r.below(Q - 1) + 1
    for b in range(B):
        for i in range(S):
            me = keys[signers[i]]
            z["k"].append(sc()); z["gamma"].append(sc()); z["blind"].append(r.bits(256)); z["r_a"].append(r.below(me.N))
            z["l"].append(sc()); z["ped_s1"].append(sc()); z["ped_s2"].append(sc()); z["heg_s1"].append(sc()); z["heg_s2"].append(sc())
            for st in range(n):
                nn = F.alice_nonces(r, me, keys[st])
                for f in ("alpha", "beta", "gamma", "rho"):
                    z["al_" + f].append(nn[f])
            for jj in range(S - 1):
                ind = jj if jj < i else jj + 1
                alice = keys[signers[ind]]
                for v in range(2):
                    z["mb_beta_tag"].append(r.below(alice.N)); z["mb_r"].append(r.below(alice.N))
                    z["mb_nonce_b"].append(sc()); z["mb_nonce_bt"].append(sc())
                nn = F.pdl_nonces(r, me, keys[signers[ind]])
                for f in ("alpha", "beta", "rho", "gamma"):
                    z["pdl_" + f].append(nn[f])
        z["msg"].append(int.from_bytes(__import__("hashlib").sha256(b"session %d" % b).digest(), "big"))
    widths = dict(k=8, gamma=8, blind=8, r_a=64, al_alpha=24, al_beta=64, al_gamma=88, al_rho=72, mb_beta_tag=64, mb_r=64,
                  mb_nonce_b=8, mb_nonce_bt=8, l=8, ped_s1=8, ped_s2=8, pdl_alpha=24, pdl_beta=64, pdl_rho=72, pdl_gamma=88,
                  heg_s1=8, heg_s2=8, msg=8)
    return {f: F.words(v, widths[f]) for f, v in z.items()}


def nonces_struct(arrs):
    s = NoncesStruct()
    for f in NONCE_FIELDS:
        setattr(s, f, arrs[f].ctypes.data)
    return s


def oracle_sign(lk, nonces, B, first=0, count=None):
    import orc
    count = B if count is None else count
    r, s = np.zeros((B, 8), dtype=np.uint32), np.zeros((B, 8), dtype=np.uint32)
    recid, status = np.zeros(B, dtype=np.int32), np.full(B, -1, dtype=np.int32)
    R = np.zeros((B, 16), dtype=np.uint32)
    ks, ns = keys_struct(lk), nonces_struct(nonces)
    orc.lib.orc_gg20_sign(C.byref(ks), C.byref(ns), first, count, orc._p(r), orc._p(s), orc._p(recid), orc._p(R), orc._p(status))
    return r, s, recid, R, status


ROUNDS = [0, 1, 2, 3, 4, 5, 7]                     # rounds that emit a message (M0..M6)


def msg_words(S, n, rnd):
    return {0: 256 * (n + 1), 1: 208 * 2 * (S - 1), 2: 96, 3: 24, 4: 450 * S, 5: 64, 7: 8}[rnd]


def oracle_sign_ex(lk, nonces, B, keyset=None):
    """All parties in lock-step on the oracle, every round message kept.
    Returns dict(slabs={round: [S][B][W] uint32}, r, s, recid, R, status [B], party_status [S][B], party_bad [S][B])."""
    import orc
    S, n = lk["S"], lk["n"]
    slabs = {rnd: np.zeros((S, B, msg_words(S, n, rnd)), dtype=np.uint32) for rnd in ROUNDS}
    ptrs = (C.c_void_p * 7)(*[slabs[rnd].ctypes.data for rnd in ROUNDS])
    r, s = np.zeros((B, 8), dtype=np.uint32), np.zeros((B, 8), dtype=np.uint32)
    recid, status = np.zeros(B, dtype=np.int32), np.full(B, -1, dtype=np.int32)
    R = np.zeros((B, 16), dtype=np.uint32)
    pst, pbad = np.zeros((S, B), dtype=np.int32), np.zeros((S, B), dtype=np.uint32)
    ks, ns = keys_struct(lk), nonces_struct(nonces)
    kset = None if keyset is None else np.ascontiguousarray(keyset, dtype=np.int32)
    orc.lib.orc_gg20_sign_ex(C.byref(ks), C.byref(ns), orc._p(kset), B, 0, B, ptrs, orc._p(r), orc._p(s), orc._p(recid), orc._p(R),
                             orc._p(status), orc._p(pst), orc._p(pbad))
    return dict(slabs=slabs, r=r, s=s, recid=recid, R=R, status=status, party_status=pst, party_bad=pbad)


def py_parties(lk, nonces, b):
    """pyref_gg20.Party objects of session b (every party only gets its own secrets)."""
    import pyref_gg20 as PG
    S, n, keys = lk["S"], lk["n"], lk["keys"]
    signers = [int(x) for x in lk["arrays"]["signers"]]
    xs = F.ints(lk["arrays"]["x"])
    pub = dict(n=n, signers=signers, N=[k.N for k in keys], Nt=[k.Nt for k in keys], h1=[k.h1 for k in keys], h2=[k.h2 for k in keys],
               X=F.points(lk["arrays"]["X"]), y=F.points(lk["arrays"]["y"])[0])
    g = lambda f, ix: F.ints(nonces[f][ix:ix + 1])[0]
    out = []
    for i in range(S):
        me = signers[i]
        pi = b * S + i
        z = dict(k=g("k", pi), gamma=g("gamma", pi), blind=g("blind", pi), r_a=g("r_a", pi), l=g("l", pi), ped_s1=g("ped_s1", pi),
                 ped_s2=g("ped_s2", pi), heg_s1=g("heg_s1", pi), heg_s2=g("heg_s2", pi))
        z["al"] = [{f: g("al_" + f, pi * n + st) for f in ("alpha", "beta", "gamma", "rho")} for st in range(n)]
        z["mb"] = [[dict(beta_tag=g("mb_beta_tag", (pi * (S - 1) + jj) * 2 + v), r=g("mb_r", (pi * (S - 1) + jj) * 2 + v),
                         nonce_b=g("mb_nonce_b", (pi * (S - 1) + jj) * 2 + v), nonce_bt=g("mb_nonce_bt", (pi * (S - 1) + jj) * 2 + v))
                    for v in range(2)] for jj in range(S - 1)]
        z["pdl"] = [{f: g("pdl_" + f, pi * (S - 1) + jj) for f in ("alpha", "beta", "rho", "gamma")} for jj in range(S - 1)]
        out.append(PG.Party(i, dict(pub, x_i=xs[me], p=keys[me].p, q=keys[me].q), z))
    return out


def py_session(lk, nonces, b):
    """(slab bytes per round: {round: [S] bytes}, per-party (r, s, recid), per-party status) of session b by the Python restatement"""
    import pyref_gg20 as PG
    parties = py_parties(lk, nonces, b)
    msgs, sigs = PG.simulate(parties, F.ints(nonces["msg"][b:b + 1])[0])
    S, n = lk["S"], lk["n"]
    packed = {rnd: [PG.pack(rnd, m, S, n) for m in msgs[q]] for q, rnd in enumerate(ROUNDS)}
    return packed, sigs, [(p.status, p.bad) for p in parties]


class OracleParty:
    """One party (signer ordinal `ord`) over B sessions on the CPU oracle: `RoundN::proceed` as functions of (state, messages).
    The key struct it receives holds ONLY this party's secrets (the other parties' x, p, q rows are zero)."""

    def __init__(self, lk, ord_, B, nonces_party, keyset=None):
        import orc
        self.orc, self.lk, self.ord, self.B = orc, lk, ord_, B
        self.S, self.n = lk["S"], lk["n"]
        me = int(lk["arrays"]["signers"][ord_])
        arr = dict(lk["arrays"])
        K = lk.get("nkeysets", 1)
        if "N" not in arr:
            arr["N"] = F.words([p_ * q_ for p_, q_ in zip(F.ints(arr["p"]), F.ints(arr["q"]))], 64)
        for f in ("x", "p", "q"):                      # nobody else's secrets
            a = arr[f].copy()
            mask = np.ones(a.shape[0], dtype=bool)
            mask[[kk * self.n + me for kk in range(K)]] = False
            a[mask] = 0
            arr[f] = a
        self._lk = dict(lk, arrays=arr)
        self._nonces = {f: np.ascontiguousarray(v) for f, v in nonces_party.items()}
        self._keyset = None if keyset is None else np.ascontiguousarray(keyset, dtype=np.int32)
        self._ks, self._ns = keys_struct(self._lk), nonces_struct(self._nonces)
        orc.lib.orc_gg20_party_new.restype = C.c_void_p
        self.h = C.c_void_p(orc.lib.orc_gg20_party_new(C.byref(self._ks), ord_, B, C.byref(self._ns), 1, 0, orc._p(self._keyset)))
        assert self.h

    def round(self, rnd, d_in=None, in_off=None):
        """d_in: numpy uint32 slab of the previous round (all senders); returns this party's [B][W] block or None"""
        W = msg_words(self.S, self.n, rnd) if rnd in ROUNDS else 0
        out = np.zeros((self.B, W), dtype=np.uint32) if W else None
        off = None if in_off is None else (C.c_int64 * self.S)(*[int(x) for x in in_off])
        self.orc.lib.orc_gg20_party_round(self.h, rnd, self.orc._p(d_in), off, self.orc._p(out), 0, self.B)
        return out

    def fault(self, step):
        self.orc.lib.orc_gg20_party_fault(self.h, step)

    def result(self):
        B = self.B
        o = dict(status=np.zeros(B, dtype=np.int32), bad_actors=np.zeros(B, dtype=np.uint32), r=np.zeros((B, 8), dtype=np.uint32),
                 s=np.zeros((B, 8), dtype=np.uint32), recid=np.zeros(B, dtype=np.int32), R=np.zeros((B, 16), dtype=np.uint32))
        self.orc.lib.orc_gg20_party_result(self.h, *[self.orc._p(o[f]) for f in ("status", "bad_actors", "r", "s", "recid", "R")])
        return o

    def close(self):
        if self.h:
            self.orc.lib.orc_gg20_party_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def party_nonces(nonces, lk, i):
    """the [B][1] slice of the [B][S] nonce arrays that belongs to signer ordinal i (msg stays whole)"""
    S, n = lk["S"], lk["n"]
    per = dict(k=1, gamma=1, blind=1, r_a=1, l=1, ped_s1=1, ped_s2=1, heg_s1=1, heg_s2=1, al_alpha=n, al_beta=n, al_gamma=n, al_rho=n,
               mb_beta_tag=2 * (S - 1), mb_r=2 * (S - 1), mb_nonce_b=2 * (S - 1), mb_nonce_bt=2 * (S - 1), pdl_alpha=S - 1, pdl_beta=S - 1,
               pdl_rho=S - 1, pdl_gamma=S - 1)
    out = {}
    for f, v in nonces.items():
        if f == "msg":
            out[f] = v
            continue
        B = v.shape[0] // (S * per[f])
        out[f] = np.ascontiguousarray(v.reshape(B, S, per[f], v.shape[1])[:, i].reshape(B * per[f], v.shape[1]))
    return out


def run_rounds(parties, msg, tamper=None):
    """Drives any set of per-party engines (OracleParty or GPU adapters exposing .round(rnd, slab) -> [B][W] numpy and
    .result()) through the protocol with an in-memory relay.  tamper(rnd, slab[S,B,W]) may modify a round's messages in place.
    Returns the slabs {round: [S,B,W]}."""
    S = len(parties)
    slabs = {}
    prev = None
    for rnd in [0, 1, 2, 3, 4, 5, 6, 7, 8]:
        outs = [p.round(rnd, prev) if rnd != 7 else p.round(7, msg) for p in parties]
        if rnd in ROUNDS:
            slab = np.ascontiguousarray(np.stack(outs))
            if tamper:
                tamper(rnd, slab)
            slabs[rnd] = slab
            prev = slab
    return slabs


# ---- identifiable abort (gg_2020/blame.rs): the openings every signer publishes, assembled from the nonces and the slabs ----
def blame5_opened(lk, nonces, slabs, B):
    """GlobalStatePhase5 (blame.rs:42-57) as arrays: [B][S] then peer slot j"""
    S, n = lk["S"], lk["n"]
    P1 = S - 1
    o = dict(k=nonces["k"].copy(), k_rand=nonces["r_a"].copy(), gamma=nonces["gamma"].copy())
    bt, br = np.zeros((B * S * P1, 64), dtype=np.uint32), np.zeros((B * S * P1, 64), dtype=np.uint32)
    cb = np.zeros((B * S * P1, 128), dtype=np.uint32)
    for b in range(B):
        for i in range(S):
            for j in range(P1):
                ind = j if j < i else j + 1
                jme = i if i < ind else i - 1
                src = ((b * S + ind) * P1 + jme) * 2 + 0                      # Bob ind's gamma MessageB to Alice i
                dst = (b * S + i) * P1 + j
                bt[dst], br[dst] = nonces["mb_beta_tag"][src], nonces["mb_r"][src]
                cb[dst] = slabs[1][ind, b, (jme * 2 + 0) * 208:(jme * 2 + 0) * 208 + 128]
    o.update(beta_tag=bt, beta_rand=br, c_b=cb)
    o["delta"] = np.ascontiguousarray(np.transpose(slabs[2][:, :, 0:8], (1, 0, 2)).reshape(B * S, 8))
    o["g_gamma"] = np.ascontiguousarray(np.transpose(slabs[3][:, :, 8:24], (1, 0, 2)).reshape(B * S, 16))
    o["c_a"] = np.ascontiguousarray(np.transpose(slabs[0][:, :, n * 256:n * 256 + 128], (1, 0, 2)).reshape(B * S, 128))
    return o


def blame6_cb(lk, slabs, B):
    """m_b_mat[i][j].c of the w_i MtAs: [B][S][S-1][128]"""
    S = lk["S"]
    P1 = S - 1
    cb = np.zeros((B * S * P1, 128), dtype=np.uint32)
    for b in range(B):
        for i in range(S):
            for j in range(P1):
                ind = j if j < i else j + 1
                jme = i if i < ind else i - 1
                cb[(b * S + i) * P1 + j] = slabs[1][ind, b, (jme * 2 + 1) * 208:(jme * 2 + 1) * 208 + 128]
    return cb


def oracle_blame(lk, which, opened, B, keyset=None):
    import orc
    S = lk["S"]
    bad = np.zeros(B, dtype=np.uint32)
    fields = dict(b5=["k", "k_rand", "gamma", "beta_tag", "beta_rand", "delta", "g_gamma", "c_a", "c_b"],
                  b6=["k", "k_rand", "miu", "miu_rand", "a1", "a2", "z", "S", "c_a", "c_b", "R"], b7=["s", "r", "R_dash", "m", "R", "S"])[which]
    keep = [np.ascontiguousarray(opened[f]) for f in fields]
    st = (C.c_void_p * len(fields))(*[a.ctypes.data for a in keep])
    if which == "b7":
        orc.lib.orc_gg20_blame7(S, B, st, orc._p(bad))
    else:
        ks = keys_struct(lk)
        kset = None if keyset is None else np.ascontiguousarray(keyset, dtype=np.int32)
        getattr(orc.lib, "orc_gg20_blame5" if which == "b5" else "orc_gg20_blame6")(C.byref(ks), orc._p(kset), B, st, orc._p(bad))
    return bad


NONCE_WIDTHS = dict(k=8, gamma=8, blind=8, r_a=64, al_alpha=24, al_beta=64, al_gamma=88, al_rho=72, mb_beta_tag=64, mb_r=64,
                    mb_nonce_b=8, mb_nonce_bt=8, l=8, ped_s1=8, ped_s2=8, pdl_alpha=24, pdl_beta=64, pdl_rho=72, pdl_gamma=88,
                    heg_s1=8, heg_s2=8, msg=8)


def nonce_rows(S, n, L, B):
    P = L * (S - 1)
    return dict(k=B * L, gamma=B * L, blind=B * L, r_a=B * L, al_alpha=B * L * n, al_beta=B * L * n, al_gamma=B * L * n, al_rho=B * L * n,
                mb_beta_tag=B * P * 2, mb_r=B * P * 2, mb_nonce_b=B * P * 2, mb_nonce_bt=B * P * 2, l=B * L, ped_s1=B * L, ped_s2=B * L,
                pdl_alpha=B * P, pdl_beta=B * P, pdl_rho=B * P, pdl_gamma=B * P, heg_s1=B * L, heg_s2=B * L, msg=B)


def oracle_sample_nonces(lk, B, seed, counter, local=None, keyset=None, msg=None):
    """oracle/sampler_oracle.c: the arrays mpe_gg20_sample_nonces must produce for (seed, counter); returns (dict, failures)"""
    import orc
    S, n = lk["S"], lk["n"]
    local = list(range(S)) if local is None else list(local)
    rows = nonce_rows(S, n, len(local), B)
    z = {f: np.zeros((rows[f], NONCE_WIDTHS[f]), dtype=np.uint32) for f in NONCE_FIELDS}
    if msg is not None:
        z["msg"] = np.ascontiguousarray(msg, dtype=np.uint32)
    ks, ns = keys_struct(lk), nonces_struct(z)
    lc = np.asarray(local, dtype=np.int32)
    kset = None if keyset is None else np.ascontiguousarray(keyset, dtype=np.int32)
    fails = orc.lib.orc_gg20_sample_nonces(C.byref(ks), B, len(local), orc._p(lc), orc._p(kset), bytes(seed), counter, C.byref(ns))
    return z, fails
