"""Independent pure-Python restatement of ONE GG20 signing session, party by party, written from the reference text
(ZenGo-X/multi-party-ecdsa v0.8.1; citations relative to /root/reference/src) — NOT from oracle/gg20_oracle.c.
Its job: pin the C oracle (two restatements in two languages must produce byte-identical round messages).

  Round0..Round7                protocols/multi_party_ecdsa/gg_2020/state_machine/sign/rounds.rs:68-692
  SignKeys / LocalSignature     protocols/multi_party_ecdsa/gg_2020/party_i.rs:526-936
  MessageA / MessageB           utilities/mta/mod.rs:52-179
  curv sigma proofs             SURVEY.md App. A.3 (un-vendored; recalled)

Every party is a `Party` object holding only what the reference's RoundN structs hold; messages are Python dicts; `pack_*`
serialises them into the word layout of include/mpecdsa_hip.h so that tests compare bytes."""
import hashlib

import pyref as R

Q, G, H2 = R.Q, R.G, R.H2


def scalar_hash(points):
    """Sha256::new().chain_points(..).result_scalar() over a proof's canonical point list — 5 points = PedersenProof
    (g, h, com, a1, a2), 7 = HomoELGamalProof (T, A3, G, H, Y, D, E), 6 = ECDDHProof (g1, h1, g2, h2, a1, a2) — in the byte
    form and order of the profile in force (pyref.ENC; default: 65-byte uncompressed points, the order listed)"""
    return R.hash_points_scalar(points, {5: "ord_pedersen", 7: "ord_heg", 6: "ord_ecddh"}[len(points)])


def hash_commitment(point, blind):
    """HashCommitment::create_commitment_with_user_defined_randomness(BigInt::from_bytes(P.to_bytes(true)), blind)
    (party_i.rs:577-580)"""
    return R.hash_bigints([R.pt_as_bigint(point), blind])


def lagrange(signers, i):
    """VerifiableSS::map_share_to_new_params(params, s[i], s): basis at 0 over the points s_j + 1"""
    num = den = 1
    for j, sj in enumerate(signers):
        if j != i:
            num = num * (sj + 1) % Q
            den = den * ((sj + 1) - (signers[i] + 1)) % Q
    return num * pow(den, -1, Q) % Q


def pedersen_prove(m, r, s1, s2):
    com = R.ec_add(R.ec_mul(m, G), R.ec_mul(r, H2))
    a1, a2 = R.ec_mul(s1, G), R.ec_mul(s2, H2)
    e = scalar_hash([G, H2, com, a1, a2])
    return dict(e=e, a1=a1, a2=a2, com=com, z1=(s1 + e * m) % Q, z2=(s2 + e * r) % Q)


def pedersen_verify(pr):
    e = scalar_hash([G, H2, pr["com"], pr["a1"], pr["a2"]])
    lhs = R.ec_add(R.ec_mul(pr["z1"], G), R.ec_mul(pr["z2"], H2))
    return lhs == R.ec_add(R.ec_add(pr["a1"], pr["a2"]), R.ec_mul(e, pr["com"]))


def heg_prove(x, r, s1, s2, Gp, H, Y, D, E):
    A1, A2, A3 = R.ec_mul(s1, H), R.ec_mul(s2, Y), R.ec_mul(s2, Gp)
    T = R.ec_add(A1, A2)
    e = scalar_hash([T, A3, Gp, H, Y, D, E])
    return dict(T=T, A3=A3, z1=(s1 + e * x) % Q if x else s1, z2=(s2 + e * r) % Q)


def heg_verify(pr, Gp, H, Y, D, E):
    e = scalar_hash([pr["T"], pr["A3"], Gp, H, Y, D, E])
    ok1 = R.ec_add(R.ec_mul(pr["z1"], H), R.ec_mul(pr["z2"], Y)) == R.ec_add(pr["T"], R.ec_mul(e, D))
    ok2 = R.ec_mul(pr["z2"], Gp) == R.ec_add(pr["A3"], R.ec_mul(e, E))
    return ok1 and ok2


class Party:
    """One signer (ordinal i of s_l).  lk: dict(n, signers, x_i, p, q, N[n], Nt[n], h1[n], h2[n], X[n], y).
    z: this party's sampled values: k, gamma, blind, r_a, al[st]{alpha,beta,gamma,rho}, mb[jj][v]{beta_tag,r,nonce_b,
    nonce_bt}, l, ped_s1, ped_s2, pdl[jj]{alpha,beta,rho,gamma}, heg_s1, heg_s2."""

    def __init__(self, i, lk, z):
        self.i, self.lk, self.z = i, lk, z
        self.S, self.n = len(lk["signers"]), lk["n"]
        self.status, self.bad = 0, []

    def fail(self, code, bad=()):
        if not self.status:
            self.status, self.bad = code, list(bad)

    def ind(self, jj):
        return jj if jj < self.i else jj + 1

    # Round0::proceed (rounds.rs:68-104)
    def round0(self):
        lk, z, sg = self.lk, self.z, self.lk["signers"]
        me = sg[self.i]
        self.k, self.gamma = z["k"] % Q, z["gamma"] % Q
        self.w = lagrange(sg, self.i) * lk["x_i"] % Q                               # SignKeys::create party_i.rs:546-571
        self.g_gamma = R.ec_mul(self.gamma, G)
        self.com = hash_commitment(self.g_gamma, z["blind"])                        # phase1_broadcast :573-589
        N = lk["N"][me]
        self.c = R.paillier_encrypt(N, self.k, z["r_a"])                            # MessageA::a mta/mod.rs:62-87
        proofs = [R.alice_generate(N, lk["Nt"][st], lk["h1"][st], lk["h2"][st], self.k, self.c, z["r_a"], **z["al"][st])
                  for st in range(self.n)]
        return dict(c=self.c, range_proofs=proofs, com=self.com)

    # Round1::proceed (rounds.rs:122-206)
    def round1(self, m0):
        lk, z, sg = self.lk, self.z, self.lk["signers"]
        self.m_a_vec, self.bc_vec = [m["c"] for m in m0], [m["com"] for m in m0]
        self.beta = []
        out = []
        for jj in range(self.S - 1):
            ind = self.ind(jj)
            alice = sg[ind]
            N = lk["N"][alice]
            NN = N * N
            pair, betas = [], []
            for v, b in enumerate((self.gamma, self.w)):                            # MessageB::b twice per peer :151-175
                for st in range(self.n):                                             # mta/mod.rs:119-131
                    if not R.alice_verify(N, lk["Nt"][st], lk["h1"][st], lk["h2"][st], m0[ind]["c"], m0[ind]["range_proofs"][st]):
                        self.fail(101)
                nn = z["mb"][jj][v]
                c_bt = R.paillier_encrypt(N, nn["beta_tag"], nn["r"])                 # :133-137
                c_b = pow(m0[ind]["c"], b, NN) * c_bt % NN                            # Paillier::mul, Paillier::add :140-145
                bt = nn["beta_tag"] % Q
                betas.append((-bt) % Q)                                               # :146
                pair.append(dict(c=c_b, b_proof=R.dlog_prove(b, nn["nonce_b"] % Q), beta_tag_proof=R.dlog_prove(bt, nn["nonce_bt"] % Q)))
            out.append(pair)
            self.beta.append(betas)
        return out                                                                    # out[jj][v]

    # Round2::proceed (rounds.rs:234-317)
    def round2(self, m1):
        lk, z, sg = self.lk, self.z, self.lk["signers"]
        me = sg[self.i]
        delta, sigma = self.k * self.gamma % Q, self.k * self.w % Q                  # party_i.rs:591-618
        self.mb_gamma_pk = []
        for jj in range(self.S - 1):
            ind = self.ind(jj)
            jme = self.i if self.i < ind else self.i - 1                             # my slot among the P2P messages of `ind`
            for v in range(2):
                mb = m1[ind][jme][v]
                alpha = R.paillier_decrypt_textbook(lk["p"], lk["q"], mb["c"]) % Q   # mta/mod.rs:160-179
                bp, btp = mb["b_proof"], mb["beta_tag_proof"]
                ok = R.ec_mul(alpha, G) == R.ec_add(R.ec_mul(self.k, bp[0]), btp[0])
                ok = ok and R.dlog_verify(*bp) and R.dlog_verify(*btp)
                if not ok:
                    self.fail(201)
                if v == 1:                                                           # rounds.rs:281 (g_w_vec: party_i.rs:527-544)
                    if bp[0] != R.ec_mul(lagrange(sg, ind), lk["X"][sg[ind]]):
                        self.fail(202)
                else:
                    self.mb_gamma_pk.append(bp[0])
                if v == 0:
                    delta = (delta + alpha + self.beta[jj][0]) % Q
                else:
                    sigma = (sigma + alpha + self.beta[jj][1]) % Q
        self.delta_i, self.sigma_i = delta, sigma
        self.l = z["l"] % Q                                                          # phase3_compute_t_i :620-634
        self.t_proof = pedersen_prove(sigma, self.l, z["ped_s1"] % Q, z["ped_s2"] % Q)
        self.T = self.t_proof["com"]
        return dict(delta=delta, T=self.T, proof=self.t_proof)

    # Round3::proceed (rounds.rs:347-402)
    def round3(self, m2):
        if any(m["T"] != m["proof"]["com"] for m in m2):
            self.fail(303)
        tot = sum(m["delta"] for m in m2) % Q
        if tot == 0:
            self.fail(301)
            self.delta_inv = 0
        else:
            self.delta_inv = pow(tot, -1, Q)                                         # phase3_reconstruct_delta :635-640
        if not all(pedersen_verify(m["proof"]) for m in m2):
            self.fail(302)
        self.t_vec = [m["T"] for m in m2]
        return dict(blind=self.z["blind"], g_gamma=self.g_gamma)

    # Round4::proceed (rounds.rs:431-498)
    def round4(self, m3):
        lk, z, sg = self.lk, self.z, self.lk["signers"]
        bad = []
        for jj in range(self.S - 1):                                                 # phase4 party_i.rs:642-687
            ind = self.ind(jj)
            d = m3[ind]
            if self.mb_gamma_pk[jj] != d["g_gamma"] or hash_commitment(d["g_gamma"], d["blind"]) != self.bc_vec[ind]:
                bad.append(ind)
        if bad:
            self.fail(401, bad)
        acc = None
        for d in m3:
            acc = R.ec_add(acc, d["g_gamma"])
        self.R = R.ec_mul(self.delta_inv, acc)
        self.R_dash = R.ec_mul(self.k, self.R)
        me = sg[self.i]
        proofs = []
        for jj in range(self.S - 1):                                                 # phase5_proof_pdl :691-717
            st = sg[self.ind(jj)]
            proofs.append(R.pdl_prove(lk["N"][me], lk["Nt"][st], lk["h1"][st], lk["h2"][st], self.c, self.R_dash, self.R,
                                      self.k, z["r_a"], **z["pdl"][jj]))
        return dict(R_dash=self.R_dash, proofs=proofs)

    # Round5::proceed (rounds.rs:525-601)
    def round5(self, m4):
        lk, z, sg = self.lk, self.z, self.lk["signers"]
        for i in range(self.S):                                                      # phase5_verify_pdl :719-766
            good = True
            for jj in range(self.S - 1):
                st = sg[jj if jj < i else jj + 1]
                good &= R.pdl_verify(lk["N"][sg[i]], lk["Nt"][st], lk["h1"][st], lk["h2"][st], self.m_a_vec[i], m4[i]["R_dash"],
                                     self.R, m4[i]["proofs"][jj])
            if not good:
                self.fail(501, [i])
                break
        acc = None
        for m in m4:
            acc = R.ec_add(acc, m["R_dash"])
        if acc != G:                                                                 # phase5_check_R_dash_sum :768-776
            self.fail(502)
        self.S_i = R.ec_mul(self.sigma_i, self.R)                                    # phase6_compute_S_i.. :778-799
        self.heg = heg_prove(self.l, self.sigma_i, z["heg_s1"] % Q, z["heg_s2"] % Q, self.R, H2, G, self.T, self.S_i)
        return dict(S_i=self.S_i, proof=self.heg)

    # Round6::proceed (rounds.rs:612-636)
    def round6(self, m5):
        bad = [j for j, m in enumerate(m5) if not heg_verify(m["proof"], self.R, H2, G, self.t_vec[j], m["S_i"])]
        if bad:
            self.fail(601, bad)
        acc = None
        for m in m5:
            acc = R.ec_add(acc, m["S_i"])
        if acc != self.lk["y"]:
            self.fail(602)

    # Round7::new (rounds.rs:672-692) -> PartialSignature
    def round7(self, msg):
        self.m = msg
        self.r = self.R[0] % Q
        self.s_i = (msg % Q * self.k + self.r * self.sigma_i) % Q                    # phase7_local_sig :850-871
        return self.s_i

    # SignManual::complete -> output_signature (party_i.rs:873-910)
    def complete(self, m6):
        s = sum(m6) % Q
        recid = (self.R[1] % Q) & 1
        if s > Q - s:
            s, recid = Q - s, recid ^ 1
        if not R.ecdsa_verify(self.lk["y"], self.m % Q, self.r, s):
            self.fail(701)
        return self.r, s, recid


def simulate(parties, msg):
    """round_based::dev::Simulation: every party's outgoing message reaches every other party (sign.rs:667-763).
    Returns the messages of every round and the per-party signatures."""
    m0 = [p.round0() for p in parties]
    m1 = [p.round1(m0) for p in parties]
    m2 = [p.round2(m1) for p in parties]
    m3 = [p.round3(m2) for p in parties]
    m4 = [p.round4(m3) for p in parties]
    m5 = [p.round5(m4) for p in parties]
    for p in parties:
        p.round6(m5)
    m6 = [p.round7(msg) for p in parties]
    sigs = [p.complete(m6) for p in parties]
    return [m0, m1, m2, m3, m4, m5, m6], sigs


# ---- serialisation into the record layout of include/mpecdsa_hip.h ("GG20 round messages") ------------------------
def _w(v, n):
    return list(int(v).to_bytes(4 * n, "little"))


def _pt(p):
    return _w(0 if p is None else p[0] | (p[1] << 256), 16)


def _pad(b, words):
    assert len(b) <= 4 * words
    return b + [0] * (4 * words - len(b))


def pack(round_, m, S, n):
    """one sender's record of round `round_` (0..5, 7) as bytes"""
    if round_ == 0:
        out = []
        for pr in m["range_proofs"]:
            out += _pad(_w(pr["z"], 64) + _w(pr["e"], 8) + _w(pr["s"], 64) + _w(pr["s1"], 25) + _w(pr["s2"], 89), 256)
        out += _pad(_w(m["c"], 128) + _w(m["com"], 8), 256)
    elif round_ == 1:
        out = []
        for jj in range(S - 1):
            for v in range(2):
                mb = m[jj][v]
                (pk, Rp, zz), (tpk, tR, tz) = mb["b_proof"], mb["beta_tag_proof"]
                out += _w(mb["c"], 128) + _pt(pk) + _pt(Rp) + _w(zz, 8) + _pt(tpk) + _pt(tR) + _w(tz, 8)
    elif round_ == 2:
        p = m["proof"]
        out = _w(m["delta"], 8) + _pt(m["T"]) + _w(p["e"], 8) + _pt(p["a1"]) + _pt(p["a2"]) + _pt(p["com"]) + _w(p["z1"], 8) + _w(p["z2"], 8)
    elif round_ == 3:
        out = _w(m["blind"], 8) + _pt(m["g_gamma"])
    elif round_ == 4:
        out = []
        for pr in m["proofs"]:
            out += _w(pr["z"], 64) + _pt(pr["u1"]) + _w(pr["u2"], 128) + _w(pr["u3"], 64) + _w(pr["s1"], 25) + _w(pr["s2"], 64) + _w(pr["s3"], 89)
        out += _pad(_pt(m["R_dash"]), 450)
    elif round_ == 5:
        p = m["proof"]
        out = _pt(m["S_i"]) + _pt(p["T"]) + _pt(p["A3"]) + _w(p["z1"], 8) + _w(p["z2"], 8)
    elif round_ == 7:
        out = _w(m, 8)
    else:
        raise ValueError(round_)
    return bytes(out)


# ---- identifiable abort: protocols/multi_party_ecdsa/gg_2020/blame.rs, restated on Python ints ---------------------------
def ecddh_prove(x, s, g1, h1, g2, h2):
    a1, a2 = R.ec_mul(s, g1), R.ec_mul(s, g2)
    e = scalar_hash([g1, h1, g2, h2, a1, a2])
    return a1, a2, (s + e * x) % Q


def ecddh_verify(g1, h1, g2, h2, a1, a2, z):
    e = scalar_hash([g1, h1, g2, h2, a1, a2])
    return R.ec_mul(z, g1) == R.ec_add(a1, R.ec_mul(e, h1)) and R.ec_mul(z, g2) == R.ec_add(a2, R.ec_mul(e, h2))


def paillier_open(p, q, c):
    """Paillier::open: (m, r) with c = (1 + m n) r^n mod n^2"""
    n = p * q
    m = R.paillier_decrypt_textbook(p, q, c)
    t = c * (1 - m * n) % (n * n) % n
    return m, pow(t, pow(n, -1, (p - 1) * (q - 1)), n)


def blame5(N, k, k_rand, gamma, beta_tag, beta_rand, delta, g_gamma, c_a, c_b):
    """GlobalStatePhase5::phase5_blame (blame.rs:116-224).  N[i]: Paillier modulus of signer i; the lists are indexed by signer
    ordinal i and peer slot j.  Returns the sorted bad-actor list."""
    S = len(k)
    bad = [i for i in range(S) if g_gamma[i] != R.ec_mul(gamma[i], G)]
    ab = []
    for i in range(S):
        ca = R.paillier_encrypt(N[i], k[i] % Q, k_rand[i])
        if ca != c_a[i]:
            bad.append(i)
        row = []
        if not bad:
            for j in range(S - 1):
                ind = j if j < i else j + 1
                NN = N[i] * N[i]
                cb = pow(ca, gamma[ind] % Q, NN) * R.paillier_encrypt(N[i], beta_tag[i][j], beta_rand[i][j]) % NN
                if cb != c_b[i][j]:
                    bad.append(ind)
                beta = (-(beta_tag[i][j] % Q)) % Q
                row.append(((k[i] * gamma[ind] - beta) % Q, beta))
        ab.append(row)
    if not bad:
        for i in range(S):
            d = k[i] * gamma[i] + sum(a for a, _ in ab[i])
            for j in range(S - 1):
                ind1, ind2 = (j if j < i else j + 1), (i - 1 if j < i else i)
                d += ab[ind1][ind2][1]
            if d % Q != delta[i] % Q:
                bad.append(i)
    return sorted(set(bad))


def blame6(N, g_w, k, k_rand, miu, miu_rand, proofs, S_vec, c_a, c_b, Rp):
    """GlobalStatePhase6::phase6_blame (blame.rs:322-421)"""
    S = len(k)
    bad = []
    for i in range(S):
        for j in range(S - 1):
            if R.paillier_encrypt(N[i], miu[i][j], miu_rand[i][j]) != c_b[i][j]:
                bad.append(i)
    for i in range(S):
        if R.paillier_encrypt(N[i], k[i] % Q, k_rand[i]) != c_a[i]:
            bad.append(i)
    if not bad:
        g_ni = [[R.ec_add(R.ec_mul(k[i], g_w[j if j < i else j + 1]), R.ec_neg(R.ec_mul(miu[i][j], G))) for j in range(S - 1)] for i in range(S)]
        for i in range(S):
            gs = R.ec_mul(k[i], g_w[i])
            for x in miu[i]:
                gs = R.ec_add(gs, R.ec_mul(x, G))
            for j in range(S - 1):
                ind1, ind2 = (j if j < i else j + 1), (i - 1 if j < i else i)
                gs = R.ec_add(gs, g_ni[ind1][ind2])
            if not ecddh_verify(G, gs, Rp, S_vec[i], *proofs[i]):
                bad.append(i)
    return sorted(set(bad))


def blame7(s_vec, r, R_dash, m, Rp, S_vec):
    """GlobalStatePhase7::phase7_blame (blame.rs:434-454)"""
    return [i for i in range(len(s_vec)) if R.ec_mul(s_vec[i], Rp) != R.ec_add(R.ec_mul(m, R_dash[i]), R.ec_mul(r, S_vec[i]))]
