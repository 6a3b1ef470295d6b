"""Independent pure-Python restatement (Python ints + hashlib) of the same reference formulas the
C oracle (oracle/mpe_oracle.c) implements.  Its only job is to pin the C oracle in the CPU tests:
two implementations in different languages, written from the reference text, must agree.
Small cases only.  File:line citations are to /root/reference (ZenGo-X/multi-party-ecdsa v0.8.1)."""
import hashlib

# ---- secp256k1 ---------------------------------------------------------------------------------
P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
Q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
# curv `Point::base_point2()`
H2 = (0x08d13221e3a7326a34dd45214ba80116dd142e4b5ff3ce66a8dc7bfa0378b795,
      0x5d41ac1477614b5c0848d50dbd565ea2807bcba1df0df07a8217e9f7f7c2be88)
INF = None


def ec_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def ec_mul(k, pt):
    k %= Q
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, pt)
        pt = ec_add(pt, pt)
        k >>= 1
    return acc


def ec_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def pt_bytes(pt, compressed):
    if compressed:
        return bytes([2 + (pt[1] & 1)]) + pt[0].to_bytes(32, "big")
    return b"\x04" + pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


# ---- the byte-level conventions of the un-vendored crates, as a profile (mirror of mpe_encoding / orc_encoding) --------
class Encoding:
    """Same fields as `mpe_encoding` (include/mpecdsa_hip.h documents each).  Defaults = what this repository believes
    curv-kzen 0.9 / zk-paillier 0.4.3 do; every alternative is exercised by tests/test_encodings_*.py."""
    FIELDS = ("chain_point", "zero_bytes", "ck_mask_order", "ck_salt", "ord_dlog", "ord_pedersen", "ord_heg", "ord_ecddh", "ord_cdlog")
    SIZES = dict(ord_dlog=3, ord_pedersen=5, ord_heg=7, ord_ecddh=6, ord_cdlog=4)

    def __init__(self, chain_point=0, zero_bytes=0, ck_mask_order=0, ck_salt=0x4B5A656E, ord_dlog=(0, 1, 2),
                 ord_pedersen=(0, 1, 2, 3, 4), ord_heg=(0, 1, 2, 3, 4, 5, 6), ord_ecddh=(0, 1, 2, 3, 4, 5), ord_cdlog=(0, 1, 2, 3)):
        self.chain_point, self.zero_bytes, self.ck_mask_order, self.ck_salt = chain_point, zero_bytes, ck_mask_order, ck_salt
        self.ord_dlog, self.ord_pedersen, self.ord_heg = tuple(ord_dlog), tuple(ord_pedersen), tuple(ord_heg)
        self.ord_ecddh, self.ord_cdlog = tuple(ord_ecddh), tuple(ord_cdlog)
        for f, n in self.SIZES.items():
            assert sorted(getattr(self, f)) == list(range(n)), f

    def as_dict(self):
        return {f: (list(getattr(self, f)) if f.startswith("ord_") else getattr(self, f)) for f in self.FIELDS}

    def replace(self, **kw):
        d = self.as_dict()
        d.update(kw)
        return Encoding(**d)

    def __eq__(self, o):
        return isinstance(o, Encoding) and self.as_dict() == o.as_dict()

    def __repr__(self):
        dflt = Encoding().as_dict()
        diff = {k: v for k, v in self.as_dict().items() if v != dflt[k]}
        return "Encoding(" + ", ".join(f"{k}={v}" for k, v in diff.items()) + ")" if diff else "Encoding(default)"


ENC = Encoding()          # the profile in force for this module (tests switch it with `use_encoding`)


class use_encoding:
    """with use_encoding(enc): ...   (restores the previous profile)"""

    def __init__(self, enc):
        self.enc = enc

    def __enter__(self):
        global ENC
        self.prev, ENC = ENC, self.enc
        return self.enc

    def __exit__(self, *a):
        global ENC
        ENC = self.prev


def chain_point_bytes(pt):
    """DigestExt::chain_point: Point::to_bytes(false) (65 B) — or to_bytes(true) under ENC.chain_point = 1"""
    return pt_bytes(pt, bool(ENC.chain_point))


def hash_points_scalar(canon, ord_field):
    """Sha256::new().chain_points(..).result_scalar() over the proof's canonical point list in the profile's order"""
    h = hashlib.sha256()
    for k in getattr(ENC, ord_field):
        h.update(chain_point_bytes(canon[k]))
    return int.from_bytes(h.digest(), "big") % Q


# ---- curv BigInt / DigestExt ------------------------------------------------------------------------
def to_bytes(x):
    """BigInt::to_bytes: big-endian magnitude, minimal length (0 -> b'\\x00' under rust-gmp; b'' under ENC.zero_bytes = 1)."""
    if x == 0 and ENC.zero_bytes:
        return b""
    return x.to_bytes(max(1, (x.bit_length() + 7) // 8), "big")


def hash_bigints(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(to_bytes(v))
    return int.from_bytes(h.digest(), "big")


def pt_as_bigint(pt):
    return int.from_bytes(pt_bytes(pt, True), "big")


# ---- kzen-paillier -----------------------------------------------------------------------------------
def paillier_encrypt(n, m, r):
    nn = n * n
    return (1 + m * n) % nn * pow(r, n, nn) % nn


def paillier_decrypt_textbook(p, q, c):
    """Textbook (non-CRT) decryption: m = L(c^lambda mod n^2) * mu mod n — a different algorithm
    from the CRT one the reference (and the oracle) uses; same unique answer."""
    n = p * q
    nn = n * n
    lam = (p - 1) * (q - 1)
    u = pow(c, lam, nn)
    return (u - 1) // n * pow(lam, -1, n) % n


# ---- commitment_unknown_order (zk_pdl_with_slack/mod.rs:182-199) -----------------------------------------
def commit(h1, h2, M, x, r):
    if r < 0:
        return pow(h1, x, M) * pow(pow(h2, -1, M), -r, M) % M
    return pow(h1, x, M) * pow(h2, r, M) % M


# ---- AliceProof (mta/range_proofs.rs:39-193) ---------------------------------------------------------
def alice_generate(N, Nt, h1, h2, a, c, r, alpha, beta, gamma, rho):
    NN = N * N
    z = pow(h1, a, Nt) * pow(h2, rho, Nt) % Nt
    u = (alpha * N + 1) * pow(beta, N, NN) % NN
    w = pow(h1, alpha, Nt) * pow(h2, gamma, Nt) % Nt
    e = hash_bigints([N, N + 1, c, z, u, w])
    s = pow(r, e, N) * beta % N
    return dict(z=z, e=e, s=s, s1=e * a + alpha, s2=e * rho + gamma)


def alice_verify(N, Nt, h1, h2, c, pr):
    NN = N * N
    if pr["s1"] > Q ** 3:
        return False
    try:
        z_e_inv = pow(pow(pr["z"], pr["e"], Nt), -1, Nt)
        c_e_inv = pow(pow(c, pr["e"], NN), -1, NN)
    except ValueError:
        return False
    w = pow(h1, pr["s1"], Nt) * pow(h2, pr["s2"], Nt) * z_e_inv % Nt
    u = (pr["s1"] * N + 1) % NN * pow(pr["s"], N, NN) * c_e_inv % NN
    return hash_bigints([N, N + 1, c, pr["z"], u, w]) == pr["e"]


# ---- PDLwSlackProof (zk_pdl_with_slack/mod.rs:68-179) ---------------------------------------------------
def pdl_prove(N, Nt, h1, h2, c, Qp, Gp, x, r, alpha, beta, rho, gamma):
    NN = N * N
    z = commit(h1, h2, Nt, x, rho)
    u1 = ec_mul(alpha, Gp)
    u2 = commit(N + 1, beta, NN, alpha, N)
    u3 = commit(h1, h2, Nt, alpha, gamma)
    e = hash_bigints([pt_as_bigint(Gp), pt_as_bigint(Qp), c, z, pt_as_bigint(u1), u2, u3])
    return dict(z=z, u1=u1, u2=u2, u3=u3, s1=e * x + alpha, s2=commit(r, beta, N, e, 1), s3=e * rho + gamma)


def pdl_verify(N, Nt, h1, h2, c, Qp, Gp, pr):
    NN = N * N
    e = hash_bigints([pt_as_bigint(Gp), pt_as_bigint(Qp), c, pr["z"], pt_as_bigint(pr["u1"]), pr["u2"], pr["u3"]])
    u1 = ec_add(ec_mul(pr["s1"], Gp), ec_mul(Q - e, Qp))
    u2 = commit(commit(N + 1, pr["s2"], NN, pr["s1"], N), c, NN, 1, -e)
    u3 = commit(commit(h1, h2, Nt, pr["s1"], pr["s3"]), pr["z"], Nt, 1, -e)
    return u1 == pr["u1"] and u2 == pr["u2"] and u3 == pr["u3"]


# ---- BobProof (mta/range_proofs.rs:218-534) -------------------------------------------------------------
def bob_generate(N, Nt, h1, h2, a_enc, mta_enc, b, beta_prim, r, alpha, beta, gamma, rho, rho_prim, sigma, tau, check):
    NN = N * N
    z = pow(h1, b, Nt) * pow(h2, rho, Nt) % Nt
    z_prim = pow(h1, alpha, Nt) * pow(h2, rho_prim, Nt) % Nt
    t = pow(h1, beta_prim, Nt) * pow(h2, sigma, Nt) % Nt
    w = pow(h1, gamma, Nt) * pow(h2, tau, Nt) % Nt
    v = pow(a_enc, alpha, NN) * (gamma * N + 1) * pow(beta, N, NN) % NN
    vals = [N, N + 1, a_enc, mta_enc, z, z_prim, t, v, w]
    u = None
    if check:
        X, u = ec_mul(b, G), ec_mul(alpha, G)
        vals += [X[0], X[1], u[0], u[1]]
    e = hash_bigints(vals)
    return dict(t=t, z=z, e=e, s=pow(r, e, N) * beta % N, s1=e * b + alpha, s2=e * rho + rho_prim,
                t1=e * beta_prim + gamma, t2=e * sigma + tau), u


def bob_verify(N, Nt, h1, h2, a_enc, mta_enc, pr, X=None, u=None):
    NN = N * N
    if pr["s1"] > Q ** 3:
        return False
    try:
        z_e_inv = pow(pow(pr["z"], pr["e"], Nt), -1, Nt)
        mta_e_inv = pow(pow(mta_enc, pr["e"], NN), -1, NN)
        t_e_inv = pow(pow(pr["t"], pr["e"], Nt), -1, Nt)
    except ValueError:
        return False
    z_prim = pow(h1, pr["s1"], Nt) * pow(h2, pr["s2"], Nt) * z_e_inv % Nt
    v = pow(a_enc, pr["s1"], NN) * pow(pr["s"], N, NN) * (pr["t1"] * N + 1) * mta_e_inv % NN
    w = pow(h1, pr["t1"], Nt) * pow(h2, pr["t2"], Nt) * t_e_inv % Nt
    vals = [N, N + 1, a_enc, mta_enc, pr["z"], z_prim, pr["t"], v, w]
    if X is not None:
        vals += [X[0], X[1], u[0], u[1]]
    if hash_bigints(vals) != pr["e"]:
        return False
    if X is not None:
        return ec_mul(pr["s1"], G) == ec_add(ec_mul(pr["e"], X), u)
    return True


# ---- curv DLogProof ------------------------------------------------------------------
def dlog_challenge(R, pk):
    return hash_points_scalar([R, G, pk], "ord_dlog")


def dlog_prove(sk, nonce):
    pk, R = ec_mul(sk, G), ec_mul(nonce, G)
    return pk, R, (nonce - dlog_challenge(R, pk) * sk) % Q


def dlog_verify(pk, R, z):
    return ec_add(ec_mul(z, G), ec_mul(dlog_challenge(R, pk), pk)) == R


# ---- ECDSA verification independent of everything above but the curve (gg_2020/test.rs:711-748) -------
def ecdsa_verify(pub, msg_int, r, s):
    if not (0 < r < Q and 0 < s < Q):
        return False
    w = pow(s, -1, Q)
    pt = ec_add(ec_mul(msg_int * w % Q, G), ec_mul(r * w % Q, pub))
    return pt is not None and pt[0] % Q == r


# ---- Lindell'17 signing (two_party_ecdsa/lindell_2017/party_two.rs:390-423, party_one.rs:519-565) ------------
def lindell_partial_sig(n, c_key, x2, k2, R1, msg, rho, r):
    """PartialSig::compute with the sampled values (rho, the encryption randomness r) as inputs -> c3"""
    nn = n * n
    rx = ec_mul(k2, R1)[0] % Q
    k2_inv = pow(k2, -1, Q)
    partial = rho * Q + k2_inv * msg % Q
    c1 = paillier_encrypt(n, partial, r)
    v = k2_inv * (rx * x2 % Q) % Q
    return pow(c_key, v, nn) * c1 % nn


def lindell_sign(p, q, c3, k1, R2):
    """Signature::compute_with_recid -> (r, s, recid); the plaintext through the textbook (non-CRT) decryption"""
    Rp = ec_mul(k1, R2)
    rx, ry = Rp[0] % Q, Rp[1] % Q
    s2 = paillier_decrypt_textbook(p, q, c3) % Q * pow(k1, -1, Q) % Q
    recid = ry & 1
    if s2 > Q - s2:
        s2, recid = Q - s2, recid ^ 1
    return rx, s2, recid


# ---- keygen verification math (gg_2020/party_i.rs:260-438); zk-paillier 0.4.3 proofs as recalled -----
def zkp_digest(vals):
    """zk-paillier compute_digest = hash_bigints"""
    return hash_bigints(vals)


def cdlog_digest(x, g, N, ni):
    canon = [x, g, N, ni]
    return zkp_digest([canon[k] for k in ENC.ord_cdlog])


def composite_dlog_prove(N, g, ni, secret, r):
    x = pow(g, r, N)
    return x, r + cdlog_digest(x, g, N, ni) * secret


def composite_dlog_verify(N, g, ni, x, y):
    import math
    if N < (1 << 128) or N % 2 == 0 or math.gcd(g, N) != 1 or math.gcd(ni, N) != 1:
        return False
    return pow(g, y, N) * pow(ni, cdlog_digest(x, g, N, ni), N) % N == x


def correct_key_rho(N, i):
    seed = zkp_digest([N, ENC.ck_salt, i])                    # SALT_STRING = b"KZen" as a BigInt by default
    msklen = N.bit_length() // 256 + 1
    return sum(zkp_digest([seed, j]) << (256 * ((msklen - 1 - j) if ENC.ck_mask_order else j)) for j in range(msklen)) % N


def correct_key_prove(p, q):
    N = p * q
    d = pow(N, -1, (p - 1) * (q - 1))
    return [pow(correct_key_rho(N, i), d, N) for i in range(11)]


def correct_key_verify(N, sigma):
    small = [p for p in range(2, 6370) if all(p % d for d in range(2, int(p ** 0.5) + 1))]
    if N <= 1 or any(N % p == 0 for p in small):
        return False
    return all(pow(sigma[i], N, N) == correct_key_rho(N, i) for i in range(11))


def vss_point(commits, index):
    acc = None
    for c in reversed(commits):
        acc = ec_add(ec_mul(index, acc) if acc is not None else None, c)
    return acc
