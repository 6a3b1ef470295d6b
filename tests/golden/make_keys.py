"""Mints the deterministic key-material fixture tests/golden/keys16.json.

The reference generates this material at keygen (Paillier::keypair and generate_h1_h2_N_tilde,
src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:137-177) from OS randomness and ships no
fixture; keygen is out of the hot path, so the tests use a fixed set minted
here with GMP's mpz_nextprime from a SHA-256 counter stream.  Run:  python tests/golden/make_keys.py
"""
import hashlib, json, os, sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402

orc.lib.orc_nextprime.restype = None


def stream(tag, nbytes):
    out, ctr = b"", 0
    while len(out) < nbytes:
        out += hashlib.sha256(b"mpecdsa-fixture|" + tag.encode() + b"|" + ctr.to_bytes(4, "big")).digest()
        ctr += 1
    return out[:nbytes]


def prime1024(tag):
    x = int.from_bytes(stream(tag, 128), "big") | (3 << 1022) | 1     # top two bits: products are exactly 2048 bit
    w = np.frombuffer(x.to_bytes(128, "little"), dtype="<u4").copy()
    out = np.zeros(32, dtype=np.uint32)
    orc.lib.orc_nextprime(32, orc._p(w), orc._p(out))
    return int.from_bytes(out.tobytes(), "little")


def main(nkeys=16):
    keys = []
    for i in range(nkeys):
        p, q = prime1024(f"paillier-p-{i}"), prime1024(f"paillier-q-{i}")
        pt, qt = prime1024(f"ntilde-p-{i}"), prime1024(f"ntilde-q-{i}")
        nt = pt * qt
        phi = (pt - 1) * (qt - 1)
        h1 = int.from_bytes(stream(f"h1-{i}", 256), "big") % nt
        ctr = 0
        while True:                                     # xhi invertible mod phi (party_i.rs:144-150)
            xhi = int.from_bytes(stream(f"xhi-{i}-{ctr}", 256), "big") % phi
            try:
                pow(xhi, -1, phi)
                break
            except ValueError:
                ctr += 1
        h2 = pow(h1, xhi, nt)
        keys.append({"p": hex(p), "q": hex(q), "n_tilde": hex(nt), "h1": hex(h1), "h2": hex(h2),
                     "nt_p": hex(pt), "nt_q": hex(qt), "xhi": hex(xhi)})
    with open(os.path.join(HERE, "keys16.json"), "w") as f:
        json.dump({"note": "deterministic test key material; see make_keys.py", "keys": keys}, f, indent=0)
    print("wrote", len(keys), "keys")


if __name__ == "__main__":
    main()
