//! Reference vectors for the MI355X engine, produced by the REAL crates (multi-party-ecdsa 0.8.1, curv-kzen 0.9,
//! kzen-paillier 0.4.2).  Not compiled in the GPU repo (there is no Rust toolchain in its image): a maintainer drops this
//! file into the reference checkout as `tests/dump_vectors.rs` and runs
//!
//!     cargo test --release --test dump_vectors -- --nocapture > ref_vectors.json
//!
//! then copies `ref_vectors.json` to `tests/golden/ref_vectors.json` of the GPU repo, where `tests/test_ref_vectors_cpu.py`
//! (and the `-m gpu` counterpart) consume it: every proof below must be ACCEPTED by the oracle's and the engine's verifiers
//! and every deterministic value (ciphertexts, commitments, challenges) must be reproduced byte for byte.  Acceptance of a
//! Fiat-Shamir proof requires the verifier to rebuild the exact transcript, so this pins the encodings the GPU repo could
//! only recall: DigestExt::chain_point, result_scalar, BigInt::to_bytes, HashCommitment, the serde forms of BigInt / Point /
//! Scalar, the field order of the curv sigma proofs.
//!
//! Only public API is used; values with private fields are emitted through their own `Serialize` impls.
#![allow(non_snake_case)]

use curv::arithmetic::traits::*;
use curv::cryptographic_primitives::commitments::hash_commitment::HashCommitment;
use curv::cryptographic_primitives::commitments::traits::Commitment;
use curv::cryptographic_primitives::proofs::sigma_correct_homomorphic_elgamal_enc::{
    HomoELGamalProof, HomoElGamalStatement, HomoElGamalWitness,
};
use curv::cryptographic_primitives::proofs::sigma_dlog::DLogProof;
use curv::cryptographic_primitives::proofs::sigma_ec_ddh::{ECDDHProof, ECDDHStatement, ECDDHWitness};
use curv::cryptographic_primitives::proofs::sigma_valid_pedersen::PedersenProof;
use curv::elliptic::curves::{secp256_k1::Secp256k1, Point, Scalar};
use curv::BigInt;
use multi_party_ecdsa::utilities::mta::range_proofs::AliceProof;
use multi_party_ecdsa::utilities::mta::{MessageA, MessageB};
use multi_party_ecdsa::utilities::zk_pdl_with_slack::{PDLwSlackProof, PDLwSlackStatement, PDLwSlackWitness};
use paillier::{
    Decrypt, DecryptionKey, EncryptWithChosenRandomness, EncryptionKey, KeyGeneration, Paillier, Randomness, RawCiphertext,
    RawPlaintext,
};
use serde_json::{json, Value};
use sha2::Sha256;
use paillier::traits::Open;
use zk_paillier::zkproofs::{CompositeDLogProof, DLogStatement, NiCorrectKeyProof, SALT_STRING};

fn hex(x: &BigInt) -> String {
    x.to_hex()
}
fn pt(p: &Point<Secp256k1>) -> Value {
    // explicit coordinates AND the crate's own serde form
    json!({"x": hex(&p.x_coord().unwrap()), "y": hex(&p.y_coord().unwrap()), "bytes_compressed": hex::encode(&*p.to_bytes(true)),
           "serde": serde_json::to_value(p).unwrap()})
}
fn sc(s: &Scalar<Secp256k1>) -> Value {
    json!({"hex": hex(&s.to_bigint()), "serde": serde_json::to_value(s).unwrap()})
}

// same construction as src/utilities/mta/range_proofs.rs:592-613 (generate_init)
fn statement_and_keys() -> (DLogStatement, EncryptionKey, DecryptionKey, BigInt) {
    let (ek_tilde, dk_tilde) = Paillier::keypair().keys();
    let one = BigInt::one();
    let phi = (&dk_tilde.p - &one) * (&dk_tilde.q - &one);
    let h1 = BigInt::sample_below(&ek_tilde.n);
    let (xhi, _) = loop {
        let xhi_ = BigInt::sample_below(&phi);
        match BigInt::mod_inv(&xhi_, &phi) {
            Some(inv) => break (xhi_, inv),
            None => continue,
        }
    };
    let h2 = BigInt::mod_pow(&h1, &xhi, &ek_tilde.n);
    let (ek, dk) = Paillier::keypair().keys();
    // the secret of the statement is phi - xhi, as generate_h1_h2_N_tilde stores it (party_i.rs:152-153)
    (DLogStatement { g: h1, ni: h2, N: ek_tilde.n }, ek, dk, &phi - &xhi)
}

// The sampling side (the device-side sampler restates curv's `Samplable for BigInt` and `Scalar::random`, mpe_sample.h).  curv
// draws from OsRng inside `sample` — no public entry point takes a caller's RngCore — so a seeded replay is impossible; what CAN
// be pinned: (1) the byte -> integer arithmetic of `sample(bits)` on KNOWN byte strings, computed here with the crate's own
// `from_bytes` and shift exactly as its source does (`BigInt::from_bytes(&buf) >> (bytes * 8 - bits)`); (2) the RANGES the crate's
// real draws fall in: sample(bits) < 2^bits and reaches bit `bits` - 1, sample_below(u) < u, sample_range(1, N - 1) in [1, N - 2],
// Scalar::random() in [1, q).
fn sampler_record(n: &BigInt) -> Value {
    let mut known = Vec::<Value>::new();
    for (len, bits) in [(32usize, 256usize), (1, 7), (1, 1), (2, 9), (5, 33), (256, 2047), (256, 2048), (257, 2050), (352, 2816)] {
        let buf: Vec<u8> = (0..len).map(|i| (i * 167 + 13 + len) as u8 | if i == 0 { 0x80 } else { 0 }).collect();
        let v = BigInt::from_bytes(&buf) >> (len * 8 - bits);
        known.push(json!({"bytes": hex::encode(&buf), "bits": bits, "value": hex(&v)}));
    }
    let u = (BigInt::one() << 300) + BigInt::from(12345u32);
    let nm1 = n - &BigInt::one();
    json!({
        "known_bytes": known,
        "sample_bits": {"bits": 7, "draws": (0..64).map(|_| hex(&BigInt::sample(7))).collect::<Vec<_>>()},
        "sample_below": {"upper": hex(&u), "draws": (0..64).map(|_| hex(&BigInt::sample_below(&u))).collect::<Vec<_>>()},
        "sample_range": {"lo": "1", "hi": hex(&nm1), "draws": (0..16).map(|_| hex(&BigInt::sample_range(&BigInt::one(), &nm1))).collect::<Vec<_>>()},
        "scalar_random": (0..16).map(|_| hex(&Scalar::<Secp256k1>::random().to_bigint())).collect::<Vec<_>>(),
    })
}

#[test]
fn dump_vectors() {
    let mut out = Vec::<Value>::new();
    for _case in 0..4 {
        let (st, ek, dk, xhi) = statement_and_keys();
        let keys = json!({"N": hex(&ek.n), "p": hex(&dk.p), "q": hex(&dk.q), "Nt": hex(&st.N), "h1": hex(&st.g), "h2": hex(&st.ni)});

        // --- Paillier: encrypt_with_chosen_randomness / decrypt (src/utilities/mta/mod.rs:68-75,165)
        let a = Scalar::<Secp256k1>::random();
        let r = BigInt::sample_below(&ek.n);
        let c = Paillier::encrypt_with_chosen_randomness(&ek, RawPlaintext::from(a.to_bigint()), &Randomness::from(&r)).0.into_owned();
        let back: RawPlaintext = Paillier::decrypt(&dk, RawCiphertext::from(c.clone()));
        assert_eq!(back.0.into_owned(), a.to_bigint());

        // --- AliceProof (src/utilities/mta/range_proofs.rs:160-193): generated by the crate, verified by the engine
        let alice = AliceProof::generate(&a.to_bigint(), &c, &ek, &st, &r);
        assert!(alice.verify(&c, &ek, &st));

        // --- MtA (src/utilities/mta/mod.rs:52-179): MessageA / MessageB as the crate serialises them
        let b = Scalar::<Secp256k1>::random();
        let (m_a, m_a_rand) = MessageA::a(&a, &ek, &[st.clone()]);
        let (m_b, beta, beta_rand, beta_tag) = MessageB::b(&b, &ek, m_a.clone(), &[st.clone()]).unwrap();
        let (alpha, alice_share) = m_b.verify_proofs_get_alpha(&dk, &a).unwrap();
        assert_eq!(&alpha + &beta, &a * &b);

        // --- PDL with slack (src/utilities/zk_pdl_with_slack/mod.rs:68-179), statement as GG20 builds it: G = some R, Q = x G
        let Rp = Point::generator() * Scalar::<Secp256k1>::random();
        let Q = &Rp * &a;
        let pdl_st = PDLwSlackStatement { ciphertext: c.clone(), ek: ek.clone(), Q: Q.clone(), G: Rp.clone(), h1: st.g.clone(), h2: st.ni.clone(),
                                          N_tilde: st.N.clone() };
        let pdl = PDLwSlackProof::prove(&PDLwSlackWitness { x: a.clone(), r: r.clone() }, &pdl_st);
        assert!(pdl.verify(&pdl_st).is_ok());

        // --- curv sigma proofs and the hash commitment, with the values GG20 feeds them (party_i.rs:573-589,620-634,778-799)
        let dlog = DLogProof::<Secp256k1, Sha256>::prove(&a);
        let l = Scalar::<Secp256k1>::random();
        let ped = PedersenProof::<Secp256k1, Sha256>::prove(&a, &l);
        let T = ped.com.clone();
        let S = &Rp * &a;
        let heg_st = HomoElGamalStatement { G: Rp.clone(), H: Point::<Secp256k1>::base_point2().clone(), Y: Point::generator().to_point(),
                                            D: T.clone(), E: S.clone() };
        let heg = HomoELGamalProof::<Secp256k1, Sha256>::prove(&HomoElGamalWitness { x: l.clone(), r: a.clone() }, &heg_st);
        assert!(heg.verify(&heg_st).is_ok());
        let ddh_st = ECDDHStatement { g1: Point::generator().to_point(), g2: Rp.clone(), h1: Point::generator() * &a, h2: S.clone() };
        let ddh = ECDDHProof::<Secp256k1, Sha256>::prove(&ECDDHWitness { x: a.clone() }, &ddh_st);
        assert!(ddh.verify(&ddh_st).is_ok());
        let blind = BigInt::sample(256);
        let g_gamma = Point::generator() * &b;
        let com = HashCommitment::<Sha256>::create_commitment_with_user_defined_randomness(&BigInt::from_bytes(g_gamma.to_bytes(true).as_ref()), &blind);

        // --- keygen proofs of zk-paillier (party_i.rs:219-258,283-301) and kzen-paillier's Open (blame.rs:252-256)
        let ck = NiCorrectKeyProof::proof(&dk, None);
        assert!(ck.verify(&ek, SALT_STRING).is_ok());
        // DLogStatement { N, g: h1, ni: h2 } with the secret phi - xhi, exactly as Keys::phase1_broadcast_... builds it (party_i.rs:225-237)
        let cd = CompositeDLogProof::prove(&st, &xhi);
        let cd_ok = cd.verify(&st).is_ok();
        let (open_m, open_r) = Paillier::open(&dk, RawCiphertext::from(c.clone()));

        out.push(json!({
            "keys": keys,
            "paillier": {"m": hex(&a.to_bigint()), "r": hex(&r), "c": hex(&c)},
            "alice_proof": {"a": hex(&a.to_bigint()), "cipher": hex(&c), "proof": serde_json::to_value(&alice).unwrap()},
            "mta": {"a": sc(&a), "b": sc(&b), "m_a": serde_json::to_value(&m_a).unwrap(), "m_a_randomness": hex(&m_a_rand),
                    "m_b": serde_json::to_value(&m_b).unwrap(), "beta": sc(&beta), "beta_randomness": hex(&beta_rand), "beta_tag": hex(&beta_tag),
                    "alpha": sc(&alpha), "alice_share": hex(&alice_share)},
            "pdl": {"x": sc(&a), "r": hex(&r), "c": hex(&c), "Q": pt(&Q), "G": pt(&Rp), "proof": serde_json::to_value(&pdl).unwrap()},
            "dlog": {"sk": sc(&a), "proof": serde_json::to_value(&dlog).unwrap(), "pk": pt(&dlog.pk)},
            "pedersen": {"m": sc(&a), "r": sc(&l), "proof": serde_json::to_value(&ped).unwrap(), "com": pt(&T)},
            "heg": {"x": sc(&l), "r": sc(&a), "G": pt(&Rp), "D": pt(&T), "E": pt(&S), "proof": serde_json::to_value(&heg).unwrap()},
            "ecddh": {"x": sc(&a), "g2": pt(&Rp), "h1": pt(&(Point::generator() * &a)), "h2": pt(&S), "proof": serde_json::to_value(&ddh).unwrap()},
            "correct_key": {"proof": serde_json::to_value(&ck).unwrap()},
            "composite_dlog": {"secret": hex(&xhi), "proof": serde_json::to_value(&cd).unwrap(), "verifies": cd_ok},
            "open": {"c": hex(&c), "m": hex(&open_m.0.into_owned()), "r": hex(&open_r.0)},
            "hash_commitment": {"point": pt(&g_gamma), "blind": hex(&blind), "com": hex(&com)},
            "base_point2": pt(&Point::<Secp256k1>::base_point2().clone()),
            "sampler": sampler_record(&ek.n),
        }));
    }
    println!("{}", serde_json::to_string(&json!({"schema": 1, "crate": "multi-party-ecdsa 0.8.1", "cases": out})).unwrap());
}
