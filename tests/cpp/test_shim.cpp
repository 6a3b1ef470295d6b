// The reference's own unit tests of the hot path, re-stated over include/mpecdsa.hpp (the C++ host layer above the C-ABI) and
// run on the GPU, batched; every value the GPU returns is also compared bit for bit with the CPU oracle (oracle/mpe_oracle.h).
//
//   alice_zkp                          src/utilities/mta/range_proofs.rs:614-633
//   bob_zkp                            src/utilities/mta/range_proofs.rs:636-709 (MtA with BobProof, MtAwc with BobProofExt)
//   test_mta                           src/utilities/mta/test.rs:6-19            (alpha + beta == a * b)
//   test_zk_pdl_with_slack             src/utilities/zk_pdl_with_slack/test.rs:12-68
//   test_zk_pdl_with_slack_soundness   src/utilities/zk_pdl_with_slack/test.rs:70-129   (#[should_panic]: Enc(x + 1) must be refused)
//   simulate_signing_t1_n2_s2 / _t1_n3_s2 / _t2_n3_s3   gg_2020/state_machine/sign.rs:667-762 (one OfflineStage per party, each with its
//                                      own secrets only, a Simulation relaying their messages, then SignManual for every party)
//   the constructor / message-store errors of sign.rs:77-101,246-330
//   test_two_party_sign                src/protocols/two_party_ecdsa/lindell_2017/test.rs:85-137
//
// Inputs (keys, scalars, every value the reference samples from OsRng) come from a fixture file written by
// tests/test_cpp_shim_gpu.py — the same seeded fixtures the Python tests use.  Test infrastructure: links the oracle and libgmp;
// the product library links neither.  Exit status 0 = all passed; prints one line per test.
#include <gmp.h>

#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "mpe_oracle.h"
#include "mpecdsa.hpp"

using namespace mpecdsa;
using paillier::DecryptionKeys;
using paillier::EncryptionKeys;
using paillier::Paillier;
using zk_paillier::DLogStatements;
namespace sm = gg_2020::state_machine::sign;

// oracle/ossl_check.c (libmpe_ossl.so): OpenSSL's ECDSA_do_verify over the interface words — the independent `verify`
extern "C" int ossl_ecdsa_verify(int batch, const uint32_t* pub, int pub_stride, const uint32_t* msg, const uint32_t* r, const uint32_t* s, uint8_t* ok);

static std::map<std::string, Batch> load_fixture(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
  std::map<std::string, Batch> m;
  for (;;) {
    uint32_t nl = 0;
    if (!f.read((char*)&nl, 4)) break;
    std::string name(nl, ' ');
    f.read(&name[0], nl);
    uint32_t words = 0, rows = 0;
    f.read((char*)&words, 4);
    f.read((char*)&rows, 4);
    Batch b(rows, (int)words);
    f.read((char*)b.w.data(), (std::streamsize)b.w.size() * 4);
    m[name] = std::move(b);
  }
  return m;
}
static Index as_index(const Batch& b) { return Index(b.w.begin(), b.w.end()); }

#define REQUIRE(cond)                                                                 \
  do {                                                                                \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return false; } \
  } while (0)

static void to_mpz(mpz_t z, const uint32_t* w, int n) { mpz_import(z, (size_t)n, -1, 4, 0, 0, w); }
static bool all_ones(const Flags& f) { for (auto x : f) if (x != 1) return false; return !f.empty(); }

struct Fixture {
  std::map<std::string, Batch> a;
  const Batch& operator[](const char* k) const {
    auto it = a.find(k);
    if (it == a.end()) { std::fprintf(stderr, "fixture lacks %s\n", k); std::exit(2); }
    return it->second;
  }
};

// range_proofs.rs:614-633
static bool alice_zkp(Context& ctx, const Fixture& F, const EncryptionKeys& ek, const DLogStatements& stm) {
  const Batch &a = F["a"], &r = F["r_a"], &N = F["N"];
  const Index ki = as_index(F["key_idx"]), si = as_index(F["st_idx"]);
  const int B = (int)a.size();
  Batch a64(B, W_N);                                               // RawPlaintext::from(a): the scalar as a plaintext
  for (int i = 0; i < B; ++i) std::memcpy(a64.row(i), a.row(i), 32);
  const Batch cipher = Paillier::encrypt_with_chosen_randomness(ctx, ek, ki, a64, r);
  Batch want_c(B, W_NN);
  orc_paillier_encrypt(B, (int)N.size(), N.w.data(), ki.data(), a64.w.data(), r.w.data(), want_c.w.data());
  REQUIRE(cipher == want_c);
  const mta::range_proofs::AliceNonces nn{F["one_alpha"], F["one_beta"], F["one_gamma"], F["one_rho"]};
  const auto proof = mta::range_proofs::AliceProof::generate(ctx, ek, stm, ki, si, a, cipher, r, nn);
  // the oracle's proof from the same inputs: byte-identical
  Batch z(B, W_N), e(B, W_SCALAR), s(B, W_N), s1(B, W_S1), s2(B, W_S2);
  orc_alice_generate(B, (int)N.size(), N.w.data(), (int)F["Nt"].size(), F["Nt"].w.data(), F["h1"].w.data(), F["h2"].w.data(), ki.data(), si.data(),
                     a.w.data(), cipher.w.data(), r.w.data(), nn.alpha.w.data(), nn.beta.w.data(), nn.gamma.w.data(), nn.rho.w.data(), z.w.data(),
                     e.w.data(), s.w.data(), s1.w.data(), s2.w.data());
  REQUIRE(proof.z == z && proof.e == e && proof.s == s && proof.s1 == s1 && proof.s2 == s2);
  REQUIRE(all_ones(proof.verify(ctx, ek, stm, ki, si, cipher)));              // assert!(alice_proof.verify(&cipher, &ek, &dlog_statement))
  auto bad = proof;
  bad.s1.row(1)[0] ^= 1u;                                                     // a tampered proof is `false`, the rest of the batch stays `true`
  const Flags v = bad.verify(ctx, ek, stm, ki, si, cipher);
  for (int i = 0; i < B; ++i) REQUIRE(v[i] == (i == 1 ? 0 : 1));
  return true;
}

// range_proofs.rs:636-709: the 5 x 5 iterations are one batch of 25 items (fresh a, b, beta_prim, r and proof nonces each)
static bool bob_zkp(Context& ctx, const Fixture& F, const EncryptionKeys& alice_public_key, const DLogStatements& dlog_statement) {
  const Batch &a = F["bz_a"], &b = F["bz_b"], &beta_prim = F["bz_beta_prim"], &r = F["bz_r"], &N = F["N"];
  const Index ki = as_index(F["bz_key_idx"]), si = as_index(F["bz_st_idx"]);
  const int B = (int)b.size();
  // Simulate Alice: encrypted_a = Paillier::encrypt(alice_public_key, a)
  const Batch encrypted_a = Paillier::encrypt_with_chosen_randomness(ctx, alice_public_key, ki, a, F["bz_r_enc_a"]);
  // Bob follows MtA: E(a) * b + E(beta_prim; r)
  const Batch b_times_enc_a = Paillier::mul(ctx, alice_public_key, ki, encrypted_a, b);
  const Batch enc_beta_prim = Paillier::encrypt_with_chosen_randomness(ctx, alice_public_key, ki, beta_prim, r);
  const Batch mta_out = Paillier::add(ctx, alice_public_key, ki, b_times_enc_a, enc_beta_prim);
  const mta::range_proofs::BobNonces nn{F["bz_alpha"], F["bz_beta"], F["bz_gamma"], F["bz_rho"], F["bz_rho_prim"], F["bz_sigma"], F["bz_tau"]};
  auto oracle_proof = [&](int check, mta::range_proofs::BobProof& w, Batch& u) {
    w = mta::range_proofs::BobProof{Batch(B, W_N), Batch(B, W_N), Batch(B, W_SCALAR), Batch(B, W_N), Batch(B, W_S1), Batch(B, W_S2), Batch(B, W_T1),
                                    Batch(B, W_S2)};
    u = Batch(B, W_POINT);
    orc_bob_generate(B, (int)N.size(), N.w.data(), (int)F["Nt"].size(), F["Nt"].w.data(), F["h1"].w.data(), F["h2"].w.data(), ki.data(), si.data(),
                     encrypted_a.w.data(), mta_out.w.data(), b.w.data(), beta_prim.w.data(), r.w.data(), nn.alpha.w.data(), nn.beta.w.data(),
                     nn.gamma.w.data(), nn.rho.w.data(), nn.rho_prim.w.data(), nn.sigma.w.data(), nn.tau.w.data(), check, w.t.w.data(), w.z.w.data(),
                     w.e.w.data(), w.s.w.data(), w.s1.w.data(), w.s2.w.data(), w.t1.w.data(), w.t2.w.data(), check ? u.w.data() : nullptr);
  };
  auto same = [](const mta::range_proofs::BobProof& x, const mta::range_proofs::BobProof& y) {
    return x.t == y.t && x.z == y.z && x.e == y.e && x.s == y.s && x.s1 == y.s1 && x.s2 == y.s2 && x.t1 == y.t1 && x.t2 == y.t2;
  };
  mta::range_proofs::BobProof want;
  Batch want_u;
  // let (bob_proof, _) = BobProof::generate(.., false);  assert!(bob_proof.verify(.., None));
  auto [bob_proof, none] = mta::range_proofs::BobProof::generate(ctx, alice_public_key, dlog_statement, ki, si, encrypted_a, mta_out, b, beta_prim, r, nn, false);
  oracle_proof(0, want, want_u);
  REQUIRE(none.size() == 0 && same(bob_proof, want));
  REQUIRE(all_ones(bob_proof.verify(ctx, alice_public_key, dlog_statement, ki, si, encrypted_a, mta_out)));
  // Bob follows MtAwc: X = G * b;  BobProofExt { proof, u };  assert!(bob_proof.verify(.., &X));
  const Batch X = ec_mul_base(ctx, b);
  auto [proof, u] = mta::range_proofs::BobProof::generate(ctx, alice_public_key, dlog_statement, ki, si, encrypted_a, mta_out, b, beta_prim, r, nn, true);
  oracle_proof(1, want, want_u);
  REQUIRE(same(proof, want) && u == want_u);
  const mta::range_proofs::BobProofExt ext{proof, u};
  REQUIRE(all_ones(ext.verify(ctx, alice_public_key, dlog_statement, ki, si, encrypted_a, mta_out, X)));
  // the plain proof does not pass as an extended one (its challenge lacks X, u), and a wrong X is refused item by item
  const mta::range_proofs::BobProofExt mixed{bob_proof, u};
  for (auto v : mixed.verify(ctx, alice_public_key, dlog_statement, ki, si, encrypted_a, mta_out, X)) REQUIRE(v == 0);
  Batch wrongX = X;
  std::memcpy(wrongX.row(3), X.row(4), 64);
  const Flags v = ext.verify(ctx, alice_public_key, dlog_statement, ki, si, encrypted_a, mta_out, wrongX);
  for (int i = 0; i < B; ++i) REQUIRE(v[i] == (i == 3 ? 0 : 1));
  return true;
}

// mta/test.rs:6-19
static bool test_mta(Context& ctx, const Fixture& F, const EncryptionKeys& ek_alice, const DecryptionKeys& dk_alice, const DLogStatements& stm) {
  const Batch &alice_input = F["a"], &bob_input = F["b"];
  const Index ki = as_index(F["key_idx"]);
  const int B = (int)alice_input.size();
  const mta::range_proofs::AliceNonces nn{F["al_alpha"], F["al_beta"], F["al_gamma"], F["al_rho"]};
  const auto m_a = mta::MessageA::a_with_predefined_randomness(ctx, dk_alice, stm, ki, alice_input, F["r_a"], nn);
  auto [m_b, beta, ok_b] = mta::MessageB::b_with_predefined_randomness(ctx, ek_alice, stm, ki, bob_input, m_a, F["mb_r"], F["beta_tag"], F["nonce_b"],
                                                                       F["nonce_bt"]);
  REQUIRE(all_ones(ok_b));                                                    // .unwrap()
  auto [alpha, alice_share, ok_a] = m_b.verify_proofs_get_alpha(ctx, dk_alice, ki, alice_input);
  REQUIRE(all_ones(ok_a));                                                    // .expect("wrong dlog or m_b")
  // let left = alpha.0 + beta; let right = alice_input * bob_input; assert_eq!(left, right);
  mpz_t q, l, rr, x, y;
  mpz_inits(q, l, rr, x, y, NULL);
  mpz_set_str(q, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", 16);
  bool same = true;
  for (int i = 0; i < B; ++i) {
    to_mpz(x, alpha.row(i), W_SCALAR); to_mpz(y, beta.row(i), W_SCALAR);
    mpz_add(l, x, y); mpz_mod(l, l, q);
    to_mpz(x, alice_input.row(i), W_SCALAR); to_mpz(y, bob_input.row(i), W_SCALAR);
    mpz_mul(rr, x, y); mpz_mod(rr, rr, q);
    same = same && mpz_cmp(l, rr) == 0;
  }
  mpz_clears(q, l, rr, x, y, NULL);
  REQUIRE(same);
  // ... and every message byte-identical to the oracle's composition of the same calls
  const Batch& N = F["N"];
  Batch a64(B, W_N), b64(B, W_N), want_ca(B, W_NN), t1(B, W_NN), t2(B, W_NN), want_cb(B, W_NN);
  for (int i = 0; i < B; ++i) { std::memcpy(a64.row(i), alice_input.row(i), 32); std::memcpy(b64.row(i), bob_input.row(i), 32); }
  orc_paillier_encrypt(B, (int)N.size(), N.w.data(), ki.data(), a64.w.data(), F["r_a"].w.data(), want_ca.w.data());
  REQUIRE(m_a.c == want_ca);
  orc_paillier_mul(B, (int)N.size(), N.w.data(), ki.data(), want_ca.w.data(), b64.w.data(), t1.w.data());
  orc_paillier_encrypt(B, (int)N.size(), N.w.data(), ki.data(), F["beta_tag"].w.data(), F["mb_r"].w.data(), t2.w.data());
  orc_paillier_add(B, (int)N.size(), N.w.data(), ki.data(), t1.w.data(), t2.w.data(), want_cb.w.data());
  REQUIRE(m_b.c == want_cb);
  Batch pk(B, W_POINT), R(B, W_POINT), z(B, W_SCALAR);
  orc_dlog_prove(B, bob_input.w.data(), F["nonce_b"].w.data(), pk.w.data(), R.w.data(), z.w.data());
  REQUIRE(m_b.b_proof.pk == pk && m_b.b_proof.pk_t_rand_commitment == R && m_b.b_proof.challenge_response == z);
  Batch share(B, W_N);
  orc_paillier_decrypt(B, (int)N.size(), F["p"].w.data(), F["q"].w.data(), ki.data(), want_cb.w.data(), share.w.data());
  REQUIRE(alice_share == share);
  // Err(InvalidKey): a MessageA whose range proof was tampered with  (mta/mod.rs:119-131)
  auto forged = m_a;
  forged.range_proofs.s.row(2 * stm.count() + 1)[3] ^= 1u;
  auto [m_b2, beta2, ok2] = mta::MessageB::b_with_predefined_randomness(ctx, ek_alice, stm, ki, bob_input, forged, F["mb_r"], F["beta_tag"], F["nonce_b"],
                                                                        F["nonce_bt"]);
  (void)m_b2; (void)beta2;
  for (int i = 0; i < B; ++i) REQUIRE(ok2[i] == (i == 2 ? 0 : 1));
  return true;
}

// zk_pdl_with_slack/test.rs:12-68 and :70-129.  `prover`: the key set the proving side computes with — the public key as in the
// reference's test, or the holder's DecryptionKeys (GG20's phase5_proof_pdl proves about the holder's own ciphertext, party_i.rs:
// 691-717: x^N through p^2 | q^2); the residues, hence the proof bytes, are the same.  The verifier only ever has `ek`.
template <class ProverKeys>
static bool test_zk_pdl_with_slack(Context& ctx, const Fixture& F, const ProverKeys& prover, const EncryptionKeys& ek, const DLogStatements& stm,
                                   bool soundness) {
  const Batch &x = F["a"], &randomness = F["r_a"], &N = F["N"];
  const Index ki = as_index(F["key_idx"]), si = as_index(F["st_idx"]);
  const int B = (int)x.size();
  const Batch Q = ec_mul_base(ctx, x);                                        // let Q = Point::generator() * &x;
  Batch G(B, W_POINT), one(1, W_SCALAR);
  one.row(0)[0] = 1;
  const Batch g1 = ec_mul_base(ctx, one);
  for (int i = 0; i < B; ++i) std::memcpy(G.row(i), g1.row(0), 64);           // G: Point::generator().to_point()
  Batch m(B, W_N);
  for (int i = 0; i < B; ++i) {
    std::memcpy(m.row(i), x.row(i), 32);
    if (soundness) {                                                          // here we encrypt x + 1 instead of x
      for (int k = 0; k < W_N; ++k) { if (++m.row(i)[k] != 0u) break; }
    }
  }
  const Batch c = Paillier::encrypt_with_chosen_randomness(ctx, prover, ki, m, randomness);
  const zk_pdl_with_slack::PDLwSlackStatement statement{c, Q, G, ki, si};
  const zk_pdl_with_slack::PDLwSlackWitness witness{x, randomness};
  const zk_pdl_with_slack::PDLwSlackNonces nn{F["pdl_alpha"], F["pdl_beta"], F["pdl_rho"], F["pdl_gamma"]};
  const auto proof = zk_pdl_with_slack::PDLwSlackProof::prove(ctx, prover, stm, witness, statement, nn);
  Batch z(B, W_N), u1(B, W_POINT), u2(B, W_NN), u3(B, W_N), s1(B, W_S1), s2(B, W_N), s3(B, W_S2);
  orc_pdl_prove(B, (int)N.size(), N.w.data(), (int)F["Nt"].size(), F["Nt"].w.data(), F["h1"].w.data(), F["h2"].w.data(), ki.data(), si.data(),
                c.w.data(), Q.w.data(), G.w.data(), x.w.data(), randomness.w.data(), nn.alpha.w.data(), nn.beta.w.data(), nn.rho.w.data(),
                nn.gamma.w.data(), z.w.data(), u1.w.data(), u2.w.data(), u3.w.data(), s1.w.data(), s2.w.data(), s3.w.data());
  REQUIRE(proof.z == z && proof.u1 == u1 && proof.u2 == u2 && proof.u3 == u3 && proof.s1 == s1 && proof.s2 == s2 && proof.s3 == s3);
  const Flags result = proof.verify(ctx, ek, stm, statement);
  Flags want(B);
  orc_pdl_verify(B, (int)N.size(), N.w.data(), (int)F["Nt"].size(), F["Nt"].w.data(), F["h1"].w.data(), F["h2"].w.data(), ki.data(), si.data(),
                 c.w.data(), Q.w.data(), G.w.data(), z.w.data(), u1.w.data(), u2.w.data(), u3.w.data(), s1.w.data(), s2.w.data(), s3.w.data(), want.data());
  REQUIRE(result == want);
  if (!soundness) {
    REQUIRE(all_ones(result));                                                // assert!(result.is_ok());
  } else {
    for (auto v : result) REQUIRE(v == 0);                                    // #[should_panic]: result.is_ok() must not hold for any item
  }
  return true;
}

// lindell_2017/test.rs:85-137: keys and ephemeral shares exist (the fixture's), party two computes the partial signature, party one
// completes it, `party_one::verify(&signature, &pubkey, &message)`
static bool test_two_party_sign(Context& ctx, const Fixture& F) {
  namespace l17 = two_party_ecdsa::lindell_2017;
  const Index ki = as_index(F["l17_key_idx"]);
  const Batch &N = F["l17_N"], &message = F["l17_msg"];
  const int B = (int)message.size();
  EncryptionKeys ek(ctx, N);                                                  // keypair.ek
  DecryptionKeys dk(ctx, F["l17_p"], F["l17_q"]);                             // party1_private
  const auto partial_sig = l17::party_two::PartialSig::compute(ctx, ek, ki, F["l17_c_key"], F["l17_x2"], F["l17_k2"], F["l17_R1"], message, F["l17_rho"],
                                                               F["l17_r"]);
  const auto signature = l17::party_one::SignatureRecid::compute_with_recid(ctx, dk, ki, partial_sig.c3, F["l17_k1"], F["l17_R2"]);
  Batch want_c3(B, W_NN), wr(B, W_SCALAR), ws(B, W_SCALAR);
  std::vector<int32_t> wrec((size_t)B);
  orc_lindell_partial_sig(B, (int)N.size(), N.w.data(), ki.data(), F["l17_c_key"].w.data(), F["l17_x2"].w.data(), F["l17_k2"].w.data(),
                          F["l17_R1"].w.data(), message.w.data(), F["l17_rho"].w.data(), F["l17_r"].w.data(), want_c3.w.data());
  orc_lindell_sign(B, (int)N.size(), F["l17_p"].w.data(), F["l17_q"].w.data(), ki.data(), want_c3.w.data(), F["l17_k1"].w.data(), F["l17_R2"].w.data(),
                   wr.w.data(), ws.w.data(), wrec.data());
  REQUIRE(partial_sig.c3 == want_c3 && signature.r == wr && signature.s == ws && signature.recid == wrec);
  // party_one::verify(&signature, &pubkey, &message).expect("Invalid signature"): pubkey = x1 x2 G, per item
  std::vector<uint8_t> ok((size_t)B);
  REQUIRE(ossl_ecdsa_verify(B, F["l17_pub"].w.data(), 16, message.w.data(), signature.r.w.data(), signature.s.w.data(), ok.data()) == B);
  REQUIRE(message.row(0)[0] == 1234u);                                        // let message = BigInt::from(1234);
  return true;
}

// ---- gg_2020/state_machine/sign.rs:667-762 ----------------------------------------------------------------------------------------
// `round_based::dev::Simulation`: every party proceeds when it can, its outgoing messages are delivered to all the others
struct Simulation {
  std::vector<sm::OfflineStage*> parties;
  std::map<std::pair<int, int>, Batch> sent;                       // (consuming round, sender ordinal) -> records, for the oracle comparison
  void add_party(sm::OfflineStage& p) { parties.push_back(&p); }
  std::vector<sm::CompletedOfflineStage> run() {
    for (int guard = 0; guard < 64; ++guard) {
      bool all = true;
      for (auto* p : parties) {
        if (p->wants_to_proceed()) p->proceed();
        for (auto& m : p->message_queue()) {
          sent[{m.round, m.sender - 1}] = m.body;
          for (auto* q : parties)
            if (q != p) q->handle_incoming(m);
        }
        p->message_queue().clear();
        all = all && p->is_finished();
      }
      if (all) break;
    }
    std::vector<sm::CompletedOfflineStage> out;
    for (auto* p : parties) out.push_back(p->pick_output().value());
    return out;
  }
};

struct SmCase {
  int t, n, S, B;
  std::vector<uint16_t> s_l;
  std::string pre;
  const Fixture& F;
  const Batch& get(const char* f) const { return F[(pre + f).c_str()]; }
  static Batch rows(const Batch& a, size_t first, size_t count) {
    Batch b(count, a.words);
    std::memcpy(b.w.data(), a.row(first), count * (size_t)a.words * 4);
    return b;
  }
  gg_2020::LocalKey local_key(int party) const {                   // `local_keys[usize::from(keygen_i - 1)]`: only this party's secrets
    gg_2020::LocalKey k;
    k.i = (uint16_t)(party + 1); k.t = (uint16_t)t; k.n = (uint16_t)n;
    k.paillier_key_vec = get("N"); k.n_tilde_vec = get("Nt"); k.h1_vec = get("h1"); k.h2_vec = get("h2"); k.y_sum_s = get("y"); k.pk_vec = get("X");
    k.x_i = rows(get("x"), (size_t)party, 1); k.p = rows(get("p"), (size_t)party, 1); k.q = rows(get("q"), (size_t)party, 1);
    return k;
  }
  Batch of_party(const char* f, int ord) const {                   // [B][S][per] -> [B][per] of signer ordinal `ord`
    const Batch& a = get(f);
    const size_t per = a.size() / ((size_t)B * S);
    Batch b((size_t)B * per, a.words);
    for (int s = 0; s < B; ++s) std::memcpy(b.row((size_t)s * per), a.row(((size_t)s * S + ord) * per), per * (size_t)a.words * 4);
    return b;
  }
  gg_2020::SignNonces sampled(int ord) const {
    auto g = [&](const char* f) { return of_party(f, ord); };
    return gg_2020::SignNonces{g("k"), g("gamma"), g("blind"), g("r_a"), g("al_alpha"), g("al_beta"), g("al_gamma"), g("al_rho"), g("mb_beta_tag"),
                               g("mb_r"), g("mb_nonce_b"), g("mb_nonce_bt"), g("l"), g("ped_s1"), g("ped_s2"), g("pdl_alpha"), g("pdl_beta"),
                               g("pdl_rho"), g("pdl_gamma"), g("heg_s1"), g("heg_s2")};
  }
};

static SmCase sm_case(const Fixture& F, int k) {
  const std::string pre = "sm" + std::to_string(k) + "_";
  const Batch& sh = F[(pre + "shape").c_str()];
  SmCase c{(int)sh.w[0], (int)sh.w[1], (int)sh.w[2], (int)sh.w[3], {}, pre, F};
  for (uint32_t v : F[(pre + "signers").c_str()].w) c.s_l.push_back((uint16_t)(v + 1));
  return c;
}

// simulate_offline_stage + simulate_signing (sign.rs:673-724) for one signer set, `B` sessions at once
static bool simulate_signing(Context& ctx, const SmCase& c) {
  std::vector<std::unique_ptr<sm::OfflineStage>> stages;
  Simulation simulation;
  for (int i = 1; i <= c.S; ++i) {                                 // for (i, &keygen_i) in (1..).zip(s_l)
    stages.emplace_back(new sm::OfflineStage(ctx, (uint16_t)i, c.s_l, c.local_key(c.s_l[(size_t)i - 1] - 1), c.B, c.sampled(i - 1)));
    simulation.add_party(*stages.back());
  }
  const auto offline = simulation.run();                           // .unwrap()
  for (auto& st : stages)
    for (int32_t v : st->status()) REQUIRE(v == 0);
  const Batch& message = c.get("msg");
  const Batch pk = offline[0].public_key();
  std::vector<sm::SignManual> parties;
  std::vector<sm::PartialSignature> local_sigs;
  for (auto& o : offline) {                                        // SignManual::new(message.clone(), o.clone())
    auto made = sm::SignManual::new_(message, o);
    parties.push_back(std::move(made.first));
    local_sigs.push_back(std::move(made.second));
  }
  // the oracle's lock-step run of the same sessions: every message of every party and the signature
  std::vector<int32_t> signers;
  for (auto v : c.s_l) signers.push_back(v - 1);
  const orc_gg20_keys K{c.t, c.n, c.S, 1, signers.data(), c.get("x").w.data(), c.get("p").w.data(), c.get("q").w.data(), c.get("N").w.data(),
                        c.get("Nt").w.data(), c.get("h1").w.data(), c.get("h2").w.data(), c.get("y").w.data(), c.get("X").w.data()};
  const orc_gg20_nonces Z{c.get("k").w.data(), c.get("gamma").w.data(), c.get("blind").w.data(), c.get("r_a").w.data(), c.get("al_alpha").w.data(),
                          c.get("al_beta").w.data(), c.get("al_gamma").w.data(), c.get("al_rho").w.data(), c.get("mb_beta_tag").w.data(),
                          c.get("mb_r").w.data(), c.get("mb_nonce_b").w.data(), c.get("mb_nonce_bt").w.data(), c.get("l").w.data(),
                          c.get("ped_s1").w.data(), c.get("ped_s2").w.data(), c.get("pdl_alpha").w.data(), c.get("pdl_beta").w.data(),
                          c.get("pdl_rho").w.data(), c.get("pdl_gamma").w.data(), c.get("heg_s1").w.data(), c.get("heg_s2").w.data(), message.w.data()};
  std::vector<std::vector<uint32_t>> slab(7);
  uint32_t* slabs[7];
  const int emitting[7] = {0, 1, 2, 3, 4, 5, 7};
  for (int m = 0; m < 7; ++m) {
    slab[(size_t)m].assign((size_t)c.S * c.B * orc_gg20_msg_words(c.S, c.n, emitting[m]), 0u);
    slabs[m] = slab[(size_t)m].data();
  }
  Batch wr(c.B, W_SCALAR), ws(c.B, W_SCALAR);
  std::vector<int32_t> wrec((size_t)c.B), wst((size_t)c.B);
  orc_gg20_sign_ex(&K, &Z, nullptr, c.B, 0, c.B, slabs, wr.w.data(), ws.w.data(), wrec.data(), nullptr, wst.data(), nullptr, nullptr);
  for (int32_t v : wst) REQUIRE(v == 0);
  for (int m = 0; m < 6; ++m)
    for (int j = 0; j < c.S; ++j) {
      const Batch& got = simulation.sent.at({m + 1, j});
      const size_t blk = (size_t)c.B * got.words;
      REQUIRE(got.words == orc_gg20_msg_words(c.S, c.n, m));
      REQUIRE(std::memcmp(got.w.data(), slab[(size_t)m].data() + (size_t)j * blk, blk * 4) == 0);       // byte-identical round message
    }
  for (int j = 0; j < c.S; ++j) REQUIRE(std::memcmp(local_sigs[(size_t)j].s_i.w.data(), slab[6].data() + (size_t)j * c.B * 8, (size_t)c.B * 32) == 0);
  // parties.into_iter().enumerate().map(|(i, p)| p.complete(&local_sigs_except(i)).unwrap()).all(|signature| verify(&signature, &pk, &message).is_ok())
  for (int i = 0; i < c.S; ++i) {
    std::vector<sm::PartialSignature> except;
    for (int j = 0; j < c.S; ++j)
      if (j != i) except.push_back(local_sigs[(size_t)j]);
    const sm::SignatureRecid sig = parties[(size_t)i].complete(except);
    for (int32_t v : sig.status) REQUIRE(v == 0);
    std::vector<uint8_t> ok((size_t)c.B);
    REQUIRE(ossl_ecdsa_verify(c.B, pk.w.data(), 0, message.w.data(), sig.r.w.data(), sig.s.w.data(), ok.data()) == c.B);
    REQUIRE(sig.r == wr && sig.s == ws && sig.recid == wrec);
  }
  return true;
}

// Party-sharded signing over the C-ABI's RCCL fan-out (mpe_comm_*, mpe_gg20_round_exchange; mpecdsa.hpp: sharded::PartySharded): every
// hosted (block, party) pair ends with the oracle's signature.  world = 1: all pairs on this rank, the all-gathers still run on the real
// communicator.  world > 1 (one process per GPU): rank r hosts party p of block s when (s + p) % world == r; block s signs the fixture's
// sessions with its own messages (the fixture's, plus s in the lowest word) — every rank checks the pairs it hosts.
static bool party_sharded(Context& ctx, const SmCase& c, int rank, int world, const std::vector<uint8_t>& id) {
  namespace sh = gg_2020::sharded;
  sh::Comm comm(ctx, id, rank, world);
  std::vector<sh::Hosted> hosted;
  const int blocks = mpe_gg20_shard_blocks(MPE_PLACE_ROTATED, c.S, world);
  auto block_msg = [&](int s) { Batch m = c.get("msg"); for (size_t b = 0; b < m.size(); ++b) m.row(b)[0] ^= (uint32_t)s; return m; };
  for (int s = 0; s < blocks; ++s)
    for (int p = 0; p < c.S; ++p) {
      int r = -1, slot = -1;
      REQUIRE(mpe_gg20_shard_where(MPE_PLACE_ROTATED, c.S, world, s, p, &r, &slot) == MPE_OK);
      if (r != rank) continue;
      hosted.push_back(sh::Hosted{s, p, c.local_key(c.s_l[(size_t)p] - 1), c.sampled(p), block_msg(s)});
    }
  sh::PartySharded ps(ctx, comm, MPE_PLACE_ROTATED, c.s_l, c.B, std::move(hosted));
  REQUIRE(ps.gather_mode() == 0 || ps.gather_mode() == 1);
  const auto res = ps.run();
  REQUIRE((int)res.size() == c.S);
  std::vector<int32_t> signers;
  for (auto v : c.s_l) signers.push_back(v - 1);
  const orc_gg20_keys K{c.t, c.n, c.S, 1, signers.data(), c.get("x").w.data(), c.get("p").w.data(), c.get("q").w.data(), c.get("N").w.data(),
                        c.get("Nt").w.data(), c.get("h1").w.data(), c.get("h2").w.data(), c.get("y").w.data(), c.get("X").w.data()};
  for (const auto& pr : res) {
    const Batch message = block_msg(pr.block);
    const orc_gg20_nonces Z{c.get("k").w.data(), c.get("gamma").w.data(), c.get("blind").w.data(), c.get("r_a").w.data(), c.get("al_alpha").w.data(),
                            c.get("al_beta").w.data(), c.get("al_gamma").w.data(), c.get("al_rho").w.data(), c.get("mb_beta_tag").w.data(),
                            c.get("mb_r").w.data(), c.get("mb_nonce_b").w.data(), c.get("mb_nonce_bt").w.data(), c.get("l").w.data(),
                            c.get("ped_s1").w.data(), c.get("ped_s2").w.data(), c.get("pdl_alpha").w.data(), c.get("pdl_beta").w.data(),
                            c.get("pdl_rho").w.data(), c.get("pdl_gamma").w.data(), c.get("heg_s1").w.data(), c.get("heg_s2").w.data(), message.w.data()};
    Batch wr(c.B, W_SCALAR), ws(c.B, W_SCALAR);
    std::vector<int32_t> wrec((size_t)c.B), wst((size_t)c.B);
    orc_gg20_sign(&K, &Z, 0, c.B, wr.w.data(), ws.w.data(), wrec.data(), nullptr, wst.data());
    for (int32_t v : wst) REQUIRE(v == 0);
    for (int32_t v : pr.status) REQUIRE(v == 0);
    REQUIRE(pr.r == wr && pr.s == ws && pr.recid == wrec);
    std::vector<uint8_t> ok((size_t)c.B);
    REQUIRE(ossl_ecdsa_verify(c.B, c.get("y").w.data(), 0, message.w.data(), pr.r.w.data(), pr.s.w.data(), ok.data()) == c.B);
  }
  return true;
}

template <class F>
static bool throws(sm::Error::Kind kind, F&& f) {
  try { f(); } catch (const sm::Error& e) { return e.kind == kind; }
  return false;
}

// OfflineStage::new's argument checks (sign.rs:78-101), the message stores (:246-297) and pick_output (:318-330)
static bool state_machine_errors(Context& ctx, const SmCase& c) {
  const auto key = c.local_key(c.s_l[0] - 1);
  const auto nn = c.sampled(0);
  REQUIRE(throws(sm::Error::TooFewParties, [&] { sm::OfflineStage(ctx, 1, {c.s_l[0]}, key, c.B, nn); }));
  REQUIRE(throws(sm::Error::InvalidPartyIndex, [&] { sm::OfflineStage(ctx, 0, c.s_l, key, c.B, nn); }));
  REQUIRE(throws(sm::Error::InvalidPartyIndex, [&] { sm::OfflineStage(ctx, (uint16_t)(c.S + 1), c.s_l, key, c.B, nn); }));
  REQUIRE(throws(sm::Error::InvalidSl, [&] { sm::OfflineStage(ctx, 1, {c.s_l[0], (uint16_t)(c.n + 1)}, key, c.B, nn); }));
  REQUIRE(throws(sm::Error::InvalidSl, [&] { sm::OfflineStage(ctx, 1, {c.s_l[0], 0}, key, c.B, nn); }));
  REQUIRE(throws(sm::Error::InvalidSl, [&] { sm::OfflineStage(ctx, 1, {c.s_l[0], c.s_l[0]}, key, c.B, nn); }));
  sm::OfflineStage a(ctx, 1, c.s_l, key, c.B, nn), b(ctx, 2, c.s_l, c.local_key(c.s_l[1] - 1), c.B, c.sampled(1));
  REQUIRE(a.current_round() == 0 && a.party_ind() == 1 && a.parties() == c.S && !a.is_finished() && a.wants_to_proceed());
  REQUIRE(!a.pick_output().has_value());                            // None while running
  a.proceed();
  b.proceed();
  REQUIRE(a.current_round() == 1 && a.message_queue().size() == 1 && a.message_queue()[0].round == 1 && a.message_queue()[0].sender == 1);
  REQUIRE(c.S > 2 || !a.wants_to_proceed());
  const sm::Msg from_b = b.message_queue()[0];
  REQUIRE(throws(sm::Error::HandleMessage, [&] { a.handle_incoming(a.message_queue()[0]); }));           // my own message
  a.handle_incoming(from_b);
  REQUIRE(throws(sm::Error::HandleMessage, [&] { a.handle_incoming(from_b); }));                          // overwrite
  sm::Msg bad = from_b;
  bad.sender = (uint16_t)(c.S + 1);
  REQUIRE(throws(sm::Error::HandleMessage, [&] { a.handle_incoming(bad); }));
  bad = from_b; bad.round = 7;
  REQUIRE(throws(sm::Error::ReceivedOutOfOrderMessage, [&] { a.handle_incoming(bad); }));
  if (c.S == 2) {
    REQUIRE(a.wants_to_proceed());
    a.proceed();
    REQUIRE(a.current_round() == 2);
    bad = from_b;                                                                                          // round 1 is over: its store is gone
    REQUIRE(throws(sm::Error::ReceivedOutOfOrderMessage, [&] { a.handle_incoming(bad); }));
  }
  return true;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: test_shim <fixture.bin>\n"); return 2; }
  Fixture F{load_fixture(argv[1])};
  int failed = 0;
  if (argc == 7 && std::string(argv[2]) == "--sharded") {
    // one rank of a multi-process party-sharded run: test_shim <fixture> --sharded <rank> <world> <id file> <case>; device = rank
    const int rank = std::atoi(argv[3]), world = std::atoi(argv[4]), k = std::atoi(argv[6]);
    try {
      Context ctx(rank);
      std::vector<uint8_t> id;
      if (rank == 0) {
        id = gg_2020::sharded::Comm::unique_id();
        const std::string tmp = std::string(argv[5]) + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id.data(), 1, id.size(), f) != id.size()) return 3;
        std::fclose(f);
        std::rename(tmp.c_str(), argv[5]);                       // the other ranks see a complete file or none
      } else {
        for (int tries = 0; tries < 600 && id.empty(); ++tries) {
          if (FILE* f = std::fopen(argv[5], "rb")) {
            id.resize(MPE_COMM_ID_BYTES);
            if (std::fread(id.data(), 1, id.size(), f) != id.size()) id.clear();
            std::fclose(f);
          }
          if (id.empty()) { struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr); }
        }
        if (id.empty()) return 4;
      }
      const bool ok = party_sharded(ctx, sm_case(F, k), rank, world, id);
      std::printf("rank %d of %d: party_sharded ... %s\n", rank, world, ok ? "ok" : "FAILED");
      return ok ? 0 : 1;
    } catch (const std::exception& e) {
      std::printf("rank %d EXCEPTION %s\n", rank, e.what());
      return 1;
    }
  }
  try {
    Context ctx(0);
    {  // options are the context's, not the environment's: an unknown key is refused, a known one round-trips
      bool refused = false;
      try { ctx.set_option("no_such_option", "1"); } catch (const mpecdsa::Error&) { refused = true; }
      ctx.set_option("fb_window_bits", "10");
      const bool ok = refused && ctx.option("fb_window_bits") == 10 && ctx.option("sampler_max_attempts") == 128;
      ctx.set_option("fb_window_bits", "13");
      std::printf("test context_options ... %s\n", ok ? "ok" : "FAILED");
      failed += ok ? 0 : 1;
    }
    EncryptionKeys ek(ctx, F["N"]);
    DecryptionKeys dk(ctx, F["p"], F["q"]);
    DLogStatements stm(ctx, F["Nt"], F["h1"], F["h2"]);
    struct { const char* name; bool ok; } results[] = {
        {"alice_zkp", alice_zkp(ctx, F, ek, stm)},
        {"bob_zkp", bob_zkp(ctx, F, ek, stm)},
        {"test_mta", test_mta(ctx, F, ek, dk, stm)},
        {"test_zk_pdl_with_slack", test_zk_pdl_with_slack(ctx, F, ek, ek, stm, false)},
        {"test_zk_pdl_with_slack_soundness", test_zk_pdl_with_slack(ctx, F, ek, ek, stm, true)},
        {"test_zk_pdl_with_slack_key_holder_proves", test_zk_pdl_with_slack(ctx, F, dk, ek, stm, false)},
        {"test_zk_pdl_with_slack_soundness_key_holder_proves", test_zk_pdl_with_slack(ctx, F, dk, ek, stm, true)},
    };
    for (auto& r : results) {
      std::printf("test %s ... %s\n", r.name, r.ok ? "ok" : "FAILED");
      failed += r.ok ? 0 : 1;
    }
    {
      const bool ok = test_two_party_sign(ctx, F);
      std::printf("test test_two_party_sign ... %s\n", ok ? "ok" : "FAILED");
      failed += ok ? 0 : 1;
    }
    const int ncases = (int)F["sm_count"].w[0];
    for (int k = 0; k < ncases; ++k) {
      const SmCase c = sm_case(F, k);
      const bool ok = simulate_signing(ctx, c);
      std::printf("test simulate_signing_t%d_n%d_s%d [", c.t, c.n, c.S);
      for (size_t j = 0; j < c.s_l.size(); ++j) std::printf("%s%d", j ? ", " : "", (int)c.s_l[j]);
      std::printf("] ... %s\n", ok ? "ok" : "FAILED");
      failed += ok ? 0 : 1;
    }
    {
      const SmCase c = sm_case(F, 0);
      bool ok = state_machine_errors(ctx, c);
      // a completed offline stage signs one message only (a second signature with the same k_i would leak the key share)
      std::vector<std::unique_ptr<sm::OfflineStage>> st;
      Simulation sim;
      for (int i = 1; i <= c.S; ++i) {
        st.emplace_back(new sm::OfflineStage(ctx, (uint16_t)i, c.s_l, c.local_key(c.s_l[(size_t)i - 1] - 1), c.B, c.sampled(i - 1)));
        sim.add_party(*st.back());
      }
      const auto done = sim.run();
      ok = ok && throws(sm::Error::DoublePickOutput, [&] { (void)st[0]->pick_output(); });
      (void)sm::SignManual::new_(c.get("msg"), done[0]);
      const sm::CompletedOfflineStage copy = done[0];
      ok = ok && throws(sm::Error::OfflineStageReused, [&] { (void)sm::SignManual::new_(c.get("msg"), copy); });
      std::printf("test state_machine_errors ... %s\n", ok ? "ok" : "FAILED");
      failed += ok ? 0 : 1;
    }
    for (int k : {1, 4}) {                                         // t=1 n=3 {1,2} and t=2 n=3 {1,2,3}: world 1, the collectives on the real RCCL communicator
      const SmCase c = sm_case(F, k);
      const bool ok = party_sharded(ctx, c, 0, 1, gg_2020::sharded::Comm::unique_id());
      std::printf("test party_sharded_rccl_world1_t%d_n%d_s%d ... %s\n", c.t, c.n, c.S, ok ? "ok" : "FAILED");
      failed += ok ? 0 : 1;
    }
    // argument errors surface as exceptions carrying mpe_last_error()
    bool threw = false;
    try {
      mpe_encoding bad;
      mpe_encoding_default(&bad);
      bad.chain_point = 9;
      ctx.set_encoding(bad);
    } catch (const Error& e) {
      threw = e.code == MPE_E_ARG && std::string(e.what()).find("permutation") != std::string::npos;
    }
    std::printf("test error_mapping ... %s\n", threw ? "ok" : "FAILED");
    failed += threw ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  std::printf("%s\n", failed ? "SOME TESTS FAILED" : "all tests passed");
  return failed ? 1 : 0;
}
