// mpecdsa.hpp — the host side ABOVE the C-ABI, in C++17, with the reference's names.
//
// The reference (ZenGo-X/multi-party-ecdsa v0.8.1) is compiled Rust; its toolchain is absent from the build image, so the layer a
// Rust maintainer would write over the `extern "C"` bindings exists here as a header-only C++ mirror of the
// reference's own call surface for the hot path — same type and method names, same argument meaning, same error behaviour — over
// `include/mpecdsa_hip.h`.  Every method is the BATCHED form of the reference call it is named after and cites it:
//
//   paillier::Paillier::{encrypt_with_chosen_randomness, decrypt, add, mul}     src/utilities/mta/mod.rs:22-24,68-75,133-145,165
//   mta::range_proofs::AliceProof::{generate, verify}                            src/utilities/mta/range_proofs.rs:105-193
//   mta::range_proofs::{BobProof::{generate, verify}, BobProofExt::verify}       src/utilities/mta/range_proofs.rs:218-534
//   mta::{MessageA::a_with_predefined_randomness, MessageB::b_with_predefined_randomness,
//         MessageB::verify_proofs_get_alpha}                                     src/utilities/mta/mod.rs:62-179
//   zk_pdl_with_slack::PDLwSlackProof::{prove, verify}                           src/utilities/zk_pdl_with_slack/mod.rs:68-179
//   curv DLogProof::{prove, verify}                                              mta/mod.rs:147-148,170-171
//   two_party_ecdsa::lindell_2017::{party_two::PartialSig::compute, party_one::Signature::compute_with_recid}
//                                                                                 src/protocols/two_party_ecdsa/lindell_2017/party_two.rs:390-423, party_one.rs:519-565
//   gg_2020::state_machine::sign::{OfflineStage, CompletedOfflineStage, SignManual}  gg_2020/state_machine/sign.rs:66-330,540-646
//       (one party of `batch` concurrent signing sessions: `RoundN::proceed`, state_machine/sign/rounds.rs:68-692)
//
// Values sampled from OsRng inside the reference's primitives are explicit arguments (`*Nonces`), which is what makes a bit-exact
// comparison possible; `bool` / `Result<(), _>` returns become one flag per item (a bad item never aborts the batch — the
// reference returns Err(InvalidKey) / false per call, mta/mod.rs:120-131,177).  A negative status of the C-ABI (bad argument,
// HIP failure) throws `mpecdsa::Error` carrying mpe_last_error().
//
// Host data model: `Batch` = item-major little-endian 32-bit words of fixed width (the interface words of mpecdsa_hip.h; a
// curv BigInt converts by reversing its big-endian bytes and zero-padding).  Buffers live on the device only inside a call: this
// layer uploads, calls, downloads — the throughput path keeps its data resident and calls the C-ABI directly (mpe_gg20_*).
// Used by tests/cpp/test_shim.cpp (mirrors the reference's own tests) — no torch, no Python.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "mpecdsa_hip.h"

namespace mpecdsa {

struct Error : std::runtime_error {
  int code;
  Error(const std::string& what, int rc) : std::runtime_error(what + ": " + (mpe_last_error() ? mpe_last_error() : "")), code(rc) {}
};
inline void check(int rc, const char* what) {
  if (rc != MPE_OK) throw Error(what, rc);
}
inline void check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// widths in 32-bit words (mpecdsa_hip.h)
enum : int { W_SCALAR = 8, W_POINT = 16, W_Q3 = 24, W_S1 = 25, W_PRIME = 32, W_N = 64, W_RHO = 72, W_Q2N = 80, W_T1 = 81, W_GAMMA = 88, W_S2 = 89, W_NN = 128 };

// item-major batch of fixed-width little-endian integers (or affine points x[8] | y[8]) on the host
struct Batch {
  int words = 0;
  std::vector<uint32_t> w;
  Batch() = default;
  Batch(size_t items, int words_) : words(words_), w(items * (size_t)words_, 0u) {}
  size_t size() const { return words ? w.size() / (size_t)words : 0; }
  uint32_t* row(size_t i) { return w.data() + i * (size_t)words; }
  const uint32_t* row(size_t i) const { return w.data() + i * (size_t)words; }
  bool operator==(const Batch& o) const { return words == o.words && w == o.w; }
  bool operator!=(const Batch& o) const { return !(*this == o); }
};

// device buffer for the duration of a call.  secret = true: key material and sampled values (x_i, p, q, k_i, gamma_i, Paillier
// randomness) — zeroed before the memory goes back to the allocator, as the C-ABI does with its own state (mpe_ctx_wipe, session
// destroy / rearm) and as the reference zeroizes its round secrets (range_proofs.rs:26-36)
template <class T>
class Dev {
 public:
  explicit Dev(size_t n, bool secret = false) : n_(n), secret_(secret) { check_hip(hipMalloc((void**)&p_, (n ? n : 1) * sizeof(T)), "hipMalloc"); }
  explicit Dev(const std::vector<T>& h, bool secret = false) : Dev(h.size(), secret) {
    if (!h.empty()) check_hip(hipMemcpy(p_, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
  }
  ~Dev() {
    if (!p_) return;
    if (secret_ && n_) (void)hipMemset(p_, 0, n_ * sizeof(T));
    (void)hipFree(p_);
  }
  Dev(const Dev&) = delete;
  Dev& operator=(const Dev&) = delete;
  T* get() const { return p_; }
  std::vector<T> download() const {
    std::vector<T> h(n_);
    if (n_) check_hip(hipMemcpy(h.data(), p_, n_ * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return h;
  }
 private:
  T* p_ = nullptr;
  size_t n_ = 0;
  bool secret_ = false;
};
inline Dev<uint32_t> up(const Batch& b) { return Dev<uint32_t>(b.w); }
inline Dev<uint32_t> up_secret(const Batch& b) { return Dev<uint32_t>(b.w, true); }
inline Batch down(const Dev<uint32_t>& d, int words) { Batch b; b.words = words; b.w = d.download(); return b; }
using Index = std::vector<int32_t>;       // per-item key / statement index
using Flags = std::vector<uint8_t>;       // per-item verdict (1 = Ok / true)

class Context {
 public:
  explicit Context(int device = 0) { check(mpe_ctx_create(&h_, device), "mpe_ctx_create"); }
  ~Context() { if (h_) (void)mpe_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  mpe_ctx* get() const { return h_; }
  void sync() const { check(mpe_sync(h_, nullptr), "mpe_sync"); }
  void set_encoding(const mpe_encoding& e) { check(mpe_ctx_set_encoding(h_, &e), "mpe_ctx_set_encoding"); }
  // the library reads no environment variable: an A/B switch of the measurements is an option of the context (mpecdsa_hip.h)
  void set_option(const char* key, const char* value) { check(mpe_ctx_set_option(h_, key, value), "mpe_ctx_set_option"); }
  long option(const char* key) const { long v = 0; check(mpe_ctx_get_option(h_, key, &v), "mpe_ctx_get_option"); return v; }
 private:
  mpe_ctx* h_ = nullptr;
};

namespace paillier {
// `EncryptionKey{n, nn}` x nkeys (kzen-paillier): what a peer holds
class EncryptionKeys {
 public:
  EncryptionKeys(Context& ctx, const Batch& n) : n_(n) {
    Dev<uint32_t> d(n.w);
    check(mpe_paillier_create_public(ctx.get(), (int)n.size(), d.get(), &h_, nullptr), "mpe_paillier_create_public");
    ctx.sync();
  }
  ~EncryptionKeys() { if (h_) (void)mpe_paillier_destroy(h_); }
  EncryptionKeys(const EncryptionKeys&) = delete;
  const mpe_paillier* get() const { return h_; }
 private:
  Batch n_;
  mpe_paillier* h_ = nullptr;
};
// `DecryptionKey{p, q}` x nkeys: the key holder (also encrypts, through p^2 | q^2)
class DecryptionKeys {
 public:
  DecryptionKeys(Context& ctx, const Batch& p, const Batch& q) {
    Dev<uint32_t> dp(p.w), dq(q.w);
    check(mpe_paillier_create_private(ctx.get(), (int)p.size(), dp.get(), dq.get(), &h_, nullptr), "mpe_paillier_create_private");
    ctx.sync();
  }
  ~DecryptionKeys() { if (h_) (void)mpe_paillier_destroy(h_); }
  DecryptionKeys(const DecryptionKeys&) = delete;
  const mpe_paillier* get() const { return h_; }
 private:
  mpe_paillier* h_ = nullptr;
};

struct Paillier {
  // `Paillier::encrypt_with_chosen_randomness(&ek, RawPlaintext::from(m), &Randomness::from(r))`   mta/mod.rs:68-75,133-137
  template <class Keys>
  static Batch encrypt_with_chosen_randomness(Context& ctx, const Keys& ek, const Index& key_idx, const Batch& m, const Batch& r) {
    const int B = (int)m.size();
    Dev<uint32_t> dm = up(m), dr = up(r), dc((size_t)B * W_NN);
    Dev<int32_t> di(key_idx);
    check(mpe_paillier_encrypt(ctx.get(), ek.get(), B, di.get(), dm.get(), dr.get(), dc.get(), nullptr), "mpe_paillier_encrypt");
    ctx.sync();
    return down(dc, W_NN);
  }
  // `Paillier::decrypt(&dk, &RawCiphertext::from(c))`   mta/mod.rs:165; party_i.rs:455-457
  static Batch decrypt(Context& ctx, const DecryptionKeys& dk, const Index& key_idx, const Batch& c) {
    const int B = (int)c.size();
    Dev<uint32_t> dc = up(c), dm((size_t)B * W_N);
    Dev<int32_t> di(key_idx);
    check(mpe_paillier_decrypt(ctx.get(), dk.get(), B, di.get(), dc.get(), dm.get(), nullptr), "mpe_paillier_decrypt");
    ctx.sync();
    return down(dm, W_N);
  }
  // `Paillier::add(&ek, c1, c2)`   mta/mod.rs:145
  template <class Keys>
  static Batch add(Context& ctx, const Keys& ek, const Index& key_idx, const Batch& c1, const Batch& c2) {
    const int B = (int)c1.size();
    Dev<uint32_t> a = up(c1), b = up(c2), o((size_t)B * W_NN);
    Dev<int32_t> di(key_idx);
    check(mpe_paillier_add(ctx.get(), ek.get(), B, di.get(), a.get(), b.get(), o.get(), nullptr), "mpe_paillier_add");
    ctx.sync();
    return down(o, W_NN);
  }
  // `Paillier::mul(&ek, c, RawPlaintext::from(k))`   mta/mod.rs:140-144
  template <class Keys>
  static Batch mul(Context& ctx, const Keys& ek, const Index& key_idx, const Batch& c, const Batch& k) {
    const int B = (int)c.size();
    Dev<uint32_t> a = up(c), b = up(k), o((size_t)B * W_NN);
    Dev<int32_t> di(key_idx);
    check(mpe_paillier_mul(ctx.get(), ek.get(), B, di.get(), a.get(), b.get(), k.words, o.get(), nullptr), "mpe_paillier_mul");
    ctx.sync();
    return down(o, W_NN);
  }
};
}  // namespace paillier

namespace zk_paillier {
// `DLogStatement{N, g, ni}` x count, stored by GG20 as (N~, h1, h2)   party_i.rs:225-229
class DLogStatements {
 public:
  DLogStatements(Context& ctx, const Batch& n_tilde, const Batch& h1, const Batch& h2) : count_((int)n_tilde.size()) {
    Dev<uint32_t> a = up(n_tilde), b = up(h1), c = up(h2);
    check(mpe_statements_create(ctx.get(), count_, a.get(), b.get(), c.get(), &h_, nullptr), "mpe_statements_create");
    ctx.sync();
  }
  ~DLogStatements() { if (h_) (void)mpe_statements_destroy(h_); }
  DLogStatements(const DLogStatements&) = delete;
  const mpe_statements* get() const { return h_; }
  int count() const { return count_; }
 private:
  mpe_statements* h_ = nullptr;
  int count_ = 0;
};
}  // namespace zk_paillier

namespace curv {
// `DLogProof<Secp256k1, Sha256>{pk, pk_t_rand_commitment, challenge_response}`
struct DLogProof {
  Batch pk, pk_t_rand_commitment, challenge_response;
  // `DLogProof::prove(&sk)` with the nonce an input
  static DLogProof prove(Context& ctx, const Batch& sk, const Batch& nonce) {
    const int B = (int)sk.size();
    Dev<uint32_t> s = up(sk), k = up(nonce), pk((size_t)B * W_POINT), R((size_t)B * W_POINT), z((size_t)B * W_SCALAR);
    check(mpe_dlog_prove(ctx.get(), B, s.get(), k.get(), pk.get(), R.get(), z.get(), nullptr), "mpe_dlog_prove");
    ctx.sync();
    return DLogProof{down(pk, W_POINT), down(R, W_POINT), down(z, W_SCALAR)};
  }
  // `DLogProof::verify(&proof) -> Result<(), ProofError>`
  Flags verify(Context& ctx) const {
    const int B = (int)pk.size();
    Dev<uint32_t> a = up(pk), b = up(pk_t_rand_commitment), c = up(challenge_response);
    Dev<uint8_t> ok((size_t)B);
    check(mpe_dlog_verify(ctx.get(), B, a.get(), b.get(), c.get(), ok.get(), nullptr), "mpe_dlog_verify");
    ctx.sync();
    return ok.download();
  }
};
}  // namespace curv

namespace mta {
namespace range_proofs {
// the values `AliceProof::generate` samples (range_proofs.rs:48-51): alpha < q^3, beta in Z*_N, gamma < q^3 N~, rho < q N~
struct AliceNonces { Batch alpha, beta, gamma, rho; };

// `AliceProof{z, e, s, s1, s2}`   range_proofs.rs:95-101
struct AliceProof {
  Batch z, e, s, s1, s2;
  // `AliceProof::generate(a, cipher, alice_ek, dlog_statement, r)`   range_proofs.rs:160-193
  template <class Keys>
  static AliceProof generate(Context& ctx, const Keys& alice_ek, const zk_paillier::DLogStatements& stm, const Index& key_idx,
                             const Index& st_idx, const Batch& a, const Batch& cipher, const Batch& r, const AliceNonces& nn) {
    const int B = (int)a.size();
    Dev<uint32_t> da = up(a), dc = up(cipher), dr = up(r), al = up(nn.alpha), be = up(nn.beta), ga = up(nn.gamma), rh = up(nn.rho);
    Dev<uint32_t> z((size_t)B * W_N), e((size_t)B * W_SCALAR), s((size_t)B * W_N), s1((size_t)B * W_S1), s2((size_t)B * W_S2);
    Dev<int32_t> ki(key_idx), si(st_idx);
    const mpe_alice_nonces n{al.get(), be.get(), ga.get(), rh.get()};
    const mpe_alice_proof p{z.get(), e.get(), s.get(), s1.get(), s2.get()};
    check(mpe_alice_generate(ctx.get(), alice_ek.get(), stm.get(), B, ki.get(), si.get(), da.get(), dc.get(), dr.get(), &n, &p, nullptr),
          "mpe_alice_generate");
    ctx.sync();
    return AliceProof{down(z, W_N), down(e, W_SCALAR), down(s, W_N), down(s1, W_S1), down(s2, W_S2)};
  }
  // `AliceProof::verify(&self, cipher, alice_ek, dlog_statement) -> bool`   range_proofs.rs:105-156
  template <class Keys>
  Flags verify(Context& ctx, const Keys& alice_ek, const zk_paillier::DLogStatements& stm, const Index& key_idx, const Index& st_idx,
               const Batch& cipher) const {
    const int B = (int)z.size();
    Dev<uint32_t> dz = up(z), de = up(e), ds = up(s), d1 = up(s1), d2 = up(s2), dc = up(cipher);
    Dev<int32_t> ki(key_idx), si(st_idx);
    Dev<uint8_t> ok((size_t)B);
    const mpe_alice_proof p{dz.get(), de.get(), ds.get(), d1.get(), d2.get()};
    check(mpe_alice_verify(ctx.get(), alice_ek.get(), stm.get(), B, ki.get(), si.get(), dc.get(), &p, ok.get(), nullptr), "mpe_alice_verify");
    ctx.sync();
    return ok.download();
  }
};

// the values `BobProof::generate` samples (range_proofs.rs:236-244): alpha < q^3, beta in Z*_N, gamma < q^2 N, rho, sigma < q N~,
// rho_prim, tau < q^3 N~
struct BobNonces { Batch alpha, beta, gamma, rho, rho_prim, sigma, tau; };

// `BobProof{t, z, e, s, s1, s2, t1, t2}`   range_proofs.rs:218-228
struct BobProof {
  Batch t, z, e, s, s1, s2, t1, t2;
  // `BobProof::generate(a_encrypted, mta_encrypted, b, beta_prim, alice_ek, dlog_statement, r, check) -> (BobProof, Option<Point>)`
  // range_proofs.rs:231-319.  check = true also returns u = alpha G and hashes X = b G, u into the challenge (`BobProofExt`, :414-424).
  template <class Keys>
  static std::pair<BobProof, Batch> generate(Context& ctx, const Keys& alice_ek, const zk_paillier::DLogStatements& stm, const Index& key_idx,
                                             const Index& st_idx, const Batch& a_encrypted, const Batch& mta_encrypted, const Batch& b,
                                             const Batch& beta_prim, const Batch& r, const BobNonces& nn, bool check_) {
    const int B = (int)b.size();
    Dev<uint32_t> ae = up(a_encrypted), me = up(mta_encrypted), db = up(b), bp = up(beta_prim), dr = up(r), al = up(nn.alpha), be = up(nn.beta),
                  ga = up(nn.gamma), rh = up(nn.rho), rp = up(nn.rho_prim), sg = up(nn.sigma), ta = up(nn.tau);
    Dev<uint32_t> t((size_t)B * W_N), z((size_t)B * W_N), e((size_t)B * W_SCALAR), s((size_t)B * W_N), s1((size_t)B * W_S1), s2((size_t)B * W_S2),
                  t1((size_t)B * W_T1), t2((size_t)B * W_S2), u((size_t)B * W_POINT);
    Dev<int32_t> ki(key_idx), si(st_idx);
    const mpe_bob_nonces n{al.get(), be.get(), ga.get(), rh.get(), rp.get(), sg.get(), ta.get()};
    const mpe_bob_proof p{t.get(), z.get(), e.get(), s.get(), s1.get(), s2.get(), t1.get(), t2.get()};
    mpecdsa::check(mpe_bob_generate(ctx.get(), alice_ek.get(), stm.get(), B, ki.get(), si.get(), ae.get(), me.get(), db.get(), bp.get(), dr.get(), &n,
                                    check_ ? 1 : 0, &p, check_ ? u.get() : nullptr, nullptr), "mpe_bob_generate");
    ctx.sync();
    return {BobProof{down(t, W_N), down(z, W_N), down(e, W_SCALAR), down(s, W_N), down(s1, W_S1), down(s2, W_S2), down(t1, W_T1), down(t2, W_S2)},
            check_ ? down(u, W_POINT) : Batch()};
  }
  // `BobProof::verify(&self, a_enc, mta_avc_out, alice_ek, dlog_statement, check: Option<&BobCheck>) -> bool`   range_proofs.rs:321-412
  // (X, u both null: `None`)
  template <class Keys>
  Flags verify(Context& ctx, const Keys& alice_ek, const zk_paillier::DLogStatements& stm, const Index& key_idx, const Index& st_idx,
               const Batch& a_enc, const Batch& mta_avc_out, const Batch* X = nullptr, const Batch* u = nullptr) const {
    const int B = (int)t.size();
    Dev<uint32_t> dt = up(t), dz = up(z), de = up(e), ds = up(s), d1 = up(s1), d2 = up(s2), e1 = up(t1), e2 = up(t2), ae = up(a_enc), me = up(mta_avc_out);
    std::unique_ptr<Dev<uint32_t>> dX, du;
    if (X && u) { dX.reset(new Dev<uint32_t>(X->w)); du.reset(new Dev<uint32_t>(u->w)); }
    Dev<int32_t> ki(key_idx), si(st_idx);
    Dev<uint8_t> ok((size_t)B);
    const mpe_bob_proof p{dt.get(), dz.get(), de.get(), ds.get(), d1.get(), d2.get(), e1.get(), e2.get()};
    mpecdsa::check(mpe_bob_verify(ctx.get(), alice_ek.get(), stm.get(), B, ki.get(), si.get(), ae.get(), me.get(), &p, dX ? dX->get() : nullptr,
                                  du ? du->get() : nullptr, ok.get(), nullptr), "mpe_bob_verify");
    ctx.sync();
    return ok.download();
  }
};

// `BobProofExt{proof, u}`   range_proofs.rs:414-424, 499-534
struct BobProofExt {
  BobProof proof;
  Batch u;
  // `BobProofExt::verify(&self, a_enc, mta_avc_out, alice_ek, dlog_statement, X) -> bool`
  template <class Keys>
  Flags verify(Context& ctx, const Keys& alice_ek, const zk_paillier::DLogStatements& stm, const Index& key_idx, const Index& st_idx,
               const Batch& a_enc, const Batch& mta_avc_out, const Batch& X) const {
    return proof.verify(ctx, alice_ek, stm, key_idx, st_idx, a_enc, mta_avc_out, &X, &u);
  }
};
}  // namespace range_proofs

// `MessageA{c, range_proofs}`   mta/mod.rs:34-38.  range_proofs holds batch * statements items, item-major [exchange][statement]
struct MessageA {
  Batch c;
  range_proofs::AliceProof range_proofs;
  // `MessageA::a_with_predefined_randomness(a, alice_ek, randomness, dlog_statements)`   mta/mod.rs:62-87
  // (alice is the key holder here, as in the protocol: she encrypts her own share)
  static MessageA a_with_predefined_randomness(Context& ctx, const paillier::DecryptionKeys& alice_dk, const zk_paillier::DLogStatements& stm,
                                               const Index& key_idx, const Batch& a, const Batch& randomness, const range_proofs::AliceNonces& nn) {
    const int B = (int)a.size(), P = B * stm.count();
    Dev<uint32_t> da = up(a), dr = up(randomness), al = up(nn.alpha), be = up(nn.beta), ga = up(nn.gamma), rh = up(nn.rho);
    Dev<uint32_t> c((size_t)B * W_NN), z((size_t)P * W_N), e((size_t)P * W_SCALAR), s((size_t)P * W_N), s1((size_t)P * W_S1), s2((size_t)P * W_S2);
    Dev<int32_t> ki(key_idx);
    const mpe_alice_nonces n{al.get(), be.get(), ga.get(), rh.get()};
    const mpe_alice_proof p{z.get(), e.get(), s.get(), s1.get(), s2.get()};
    check(mpe_mta_message_a(ctx.get(), alice_dk.get(), stm.get(), B, ki.get(), da.get(), dr.get(), &n, c.get(), &p, nullptr), "mpe_mta_message_a");
    ctx.sync();
    return MessageA{down(c, W_NN), range_proofs::AliceProof{down(z, W_N), down(e, W_SCALAR), down(s, W_N), down(s1, W_S1), down(s2, W_S2)}};
  }
};

// `MessageB{c, b_proof, beta_tag_proof}`   mta/mod.rs:40-45
struct MessageB {
  Batch c;
  curv::DLogProof b_proof, beta_tag_proof;

  // `MessageB::b_with_predefined_randomness(b, alice_ek, m_a, randomness, beta_tag, dlog_statements)
  //     -> Result<(MessageB, Scalar /* beta */, ..), Error>`   mta/mod.rs:111-158
  // ok[i] == 0 is the reference's Err(InvalidKey): one of m_a's range proofs failed (:119-131).  nonce_b / nonce_bt are the
  // nonces of the two DLogProofs (:147-148).
  template <class Keys>
  static std::tuple<MessageB, Batch, Flags> b_with_predefined_randomness(Context& ctx, const Keys& alice_ek, const zk_paillier::DLogStatements& stm,
                                                                          const Index& key_idx, const Batch& b, const MessageA& m_a,
                                                                          const Batch& randomness, const Batch& beta_tag, const Batch& nonce_b,
                                                                          const Batch& nonce_bt) {
    const int B = (int)b.size();
    Dev<uint32_t> db = up(b), ca = up(m_a.c), z = up(m_a.range_proofs.z), e = up(m_a.range_proofs.e), s = up(m_a.range_proofs.s),
                  s1 = up(m_a.range_proofs.s1), s2 = up(m_a.range_proofs.s2), dr = up(randomness), bt = up(beta_tag), nb = up(nonce_b), nbt = up(nonce_bt);
    Dev<uint32_t> cb((size_t)B * W_NN), beta((size_t)B * W_SCALAR), pk((size_t)B * W_POINT), R((size_t)B * W_POINT), zz((size_t)B * W_SCALAR),
                  tpk((size_t)B * W_POINT), tR((size_t)B * W_POINT), tz((size_t)B * W_SCALAR);
    Dev<int32_t> ki(key_idx);
    Dev<uint8_t> ok((size_t)B);
    const mpe_alice_proof rp{z.get(), e.get(), s.get(), s1.get(), s2.get()};
    const mpe_dlog_proof p1{pk.get(), R.get(), zz.get()}, p2{tpk.get(), tR.get(), tz.get()};
    check(mpe_mta_message_b(ctx.get(), alice_ek.get(), stm.get(), B, ki.get(), db.get(), ca.get(), &rp, dr.get(), bt.get(), nb.get(), nbt.get(),
                            cb.get(), beta.get(), &p1, &p2, ok.get(), nullptr), "mpe_mta_message_b");
    ctx.sync();
    MessageB m{down(cb, W_NN), curv::DLogProof{down(pk, W_POINT), down(R, W_POINT), down(zz, W_SCALAR)},
               curv::DLogProof{down(tpk, W_POINT), down(tR, W_POINT), down(tz, W_SCALAR)}};
    return {std::move(m), down(beta, W_SCALAR), ok.download()};
  }

  // `MessageB::verify_proofs_get_alpha(&self, dk, a) -> Result<(Scalar /* alpha */, BigInt /* alice_share */), Error>`   mta/mod.rs:160-179
  std::tuple<Batch, Batch, Flags> verify_proofs_get_alpha(Context& ctx, const paillier::DecryptionKeys& dk, const Index& key_idx, const Batch& a) const {
    const int B = (int)a.size();
    Dev<uint32_t> cb = up(c), pk = up(b_proof.pk), R = up(b_proof.pk_t_rand_commitment), z = up(b_proof.challenge_response),
                  tpk = up(beta_tag_proof.pk), tR = up(beta_tag_proof.pk_t_rand_commitment), tz = up(beta_tag_proof.challenge_response), da = up(a);
    Dev<uint32_t> alpha((size_t)B * W_SCALAR), share((size_t)B * W_N);
    Dev<int32_t> ki(key_idx);
    Dev<uint8_t> ok((size_t)B);
    const mpe_dlog_proof p1{pk.get(), R.get(), z.get()}, p2{tpk.get(), tR.get(), tz.get()};
    check(mpe_mta_verify_get_alpha(ctx.get(), dk.get(), B, ki.get(), cb.get(), &p1, &p2, da.get(), alpha.get(), share.get(), ok.get(), nullptr),
          "mpe_mta_verify_get_alpha");
    ctx.sync();
    return {down(alpha, W_SCALAR), down(share, W_N), ok.download()};
  }
};
}  // namespace mta

namespace zk_pdl_with_slack {
// `PDLwSlackStatement{ciphertext, ek, Q, G, h1, h2, N_tilde}`   zk_pdl_with_slack/mod.rs:37-46 — ek and (h1, h2, N_tilde) by index
struct PDLwSlackStatement { Batch ciphertext, Q, G; Index ek_idx, st_idx; };
// `PDLwSlackWitness{x, r}`   :48-51
struct PDLwSlackWitness { Batch x, r; };
// what `prove` samples (:73-77): alpha < q^3, beta in [1, N), rho < q N~, gamma < q^3 N~
struct PDLwSlackNonces { Batch alpha, beta, rho, gamma; };

// `PDLwSlackProof{z, u1, u2, u3, s1, s2, s3}`   :56-65
struct PDLwSlackProof {
  Batch z, u1, u2, u3, s1, s2, s3;
  // `PDLwSlackProof::prove(witness, statement)`   :68-125   (the prover holds the key: GG20's phase5_proof_pdl, party_i.rs:691-717)
  template <class Keys>
  static PDLwSlackProof prove(Context& ctx, const Keys& ek, const zk_paillier::DLogStatements& stm, const PDLwSlackWitness& w,
                              const PDLwSlackStatement& st, const PDLwSlackNonces& nn) {
    const int B = (int)w.x.size();
    Dev<uint32_t> c = up(st.ciphertext), Q = up(st.Q), G = up(st.G), x = up(w.x), r = up(w.r), al = up(nn.alpha), be = up(nn.beta), rh = up(nn.rho),
                  ga = up(nn.gamma);
    Dev<uint32_t> z((size_t)B * W_N), u1((size_t)B * W_POINT), u2((size_t)B * W_NN), u3((size_t)B * W_N), s1((size_t)B * W_S1), s2((size_t)B * W_N),
                  s3((size_t)B * W_S2);
    Dev<int32_t> ki(st.ek_idx), si(st.st_idx);
    const mpe_pdl_nonces n{al.get(), be.get(), rh.get(), ga.get()};
    const mpe_pdl_proof p{z.get(), u1.get(), u2.get(), u3.get(), s1.get(), s2.get(), s3.get()};
    check(mpe_pdl_prove(ctx.get(), ek.get(), stm.get(), B, ki.get(), si.get(), c.get(), Q.get(), G.get(), x.get(), r.get(), &n, &p, nullptr),
          "mpe_pdl_prove");
    ctx.sync();
    return PDLwSlackProof{down(z, W_N), down(u1, W_POINT), down(u2, W_NN), down(u3, W_N), down(s1, W_S1), down(s2, W_N), down(s3, W_S2)};
  }
  // `PDLwSlackProof::verify(&self, statement) -> Result<(), ZkPdlWithSlackError>`   :127-179 (ok[i] = 1: Ok(()))
  template <class Keys>
  Flags verify(Context& ctx, const Keys& ek, const zk_paillier::DLogStatements& stm, const PDLwSlackStatement& st) const {
    const int B = (int)z.size();
    Dev<uint32_t> c = up(st.ciphertext), Q = up(st.Q), G = up(st.G), dz = up(z), d1 = up(u1), d2 = up(u2), d3 = up(u3), e1 = up(s1), e2 = up(s2), e3 = up(s3);
    Dev<int32_t> ki(st.ek_idx), si(st.st_idx);
    Dev<uint8_t> ok((size_t)B);
    const mpe_pdl_proof p{dz.get(), d1.get(), d2.get(), d3.get(), e1.get(), e2.get(), e3.get()};
    check(mpe_pdl_verify(ctx.get(), ek.get(), stm.get(), B, ki.get(), si.get(), c.get(), Q.get(), G.get(), &p, ok.get(), nullptr), "mpe_pdl_verify");
    ctx.sync();
    return ok.download();
  }
};
}  // namespace zk_pdl_with_slack

// `Point::generator() * k`, `P * k`   (party_i.rs:546-936)
inline Batch ec_mul_base(Context& ctx, const Batch& k) {
  const int B = (int)k.size();
  Dev<uint32_t> dk = up(k), o((size_t)B * W_POINT);
  check(mpe_ec_mul_base(ctx.get(), B, dk.get(), k.words, o.get(), nullptr), "mpe_ec_mul_base");
  ctx.sync();
  return down(o, W_POINT);
}

// ---- Lindell'17 two-party ECDSA, signing ----------------------------------------------------------------------------------------
namespace two_party_ecdsa {
namespace lindell_2017 {
namespace party_two {
// `PartialSig{c3}`   party_two.rs:383-388
struct PartialSig {
  Batch c3;
  // `PartialSig::compute(ek, encrypted_secret_share, local_share, ephemeral_local_share, ephemeral_other_public_share, message)`
  // party_two.rs:390-423.  ek = party ONE's Paillier key; x2 = party two's share, k2 its ephemeral secret, R1 = party one's ephemeral
  // public share; rho (< q^2) and r (the randomness Paillier::encrypt draws) are what the reference samples.
  template <class Keys>
  static PartialSig compute(Context& ctx, const Keys& ek, const Index& key_idx, const Batch& encrypted_secret_share, const Batch& x2, const Batch& k2,
                            const Batch& R1, const Batch& message, const Batch& rho, const Batch& r) {
    const int B = (int)x2.size();
    Dev<uint32_t> ck = up(encrypted_secret_share), dx = up(x2), dk = up(k2), dR = up(R1), dm = up(message), drho = up(rho), dr = up(r), c3((size_t)B * W_NN);
    Dev<int32_t> ki(key_idx);
    check(mpe_lindell_partial_sig(ctx.get(), ek.get(), B, ki.get(), ck.get(), dx.get(), dk.get(), dR.get(), dm.get(), drho.get(), dr.get(), c3.get(),
                                  nullptr), "mpe_lindell_partial_sig");
    ctx.sync();
    return PartialSig{down(c3, W_NN)};
  }
};
}  // namespace party_two
namespace party_one {
// `SignatureRecid{s, r, recid}`   party_one.rs:106-111
struct SignatureRecid {
  Batch r, s;
  std::vector<int32_t> recid;
  // `Signature::compute_with_recid(party_one_private, partial_sig_c3, ephemeral_local_share, ephemeral_other_public_share)`
  // party_one.rs:519-565 (`Signature::compute`, :486-517, is the same without recid).  dk = party one's key; k1 its ephemeral secret,
  // R2 = party two's ephemeral public share.  s is low: min(s, q - s).
  static SignatureRecid compute_with_recid(Context& ctx, const paillier::DecryptionKeys& dk, const Index& key_idx, const Batch& partial_sig_c3,
                                           const Batch& k1, const Batch& R2) {
    const int B = (int)k1.size();
    Dev<uint32_t> c3 = up(partial_sig_c3), dk1 = up(k1), dR = up(R2), r((size_t)B * W_SCALAR), s((size_t)B * W_SCALAR);
    Dev<int32_t> ki(key_idx), rec((size_t)B);
    check(mpe_lindell_sign(ctx.get(), dk.get(), B, ki.get(), c3.get(), dk1.get(), dR.get(), r.get(), s.get(), rec.get(), nullptr), "mpe_lindell_sign");
    ctx.sync();
    return SignatureRecid{down(r, W_SCALAR), down(s, W_SCALAR), rec.download()};
  }
};
}  // namespace party_one
}  // namespace lindell_2017
}  // namespace two_party_ecdsa

// ---- GG20 signing, the state-machine surface ------------------------------------------------------------------------------------
namespace gg_2020 {

// `LocalKey<Secp256k1>` as keygen leaves it with party `i` (state_machine/keygen/rounds.rs:311-322), in interface words
struct LocalKey {
  uint16_t i = 0, t = 0, n = 0;                 // `i` in [1, n]
  Batch paillier_key_vec;                       // [n][64]  EncryptionKey.n of every party
  Batch n_tilde_vec, h1_vec, h2_vec;            // [n][64]  `h1_h2_n_tilde_vec` (DLogStatement{N, g, ni})
  Batch y_sum_s;                                // [1][16]  the joint public key
  Batch pk_vec;                                 // [n][16]  X_j = x_j G
  Batch x_i;                                    // [1][8]   `keys_linear.x_i`
  Batch p, q;                                   // [1][32]  `paillier_dk`
};

// Every value ONE party samples from OsRng while signing, for `batch` sessions: the C-ABI's mpe_gg20_nonces with one local party
// (leading dimension [batch]; then statement st (n), peer slot jj (S-1), MessageB variant v (2) — see include/mpecdsa_hip.h).
struct SignNonces {
  Batch k, gamma, blind, r_a, al_alpha, al_beta, al_gamma, al_rho, mb_beta_tag, mb_r, mb_nonce_b, mb_nonce_bt, l, ped_s1, ped_s2, pdl_alpha,
      pdl_beta, pdl_rho, pdl_gamma, heg_s1, heg_s2;
};

namespace state_machine {
namespace sign {

// `sign::Error` (sign.rs:520-560).  A failed CHECK of the protocol is not an exception here: the batch goes on and the session's
// status says which check failed (`ProceedRound(rounds::Error)` per session: OfflineStage::status(), mpecdsa_hip.h lists the codes).
struct Error : std::runtime_error {
  enum Kind { TooFewParties, InvalidPartyIndex, InvalidSl, ReceivedOutOfOrderMessage, HandleMessage, DoublePickOutput, OfflineStageReused };
  Kind kind;
  Error(Kind k, const std::string& what) : std::runtime_error(what), kind(k) {}
};

// `Msg<OfflineProtocolMessage>` (sign.rs:340-366): `sender` = position in s_l (1-based), `round` = the round that consumes it
// (M1..M6), `body` = one fixed-layout record per session.  Every message travels as a broadcast and the receiver picks what is
// addressed to it, as the relay of examples/gg20_sm_manager.rs does.
struct Msg {
  uint16_t sender = 0, round = 0;
  Batch body;
};

namespace detail {
struct Party {
  Context& ctx;
  mpe_gg20_keys* keys = nullptr;
  mpe_gg20_session* sess = nullptr;
  int batch = 0, S = 0, n = 0, me = 0;                             // me: 0-based position in s_l
  Batch y_sum_s;
  std::vector<std::unique_ptr<Dev<uint32_t>>> sampled;              // the nonce arrays: valid until round 5 is queued (mpecdsa_hip.h)
  std::map<int, Batch> mine;                                        // what this party sent, by consuming round (1..6; 8 = partial signature)
  bool signed_once = false;
  explicit Party(Context& c) : ctx(c) {}
  ~Party() {
    if (sess) (void)mpe_gg20_session_destroy(sess, nullptr);        // zeroes k_i, gamma_i, w_i, sigma_i
    if (keys) (void)mpe_gg20_keys_destroy(keys);
  }
  Party(const Party&) = delete;
  int words(int emitting_round) const { return mpe_gg20_msg_words(S, n, emitting_round); }
  // the records of ALL senders for one round, sender-major: own block at `me`
  Dev<uint32_t> slab(const Batch& own, const std::map<uint16_t, Batch>& peers) const {
    const size_t blk = own.w.size();
    std::vector<uint32_t> h((size_t)S * blk);
    for (int j = 0; j < S; ++j) {
      const Batch& b = j == me ? own : peers.at((uint16_t)(j + 1));
      std::copy(b.w.begin(), b.w.end(), h.begin() + (size_t)j * blk);
    }
    return Dev<uint32_t>(h);
  }
};
}  // namespace detail

// `CompletedOfflineStage` (rounds.rs:560-570): everything `SignManual` needs; may be copied like the reference's (it is `Clone`),
// but signs ONE message — a second SignManual::new on any copy throws (two signatures with one k_i leak the key share).
class CompletedOfflineStage {
 public:
  const Batch& public_key() const { return p_->y_sum_s; }           // `public_key()` (rounds.rs:572-574)
 private:
  friend class OfflineStage;
  friend class SignManual;
  explicit CompletedOfflineStage(std::shared_ptr<detail::Party> p) : p_(std::move(p)) {}
  std::shared_ptr<detail::Party> p_;
};

// `OfflineStage` (sign.rs:43-330): one party of the offline stage, for `batch` concurrent sessions with the same signer set.
class OfflineStage {
 public:
  // `OfflineStage::new(i, s_l, local_key)` (sign.rs:77-121): i in [1, |s_l|], s_l = keygen indices of the signers.  This layer
  // takes s_l in ascending order (the order the C-ABI's signer ordinals use; every test of the reference does).
  OfflineStage(Context& ctx, uint16_t i, const std::vector<uint16_t>& s_l, const LocalKey& local_key, int batch, const SignNonces& sampled)
      : p_(std::make_shared<detail::Party>(ctx)) {
    if (s_l.size() < 2) throw Error(Error::TooFewParties, "at least 2 parties are required for signing");
    if (i == 0 || i > s_l.size()) throw Error(Error::InvalidPartyIndex, "party index is not in range [1; n]");
    for (size_t a = 0; a < s_l.size(); ++a) {
      if (s_l[a] == 0 || s_l[a] > local_key.n) throw Error(Error::InvalidSl, "invalid s_l");
      if (a && s_l[a] <= s_l[a - 1]) throw Error(Error::InvalidSl, s_l[a] == s_l[a - 1] ? "invalid s_l" : "invalid s_l (this layer: ascending order)");
    }
    if (s_l[i - 1] != local_key.i) throw Error(Error::InvalidSl, "invalid s_l (s_l[i] is not this key's keygen index)");
    detail::Party& P = *p_;
    P.batch = batch; P.S = (int)s_l.size(); P.n = local_key.n; P.me = i - 1; P.y_sum_s = local_key.y_sum_s;
    std::vector<int32_t> signers(s_l.begin(), s_l.end());
    for (auto& v : signers) v -= 1;
    const int32_t own = local_key.i - 1, local = P.me;
    {
      Dev<uint32_t> x = up_secret(local_key.x_i), dp = up_secret(local_key.p), dq = up_secret(local_key.q), N = up(local_key.paillier_key_vec), Nt = up(local_key.n_tilde_vec),
                    h1 = up(local_key.h1_vec), h2 = up(local_key.h2_vec), y = up(local_key.y_sum_s), X = up(local_key.pk_vec);
      check(mpe_gg20_keys_create(ctx.get(), local_key.t, local_key.n, P.S, signers.data(), 1, 1, &own, x.get(), dp.get(), dq.get(), N.get(), Nt.get(),
                                 h1.get(), h2.get(), y.get(), X.get(), &P.keys, nullptr), "mpe_gg20_keys_create");
      ctx.sync();
    }
    const Batch* f[] = {&sampled.k, &sampled.gamma, &sampled.blind, &sampled.r_a, &sampled.al_alpha, &sampled.al_beta, &sampled.al_gamma, &sampled.al_rho,
                        &sampled.mb_beta_tag, &sampled.mb_r, &sampled.mb_nonce_b, &sampled.mb_nonce_bt, &sampled.l, &sampled.ped_s1, &sampled.ped_s2,
                        &sampled.pdl_alpha, &sampled.pdl_beta, &sampled.pdl_rho, &sampled.pdl_gamma, &sampled.heg_s1, &sampled.heg_s2};
    for (const Batch* b : f) P.sampled.emplace_back(new Dev<uint32_t>(b->w, true));       // wiped when released (after round 5 / with the party)
    auto d = [&](int k) { return (const uint32_t*)P.sampled[(size_t)k]->get(); };
    const mpe_gg20_nonces nn{d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11), d(12), d(13), d(14), d(15), d(16), d(17), d(18),
                             d(19), d(20), nullptr};
    check(mpe_gg20_session_create(ctx.get(), P.keys, batch, 1, &local, nullptr, &nn, 0, &P.sess, nullptr), "mpe_gg20_session_create");
  }

  uint16_t current_round() const { return round_; }                                   // 0..6, 7 once finished (sign.rs:300-312)
  uint16_t party_ind() const { return (uint16_t)(p_->me + 1); }
  uint16_t parties() const { return (uint16_t)p_->S; }
  bool is_finished() const { return round_ == 7; }
  std::vector<Msg>& message_queue() { return queue_; }

  // `handle_incoming(msg)` (sign.rs:246-297): stores a peer's message for the round it belongs to — early messages wait in
  // their round's store, a message for a round that is over is `ReceivedOutOfOrderMessage`
  void handle_incoming(const Msg& m) {
    if (m.round < 1 || m.round > 6 || m.round < round_)                               // the store of a finished round is gone
      throw Error(Error::ReceivedOutOfOrderMessage, "didn't expect to receive message from round " + std::to_string(m.round) + " (being at round " +
                                                        std::to_string(round_) + ")");
    if (m.sender == 0 || m.sender > p_->S || m.sender == party_ind()) throw Error(Error::HandleMessage, "received message didn't pass pre-validation: sender");
    if ((int)m.body.size() != p_->batch || m.body.words != p_->words(m.round - 1))
      throw Error(Error::HandleMessage, "received message didn't pass pre-validation: record layout");
    auto& st = store_[m.round];
    if (st.count(m.sender)) throw Error(Error::HandleMessage, "received message didn't pass pre-validation: message overwrite");
    st[m.sender] = m.body;
  }

  // `wants_to_proceed()` (sign.rs:299-311): round 0 always; round r once the S - 1 peers' messages for r are in
  bool wants_to_proceed() const {
    if (round_ == 7) return false;
    if (round_ == 0) return true;
    auto it = store_.find(round_);
    return it != store_.end() && (int)it->second.size() == p_->S - 1;
  }

  // `proceed()` (sign.rs:313-316): RoundN::proceed for every session (rounds.rs:68,122,234,347,431,525,612); no-op when the
  // round's messages are not all in.  The outgoing message is appended to message_queue().
  void proceed() {
    if (!wants_to_proceed()) return;
    detail::Party& P = *p_;
    const int r = round_;
    if (r == 0) {
      Dev<uint32_t> out((size_t)P.batch * P.words(0));
      check(mpe_gg20_round0(P.sess, out.get(), nullptr), "mpe_gg20_round0");
      emit(1, down(out, P.words(0)));
    } else {
      Dev<uint32_t> in = P.slab(P.mine.at(r), store_.at((uint16_t)r));
      if (r < 6) {
        Dev<uint32_t> out((size_t)P.batch * P.words(r));
        using Fn = int (*)(mpe_gg20_session*, const uint32_t*, const int64_t*, uint32_t*, void*);
        static const Fn fn[] = {nullptr, mpe_gg20_round1, mpe_gg20_round2, mpe_gg20_round3, mpe_gg20_round4, mpe_gg20_round5};
        check(fn[r](P.sess, in.get(), nullptr, out.get(), nullptr), "mpe_gg20_roundN");
        emit((uint16_t)(r + 1), down(out, P.words(r)));
      } else {
        check(mpe_gg20_round6(P.sess, in.get(), nullptr, nullptr), "mpe_gg20_round6");
        P.ctx.sync();
      }
      store_.erase((uint16_t)r);
    }
    if (r == 5) P.sampled.clear();                                                    // round 5 has run: the sampled values are no longer read
    round_ = (uint16_t)(r + 1);
  }

  // per-session status so far (0 = every check passed): the reference's Err(ProceedRound(..)) of that session
  std::vector<int32_t> status() const {
    Dev<int32_t> st((size_t)p_->batch);
    check(mpe_gg20_session_result(p_->sess, st.get(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "mpe_gg20_session_result");
    p_->ctx.sync();
    return st.download();
  }

  // `pick_output()` (sign.rs:318-330): None while the stage is running, the output once, DoublePickOutput afterwards
  std::optional<CompletedOfflineStage> pick_output() {
    if (!is_finished()) return std::nullopt;
    if (picked_) throw Error(Error::DoublePickOutput, "pick_output called twice");
    picked_ = true;
    return CompletedOfflineStage(p_);
  }

 private:
  void emit(uint16_t consuming_round, Batch body) {
    p_->ctx.sync();
    p_->mine[consuming_round] = body;
    queue_.push_back(Msg{party_ind(), consuming_round, std::move(body)});
  }
  std::shared_ptr<detail::Party> p_;
  std::map<uint16_t, std::map<uint16_t, Batch>> store_;              // msgs1..msgs6 (sign.rs:52-57)
  std::vector<Msg> queue_;
  uint16_t round_ = 0;
  bool picked_ = false;
};

// `PartialSignature` (rounds.rs:594) and `SignatureRecid{r, s, recid}` (party_i.rs:131-135), one per session; status[b] != 0:
// the reference's Err(CompleteSigning(..)) for that session (701: the assembled signature does not verify)
struct PartialSignature { Batch s_i; };
struct SignatureRecid {
  Batch r, s;
  std::vector<int32_t> recid, status;
  std::vector<uint32_t> bad_actors;
};

// `SignManual` (sign.rs:540-646)
class SignManual {
 public:
  // `SignManual::new(message, completed_offline_stage) -> (SignManual, PartialSignature)` (sign.rs:551-558, Round7::new
  // rounds.rs:612-660); message [batch][8]: the BigInt to sign, reduced mod q like Scalar::from (party_i.rs:857)
  static std::pair<SignManual, PartialSignature> new_(const Batch& message, const CompletedOfflineStage& completed_offline_stage) {
    std::shared_ptr<detail::Party> p = completed_offline_stage.p_;
    if (p->signed_once) throw Error(Error::OfflineStageReused, "this offline stage has already signed a message");
    if ((int)message.size() != p->batch || message.words != W_SCALAR) throw mpecdsa::Error("SignManual::new: message layout", MPE_E_ARG);
    p->signed_once = true;
    Dev<uint32_t> m = up(message), out((size_t)p->batch * W_SCALAR);
    check(mpe_gg20_round7(p->sess, m.get(), out.get(), nullptr), "mpe_gg20_round7");
    p->ctx.sync();
    PartialSignature ps{down(out, W_SCALAR)};
    p->mine[8] = ps.s_i;
    return {SignManual(std::move(p)), std::move(ps)};
  }
  // `complete(self, sigs)` (sign.rs:562-568): sigs = the partial signatures of the OTHER parties, in any order
  SignatureRecid complete(const std::vector<PartialSignature>& sigs) {
    detail::Party& P = *p_;
    if ((int)sigs.size() != P.S - 1) throw mpecdsa::Error("SignManual::complete: expected the other parties' partial signatures", MPE_E_ARG);
    std::map<uint16_t, Batch> peers;
    size_t k = 0;
    for (int j = 0; j < P.S; ++j)
      if (j != P.me) peers[(uint16_t)(j + 1)] = sigs[k++].s_i;
    Dev<uint32_t> in = P.slab(P.mine.at(8), peers);
    check(mpe_gg20_complete(P.sess, in.get(), nullptr, nullptr), "mpe_gg20_complete");
    const size_t B = (size_t)P.batch;
    Dev<int32_t> st(B), rec(B);
    Dev<uint32_t> bad(B), r(B * W_SCALAR), s(B * W_SCALAR);
    check(mpe_gg20_session_result(P.sess, st.get(), bad.get(), r.get(), s.get(), rec.get(), nullptr, nullptr), "mpe_gg20_session_result");
    P.ctx.sync();
    return SignatureRecid{down(r, W_SCALAR), down(s, W_SCALAR), rec.download(), st.download(), bad.download()};
  }
 private:
  explicit SignManual(std::shared_ptr<detail::Party> p) : p_(std::move(p)) {}
  std::shared_ptr<detail::Party> p_;
};

}  // namespace sign
}  // namespace state_machine
// ---- party-sharded signing across GPUs: the round fan-out behind the C-ABI (mpe_comm_*, RCCL over xGMI) -----------------------------
// In the reference a party broadcasts every message — P2P ones included — to the room and every client filters what is addressed
// to it (examples/gg20_sm_client.rs:35-40); the state machine consumes them round by round (state_machine/sign.rs:252-438).  Here one
// process per GPU hosts some (session block, party) pairs; a round's records are written by mpe_gg20_roundN straight into this rank's
// rows of a gather buffer, ONE ncclAllGather per round (mpe_gg20_round_exchange) delivers everybody's rows, and the next round reads
// the gathered buffer in place through h_in_off (mpe_gg20_shard_in_off) — no host copy of any message.
namespace sharded {

// the communicator: rank 0 makes the id, the HOST carries its bytes to the other ranks (a file, a socket, MPI, the relay)
class Comm {
 public:
  static std::vector<uint8_t> unique_id() {
    std::vector<uint8_t> id(MPE_COMM_ID_BYTES);
    check(mpe_comm_unique_id(id.data()), "mpe_comm_unique_id");
    return id;
  }
  Comm(Context& ctx, const std::vector<uint8_t>& id, int rank, int world) {
    if (id.size() != MPE_COMM_ID_BYTES) throw Error("Comm: the id is MPE_COMM_ID_BYTES bytes", MPE_E_ARG);
    check(mpe_comm_create(ctx.get(), id.data(), rank, world, &h_), "mpe_comm_create");
  }
  ~Comm() { if (h_) (void)mpe_comm_destroy(h_); }
  Comm(const Comm&) = delete;
  mpe_comm* get() const { return h_; }
  int rank() const { return mpe_comm_rank(h_); }
  int world() const { return mpe_comm_world(h_); }
  // the gather layout on the real communicator, before any signing work; returns 0 (in place) / 1 (copy form)
  int layout_self_test(int rows_per_rank) {
    int mode = -1, ok = 0;
    check(mpe_comm_layout_self_test(h_, rows_per_rank, &mode, &ok, nullptr), "mpe_comm_layout_self_test");
    return mode;
  }
 private:
  mpe_comm* h_ = nullptr;
};

// what one hosted (block, party) pair needs: the party's LocalKey (its own secrets only) and the values it samples for its block
struct Hosted {
  int block = 0, party = 0;                      // party: 0-based position in s_l
  LocalKey local_key;
  SignNonces sampled;
  Batch message;                                 // [batch][8] the block's messages
};
struct PairResult {
  int block = 0, party = 0;
  Batch r, s;
  std::vector<int32_t> recid, status;
  std::vector<uint32_t> bad_actors;
};

class PartySharded {
 public:
  // placement: MPE_PLACE_PARTY / MPE_PLACE_ROTATED (mpecdsa_hip.h); `hosted`: exactly the pairs mpe_gg20_shard_where puts on this rank
  PartySharded(Context& ctx, Comm& comm, int placement, const std::vector<uint16_t>& s_l, int batch, std::vector<Hosted> hosted)
      : ctx_(ctx), comm_(comm), placement_(placement), S_((int)s_l.size()), batch_(batch) {
    const int world = comm.world(), rank = comm.rank();
    per_rank_ = mpe_gg20_shard_per_rank(placement, S_, world);
    if (per_rank_ < 1 || (int)hosted.size() != per_rank_) throw Error("PartySharded: this rank hosts mpe_gg20_shard_per_rank pairs", MPE_E_ARG);
    std::vector<int32_t> signers(s_l.begin(), s_l.end());
    for (auto& v : signers) v -= 1;
    for (Hosted& h : hosted) {
      int r = -1, slot = -1;
      check(mpe_gg20_shard_where(placement, S_, world, h.block, h.party, &r, &slot), "mpe_gg20_shard_where");
      if (r != rank) throw Error("PartySharded: a pair that lives on another rank", MPE_E_ARG);
      std::unique_ptr<Pair> P(new Pair(ctx));
      P->block = h.block; P->party = h.party; P->slot = slot; P->message = std::move(h.message);
      n_ = h.local_key.n;
      const LocalKey& k = h.local_key;
      const int32_t own = k.i - 1, local = h.party;
      {
        Dev<uint32_t> x = up_secret(k.x_i), dp = up_secret(k.p), dq = up_secret(k.q), N = up(k.paillier_key_vec), Nt = up(k.n_tilde_vec), h1 = up(k.h1_vec), h2 = up(k.h2_vec),
                      y = up(k.y_sum_s), X = up(k.pk_vec);
        check(mpe_gg20_keys_create(ctx.get(), k.t, k.n, S_, signers.data(), 1, 1, &own, x.get(), dp.get(), dq.get(), N.get(), Nt.get(), h1.get(), h2.get(),
                                   y.get(), X.get(), &P->keys, nullptr), "mpe_gg20_keys_create");
        ctx.sync();
      }
      const SignNonces& z = h.sampled;
      const Batch* f[] = {&z.k, &z.gamma, &z.blind, &z.r_a, &z.al_alpha, &z.al_beta, &z.al_gamma, &z.al_rho, &z.mb_beta_tag, &z.mb_r, &z.mb_nonce_b,
                          &z.mb_nonce_bt, &z.l, &z.ped_s1, &z.ped_s2, &z.pdl_alpha, &z.pdl_beta, &z.pdl_rho, &z.pdl_gamma, &z.heg_s1, &z.heg_s2};
      for (const Batch* b : f) P->sampled.emplace_back(new Dev<uint32_t>(b->w, true));
      auto d = [&](int q) { return (const uint32_t*)P->sampled[(size_t)q]->get(); };
      const mpe_gg20_nonces nn{d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11), d(12), d(13), d(14), d(15), d(16), d(17), d(18),
                               d(19), d(20), nullptr};
      check(mpe_gg20_session_create(ctx.get(), P->keys, batch, 1, &local, nullptr, &nn, 0, &P->sess, nullptr), "mpe_gg20_session_create");
      check(mpe_gg20_shard_in_off(placement, S_, world, batch, h.block, P->in_off), "mpe_gg20_shard_in_off");
      pairs_.push_back(std::move(P));
    }
    int maxw = 0;
    for (int r = 0; r < 8; ++r) maxw = std::max(maxw, mpe_gg20_msg_words(S_, n_, r));
    const size_t words = (size_t)world * per_rank_ * batch * maxw;
    for (auto& b : buf_) b.reset(new Dev<uint32_t>(words));
    gather_mode_ = comm.layout_self_test(per_rank_);
  }

  int gather_mode() const { return gather_mode_; }
  // bytes every rank receives per round (all ranks' rows), by emitting round
  std::map<int, size_t> bytes_per_round() const {
    std::map<int, size_t> m;
    for (int r : {0, 1, 2, 3, 4, 5, 7}) m[r] = (size_t)comm_.world() * per_rank_ * batch_ * mpe_gg20_msg_words(S_, n_, r) * 4;
    return m;
  }

  // Round0..Round7 + SignManual::complete for every hosted pair, one all-gather after every emitting round
  std::vector<PairResult> run() {
    const uint32_t* gathered = nullptr;
    int q = 0;
    for (int rnd = 0; rnd <= 8; ++rnd) {
      const int W = (rnd == 6 || rnd == 8) ? 0 : mpe_gg20_msg_words(S_, n_, rnd);
      uint32_t* cur = W ? buf_[q & 1]->get() : nullptr;
      for (auto& Pp : pairs_) {
        Pair& P = *Pp;
        uint32_t* mine = W ? cur + ((size_t)comm_.rank() * per_rank_ + P.slot) * (size_t)batch_ * W : nullptr;
        int rc = MPE_OK;
        switch (rnd) {
          case 0: rc = mpe_gg20_round0(P.sess, mine, nullptr); break;
          case 1: rc = mpe_gg20_round1(P.sess, gathered, P.in_off, mine, nullptr); break;
          case 2: rc = mpe_gg20_round2(P.sess, gathered, P.in_off, mine, nullptr); break;
          case 3: rc = mpe_gg20_round3(P.sess, gathered, P.in_off, mine, nullptr); break;
          case 4: rc = mpe_gg20_round4(P.sess, gathered, P.in_off, mine, nullptr); break;
          case 5: rc = mpe_gg20_round5(P.sess, gathered, P.in_off, mine, nullptr); break;
          case 6: rc = mpe_gg20_round6(P.sess, gathered, P.in_off, nullptr); break;
          case 7: { Dev<uint32_t> m = up(P.message); rc = mpe_gg20_round7(P.sess, m.get(), mine, nullptr); ctx_.sync(); break; }
          default: rc = mpe_gg20_complete(P.sess, gathered, P.in_off, nullptr); break;
        }
        check(rc, "mpe_gg20_roundN");
      }
      if (W) {
        check(mpe_gg20_round_exchange(comm_.get(), S_, n_, rnd, per_rank_, batch_, cur, nullptr), "mpe_gg20_round_exchange");
        gathered = cur;
        ++q;
      }
    }
    std::vector<PairResult> out;
    const size_t B = (size_t)batch_;
    for (auto& Pp : pairs_) {
      Dev<int32_t> st(B), rec(B);
      Dev<uint32_t> bad(B), r(B * W_SCALAR), s(B * W_SCALAR);
      check(mpe_gg20_session_result(Pp->sess, st.get(), bad.get(), r.get(), s.get(), rec.get(), nullptr, nullptr), "mpe_gg20_session_result");
      ctx_.sync();
      out.push_back(PairResult{Pp->block, Pp->party, down(r, W_SCALAR), down(s, W_SCALAR), rec.download(), st.download(), bad.download()});
    }
    return out;
  }

 private:
  struct Pair {
    Context& ctx;
    int block = 0, party = 0, slot = 0;
    mpe_gg20_keys* keys = nullptr;
    mpe_gg20_session* sess = nullptr;
    int64_t in_off[8] = {0};
    Batch message;
    std::vector<std::unique_ptr<Dev<uint32_t>>> sampled;
    explicit Pair(Context& c) : ctx(c) {}
    ~Pair() {
      if (sess) (void)mpe_gg20_session_destroy(sess, nullptr);
      if (keys) (void)mpe_gg20_keys_destroy(keys);
    }
  };
  Context& ctx_;
  Comm& comm_;
  int placement_, S_, batch_, n_ = 0, per_rank_ = 0, gather_mode_ = -1;
  std::vector<std::unique_ptr<Pair>> pairs_;
  std::unique_ptr<Dev<uint32_t>> buf_[2];
};

}  // namespace sharded
}  // namespace gg_2020

}  // namespace mpecdsa
