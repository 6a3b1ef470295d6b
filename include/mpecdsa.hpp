// mpecdsa.hpp — the host side ABOVE the C-ABI, in C++17, with the reference's names.
//
// The reference (ZenGo-X/multi-party-ecdsa v0.8.1) is compiled Rust; its toolchain is absent from the build image, so the layer a
// Rust maintainer would write over the `extern "C"` bindings exists here as a header-only C++ mirror of the
// reference's own call surface for the hot path — same type and method names, same argument meaning, same error behaviour — over
// `include/mpecdsa_hip.h`.  Every method is the BATCHED form of the reference call it is named after and cites it:
//
//   paillier::Paillier::{encrypt_with_chosen_randomness, decrypt, add, mul}     src/utilities/mta/mod.rs:22-24,68-75,133-145,165
//   mta::range_proofs::AliceProof::{generate, verify}                            src/utilities/mta/range_proofs.rs:105-193
//   mta::{MessageA::a_with_predefined_randomness, MessageB::b_with_predefined_randomness,
//         MessageB::verify_proofs_get_alpha}                                     src/utilities/mta/mod.rs:62-179
//   zk_pdl_with_slack::PDLwSlackProof::{prove, verify}                           src/utilities/zk_pdl_with_slack/mod.rs:68-179
//   curv DLogProof::{prove, verify}                                              mta/mod.rs:147-148,170-171
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

#include <cstdint>
#include <cstring>
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
enum : int { W_SCALAR = 8, W_POINT = 16, W_Q3 = 24, W_S1 = 25, W_PRIME = 32, W_N = 64, W_RHO = 72, W_GAMMA = 88, W_S2 = 89, W_NN = 128 };

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

// device buffer for the duration of a call
template <class T>
class Dev {
 public:
  explicit Dev(size_t n) : n_(n) { check_hip(hipMalloc((void**)&p_, (n ? n : 1) * sizeof(T)), "hipMalloc"); }
  explicit Dev(const std::vector<T>& h) : Dev(h.size()) {
    if (!h.empty()) check_hip(hipMemcpy(p_, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
  }
  ~Dev() { if (p_) (void)hipFree(p_); }
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
};
inline Dev<uint32_t> up(const Batch& b) { return Dev<uint32_t>(b.w); }
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

}  // namespace mpecdsa
