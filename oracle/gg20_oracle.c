/* gg20_oracle.c — CPU ORACLE (test infrastructure, NOT product code): GG20 signing as the reference structures it,
 * ONE PARTY AT A TIME.  Each `gg_roundN` below is the restatement of `RoundN::proceed`
 *   src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign/rounds.rs:68,122,234,347,431,525,612,672
 * a pure function of (this party's state, the messages of the previous round) -> (next state, outgoing message),
 * over SignKeys / LocalSignature (src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:526-936) and
 * MessageA / MessageB (src/utilities/mta/mod.rs:52-179); for the un-vendored curv sigma proofs see SURVEY.md
 * App. A.3 (PedersenProof, HomoELGamalProof, HashCommitment, VerifiableSS::map_share_to_new_params).
 * Every value the reference samples is an input.  PARITY UNPINNED — see mpe_oracle.h.
 *
 * Messages are fixed-size records of 32-bit words, one per (sender, session); the layouts are those of
 * include/mpecdsa_hip.h ("GG20 round messages"), so a test can compare whole message slabs byte for byte:
 *   M0 Round0 out  (MessageA, SignBroadcastPhase1)     (n+1) sub-records of 256: AliceProof st | {c, com}
 *   M1 Round1 out  (GammaI, WI) for every peer          2(S-1) sub-records of 208: MessageB (jj, v)
 *   M2 Round2 out  (DeltaI, TI, TIProof)                96
 *   M3 Round3 out  SignDecommitPhase1                   24
 *   M4 Round4 out  (RDash, Vec<PDLwSlackProof>)         S sub-records of 450: proof for peer jj | R_dash
 *   M5 Round5 out  (SI, HEGProof)                       64
 *   M6 Round7 out  PartialSignature                     8
 * P2P messages travel like broadcast ones and are filtered by the receiver, as in the reference's relay
 * (examples/gg20_sm_client.rs:35-40).
 *
 * Status of a party in a session: 0, or 100*round + detail of its FIRST failed check, in the order the reference
 * evaluates them (sticky):
 *   101 MessageB::b -> InvalidKey (a peer's range proof)              rounds.rs:151-175  (Error::Round1)
 *   201 verify_proofs_get_alpha                                        rounds.rs:264-279  (Error::Round3 [sic])
 *   202 assert_eq!(m_b.b_proof.pk, g_w_vec[ind])                       rounds.rs:281      (panic)
 *   303 assert_eq!(t_vec[i], t_proof_vec[i].com)                       rounds.rs:365-367  (panic)
 *   301 phase3_reconstruct_delta: sum not invertible                   party_i.rs:635-640 (unwrap panic)
 *   302 PedersenProof::verify                                          rounds.rs:371-378  (Error::Round3)
 *   401 phase4 "bad gamma_i decommit", bad_actors = the peers          party_i.rs:642-687 (Error::Round5 [sic])
 *   501 "Bad PDLwSlack proof", bad_actors = the first failing prover   party_i.rs:719-766, rounds.rs:546-558
 *   502 phase5_check_R_dash_sum                                        party_i.rs:768-776
 *   601 phase6_verify_proof, bad_actors = every failing prover         party_i.rs:801-833 (Error::Round6VerifyProof)
 *   602 phase6_check_S_i_sum                                           party_i.rs:835-848 (Error::Round6CheckSig)
 *   701 output_signature: verify failed                                party_i.rs:873-910 (Error::Round7)
 *   r90 (290 .. 690) a point of a message round r reads is malformed (off the curve, coordinate >= p, infinity): curv's
 *       deserialisation rejects the message before RoundN::proceed sees it; bad_actors = the senders
 * bad_actors is a bit mask over signer ordinals (positions in s_l), as the reference's Vec<usize>.
 *
 * Compiled into libmpe_oracle.so by #include from mpe_oracle.c (shares its static helpers). */

#define GG_MAXS 8
#define GG_MAXN 8
#define GG_SUB0 256
#define GG_SUB1 208
#define GG_W2 96
#define GG_W3 24
#define GG_SUB4 450
#define GG_W5 64
#define GG_W6 8

int orc_gg20_msg_words(int S, int n, int round) {
  switch (round) {
    case 0: return GG_SUB0 * (n + 1);
    case 1: return GG_SUB1 * 2 * (S - 1);
    case 2: return GG_W2;
    case 3: return GG_W3;
    case 4: return GG_SUB4 * S;
    case 5: return GG_W5;
    case 7: return GG_W6;
    default: return 0;
  }
}

static void sc_mod(mpz_t r) { mpz_mod(r, r, EC_Q); }
static int ind_of(int i, int jj) { return jj < i ? jj : jj + 1; }          /* rounds.rs:149,261,464 */
static int jme_of(int i, int ind) { return i < ind ? i : i - 1; }          /* my slot in the P2P messages of `ind` */

/* HashCommitment::create_commitment_with_user_defined_randomness(m, r) = SHA256(bytes(m) || bytes(r)) as BigInt
 * with m = BigInt::from_bytes(point.to_bytes(true))   (party_i.rs:577-580,654-659) */
static void hash_commit_point(mpz_t out, const pt_t* P, const mpz_t blind) {
  sha_t sh; sha_init(&sh);
  mpz_t m; mpz_init(m);
  pt_as_bigint(m, P);
  chain_bigint(&sh, m); chain_bigint(&sh, blind);
  result_bigint(&sh, out);
  mpz_clear(m);
}
/* pts: the canonical list of the proof — 5 = PedersenProof (g, h, com, a1, a2), 7 = HomoELGamalProof (T, A3, G, H, Y, D, E),
 * 6 = ECDDHProof (g1, h1, g2, h2, a1, a2); hashed in the order ORC_ENC names (identity by default) */
static void hash_points_scalar(mpz_t out, const pt_t** pts, int n) {
  sha_t sh; sha_init(&sh);
  const uint8_t* ord = n == 5 ? ORC_ENC.ord_pedersen : (n == 7 ? ORC_ENC.ord_heg : ORC_ENC.ord_ecddh);
  for (int i = 0; i < n; ++i) chain_point(&sh, pts[ord[i] % n]);
  result_bigint(&sh, out);
  sc_mod(out);
}
static void pt_h2(pt_t* r) {
  ec_setup();
  mpz_set_str(r->x, "08d13221e3a7326a34dd45214ba80116dd142e4b5ff3ce66a8dc7bfa0378b795", 16);
  mpz_set_str(r->y, "5d41ac1477614b5c0848d50dbd565ea2807bcba1df0df07a8217e9f7f7c2be88", 16);
  r->inf = 0;
}
/* VerifiableSS::map_share_to_new_params: Lagrange basis at 0 for x-coordinates s_j + 1 */
static void lagrange_at_zero(mpz_t out, const int32_t* signers, int S, int i) {
  mpz_t num, den, t; mpz_inits(num, den, t, NULL);
  mpz_set_ui(num, 1); mpz_set_ui(den, 1);
  for (int j = 0; j < S; ++j) {
    if (j == i) continue;
    mpz_mul_ui(num, num, (unsigned long)(signers[j] + 1)); sc_mod(num);
    mpz_set_si(t, (long)signers[j] - (long)signers[i]); sc_mod(t);
    mpz_mul(den, den, t); sc_mod(den);
  }
  mpz_invert(den, den, EC_Q);
  mpz_mul(out, num, den); sc_mod(out);
  mpz_clears(num, den, t, NULL);
}

/* ---- curv PedersenProof / HomoELGamalProof, word interface --------------------------- */
/* PedersenProof::prove(m, r) with the nonces s1, s2 as inputs: com = m G + r H, a1 = s1 G, a2 = s2 H,
 * e = H(G, H, com, a1, a2), z1 = s1 + e m, z2 = s2 + e r */
void orc_pedersen_prove(int batch, const uint32_t* m, const uint32_t* r, const uint32_t* s1, const uint32_t* s2, uint32_t* com,
                        uint32_t* e_out, uint32_t* a1_out, uint32_t* a2_out, uint32_t* z1_out, uint32_t* z2_out) {
  mpz_t M, R, S1, S2, e, z; mpz_inits(M, R, S1, S2, e, z, NULL);
  pt_t G, H, C, A1, A2, t; pt_init(&G); pt_init(&H); pt_init(&C); pt_init(&A1); pt_init(&A2); pt_init(&t);
  pt_gen(&G); pt_h2(&H);
  for (int i = 0; i < batch; ++i) {
    zin(M, m + (size_t)i * 8, 8); sc_mod(M); zin(R, r + (size_t)i * 8, 8); sc_mod(R);
    zin(S1, s1 + (size_t)i * 8, 8); sc_mod(S1); zin(S2, s2 + (size_t)i * 8, 8); sc_mod(S2);
    pt_mul(&C, M, &G); pt_mul(&t, R, &H); pt_add(&C, &C, &t);
    pt_mul(&A1, S1, &G); pt_mul(&A2, S2, &H);
    const pt_t* hp[5] = {&G, &H, &C, &A1, &A2};
    hash_points_scalar(e, hp, 5);
    pt_out(com + (size_t)i * 16, &C); pt_out(a1_out + (size_t)i * 16, &A1); pt_out(a2_out + (size_t)i * 16, &A2);
    zout(e_out + (size_t)i * 8, 8, e);
    mpz_mul(z, e, M); mpz_add(z, z, S1); sc_mod(z); zout(z1_out + (size_t)i * 8, 8, z);
    mpz_mul(z, e, R); mpz_add(z, z, S2); sc_mod(z); zout(z2_out + (size_t)i * 8, 8, z);
  }
  pt_clear(&G); pt_clear(&H); pt_clear(&C); pt_clear(&A1); pt_clear(&A2); pt_clear(&t);
  mpz_clears(M, R, S1, S2, e, z, NULL);
}
/* PedersenProof::verify: e recomputed; z1 G + z2 H == a1 + a2 + e com */
void orc_pedersen_verify(int batch, const uint32_t* com, const uint32_t* a1, const uint32_t* a2, const uint32_t* z1,
                         const uint32_t* z2, uint8_t* ok) {
  mpz_t e, Z1, Z2; mpz_inits(e, Z1, Z2, NULL);
  pt_t G, H, C, A1, A2, l, r, t; pt_init(&G); pt_init(&H); pt_init(&C); pt_init(&A1); pt_init(&A2); pt_init(&l); pt_init(&r); pt_init(&t);
  pt_gen(&G); pt_h2(&H);
  for (int i = 0; i < batch; ++i) {
    pt_in(&C, com + (size_t)i * 16); pt_in(&A1, a1 + (size_t)i * 16); pt_in(&A2, a2 + (size_t)i * 16);
    zin(Z1, z1 + (size_t)i * 8, 8); zin(Z2, z2 + (size_t)i * 8, 8);
    const pt_t* hp[5] = {&G, &H, &C, &A1, &A2};
    hash_points_scalar(e, hp, 5);
    pt_mul(&l, Z1, &G); pt_mul(&t, Z2, &H); pt_add(&l, &l, &t);
    pt_add(&r, &A1, &A2); pt_mul(&t, e, &C); pt_add(&r, &r, &t);
    ok[i] = (uint8_t)pt_eq(&l, &r);
  }
  pt_clear(&G); pt_clear(&H); pt_clear(&C); pt_clear(&A1); pt_clear(&A2); pt_clear(&l); pt_clear(&r); pt_clear(&t);
  mpz_clears(e, Z1, Z2, NULL);
}
/* HomoELGamalProof::prove(w{x, r}, delta{G, H, Y, D, E}) with the nonces s1, s2 as inputs:
 * A1 = s1 H, A2 = s2 Y, A3 = s2 G, T = A1 + A2, e = H(T, A3, G, H, Y, D, E), z1 = s1 + e x (s1 if x = 0), z2 = s2 + e r */
void orc_heg_prove(int batch, const uint32_t* x, const uint32_t* r, const uint32_t* s1, const uint32_t* s2, const uint32_t* Gp,
                   const uint32_t* Hp, const uint32_t* Yp, const uint32_t* Dp, const uint32_t* Ep, uint32_t* T_out, uint32_t* A3_out,
                   uint32_t* z1_out, uint32_t* z2_out) {
  mpz_t X, R, S1, S2, e, z; mpz_inits(X, R, S1, S2, e, z, NULL);
  pt_t G, H, Y, D, E, A1, A2, A3, T; pt_init(&G); pt_init(&H); pt_init(&Y); pt_init(&D); pt_init(&E); pt_init(&A1); pt_init(&A2); pt_init(&A3); pt_init(&T);
  for (int i = 0; i < batch; ++i) {
    zin(X, x + (size_t)i * 8, 8); sc_mod(X); zin(R, r + (size_t)i * 8, 8); sc_mod(R);
    zin(S1, s1 + (size_t)i * 8, 8); sc_mod(S1); zin(S2, s2 + (size_t)i * 8, 8); sc_mod(S2);
    pt_in(&G, Gp + (size_t)i * 16); pt_in(&H, Hp + (size_t)i * 16); pt_in(&Y, Yp + (size_t)i * 16); pt_in(&D, Dp + (size_t)i * 16); pt_in(&E, Ep + (size_t)i * 16);
    pt_mul(&A1, S1, &H); pt_mul(&A2, S2, &Y); pt_mul(&A3, S2, &G); pt_add(&T, &A1, &A2);
    const pt_t* hp[7] = {&T, &A3, &G, &H, &Y, &D, &E};
    hash_points_scalar(e, hp, 7);
    if (mpz_sgn(X) != 0) { mpz_mul(z, e, X); mpz_add(z, z, S1); sc_mod(z); } else mpz_set(z, S1);
    zout(z1_out + (size_t)i * 8, 8, z);
    mpz_mul(z, e, R); mpz_add(z, z, S2); sc_mod(z); zout(z2_out + (size_t)i * 8, 8, z);
    pt_out(T_out + (size_t)i * 16, &T); pt_out(A3_out + (size_t)i * 16, &A3);
  }
  pt_clear(&G); pt_clear(&H); pt_clear(&Y); pt_clear(&D); pt_clear(&E); pt_clear(&A1); pt_clear(&A2); pt_clear(&A3); pt_clear(&T);
  mpz_clears(X, R, S1, S2, e, z, NULL);
}
/* HomoELGamalProof::verify: z1 H + z2 Y == T + e D  and  z2 G == A3 + e E */
void orc_heg_verify(int batch, const uint32_t* Gp, const uint32_t* Hp, const uint32_t* Yp, const uint32_t* Dp, const uint32_t* Ep,
                    const uint32_t* Tp, const uint32_t* A3p, const uint32_t* z1, const uint32_t* z2, uint8_t* ok) {
  mpz_t e, Z1, Z2; mpz_inits(e, Z1, Z2, NULL);
  pt_t G, H, Y, D, E, A3, T, l, r, t; pt_init(&G); pt_init(&H); pt_init(&Y); pt_init(&D); pt_init(&E); pt_init(&A3); pt_init(&T); pt_init(&l); pt_init(&r); pt_init(&t);
  for (int i = 0; i < batch; ++i) {
    pt_in(&G, Gp + (size_t)i * 16); pt_in(&H, Hp + (size_t)i * 16); pt_in(&Y, Yp + (size_t)i * 16); pt_in(&D, Dp + (size_t)i * 16); pt_in(&E, Ep + (size_t)i * 16);
    pt_in(&T, Tp + (size_t)i * 16); pt_in(&A3, A3p + (size_t)i * 16);
    zin(Z1, z1 + (size_t)i * 8, 8); zin(Z2, z2 + (size_t)i * 8, 8);
    const pt_t* hp[7] = {&T, &A3, &G, &H, &Y, &D, &E};
    hash_points_scalar(e, hp, 7);
    pt_mul(&l, Z1, &H); pt_mul(&t, Z2, &Y); pt_add(&l, &l, &t);
    pt_mul(&t, e, &D); pt_add(&r, &T, &t);
    int good = pt_eq(&l, &r);
    pt_mul(&l, Z2, &G); pt_mul(&t, e, &E); pt_add(&r, &A3, &t);
    good = good && pt_eq(&l, &r);
    ok[i] = (uint8_t)good;
  }
  pt_clear(&G); pt_clear(&H); pt_clear(&Y); pt_clear(&D); pt_clear(&E); pt_clear(&A3); pt_clear(&T); pt_clear(&l); pt_clear(&r); pt_clear(&t);
  mpz_clears(e, Z1, Z2, NULL);
}
/* HashCommitment::create_commitment_with_user_defined_randomness(BigInt::from_bytes(P.to_bytes(true)), blind) */
void orc_hash_commit_point(int batch, const uint32_t* P, const uint32_t* blind, uint32_t* com) {
  mpz_t b, c; mpz_inits(b, c, NULL);
  pt_t p; pt_init(&p);
  for (int i = 0; i < batch; ++i) {
    pt_in(&p, P + (size_t)i * 16); zin(b, blind + (size_t)i * 8, 8);
    hash_commit_point(c, &p, b);
    zout(com + (size_t)i * 8, 8, c);
  }
  pt_clear(&p); mpz_clears(b, c, NULL);
}

/* ---- one party's state in one session (what the reference carries from RoundN to RoundN+1) ------------------ */
typedef struct {
  uint32_t k[8], gamma[8], w[8], blind[8], ra[64];                 /* sign_keys, phase1_decom, m_a.1 */
  uint32_t g_gamma[16], com[8], ca[128];
  uint32_t beta[GG_MAXS][2][8];                                    /* beta_vec / ni_vec: [jj][v] */
  uint32_t ca_all[GG_MAXS][128], com_all[GG_MAXS][8];              /* m_a_vec[..].c, bc_vec */
  uint32_t bpk_in[GG_MAXS][16];                                    /* mb_gamma_s[jj].b_proof.pk */
  uint32_t delta_i[8], sigma_i[8], l[8], T[16];
  uint32_t ped[5][16];                                             /* TIProof: e, a1, a2, z1, z2 (8- or 16-word fields) */
  uint32_t tvec[GG_MAXS][16];
  uint32_t dinv[8], R[16], Rbar[16];
  uint32_t S_i[16], heg[4][16];                                    /* HEGProof: T, A3, z1, z2 */
  uint32_t r[8], s_i[8], m[8], sig_s[8];
  int32_t recid;
  int32_t status;
  uint32_t bad;
} gg_sess;

struct orc_gg20_party {
  orc_gg20_keys K;
  int ord, B, L, li;
  int fault_step;                  /* the reference tests' corrupt_step (test.rs:282-289): 5 / 6 / 7, 0 = honest */
  const int32_t* keyset;
  orc_gg20_nonces Z;
  gg_sess* s;
};

static void gg_fail(gg_sess* s, int code, uint32_t bad) {
  if (s->status == 0) { s->status = code; s->bad = bad; }
}

/* key material of party `a` of the key set of session b */
typedef struct { const uint32_t *x, *p, *q, *N, *Nt, *h1, *h2, *y, *X; } gg_kv;
static gg_kv gg_keys_of(const orc_gg20_party* P, int b) {
  const orc_gg20_keys* K = &P->K;
  const size_t ks = P->keyset ? (size_t)P->keyset[b] : 0, n = (size_t)K->n;
  gg_kv v;
  v.x = K->x + ks * n * 8; v.p = K->p + ks * n * 32; v.q = K->q + ks * n * 32; v.N = K->N ? K->N + ks * n * 64 : NULL;
  v.Nt = K->Nt + ks * n * 64; v.h1 = K->h1 + ks * n * 64; v.h2 = K->h2 + ks * n * 64; v.y = K->y + ks * 16; v.X = K->X + ks * n * 16;
  return v;
}
/* N of party a: paillier_key_vec[a] (public), or p*q when the fixture holds every party's primes */
static void gg_N_words(const gg_kv* kv, int a, uint32_t* Nw) {
  if (kv->N) { memcpy(Nw, kv->N + (size_t)a * 64, 256); return; }
  mpz_t p, q; mpz_inits(p, q, NULL);
  zin(p, kv->p + (size_t)a * 32, 32); zin(q, kv->q + (size_t)a * 32, 32);
  mpz_mul(p, p, q); zout(Nw, 64, p);
  mpz_clears(p, q, NULL);
}
static const uint32_t* gg_rec(const uint32_t* in, const int64_t* off, int B, int W, int j, int b) {
  const int64_t o = off ? off[j] : (int64_t)j * B;
  return in + (size_t)(o + b) * (size_t)W;
}
static int words_eq(const uint32_t* a, const uint32_t* b, int n) { return memcmp(a, b, (size_t)n * 4) == 0; }

/* Malformed points in the messages a party is about to read: curv's Point deserialisation (coordinates < p, on the curve; the
 * identity never occurs in an honest message) rejects them before the round runs.  status 100*round + 90, bad_actors = senders. */
static int pt_words_valid(const uint32_t* w) {
  ec_setup();
  mpz_t x, y, l, r; mpz_inits(x, y, l, r, NULL);
  zin(x, w, 8); zin(y, w + 8, 8);
  int ok = !(mpz_sgn(x) == 0 && mpz_sgn(y) == 0) && mpz_cmp(x, EC_P) < 0 && mpz_cmp(y, EC_P) < 0;
  if (ok) {
    mpz_mul(l, y, y); mpz_mod(l, l, EC_P);
    mpz_mul(r, x, x); mpz_mul(r, r, x); mpz_add_ui(r, r, 7); mpz_mod(r, r, EC_P);
    ok = mpz_cmp(l, r) == 0;
  }
  mpz_clears(x, y, l, r, NULL);
  return ok;
}
static void gg_validate(orc_gg20_party* P, int b, int round, const uint32_t* in, const int64_t* off) {
  const int S = P->K.S, n = P->K.n, i = P->ord;
  const int W = orc_gg20_msg_words(S, n, round - 1);
  uint32_t mask = 0;
  for (int j = 0; j < S; ++j) {
    if (j == i) continue;
    const uint32_t* m = gg_rec(in, off, P->B, W, j, b);
    int good = 1;
    if (round == 2) {
      for (int v = 0; v < 2; ++v) {
        const uint32_t* mb = m + (size_t)(jme_of(i, j) * 2 + v) * GG_SUB1;
        good = good && pt_words_valid(mb + 128) && pt_words_valid(mb + 144) && pt_words_valid(mb + 168) && pt_words_valid(mb + 184);
      }
    } else if (round == 3) good = pt_words_valid(m + 8) && pt_words_valid(m + 32) && pt_words_valid(m + 48) && pt_words_valid(m + 64);
    else if (round == 4) good = pt_words_valid(m + 8);
    else if (round == 5) {
      for (int jj = 0; jj < S - 1; ++jj) good = good && pt_words_valid(m + (size_t)jj * GG_SUB4 + 64);
      good = good && pt_words_valid(m + (size_t)(S - 1) * GG_SUB4);
    } else if (round == 6) good = pt_words_valid(m) && pt_words_valid(m + 16) && pt_words_valid(m + 32);
    if (!good) mask |= 1u << j;
  }
  if (mask) gg_fail(&P->s[b], 100 * round + 90, mask);
}

/* ---- Round 0 (rounds.rs:68-104): SignKeys::create, phase1_broadcast, MessageA::a ----------------------------- */
static void gg_round0(orc_gg20_party* P, int b, uint32_t* out) {
  const orc_gg20_keys* K = &P->K;
  const int S = K->S, n = K->n, i = P->ord, me = K->signers[i];
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  const size_t pi = (size_t)b * P->L + P->li;
  mpz_t k, g, lam, x, w, blind, com; mpz_inits(k, g, lam, x, w, blind, com, NULL);
  pt_t G, gg; pt_init(&G); pt_init(&gg); pt_gen(&G);
  zin(k, P->Z.k + pi * 8, 8);
  zin(g, P->Z.gamma + pi * 8, 8);
  /* k_i, gamma_i = Scalar::random() (party_i.rs:561-563): 0 < x < q.  Anything else is the mark the device sampler leaves when a
   * rejection loop gave up (or a caller's mistake): the party stops, status 91 (MPE_GG20_STATUS_BAD_NONCE, include/mpecdsa_hip.h) */
  if (mpz_sgn(k) == 0 || mpz_cmp(k, EC_Q) >= 0 || mpz_sgn(g) == 0 || mpz_cmp(g, EC_Q) >= 0) gg_fail(s, 91, 0);
  sc_mod(k); zout(s->k, 8, k);
  sc_mod(g); zout(s->gamma, 8, g);
  memcpy(s->blind, P->Z.blind + pi * 8, 32);
  memcpy(s->ra, P->Z.r_a + pi * 64, 256);
  lagrange_at_zero(lam, K->signers, S, i);                              /* party_i.rs:553-557 */
  zin(x, kv.x + (size_t)me * 8, 8);
  mpz_mul(w, lam, x); sc_mod(w); zout(s->w, 8, w);                       /* w_i = li * x_i :558 */
  pt_mul(&gg, g, &G); pt_out(s->g_gamma, &gg);                           /* :562 */
  zin(blind, s->blind, 8);
  hash_commit_point(com, &gg, blind); zout(s->com, 8, com);              /* phase1_broadcast :573-589 */
  uint32_t Nw[64], k64[64] = {0};
  gg_N_words(&kv, me, Nw);
  memcpy(k64, s->k, 32);
  orc_paillier_encrypt(1, 1, Nw, NULL, k64, s->ra, s->ca);               /* MessageA::a mta/mod.rs:68-75 */
  memset(out, 0, (size_t)GG_SUB0 * (n + 1) * 4);
  for (int st = 0; st < n; ++st) {                                       /* :76-81, all n statements (rounds.rs:87) */
    const size_t ix = pi * n + st;
    uint32_t* o = out + (size_t)st * GG_SUB0;
    orc_alice_generate(1, 1, Nw, 1, kv.Nt + (size_t)st * 64, kv.h1 + (size_t)st * 64, kv.h2 + (size_t)st * 64, NULL, NULL, s->k, s->ca,
                       s->ra, P->Z.al_alpha + ix * 24, P->Z.al_beta + ix * 64, P->Z.al_gamma + ix * 88, P->Z.al_rho + ix * 72,
                       o, o + 64, o + 72, o + 136, o + 161);
  }
  memcpy(out + (size_t)n * GG_SUB0, s->ca, 512);
  memcpy(out + (size_t)n * GG_SUB0 + 128, s->com, 32);
  pt_clear(&G); pt_clear(&gg); mpz_clears(k, g, lam, x, w, blind, com, NULL);
}

/* ---- Round 1 (rounds.rs:122-206): MessageB::b for gamma_i and w_i towards every peer ------------------------- */
static void gg_round1(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off, uint32_t* out) {
  const orc_gg20_keys* K = &P->K;
  const int S = K->S, n = K->n, i = P->ord, W0 = GG_SUB0 * (n + 1);
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  const size_t pi = (size_t)b * P->L + P->li;
  for (int j = 0; j < S; ++j) {                                          /* into_vec_including_me: m_a_vec, bc_vec */
    const uint32_t* rec = gg_rec(in, off, P->B, W0, j, b) + (size_t)n * GG_SUB0;
    memcpy(s->ca_all[j], rec, 512); memcpy(s->com_all[j], rec + 128, 32);
  }
  mpz_t bt, t; mpz_inits(bt, t, NULL);
  memset(out, 0, (size_t)GG_SUB1 * 2 * (S - 1) * 4);
  for (int jj = 0; jj < S - 1; ++jj) {
    const int ind = ind_of(i, jj), alice = K->signers[ind];
    uint32_t Nw[64];
    gg_N_words(&kv, alice, Nw);
    const uint32_t* arec = gg_rec(in, off, P->B, W0, ind, b);
    for (int v = 0; v < 2; ++v) {
      /* verify Alice's n range proofs (mta/mod.rs:119-131); executed for both calls as the reference does */
      for (int st = 0; st < n; ++st) {
        const uint32_t* pr = arec + (size_t)st * GG_SUB0;
        uint8_t ok = 0;
        orc_alice_verify(1, 1, Nw, 1, kv.Nt + (size_t)st * 64, kv.h1 + (size_t)st * 64, kv.h2 + (size_t)st * 64, NULL, NULL,
                         s->ca_all[ind], pr, pr + 64, pr + 72, pr + 136, pr + 161, &ok);
        if (!ok) gg_fail(s, 101, 0);
      }
      const size_t ix = (pi * (S - 1) + jj) * 2 + v;
      uint32_t* o = out + (size_t)(jj * 2 + v) * GG_SUB1;
      const uint32_t* bsel = v == 0 ? s->gamma : s->w;
      uint32_t cbt[128], bca[128], b64[64] = {0};
      orc_paillier_encrypt(1, 1, Nw, NULL, P->Z.mb_beta_tag + ix * 64, P->Z.mb_r + ix * 64, cbt);       /* :133-137 */
      memcpy(b64, bsel, 32);
      orc_paillier_mul(1, 1, Nw, NULL, s->ca_all[ind], b64, bca);                                        /* :140-144 */
      orc_paillier_add(1, 1, Nw, NULL, bca, cbt, o);                                                     /* :145 */
      zin(bt, P->Z.mb_beta_tag + ix * 64, 64); sc_mod(bt);                                               /* beta_tag_fe :132 */
      mpz_neg(t, bt); sc_mod(t); zout(s->beta[jj][v], 8, t);                                             /* beta = -beta_tag :146 */
      orc_dlog_prove(1, bsel, P->Z.mb_nonce_b + ix * 8, o + 128, o + 144, o + 160);                      /* :147 */
      uint32_t btw[8]; zout(btw, 8, bt);
      orc_dlog_prove(1, btw, P->Z.mb_nonce_bt + ix * 8, o + 168, o + 184, o + 200);                      /* :148 */
    }
  }
  mpz_clears(bt, t, NULL);
}

/* ---- Round 2 (rounds.rs:234-317): verify_proofs_get_alpha, delta_i, sigma_i, T_i + PedersenProof -------------- */
static void gg_round2(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off, uint32_t* out) {
  const orc_gg20_keys* K = &P->K;
  const int S = K->S, i = P->ord, me = K->signers[i], W1 = GG_SUB1 * 2 * (S - 1);
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  const size_t pi = (size_t)b * P->L + P->li;
  mpz_t k, de, si, t, lam, al; mpz_inits(k, de, si, t, lam, al, NULL);
  pt_t G, Bp, BTp, gw, a, c, Xp; pt_init(&G); pt_init(&Bp); pt_init(&BTp); pt_init(&gw); pt_init(&a); pt_init(&c); pt_init(&Xp); pt_gen(&G);
  zin(k, s->k, 8);
  zin(t, s->gamma, 8); mpz_mul(de, k, t); sc_mod(de);                     /* phase2_delta_i :591-604 */
  zin(t, s->w, 8); mpz_mul(si, k, t); sc_mod(si);                         /* phase2_sigma_i :606-618 */
  for (int jj = 0; jj < S - 1; ++jj) {
    const int ind = ind_of(i, jj), jme = jme_of(i, ind);
    const uint32_t* rec = gg_rec(in, off, P->B, W1, ind, b);
    for (int v = 0; v < 2; ++v) {
      const uint32_t* mb = rec + (size_t)(jme * 2 + v) * GG_SUB1;
      uint32_t mw[64];
      orc_paillier_decrypt(1, 1, kv.p + (size_t)me * 32, kv.q + (size_t)me * 32, NULL, mb, mw);   /* mta/mod.rs:165 */
      zin(al, mw, 64); sc_mod(al);                                         /* alpha :167 */
      pt_mul(&a, al, &G);                                                  /* g_alpha :168 */
      pt_in(&Bp, mb + 128); pt_in(&BTp, mb + 168);
      pt_mul(&c, k, &Bp); pt_add(&c, &c, &BTp);                            /* ba_btag :169 */
      uint8_t ok1 = 0, ok2 = 0;
      orc_dlog_verify(1, mb + 128, mb + 144, mb + 160, &ok1);              /* :170 */
      orc_dlog_verify(1, mb + 168, mb + 184, mb + 200, &ok2);              /* :171 */
      if (!ok1 || !ok2 || !pt_eq(&a, &c)) gg_fail(s, 201, 0);              /* :173-177 */
      if (v == 1) {                                                        /* rounds.rs:281: g_w_vec[ind] = lambda_ind X_ind (party_i.rs:527-544) */
        lagrange_at_zero(lam, K->signers, S, ind);
        pt_in(&Xp, kv.X + (size_t)K->signers[ind] * 16);
        pt_mul(&gw, lam, &Xp);
        if (!pt_eq(&Bp, &gw)) gg_fail(s, 202, 0);
      } else {
        memcpy(s->bpk_in[jj], mb + 128, 64);                               /* mb_gamma_s[jj].b_proof.pk, used by phase4 */
      }
      zin(t, s->beta[jj][v], 8); mpz_add(t, t, al);
      if (v == 0) { mpz_add(de, de, t); sc_mod(de); } else { mpz_add(si, si, t); sc_mod(si); }
    }
  }
  if (P->fault_step == 5) { mpz_add(de, de, de); sc_mod(de); }            /* test.rs:458-461 */
  if (P->fault_step == 6) { mpz_add(si, si, si); sc_mod(si); }            /* test.rs:462-465 */
  zout(s->delta_i, 8, de); zout(s->sigma_i, 8, si);
  /* phase3_compute_t_i :620-634: T = sigma G + l H and PedersenProof::prove(sigma_i, l) */
  zin(t, P->Z.l + pi * 8, 8); sc_mod(t); zout(s->l, 8, t);
  memset(s->ped, 0, sizeof s->ped);
  orc_pedersen_prove(1, s->sigma_i, s->l, P->Z.ped_s1 + pi * 8, P->Z.ped_s2 + pi * 8, s->T, s->ped[0], s->ped[1], s->ped[2], s->ped[3], s->ped[4]);
  memset(out, 0, GG_W2 * 4);
  memcpy(out, s->delta_i, 32); memcpy(out + 8, s->T, 64);
  memcpy(out + 24, s->ped[0], 32); memcpy(out + 32, s->ped[1], 64); memcpy(out + 48, s->ped[2], 64); memcpy(out + 64, s->T, 64);
  memcpy(out + 80, s->ped[3], 32); memcpy(out + 88, s->ped[4], 32);
  pt_clear(&G); pt_clear(&Bp); pt_clear(&BTp); pt_clear(&gw); pt_clear(&a); pt_clear(&c); pt_clear(&Xp);
  mpz_clears(k, de, si, t, lam, al, NULL);
}

/* ---- Round 3 (rounds.rs:347-402): T_i == proof.com, delta^-1, PedersenProof::verify; decommit ------------------ */
static void gg_round3(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off, uint32_t* out) {
  const int S = P->K.S;
  gg_sess* s = &P->s[b];
  mpz_t sum, t; mpz_inits(sum, t, NULL);
  int com_ok = 1, ped_ok = 1;
  for (int j = 0; j < S; ++j) {
    const uint32_t* rec = gg_rec(in, off, P->B, GG_W2, j, b);
    memcpy(s->tvec[j], rec + 8, 64);
    if (!words_eq(rec + 8, rec + 64, 16)) com_ok = 0;                     /* rounds.rs:365-367 */
    zin(t, rec, 8); mpz_add(sum, sum, t); sc_mod(sum);
    uint8_t ok = 0;
    orc_pedersen_verify(1, rec + 64, rec + 32, rec + 48, rec + 80, rec + 88, &ok);
    if (!ok) ped_ok = 0;
  }
  if (!com_ok) gg_fail(s, 303, 0);
  if (!mpz_invert(t, sum, EC_Q)) { gg_fail(s, 301, 0); mpz_set_ui(t, 0); }   /* phase3_reconstruct_delta :635-640 */
  zout(s->dinv, 8, t);
  if (!ped_ok) gg_fail(s, 302, 0);
  memset(out, 0, GG_W3 * 4);
  memcpy(out, s->blind, 32); memcpy(out + 8, s->g_gamma, 64);
  mpz_clears(sum, t, NULL);
}

/* ---- Round 4 (rounds.rs:431-498): phase4 -> R, R_dash = k_i R, one PDLwSlackProof per peer ------------------- */
static void gg_round4(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off, uint32_t* out) {
  const orc_gg20_keys* K = &P->K;
  const int S = K->S, i = P->ord, me = K->signers[i];
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  const size_t pi = (size_t)b * P->L + P->li;
  mpz_t t, blind, com; mpz_inits(t, blind, com, NULL);
  pt_t gg, acc, R, Rb; pt_init(&gg); pt_init(&acc); pt_init(&R); pt_init(&Rb);
  uint32_t bad = 0;
  for (int jj = 0; jj < S - 1; ++jj) {                                    /* phase4 :642-687 */
    const int ind = ind_of(i, jj);
    const uint32_t* rec = gg_rec(in, off, P->B, GG_W3, ind, b);
    pt_in(&gg, rec + 8); zin(blind, rec, 8);
    hash_commit_point(com, &gg, blind);
    zin(t, s->com_all[ind], 8);
    if (!words_eq(s->bpk_in[jj], rec + 8, 16) || mpz_cmp(com, t) != 0) bad |= 1u << ind;
  }
  if (bad) gg_fail(s, 401, bad);
  for (int j = 0; j < S; ++j) {                                           /* gamma_sum over the decommitments, mine included */
    pt_in(&gg, gg_rec(in, off, P->B, GG_W3, j, b) + 8);
    pt_add(&acc, &acc, &gg);
  }
  zin(t, s->dinv, 8); pt_mul(&R, t, &acc); pt_out(s->R, &R);              /* R = gamma_sum * delta_inv */
  zin(t, s->k, 8); pt_mul(&Rb, t, &R); pt_out(s->Rbar, &Rb);              /* R_dash = R * k_i  rounds.rs:452 */
  uint32_t Nw[64];
  gg_N_words(&kv, me, Nw);
  memset(out, 0, (size_t)GG_SUB4 * S * 4);
  for (int jj = 0; jj < S - 1; ++jj) {                                    /* phase5_proof_pdl :691-717 */
    const int st = K->signers[ind_of(i, jj)];
    const size_t ix = pi * (S - 1) + jj;
    uint32_t* o = out + (size_t)jj * GG_SUB4;
    orc_pdl_prove(1, 1, Nw, 1, kv.Nt + (size_t)st * 64, kv.h1 + (size_t)st * 64, kv.h2 + (size_t)st * 64, NULL, NULL, s->ca, s->Rbar,
                  s->R, s->k, s->ra, P->Z.pdl_alpha + ix * 24, P->Z.pdl_beta + ix * 64, P->Z.pdl_rho + ix * 72, P->Z.pdl_gamma + ix * 88,
                  o, o + 64, o + 80, o + 208, o + 272, o + 297, o + 361);
  }
  memcpy(out + (size_t)(S - 1) * GG_SUB4, s->Rbar, 64);
  pt_clear(&gg); pt_clear(&acc); pt_clear(&R); pt_clear(&Rb); mpz_clears(t, blind, com, NULL);
}

/* ---- Round 5 (rounds.rs:525-601): all PDL proofs, sum R_dash, S_i + HomoELGamalProof ------------------------- */
static void gg_round5(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off, uint32_t* out) {
  const orc_gg20_keys* K = &P->K;
  const int S = K->S, W4 = GG_SUB4 * S;
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  const size_t pi = (size_t)b * P->L + P->li;
  pt_t acc, G, t; pt_init(&acc); pt_init(&G); pt_init(&t); pt_gen(&G);
  uint32_t bad = 0;
  for (int i = 0; i < S && !bad; ++i) {                                    /* `?` stops at the first failing prover */
    const uint32_t* rec = gg_rec(in, off, P->B, W4, i, b);
    const uint32_t* rdash = rec + (size_t)(S - 1) * GG_SUB4;
    uint32_t Nw[64];
    gg_N_words(&kv, K->signers[i], Nw);
    for (int jj = 0; jj < S - 1; ++jj) {                                   /* phase5_verify_pdl :719-766, G = MY R */
      const int st = K->signers[ind_of(i, jj)];
      const uint32_t* o = rec + (size_t)jj * GG_SUB4;
      uint8_t ok = 0;
      orc_pdl_verify(1, 1, Nw, 1, kv.Nt + (size_t)st * 64, kv.h1 + (size_t)st * 64, kv.h2 + (size_t)st * 64, NULL, NULL, s->ca_all[i],
                     rdash, s->R, o, o + 64, o + 80, o + 208, o + 272, o + 297, o + 361, &ok);
      if (!ok) bad |= 1u << i;
    }
  }
  if (bad) gg_fail(s, 501, bad);
  for (int j = 0; j < S; ++j) {                                            /* phase5_check_R_dash_sum :768-776 */
    pt_in(&t, gg_rec(in, off, P->B, W4, j, b) + (size_t)(S - 1) * GG_SUB4);
    pt_add(&acc, &acc, &t);
  }
  if (!pt_eq(&acc, &G)) gg_fail(s, 502, 0);
  /* phase6_compute_S_i_and_proof_of_consistency :778-799: G = R, H = base_point2, Y = generator, D = T_i, E = S_i, x = l_i, r = sigma_i */
  mpz_t si; mpz_init(si);
  pt_t R, Sp, H; pt_init(&R); pt_init(&Sp); pt_init(&H); pt_h2(&H);
  pt_in(&R, s->R); zin(si, s->sigma_i, 8);
  pt_mul(&Sp, si, &R); pt_out(s->S_i, &Sp);
  uint32_t Gw[16], Hw[16];
  pt_out(Gw, &G); pt_out(Hw, &H);
  memset(s->heg, 0, sizeof s->heg);
  orc_heg_prove(1, s->l, s->sigma_i, P->Z.heg_s1 + pi * 8, P->Z.heg_s2 + pi * 8, s->R, Hw, Gw, s->T, s->S_i, s->heg[0], s->heg[1], s->heg[2], s->heg[3]);
  memset(out, 0, GG_W5 * 4);
  memcpy(out, s->S_i, 64); memcpy(out + 16, s->heg[0], 64); memcpy(out + 32, s->heg[1], 64);
  memcpy(out + 48, s->heg[2], 32); memcpy(out + 56, s->heg[3], 32);
  pt_clear(&acc); pt_clear(&G); pt_clear(&t); pt_clear(&R); pt_clear(&Sp); pt_clear(&H); mpz_clear(si);
}

/* ---- Round 6 (rounds.rs:612-636): every HomoELGamalProof, sum S_i == y ------------------------------------------- */
static void gg_round6(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off) {
  const int S = P->K.S;
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  pt_t acc, G, H, t, y; pt_init(&acc); pt_init(&G); pt_init(&H); pt_init(&t); pt_init(&y); pt_gen(&G); pt_h2(&H);
  uint32_t Gw[16], Hw[16], bad = 0;
  pt_out(Gw, &G); pt_out(Hw, &H);
  for (int j = 0; j < S; ++j) {                                            /* phase6_verify_proof :801-833: collects every failure */
    const uint32_t* rec = gg_rec(in, off, P->B, GG_W5, j, b);
    uint8_t ok = 0;
    orc_heg_verify(1, s->R, Hw, Gw, s->tvec[j], rec, rec + 16, rec + 32, rec + 48, rec + 56, &ok);
    if (!ok) bad |= 1u << j;
    pt_in(&t, rec); pt_add(&acc, &acc, &t);
  }
  if (bad) gg_fail(s, 601, bad);
  pt_in(&y, kv.y);
  if (!pt_eq(&acc, &y)) gg_fail(s, 602, 0);                                /* phase6_check_S_i_sum :835-848 */
  pt_clear(&acc); pt_clear(&G); pt_clear(&H); pt_clear(&t); pt_clear(&y);
}

/* ---- Round 7 (rounds.rs:672-692, party_i.rs:850-871): phase7_local_sig -> PartialSignature ----------------------- */
static void gg_round7(orc_gg20_party* P, int b, uint32_t* out) {
  gg_sess* s = &P->s[b];
  mpz_t m, r, t, t2; mpz_inits(m, r, t, t2, NULL);
  memcpy(s->m, P->Z.msg + (size_t)b * 8, 32);
  zin(m, s->m, 8); sc_mod(m);
  zin(r, s->R, 8); sc_mod(r); zout(s->r, 8, r);                            /* r = R.x mod q */
  zin(t, s->k, 8); mpz_mul(t, t, m);
  zin(t2, s->sigma_i, 8); mpz_mul(t2, t2, r); mpz_add(t, t, t2); sc_mod(t);   /* s_i = m k_i + r sigma_i :864 */
  if (P->fault_step == 7) { mpz_add(t, t, t); sc_mod(t); }                    /* test.rs:679-686 */
  zout(s->s_i, 8, t);
  memcpy(out, s->s_i, 32);
  mpz_clears(m, r, t, t2, NULL);
}
/* SignManual::complete -> output_signature (party_i.rs:873-910) + verify (:913-936) */
static void gg_complete(orc_gg20_party* P, int b, const uint32_t* in, const int64_t* off) {
  const int S = P->K.S, i = P->ord;
  gg_sess* s = &P->s[b];
  const gg_kv kv = gg_keys_of(P, b);
  mpz_t sg, t, r, m, half, ry; mpz_inits(sg, t, r, m, half, ry, NULL);
  zin(sg, s->s_i, 8);
  for (int j = 0; j < S; ++j) {
    if (j == i) continue;
    zin(t, gg_rec(in, off, P->B, GG_W6, j, b), 8); mpz_add(sg, sg, t); sc_mod(sg);
  }
  zin(r, s->r, 8);
  zin(ry, s->R + 8, 8); mpz_mod(ry, ry, EC_Q);
  int recid = mpz_tstbit(ry, 0) ? 1 : 0;
  mpz_sub(half, EC_Q, sg);
  if (mpz_cmp(sg, half) > 0) { mpz_set(sg, half); recid ^= 1; }             /* :896-900 */
  pt_t G, y, a, c; pt_init(&G); pt_init(&y); pt_init(&a); pt_init(&c); pt_gen(&G);
  int okv = mpz_invert(t, sg, EC_Q) != 0;
  if (okv) {
    mpz_t u1, u2; mpz_inits(u1, u2, NULL);
    zin(m, s->m, 8); sc_mod(m);
    mpz_mul(u1, m, t); sc_mod(u1); mpz_mul(u2, r, t); sc_mod(u2);
    pt_in(&y, kv.y);
    pt_mul(&a, u1, &G); pt_mul(&c, u2, &y); pt_add(&a, &a, &c);
    mpz_mod(t, a.x, EC_Q);
    okv = !a.inf && mpz_cmp(t, r) == 0;
    mpz_clears(u1, u2, NULL);
  }
  if (!okv) gg_fail(s, 701, 0);
  zout(s->sig_s, 8, sg); s->recid = recid;
  pt_clear(&G); pt_clear(&y); pt_clear(&a); pt_clear(&c);
  mpz_clears(sg, t, r, m, half, ry, NULL);
}

/* ---- the per-party object ------------------------------------------------------------------------------------------ */
/* K: tables [nkeysets][n][..]; this party reads x, p, q of its OWN index only (signers[ord]) and N of everybody (NULL: p*q).
 * Z: nonces with leading dimensions [B][L], this party at position li (L = 1, li = 0 when the arrays hold one party).
 * keyset: [B] key-set index per session, or NULL. */
orc_gg20_party* orc_gg20_party_new(const orc_gg20_keys* K, int ord, int B, const orc_gg20_nonces* Z, int L, int li, const int32_t* keyset) {
  if (K->S > GG_MAXS || K->n > GG_MAXN || ord < 0 || ord >= K->S) return NULL;
  ec_setup();
  orc_gg20_party* P = (orc_gg20_party*)calloc(1, sizeof *P);
  P->K = *K; P->ord = ord; P->B = B; P->L = L; P->li = li; P->keyset = keyset; P->Z = *Z;
  P->s = (gg_sess*)calloc((size_t)B, sizeof(gg_sess));
  return P;
}
void orc_gg20_party_free(orc_gg20_party* P) {
  if (!P) return;
  if (P->s) { memset(P->s, 0, (size_t)P->B * sizeof(gg_sess)); free(P->s); }      /* secrets do not outlive the object */
  free(P);
}
/* round 0..7 = RoundN::proceed / Round7::new; round 8 = SignManual::complete.  in: the previous round's records of all S senders
 * (sender j's [B][W] block at record offset in_off[j]; NULL: j*B); out: this party's [B][W] block.  Sessions [first, first+count). */
void orc_gg20_party_round(orc_gg20_party* P, int round, const uint32_t* in, const int64_t* in_off, uint32_t* out, int first, int count) {
  const int S = P->K.S, n = P->K.n;
  const size_t W = (size_t)orc_gg20_msg_words(S, n, round);
  for (int b = first; b < first + count; ++b) {
    uint32_t* o = out ? out + (size_t)b * W : NULL;
    switch (round) {
      case 0: gg_round0(P, b, o); break;
      case 1: gg_round1(P, b, in, in_off, o); break;
      case 2: gg_validate(P, b, 2, in, in_off); gg_round2(P, b, in, in_off, o); break;
      case 3: gg_validate(P, b, 3, in, in_off); gg_round3(P, b, in, in_off, o); break;
      case 4: gg_validate(P, b, 4, in, in_off); gg_round4(P, b, in, in_off, o); break;
      case 5: gg_validate(P, b, 5, in, in_off); gg_round5(P, b, in, in_off, o); break;
      case 6: gg_validate(P, b, 6, in, in_off); gg_round6(P, b, in, in_off); break;
      case 7: gg_round7(P, b, o); break;
      case 8: gg_complete(P, b, in, in_off); break;
      default: break;
    }
    if (o && P->s[b].status) memset(o, 0, W * 4);      /* a party that failed has stopped: RoundN::proceed returned Err, nothing is sent */
  }
}
/* per-session results of this party: status, bad_actors, SignatureRecid (zero when status != 0), R */
void orc_gg20_party_result(const orc_gg20_party* P, int32_t* status, uint32_t* bad, uint32_t* r, uint32_t* s, int32_t* recid, uint32_t* R) {
  for (int b = 0; b < P->B; ++b) {
    const gg_sess* x = &P->s[b];
    if (status) status[b] = x->status;
    if (bad) bad[b] = x->bad;
    const int good = x->status == 0;
    if (r) { if (good) memcpy(r + (size_t)b * 8, x->r, 32); else memset(r + (size_t)b * 8, 0, 32); }
    if (s) { if (good) memcpy(s + (size_t)b * 8, x->sig_s, 32); else memset(s + (size_t)b * 8, 0, 32); }
    if (recid) recid[b] = good ? x->recid : 0;
    if (R) memcpy(R + (size_t)b * 16, x->R, 64);
  }
}
/* test hook = the reference tests' corrupt_step (gg_2020/test.rs:282-289,458-465,679-686): this party doubles its delta_i (step 5),
 * sigma_i (step 6) or s_i (step 7) in every session; 0 = honest */
void orc_gg20_party_fault(orc_gg20_party* P, int step) { P->fault_step = step; }

/* ---- all parties of sessions [first, first+count) in lock-step: round_based::dev::Simulation (sign.rs:667-763) ----------
 * Z: [B][S] layout (every party's nonces).  slabs: NULL, or 7 pointers (M0..M6) to [S][B][W] arrays that receive every
 * message.  party_status / party_bad: NULL or [S][B].  status[b] = the smallest non-zero party status (0 = all parties signed);
 * r, s, recid, R: party 0's output. */
void orc_gg20_sign_ex(const orc_gg20_keys* K, const orc_gg20_nonces* Z, const int32_t* keyset, int B, int first, int count,
                      uint32_t* const* slabs, uint32_t* r_out, uint32_t* s_out, int32_t* recid_out, uint32_t* R_out, int32_t* status,
                      int32_t* party_status, uint32_t* party_bad) {
  const int S = K->S, n = K->n;
  static const int rounds[7] = {0, 1, 2, 3, 4, 5, 7};
  uint32_t* own[7] = {0};
  uint32_t* m[7];
  for (int q = 0; q < 7; ++q) {
    if (slabs && slabs[q]) m[q] = slabs[q];
    else m[q] = own[q] = (uint32_t*)calloc((size_t)S * B * orc_gg20_msg_words(S, n, rounds[q]), 4);
  }
  orc_gg20_party* P[GG_MAXS];
  for (int i = 0; i < S; ++i) P[i] = orc_gg20_party_new(K, i, B, Z, S, i, keyset);
  for (int q = 0; q < 7; ++q) {
    const int round = rounds[q];
    const size_t W = (size_t)orc_gg20_msg_words(S, n, round);
    for (int i = 0; i < S; ++i) {
      if (round == 7) orc_gg20_party_round(P[i], 6, m[5], NULL, NULL, first, count);
      orc_gg20_party_round(P[i], round, q ? m[q - 1] : NULL, NULL, m[q] + (size_t)i * B * W, first, count);
    }
  }
  for (int i = 0; i < S; ++i) orc_gg20_party_round(P[i], 8, m[6], NULL, NULL, first, count);
  for (int b = first; b < first + count; ++b) {
    int st = 0;
    for (int i = 0; i < S; ++i) {
      const gg_sess* x = &P[i]->s[b];
      if (x->status && (!st || x->status < st)) st = x->status;
      if (party_status) party_status[(size_t)i * B + b] = x->status;
      if (party_bad) party_bad[(size_t)i * B + b] = x->bad;
    }
    const gg_sess* x = &P[0]->s[b];
    if (status) status[b] = st;
    if (st == 0) { memcpy(r_out + (size_t)b * 8, x->r, 32); memcpy(s_out + (size_t)b * 8, x->sig_s, 32); recid_out[b] = x->recid; }
    else { memset(r_out + (size_t)b * 8, 0, 32); memset(s_out + (size_t)b * 8, 0, 32); recid_out[b] = 0; }
    if (R_out) memcpy(R_out + (size_t)b * 16, x->R, 64);
  }
  for (int i = 0; i < S; ++i) orc_gg20_party_free(P[i]);
  for (int q = 0; q < 7; ++q) free(own[q]);
}
void orc_gg20_sign(const orc_gg20_keys* K, const orc_gg20_nonces* Z, int first, int count, uint32_t* r_out, uint32_t* s_out,
                   int32_t* recid_out, uint32_t* R_out, int32_t* status) {
  /* the arrays are indexed by absolute session number: B = first + count covers them */
  orc_gg20_sign_ex(K, Z, NULL, first + count, first, count, NULL, r_out, s_out, recid_out, R_out, status, NULL, NULL);
}

/* ==========================================================================================================================
 * Identifiable abort: src/protocols/multi_party_ecdsa/gg_2020/blame.rs.  Every signer opens the values the failing phase used;
 * the blame functions re-derive the public ciphertexts from them and name the parties whose openings do not match (or whose
 * broadcast delta_i / S_i / s_i is inconsistent).  Arrays: leading dimensions [B][S] (signer ordinal), then the peer slot j
 * (S-1; ind = j < i ? j : j+1).  N: [nkeysets][n][64] must be given or derivable (p*q).  bad[b]: bit mask over signer ordinals.
 * ========================================================================================================================== */
/* curv ECDDHProof (SURVEY.md App. A.3 family; recalled): a1 = s g1, a2 = s g2, e = H(g1, h1, g2, h2, a1, a2), z = s + e x */
void orc_ecddh_prove(int batch, const uint32_t* x, const uint32_t* s_in, const uint32_t* g1, const uint32_t* h1, const uint32_t* g2,
                     const uint32_t* h2, uint32_t* a1_out, uint32_t* a2_out, uint32_t* z_out) {
  mpz_t X, S, e, z; mpz_inits(X, S, e, z, NULL);
  pt_t G1, H1, G2, H2, A1, A2; pt_init(&G1); pt_init(&H1); pt_init(&G2); pt_init(&H2); pt_init(&A1); pt_init(&A2);
  for (int i = 0; i < batch; ++i) {
    zin(X, x + (size_t)i * 8, 8); sc_mod(X); zin(S, s_in + (size_t)i * 8, 8); sc_mod(S);
    pt_in(&G1, g1 + (size_t)i * 16); pt_in(&H1, h1 + (size_t)i * 16); pt_in(&G2, g2 + (size_t)i * 16); pt_in(&H2, h2 + (size_t)i * 16);
    pt_mul(&A1, S, &G1); pt_mul(&A2, S, &G2);
    const pt_t* hp[6] = {&G1, &H1, &G2, &H2, &A1, &A2};
    hash_points_scalar(e, hp, 6);
    mpz_mul(z, e, X); mpz_add(z, z, S); sc_mod(z);
    pt_out(a1_out + (size_t)i * 16, &A1); pt_out(a2_out + (size_t)i * 16, &A2); zout(z_out + (size_t)i * 8, 8, z);
  }
  pt_clear(&G1); pt_clear(&H1); pt_clear(&G2); pt_clear(&H2); pt_clear(&A1); pt_clear(&A2); mpz_clears(X, S, e, z, NULL);
}
static int ecddh_verify_one(const pt_t* G1, const pt_t* H1, const pt_t* G2, const pt_t* H2, const pt_t* A1, const pt_t* A2, const mpz_t z) {
  mpz_t e; mpz_init(e);
  pt_t l, r, t; pt_init(&l); pt_init(&r); pt_init(&t);
  const pt_t* hp[6] = {G1, H1, G2, H2, A1, A2};
  hash_points_scalar(e, hp, 6);
  pt_mul(&l, z, G1); pt_mul(&t, e, H1); pt_add(&r, A1, &t);
  int ok = pt_eq(&l, &r);
  pt_mul(&l, z, G2); pt_mul(&t, e, H2); pt_add(&r, A2, &t);
  ok = ok && pt_eq(&l, &r);
  pt_clear(&l); pt_clear(&r); pt_clear(&t); mpz_clear(e);
  return ok;
}
void orc_ecddh_verify(int batch, const uint32_t* g1, const uint32_t* h1, const uint32_t* g2, const uint32_t* h2, const uint32_t* a1,
                      const uint32_t* a2, const uint32_t* z, uint8_t* ok) {
  mpz_t Z; mpz_init(Z);
  pt_t G1, H1, G2, H2, A1, A2; pt_init(&G1); pt_init(&H1); pt_init(&G2); pt_init(&H2); pt_init(&A1); pt_init(&A2);
  for (int i = 0; i < batch; ++i) {
    pt_in(&G1, g1 + (size_t)i * 16); pt_in(&H1, h1 + (size_t)i * 16); pt_in(&G2, g2 + (size_t)i * 16); pt_in(&H2, h2 + (size_t)i * 16);
    pt_in(&A1, a1 + (size_t)i * 16); pt_in(&A2, a2 + (size_t)i * 16); zin(Z, z + (size_t)i * 8, 8);
    ok[i] = (uint8_t)ecddh_verify_one(&G1, &H1, &G2, &H2, &A1, &A2, Z);
  }
  pt_clear(&G1); pt_clear(&H1); pt_clear(&G2); pt_clear(&H2); pt_clear(&A1); pt_clear(&A2); mpz_clear(Z);
}
/* Paillier::open (kzen-paillier; blame.rs:252-256): m = Dec(c), r = (c (1 - m N) mod N^2)^(N^-1 mod phi) mod N — the unique r in [0, N)
 * with c = (1 + m N) r^N mod N^2 */
void orc_paillier_open(int batch, int nkeys, const uint32_t* p, const uint32_t* q, const int32_t* key_idx, const uint32_t* c, uint32_t* m,
                       uint32_t* r) {
  mpz_t P, Q, N, NN, phi, d, C, M, t; mpz_inits(P, Q, N, NN, phi, d, C, M, t, NULL);
  for (int i = 0; i < batch; ++i) {
    const int k = pick(key_idx, nkeys, i);
    zin(P, p + (size_t)k * 32, 32); zin(Q, q + (size_t)k * 32, 32); zin(C, c + (size_t)i * 128, 128);
    mpz_mul(N, P, Q); mpz_mul(NN, N, N);
    paillier_dec(M, P, Q, C);
    mpz_mul(t, M, N); mpz_ui_sub(t, 1, t); mpz_mod(t, t, NN);
    mpz_mul(t, t, C); mpz_mod(t, t, NN); mpz_mod(t, t, N);
    mpz_sub_ui(P, P, 1); mpz_sub_ui(Q, Q, 1); mpz_mul(phi, P, Q);
    mpz_invert(d, N, phi);
    mpz_powm(t, t, d, N);
    zout(m + (size_t)i * 64, 64, M); zout(r + (size_t)i * 64, 64, t);
  }
  mpz_clears(P, Q, N, NN, phi, d, C, M, t, NULL);
}

/* GlobalStatePhase5::phase5_blame (blame.rs:116-224) */
void orc_gg20_blame5(const orc_gg20_keys* K, const int32_t* keyset, int B, const orc_blame5_in* in, uint32_t* bad_out) {
  const int S = K->S, n = K->n, P1 = S - 1;
  ec_setup();
  mpz_t t, u, al, be, kk, g; mpz_inits(t, u, al, be, kk, g, NULL);
  pt_t G, a, b_; pt_init(&G); pt_init(&a); pt_init(&b_); pt_gen(&G);
  orc_gg20_party fake; memset(&fake, 0, sizeof fake); fake.K = *K; fake.keyset = keyset;
  for (int b = 0; b < B; ++b) {
    const gg_kv kv = gg_keys_of(&fake, b);
    uint32_t bad = 0;
    mpz_t alpha[GG_MAXS][GG_MAXS], beta[GG_MAXS][GG_MAXS];
    for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) { mpz_init(alpha[i][j]); mpz_init(beta[i][j]); }
    for (int i = 0; i < S; ++i) {                                       /* commitment to g_gamma :121-125 */
      zin(g, in->gamma + ((size_t)b * S + i) * 8, 8);
      pt_mul(&a, g, &G); pt_in(&b_, in->g_gamma + ((size_t)b * S + i) * 16);
      if (!pt_eq(&a, &b_)) bad |= 1u << i;
    }
    for (int i = 0; i < S; ++i) {
      uint32_t Nw[64], k64[64] = {0}, ca[128];
      gg_N_words(&kv, K->signers[i], Nw);
      memcpy(k64, in->k + ((size_t)b * S + i) * 8, 32);
      orc_paillier_encrypt(1, 1, Nw, NULL, k64, in->k_rand + ((size_t)b * S + i) * 64, ca);          /* message a :128-138 */
      if (!words_eq(ca, in->c_a + ((size_t)b * S + i) * 128, 128)) bad |= 1u << i;
      if (bad) continue;                                                                              /* :140 */
      zin(kk, in->k + ((size_t)b * S + i) * 8, 8);
      for (int j = 0; j < P1; ++j) {
        const int ind = ind_of(i, j);
        const size_t ix = ((size_t)b * S + i) * P1 + j;
        uint32_t cbt[128], bca[128], cb[128], g64[64] = {0};
        memcpy(g64, in->gamma + ((size_t)b * S + ind) * 8, 32);
        orc_paillier_encrypt(1, 1, Nw, NULL, in->beta_tag + ix * 64, in->beta_rand + ix * 64, cbt);   /* MessageB::b_with_predefined_randomness :144-152 */
        orc_paillier_mul(1, 1, Nw, NULL, ca, g64, bca);
        orc_paillier_add(1, 1, Nw, NULL, bca, cbt, cb);
        if (!words_eq(cb, in->c_b + ix * 128, 128)) bad |= 1u << ind;                                  /* :154-156 */
        zin(be, in->beta_tag + ix * 64, 64); sc_mod(be); mpz_neg(be, be); sc_mod(be);                  /* beta = -beta_tag */
        zin(g, in->gamma + ((size_t)b * S + ind) * 8, 8);
        mpz_mul(al, kk, g); mpz_sub(al, al, be); sc_mod(al);                                           /* alpha = k_i gamma_j - beta :158-159 */
        mpz_set(alpha[i][j], al); mpz_set(beta[i][j], be);
      }
    }
    if (!bad) {                                                                                        /* :181-211 */
      for (int i = 0; i < S; ++i) {
        zin(kk, in->k + ((size_t)b * S + i) * 8, 8); zin(g, in->gamma + ((size_t)b * S + i) * 8, 8);
        mpz_mul(t, kk, g);
        for (int j = 0; j < P1; ++j) {
          const int ind1 = ind_of(i, j), ind2 = j < i ? i - 1 : i;
          mpz_add(t, t, alpha[i][j]); mpz_add(t, t, beta[ind1][ind2]);
        }
        sc_mod(t);
        zin(u, in->delta + ((size_t)b * S + i) * 8, 8); sc_mod(u);
        if (mpz_cmp(t, u) != 0) bad |= 1u << i;
      }
    }
    for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) { mpz_clear(alpha[i][j]); mpz_clear(beta[i][j]); }
    bad_out[b] = bad;
  }
  pt_clear(&G); pt_clear(&a); pt_clear(&b_); mpz_clears(t, u, al, be, kk, g, NULL);
  (void)n;
}

/* GlobalStatePhase6::phase6_blame (blame.rs:322-421) */
void orc_gg20_blame6(const orc_gg20_keys* K, const int32_t* keyset, int B, const orc_blame6_in* in, uint32_t* bad_out) {
  const int S = K->S, P1 = S - 1;
  ec_setup();
  mpz_t kk, t, lam; mpz_inits(kk, t, lam, NULL);
  pt_t G, X, gw[GG_MAXS], gni[GG_MAXS][GG_MAXS], gs, tmp, tmp2, R, Sp, A1, A2;
  pt_init(&G); pt_init(&X); pt_init(&gs); pt_init(&tmp); pt_init(&tmp2); pt_init(&R); pt_init(&Sp); pt_init(&A1); pt_init(&A2); pt_gen(&G);
  for (int i = 0; i < GG_MAXS; ++i) { pt_init(&gw[i]); for (int j = 0; j < GG_MAXS; ++j) pt_init(&gni[i][j]); }
  orc_gg20_party fake; memset(&fake, 0, sizeof fake); fake.K = *K; fake.keyset = keyset;
  for (int b = 0; b < B; ++b) {
    const gg_kv kv = gg_keys_of(&fake, b);
    uint32_t bad = 0;
    for (int i = 0; i < S; ++i) {                                         /* correctness of miu :327-339 */
      uint32_t Nw[64], c[128];
      gg_N_words(&kv, K->signers[i], Nw);
      for (int j = 0; j < P1; ++j) {
        const size_t ix = ((size_t)b * S + i) * P1 + j;
        orc_paillier_encrypt(1, 1, Nw, NULL, in->miu + ix * 64, in->miu_rand + ix * 64, c);
        if (!words_eq(c, in->c_b + ix * 128, 128)) bad |= 1u << i;
      }
    }
    for (int i = 0; i < S; ++i) {                                         /* correctness of k :342-354 */
      uint32_t Nw[64], k64[64] = {0}, ca[128];
      gg_N_words(&kv, K->signers[i], Nw);
      memcpy(k64, in->k + ((size_t)b * S + i) * 8, 32);
      orc_paillier_encrypt(1, 1, Nw, NULL, k64, in->k_rand + ((size_t)b * S + i) * 64, ca);
      if (!words_eq(ca, in->c_a + ((size_t)b * S + i) * 128, 128)) bad |= 1u << i;
    }
    if (!bad) {
      for (int i = 0; i < S; ++i) {                                       /* g_w_vec as in SignKeys::g_w_vec */
        lagrange_at_zero(lam, K->signers, S, i);
        pt_in(&X, kv.X + (size_t)K->signers[i] * 16);
        pt_mul(&gw[i], lam, &X);
      }
      for (int i = 0; i < S; ++i) {                                       /* g_ni :360-376 */
        zin(kk, in->k + ((size_t)b * S + i) * 8, 8);
        for (int j = 0; j < P1; ++j) {
          const int ind = ind_of(i, j);
          pt_mul(&tmp, kk, &gw[ind]);
          zin(t, in->miu + (((size_t)b * S + i) * P1 + j) * 64, 64);
          pt_mul(&tmp2, t, &G); pt_neg(&tmp2, &tmp2);
          pt_add(&gni[i][j], &tmp, &tmp2);
        }
      }
      pt_in(&R, in->R + (size_t)b * 16);
      for (int i = 0; i < S; ++i) {                                       /* g_sigma_i :380-397, ECDDH proof :400-414 */
        zin(kk, in->k + ((size_t)b * S + i) * 8, 8);
        pt_mul(&gs, kk, &gw[i]);
        for (int j = 0; j < P1; ++j) {
          zin(t, in->miu + (((size_t)b * S + i) * P1 + j) * 64, 64);
          pt_mul(&tmp, t, &G); pt_add(&gs, &gs, &tmp);
        }
        for (int j = 0; j < P1; ++j) {
          const int ind1 = ind_of(i, j), ind2 = j < i ? i - 1 : i;
          pt_add(&gs, &gs, &gni[ind1][ind2]);
        }
        pt_in(&Sp, in->S + ((size_t)b * S + i) * 16); pt_in(&A1, in->a1 + ((size_t)b * S + i) * 16); pt_in(&A2, in->a2 + ((size_t)b * S + i) * 16);
        zin(t, in->z + ((size_t)b * S + i) * 8, 8);
        if (!ecddh_verify_one(&G, &gs, &R, &Sp, &A1, &A2, t)) bad |= 1u << i;
      }
    }
    bad_out[b] = bad;
  }
  for (int i = 0; i < GG_MAXS; ++i) { pt_clear(&gw[i]); for (int j = 0; j < GG_MAXS; ++j) pt_clear(&gni[i][j]); }
  pt_clear(&G); pt_clear(&X); pt_clear(&gs); pt_clear(&tmp); pt_clear(&tmp2); pt_clear(&R); pt_clear(&Sp); pt_clear(&A1); pt_clear(&A2);
  mpz_clears(kk, t, lam, NULL);
}

/* GlobalStatePhase7::phase7_blame (blame.rs:434-454): R s_i == m R_dash_i + r S_i */
void orc_gg20_blame7(int S, int B, const orc_blame7_in* in, uint32_t* bad_out) {
  ec_setup();
  mpz_t t; mpz_init(t);
  pt_t R, l, a, b_, p; pt_init(&R); pt_init(&l); pt_init(&a); pt_init(&b_); pt_init(&p);
  for (int b = 0; b < B; ++b) {
    uint32_t bad = 0;
    pt_in(&R, in->R + (size_t)b * 16);
    for (int i = 0; i < S; ++i) {
      zin(t, in->s + ((size_t)b * S + i) * 8, 8); pt_mul(&l, t, &R);
      pt_in(&p, in->R_dash + ((size_t)b * S + i) * 16); zin(t, in->m + (size_t)b * 8, 8); pt_mul(&a, t, &p);
      pt_in(&p, in->S + ((size_t)b * S + i) * 16); zin(t, in->r + (size_t)b * 8, 8); pt_mul(&b_, t, &p);
      pt_add(&a, &a, &b_);
      if (!pt_eq(&l, &a)) bad |= 1u << i;
    }
    bad_out[b] = bad;
  }
  pt_clear(&R); pt_clear(&l); pt_clear(&a); pt_clear(&b_); pt_clear(&p); mpz_clear(t);
}
/* read-back of a party's state for the blame protocols (what LocalStatePhase5/6 publish): sigma_i, the plaintexts miu_ij before
 * reduction are NOT kept by the party object; tests recompute them with orc_paillier_open / decrypt */
void orc_gg20_party_sigma(const orc_gg20_party* P, uint32_t* sigma /*[B][8]*/) {
  for (int b = 0; b < P->B; ++b) memcpy(sigma + (size_t)b * 8, P->s[b].sigma_i, 32);
}

/* ==========================================================================================================================
 * Keygen VERIFICATION math: what every party checks about every other party's first keygen messages and
 * shares — src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:260-320 (phase1_verify_com_phase3_verify_correct_key_verify_dlog…),
 * :322-367 (phase2_verify_vss…), :405-438 (verify_dlog_proofs_check_against_vss).  The two zk-paillier 0.4.3 proofs are
 * un-vendored; their definitions below are RECALLED — parity unpinned like the rest of this oracle.
 * ========================================================================================================================== */
/* zk-paillier `compute_digest`: SHA-256 over the minimal big-endian bytes of every value, digest as BigInt */
static void zkp_digest(mpz_t out, const mpz_t* vals, int n) {
  sha_t sh; sha_init(&sh);
  for (int i = 0; i < n; ++i) chain_bigint(&sh, vals[i]);
  result_bigint(&sh, out);
}
/* e = H(x, g, N, ni) of CompositeDLogProof; v = the canonical list, hashed in ORC_ENC.ord_cdlog order */
static void cdlog_digest(mpz_t e, const mpz_t* v) {
  sha_t sh; sha_init(&sh);
  for (int i = 0; i < 4; ++i) chain_bigint(&sh, v[ORC_ENC.ord_cdlog[i] & 3]);
  result_bigint(&sh, e);
}
/* CompositeDLogProof{x, y}::verify(statement{N, g, ni}) (zk-paillier composite_dlog_proof.rs): N >= 2^128, gcd(g, N) = gcd(ni, N) = 1,
 * e = H(x, g, N, ni), x == g^y ni^e mod N.   y: [B][73] (r < 2^512, e 256 bit, secret < phi: y < 2^2306) */
void orc_composite_dlog_verify(int batch, const uint32_t* N, const uint32_t* g, const uint32_t* ni, const uint32_t* x, const uint32_t* y,
                               uint8_t* ok) {
  mpz_t v[4], Y, e, a, b, lim; mpz_inits(v[0], v[1], v[2], v[3], Y, e, a, b, lim, NULL);
  mpz_ui_pow_ui(lim, 2, 128);
  for (int i = 0; i < batch; ++i) {
    zin(v[0], x + (size_t)i * 64, 64); zin(v[1], g + (size_t)i * 64, 64); zin(v[2], N + (size_t)i * 64, 64); zin(v[3], ni + (size_t)i * 64, 64);
    zin(Y, y + (size_t)i * 73, 73);
    ok[i] = 0;
    if (mpz_cmp(v[2], lim) < 0 || mpz_even_p(v[2])) continue;
    mpz_gcd(a, v[1], v[2]); if (mpz_cmp_ui(a, 1) != 0) continue;
    mpz_gcd(a, v[3], v[2]); if (mpz_cmp_ui(a, 1) != 0) continue;
    cdlog_digest(e, (const mpz_t*)v);
    mpz_powm(a, v[1], Y, v[2]); mpz_powm(b, v[3], e, v[2]);
    mpz_mul(a, a, b); mpz_mod(a, a, v[2]);
    ok[i] = (uint8_t)(mpz_cmp(a, v[0]) == 0);
  }
  mpz_clears(v[0], v[1], v[2], v[3], Y, e, a, b, lim, NULL);
}
/* CompositeDLogProof::prove with the nonce r (< 2^512) as input: x = g^r mod N, e = H(x, g, N, ni), y = r + e secret  (fixtures) */
void orc_composite_dlog_prove(int batch, const uint32_t* N, const uint32_t* g, const uint32_t* ni, const uint32_t* secret, const uint32_t* r,
                              uint32_t* x, uint32_t* y) {
  mpz_t v[4], S, R, e; mpz_inits(v[0], v[1], v[2], v[3], S, R, e, NULL);
  for (int i = 0; i < batch; ++i) {
    zin(v[1], g + (size_t)i * 64, 64); zin(v[2], N + (size_t)i * 64, 64); zin(v[3], ni + (size_t)i * 64, 64);
    zin(S, secret + (size_t)i * 64, 64); zin(R, r + (size_t)i * 16, 16);
    mpz_powm(v[0], v[1], R, v[2]);
    cdlog_digest(e, (const mpz_t*)v);
    mpz_mul(e, e, S); mpz_add(e, e, R);
    zout(x + (size_t)i * 64, 64, v[0]); zout(y + (size_t)i * 73, 73, e);
  }
  mpz_clears(v[0], v[1], v[2], v[3], S, R, e, NULL);
}
/* NiCorrectKeyProof (zk-paillier correct_key_ni.rs): M2 = 11 values sigma_i; rho_i = mask_generation(bit_length(N), H(N, salt, i)) mod N with
 * mask_generation(len, seed) = sum_j H(seed, j) << (256 j), j < len/256 + 1; verify: sigma_i^N == rho_i mod N for all i and
 * gcd(N, primorial(6370)) == 1 (no prime below 6370 divides N).  salt = SALT_STRING = "KZen" as BigInt. */
#define ORC_CK_M2 11
static void correct_key_rho(mpz_t rho, const mpz_t N, int i) {
  mpz_t v[3], seed, d, acc, w[2]; mpz_inits(v[0], v[1], v[2], seed, d, acc, w[0], w[1], NULL);
  mpz_set(v[0], N); mpz_set_ui(v[1], (unsigned long)ORC_ENC.ck_salt); mpz_set_ui(v[2], (unsigned long)i);   /* salt: b"KZen" as a BigInt */
  zkp_digest(seed, (const mpz_t*)v, 3);
  const int msklen = (int)(mpz_sizeinbase(N, 2) / 256) + 1;
  mpz_set_ui(acc, 0);
  for (int j = 0; j < msklen; ++j) {
    mpz_set(w[0], seed); mpz_set_ui(w[1], (unsigned long)j);
    zkp_digest(d, (const mpz_t*)w, 2);
    mpz_mul_2exp(d, d, 256u * (unsigned)(ORC_ENC.ck_mask_order ? msklen - 1 - j : j)); mpz_add(acc, acc, d);
  }
  mpz_mod(rho, acc, N);
  mpz_clears(v[0], v[1], v[2], seed, d, acc, w[0], w[1], NULL);
}
static int no_small_factor(const mpz_t N) {          /* gcd(N, prod of the primes < 6370) == 1 */
  for (unsigned long p = 2; p < 6370; ++p) {
    int prime = 1;
    for (unsigned long d = 2; d * d <= p; ++d) if (p % d == 0) { prime = 0; break; }
    if (prime && mpz_divisible_ui_p(N, p)) return 0;
  }
  return 1;
}
void orc_correct_key_verify(int batch, const uint32_t* N, const uint32_t* sigma /*[B][11][64]*/, uint8_t* ok) {
  mpz_t n, s, rho; mpz_inits(n, s, rho, NULL);
  for (int b = 0; b < batch; ++b) {
    zin(n, N + (size_t)b * 64, 64);
    int good = mpz_cmp_ui(n, 1) > 0 && no_small_factor(n);
    for (int i = 0; i < ORC_CK_M2 && good; ++i) {
      correct_key_rho(rho, n, i);
      zin(s, sigma + ((size_t)b * ORC_CK_M2 + i) * 64, 64);
      mpz_powm(s, s, n, n);
      good = mpz_cmp(s, rho) == 0;
    }
    ok[b] = (uint8_t)good;
  }
  mpz_clears(n, s, rho, NULL);
}
/* NiCorrectKeyProof::proof (fixtures): sigma_i = rho_i^(N^-1 mod phi) mod N */
void orc_correct_key_prove(int batch, const uint32_t* p, const uint32_t* q, uint32_t* sigma) {
  mpz_t P, Q, n, phi, d, rho; mpz_inits(P, Q, n, phi, d, rho, NULL);
  for (int b = 0; b < batch; ++b) {
    zin(P, p + (size_t)b * 32, 32); zin(Q, q + (size_t)b * 32, 32);
    mpz_mul(n, P, Q); mpz_sub_ui(P, P, 1); mpz_sub_ui(Q, Q, 1); mpz_mul(phi, P, Q);
    mpz_invert(d, n, phi);
    for (int i = 0; i < ORC_CK_M2; ++i) {
      correct_key_rho(rho, n, i);
      mpz_powm(rho, rho, d, n);
      zout(sigma + ((size_t)b * ORC_CK_M2 + i) * 64, 64, rho);
    }
  }
  mpz_clears(P, Q, n, phi, d, rho, NULL);
}
/* Feldman VSS (curv VerifiableSS): validate_share(share, index): share G == sum_k index^k C_k; get_point_commitment(index) = that sum.
 * commitments: [B][t+1][16]; index: [B] (1-based party index). */
static void vss_point(pt_t* out, const uint32_t* commits, int t1, unsigned long index) {
  pt_t c; pt_init(&c);
  mpz_t idx; mpz_init_set_ui(idx, index);
  out->inf = 1;
  for (int k = t1 - 1; k >= 0; --k) {                 /* Horner: acc = acc * index + C_k */
    pt_mul(out, idx, out);
    pt_in(&c, commits + (size_t)k * 16);
    pt_add(out, out, &c);
  }
  pt_clear(&c); mpz_clear(idx);
}
void orc_vss_validate_share(int batch, int t1, const uint32_t* commits, const uint32_t* share, const int32_t* index, uint8_t* ok) {
  ec_setup();
  mpz_t s; mpz_init(s);
  pt_t G, l, r; pt_init(&G); pt_init(&l); pt_init(&r); pt_gen(&G);
  for (int b = 0; b < batch; ++b) {
    zin(s, share + (size_t)b * 8, 8);
    pt_mul(&l, s, &G);
    vss_point(&r, commits + (size_t)b * t1 * 16, t1, (unsigned long)index[b]);
    ok[b] = (uint8_t)pt_eq(&l, &r);
  }
  pt_clear(&G); pt_clear(&l); pt_clear(&r); mpz_clear(s);
}
void orc_vss_point_commitment(int batch, int t1, const uint32_t* commits, const int32_t* index, uint32_t* out) {
  ec_setup();
  pt_t r; pt_init(&r);
  for (int b = 0; b < batch; ++b) {
    vss_point(&r, commits + (size_t)b * t1 * 16, t1, (unsigned long)index[b]);
    pt_out(out + (size_t)b * 16, &r);
  }
  pt_clear(&r);
}

/* ---- the two keygen verdicts as the reference composes them (gg_2020/party_i.rs:260-367) --------------------------------
 * items = (keygen session, prover i), n_parties consecutive items per session; bad [batch / n_parties]: bit i = prover i of the
 * session is in `bad_actors`.  Row widths as in orc_correct_key_verify / orc_composite_dlog_verify / orc_hash_commit_point. */
#define ORC_PAILLIER_MIN_BIT_LENGTH 2047      /* party_i.rs:49 */
#define ORC_PAILLIER_MAX_BIT_LENGTH 2048      /* party_i.rs:50 */
void orc_keygen_verify_round1(int batch, int n_parties, const uint32_t* y, const uint32_t* blind, const uint32_t* com, const uint32_t* N,
                              const uint32_t* sigma, const uint32_t* Nt, const uint32_t* h1, const uint32_t* h2, const uint32_t* x_h1,
                              const uint32_t* y_h1, const uint32_t* x_h2, const uint32_t* y_h2, uint8_t* ok, uint32_t* bad) {
  mpz_t n, nt; mpz_inits(n, nt, NULL);
  if (bad) memset(bad, 0, (size_t)(batch / n_parties) * 4);
  for (int i = 0; i < batch; ++i) {
    uint32_t c[8];
    uint8_t ck = 0, cd1 = 0, cd2 = 0;
    orc_hash_commit_point(1, y + (size_t)i * 16, blind + (size_t)i * 8, c);                                  /* :278-283 */
    orc_correct_key_verify(1, N + (size_t)i * 64, sigma + (size_t)i * 11 * 64, &ck);                         /* :284-287 */
    zin(n, N + (size_t)i * 64, 64); zin(nt, Nt + (size_t)i * 64, 64);
    orc_composite_dlog_verify(1, Nt + (size_t)i * 64, h1 + (size_t)i * 64, h2 + (size_t)i * 64, x_h1 + (size_t)i * 64, y_h1 + (size_t)i * 73, &cd1);   /* :292-295 */
    /* dlog_statement_base_h2 = { N, g: ni, ni: g } (:271-275) */
    orc_composite_dlog_verify(1, Nt + (size_t)i * 64, h2 + (size_t)i * 64, h1 + (size_t)i * 64, x_h2 + (size_t)i * 64, y_h2 + (size_t)i * 73, &cd2);   /* :296-299 */
    const size_t bn = mpz_sgn(n) ? mpz_sizeinbase(n, 2) : 0, bt = mpz_sgn(nt) ? mpz_sizeinbase(nt, 2) : 0;
    const int test_res = memcmp(c, com + (size_t)i * 8, 32) == 0 && ck &&
                         bn >= ORC_PAILLIER_MIN_BIT_LENGTH && bn <= ORC_PAILLIER_MAX_BIT_LENGTH &&             /* :288-289 */
                         bt >= ORC_PAILLIER_MIN_BIT_LENGTH && bt <= ORC_PAILLIER_MAX_BIT_LENGTH &&             /* :290-291 */
                         cd1 && cd2;
    ok[i] = test_res ? 1 : 0;
    if (!test_res && bad) bad[i / n_parties] |= 1u << (i % n_parties);                                        /* :300-302 bad_actors_vec.push(i) */
  }
  mpz_clears(n, nt, NULL);
}
/* phase2_verify_vss_construct_keypair_phase3_pok_dlog, the verdict (:337-349) */
void orc_keygen_verify_round2(int batch, int n_parties, int t1, const uint32_t* commits, const uint32_t* share, const int32_t* index,
                              const uint32_t* y, uint8_t* ok, uint32_t* bad) {
  if (bad) memset(bad, 0, (size_t)(batch / n_parties) * 4);
  for (int i = 0; i < batch; ++i) {
    uint8_t v = 0;
    orc_vss_validate_share(1, t1, commits + (size_t)i * t1 * 16, share + (size_t)i * 8, index + i, &v);      /* :338-340 */
    const int res = v && memcmp(commits + (size_t)i * t1 * 16, y + (size_t)i * 16, 64) == 0;                  /* :341 commitments[0] == y_vec[i] */
    ok[i] = res ? 1 : 0;
    if (!res && bad) bad[i / n_parties] |= 1u << (i % n_parties);
  }
}
