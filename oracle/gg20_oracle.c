/* gg20_oracle.c — CPU ORACLE (test infrastructure, NOT product code): one party-complete GG20
 * signing session, all parties simulated in lock-step the way `round_based::dev::Simulation` does in
 * the reference's own test (src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign.rs:667-763),
 * with every sampled value passed in.  Follows
 *   src/protocols/multi_party_ecdsa/gg_2020/state_machine/sign/rounds.rs:67-692  (Round0..Round7)
 *   src/protocols/multi_party_ecdsa/gg_2020/party_i.rs:526-936                   (SignKeys, LocalSignature)
 *   src/utilities/mta/mod.rs:52-179                                              (MessageA / MessageB)
 * and, for the un-vendored curv sigma proofs, SURVEY.md App. A.3 (PedersenProof, HomoELGamalProof,
 * HashCommitment, VerifiableSS::map_share_to_new_params).  PARITY UNPINNED — see mpe_oracle.h.
 *
 * Compiled into libmpe_oracle.so by #include from mpe_oracle.c (shares its static helpers). */

/* ---- layouts ------------------------------------------------------------------------------ */
/* keys: n parties; signers: S ascending indices into 0..n-1 */
/* pair index pp = i*(S-1) + jj  (i = sender/owner signer ordinal, jj = peer ordinal, ind = jj<i ? jj : jj+1)
 * (the `ind` convention of rounds.rs:149,261,464) */

static void sc_mod(mpz_t r) { mpz_mod(r, r, EC_Q); }

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
static void hash_points_scalar(mpz_t out, const pt_t** pts, int n) {
  sha_t sh; sha_init(&sh);
  for (int i = 0; i < n; ++i) chain_point(&sh, pts[i]);
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

#define MAXS 8
#define MAXN 8

/* returns 0 on success, otherwise 100*round + detail */
static int gg20_sign_one(const orc_gg20_keys* K, const orc_gg20_nonces* Z, int sess, uint32_t* r_out, uint32_t* s_out,
                         int32_t* recid_out, uint32_t* R_out) {
  const int S = K->S, n = K->n, P = S * (S - 1);
  if (S > MAXS || n > MAXN) return 1;
  int rc = 0;
  ec_setup();
  /* per-session slices */
  const uint32_t* k_w = Z->k + (size_t)sess * S * 8;
  const uint32_t* gam_w = Z->gamma + (size_t)sess * S * 8;
  const uint32_t* blind_w = Z->blind + (size_t)sess * S * 8;
  const uint32_t* ra_w = Z->r_a + (size_t)sess * S * 64;
  const size_t an = (size_t)sess * S * n;
  const size_t pb = (size_t)sess * P * 2, pp0 = (size_t)sess * P;

  mpz_t N[MAXN], NN[MAXN], pw[MAXN], qw[MAXN];
  for (int a = 0; a < n; ++a) {
    mpz_inits(N[a], NN[a], pw[a], qw[a], NULL);
    zin(pw[a], K->p + (size_t)a * 32, 32); zin(qw[a], K->q + (size_t)a * 32, 32);
    mpz_mul(N[a], pw[a], qw[a]); mpz_mul(NN[a], N[a], N[a]);
  }
  /* tables as words for the batch helpers of mpe_oracle.c */
  uint32_t Nw[MAXN][64];
  for (int a = 0; a < n; ++a) zout(Nw[a], 64, N[a]);

  mpz_t k[MAXS], gam[MAXS], w[MAXS], blind[MAXS], com[MAXS], ca[MAXS], ra[MAXS], delta_i[MAXS], sigma_i[MAXS], l[MAXS], lam, t, t2, e;
  pt_t G, H2, g_gamma[MAXS], g_w[MAXS], Tpt[MAXS], Rbar[MAXS], Spt[MAXS], R, tmp, tmp2, tmp3;
  mpz_inits(lam, t, t2, e, NULL);
  pt_init(&G); pt_init(&H2); pt_init(&R); pt_init(&tmp); pt_init(&tmp2); pt_init(&tmp3);
  pt_gen(&G); pt_h2(&H2);
  for (int i = 0; i < S; ++i) {
    mpz_inits(k[i], gam[i], w[i], blind[i], com[i], ca[i], ra[i], delta_i[i], sigma_i[i], l[i], NULL);
    pt_init(&g_gamma[i]); pt_init(&g_w[i]); pt_init(&Tpt[i]); pt_init(&Rbar[i]); pt_init(&Spt[i]);
  }
  /* MessageB state: [i][jj][v] */
  static __thread uint32_t cb[MAXS][MAXS][2][128];
  mpz_t beta[MAXS][MAXS][2];
  pt_t Bpk[MAXS][MAXS][2], BR[MAXS][MAXS][2], BTpk[MAXS][MAXS][2], BTR[MAXS][MAXS][2];
  mpz_t Bz[MAXS][MAXS][2], BTz[MAXS][MAXS][2];
  for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) for (int v = 0; v < 2; ++v) {
    mpz_inits(beta[i][j][v], Bz[i][j][v], BTz[i][j][v], NULL);
    pt_init(&Bpk[i][j][v]); pt_init(&BR[i][j][v]); pt_init(&BTpk[i][j][v]); pt_init(&BTR[i][j][v]);
  }
  /* Alice proofs [i][st] */
  static __thread uint32_t az[MAXS][MAXN][64], ae[MAXS][MAXN][8], as_[MAXS][MAXN][64], as1[MAXS][MAXN][25], as2[MAXS][MAXN][89];
  static __thread uint32_t caw[MAXS][128];

  /* ---------------- Round 0 (rounds.rs:68-104): SignKeys::create, phase1_broadcast, MessageA::a ---------------- */
  for (int i = 0; i < S; ++i) {
    const int me = K->signers[i];
    zin(k[i], k_w + i * 8, 8); sc_mod(k[i]);
    zin(gam[i], gam_w + i * 8, 8); sc_mod(gam[i]);
    zin(blind[i], blind_w + i * 8, 8);
    zin(ra[i], ra_w + i * 64, 64);
    lagrange_at_zero(lam, K->signers, S, i);                          /* party_i.rs:553-557 */
    zin(t, K->x + (size_t)me * 8, 8);
    mpz_mul(w[i], lam, t); sc_mod(w[i]);                               /* w_i = li * x_i :558 */
    pt_mul(&g_gamma[i], gam[i], &G);                                   /* :562 */
    hash_commit_point(com[i], &g_gamma[i], blind[i]);                  /* phase1_broadcast :573-589 */
    paillier_enc(ca[i], N[me], NN[me], k[i], ra[i]);                   /* MessageA::a_with_predefined_randomness mta/mod.rs:68-75 */
    zout(caw[i], 128, ca[i]);
    for (int st = 0; st < n; ++st) {                                   /* :76-81, all n statements (rounds.rs:87) */
      const size_t ix = an + (size_t)i * n + st;
      orc_alice_generate(1, 1, Nw[me], 1, K->Nt + (size_t)st * 64, K->h1 + (size_t)st * 64, K->h2 + (size_t)st * 64, NULL, NULL,
                         k_w + i * 8, caw[i], ra_w + i * 64, Z->al_alpha + ix * 24, Z->al_beta + ix * 64, Z->al_gamma + ix * 88,
                         Z->al_rho + ix * 72, az[i][st], ae[i][st], as_[i][st], as1[i][st], as2[i][st]);
    }
  }
  /* g_w_vec as every party recomputes it in Round2 (party_i.rs:527-544): lambda_j * X_j */
  for (int i = 0; i < S; ++i) {
    lagrange_at_zero(lam, K->signers, S, i);
    pt_in(&tmp, K->X + (size_t)K->signers[i] * 16);
    pt_mul(&g_w[i], lam, &tmp);          /* what the peers hold for signer i: from pk_vec, NOT from its secret share */
  }

  /* ---------------- Round 1 (rounds.rs:122-206): MessageB::b for gamma_i and w_i towards every peer ---------------- */
  for (int i = 0; i < S; ++i) {
    for (int jj = 0; jj < S - 1; ++jj) {
      const int ind = jj < i ? jj : jj + 1, alice = K->signers[ind];
      for (int v = 0; v < 2; ++v) {
        /* verify Alice's n range proofs (mta/mod.rs:119-131); executed for both calls as the reference does */
        for (int st = 0; st < n; ++st) {
          uint8_t ok = 0;
          orc_alice_verify(1, 1, Nw[alice], 1, K->Nt + (size_t)st * 64, K->h1 + (size_t)st * 64, K->h2 + (size_t)st * 64, NULL, NULL,
                           caw[ind], az[ind][st], ae[ind][st], as_[ind][st], as1[ind][st], as2[ind][st], &ok);
          if (!ok) { rc = 101; goto done; }
        }
        const size_t ix = pb + ((size_t)i * (S - 1) + jj) * 2 + v;
        mpz_t bt, rr, cbt, bca; mpz_inits(bt, rr, cbt, bca, NULL);
        zin(bt, Z->mb_beta_tag + ix * 64, 64);
        zin(rr, Z->mb_r + ix * 64, 64);
        paillier_enc(cbt, N[alice], NN[alice], bt, rr);                 /* :133-137 */
        mpz_powm(bca, ca[ind], v == 0 ? gam[i] : w[i], NN[alice]);      /* Paillier::mul :140-144 */
        mpz_mul(bca, bca, cbt); mpz_mod(bca, bca, NN[alice]);           /* Paillier::add :145 */
        zout(cb[i][jj][v], 128, bca);
        mpz_mod(t, bt, EC_Q);                                           /* beta_tag_fe :132 */
        mpz_neg(beta[i][jj][v], t); sc_mod(beta[i][jj][v]);             /* beta = -beta_tag :146 */
        /* DLogProof::prove(b), DLogProof::prove(beta_tag_fe) :147-148 */
        uint32_t skw[8], pkw[16], Rw[16], zw[8];
        zout(skw, 8, v == 0 ? gam[i] : w[i]);
        orc_dlog_prove(1, skw, Z->mb_nonce_b + ix * 8, pkw, Rw, zw);
        pt_in(&Bpk[i][jj][v], pkw); pt_in(&BR[i][jj][v], Rw); zin(Bz[i][jj][v], zw, 8);
        zout(skw, 8, t);
        orc_dlog_prove(1, skw, Z->mb_nonce_bt + ix * 8, pkw, Rw, zw);
        pt_in(&BTpk[i][jj][v], pkw); pt_in(&BTR[i][jj][v], Rw); zin(BTz[i][jj][v], zw, 8);
        mpz_clears(bt, rr, cbt, bca, NULL);
      }
    }
  }

  /* ---------------- Round 2 (rounds.rs:234-317): verify_proofs_get_alpha, delta_i, sigma_i, T_i ---------------- */
  for (int i = 0; i < S; ++i) {
    const int me = K->signers[i];
    mpz_mul(delta_i[i], k[i], gam[i]); sc_mod(delta_i[i]);             /* phase2_delta_i :591-604 */
    mpz_mul(sigma_i[i], k[i], w[i]); sc_mod(sigma_i[i]);               /* phase2_sigma_i :606-618 */
    for (int jj = 0; jj < S - 1; ++jj) {
      const int ind = jj < i ? jj : jj + 1;
      /* the message peer `ind` sent to me: its pair ordinal for me */
      const int jme = i < ind ? i : i - 1;
      for (int v = 0; v < 2; ++v) {
        mpz_t c, m; mpz_inits(c, m, NULL);
        zin(c, cb[ind][jme][v], 128);
        paillier_dec(m, pw[me], qw[me], c);                             /* mta/mod.rs:165 */
        mpz_mod(t, m, EC_Q);                                            /* alpha :167 */
        pt_mul(&tmp, t, &G);                                            /* g_alpha :168 */
        pt_mul(&tmp2, k[i], &Bpk[ind][jme][v]); pt_add(&tmp2, &tmp2, &BTpk[ind][jme][v]);   /* ba_btag :169 */
        uint32_t pkw[16], Rw[16], zw[8]; uint8_t ok1, ok2;
        pt_out(pkw, &Bpk[ind][jme][v]); pt_out(Rw, &BR[ind][jme][v]); zout(zw, 8, Bz[ind][jme][v]);
        orc_dlog_verify(1, pkw, Rw, zw, &ok1);
        pt_out(pkw, &BTpk[ind][jme][v]); pt_out(Rw, &BTR[ind][jme][v]); zout(zw, 8, BTz[ind][jme][v]);
        orc_dlog_verify(1, pkw, Rw, zw, &ok2);
        if (!ok1 || !ok2 || !pt_eq(&tmp, &tmp2)) { mpz_clears(c, m, NULL); rc = 201; goto done; }   /* :170-177 */
        if (v == 1 && !pt_eq(&Bpk[ind][jme][1], &g_w[ind])) { mpz_clears(c, m, NULL); rc = 202; goto done; }  /* rounds.rs:281 */
        /* alpha_ij + beta_ij (my own beta from the MessageB I built for this peer) */
        mpz_add(t, t, beta[i][jj][v]);
        if (v == 0) { mpz_add(delta_i[i], delta_i[i], t); sc_mod(delta_i[i]); }
        else { mpz_add(sigma_i[i], sigma_i[i], t); sc_mod(sigma_i[i]); }
        mpz_clears(c, m, NULL);
      }
    }
    /* phase3_compute_t_i :620-634 */
    zin(l[i], Z->l + ((size_t)sess * S + i) * 8, 8); sc_mod(l[i]);
    pt_mul(&tmp, sigma_i[i], &G); pt_mul(&tmp2, l[i], &H2); pt_add(&Tpt[i], &tmp, &tmp2);
  }

  /* ---------------- Round 3 (rounds.rs:347-402): PedersenProof prove/verify, delta^-1 ---------------- */
  mpz_t dinv; mpz_init(dinv);
  mpz_set_ui(dinv, 0);
  for (int i = 0; i < S; ++i) { mpz_add(dinv, dinv, delta_i[i]); sc_mod(dinv); }
  if (!mpz_invert(dinv, dinv, EC_Q)) { rc = 301; goto done2; }         /* phase3_reconstruct_delta :635-640 */
  for (int i = 0; i < S; ++i) {
    /* prove (App. A.3): a1 = s1 g, a2 = s2 h, com = m g + r h, e = H(g,h,com,a1,a2), z1 = s1 + e m, z2 = s2 + e r */
    mpz_t s1, s2, z1, z2; mpz_inits(s1, s2, z1, z2, NULL);
    zin(s1, Z->ped_s1 + ((size_t)sess * S + i) * 8, 8); sc_mod(s1);
    zin(s2, Z->ped_s2 + ((size_t)sess * S + i) * 8, 8); sc_mod(s2);
    pt_t a1, a2; pt_init(&a1); pt_init(&a2);
    pt_mul(&a1, s1, &G); pt_mul(&a2, s2, &H2);
    const pt_t* hp[5] = {&G, &H2, &Tpt[i], &a1, &a2};                  /* com == T_i (rounds.rs:366) */
    hash_points_scalar(e, hp, 5);
    mpz_mul(z1, e, sigma_i[i]); mpz_add(z1, z1, s1); sc_mod(z1);
    mpz_mul(z2, e, l[i]); mpz_add(z2, z2, s2); sc_mod(z2);
    /* verify: z1 g + z2 h == a1 + a2 + e com  (every party verifies every proof; identical outcome) */
    pt_mul(&tmp, z1, &G); pt_mul(&tmp2, z2, &H2); pt_add(&tmp, &tmp, &tmp2);
    pt_mul(&tmp2, e, &Tpt[i]); pt_add(&tmp3, &a1, &a2); pt_add(&tmp3, &tmp3, &tmp2);
    const int okp = pt_eq(&tmp, &tmp3);
    pt_clear(&a1); pt_clear(&a2); mpz_clears(s1, s2, z1, z2, NULL);
    if (!okp) { rc = 302; goto done2; }
  }

  /* ---------------- Round 4 (rounds.rs:431-498): phase4 -> R, R_dash, PDL proofs ---------------- */
  for (int i = 0; i < S; ++i) {                                          /* phase4 :642-687, as run by party i */
    for (int jj = 0; jj < S - 1; ++jj) {
      const int ind = jj < i ? jj : jj + 1, jme = i < ind ? i : i - 1;
      hash_commit_point(t, &g_gamma[ind], blind[ind]);
      if (!pt_eq(&Bpk[ind][jme][0], &g_gamma[ind]) || mpz_cmp(t, com[ind]) != 0) { rc = 401; goto done2; }
    }
  }
  pt_set(&tmp, &g_gamma[0]);
  for (int i = 1; i < S; ++i) pt_add(&tmp, &tmp, &g_gamma[i]);
  pt_mul(&R, dinv, &tmp);                                                 /* R = (sum Gamma_i) * delta^-1 */
  /* PDL proofs [i][jj] */
  static __thread uint32_t pz[MAXS][MAXS][64], pu1[MAXS][MAXS][16], pu2[MAXS][MAXS][128], pu3[MAXS][MAXS][64], ps1[MAXS][MAXS][25],
      ps2[MAXS][MAXS][64], ps3[MAXS][MAXS][89];
  uint32_t Rw16[16], Rbw[MAXS][16];
  pt_out(Rw16, &R);
  for (int i = 0; i < S; ++i) {
    const int me = K->signers[i];
    pt_mul(&Rbar[i], k[i], &R);                                           /* R_dash = R * k_i  rounds.rs:452 */
    pt_out(Rbw[i], &Rbar[i]);
    for (int jj = 0; jj < S - 1; ++jj) {
      const int ind = jj < i ? jj : jj + 1, st = K->signers[ind];
      const size_t ix = pp0 + (size_t)i * (S - 1) + jj;
      uint32_t kw8[8]; zout(kw8, 8, k[i]);
      orc_pdl_prove(1, 1, Nw[me], 1, K->Nt + (size_t)st * 64, K->h1 + (size_t)st * 64, K->h2 + (size_t)st * 64, NULL, NULL, caw[i],
                    Rbw[i], Rw16, kw8, ra_w + i * 64, Z->pdl_alpha + ix * 24, Z->pdl_beta + ix * 64, Z->pdl_rho + ix * 72,
                    Z->pdl_gamma + ix * 88, pz[i][jj], pu1[i][jj], pu2[i][jj], pu3[i][jj], ps1[i][jj], ps2[i][jj], ps3[i][jj]);
    }
  }

  /* ---------------- Round 5 (rounds.rs:525-601): verify all PDL proofs, sum R_dash, S_i + HEG proof ---------------- */
  for (int verifier = 0; verifier < S; ++verifier) {                      /* every party verifies all S(S-1) proofs */
    for (int i = 0; i < S; ++i) {
      const int me = K->signers[i];
      for (int jj = 0; jj < S - 1; ++jj) {
        const int ind = jj < i ? jj : jj + 1, st = K->signers[ind];
        uint8_t ok = 0;
        orc_pdl_verify(1, 1, Nw[me], 1, K->Nt + (size_t)st * 64, K->h1 + (size_t)st * 64, K->h2 + (size_t)st * 64, NULL, NULL, caw[i],
                       Rbw[i], Rw16, pz[i][jj], pu1[i][jj], pu2[i][jj], pu3[i][jj], ps1[i][jj], ps2[i][jj], ps3[i][jj], &ok);
        if (!ok) { rc = 501; goto done2; }
      }
    }
  }
  pt_set(&tmp, &Rbar[0]);
  for (int i = 1; i < S; ++i) pt_add(&tmp, &tmp, &Rbar[i]);
  if (!pt_eq(&tmp, &G)) { rc = 502; goto done2; }                          /* phase5_check_R_dash_sum :768-776 */
  pt_t ysum; pt_init(&ysum);
  for (int i = 0; i < S; ++i) {
    pt_mul(&Spt[i], sigma_i[i], &R);                                       /* phase6_compute_S_i :784 */
    /* HomoELGamalProof (App. A.3) with G=R, H=base_point2, Y=generator, D=T_i, E=S_i, x=l_i, r=sigma_i */
    mpz_t s1, s2, z1, z2; mpz_inits(s1, s2, z1, z2, NULL);
    zin(s1, Z->heg_s1 + ((size_t)sess * S + i) * 8, 8); sc_mod(s1);
    zin(s2, Z->heg_s2 + ((size_t)sess * S + i) * 8, 8); sc_mod(s2);
    pt_t A1, A2, A3, TT; pt_init(&A1); pt_init(&A2); pt_init(&A3); pt_init(&TT);
    pt_mul(&A1, s1, &H2); pt_mul(&A2, s2, &G); pt_mul(&A3, s2, &R); pt_add(&TT, &A1, &A2);
    const pt_t* hp[7] = {&TT, &A3, &R, &H2, &G, &Tpt[i], &Spt[i]};
    hash_points_scalar(e, hp, 7);
    if (mpz_sgn(l[i]) != 0) { mpz_mul(z1, e, l[i]); mpz_add(z1, z1, s1); sc_mod(z1); } else mpz_set(z1, s1);
    mpz_mul(z2, e, sigma_i[i]); mpz_add(z2, z2, s2); sc_mod(z2);
    /* Round 6 verify (party_i.rs:801-833): z1 H + z2 Y == T + e D  and  z2 G == A3 + e E */
    pt_mul(&tmp, z1, &H2); pt_mul(&tmp2, z2, &G); pt_add(&tmp, &tmp, &tmp2);
    pt_mul(&tmp2, e, &Tpt[i]); pt_add(&tmp2, &TT, &tmp2);
    int okh = pt_eq(&tmp, &tmp2);
    pt_mul(&tmp, z2, &R); pt_mul(&tmp2, e, &Spt[i]); pt_add(&tmp2, &A3, &tmp2);
    okh = okh && pt_eq(&tmp, &tmp2);
    pt_clear(&A1); pt_clear(&A2); pt_clear(&A3); pt_clear(&TT); mpz_clears(s1, s2, z1, z2, NULL);
    if (!okh) { pt_clear(&ysum); rc = 601; goto done2; }
    pt_add(&ysum, &ysum, &Spt[i]);
  }
  pt_in(&tmp, K->y);
  if (!pt_eq(&ysum, &tmp)) { pt_clear(&ysum); rc = 602; goto done2; }      /* phase6_check_S_i_sum :835-848 */
  pt_clear(&ysum);

  /* ---------------- Round 7 (party_i.rs:850-936): local sigs, output_signature, verify ---------------- */
  {
    mpz_t m, r, s, half; mpz_inits(m, r, s, half, NULL);
    zin(m, Z->msg + (size_t)sess * 8, 8);
    mpz_mod(r, R.x, EC_Q);
    mpz_set_ui(s, 0);
    for (int i = 0; i < S; ++i) {
      mpz_mod(t, m, EC_Q); mpz_mul(t, t, k[i]);
      mpz_mul(t2, r, sigma_i[i]); mpz_add(t, t, t2); sc_mod(t);            /* s_i = m k_i + r sigma_i :864 */
      mpz_add(s, s, t); sc_mod(s);
    }
    int recid = mpz_tstbit(R.y, 0) ? 1 : 0;        /* ry = R.y mod q parity: y < p; the reference reduces mod q first (:890-895) */
    { mpz_t ry; mpz_init(ry); mpz_mod(ry, R.y, EC_Q); recid = mpz_tstbit(ry, 0) ? 1 : 0; mpz_clear(ry); }
    mpz_sub(half, EC_Q, s);
    if (mpz_cmp(s, half) > 0) { mpz_set(s, half); recid ^= 1; }             /* :896-900 */
    /* verify :913-936 */
    mpz_t b, u1, u2; mpz_inits(b, u1, u2, NULL);
    int okv = mpz_invert(b, s, EC_Q) != 0;
    if (okv) {
      mpz_mod(u1, m, EC_Q); mpz_mul(u1, u1, b); sc_mod(u1);
      mpz_mul(u2, r, b); sc_mod(u2);
      pt_in(&tmp3, K->y);
      pt_mul(&tmp, u1, &G); pt_mul(&tmp2, u2, &tmp3); pt_add(&tmp, &tmp, &tmp2);
      mpz_mod(t, tmp.x, EC_Q);
      okv = !tmp.inf && mpz_cmp(t, r) == 0;
    }
    mpz_clears(b, u1, u2, NULL);
    zout(r_out, 8, r); zout(s_out, 8, s); *recid_out = recid;
    if (R_out) pt_out(R_out, &R);
    mpz_clears(m, r, s, half, NULL);
    if (!okv) rc = 701;
  }
done2:
  mpz_clear(dinv);
done:
  for (int a = 0; a < n; ++a) mpz_clears(N[a], NN[a], pw[a], qw[a], NULL);
  for (int i = 0; i < S; ++i) {
    mpz_clears(k[i], gam[i], w[i], blind[i], com[i], ca[i], ra[i], delta_i[i], sigma_i[i], l[i], NULL);
    pt_clear(&g_gamma[i]); pt_clear(&g_w[i]); pt_clear(&Tpt[i]); pt_clear(&Rbar[i]); pt_clear(&Spt[i]);
  }
  for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) for (int v = 0; v < 2; ++v) {
    mpz_clears(beta[i][j][v], Bz[i][j][v], BTz[i][j][v], NULL);
    pt_clear(&Bpk[i][j][v]); pt_clear(&BR[i][j][v]); pt_clear(&BTpk[i][j][v]); pt_clear(&BTR[i][j][v]);
  }
  mpz_clears(lam, t, t2, e, NULL);
  pt_clear(&G); pt_clear(&H2); pt_clear(&R); pt_clear(&tmp); pt_clear(&tmp2); pt_clear(&tmp3);
  return rc;
}

/* batch driver: sessions [first, first+count) ; status[i] = 0 on success */
void orc_gg20_sign(const orc_gg20_keys* K, const orc_gg20_nonces* Z, int first, int count, uint32_t* r_out, uint32_t* s_out,
                   int32_t* recid_out, uint32_t* R_out, int32_t* status) {
  for (int sidx = first; sidx < first + count; ++sidx)
    status[sidx] = gg20_sign_one(K, Z, sidx, r_out + (size_t)sidx * 8, s_out + (size_t)sidx * 8, recid_out + sidx,
                                 R_out ? R_out + (size_t)sidx * 16 : NULL);
}
