/* lindell_oracle.c — CPU ORACLE (test infrastructure, NOT product code; included by mpe_oracle.c).
 * The signing half of Lindell'17 two-party ECDSA, restated over libgmp with every sampled value as an input:
 *   party two  PartialSig::compute            src/protocols/two_party_ecdsa/lindell_2017/party_two.rs:390-423
 *   party one  Signature::compute_with_recid  src/protocols/two_party_ecdsa/lindell_2017/party_one.rs:519-565
 * PARITY UNPINNED like the rest of the oracle (the reference has no vectors for this path); pinned by an independent
 * pure-Python restatement (tests/pyref.py: other big-integer engine, textbook decryption) and by the independent
 * ECDSA verification of the signatures it produces (tests/test_lindell_cpu.py; the reference's own test asserts
 * party_one::verify on the result, lindell_2017/test.rs). */

/* c3 = Enc(rho q + k2^-1 m; r) * c_key^(k2^-1 rx x2)  mod N^2 */
void orc_lindell_partial_sig(int batch, int nkeys, const uint32_t* N, const int32_t* key_idx, const uint32_t* c_key,
                             const uint32_t* x2, const uint32_t* k2, const uint32_t* R1, const uint32_t* msg,
                             const uint32_t* rho, const uint32_t* r, uint32_t* c3) {
  ec_setup();
  mpz_t n, nn, k, kinv, rx, m, ps, rr, c1, c2, v, t, ck;
  mpz_inits(n, nn, k, kinv, rx, m, ps, rr, c1, c2, v, t, ck, NULL);
  pt_t P, Rp; pt_init(&P); pt_init(&Rp);
  for (int i = 0; i < batch; ++i) {
    zin(n, N + (size_t)pick(key_idx, nkeys, i) * ORC_W2048, ORC_W2048);
    mpz_mul(nn, n, n);
    zin(k, k2 + (size_t)i * 8, 8); mpz_mod(k, k, EC_Q);
    pt_in(&P, R1 + (size_t)i * 16);
    pt_mul(&Rp, k, &P);                                   /* r = R1 * k2                       :401 */
    mpz_mod(rx, Rp.x, EC_Q);                              /* rx = r.x mod q                    :403 */
    mpz_invert(kinv, k, EC_Q);                            /* k2_inv                            :405 */
    zin(m, msg + (size_t)i * 8, 8);
    mpz_mul(t, kinv, m); mpz_mod(t, t, EC_Q);
    zin(ps, rho + (size_t)i * 16, 16);
    mpz_mul(ps, ps, EC_Q); mpz_add(ps, ps, t);            /* partial_sig = rho q + k2_inv m    :406 */
    zin(rr, r + (size_t)i * ORC_W2048, ORC_W2048);
    paillier_enc(c1, n, nn, ps, rr);                      /* c1 = Paillier::encrypt            :408 */
    zin(t, x2 + (size_t)i * 8, 8); mpz_mod(t, t, EC_Q);
    mpz_mul(v, rx, t); mpz_mod(v, v, EC_Q);
    mpz_mul(v, v, kinv); mpz_mod(v, v, EC_Q);             /* v = k2_inv (rx x2)                :409-413 */
    zin(ck, c_key + (size_t)i * ORC_W4096, ORC_W4096);
    mpz_powm(c2, ck, v, nn);                              /* c2 = Paillier::mul(c_key, v)      :414-418 */
    mpz_mul(c1, c1, c2); mpz_mod(c1, c1, nn);             /* c3 = Paillier::add(c2, c1)        :421 */
    zout(c3 + (size_t)i * ORC_W4096, ORC_W4096, c1);
  }
  pt_clear(&P); pt_clear(&Rp);
  mpz_clears(n, nn, k, kinv, rx, m, ps, rr, c1, c2, v, t, ck, NULL);
}

/* (r, s, recid) from the partial signature: s = min(s'', q - s''), s'' = Dec(c3) k1^-1 mod q */
void orc_lindell_sign(int batch, int nkeys, const uint32_t* p, const uint32_t* q, const int32_t* key_idx, const uint32_t* c3,
                      const uint32_t* k1, const uint32_t* R2, uint32_t* r_out, uint32_t* s_out, int32_t* recid) {
  ec_setup();
  mpz_t pp, qq, k, kinv, rx, ry, st, s2, neg, cc;
  mpz_inits(pp, qq, k, kinv, rx, ry, st, s2, neg, cc, NULL);
  pt_t P, Rp; pt_init(&P); pt_init(&Rp);
  for (int i = 0; i < batch; ++i) {
    const int kk = pick(key_idx, nkeys, i);
    zin(pp, p + (size_t)kk * ORC_W1024, ORC_W1024);
    zin(qq, q + (size_t)kk * ORC_W1024, ORC_W1024);
    zin(k, k1 + (size_t)i * 8, 8); mpz_mod(k, k, EC_Q);
    pt_in(&P, R2 + (size_t)i * 16);
    pt_mul(&Rp, k, &P);                                   /* r = R2 * k1                       :526 */
    mpz_mod(rx, Rp.x, EC_Q); mpz_mod(ry, Rp.y, EC_Q);     /*                                   :528-535 */
    mpz_invert(kinv, k, EC_Q);                            /* k1_inv                            :536 */
    zin(cc, c3 + (size_t)i * ORC_W4096, ORC_W4096);
    paillier_dec(st, pp, qq, cc);                         /* s_tag = Paillier::decrypt         :538-542 */
    mpz_mod(st, st, EC_Q);                                /* Scalar::from(s_tag)               :543 */
    mpz_mul(s2, st, kinv); mpz_mod(s2, s2, EC_Q);         /* s_tag_tag                         :544 */
    mpz_sub(neg, EC_Q, s2);
    int rec = mpz_tstbit(ry, 0) ? 1 : 0;                  /*                                   :557-558 */
    if (mpz_cmp(s2, neg) > 0) { rec ^= 1; mpz_set(s2, neg); }      /* s = min(..); recid ^= 1  :546-561 */
    zout(r_out + (size_t)i * 8, 8, rx);
    zout(s_out + (size_t)i * 8, 8, s2);
    recid[i] = rec;
  }
  pt_clear(&P); pt_clear(&Rp);
  mpz_clears(pp, qq, k, kinv, rx, ry, st, s2, neg, cc, NULL);
}
