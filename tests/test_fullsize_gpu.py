"""BASELINE's configurations at the sizes SURVEY.md 8d names, compared with the checkers in full (not a prefix):
 * config 2: ALL 65 536 Paillier ciphertexts and decrypted plaintexts bit-exact against the GMP oracle;
 * config 3: the 4 096-instance prefix of both scalar multiplications against OpenSSL's EC_POINT_mul;
 * config 4: ALL 1 024 signatures (r, s, recid, R) byte-identical to the oracle, and every one verifies under the wallet's
   public key with OpenSSL's ECDSA_do_verify (the reference's independent check: gg_2020/test.rs:711-748);
 * config 5: one GPU's share (8 192 sessions, t=2, n=5): a 128-session sample byte-identical to the oracle, all 8 192
   signatures verify under OpenSSL.
The oracle runs on the host threads the cgroup grants (tens of seconds each)."""
import numpy as np
import pytest
import torch

import fixtures as F
import gg20_fixture as G
import orc
import ossl

pytestmark = pytest.mark.gpu


def _threads():
    import bench
    return min(bench.host_cores()[0], 64)


def test_config2_all_65536_ciphertexts_and_plaintexts_equal_the_oracle(gpu_ctx, keys):
    import bench
    from multi_party_ecdsa_amd import engine as E
    res = bench.paillier_config2(gpu_ctx, E, keys, F, oracle_threads=_threads(), oracle_items=65536)
    assert res["parity_prefix"] == 65536 and res["parity_vs_oracle_on_prefix"]
    assert res["roundtrip_ok"] and res["holder_equals_public_ciphertext"]


def test_config3_prefix_scalar_multiplications_equal_openssl(gpu_ctx):
    from multi_party_ecdsa_amd import engine as E
    import bench
    n = 4096
    g = torch.Generator(device=gpu_ctx.device)
    g.manual_seed(3)
    kb = bench.rand_words(g, gpu_ctx.device, n, 8, 7)
    x = bench.rand_words(g, gpu_ctx.device, n, 8, 8)
    x[:, 7] &= 0x7FFFFFFF
    Gp = E.ec_mul_base(gpu_ctx, kb)
    Qp = E.ec_mul(gpu_ctx, x, Gp)
    gpu_ctx.sync()
    h = lambda t: np.ascontiguousarray(t.cpu().numpy().view(np.uint32))
    th = _threads()
    assert np.array_equal(h(Gp), ossl.ec_mul(h(kb), threads=th))
    assert np.array_equal(h(Qp), ossl.ec_mul(h(x), h(Gp), threads=th))


def _sign_and_check(gpu_ctx, keys, t, n, signers, B, sample, seed):
    import bench
    from multi_party_ecdsa_amd import engine as E
    lk = G.make_local_keys(keys, t, n, signers)
    gk = E.Gg20Keys(gpu_ctx, t, n, signers, lk["arrays"])
    S = len(signers)
    gen = torch.Generator(device=gpu_ctx.device)
    gen.manual_seed(seed)
    nonces = bench.make_device_nonces(gen, gpu_ctx.device, B, S, S, n)
    r, s, recid, status, R = E.gg20_sign(gpu_ctx, gk, nonces, B, want_R=True)
    gpu_ctx.sync()
    r, s, recid, status, R = [o.cpu().numpy() for o in (r, s, recid, status, R)]
    assert not status.any()
    th = _threads()
    hn = bench._host({f: v[: sample * (v.shape[0] // B)] for f, v in nonces.items()})
    _, wr, ws, wrecid, wstatus = bench.cpu_baseline_gg20(lk, hn, sample, min(th, sample))
    assert not wstatus.any()
    assert np.array_equal(r[:sample].view(np.uint32), wr) and np.array_equal(s[:sample].view(np.uint32), ws)
    assert np.array_equal(recid[:sample], wrecid)
    ok = ossl.ecdsa_verify(lk["arrays"]["y"][0], nonces["msg"].cpu().numpy().view(np.uint32), r.view(np.uint32), s.view(np.uint32), threads=th)
    assert ok.all(), f"{int((~ok).sum())} signatures rejected by OpenSSL"
    # r = R.x mod q for the R the parties agreed on
    Rx = F.ints(R.view(np.uint32)[:64, :8])
    assert [v % 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141 for v in Rx] == F.ints(r.view(np.uint32)[:64])
    gk.close()


def test_config4_all_1024_signatures_equal_the_oracle_and_verify_under_openssl(gpu_ctx, keys):
    _sign_and_check(gpu_ctx, keys, 1, 3, [0, 1], 1024, 1024, 44)


def test_config5_share_128_sample_equals_the_oracle_all_8192_verify_under_openssl(gpu_ctx, keys):
    # (round 6: the oracle sample was 512 sessions = 34 s of host time; every one of the 8 192 signatures still goes through OpenSSL)
    _sign_and_check(gpu_ctx, keys, 2, 5, [0, 1, 2], 8192, 128, 55)
