"""ctypes binding of libmpecdsa_hip.so (the C-ABI declared in include/mpecdsa_hip.h).

The HIP library is the product; there is no CPU fallback.  Importing this module without the
built shared object raises ImportError with the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPE_LIB_PATH: another build of the SAME library (A/B measurements of kernel variants, tools/ab.sh); never a different backend
LIB_PATH = os.environ.get("MPE_LIB_PATH") or os.path.join(_HERE, "libmpecdsa_hip.so")

MPE_OK, MPE_E_ARG, MPE_E_HIP, MPE_E_NOMEM = 0, -1, -2, -3
GG20_STATUS_BAD_NONCE = 91                    # include/mpecdsa_hip.h MPE_GG20_STATUS_BAD_NONCE


def gg20_status_pass_failed(rc):              # MPE_GG20_STATUS_PASS_FAILED(rc)
    return 9000 - rc



class MpeError(RuntimeError):
    pass


class LaunchInfo(C.Structure):
    _fields_ = [("waves", C.c_int), ("ints_per_wave", C.c_int), ("limbs", C.c_int), ("limb_bits", C.c_int),
                ("lds_bytes_per_wave", C.c_int), ("table_scratch_bytes", C.c_size_t)]


class ProfRec(C.Structure):
    _fields_ = [("kind", C.c_int), ("bits", C.c_int), ("exp_words", C.c_int), ("batch", C.c_int), ("ms", C.c_float), ("exp2_words", C.c_int),
                ("sliding_frac", C.c_float)]


class Encoding(C.Structure):
    """`mpe_encoding` (include/mpecdsa_hip.h): the recalled byte-level conventions of curv / zk-paillier as a run-time profile"""
    _fields_ = [("chain_point", C.c_uint8), ("zero_bytes", C.c_uint8), ("ck_mask_order", C.c_uint8), ("reserved", C.c_uint8),
                ("ck_salt", C.c_uint32), ("ord_dlog", C.c_uint8 * 4), ("ord_pedersen", C.c_uint8 * 8), ("ord_heg", C.c_uint8 * 8),
                ("ord_ecddh", C.c_uint8 * 8), ("ord_cdlog", C.c_uint8 * 4)]
    SIZES = dict(ord_dlog=3, ord_pedersen=5, ord_heg=7, ord_ecddh=6, ord_cdlog=4)

    @classmethod
    def from_dict(cls, d):
        """d: {"chain_point": 0|1, "zero_bytes": 0|1, "ck_mask_order": 0|1, "ck_salt": int, "ord_*": [..]} (missing = default)"""
        e = cls()
        lib.mpe_encoding_default(C.byref(e))
        for k, v in d.items():
            if k.startswith("ord_"):
                arr = getattr(e, k)
                for i, x in enumerate(v):
                    arr[i] = x
            else:
                setattr(e, k, v)
        return e

    def as_dict(self):
        out = {k: getattr(self, k) for k in ("chain_point", "zero_bytes", "ck_mask_order", "ck_salt")}
        out.update({k: list(getattr(self, k))[:n] for k, n in self.SIZES.items()})
        return out


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, C.c_void_p) for f in fields]})


AliceProof = _ptr_struct("AliceProof", ["z", "e", "s", "s1", "s2"])
AliceNonces = _ptr_struct("AliceNonces", ["alpha", "beta", "gamma", "rho"])
PdlProof = _ptr_struct("PdlProof", ["z", "u1", "u2", "u3", "s1", "s2", "s3"])
PdlNonces = _ptr_struct("PdlNonces", ["alpha", "beta", "rho", "gamma"])
DlogProof = _ptr_struct("DlogProof", ["pk", "R", "z"])
BobProof = _ptr_struct("BobProof", ["t", "z", "e", "s", "s1", "s2", "t1", "t2"])
BobNonces = _ptr_struct("BobNonces", ["alpha", "beta", "gamma", "rho", "rho_prim", "sigma", "tau"])


GG20_NONCE_FIELDS = ["k", "gamma", "blind", "r_a", "al_alpha", "al_beta", "al_gamma", "al_rho", "mb_beta_tag", "mb_r",
                     "mb_nonce_b", "mb_nonce_bt", "l", "ped_s1", "ped_s2", "pdl_alpha", "pdl_beta", "pdl_rho", "pdl_gamma",
                     "heg_s1", "heg_s2", "msg"]
Gg20Nonces = _ptr_struct("Gg20Nonces", GG20_NONCE_FIELDS)
KEYGEN_ROUND1_FIELDS = ["y", "blind", "com", "N", "sigma", "Nt", "h1", "h2", "x_h1", "y_h1", "x_h2", "y_h2"]
KeygenRound1 = _ptr_struct("KeygenRound1", KEYGEN_ROUND1_FIELDS)
PedersenProof = _ptr_struct("PedersenProof", ["com", "e", "a1", "a2", "z1", "z2"])
HegStatement = _ptr_struct("HegStatement", ["G", "H", "Y", "D", "E"])
HegProof = _ptr_struct("HegProof", ["T", "A3", "z1", "z2"])
Blame5In = _ptr_struct("Blame5In", ["k", "k_rand", "gamma", "beta_tag", "beta_rand", "delta", "g_gamma", "c_a", "c_b"])
Blame6In = _ptr_struct("Blame6In", ["k", "k_rand", "miu", "miu_rand", "a1", "a2", "z", "S", "c_a", "c_b", "R"])
Blame7In = _ptr_struct("Blame7In", ["s", "r", "R_dash", "m", "R", "S"])
EcddhStatement = _ptr_struct("EcddhStatement", ["g1", "h1", "g2", "h2"])
EcddhProof = _ptr_struct("EcddhProof", ["a1", "a2", "z"])


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    vp, ip, u32p, i32p = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p
    sig = {
        "mpe_version": (C.c_char_p, []),
        "mpe_last_error": (C.c_char_p, []),
        "mpe_ctx_create": (ip, [C.POINTER(vp), ip]),
        "mpe_ctx_destroy": (ip, [vp]),
        "mpe_sync": (ip, [vp, vp]),
        "mpe_ctx_set_device_share": (ip, [vp, ip]),
        "mpe_ctx_set_option": (ip, [vp, C.c_char_p, C.c_char_p]),
        "mpe_ctx_get_option": (ip, [vp, C.c_char_p, C.POINTER(C.c_long)]),
        "mpe_ctx_option_count": (ip, []),
        "mpe_ctx_option_name": (C.c_char_p, [ip]),
        "mpe_comm_library": (ip, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mpe_gg20_pipeline_ticket_rc": (ip, [vp, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mpe_gg20_pipeline_inject_fault": (ip, [vp, ip, ip]),
        "mpe_gg20_pipeline_set_deadline_us": (ip, [vp, C.c_int64]),
        "mpe_gg20_pipeline_set_eager": (ip, [vp, ip]),
        "mpe_gg20_pipeline_poll": (ip, [vp, C.POINTER(C.c_int)]),
        "mpe_gg20_pipeline_counters": (ip, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "mpe_encoding_default": (None, [C.POINTER(Encoding)]),
        "mpe_ctx_set_encoding": (ip, [vp, C.POINTER(Encoding)]),
        "mpe_ctx_get_encoding": (ip, [vp, C.POINTER(Encoding)]),
        "mpe_modset_create": (ip, [vp, ip, ip, u32p, C.POINTER(vp), vp]),
        "mpe_modset_destroy": (ip, [vp]),
        "mpe_modset_count": (ip, [vp]),
        "mpe_modset_bits": (ip, [vp]),
        "mpe_modexp": (ip, [vp, vp, ip, i32p, u32p, u32p, ip, u32p, vp]),
        "mpe_modexp2": (ip, [vp, vp, ip, i32p, u32p, u32p, ip, u32p, u32p, ip, u32p, vp]),
        "mpe_modmul": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, vp]),
        "mpe_lindell_partial_sig": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p, vp]),
        "mpe_lindell_sign": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, u32p, u32p, i32p, vp]),
        "mpe_correct_key_prove": (ip, [vp, vp, u32p, vp]),
        "mpe_composite_dlog_prove": (ip, [vp, ip, u32p, u32p, u32p, u32p, u32p, u32p, u32p, vp]),
        "mpe_paillier_open": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, vp]),
        "mpe_lindell_pdl_proof": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, u32p, u32p, vp, u32p, vp, vp]),
        "mpe_lindell_pdl_verify": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p, vp, vp, vp]),
        "mpe_last_launch_info": (ip, [vp, C.POINTER(LaunchInfo)]),
        "mpe_modinv": (ip, [vp, vp, ip, i32p, u32p, u32p, vp, vp]),
        "mpe_ec_mul_base": (ip, [vp, ip, u32p, ip, u32p, vp]),
        "mpe_ec_mul": (ip, [vp, ip, u32p, ip, u32p, u32p, vp]),
        "mpe_ec_add": (ip, [vp, ip, u32p, u32p, u32p, vp]),
        "mpe_dlog_prove": (ip, [vp, ip, u32p, u32p, u32p, u32p, u32p, vp]),
        "mpe_dlog_verify": (ip, [vp, ip, u32p, u32p, u32p, vp, vp]),
        "mpe_statements_create": (ip, [vp, ip, u32p, u32p, u32p, C.POINTER(vp), vp]),
        "mpe_statements_destroy": (ip, [vp]),
        "mpe_alice_generate": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, u32p, u32p, C.POINTER(AliceNonces),
                                    C.POINTER(AliceProof), vp]),
        "mpe_alice_verify": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, C.POINTER(AliceProof), vp, vp]),
        "mpe_pdl_prove": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, u32p, u32p, u32p, u32p, C.POINTER(PdlNonces),
                               C.POINTER(PdlProof), vp]),
        "mpe_pdl_verify": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, u32p, u32p, C.POINTER(PdlProof), vp, vp]),
        "mpe_mta_message_a": (ip, [vp, vp, vp, ip, i32p, u32p, u32p, C.POINTER(AliceNonces), u32p, C.POINTER(AliceProof), vp]),
        "mpe_mta_message_b": (ip, [vp, vp, vp, ip, i32p, u32p, u32p, C.POINTER(AliceProof), u32p, u32p, u32p, u32p, u32p, u32p,
                                   C.POINTER(DlogProof), C.POINTER(DlogProof), vp, vp]),
        "mpe_mta_verify_get_alpha": (ip, [vp, vp, ip, i32p, u32p, C.POINTER(DlogProof), C.POINTER(DlogProof), u32p, u32p, u32p,
                                          vp, vp]),
        "mpe_bob_generate": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, u32p, u32p, u32p, u32p, C.POINTER(BobNonces), ip,
                                  C.POINTER(BobProof), u32p, vp]),
        "mpe_bob_verify": (ip, [vp, vp, vp, ip, i32p, i32p, u32p, u32p, C.POINTER(BobProof), u32p, u32p, vp, vp]),
        "mpe_gg20_keys_create": (ip, [vp, ip, ip, ip, C.POINTER(C.c_int32), ip, ip, C.POINTER(C.c_int32)] + [u32p] * 9 + [C.POINTER(vp), vp]),
        "mpe_gg20_keys_destroy": (ip, [vp]),
        "mpe_gg20_keys_fb_window_bits": (ip, [vp]),
        "mpe_gg20_msg_words": (ip, [ip, ip, ip]),
        "mpe_gg20_session_create": (ip, [vp, vp, ip, ip, C.POINTER(C.c_int32), i32p, C.POINTER(Gg20Nonces), ip, C.POINTER(vp), vp]),
        "mpe_gg20_session_destroy": (ip, [vp, vp]),
        "mpe_gg20_round0": (ip, [vp, u32p, vp]),
        "mpe_gg20_round1": (ip, [vp, u32p, C.POINTER(C.c_int64), u32p, vp]),
        "mpe_gg20_round2": (ip, [vp, u32p, C.POINTER(C.c_int64), u32p, vp]),
        "mpe_gg20_round3": (ip, [vp, u32p, C.POINTER(C.c_int64), u32p, vp]),
        "mpe_gg20_round4": (ip, [vp, u32p, C.POINTER(C.c_int64), u32p, vp]),
        "mpe_gg20_round5": (ip, [vp, u32p, C.POINTER(C.c_int64), u32p, vp]),
        "mpe_gg20_round6": (ip, [vp, u32p, C.POINTER(C.c_int64), vp]),
        "mpe_gg20_round7": (ip, [vp, u32p, u32p, vp]),
        "mpe_gg20_complete": (ip, [vp, u32p, C.POINTER(C.c_int64), vp]),
        "mpe_gg20_session_result": (ip, [vp, vp, vp, u32p, u32p, vp, u32p, vp]),
        "mpe_gg20_sign": (ip, [vp, vp, ip, i32p, C.POINTER(Gg20Nonces), u32p, u32p, vp, u32p, vp, ip, ip, vp]),
        "mpe_pedersen_prove": (ip, [vp, ip, u32p, u32p, u32p, u32p, C.POINTER(PedersenProof), vp]),
        "mpe_pedersen_verify": (ip, [vp, ip, C.POINTER(PedersenProof), vp, vp]),
        "mpe_heg_prove": (ip, [vp, ip, u32p, u32p, u32p, u32p, C.POINTER(HegStatement), C.POINTER(HegProof), vp]),
        "mpe_heg_verify": (ip, [vp, ip, C.POINTER(HegStatement), C.POINTER(HegProof), vp, vp]),
        "mpe_hash_commit_point": (ip, [vp, ip, u32p, u32p, u32p, vp]),
        "mpe_gg20_session_fault_inject": (ip, [vp, ip, C.c_uint32]),
        "mpe_gg20_session_rearm": (ip, [vp, i32p, C.POINTER(Gg20Nonces), vp]),
        "mpe_gg20_session_abort": (ip, [vp, vp]),
        "mpe_gg20_blame5": (ip, [vp, vp, ip, i32p, C.POINTER(Blame5In), u32p, vp]),
        "mpe_gg20_blame6": (ip, [vp, vp, ip, i32p, C.POINTER(Blame6In), u32p, vp]),
        "mpe_gg20_blame7": (ip, [vp, ip, ip, C.POINTER(Blame7In), u32p, vp]),
        "mpe_gg20_session_blame6_state": (ip, [vp, u32p, u32p, u32p, u32p, u32p, vp]),
        "mpe_ecddh_prove": (ip, [vp, ip, u32p, u32p, C.POINTER(EcddhStatement), C.POINTER(EcddhProof), vp]),
        "mpe_ecddh_verify": (ip, [vp, ip, C.POINTER(EcddhStatement), C.POINTER(EcddhProof), vp, vp]),
        "mpe_correct_key_verify": (ip, [vp, ip, u32p, u32p, vp, vp]),
        "mpe_composite_dlog_verify": (ip, [vp, ip, u32p, u32p, u32p, u32p, u32p, vp, vp]),
        "mpe_vss_validate_share": (ip, [vp, ip, ip, u32p, u32p, i32p, vp, vp]),
        "mpe_vss_point_commitment": (ip, [vp, ip, ip, u32p, i32p, u32p, vp]),
        "mpe_ctx_wipe": (ip, [vp, vp]),
        "mpe_ctx_scratch_audit": (ip, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]),
        "mpe_statements_create_wb": (ip, [vp, ip, u32p, u32p, u32p, ip, C.POINTER(vp), vp]),
        "mpe_prof_enable": (ip, [vp, ip]),
        "mpe_prof_collect": (ip, [vp, C.POINTER(ProfRec), ip, C.POINTER(C.c_int)]),
        "mpe_paillier_create_public": (ip, [vp, ip, u32p, C.POINTER(vp), vp]),
        "mpe_paillier_create_private": (ip, [vp, ip, u32p, u32p, C.POINTER(vp), vp]),
        "mpe_paillier_destroy": (ip, [vp]),
        "mpe_paillier_nkeys": (ip, [vp]),
        "mpe_paillier_n": (vp, [vp]),
        "mpe_paillier_encrypt": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, vp]),
        "mpe_paillier_decrypt": (ip, [vp, vp, ip, i32p, u32p, u32p, vp]),
        "mpe_paillier_add": (ip, [vp, vp, ip, i32p, u32p, u32p, u32p, vp]),
        "mpe_paillier_mul": (ip, [vp, vp, ip, i32p, u32p, u32p, ip, u32p, vp]),
        "mpe_keygen_verify_round1": (ip, [vp, ip, ip, C.POINTER(KeygenRound1), vp, u32p, vp]),
        "mpe_keygen_verify_round2": (ip, [vp, ip, ip, ip, u32p, u32p, i32p, u32p, vp, u32p, vp]),
        "mpe_comm_unique_id": (ip, [C.c_char_p]),
        "mpe_comm_create": (ip, [vp, C.c_char_p, ip, ip, C.POINTER(vp)]),
        "mpe_comm_destroy": (ip, [vp]),
        "mpe_comm_rank": (ip, [vp]),
        "mpe_comm_world": (ip, [vp]),
        "mpe_comm_gather_mode": (ip, [vp]),
        "mpe_comm_all_gather": (ip, [vp, vp, C.c_size_t, vp]),
        "mpe_comm_layout_self_test": (ip, [vp, ip, C.POINTER(C.c_int), C.POINTER(C.c_int), vp]),
        "mpe_gg20_shard_where": (ip, [ip, ip, ip, ip, ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mpe_gg20_shard_blocks": (ip, [ip, ip, ip]),
        "mpe_gg20_shard_per_rank": (ip, [ip, ip, ip]),
        "mpe_gg20_shard_in_off": (ip, [ip, ip, ip, ip, ip, C.POINTER(C.c_int64)]),
        "mpe_gg20_round_exchange": (ip, [vp, ip, ip, ip, ip, ip, u32p, vp]),
        "mpe_sample_bits": (ip, [vp, ip, C.c_char_p, C.c_uint64, ip, ip, u32p, vp]),
        "mpe_sample_below": (ip, [vp, ip, C.c_char_p, C.c_uint64, u32p, ip, ip, i32p, ip, ip, u32p, i32p, vp]),
        "mpe_sample_scalar": (ip, [vp, ip, C.c_char_p, C.c_uint64, u32p, i32p, vp]),
        "mpe_gg20_nonces_alloc": (ip, [vp, vp, ip, ip, C.POINTER(vp)]),
        "mpe_gg20_nonces_view": (ip, [vp, C.POINTER(Gg20Nonces)]),
        "mpe_gg20_nonces_free": (ip, [vp]),
        "mpe_gg20_pipeline_create": (ip, [vp, vp, ip, ip, ip, ip, C.POINTER(vp)]),
        "mpe_gg20_pipeline_destroy": (ip, [vp]),
        "mpe_gg20_pipeline_submit": (ip, [vp, i32p, C.POINTER(Gg20Nonces), u32p, u32p, vp, u32p, vp, vp, C.POINTER(C.c_uint64)]),
        "mpe_gg20_pipeline_submit_seeded": (ip, [vp, i32p, C.c_char_p, C.c_uint64, u32p, u32p, u32p, vp, u32p, vp, vp, C.POINTER(C.c_uint64)]),
        "mpe_gg20_pipeline_flush": (ip, [vp]),
        "mpe_gg20_pipeline_query": (ip, [vp, C.c_uint64, C.POINTER(C.c_int)]),
        "mpe_gg20_pipeline_wait": (ip, [vp, C.c_uint64]),
        "mpe_gg20_pipeline_stream_wait": (ip, [vp, C.c_uint64, vp]),
        "mpe_gg20_pipeline_latency_ms": (ip, [vp, C.c_uint64, C.POINTER(C.c_float)]),
        "mpe_gg20_pipeline_pass_ms": (ip, [vp, C.c_uint64, C.POINTER(C.c_float)]),
        "mpe_gg20_pipeline_sampler_failures": (ip, [vp, C.POINTER(C.c_int32)]),
        "mpe_gg20_sample_nonces": (ip, [vp, vp, ip, ip, C.POINTER(C.c_int32), i32p, C.c_char_p, C.c_uint64, C.POINTER(Gg20Nonces), i32p, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

# every symbol include/mpecdsa_hip.h declares; tests check the library exports all of them
EXPORTED = ["mpe_version", "mpe_last_error", "mpe_ctx_create", "mpe_ctx_destroy", "mpe_sync",
            "mpe_encoding_default", "mpe_ctx_set_encoding", "mpe_ctx_get_encoding", "mpe_ctx_set_device_share",
            "mpe_modset_create", "mpe_modset_destroy", "mpe_modset_count", "mpe_modset_bits",
            "mpe_modexp", "mpe_modexp2", "mpe_modmul", "mpe_last_launch_info", "mpe_paillier_create_public",
            "mpe_paillier_create_private", "mpe_paillier_destroy", "mpe_paillier_nkeys", "mpe_paillier_n",
            "mpe_paillier_encrypt", "mpe_paillier_decrypt", "mpe_paillier_add", "mpe_paillier_mul",
            "mpe_prof_enable", "mpe_prof_collect", "mpe_modinv", "mpe_ec_mul_base", "mpe_ec_mul", "mpe_ec_add",
            "mpe_dlog_prove", "mpe_dlog_verify", "mpe_statements_create", "mpe_statements_destroy",
            "mpe_alice_generate", "mpe_alice_verify", "mpe_pdl_prove", "mpe_pdl_verify", "mpe_gg20_keys_create",
            "mpe_gg20_keys_destroy", "mpe_gg20_sign", "mpe_bob_generate", "mpe_bob_verify", "mpe_mta_message_a",
            "mpe_mta_message_b", "mpe_mta_verify_get_alpha", "mpe_lindell_partial_sig", "mpe_lindell_sign", "mpe_lindell_pdl_proof", "mpe_lindell_pdl_verify", "mpe_paillier_open", "mpe_correct_key_prove", "mpe_composite_dlog_prove",
            "mpe_gg20_keys_fb_window_bits", "mpe_gg20_msg_words", "mpe_gg20_session_create", "mpe_gg20_session_destroy",
            "mpe_gg20_round0", "mpe_gg20_round1", "mpe_gg20_round2", "mpe_gg20_round3", "mpe_gg20_round4", "mpe_gg20_round5",
            "mpe_gg20_round6", "mpe_gg20_round7", "mpe_gg20_complete", "mpe_gg20_session_result", "mpe_pedersen_prove",
            "mpe_pedersen_verify", "mpe_heg_prove", "mpe_heg_verify", "mpe_hash_commit_point", "mpe_ctx_wipe", "mpe_ctx_scratch_audit", "mpe_gg20_session_fault_inject", "mpe_gg20_blame5",
            "mpe_gg20_blame6", "mpe_gg20_blame7", "mpe_correct_key_verify", "mpe_composite_dlog_verify", "mpe_vss_validate_share",
            "mpe_vss_point_commitment", "mpe_gg20_session_blame6_state", "mpe_ecddh_prove", "mpe_ecddh_verify",
            "mpe_statements_create_wb", "mpe_gg20_session_rearm", "mpe_sample_bits", "mpe_sample_below", "mpe_sample_scalar",
            "mpe_gg20_nonces_alloc", "mpe_gg20_nonces_view", "mpe_gg20_nonces_free", "mpe_gg20_sample_nonces",
            "mpe_gg20_pipeline_create", "mpe_gg20_pipeline_destroy", "mpe_gg20_pipeline_submit", "mpe_gg20_pipeline_submit_seeded",
            "mpe_gg20_pipeline_flush", "mpe_gg20_pipeline_query", "mpe_gg20_pipeline_wait", "mpe_gg20_pipeline_stream_wait",
            "mpe_gg20_pipeline_latency_ms", "mpe_gg20_pipeline_pass_ms", "mpe_gg20_pipeline_sampler_failures", "mpe_keygen_verify_round1", "mpe_keygen_verify_round2",
            "mpe_comm_unique_id", "mpe_comm_create", "mpe_comm_destroy", "mpe_comm_rank", "mpe_comm_world", "mpe_comm_gather_mode", "mpe_comm_all_gather",
            "mpe_comm_layout_self_test", "mpe_gg20_shard_where", "mpe_gg20_shard_blocks", "mpe_gg20_shard_per_rank", "mpe_gg20_shard_in_off",
            "mpe_gg20_round_exchange", "mpe_gg20_session_abort", "mpe_ctx_set_option", "mpe_ctx_get_option", "mpe_ctx_option_count",
            "mpe_ctx_option_name", "mpe_comm_library", "mpe_gg20_pipeline_ticket_rc", "mpe_gg20_pipeline_inject_fault",
            "mpe_gg20_pipeline_set_deadline_us", "mpe_gg20_pipeline_set_eager", "mpe_gg20_pipeline_poll", "mpe_gg20_pipeline_counters"]


def check(rc, what):
    if rc != MPE_OK:
        raise MpeError(f"{what} failed: rc={rc} ({lib.mpe_last_error().decode()})")
