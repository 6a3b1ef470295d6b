"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/mpecdsa_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_if_missing():
    lib = os.path.join(ROOT, "multi_party_ecdsa_amd", "libmpecdsa_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as ge
        ge.build()
    return lib


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_build_if_missing())
    hdr = open(os.path.join(ROOT, "include", "mpecdsa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mpe_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 12
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mpecdsa_hip.h but not exported"
    from multi_party_ecdsa_amd import _native
    assert set(_native.EXPORTED) == declared


def test_version_and_argument_errors_without_gpu():
    from multi_party_ecdsa_amd import _native as N
    assert b"gfx950" in N.lib.mpe_version()
    # NULL / bad arguments are rejected before any HIP call
    assert N.lib.mpe_ctx_create(None, 0) == N.MPE_E_ARG
    assert N.lib.mpe_modexp(None, None, 1, None, None, None, 1, None, None) == N.MPE_E_ARG
    assert N.lib.mpe_modset_create(None, 2048, 1, None, None, None) == N.MPE_E_ARG


def test_product_does_not_link_the_oracle():
    """The shipped library must not depend on the oracle or on libgmp (no CPU fallback path)."""
    import subprocess
    out = subprocess.run(["ldd", _build_if_missing()], capture_output=True, text=True).stdout
    assert "mpe_oracle" not in out and "libgmp" not in out


def test_word_conversion_roundtrip():
    from multi_party_ecdsa_amd.words import ints_to_words, words_to_ints
    vals = [0, 1, (1 << 2048) - 1, 0x1234567890abcdef << 1000]
    assert words_to_ints(ints_to_words(vals, 64)) == vals
