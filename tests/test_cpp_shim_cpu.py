"""Without a GPU: the C++ host layer above the C-ABI (include/mpecdsa.hpp: the reference's names over mpecdsa_hip.h) and the
re-statement of the reference's unit tests over it (tests/cpp/test_shim.cpp) compile with -Wall -Werror and link against the
product library; run here, the program fails the way the layer promises — an mpecdsa::Error carrying mpe_last_error(), exit
status 1, no crash, no CPU fallback.  The -m gpu counterpart (tests/test_cpp_shim_gpu.py) runs the tests."""
import os
import subprocess

import cpp_shim


def test_cpp_host_layer_builds_and_fails_loudly_without_a_gpu(tmp_path, keys):
    exe = cpp_shim.build(str(tmp_path))
    fx = os.path.join(str(tmp_path), "fixture.bin")
    arrays = cpp_shim.write_fixture(fx, keys)
    assert arrays["al_alpha"].shape == (12, 24) and os.path.getsize(fx) > 10000
    import torch
    if torch.cuda.is_available():
        return                                        # on a GPU box the gpu-marked test does the run
    p = subprocess.run([exe, fx], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "EXCEPTION mpe_ctx_create" in p.stdout, (p.returncode, p.stdout, p.stderr)
