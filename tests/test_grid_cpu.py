"""The persistent-grid rule of the ladder kernels (csrc/mpe_internal.h: persistent_grid — equal trips | full trips + a tail of lone waves |
hybrid) checked on the host: tests/cpp/test_grid.cpp is compiled with `hipcc --cuda-host-only` against the library's own header and run
here; no GPU involved.  (What the rule is worth is measured: profiles/r05/ab_grid_three_modes.jsonl.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_persistent_grid_rule_covers_every_group_in_the_minimum_number_of_passes(tmp_path):
    exe = str(tmp_path / "test_grid")
    subprocess.check_call(["hipcc", "--cuda-host-only", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_grid.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
