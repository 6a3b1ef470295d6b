"""The bit-level Python models of the lane algorithms (tools/model/): column accumulators must stay below 2^64 and the
results must be the right residues, for random operands and with every limb at its lazy maximum."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "model", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pair_multiplication_model_bounds_and_residues():
    pm = _load("pair_model")
    for bits, tpi in ((2048, 4), (1024, 2)):
        for stress in (False, True):
            st = pm.run(bits, tpi, 2, 11, stress)
            assert st["maxcol"] < 1 << 64
            assert st["maxlimb"] < (1 << 29) + (1 << 12)


def test_montmul_model_bounds():
    mm = _load("montmul_model")
    for bits, w, l, tpi in ((4096, 29, 18, 8), (2048, 29, 18, 4)):
        st = mm.run(bits, w, l, tpi, iters=2)
        assert st["maxcol"] < 1 << 64
