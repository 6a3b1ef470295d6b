"""Host-side conversion between Python integers and the C-ABI's little-endian u32 word arrays
(the analogue of curv `Converter::to_bytes/from_bytes`, big-endian there)."""
import numpy as np


def ints_to_words(vals, nwords):
    """list of non-negative ints -> np.uint32 array [len(vals), nwords] (little-endian words)."""
    buf = b"".join(int(v).to_bytes(nwords * 4, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u4").reshape(len(vals), nwords).copy()


def words_to_ints(arr):
    """np.uint32 array [n, nwords] -> list of ints."""
    a = np.ascontiguousarray(arr, dtype="<u4")
    nbytes = a.shape[1] * 4
    raw = a.tobytes()
    return [int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "little") for i in range(a.shape[0])]
