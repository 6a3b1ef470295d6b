"""MI355X-native batched crypto core for GG20 threshold signing (drop-in for the curv/paillier
call surface of ZenGo-X/multi-party-ecdsa's hot path).  See DESIGN.md / INTEGRATION.md."""
import torch  # noqa: F401  (first: one HIP runtime per process — the one torch ships)

from . import _native  # noqa: F401  (raises ImportError when the HIP library has not been built)
from .words import ints_to_words, words_to_ints  # noqa: F401
