"""Counter-based PRNG shared (bit-exactly) by Python and C.

There are no real checkpoints in the build container or on the GPU box (no network), so
full-size models are *generated*: element ``i`` of tensor ``name`` under ``seed`` is a pure
function of ``(seed, fnv1a64(name), i)``.  The C twin is ``glc_prng_fill`` in
``gliclass/c_amd/host/glc_weights.c``; ``tests/test_weights.py`` asserts both produce the same
bits.  Only integer ops and one double multiply-add are used so there is no libm dependence.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MASK = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def hash_u64(seed: int, tid: int, n: int, start: int = 0) -> np.ndarray:
    """splitmix64 finaliser over the counter stream ``start .. start+n``."""
    base = (tid ^ ((seed * 0x9E3779B97F4A7C15) & _MASK)) & _MASK
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(base) + i * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_f32(seed: int, name: str, n: int, amp: float, mean: float = 0.0) -> np.ndarray:
    """n floats uniform in [mean-amp, mean+amp) (double arithmetic, one cast to f32)."""
    z = hash_u64(seed, fnv1a64(name), n)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return ((2.0 * u - 1.0) * float(amp) + float(mean)).astype(np.float32)


def randint(seed: int, name: str, n: int, lo: int, hi: int) -> np.ndarray:
    """n ints uniform in [lo, hi)."""
    z = hash_u64(seed, fnv1a64(name), n)
    return (lo + (z >> np.uint64(11)) % np.uint64(hi - lo)).astype(np.int64)
