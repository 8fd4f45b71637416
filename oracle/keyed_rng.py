"""
TEST INFRASTRUCTURE - keyed (counter-based) random streams for the oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The shipped renderer has its own device-side
implementation of the same generator.

The reference renderer draws from three unseeded, module-level
numpy.random.default_rng() objects (camera.py:5, materials.py:13,
main.py:23) whose consumption order depends on batch composition.  To make
"identical injected random streams" well defined, every draw is re-keyed by
what it is used for:

    Philox-4x32-10( counter = (x, y, sample, slot), key = (seed_lo, seed_hi) )
    slot = purpose | bounce << 4 | attempt << 12

where (x, y) is the pixel, sample the global sample index of the path, bounce
the bounce number of the scatter event, purpose one of the constants below
and attempt the index of the rejection-sampling retry.  One Philox block
yields four 32-bit words r0..r3; a uniform double in [0, 1) is r * 2**-32 and
uniform(low, high) = low + (high - low) * u evaluated in float64, which is
what numpy's Generator.uniform computes from its own u.

Purposes (words used):
    JITTER  (r0, r1)      sub-pixel offset in [0,1)^2       camera.py:130
    LENS    (r0, r1)      lens-disk candidate in [-1,1)^2   camera.py:138-148
    UNITVEC (r0, r1, r2)  unit-sphere candidate in [-1,1)^3 materials.py:407-426
    SCHLICK (r0)          reflect/refract coin in [0,1)     materials.py:331
"""
import numpy as np

JITTER = 0
LENS = 1
UNITVEC = 2
SCHLICK = 3

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_SH = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  Counter words are array-likes of equal
    (broadcastable) shape holding values < 2**32; key words are Python ints.
    Returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *(np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    )
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _SH, p0 & _MASK
        hi1, lo1 = p1 >> _SH, p1 & _MASK
        c0, c1, c2, c3 = (
            hi1 ^ c1 ^ np.uint64(k0),
            lo1,
            hi0 ^ c3 ^ np.uint64(k1),
            lo0,
        )
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def slot(purpose, bounce, attempt):
    return (
        np.asarray(purpose, dtype=np.uint64)
        | (np.asarray(bounce, dtype=np.uint64) << np.uint64(4))
        | (np.asarray(attempt, dtype=np.uint64) << np.uint64(12))
    )


def keyed_words(seed, x, y, sample, purpose, bounce, attempt):
    """The four 32-bit words of the block keyed by the arguments."""
    seed = int(seed)
    return philox4x32_10(
        x, y, sample, slot(purpose, bounce, attempt),
        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
    )


def to_unit(words):
    """uint32 -> float64 uniform in [0, 1)."""
    return words.astype(np.float64) * (1.0 / 4294967296.0)


def keyed_uniform3(seed, x, y, sample, purpose, bounce, attempt, low, high):
    """(n, 3) float64 uniforms in [low, high) from words r0, r1, r2."""
    r0, r1, r2, _ = keyed_words(seed, x, y, sample, purpose, bounce, attempt)
    u = np.stack([to_unit(r0), to_unit(r1), to_unit(r2)], axis=-1)
    return low + (high - low) * u


def keyed_unit_vectors(seed, x, y, sample, bounce):
    """Keyed version of the reference's numpy_random_unit_vecs
    (materials.py:401-431): draw in the cube [-1,1)^3 as float64, cast to
    float32, accept iff 1e-5 <= |v|^2 <= 1 (float32 arithmetic), redraw only
    the rejected entries with the next attempt number, normalise in float32.
    x, y, sample are 1-D integer arrays naming the path of every entry."""
    x = np.asarray(x)
    y = np.asarray(y)
    sample = np.asarray(sample)
    n = x.shape[0]
    vecs = np.zeros((n, 3), dtype=np.single)
    pending = np.arange(n)
    attempt = 0
    while pending.size:
        cand = keyed_uniform3(
            seed, x[pending], y[pending], sample[pending],
            UNITVEC, bounce, attempt, -1.0, 1.0,
        ).astype(np.single)
        vecs[pending] = cand
        len2 = np.einsum("ij,ij->i", cand, cand)
        bad = np.logical_or(len2 > 1.0, len2 < 0.00001)
        pending = pending[bad]
        attempt += 1
    vecs /= np.sqrt(np.einsum("ij,ij->i", vecs, vecs))[..., np.newaxis]
    return vecs


def keyed_schlick(seed, x, y, sample, bounce):
    r0, _, _, _ = keyed_words(seed, x, y, sample, SCHLICK, bounce, 0)
    return to_unit(r0)
