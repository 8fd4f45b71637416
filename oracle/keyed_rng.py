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

    Philox-4x32-7( counter = (x, y, sample, slot), key = (seed_lo, seed_hi) )
    slot = purpose | bounce << 4 | call << 12

where (x, y) is the pixel, sample the global sample index of the path, bounce
the bounce number of the scatter event, purpose one of the constants below
and call the number of the block inside that stream.  One Philox block
yields four 32-bit words w0..w3; a uniform double in [0, 1) is w * 2**-32 and
uniform(low, high) = low + (high - low) * u evaluated in float64, which is
what numpy's Generator.uniform computes from its own u (whose resolution is
53 bits where this one's is 32, or 21 for the unit-sphere candidates).

A rejection loop is paid for by the unluckiest of the 64 paths that share a
GPU wavefront, so one block yields two candidates:

    JITTER  call 0: (w0, w1) sub-pixel offset in [0,1)^2 (camera.py:130);
            (w2, w3) lens-disk candidate 0 in [-1,1)^2 (camera.py:138)
    LENS    call c: (w0, w1) lens-disk candidate 1 + 2c, (w2, w3) candidate
            2 + 2c (camera.py:141-148)
    UNITVEC call c: unit-sphere candidates 2c and 2c + 1 in [-1,1)^3
            (materials.py:407-426), coordinate = -1 + v * 2**-20 with a
            21-bit integer v: the even candidate takes the top 21 bits of
            w0, w1, w2; the odd one the low 11 bits of each followed by 10
            bits of w3 (bits 22.., 12.., 2..)
    SCHLICK call 0: w0, reflect/refract coin in [0,1) (materials.py:331)
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


# Rounds of the keyed streams: seven, the fewest with which Philox-4x32 passes BigCrush (Salmon et al.,
# SC'11; ten is Random123's default).  The device generator uses the same number.
ROUNDS = 7


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=ROUNDS):
    """Vectorised Philox-4x32 with `rounds` rounds.  Counter words are array-likes of equal
    (broadcastable) shape holding values < 2**32; key words are Python ints.
    Returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *(np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    )
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
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
    return philox4x32(
        x, y, sample, slot(purpose, bounce, attempt),
        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
    )


def to_unit(words):
    """uint32 -> float64 uniform in [0, 1)."""
    return words.astype(np.float64) * (1.0 / 4294967296.0)


def jitter(seed, x, y, sample):
    """(n, 2) float64 sub-pixel offsets in [0, 1)."""
    w0, w1, _, _ = keyed_words(seed, x, y, sample, JITTER, 0, 0)
    return np.stack([to_unit(w0), to_unit(w1)], axis=-1)


def lens_candidate(seed, x, y, sample, attempt):
    """(n, 2) float64 lens-disk candidate number `attempt` in [-1, 1)^2."""
    if attempt == 0:
        _, _, a, b = keyed_words(seed, x, y, sample, JITTER, 0, 0)
    else:
        w = keyed_words(seed, x, y, sample, LENS, 0, (attempt - 1) // 2)
        a, b = (w[0], w[1]) if (attempt - 1) % 2 == 0 else (w[2], w[3])
    return -1.0 + 2.0 * np.stack([to_unit(a), to_unit(b)], axis=-1)


def unit_candidate(seed, x, y, sample, bounce, attempt):
    """(n, 3) float64 unit-sphere candidate number `attempt` in [-1, 1)^3."""
    w = keyed_words(seed, x, y, sample, UNITVEC, bounce, attempt // 2)
    if attempt % 2 == 0:
        v = [w[a] >> np.uint32(11) for a in range(3)]
    else:
        v = [((w[a] & np.uint32(0x7FF)) << np.uint32(10)) | ((w[3] >> np.uint32(22 - 10 * a)) & np.uint32(0x3FF))
             for a in range(3)]
    return -1.0 + np.stack(v, axis=-1).astype(np.float64) * (1.0 / 1048576.0)


def keyed_unit_vectors(seed, x, y, sample, bounce):
    """Keyed version of the reference's numpy_random_unit_vecs
    (materials.py:401-431): draw in the cube [-1,1)^3 as float64, cast to
    float32, accept iff 1e-5 <= |v|^2 <= 1 (float32 arithmetic), redraw only
    the rejected entries with their next candidate, normalise in float32.
    x, y, sample are 1-D integer arrays naming the path of every entry."""
    x = np.asarray(x)
    y = np.asarray(y)
    sample = np.asarray(sample)
    n = x.shape[0]
    vecs = np.zeros((n, 3), dtype=np.single)
    pending = np.arange(n)
    attempt = 0
    while pending.size:
        cand = unit_candidate(seed, x[pending], y[pending], sample[pending], bounce, attempt).astype(np.single)
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
