"""The context cache of `Propagator` is keyed on CONTENT: an in-place edit of one coefficient of a large table
must change the key - the round-3 memo revalidated a strided sample of 256 elements and missed it."""
import hashlib

import numpy as np

from nyx_amd import propagator as P


def _key(obj):
    h = hashlib.blake2b(digest_size=16)
    P._feed(h, obj)
    return h.digest()


def test_in_place_edit_of_one_stokes_coefficient_changes_the_key():
    c = np.random.default_rng(0).standard_normal(71 * 72 // 2)   # a 70x70 field: 20 KB, over the small-array threshold
    k0 = _key(c)
    assert _key(c) == k0
    j2 = 2 * 3 // 2 + 0
    assert j2 % max(1, c.size // 256) != 0          # (not one of the elements the old sample looked at)
    c[j2] *= 1.0 + 1e-9
    assert _key(c) != k0


def test_only_frozen_arrays_are_memoised():
    a = np.arange(4096, dtype=np.float64)
    P._ARRAY_DIGESTS.clear()
    d0 = P._array_digest(a)
    assert not P._ARRAY_DIGESTS                       # writable: hashed in full, nothing remembered
    a.setflags(write=False)
    assert P._array_digest(a) == d0 and len(P._ARRAY_DIGESTS) == 1
    v = a[::2]                                        # a read-only view of a frozen owner is frozen too
    P._array_digest(v)
    assert len(P._ARRAY_DIGESTS) == 2
    a.setflags(write=True)
    a[1] = -1.0
    d1 = P._array_digest(a)                           # seen writable: the memo entry is dropped
    assert d1 != d0 and len(P._ARRAY_DIGESTS) == 1
    a.setflags(write=False)
    assert P._array_digest(a) == d1
    w = np.arange(4096, dtype=np.float64)
    ro = w[:]
    ro.setflags(write=False)                          # read-only VIEW of a writable owner: not frozen
    n = len(P._ARRAY_DIGESTS)
    P._array_digest(ro)
    assert len(P._ARRAY_DIGESTS) == n


def test_a_freed_tables_digest_is_not_inherited_by_its_successor():
    """free a frozen table, load another frozen table of the same shape - the allocator may well hand out the
    same address - and the memo must NOT answer with the first table's digest: the entry dies with the owner, and a hit is
    revalidated against a sample of the content."""
    import gc
    P._ARRAY_DIGESTS.clear()
    a = np.random.default_rng(1).standard_normal(8192)
    a.setflags(write=False)
    da = P._array_digest(a)
    assert len(P._ARRAY_DIGESTS) == 1
    key = next(iter(P._ARRAY_DIGESTS))
    del a
    gc.collect()
    assert not P._ARRAY_DIGESTS                       # the owner is gone: so is its entry
    # the same address, shape and strides answering for other content (forced: a stale entry planted under the new array's key)
    b = np.random.default_rng(2).standard_normal(8192)
    b.setflags(write=False)
    kb = (b.__array_interface__["data"][0], b.shape, b.strides, b.dtype.str)
    P._ARRAY_DIGESTS[kb] = (None, da, b"stale___")     # (no owner reference: only the sample stands guard)
    db = P._array_digest(b)
    assert db != da and db == hashlib.blake2b(b.tobytes(), digest_size=16).digest()
    assert key is not None
