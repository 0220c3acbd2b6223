"""CPU: pin the matching oracle with hand-computable known answers (the reference holds no
golden vectors for this path)."""
import numpy as np

import oracle


def _desc(bits_set):
    d = np.zeros(32, dtype=np.uint8)
    for b in bits_set:
        d[b // 8] |= 1 << (b % 8)
    return d


def test_known_distances_and_first_minimum_wins():
    q = np.stack([_desc([]), _desc(range(256)), _desc([0, 9, 200])])
    t = np.stack([_desc([1]), _desc([]), _desc([2]), _desc(range(255)), _desc([])])
    keys = oracle.match_hamming(q, t, k=1)[:, 0]
    dist, idx = keys >> 20, keys & 0xFFFFF
    # q0 = zeros: distances 1,0,1,255,0 -> min 0 first at train 1 (not 4)
    assert (dist[0], idx[0]) == (0, 1)
    # q1 = ones: distances 255,256,255,1,256 -> train 3
    assert (dist[1], idx[1]) == (1, 3)
    # q2 has 3 bits: distances 4,3,4,252(=255-3+0)...,3 -> 3 at train 1
    assert (dist[2], idx[2]) == (3, 1)


def test_knn2_order_and_missing():
    q = np.stack([_desc([])])
    t = np.stack([_desc([0, 1]), _desc([0]), _desc([5]), _desc([])])
    k2 = oracle.match_hamming(q, t, k=2)
    assert [(int(x) >> 20, int(x) & 0xFFFFF) for x in k2[0]] == [(0, 3), (1, 1)]
    one = oracle.match_hamming(q, t[:1], k=2)
    assert (int(one[0, 0]) >> 20, int(one[0, 0]) & 0xFFFFF) == (2, 0)
    assert int(one[0, 1]) == oracle.KEY_NONE
    none = oracle.match_hamming(q, t[:0], k=1)
    assert int(none[0, 0]) == oracle.KEY_NONE


def test_against_numpy_bruteforce():
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (57, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (91, 32), dtype=np.uint8)
    t[17] = q[5]
    t[40] = q[5]  # duplicate: first wins
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=-1).sum(-1)
    keys = oracle.match_hamming(q, t, k=1)[:, 0]
    assert np.array_equal(keys & 0xFFFFF, d.argmin(1))  # argmin returns the first minimum
    assert np.array_equal(keys >> 20, d.min(1))
    assert keys[5] == 17


def test_sort_is_stable_by_distance():
    dist = np.array([5, 3, 5, 0, 3, 3, 256, 0], dtype=np.uint32)
    keys = (dist << 20) | np.arange(8, dtype=np.uint32)[::-1]  # train idx must not affect order
    order = oracle.sort_matches(keys)
    assert order.tolist() == [3, 7, 1, 4, 5, 0, 2, 6]
    assert order.tolist() == sorted(range(8), key=lambda i: dist[i])
    assert oracle.sort_matches(np.zeros(0, np.uint32)).shape == (0,)


def test_radius_match_against_numpy_and_truncation():
    rng = np.random.default_rng(9)
    q = rng.integers(0, 256, (23, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    t[7] = t[90] = q[3]                       # two exact matches of query 3: train order breaks the tie
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=-1).sum(-1)
    r = 118
    keys, counts = oracle.match_radius(q, t, r, 64)
    assert np.array_equal(counts, (d <= r).sum(1))              # inclusive radius
    for i in range(23):
        want = sorted((int(d[i, j]) << 20) | j for j in range(150) if d[i, j] <= r)
        assert keys[i, : min(len(want), 64)].tolist() == want[:64]
        assert (keys[i, len(want):] == oracle.KEY_NONE).all()
    assert keys[3, :2].tolist() == [7, 90]
    k2, c2 = oracle.match_radius(q, t, 256, 5)                  # everything matches: the 5 smallest keys are kept
    assert (c2 == 150).all()
    assert k2[0].tolist() == sorted((int(d[0, j]) << 20) | j for j in range(150))[:5]
    k0, c0 = oracle.match_radius(q, t[:0], 10, 4)
    assert (c0 == 0).all() and (k0 == oracle.KEY_NONE).all()


def test_l2_matching_of_float_descriptors_against_numpy():
    """BFMatcher() with NORM_L2 on float descriptors: the nearest two train rows per query, ties to the lower index; the
    distances agree with a float64 evaluation to float32 rounding, and the first-index rule is exercised by duplicates."""
    rng = np.random.default_rng(8)
    for dim in (128, 64, 61, 3):
        q = rng.normal(size=(40, dim)).astype(np.float32)
        t = rng.normal(size=(55, dim)).astype(np.float32)
        t[20] = t[7]                                               # a duplicate train row: index 7 must win
        q[0] = t[7]
        keys = oracle.match_l2(q, t, k=2)
        d = np.sqrt(((q[:, None, :].astype(np.float64) - t[None].astype(np.float64)) ** 2).sum(-1))
        idx = np.argsort(d, axis=1, kind="stable")[:, :2]
        assert np.array_equal((keys & 0xFFFFFFFF).astype(np.int64), idx)
        got = (keys >> 32).astype(np.uint32).view(np.float32)
        assert np.abs(got - np.take_along_axis(d, idx, 1)).max() < 1e-5
        assert got[0, 0] == 0.0 and (keys[0] & 0xFFFFFFFF).tolist() == [7, 20]
    assert np.array_equal(oracle.match_l2(q[:2], t[:1], k=2)[:, 1], np.full(2, 2 ** 64 - 1, dtype=np.uint64))   # no second row
