"""GPU parity: sosvo_match_hamming / sosvo_sort_matches (through the C ABI) against the CPU
oracle, bit-exact (integer work).  Edge cases follow the reference call site semantics
(camera_models.py:404-446): empty sides, ties -> first train index, stable sort."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _run(ctx, q, t, nq, nt, k):
    dev = ctx.device
    keys = ctx.match_hamming(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev),
                             torch.from_numpy(nq).to(dev), torch.from_numpy(nt).to(dev), k=k)
    ctx.synchronize()
    return keys.cpu().numpy()


def _check(ctx, P, Sq, St, nq, nt, k, seed, planted=True):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, (P, Sq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (P, St, 32), dtype=np.uint8)
    nq = np.asarray(nq, dtype=np.int32)
    nt = np.asarray(nt, dtype=np.int32)
    if planted:  # plant noisy copies and exact duplicates so that ties really occur
        for p in range(P):
            m = min(nq[p], nt[p]) // 2
            if m > 1:
                src = rng.integers(0, nq[p], m)
                dst = rng.integers(0, nt[p], m)
                noise = (rng.random((m, 32, 8)) < 0.05)
                t[p, dst] = q[p, src] ^ np.packbits(noise, axis=-1)[..., 0]
                t[p, dst[: m // 4]] = q[p, src[: m // 4]]
                if nt[p] > 3:
                    t[p, nt[p] - 1] = t[p, 0]  # duplicate train rows: lowest index must win
    got = _run(ctx, q, t, nq, nt, k)
    for p in range(P):
        want = oracle.match_hamming(q[p, : nq[p]], t[p, : nt[p]], k=k)
        assert np.array_equal(got[p, : nq[p]], want), "problem %d" % p
    return q, t, nq, nt, got


def test_single_small(ctx):
    _check(ctx, 1, 64, 64, [64], [64], 1, seed=1)


def test_ragged_batch_and_edge_counts(ctx):
    nq = [0, 1, 5, 64, 65, 255, 256, 257, 300, 300]
    nt = [7, 1, 0, 300, 1, 256, 257, 255, 300, 2]
    _check(ctx, 10, 300, 300, nq, nt, 1, seed=2)
    _check(ctx, 10, 300, 300, nq, nt, 2, seed=3)


def test_bucket_sized_batch(ctx):
    # 48 stereo-bucket sized problems (12 masks x 2 frames x 2), ~170 keypoints each
    rng = np.random.default_rng(5)
    nq = rng.integers(100, 200, 48)
    nt = rng.integers(100, 200, 48)
    _check(ctx, 48, 200, 200, nq, nt, 1, seed=6)


def test_c2_size_split_path(ctx):
    # BASELINE config 2: 2000 x 2000, 1-NN; few problems -> the train range is split over
    # grid.z and merged with atomicMin on the packed key.
    _check(ctx, 2, 2048, 2048, [2000, 1777], [2000, 2048], 1, seed=7)


def test_c2_size_batched_no_split(ctx):
    P = 40
    rng = np.random.default_rng(8)
    _check(ctx, P, 2048, 2048, rng.integers(1500, 2049, P), rng.integers(1500, 2049, P), 1, seed=9)


def test_c3_size_knn2(ctx):
    # BASELINE config 3: 8000 x 8000 with the two nearest neighbours (ratio-test input)
    _check(ctx, 1, 8000, 8000, [8000], [8000], 2, seed=10)


def test_sort_matches_stable(ctx):
    P, S = 6, 2048
    nq = np.array([0, 1, 2, 500, 2000, 2048], dtype=np.int32)
    q, t, nq, nt, keys = _check(ctx, P, S, S, nq, [5, 0, 700, 1000, 2000, 64], 1, seed=11)
    dev = ctx.device
    order = ctx.sort_matches(torch.from_numpy(keys).to(dev), torch.from_numpy(nq).to(dev))
    ctx.synchronize()
    order = order.cpu().numpy()
    for p in range(P):
        want = oracle.sort_matches(keys[p, : nq[p], 0])
        assert np.array_equal(order[p, : nq[p]], want), "problem %d" % p
        d = keys[p, order[p, : nq[p]], 0] >> 20
        assert np.all(np.diff(d.astype(np.int64)) >= 0)


def test_properties_at_full_size(ctx):
    # size-independent properties: a query planted verbatim in the train set matches it with
    # distance 0; permuting the train rows permutes the matched indices' descriptors only.
    rng = np.random.default_rng(12)
    P, S = 3, 4096
    q = rng.integers(0, 256, (P, S, 32), dtype=np.uint8)
    perm = np.stack([rng.permutation(S) for _ in range(P)])
    t = np.stack([q[p][perm[p]] for p in range(P)])
    n = np.full(P, S, dtype=np.int32)
    keys = _run(ctx, q, t, n, n, 1)[..., 0]
    assert np.all(keys >> 20 == 0)
    inv = np.stack([np.argsort(perm[p]) for p in range(P)])
    assert np.array_equal((keys & 0xFFFFF).astype(np.int64), inv)


def test_radius_match_ragged_batch_with_slots_and_truncation(ctx):
    """sosvo_match_radius against the oracle: ragged problems sharing descriptor blocks through slots, an empty
    train set, a radius that matches everything (cap smallest keys kept), duplicates (train order breaks ties)."""
    rng = np.random.default_rng(21)
    S = 96
    blocks = rng.integers(0, 256, (4, S, 32), dtype=np.uint8)
    blocks[1, 5] = blocks[0, 3]
    blocks[1, 70] = blocks[0, 3]
    n = np.array([96, 80, 17, 0], dtype=np.int32)
    q_slot = np.array([0, 1, 2, 0, 3], dtype=np.int32)
    t_slot = np.array([1, 0, 1, 3, 0], dtype=np.int32)
    dev = ctx.device
    tb, tn = torch.from_numpy(blocks).to(dev), torch.from_numpy(n).to(dev)
    for radius, cap in ((110, 32), (256, 8), (0, 4)):
        keys, counts = ctx.match_radius(tb, tb, tn, tn, radius, cap, q_slot=torch.from_numpy(q_slot).to(dev),
                                        t_slot=torch.from_numpy(t_slot).to(dev))
        ctx.synchronize()
        keys, counts = keys.cpu().numpy(), counts.cpu().numpy()
        for p in range(5):
            nq, nt = n[q_slot[p]], n[t_slot[p]]
            wk, wc = oracle.match_radius(blocks[q_slot[p], :nq], blocks[t_slot[p], :nt], radius, cap)
            assert np.array_equal(counts[p, :nq], wc), (radius, p)
            assert np.array_equal(keys[p, :nq], wk), (radius, p)
    with pytest.raises(Exception):
        ctx.match_radius(tb, tb, tn, tn, 10, 513)


def test_l2_float_descriptors_match_the_oracle_bit_for_bit(ctx):
    """sosvo_match_l2 (BFMatcher() with NORM_L2, the reference's matcher for "SIFT" / "SURF" descriptors) against the
    oracle: keys (float32 distance bits, train index) identical, ragged problems, k = 1 and 2, odd dimensions."""
    rng = np.random.default_rng(21)
    for dim, k in ((128, 2), (64, 1), (61, 2), (5, 1)):
        nq = np.array([300, 1, 0, 77], dtype=np.int32)
        nt = np.array([257, 40, 9, 1], dtype=np.int32)
        q = rng.normal(size=(4, 300, dim)).astype(np.float32)
        t = rng.normal(size=(4, 260, dim)).astype(np.float32)
        t[0, 100] = t[0, 3]
        q[0, 5] = t[0, 3]
        dev = ctx.device
        keys = ctx.match_l2(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(nq).to(dev),
                            torch.from_numpy(nt).to(dev), k=k)
        ctx.synchronize()
        got = keys.cpu().numpy().view(np.uint64)
        for p in range(4):
            want = oracle.match_l2(q[p, :nq[p]], t[p, :nt[p]], k=k)
            assert np.array_equal(got[p, :nq[p]], want), (dim, k, p)
            # rows past the query count read "absent" (all ones), whatever the buffer held before the call
            assert (got[p, nq[p]:] == np.uint64(2 ** 64 - 1)).all(), (dim, k, p)


def test_feature_matcher_mirror_on_float_descriptors_and_flann(ctx):
    """FeatureMatcher("SIFT", ...) on float32 descriptors: best match, the k = 2 ratio rule (:421-423), the flattened
    2-NN lists; matcher_type "FLANN" answers with the exact neighbours."""
    from vo_single_camera_sos_amd.omnistereo.camera_models import FeatureMatcher
    rng = np.random.default_rng(22)
    t = rng.normal(size=(120, 128)).astype(np.float32)
    q = (t[rng.permutation(120)[:80]] + 0.05 * rng.normal(size=(80, 128))).astype(np.float32)
    keys = oracle.match_l2(q, t, k=2)
    d = (keys >> 32).astype(np.uint32).view(np.float32)
    i = (keys & 0xFFFFFFFF).astype(np.int64)
    qi, ti, di = FeatureMatcher("SIFT", "BF", 1, context=ctx).match_arrays(q, t)
    o = np.argsort(d[:, 0], kind="stable")
    assert np.array_equal(qi, o) and np.array_equal(ti, i[o, 0]) and np.array_equal(di, d[o, 0])
    qi, ti, di = FeatureMatcher("SIFT", "FLANN", 2, context=ctx).match_arrays(q, t)
    keep = d[o, 0] < d[o, 1] * np.float32(0.75)
    assert np.array_equal(qi, o[keep]) and np.array_equal(ti, i[o, 0][keep]) and len(qi) > 40
    qi, ti, di = FeatureMatcher("SURF", "BF", 2, context=ctx).match_arrays(q, t)
    o2 = np.argsort(d.reshape(-1), kind="stable")
    assert np.array_equal(qi, o2 // 2) and np.array_equal(ti, i.reshape(-1)[o2])
    ml = FeatureMatcher("ORB", "FLANN", 1, context=ctx).match(rng.integers(0, 256, (30, 32), dtype=np.uint8),
                                                               rng.integers(0, 256, (40, 32), dtype=np.uint8))
    assert len(ml) == 30


def test_matrix_pipe_tile_boundaries_and_extreme_descriptors(ctx):
    """The MFMA matcher's tiling (32-query tiles, 32-row blocks, 128-train LDS tiles, train-range splits merged by atomicMin,
    two query tiles per wave from a stride of 1024 on) at every boundary, 1-NN and 2-NN; and the arithmetic's extremes: all-zero
    and all-one descriptors (|q| + |t| - 2 q.t = 0 and 256, partial keys of both signs), identical rows (ties -> first
    train index)."""
    sizes = [0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 385]
    rng = np.random.default_rng(77)
    for stride in (400, 1100):
        P = len(sizes)
        nq = np.array(sizes, dtype=np.int32)
        nt = np.array(sizes[::-1], dtype=np.int32)
        if stride > 1024:
            nq[-1], nt[0], nq[3], nt[5] = 1100, 1100, 1025, 1024
        for k in (1, 2):
            _check(ctx, P, stride, stride, nq, nt, k, seed=int(stride + k))
    # extremes
    q = rng.integers(0, 256, (2, 96, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (2, 160, 32), dtype=np.uint8)
    q[0, 0] = 0
    q[0, 1] = 255
    t[0, 5] = 255
    t[0, 9] = 0
    t[0, 130] = 255          # a second all-ones row in the next LDS tile: the first one (index 5) must win for q[0, 1]
    t[1, :] = t[1, 0]        # every train row identical: index 0 wins everywhere, the second neighbour is index 1
    nq, nt = np.array([96, 96], np.int32), np.array([160, 160], np.int32)
    for k in (1, 2):
        got = _run(ctx, q, t, nq, nt, k)
        for p in range(2):
            assert np.array_equal(got[p], oracle.match_hamming(q[p], t[p], k=k)), (k, p)
    got = _run(ctx, q, t, nq, nt, 2)
    assert got[0, 0, 0] == (0 << 20 | 9) and got[0, 1, 0] == (0 << 20 | 5) and got[0, 1, 1] == (0 << 20 | 130)
    assert (got[1, :, 0] & 0xFFFFF == 0).all() and (got[1, :, 1] & 0xFFFFF == 1).all()
    assert int(np.unpackbits(q[0, 0] ^ t[0, 5]).sum()) == 256   # (the table holds the largest distance, too)
