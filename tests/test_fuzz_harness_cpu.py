"""CPU: the generators of scripts/fuzz_parity.py (the stages themselves need the GPU: tests/test_gpu_fuzz_script.py) -- the
harness must import without touching the GPU, and its inputs must cover what its docstring promises."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _harness():
    spec = importlib.util.spec_from_file_location("fuzz_parity_module", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generators_cover_the_promised_input_kinds():
    fp = _harness()
    assert set(fp.STAGES) >= {"median", "gft", "fast", "agast", "match", "radius", "orb", "unwrap", "ransac", "describe", "relpose",
                              "l2sort", "pipeline", "rgbd"}
    rng = np.random.default_rng(1)
    kinds = [fp._image(rng, 20, 30, k) for k in range(4)]
    assert all(im.shape == (20, 30) and im.dtype == np.uint8 for im in kinds)
    assert np.unique(kinds[0]).size > 100 and np.unique(kinds[3]).size == 1          # noise ... constant
    seen_empty = seen_full = False
    for _ in range(60):
        bits = fp._masks(rng, 20, 30, 3)
        assert bits.dtype == np.uint32 and bits.max() < 8
        seen_empty |= any(((bits >> m) & 1).sum() == 0 for m in range(3))
        seen_full |= bool((bits == 7).all())
    assert seen_empty and seen_full
    nq, nt = np.array([5, 0, 9], np.int32), np.array([7, 3, 0], np.int32)
    ties = False
    for _ in range(40):
        q, t = fp._descriptors(rng, 3, 9, 7, nq, nt)
        assert q.shape == (3, 9, 32) and t.shape == (3, 7, 32)
        ties |= bool((t[0, 0] == t[0, 1]).all())
    assert ties                                                                       # duplicate train rows occur
