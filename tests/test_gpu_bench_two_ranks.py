"""GPU: BASELINE config 4's code path with TWO ranks of the real engine.  `python bench.py --gpus 2` (no launcher
around it) must start its two ranks itself, shard the job's pairs over them, gather the [2B,16] records and report
n_gpus 2; the gathered records must equal, bit for bit, those of ONE process running the same 2B pairs.  Both ranks
share the box's one GPU, so the exchange runs on gloo (RCCL refuses two ranks on one device); the collective code is
the same (`parallel.gather_records`)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--streams", "2", "--iters", "300", "--features-per-mask", "120", "--no-cpu",
          "--no-h2d", "--no-isolated", "--render-workers", "2"]


def _bench(extra, dump):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + COMMON + ["--dump-records", dump],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line for the whole job
    return json.loads(lines[0]), np.load(dump)


def test_two_self_launched_ranks_equal_one_process(tmp_path):
    d2, rec2 = _bench(["--gpus", "2", "--dist-backend", "gloo", "--pairs-per-gpu", "6"], str(tmp_path / "two.npy"))
    d1, rec1 = _bench(["--gpus", "1", "--pairs-per-gpu", "12"], str(tmp_path / "one.npy"))
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1
    assert d2["config"]["global_pairs_per_step"] == 12 and d2["config"]["pairs_per_gpu"] == 6
    assert d2["scaling"] == "weak" and d2["value"] > 0
    assert rec2.shape == rec1.shape == (12, 16)
    assert (rec1[:, 14] == 0).sum() >= 10, rec1[:, 12:16]   # the pairs are tracked, not merely equal
    assert np.array_equal(rec2, rec1), np.argwhere(rec2 != rec1)[:8]


def _visible_gpus():
    """Device count WITHOUT initialising the GPU in the pytest process (torch.cuda.device_count() does not, on this image)."""
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_rccl_ranks_equal_one_process(tmp_path):
    """The first time a box with two (or more) MI355X runs this suite, RCCL itself carries the N > 1 gather: two ranks on
    two GPUs, backend nccl (= RCCL over xGMI), and the gathered records must equal one process over the same 16 pairs bit
    for bit.  Skipped on the one-GPU boxes (there the same flow runs on gloo, above)."""
    d2, rec2 = _bench(["--gpus", "2", "--dist-backend", "nccl", "--pairs-per-gpu", "8"], str(tmp_path / "two_rccl.npy"))
    d1, rec1 = _bench(["--gpus", "1", "--pairs-per-gpu", "16"], str(tmp_path / "one16.npy"))
    assert d2["n_gpus"] == 2 and d2["config"]["global_pairs_per_step"] == 16 and d2["scaling"] == "weak"
    assert rec2.shape == rec1.shape == (16, 16)
    assert (rec1[:, 14] == 0).sum() >= 13, rec1[:, 12:16]
    assert np.array_equal(rec2, rec1), np.argwhere(rec2 != rec1)[:8]
