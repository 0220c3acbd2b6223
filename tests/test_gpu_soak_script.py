"""GPU: tests/soak_parity.py at a small size -- the bench configuration itself (1440 x 146 panoramas, ~2000 keypoints
per view, 2000 RANSAC iterations) through the GPU engine and through the oracle flow on host processes, every record
compared (the tool exits non-zero on any difference); likewise the RGB-D path with the reference's default "EPNP", the
generalised-P3P hypothesis generator, and the ORB and FAST detectors."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--rgbd", "EPNP"], ["--rgbd", "KNEIP"], ["--solver", "GP3P"], ["--detector", "ORB", "--kp-cap", "1280"],
                                   ["--detector", "FAST", "--kp-cap", "2048"]])
def test_soak_parity_small(extra):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "soak_parity.py"), "--pairs", "16", "--workers", "4", "--seed", "12321"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    last = [l for l in out.stdout.splitlines() if "soak" in l][-1]
    assert ("16 / 16 pairs identical" in last and "16 / 16 refined poses bit-identical" in last) or "16 / 16 records bit-identical" in last, last
