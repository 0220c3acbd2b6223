"""GPU: scripts/fuzz_parity.py for a fixed number of cases per stage -- randomly drawn sizes / parameters of the median,
the GFT / FAST / AGAST / ORB detectors and the Hamming matchers through the C ABI against the CPU oracle, bit for bit (the tool
exits non-zero on the first difference and prints the case to replay)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_parity_fixed_cases():
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--cases", "25", "--minutes", "6", "--seed", "20261004"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "all identical" in out.stdout.splitlines()[-1], out.stdout[-500:]
