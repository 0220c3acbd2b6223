"""CPU: the FAST-9 score formulation of csrc/orb.hip (one polarity per lane on raw circle values, van Herk arc network) equals
the two-sided network on biased values -- tests/fast_network_check.cpp restates both in plain C++ and compares them on random
and adversarial circles (thresholds 10, 20 and random; saturated centres; circles with a brighter AND a darker arc)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_one_polarity_fast_score_equals_the_two_sided_network(tmp_path):
    exe = str(tmp_path / "fast_network_check")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(HERE, "fast_network_check.cpp")])
    out = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout and "ambiguous" in out.stdout, out.stdout
