"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (scripts/oracle_sanitize.sh): the reference
flow on one rendered frame pair per detector / solver plus the oracle's own CPU tests, against the sanitizer build
of oracle/*.c.  The HIP path is compared with this oracle, so undefined behaviour in it would be undefined parity.
CPU only (the GPU pool has no sanitizer runs)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have(libname):
    out = subprocess.run(["gcc", "-print-file-name=" + libname], capture_output=True, text=True).stdout.strip()
    return os.path.isabs(out) and os.path.exists(out)


@pytest.mark.skipif(not (_have("libasan.so") and _have("libubsan.so")), reason="gcc's sanitizer runtimes are not installed")
def test_oracle_clean_under_asan_ubsan():
    env = {k: v for k, v in os.environ.items() if k not in ("SOSVO_ORACLE_LIB", "LD_PRELOAD")}
    r = subprocess.run([os.path.join(ROOT, "scripts", "oracle_sanitize.sh")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail
    assert "RGB-D Kneip" in r.stdout and " passed" in r.stdout, tail
