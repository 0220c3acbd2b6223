"""CPU: bench.py's launch logic -- `--gpus N` without a launcher environment starts N ranks itself as child
processes (never an exec), a launcher whose WORLD_SIZE disagrees with --gpus is an error, not a warning."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_self_launch_starts_n_ranks_as_children(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--pairs-per-gpu", "64"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    args = bench.parse()
    assert bench.self_launch(args) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--pairs-per-gpu", "64"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
