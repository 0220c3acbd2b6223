"""GPU: bench.py's multi-rank code path -- init_process_group("nccl") (= RCCL), the per-step all_gather of the
[B,16] records on the engine's results, barrier + max-over-ranks timing -- rehearsed with ONE launcher-started rank
(the one-GPU box cannot hold two RCCL ranks); the N > 1 sharding itself is covered on gloo by test_parallel_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_bench_single_rank_through_rccl(tmp_path):
    port = _free_port()
    env = dict(os.environ, SOSVO_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--pairs-per-gpu", "8", "--streams", "2", "--iters", "200", "--features-per-mask", "100", "--no-cpu",
           "--render-workers", "2"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["tracked_ok"] >= 6, d["config"]
