"""N > 1 path on CPU: two ranks over gloo run the model-load broadcast exactly as bench.py does over
RCCL (rank 0 parses, everyone else builds from the blob); sessions are sharded, never exchanged."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_blob_broadcast_world2(built, tiny_model, tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), tiny_model["path"], str(tmp_path)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    a = json.load(open(tmp_path / "rank0.json")); b = json.load(open(tmp_path / "rank1.json"))
    for k in ("name", "params", "vocab", "tokens", "blob_bytes", "blob_sum"):
        assert a[k] == b[k], k
    assert a["tokens"] == tiny_model["tokens"]
    assert a["sessions"] == [0, 1, 2, 3] and b["sessions"] == [4, 5, 6, 7]
