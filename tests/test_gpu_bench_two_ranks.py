"""Two-rank readiness on a one-GPU box: bench.py launched the way the driver launches it for N > 1
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...`), both ranks wrapped onto the one GPU with
the gloo backend carrying the weight blob (ranks that share a device cannot form an RCCL communicator).  What this pins is the
multi-process control flow of the bench -- rendezvous on 127.0.0.1, rank 0 parses / rank 1 builds from the broadcast blob,
per-rank session shards, barrier + max-over-ranks timing, one JSON line from rank 0 -- so that the first 8-GPU run is not the
first time it executes.  The RCCL leg itself (aprilx_model_broadcast) is covered with a communicator of one in
test_gpu_parity.py::test_model_broadcast_in_library; what an N-GPU SCALE line must show is in DESIGN.md section 7."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_bench_two_ranks_on_one_gpu(built, v0_model):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--sessions", "16",
           "--steps", "3", "--warmup", "2", "--no-sweep", "--no-cpu-baseline", "--no-config5", "--steady-steps", "12", "--profile-steps", "2"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", APRIL_MODEL=v0_model["path"],
               APRIL_MAX_SESSIONS="64", APRIL_MAX_BATCH="256")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, (out[-1500:], r.stderr.decode()[-3000:])
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out[-2000:]            # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["sessions_total"] == 32 and j["scaling"] == "weak"
    assert j["steps"] == 3 and j["warmup"] == 2 and j["value"] > 0 and j["ms_per_step"] > 0
    assert j["weight_broadcast"]["ranks"] == 2 and "gloo" in j["weight_broadcast"]["where"]
    assert j["rccl_fallback"] is False
    assert j["replay_mismatch"] == 0
    assert j["steady"]["steps"] == 8 and j["steady"]["ms_per_step"] > 0
    # value = the units ALL ranks processed / the slowest rank's time
    assert abs(j["value"] - 32 * 3 * 0.1 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-2
    assert len(j["rccl_libs_mapped"]) <= 1, j["rccl_libs_mapped"]       # one RCCL build per process (shared SONAME librccl.so.1)
    # every rank reports its own view of the load: device, load time, what the library used, which RCCL it has mapped
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all("error" not in r for r in pr), pr
    assert all(r["model_load_s"] >= 0 and len(r["rccl_libs_mapped"]) <= 1 and r["rccl_fallback"] is False for r in pr)
    assert pr[0]["pid"] != pr[1]["pid"]
