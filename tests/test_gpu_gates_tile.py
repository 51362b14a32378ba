"""The fp32 gates GEMM has two schedules -- the hand-scheduled K-split 64 x 64 tiles (small launches) and GM_TILE (from ~2000
rows per launch, csrc/engine.cc gates_tile_rows) -- that must produce the same bits, since which one runs depends on the batch
size.  Two processes stream the same sessions with one schedule forced each; every logit and every callback must be identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, mode, nsess, steps):
    env = dict(os.environ, APRIL_GATES_TILE=str(mode), APRIL_MAX_SESSIONS="256", APRIL_MAX_BATCH="1024")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gates_tile_worker.py"), path, str(nsess), str(steps)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("DIGEST")][-1].split()
    return line[1], int(line[2]), int(line[3])


@pytest.mark.parametrize("which,nsess", [("medium", 48), ("v0", 96)])
def test_gates_schedules_are_bit_identical(built, medium_model, v0_model, which, nsess):
    path = (medium_model if which == "medium" else v0_model)["path"]
    a = run(path, 0, nsess, 6)
    b = run(path, 1, nsess, 6)
    assert a[1] == b[1] and a[1] > 0 and a[2] == 0 and b[2] == 0
    assert a[0] == b[0], "the two gates schedules differ"
