"""GM_KW (csrc/kernels_gemm_kw.hip, round 5): the LSTM projection and the FFN-down GEMM at a few dozen to a thousand rows per launch
run with K split over eight waves whose activation rows arrive through wave-private LDS rings.  Which schedule runs depends on
the batch size (<= 32 rows: the weight streams / full-K tiles, 33 .. ~1000 rows: GM_KW with 16- or 32-row tiles, above: GM_TILE),
so all of them must produce the same bits:
  * tools/kw_bench compares every output (rows, state rows, sums of squares) of GM_KW against the round-4 schedules bitwise on 27
    shapes (ragged rows, z-batched 1..3 problems, both tile heights, the larger encoder's kz = 2 chunk form, FFN up on four waves);
  * whole sessions streamed with GM_KW switched off and on (and with either tile height pinned) give identical logits and callbacks,
    at sizes where the projection / FFN-down launches cross the schedule boundaries inside one run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, nsess, steps, **env):
    e = dict(os.environ, APRIL_MAX_SESSIONS="512", APRIL_MAX_BATCH="2048")
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gates_tile_worker.py"), path, str(nsess), str(steps)],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("DIGEST")][-1].split()
    return line[1], int(line[2]), int(line[3])


@pytest.mark.parametrize("which,nsess", [("medium", 40), ("v0", 96), ("v0", 300)])
def test_kw_schedule_is_bit_identical_on_whole_sessions(built, medium_model, v0_model, which, nsess):
    path = (medium_model if which == "medium" else v0_model)["path"]
    off = run(path, nsess, 5, APRIL_GM_KW=0)
    assert off[1] > 0 and off[2] == 0
    for env in ({"APRIL_GM_KW": 1}, {"APRIL_GM_KW": 1, "APRIL_KW_MT": 1}, {"APRIL_GM_KW": 1, "APRIL_KW_MT": 2}, {"APRIL_GM_KW": 1, "APRIL_KW_RING": 4}):
        on = run(path, nsess, 5, **env)
        assert on[1] == off[1] and on[2] == 0
        assert on[0] == off[0], "GM_KW %r: logits or callbacks differ from the round-4 schedules" % env


@pytest.mark.parametrize("xcd", ["0", "2"])
def test_kw_bench_every_output_bitwise(built, xcd):
    exe = os.path.join(ROOT, "tools", "kw_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_kw_bench.sh")], timeout=900)
    # (APRIL_KW_GATES: the gates form of GM_KW is a measurement form, off by default -- its outputs are still checked bit for bit here;
    # APRIL_KW_XCD=2: the 2 x 4 XCD order of the tiles, measured neutral and off by default, likewise)
    r = subprocess.run([exe, "20"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, APRIL_KW_GATES="1", APRIL_KW_GATES_MAX_ROWS="100000", APRIL_KW_XCD=xcd))
    out = r.stdout.decode()
    assert r.returncode == 0 and "all configurations bit-identical" in out, out[-3000:] + r.stderr.decode()[-1000:]
    assert out.count("bit-identical") > 60 and "MISMATCH" not in out
