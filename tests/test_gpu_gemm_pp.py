"""GM_PP (csrc/kernels_gemm_pp.hip, round 6): the fp16 gates and FFN-up GEMMs from a few hundred rows per launch run on 256 / 128 x 128
tiles whose two wave groups work one phase apart.  Smaller launches keep GM_TILE's forms, so both must produce the same bits
(canonical chains: kernels.h):
  * tools/pp_bench compares every output (cell state, binary16 rows, fp32 partial rows) of GM_PP -- planner's choice, 256-row and
    128-row tiles pinned -- against GM_TILE bitwise: gates with two A segments + BasicNorm scale + LSTM cell, FFN up + DoubleSwish, the
    layer-major halves (EPI_XPART, EPI_LSTM + p_add), ragged row counts, z-batched 1..3 problems, larger-encoder and aprilv0 dims;
  * whole fp16 sessions streamed with GM_PP switched off and on (and with either tile height pinned) give identical logits and
    callbacks at a size where the gates launches cross the schedule boundary inside one run (wavefront of 1..3 chunk steps);
  * the 256 x 192 form (csrc/kernels_gemm_pw.hip: N a multiple of 192, i.e. the larger encoder's gates) is one of pp_bench's pinned forms
    ("ppw") and is switched off / on under whole sessions of the larger encoder at 512 sessions, where the three-problem launches of
    the feed wavefront take it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, nsess, steps, **env):
    e = dict(os.environ, APRIL_MAX_SESSIONS="512", APRIL_MAX_BATCH="2048", APRIL_PRECISION="f16")
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gates_tile_worker.py"), path, str(nsess), str(steps)],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("DIGEST")][-1].split()
    return line[1], int(line[2]), int(line[3])


def test_pp_bench_every_output_bitwise(built):
    exe = os.path.join(ROOT, "tools", "pp_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_pp_bench.sh")], timeout=900)
    r = subprocess.run([exe, "10", "both"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "all forms bit-identical" in out, out[-3000:] + r.stderr.decode()[-1000:]
    assert out.count(" same") >= 80 and "DIFF" not in out and "MISMATCH" not in r.stderr.decode()


def test_pp_bench_stress_no_timing_dependence(built):
    """every ping-pong form 120 times, every other run beside a 512 MB device copy (the DMA pieces land late and out of their usual
    order): the bits of the GM_TILE form every time.  (The first 256 x 192 kernel read k block 1 before one group's pieces of it had
    been waited for -- right in a quiet benchmark, wrong inside the engine.)"""
    exe = os.path.join(ROOT, "tools", "pp_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_pp_bench.sh")], timeout=900)
    r = subprocess.run([exe, "120", "stress"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and out.count("all same") == 7 and "DIFF" not in out, out[-2000:] + r.stderr.decode()[-1000:]


@pytest.mark.parametrize("nsess", [96, 300])
def test_pp_schedule_is_bit_identical_on_whole_f16_sessions(built, v0_model, nsess):
    path = v0_model["path"]
    off = run(path, nsess, 5, APRIL_GM_PP=0)
    assert off[1] > 0 and off[2] == 0
    for env in ({"APRIL_GM_PP": 1}, {"APRIL_GM_PP": 1, "APRIL_PP_MT": 16}, {"APRIL_GM_PP": 1, "APRIL_PP_MT": 8}):
        on = run(path, nsess, 5, **env)
        assert on[1] == off[1] and on[2] == 0
        assert on[0] == off[0], "GM_PP %r: logits or callbacks differ from the GM_TILE forms" % env


def test_wide_pp_tiles_are_bit_identical_on_the_larger_encoder(built, large_model):
    """512 sessions of the larger encoder (BASELINE configs[4]): the gates launches of three layer problems run on 256 x 192 tiles
    (APRIL_PP_WIDE, default on), the others on 256 / 128 x 128 -- the same bits as with the wide form off and as GM_TILE alone."""
    path = large_model["path"]
    ref = run(path, 512, 3, APRIL_GM_PP=0)
    assert ref[1] > 0 and ref[2] == 0
    for env in ({"APRIL_PP_WIDE": 0}, {"APRIL_PP_WIDE": 1}, {"APRIL_PP_MT": 12}):
        got = run(path, 512, 3, **env)
        assert got[1] == ref[1] and got[2] == 0
        assert got[0] == ref[0], "GM_PP %r: logits or callbacks differ from the GM_TILE forms" % env
