"""The stepping thread keeps two flights in the air (csrc/session.cc Scheduler::loop): flight k + 1 is launched before flight k
is completed whenever its work is already queued.  Which feeds share a flight, and whether a caller lent its buffers or had them
copied, must not change a single callback: the same sessions are streamed in separate processes through the lock-step group feed
(aprilx_feed_many), the pipelined group feed at depth 1 and 2 (aprilx_feed_many_pipelined), as asynchronous sessions, and with
pipelining switched off in the library (APRIL_PIPELINE=1); every token, log-probability (bit for bit), flag and time must agree.
Also here: the bound on the samples staged per pass (FbankFrameDesc::pcm_off is 32 bits) crossed with small numbers."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(path, nsess, steps, mode, feed=1600, flush=1, **env):
    e = dict(os.environ, APRIL_MAX_SESSIONS="512", APRIL_MAX_BATCH="2048")
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_worker.py"), path, str(nsess), str(steps), mode, str(feed), str(flush)],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("DIGEST")][-1].split()
    lat = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("LATENCY")][-1].split()
    eng = [int(x) for x in [ln for ln in r.stdout.decode().splitlines() if ln.startswith("ENGINES")][-1].split()[1:]]
    return dict(digest=line[1], chunks=int(line[2]), mismatch=int(line[3]), calls=int(line[4]), tokens=int(line[5]), flights=int(line[6]),
                lat_n=int(lat[1]), lat_min=float(lat[2]), lat_max=float(lat[3]), engines=eng)


def test_ingest_modes_give_the_same_callbacks(built, medium_model):
    path = medium_model["path"]
    ref = run(path, 24, 12, "sync")
    assert ref["chunks"] > 0 and ref["calls"] > 0 and ref["mismatch"] == 0
    for mode, env in (("pipe2", {}), ("pipe1", {}), ("async", {}), ("pipe2", {"APRIL_PIPELINE": 1}), ("sync", {"APRIL_PIPELINE": 1}), ("pipe2", {"APRIL_NO_GRAPHS": 1})):
        got = run(path, 24, 12, mode, **env)
        assert got["mismatch"] == 0 and got["chunks"] == ref["chunks"], (mode, env, got)
        assert got["digest"] == ref["digest"], "callbacks differ in mode %s %r" % (mode, env)


def test_pipelined_feeds_at_aprilv0_dims(built, v0_model):
    path = v0_model["path"]
    a = run(path, 64, 8, "sync")
    b = run(path, 64, 8, "pipe2")
    assert a["mismatch"] == 0 and b["mismatch"] == 0 and a["chunks"] == b["chunks"] > 0
    assert a["digest"] == b["digest"]
    assert b["flights"] >= 5          # (the first feeds may share a flight when the first launch -- plan building -- is slow)


def test_irregular_feeds_pipelined(built, medium_model):
    """feeds that are not a whole number of frames (1234 samples): chunk counts per flight vary (0..1 chunk), flights with
    nothing to do for some sessions"""
    path = medium_model["path"]
    a = run(path, 5, 40, "sync", feed=1234)
    b = run(path, 5, 40, "pipe2", feed=1234)
    c = run(path, 5, 40, "async", feed=1234)
    assert a["mismatch"] == b["mismatch"] == c["mismatch"] == 0 and a["chunks"] == b["chunks"] == c["chunks"] > 0
    assert a["digest"] == b["digest"] == c["digest"]


def test_staged_samples_per_pass_are_bounded(built, medium_model):
    """3 s per session in ONE feed, 6 sessions = 288 000 samples; with the limit at 20 000 samples a pass stages one session's
    worth of frames at most and the rest waits for the next pass -- the callbacks are those of the unbounded run"""
    path = medium_model["path"]
    a = run(path, 6, 1, "sync", feed=48000)
    b = run(path, 6, 1, "sync", feed=48000, APRIL_STAGE_LIMIT_SAMPLES=20000)
    c = run(path, 6, 1, "pipe2", feed=48000, APRIL_STAGE_LIMIT_SAMPLES=20000)
    assert a["mismatch"] == b["mismatch"] == c["mismatch"] == 0 and a["chunks"] == b["chunks"] == c["chunks"] > 0
    assert a["digest"] == b["digest"] == c["digest"]


def test_flights_with_several_steps_pipelined(built, medium_model):
    """0.5 s per feed with the layer-major path switched off: every flight holds TWO feed wavefronts (7 + 5 chunks).  Only the
    first step of a flight may use the three streams -- the second one shares the flight's buffers with it (this is what a
    flush of many sessions looks like on an fp16 engine: tests/test_gpu_f16.py caught it at 512 sessions)"""
    path = medium_model["path"]
    a = run(path, 16, 6, "sync", feed=8000)
    b = run(path, 16, 6, "pipe2", feed=8000, APRIL_LM_MIN_CHUNKS=0)
    c = run(path, 16, 6, "async", feed=8000, APRIL_LM_MIN_CHUNKS=0)
    assert a["mismatch"] == b["mismatch"] == c["mismatch"] == 0 and a["chunks"] == b["chunks"] == c["chunks"] > 0
    assert a["digest"] == b["digest"] == c["digest"]


def test_feed_latency_is_stamped_inside_the_library(built, medium_model):
    """aprilx_model_feed_latency: one hand-over -> delivery latency per completed tick.  Lock-step feeds: one tick per feed call;
    pipelined feeds: feeds may share a tick, never more ticks than feeds; every latency is positive and far below a second here."""
    path = medium_model["path"]
    a = run(path, 8, 10, "sync", flush=0)
    assert a["lat_n"] == 10 and 0.0 < a["lat_min"] <= a["lat_max"] < 2000.0, a
    b = run(path, 8, 10, "pipe2", flush=0)
    assert 1 <= b["lat_n"] <= 10 and 0.0 < b["lat_min"] <= b["lat_max"] < 2000.0, b


def test_eight_engines_on_one_device(built, medium_model, v0_model):
    """The per-GPU sharding of BASELINE configs[3] on the one GPU a test box has: APRIL_GPU_DEVICES=0,0,0,0,0,0,0,0 builds eight
    engines (weights copied device to device from the first), each with its own stepping thread and streams; least-loaded placement
    (reference load site src/april_model.c:57-61 -> april_api.cc aas_create_session) deals the sessions out evenly, the eight
    threads step concurrently, and every session produces exactly the callbacks it produces on a single engine."""
    for path, nsess, steps in ((medium_model["path"], 64, 10), (v0_model["path"], 256, 4)):
        one = run(path, nsess, steps, "pipe2")
        eight = run(path, nsess, steps, "pipe2", APRIL_GPU_DEVICES="0,0,0,0,0,0,0,0", APRIL_MAX_SESSIONS=64)
        assert one["mismatch"] == eight["mismatch"] == 0 and one["chunks"] == eight["chunks"] > 0
        assert one["digest"] == eight["digest"], "callbacks differ between one engine and eight"
        used = [c for c in eight["engines"] if c]
        assert len(used) == 8 and len(set(used)) == 1, "sessions were not dealt evenly: chunks per engine %r" % eight["engines"]
        assert [c for c in one["engines"] if c] == [one["chunks"]]


def test_gates_clock_times_replayed_launches_and_changes_nothing(built, v0_model):
    """aprilx_model_profile(model, 2) -- bench.py's roofline clock: feeds run as always (graphs replay, flights overlap) while the gates
    kernels of the feed wavefronts stamp their own start and end.  Every callback stays the same, and the clock reports the launches,
    their rows and a plausible time."""
    path = v0_model["path"]
    ref = run(path, 64, 12, "pipe2")
    e = dict(os.environ, APRIL_MAX_SESSIONS="512", APRIL_MAX_BATCH="2048", APRIL_TEST_PROFILE="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_worker.py"), path, "64", "12", "pipe2", "1600", "1"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().splitlines()
    dig = [ln for ln in out if ln.startswith("DIGEST")][-1].split()
    clk = [ln for ln in out if ln.startswith("GCLOCK")][-1].split()
    assert dig[1] == ref["digest"] and int(dig[3]) == 0
    launches, ms, rows = int(clk[1]), float(clk[2]), int(clk[3])
    # 8 clocked feeds of 2..3 chunks x 12 layers: a wavefront of T chunks has L + T - 1 gates launches; rows = sessions x layer-chunks
    assert launches >= 8 * 13, clk
    assert rows % 64 == 0 and rows // 64 >= 8 * 2 * 12, clk
    assert 1e-3 * launches < ms < 1.0 * launches, clk                                        # 1 us .. 1 ms per launch
