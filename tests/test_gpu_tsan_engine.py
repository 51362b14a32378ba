"""The REAL engine under ThreadSanitizer on the GPU (round 5).  tests/test_sched_sanitizers.py covers the host scheduler against a fake
engine; here the same scenario driver (tests/sched_harness/driver.cc: lock-step / pipelined / asynchronous ingest with identical
callbacks, eight client threads, a monitoring thread that reads statistics, latencies and an idle session's state while others
stream, session churn, frees from inside handlers, queue overflow, irregular and five-second feeds) is linked against the product's
own host code -- april_api.cc, engine.cc (capture mutex, three streams and their dependency flags, slot resets from client threads,
the process-wide lock between graph captures and legacy-stream users), session.cc -- compiled with -fsanitize=thread, and the
product's device objects; the uninstrumented HIP / HSA / RCCL runtimes are suppressed (tools/tsan_gpu.supp).
What it found when it first ran: a client thread's hipMemcpy (aprilx_session_context / aprilx_session_read_frames on an idle session)
aborts the process while the stepping thread captures a graph -> hip_legacy_mutex (engine.h); Engine::kernels_per_step() and
Engine::live_slots() read plain counters the stepping thread / other clients write -> atomics."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_real_engine_scenarios_under_tsan(built, tiny_model):
    exe = os.path.join(ROOT, "tools", "tsan_gpu_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_tsan_gpu_driver.sh")], timeout=1500)
    env = dict(os.environ, APRIL_LOG_LEVEL="NONE",
               TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 suppressions=%s" % os.path.join(ROOT, "tools", "tsan_gpu.supp"))
    for extra in ({}, {"APRIL_GPU_DEVICES": "0,0,0"}, {"APRIL_PIPELINE": "1"}):      # default; three engines (captures of one beside the others); one flight at a time
        r = subprocess.run([exe, tiny_model["path"]], env=dict(env, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        out = r.stdout.decode() + r.stderr.decode()
        assert "HARNESS ok" in out, (extra, out[-4000:])
        assert "WARNING: ThreadSanitizer" not in out, (extra, out[:6000])
        assert r.returncode == 0, (extra, out[-2000:])


def test_real_engine_scenarios_under_asan_ubsan(built, tiny_model):
    """the same driver and engine under -fsanitize=address,undefined (host heap / stack / UB in engine.cc's staging, index rings, plans)"""
    exe = os.path.join(ROOT, "tools", "asan_gpu_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_tsan_gpu_driver.sh")], timeout=1500)
    env = dict(os.environ, APRIL_LOG_LEVEL="NONE", ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="halt_on_error=1")
    for extra in ({}, {"APRIL_GPU_DEVICES": "0,0"}):
        r = subprocess.run([exe, tiny_model["path"]], env=dict(env, **extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        out = r.stdout.decode() + r.stderr.decode()
        assert "HARNESS ok" in out and r.returncode == 0, (extra, out[-4000:])
        assert "AddressSanitizer" not in out and "runtime error" not in out, (extra, out[:6000])
