"""The control flow of the grouped RCCL weight broadcast (csrc/rccl_group.h, used by april_api.cc broadcast_local for one process
driving several GPUs) against a failing stub, on the CPU: a broadcast that fails inside the open group must be followed by the
group end FIRST and the communicator aborts AFTER it (APRIL_FAULT_RCCL=2 needs two real devices: never executed on hardware)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_broadcast_failure_paths_run_in_order(tmp_path):
    exe = str(tmp_path / "rccl_group_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "april_asr_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "rccl_group_test.cc"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    out = r.stdout.decode()
    assert r.returncode == 0 and "all paths in order" in out, out
    assert "A!" not in out          # no communicator was aborted while the group was open
