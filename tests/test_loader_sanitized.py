"""Round-6 loader additions under AddressSanitizer + UBSan on the CPU (tests/cpp/loader_tables_test.cc): build_fbank_tables for every frame
length 8 .. 2100 (factor lists, twiddle / root-of-unity table sizes the device passes index, values on the unit circle) and
pad_host_model on 60 random models whose widths are multiples of 16."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "april_asr_amd", "csrc")


def test_fbank_tables_and_width_padding_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "loader_tables_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", C, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "loader_tables_test.cc"),
                           os.path.join(C, "fbank_tables.cc"), os.path.join(C, "model_loader.cc"), os.path.join(C, "onnx_reader.cc"), "-o", exe], timeout=600)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "all checks passed" in out, out[-3000:]
