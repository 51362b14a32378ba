"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares; struct
layouts and enum values equal the reference ABI (AprilToken 32 B {0,8,12,16,24},
AprilConfig 40 B {0,16,24,32}; captured from the reference's april_api.h and its ctypes binding)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from april_asr_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"APRIL_EXPORT[^;(]*?\b(a(?:am|as|prilx)_\w+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(built):
    L = _ffi.lib()
    ref_syms = declared_symbols("april_api.h")
    eng_syms = declared_symbols("aprilx_engine.h")
    assert sorted(ref_syms) == sorted(_ffi.EXPORTED_REFERENCE_SYMBOLS) and len(ref_syms) == 12
    assert set(eng_syms) == set(_ffi.EXPORTED_ENGINE_SYMBOLS)
    for s in ref_syms + eng_syms:
        assert hasattr(L, s), s


def test_only_the_abi_is_exported(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH]).decode()
    # every defined dynamic symbol of any kind (code, data, weak, vague-linkage: the HIP kernel handle objects used to leak
    # out as D / V symbols; csrc/exports.map keeps them local)
    names = {l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3}
    assert names == set(_ffi.EXPORTED_REFERENCE_SYMBOLS) | set(_ffi.EXPORTED_ENGINE_SYMBOLS), names


def test_struct_layouts_ctypes():
    assert C.sizeof(_ffi.AprilToken) == 32
    assert [getattr(_ffi.AprilToken, f).offset for f in ("token", "logprob", "flags", "time_ms", "reserved")] == [0, 8, 12, 16, 24]
    assert C.sizeof(_ffi.AprilConfig) == 40
    assert [getattr(_ffi.AprilConfig, f).offset for f in ("speaker", "handler", "userdata", "flags")] == [0, 16, 24, 32]
    assert C.sizeof(_ffi.AprilSpeakerID) == 16


def test_struct_layouts_compiled_header(tmp_path):
    """Compile include/april_api.h with gcc and print the layout the C compiler sees."""
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "april_api.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu ", sizeof(AprilToken), offsetof(AprilToken,token), offsetof(AprilToken,logprob),'
                   ' offsetof(AprilToken,flags), offsetof(AprilToken,time_ms), offsetof(AprilToken,reserved));'
                   'printf("%zu %zu %zu %zu %zu ", sizeof(AprilConfig), offsetof(AprilConfig,speaker), offsetof(AprilConfig,handler),'
                   ' offsetof(AprilConfig,userdata), offsetof(AprilConfig,flags));'
                   'printf("%d %d %d %d %d %d %d %d %d", APRIL_RESULT_RECOGNITION_PARTIAL, APRIL_RESULT_RECOGNITION_FINAL,'
                   ' APRIL_RESULT_ERROR_CANT_KEEP_UP, APRIL_RESULT_SILENCE, APRIL_TOKEN_FLAG_WORD_BOUNDARY_BIT,'
                   ' APRIL_TOKEN_FLAG_SENTENCE_END_BIT, APRIL_CONFIG_FLAG_ASYNC_RT_BIT, APRIL_CONFIG_FLAG_ASYNC_NO_RT_BIT, APRIL_VERSION);'
                   'return 0;}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert [int(x) for x in out] == [32, 0, 8, 12, 16, 24, 40, 0, 16, 24, 32, 1, 2, 3, 4, 1, 2, 1, 2, 1]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_layout_equals_reference_header(tmp_path):
    """Same probe against the reference's own april_api.h: identical numbers."""
    def probe(inc):
        src = tmp_path / "p.c"
        src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "april_api.h"\nint main(void){'
                       'printf("%zu %zu %zu %zu %zu %zu %zu %zu", sizeof(AprilToken), offsetof(AprilToken,logprob), offsetof(AprilToken,flags),'
                       ' offsetof(AprilToken,time_ms), sizeof(AprilConfig), offsetof(AprilConfig,handler), offsetof(AprilConfig,userdata),'
                       ' offsetof(AprilConfig,flags)); return 0;}\n')
        exe = tmp_path / "p"
        subprocess.check_call(["gcc", "-I", inc, str(src), "-o", str(exe)])
        return subprocess.check_output([str(exe)]).decode()
    assert probe(os.path.join(ROOT, "include")) == probe("/root/reference")


def test_fails_loudly_without_gpu(built, tiny_model, capfd):
    """No CPU path: on a machine without a HIP device the model cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _ffi.lib()
    L.aam_api_init(1)
    assert not L.aam_create_model(tiny_model["path"].encode())
    err = capfd.readouterr().err
    assert "HIP device" in err


def test_generated_gemm_loop_is_current():
    """april_asr_amd/csrc/gemm_mainloop_asm.inc is generated code: it must be what tools/gen_gemm_asm.py prints."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = subprocess.check_output([sys.executable, os.path.join(root, "tools", "gen_gemm_asm.py")]).decode()
    have = open(os.path.join(root, "april_asr_amd", "csrc", "gemm_mainloop_asm.inc")).read()
    assert want == have


def test_host_pool(tmp_path):
    """The helper-thread pool of the stepping thread (csrc/host_pool.h) under ThreadSanitizer: 12000 jobs of random
    size and grain, every index visited exactly once."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_pool_test")
    base = ["g++", "-std=c++17", "-O1", "-g", "-I" + os.path.join(root, "april_asr_amd", "csrc"),
            os.path.join(root, "tests", "cpp", "host_pool_test.cc"), "-o", exe, "-lpthread"]
    if subprocess.run(base[:5] + ["-fsanitize=thread"] + base[5:], capture_output=True).returncode != 0:
        subprocess.check_call(base)                       # toolchain without libtsan: plain build
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
