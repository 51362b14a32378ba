import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- first, so the process uses ONE HIP runtime (torch's) for both torch and libaprilasr.so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS_DIR = os.path.dirname(os.path.abspath(__file__))
if TESTS_DIR not in sys.path:
    sys.path.insert(0, TESTS_DIR)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the product library and the oracle exist (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from april_asr_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        g.build()
    from oracle import orc_py as O
    O.build()
    return True


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("models")


@pytest.fixture(scope="session")
def tiny_model(model_dir, built):
    from april_asr_amd import synth_model as SM
    p = str(model_dir / "tiny.april")
    dims, w, toks = SM.write_model(p, SM.TINY_DIMS)
    return dict(path=p, dims=dims, weights=w, tokens=toks)


@pytest.fixture(scope="session")
def tiny_model_variant(model_dir, built):
    """Same weights as tiny_model, but spelled differently in ONNX (MatMul+Add instead of Gemm,
    BasicNorm eps as Exp(initializer) instead of a folded constant)."""
    from april_asr_amd import synth_model as SM
    p = str(model_dir / "tiny_variant.april")
    dims, w, toks = SM.write_model(p, SM.TINY_DIMS, variant=dict(lstm_gemm=False, fold_eps=False))
    return dict(path=p, dims=dims, weights=w, tokens=toks)


@pytest.fixture(scope="session")
def medium_model(model_dir, built):
    from april_asr_amd import synth_model as SM
    p = str(model_dir / "medium.april")
    dims, w, toks = SM.write_model(p, SM.MEDIUM_DIMS, seed=77)
    return dict(path=p, dims=dims, weights=w, tokens=toks)


@pytest.fixture(scope="session")
def narrow_model(model_dir, built):
    """widths that are multiples of 16 but not of 64 (d 144, cell 208, ffn 304, joiner 80, 48 conv channels): padded at load"""
    from april_asr_amd import synth_model as SM
    p = str(model_dir / "narrow.april")
    dims, w, toks = SM.write_model(p, SM.NARROW_DIMS, seed=3)
    return dict(path=p, dims=dims, weights=w, tokens=toks)


@pytest.fixture(scope="session")
def v0_model(model_dir, built):
    """aprilv0 dimensions (12 layers, 84 M parameters) -- ~10 s to write."""
    from april_asr_amd import synth_model as SM
    p = str(model_dir / "v0.april")
    dims, w, toks = SM.write_model(p, SM.APRILV0_DIMS)
    return dict(path=p, dims=dims, weights=None, tokens=toks)


@pytest.fixture(scope="session")
def large_model(model_dir, built):
    """BASELINE configs[4] "larger encoder": 16 layers, d_model 768, hidden 1536, ffn 3072, joiner 768 (~250 M parameters,
    ~1 GB on disk, ~40 s to write)."""
    from april_asr_amd import synth_model as SM
    p = str(model_dir / "large.april")
    dims, w, toks = SM.write_model(p, SM.LARGE_DIMS, seed=5)
    return dict(path=p, dims=dims, weights=None, tokens=toks)


def speech_like_pcm(seconds, seed=0, rate=16000, silence=(0.0, 0.0)):
    """Three seeded sinusoids x 4 Hz envelope + noise at -30 dB, <= 0.5 FS, optional digital silence span."""
    rng = np.random.RandomState(seed)
    n = int(seconds * rate)
    t = np.arange(n) / rate
    sig = np.zeros(n)
    for f in rng.uniform(200, 3000, size=3):
        sig += np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    sig *= 0.5 * (1 + np.sin(2 * np.pi * 4 * t)) / 3
    sig += rng.normal(0, 10 ** (-30 / 20), size=n)
    sig = np.clip(sig * 0.45, -0.5, 0.5)
    a, b = int(silence[0] * rate), int(silence[1] * rate)
    sig[a:b] = 0
    return (sig * 32767).astype(np.int16)
