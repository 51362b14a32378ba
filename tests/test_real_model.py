"""Real-model legs, skipped unless the environment provides them:
  APRIL_MODEL=/path/aprilv0_en-us.april   a model written by the reference's extra/export-april.py
  onnxruntime (Python module)             the reference's network backend (v1.13.1 CPU in the reference build)
With both, the GPU engine is compared with the ONNXRuntime-CPU path token for token (logits within 1e-3), which is the
north-star parity statement; with the model alone, the loader must accept the exporter's graphs."""
import os

import numpy as np
import pytest

from conftest import speech_like_pcm

MODEL = os.environ.get("APRIL_MODEL")
pytestmark = pytest.mark.skipif(not MODEL or not os.path.exists(MODEL), reason="APRIL_MODEL not set")


def test_real_model_loads_host_only(built):
    import april_asr_amd as A
    m = A.Model.load_host_only(MODEL)
    assert m.dims.n_layers > 0 and m.dims.vocab > 1
    m.close()


@pytest.mark.gpu
def test_real_model_token_exact_vs_onnxruntime(built):
    from oracle import ort_leg as OL
    if not OL.available():
        pytest.skip("onnxruntime not installed")
    import april_asr_amd as A
    from test_gpu_parity import assert_same_transcript, run_gpu
    pcm = speech_like_pcm(10.0, seed=3, silence=(4.0, 6.5))
    ref = OL.OrtSession(MODEL, trace_logits=4000)
    for o in range(0, pcm.size, 1600):
        ref.feed(pcm[o:o + 1600])
    ref.flush()
    gm = A.Model(MODEL)
    want = [(t, [(gm.token(i).encode(), lp, fl, ms) for (i, lp, fl, ms) in toks]) for t, toks in ref.events]
    got, lg, n = run_gpu(gm, pcm, 1600)
    assert n == ref.chunks()
    assert lg.shape == ref.logits().shape and np.abs(lg - ref.logits()).max() < 1e-3
    assert_same_transcript(want, got)
    gm.close(); ref.close()
