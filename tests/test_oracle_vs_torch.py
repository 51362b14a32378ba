"""The CPU oracle's generic ONNX interpreter against a plain PyTorch fp32 statement of the same architecture
(tests/torch_ref.py): pins the oracle's operator semantics (Conv, MatMul/Gemm, Split order of the LSTM gates, the
LSTM projection, BasicNorm, DoubleSwish, grouped Conv1d of the decoder) on an implementation that shares no code
with it.  fp32 on both sides, different summation orders: 2e-5."""
import numpy as np
import pytest

import torch_ref as TR
from april_asr_amd import synth_model as SM


@pytest.mark.parametrize("which", ["tiny_model", "tiny_model_variant", "medium_model"])
def test_oracle_networks_match_torch(which, request):
    from oracle import orc_py as O
    mdl = request.getfixturevalue(which)
    dims, w = mdl["dims"], mdl["weights"]
    om = O.Model(mdl["path"])
    rng = np.random.RandomState(3)
    for _ in range(3):
        x = rng.uniform(-16, 8, size=(dims["seg"], dims["mel"])).astype(np.float32)
        h = rng.uniform(-0.5, 0.5, size=(dims["n_layers"], dims["d_model"])).astype(np.float32)
        c = rng.uniform(-1, 1, size=(dims["n_layers"], dims["hidden"])).astype(np.float32)
        e0, h0, c0 = om.encoder(x[None], h[:, None, :], c[:, None, :])
        e1, h1, c1 = TR.encoder(w, dims, x, h, c)
        assert np.abs(e0.ravel() - e1).max() < 2e-5 and np.abs(h0[:, 0, :] - h1).max() < 2e-5 and np.abs(c0[:, 0, :] - c1).max() < 2e-5
        ctx = rng.randint(0, dims["vocab"], size=dims["context"])
        d0 = om.decoder(ctx).ravel()
        assert np.abs(d0 - TR.decoder(w, dims, ctx)).max() < 2e-5
        ee = rng.uniform(-2, 2, size=dims["joiner"]).astype(np.float32)
        l0 = om.joiner(ee.reshape(1, 1, -1), d0.reshape(1, 1, -1)).ravel()
        assert np.abs(l0 - TR.joiner(w, dims, ee, d0)).max() < 5e-5
    om.close()
