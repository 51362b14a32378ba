"""The conditional ONNXRuntime leg (oracle/ort_leg.py) without ONNXRuntime: the session driver + network hooks are run with
the oracle's own graph interpreter plugged in behind the `InferenceSession.run` interface; the transcript and the logits must
equal the plain oracle session's.  (With the real `onnxruntime` module and a real model: tests/test_real_model.py.)"""
import numpy as np

from conftest import speech_like_pcm


class _InterpreterAsOrt:
    """Quacks like onnxruntime.InferenceSession for one of the three graphs."""
    def __init__(self, model, which):
        self.m, self.which = model, which

    def run(self, outputs, feeds):
        if self.which == 0:
            e, h, c = self.m.encoder(feeds["x"], feeds["h"], feeds["c"])
            return [e, h, c]
        if self.which == 1:
            return [self.m.decoder(feeds["context"].ravel())]
        return [self.m.joiner(feeds["encoder_out"], feeds["decoder_out"])]


def test_ort_leg_plumbing_matches_plain_oracle(built, tiny_model):
    from oracle import orc_py as O
    from oracle import ort_leg as OL
    om = O.Model(tiny_model["path"])
    order = iter(range(3))
    s = OL.OrtSession(tiny_model["path"], make_session=lambda graph_bytes: _InterpreterAsOrt(om, next(order)), trace_logits=2000)
    ref = O.Session(om, trace_logits=2000)
    pcm = np.concatenate([speech_like_pcm(2.0, seed=1), np.zeros(16000 * 3, np.int16)])
    for o in range(0, pcm.size, 1600):
        s.feed(pcm[o:o + 1600]); ref.feed(pcm[o:o + 1600])
    s.flush(); ref.flush()
    assert s.chunks() == ref.chunks() and s.events == ref.events and len(s.events) > 0
    assert np.array_equal(s.logits(), ref.logits())
    s.close(); ref.close(); om.close()
