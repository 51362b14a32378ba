"""The reference's own test suite, mirrored (bindings/java/lib/lib/src/test/java/aprilasr/LibraryTest.java): five JUnit
smoke tests against a developer-local model and a wav from the network.  Neither exists offline, so the same calls run
on the synthetic model and, where the Java test greps the FINAL text for "ELEPHANT"/"COOL", the transcript is compared
with the CPU oracle's instead.  Also `./main ? model` (example.cpp:151-156: 3200 zeros + flush, "memory leak testing")."""
import numpy as np
import pytest

from conftest import speech_like_pcm
from test_gpu_parity import assert_same_transcript, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


def test_canLoadModel(tiny_model):                                  # LibraryTest.java:23
    import april_asr_amd as A
    m = A.Model(tiny_model["path"])
    assert m.get_sample_rate() == 16000 and m.get_name()
    m.close()


def test_cantLoadFakeModel(tmp_path):                               # LibraryTest.java:30 (NULL handle on a bad path)
    import april_asr_amd as A
    with pytest.raises(Exception):
        A.Model("/nonexistent/aprilv0_en-us.april")
    bad = tmp_path / "fake.april"
    bad.write_bytes(b"not a model")
    with pytest.raises(Exception):
        A.Model(str(bad))


def test_canUseModel(tiny_model):                                   # LibraryTest.java:87-117: 1792 zero samples, no crash
    import april_asr_amd as A
    m = A.Model(tiny_model["path"])
    got = []
    s = A.Session(m, lambda t, toks: got.append((t, toks)))
    s.feed_pcm16(np.zeros(1792, np.int16))
    s.close()
    s = A.Session(m, lambda t, toks: got.append((t, toks)))        # example.cpp:151-156
    s.feed_pcm16(np.zeros(3200, np.int16))
    s.flush()
    s.close()
    m.close()


def test_testZoo(v0_model):                                         # LibraryTest.java:35-85: whole file in ONE feed, sync, then flush
    import april_asr_amd as A
    from oracle import orc_py as O
    gm = A.Model(v0_model["path"]); om = O.Model(v0_model["path"])
    pcm = speech_like_pcm(12.0, seed=5, silence=(8.0, 11.5))        # aprilv0 dimensions: this input yields partial, final and silence results
    want, _, _ = run_oracle(om, pcm, pcm.size)
    got, _, _ = run_gpu(gm, pcm, pcm.size)
    assert_same_transcript(want, got)
    assert {t for t, _ in got} >= {1, 2, 4}                         # PARTIAL, FINAL, SILENCE (april_api.h:86-106)
    assert gm.get_sample_rate() == 16000
    gm.close(); om.close()


def test_asynchronousTest(tiny_model):                              # LibraryTest.java:119-178: 3600-sample chunks, async session
    import april_asr_amd as A
    from oracle import orc_py as O
    gm = A.Model(tiny_model["path"]); om = O.Model(tiny_model["path"])
    pcm = np.concatenate([speech_like_pcm(4.0, seed=13), np.zeros(16000 * 3, np.int16)])
    want, _, _ = run_oracle(om, pcm, 3600)
    got, _, _ = run_gpu(gm, pcm, 3600, asynchronous=True)
    assert_same_transcript(want, got)
    gm.close(); om.close()
