"""The HIP fbank path against the committed golden vectors DIRECTLY (tests/golden/*.npz, generated from the reference's own
fbank.c + pocketfft.c by tests/golden/make_golden.py): the kernel on the 40 single frames, and the whole online fbank of a
session -- framing, the HBM feature ring, both flush phases (src/fbank.c:174-349, src/april_session.c:547-564) -- on the
LCG-noise recipe of SURVEY.md Appendix E: every 9 x 80 chunk bit for bit, the chunk counts 248 + 9 / 28 for ten seconds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gm(tiny_model):
    import april_asr_amd as A
    m = A.Model(tiny_model["path"])      # (the fbank front end does not depend on the network's size)
    yield m
    m.close()


def test_fbank_kernel_on_golden_frames(gm):
    g = np.load(os.path.join(G, "fbank_frames.npz"))
    assert gm.dims.fft_size == g["pcm"].shape[1] and gm.dims.mel == g["logmel"].shape[1]
    got = gm.run_fbank(g["pcm"])
    assert np.array_equal(bits(got), bits(g["logmel"])), "max |diff| = %g" % np.abs(got - g["logmel"]).max()


def session_chunks(gm, pcm, seg, pipelined=False):
    """feeds pcm in `seg`-sample calls, flushes, and returns (chunks while feeding, all chunks) as the session's ring saw them"""
    import april_asr_amd as A
    from oracle import orc_py as O  # noqa: F401  (checker side only: LCG recipe)
    s = A.Session(gm, lambda *_: None)
    for i in range(0, pcm.size, seg):
        s.feed_pcm16(pcm[i:i + seg])
    n_feed = s.chunks()
    s.flush()
    n_all = s.chunks()
    rows = s.frames()
    d = gm.dims
    step = 4                                             # segment_step of every april model (params.c: seg 9, step 4)
    assert rows.shape[0] >= (n_all - 1) * step + d.seg
    chunks = np.stack([rows[j * step:j * step + d.seg] for j in range(n_all)])
    s.close()
    return n_feed, chunks


def test_online_fbank_on_golden_lcg_1s(gm):
    from oracle import orc_py as O
    g = np.load(os.path.join(G, "fbank_lcg.npz"))
    pcm = O.lcg_pcm16_fast(160000, seed=int(g["seed"]))[:16000]
    want = np.concatenate([g["feed_1s"], g["flush1_1s"], g["flush2_1s"]])
    for seg in (3200, 1600, 333, 16000):
        n_feed, chunks = session_chunks(gm, pcm, seg)
        assert n_feed == len(g["feed_1s"]) and chunks.shape == want.shape, (seg, n_feed, chunks.shape, want.shape)
        assert np.array_equal(bits(chunks), bits(want)), "segments of %d samples" % seg


def test_online_fbank_on_golden_lcg_10s(gm):
    from oracle import orc_py as O
    g = np.load(os.path.join(G, "fbank_lcg.npz"))
    pcm = O.lcg_pcm16_fast(160000, seed=int(g["seed"]))
    n_feed, chunks = session_chunks(gm, pcm, 3200)
    assert n_feed == int(g["n_feed_10s"]) == 248
    assert chunks.shape[0] == 248 + int(g["n_flush_total_10s"]) == 276
    assert chunks[:248].astype(np.float64).sum() == float(g["sum_feed_10s"])
    assert np.array_equal(bits(chunks[0]), bits(g["first_chunk_10s"]))
    assert np.array_equal(bits(chunks[-1]), bits(g["last_flush_chunk_10s"]))
    assert chunks[-1][-1][-1] == np.float32(-15.9423847)
