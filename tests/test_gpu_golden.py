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


# ------------------------------------------------------------------ round_pow2 = 0: the FFT length is the frame length
# (src/fbank.c:135-138).  400 = 4 4 5 5, 320 = 4 4 4 5, 480 = 2 4 4 3 5, 200 = 2 4 5 5 (8 kHz): pocketfft's radix 3 / 5 passes and
# radix 4 / 2 with an odd inner stride, and (round 6) the generic pass for any other factor with all three twiddle constructions, on the
# device, against vectors generated from the reference's own fbank.c + pocketfft.c.
NONPOW2 = [("n400", dict(round_pow2=0), 400), ("n320", dict(round_pow2=0, length_ms=20), 320), ("n480", dict(round_pow2=0, length_ms=30), 480),
           ("n200", dict(round_pow2=0, rate=8000), 200),
           # lengths with other factors (pocketfft's generic pass) and not multiples of 4 (the other two twiddle constructions):
           # 882 = 2 3 3 7 7 (44.1 kHz / 20 ms), 1102 = 2 19 29 (44.1 kHz / 25 ms), 441 = 3 3 7 7 (odd), 220 = 4 5 11
           ("n882", dict(round_pow2=0, rate=44100, length_ms=20), 882), ("n1102", dict(round_pow2=0, rate=44100), 1102),
           ("n441", dict(round_pow2=0, rate=44100, length_ms=10), 441), ("n220", dict(round_pow2=0, rate=22050, length_ms=10), 220)]


@pytest.fixture(scope="module")
def nonpow2_models(model_dir, built):
    import april_asr_amd as A
    from april_asr_amd import synth_model as SM
    ms = {}
    for name, params, n in NONPOW2:
        p = str(model_dir / ("tiny_%s.april" % name))
        SM.write_model(p, SM.TINY_DIMS, params=params)
        ms[name] = A.Model(p)
        assert ms[name].dims.fft_size == n
    yield ms
    for m in ms.values():
        m.close()


@pytest.mark.parametrize("name", [g[0] for g in NONPOW2])
def test_fbank_kernel_on_golden_nonpow2_frames(nonpow2_models, name):
    g = np.load(os.path.join(G, "fbank_nonpow2.npz"))
    got = nonpow2_models[name].run_fbank(g[name + "_pcm"])
    want = g[name + "_logmel"]
    assert np.array_equal(bits(got), bits(want)), "max |diff| = %g" % np.abs(got - want).max()


def test_online_fbank_on_golden_nonpow2_400(nonpow2_models):
    from oracle import orc_py as O
    g = np.load(os.path.join(G, "fbank_nonpow2.npz"))
    pcm = O.lcg_pcm16_fast(16000, seed=int(g["seed"]))
    want = np.concatenate([g["n400_feed_1s"], g["n400_flush1_1s"], g["n400_flush2_1s"]])
    for seg in (3200, 333, 16000):
        n_feed, chunks = session_chunks(nonpow2_models["n400"], pcm, seg)
        assert n_feed == len(g["n400_feed_1s"]) and chunks.shape == want.shape, (seg, n_feed, chunks.shape, want.shape)
        assert np.array_equal(bits(chunks), bits(want)), "segments of %d samples" % seg


def test_session_transcript_nonpow2_400(nonpow2_models, model_dir):
    """the whole path on a round_pow2 = 0 model against the oracle session: logits within 1e-3, the same callbacks"""
    from oracle import orc_py as O
    from test_gpu_parity import run_oracle, run_gpu, assert_same_transcript, speech_like_pcm
    om = O.Model(str(model_dir / "tiny_n400.april"))
    pcm = np.concatenate([speech_like_pcm(2.0, seed=6), np.zeros(16000, np.int16)])
    want, lg0, n0 = run_oracle(om, pcm, 1600)
    got, lg1, n1 = run_gpu(nonpow2_models["n400"], pcm, 1600)
    om.close()
    assert n0 == n1
    assert lg0.shape == lg1.shape and np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)


def test_session_transcript_nonpow2_882(nonpow2_models, model_dir):
    """a 44.1 kHz model with 20 ms frames (882-point FFT = 2 3 3 7 7: the generic radix pass, the n mod 4 = 2 twiddle construction):
    the whole path against the oracle session fed in 100 ms pieces of 4410 samples: the same chunk count, logits within 1e-3, the same callbacks"""
    from oracle import orc_py as O
    from test_gpu_parity import run_oracle, run_gpu, assert_same_transcript, speech_like_pcm
    om = O.Model(str(model_dir / "tiny_n882.april"))
    pcm = np.concatenate([speech_like_pcm(2.0, seed=16, rate=44100), np.zeros(44100, np.int16)])
    want, lg0, n0 = run_oracle(om, pcm, 4410)
    got, lg1, n1 = run_gpu(nonpow2_models["n882"], pcm, 4410)
    om.close()
    assert n0 == n1 and n0 > 40
    assert lg0.shape == lg1.shape and np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)


MANY_LENGTHS = [14, 22, 26, 49, 63, 77, 98, 105, 121, 143, 169, 187, 198, 209, 221, 242, 245, 289, 294, 343, 361, 363, 429, 455, 462, 507, 529,
                539, 550, 625, 637, 663, 729, 741, 847, 875, 961, 1001, 1023, 1029, 1083, 1183, 1250, 1287]


def test_fbank_kernel_on_many_frame_lengths(model_dir, built):
    """44 frame lengths with every kind of factor list pocketfft's radix plan produces (7, 11, 13, 17, 19, 23, 29, 31 and their squares and
    mixes; lengths with n mod 4 = 0, 2 and odd; 49, 343, 2401-style prime powers; 961 = 31 x 31): the device's filterbank rows equal the oracle's bit
    for bit (the oracle equals the compiled reference at every length 8 .. 1299, tests/test_oracle_fbank.py)."""
    import april_asr_amd as A
    from april_asr_amd import synth_model as SM
    from oracle import orc_py as O
    rng = np.random.RandomState(31)
    for n in MANY_LENGTHS:
        p = str(model_dir / ("tiny_len%d.april" % n))
        SM.write_model(p, SM.TINY_DIMS, params=dict(round_pow2=0, rate=40 * n))
        m = A.Model(p)
        assert m.dims.fft_size == n
        frames = np.concatenate([rng.randint(-32768, 32768, size=(4, n)), rng.randint(-300, 300, size=(1, n)), np.full((1, n), 32767)]).astype(np.int16)
        got = m.run_fbank(frames)
        fb = O.OrcFbank(round_pow2=0, rate=40 * n)
        want = np.stack([fb.frame(f.astype(np.float32) / np.float32(32768.0)) for f in frames])
        assert np.array_equal(bits(got), bits(want)), "frame length %d: max |diff| = %g" % (n, np.abs(got - want).max())
        m.close()


def test_fbank_kernel_on_frames_above_4096_samples(model_dir, built):
    """frames of 4400, 6000 and 8192 samples (100 ms at 44 / 60 / 81.92 kHz): the two fp64 LDS buffers of a frame pass 64 KB, which a launch has to
    announce (csrc/kernels_fbank.hip launch_fbank) -- the sequential DC sum of frames above 512 samples, the generic radix pass (4400 = 2 4 5 5 11) and
    the power of two 8192 against the oracle, bit for bit"""
    import april_asr_amd as A
    from april_asr_amd import synth_model as SM
    from oracle import orc_py as O
    for n in (4400, 6000, 8192):
        p = str(model_dir / ("tiny_long%d.april" % n))
        SM.write_model(p, SM.TINY_DIMS, params=dict(round_pow2=0, rate=10 * n, length_ms=100))
        m = A.Model(p)
        assert m.dims.fft_size == n
        frames = np.random.RandomState(n).randint(-32768, 32768, size=(3, n)).astype(np.int16)
        got = m.run_fbank(frames)
        fb = O.OrcFbank(round_pow2=0, rate=10 * n, len_ms=100)
        want = np.stack([fb.frame(f.astype(np.float32) / np.float32(32768.0)) for f in frames])
        assert np.array_equal(bits(got), bits(want)), "frame length %d: max |diff| = %g" % (n, np.abs(got - want).max())
        m.close()
