"""Pins the fbank oracle (oracle/orc_fbank.c): bit-exact against the committed golden vectors
(generated from the reference's compiled fbank.c/pocketfft.c by tests/golden/make_golden.py) and,
where /root/reference exists, against the compiled reference itself on more chunkings."""
import os

import numpy as np
import pytest

from oracle import orc_py as O

G = os.path.join(os.path.dirname(__file__), "golden")


def wave(pcm):
    return pcm.astype(np.float32) / np.float32(32768.0)


def run(fb, pcm, seg):
    feed = []
    w = wave(pcm)
    for i in range(0, w.size, seg):
        fb.accept(w[i:i + seg])
        feed += fb.pull_all()
    p1 = []
    while fb.flush():
        p1 += fb.pull_all()
    fb.accept_zeros(3200); fb.accept_zeros(3200)
    p2 = []
    while fb.flush():
        p2 += fb.pull_all()
    return np.array(feed), np.array(p1), np.array(p2)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_golden_lcg_chunks(built):
    g = np.load(os.path.join(G, "fbank_lcg.npz"))
    pcm = O.lcg_pcm16_fast(160000, seed=int(g["seed"]))
    feed, p1, p2 = run(O.OrcFbank(), pcm[:16000], 3200)
    assert np.array_equal(bits(feed), bits(g["feed_1s"]))
    assert np.array_equal(bits(p1), bits(g["flush1_1s"])) and np.array_equal(bits(p2), bits(g["flush2_1s"]))
    feed, p1, p2 = run(O.OrcFbank(), pcm, 3200)
    # 248 chunks while feeding 10 s, 9 after the first flush drain, 28 in total
    assert len(feed) == int(g["n_feed_10s"]) == 248
    assert len(p1) == int(g["n_flush1_10s"]) == 9 and len(p1) + len(p2) == int(g["n_flush_total_10s"]) == 28
    assert feed.astype(np.float64).sum() == float(g["sum_feed_10s"])
    assert np.array_equal(bits(feed[0]), bits(g["first_chunk_10s"]))
    assert np.array_equal(bits(p2[-1]), bits(g["last_flush_chunk_10s"]))
    assert p2[-1][-1][-1] == np.float32(-15.9423847)


def test_golden_tables(built):
    g = np.load(os.path.join(G, "fbank_lcg.npz"))
    fb = O.OrcFbank()
    assert np.array_equal(bits(fb.window()), bits(g["window"]))
    assert np.array_equal(bits(fb.mel()), bits(g["mel"]))


def test_golden_single_frames(built):
    g = np.load(os.path.join(G, "fbank_frames.npz"))
    fb = O.OrcFbank()
    got = np.stack([fb.frame(wave(p)) for p in g["pcm"]])
    assert np.array_equal(bits(got), bits(g["logmel"]))


@pytest.mark.parametrize("seg", [1600, 512, 333, 7, 160000])
def test_chunking_invariance(built, seg):
    """G3: the same PCM fed in any call sizes gives identical chunks (fbank.c leftover logic == FIFO)."""
    pcm = O.lcg_pcm16_fast(24000, seed=777)
    a = run(O.OrcFbank(), pcm, 3200)
    b = run(O.OrcFbank(), pcm, seg)
    for x, y in zip(a, b):
        assert np.array_equal(bits(x), bits(y))


@pytest.mark.skipif(not O.ref_available(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("seg,seed", [(3200, 1), (1600, 2), (512, 3), (100, 4), (48000, 5)])
def test_against_compiled_reference(built, seg, seed):
    rng = np.random.RandomState(seed)
    pcm = np.concatenate([O.lcg_pcm16_fast(20000, seed=seed), np.zeros(3000, np.int16),
                          rng.randint(-2000, 2000, size=9000).astype(np.int16)])
    a = run(O.OrcFbank(), pcm, seg)
    b = run(O.RefFbank(), pcm, seg)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(bits(x), bits(y))


@pytest.mark.skipif(not O.ref_available(), reason="compiled reference (oracle/_ref) not present")
def test_other_geometry_against_reference(built):
    """8 kHz (256-point FFT: factors 4,4,4,4) and 40 mel bins."""
    kw = dict(rate=8000, nbins=40, seg_count=9, seg_step=4)
    pcm = O.lcg_pcm16_fast(16000, seed=9)
    a = run(O.OrcFbank(**kw), pcm, 800)
    b = run(O.RefFbank(**kw), pcm, 800)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(bits(x), bits(y))


# ------------------------------------------------------------------ round_pow2 = 0: the FFT length is the frame length
# (src/fbank.c:135-138; pocketfft's radix 3 and 5 passes, radix 4 / 2 with an odd inner stride)
NONPOW2 = [("n400", dict(round_pow2=0), 400), ("n320", dict(round_pow2=0, len_ms=20), 320), ("n480", dict(round_pow2=0, len_ms=30), 480),
           ("n200", dict(round_pow2=0, rate=8000), 200),
           # lengths with other factors (pocketfft's generic pass) and not multiples of 4 (the other two twiddle constructions):
           # 882 = 2 3 3 7 7 (44.1 kHz / 20 ms), 1102 = 2 19 29 (44.1 kHz / 25 ms), 441 = 3 3 7 7 (odd), 220 = 4 5 11
           ("n882", dict(round_pow2=0, rate=44100, len_ms=20), 882), ("n1102", dict(round_pow2=0, rate=44100), 1102),
           ("n441", dict(round_pow2=0, rate=44100, len_ms=10), 441), ("n220", dict(round_pow2=0, rate=22050, len_ms=10), 220)]


@pytest.mark.parametrize("name,kw,n", NONPOW2)
def test_golden_nonpow2_single_frames(built, name, kw, n):
    g = np.load(os.path.join(G, "fbank_nonpow2.npz"))
    pcm = g[name + "_pcm"]
    assert pcm.shape[1] == n
    fb = O.OrcFbank(**kw)
    got = np.stack([fb.frame(wave(p)) for p in pcm])
    assert np.array_equal(bits(got), bits(g[name + "_logmel"]))


def test_golden_nonpow2_online_400(built):
    g = np.load(os.path.join(G, "fbank_nonpow2.npz"))
    pcm = O.lcg_pcm16_fast(16000, seed=int(g["seed"]))
    for seg in (3200, 333):
        feed, p1, p2 = run(O.OrcFbank(round_pow2=0), pcm, seg)
        assert np.array_equal(bits(feed), bits(g["n400_feed_1s"]))
        assert np.array_equal(bits(p1), bits(g["n400_flush1_1s"])) and np.array_equal(bits(p2), bits(g["n400_flush2_1s"]))


@pytest.mark.skipif(not O.ref_available(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("kw", [dict(round_pow2=0), dict(round_pow2=0, len_ms=20), dict(round_pow2=0, len_ms=30), dict(round_pow2=0, len_ms=10),
                                dict(round_pow2=0, len_ms=15), dict(round_pow2=0, rate=8000, nbins=40), dict(round_pow2=0, rate=48000, nbins=80),
                                dict(round_pow2=0, rate=8000, len_ms=30, nbins=23),
                                dict(round_pow2=0, rate=44100, len_ms=20), dict(round_pow2=0, rate=44100), dict(round_pow2=0, rate=44100, len_ms=10),
                                dict(round_pow2=0, rate=22050, len_ms=10), dict(round_pow2=0, rate=22050), dict(round_pow2=0, rate=11025, nbins=40),
                                dict(round_pow2=0, rate=32000, len_ms=11), dict(round_pow2=0, len_ms=13)])
def test_nonpow2_against_compiled_reference(built, kw):
    """FFT lengths 400, 320, 480, 160, 240, 200, 1200, 240 (8 kHz): every factor list of 4 / 2 / 3 / 5 that frame lengths produce; then
    882 = 2 3 3 7 7, 1102 = 2 19 29, 441 = 3 3 7 7 (odd), 220 = 4 5 11, 551 = 19 29 (odd), 275 = 5 5 11 (odd), 352 = 2 4 4 11, 208 = 4 4 13:
    pocketfft's generic pass and its three twiddle constructions (n mod 4 = 0, 2, odd)."""
    pcm = np.concatenate([O.lcg_pcm16_fast(20000, seed=5), np.zeros(1500, np.int16)])
    a = run(O.OrcFbank(**kw), pcm, 1600)
    b = run(O.RefFbank(**kw), pcm, 1600)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(bits(x), bits(y))


def test_unsupported_fft_lengths_are_refused(built):
    """lengths pocketfft hands to Bluestein's algorithm (a large prime factor: make_rfft_plan, pocketfft.c:2155-2182) are refused, not
    approximated -- the compiled reference still runs them; lengths with small odd factors are accepted (generic radix pass)"""
    L = O.lib()

    def new(**kw):
        o = dict(O.APRILV0_FBANK); o.update(kw); o["round_pow2"] = 0
        return L.orc_fbank_new(o["rate"], o["shift_ms"], o["len_ms"], o["nbins"], 0, o["mel_lo"], o["mel_hi"], o["seg_count"], o["seg_step"])
    for kw in (dict(rate=40360, len_ms=25), dict(rate=80720, len_ms=25)):      # 1009 (prime), 2018 = 2 x 1009
        assert not new(**kw)
        if O.ref_available():
            O.RefFbank(round_pow2=0, **kw)               # (the reference has a plan for them)
    for kw in (dict(rate=22400, len_ms=25), dict(rate=16000, len_ms=13), dict(rate=44000, len_ms=21), dict(rate=20560, len_ms=25)):      # 560 = 2^4 5 7, 208 = 2^4 13, 924 = 2^2 3 7 11, 514 = 2 x 257 (pocketfft's cost model keeps the radix plan)
        h = new(**kw)
        assert h
        L.orc_fbank_free(h)


@pytest.mark.skipif(not O.ref_available(), reason="compiled reference (oracle/_ref) not present")
def test_every_frame_length_against_compiled_reference(built):
    """round_pow2 = 0 with every frame length 8 .. 1299 (sample rate 40 n, 25 ms): wherever the oracle takes the length (whatever
    pocketfft runs through its radix passes: 1034 lengths, every factor list and all three twiddle constructions) its frames equal the
    compiled reference's bit for bit; the rest (258 lengths: primes from 191 and their small multiples) are the ones its restatement of
    make_rfft_plan's choice hands to Bluestein -- refused, never approximated."""
    rng = np.random.RandomState(3)
    same, refused = 0, []
    for n in range(8, 1300):
        kw = dict(round_pow2=0, rate=40 * n, nbins=23)
        w = wave(rng.randint(-20000, 20000, size=n + 8 * (40 * n // 100)).astype(np.int16))
        try:
            a = O.OrcFbank(**kw)
        except Exception:
            refused.append(n)
            continue
        b = O.RefFbank(**kw)
        a.accept(w); b.accept(w)
        x, y = np.array(a.pull_all()), np.array(b.pull_all())
        assert x.shape == y.shape and x.size and np.array_equal(bits(x), bits(y)), n
        same += 1
    assert same > 1000 and len(refused) < 300 and min(refused) > 150
    # a refused length has a prime factor above its square root (pocketfft.c:2162) -- necessary, not sufficient (the cost comparison decides)
    for n in refused:
        p, m = 1, n
        f = 2
        while f * f <= m:
            while m % f == 0:
                p, m = f, m // f
            f += 1
        p = m if m > 1 else p
        assert p * p > n, n
