"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(april_asr_amd/libaprilasr.so), against the CPU oracle on the same seeded inputs.

Bars: fbank bit-exact; network outputs within 1e-4 per call (fp32, different summation order);
full-session logits within 1e-3 and the callback transcript token-for-token
(BASELINE.json north_star: "token-for-token ... logits within 1e-3 fp32").
"""
import numpy as np
import pytest

from conftest import speech_like_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_tiny(tiny_model):
    import april_asr_amd as A
    m = A.Model(tiny_model["path"])
    yield m
    m.close()


@pytest.fixture(scope="module")
def orc_tiny(tiny_model):
    from oracle import orc_py as O
    m = O.Model(tiny_model["path"])
    yield m
    m.close()


@pytest.fixture(scope="module")
def gpu_v0(v0_model):
    import april_asr_amd as A
    m = A.Model(v0_model["path"])
    yield m
    m.close()


@pytest.fixture(scope="module")
def orc_v0(v0_model):
    from oracle import orc_py as O
    m = O.Model(v0_model["path"])
    yield m
    m.close()


def test_native_library_is_loaded(gpu_tiny):
    """The product path is the in-tree HIP library; nothing here can fall back to the CPU."""
    with open("/proc/self/maps") as f:
        assert "april_asr_amd/libaprilasr.so" in f.read()
    assert gpu_tiny.dims.n_devices >= 1


# ------------------------------------------------------------------ fbank
def test_fbank_kernel_bit_exact(gpu_tiny):
    from oracle import orc_py as O
    rng = np.random.RandomState(1)
    n = gpu_tiny.dims.fft_size
    frames = [O.lcg_pcm16_fast(n * 64, seed=99).reshape(64, n),
              rng.randint(-32768, 32768, size=(64, n)).astype(np.int16),
              (rng.randint(-300, 300, size=(16, n))).astype(np.int16),
              np.zeros((2, n), np.int16),
              np.full((2, n), -32768, np.int16), np.full((2, n), 32767, np.int16)]
    pcm = np.concatenate(frames)
    got = gpu_tiny.run_fbank(pcm)
    fb = O.OrcFbank()
    want = np.stack([fb.frame(pcm[i].astype(np.float32) / np.float32(32768.0)) for i in range(pcm.shape[0])])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        "max |diff| = %g" % np.abs(got - want).max()


# ------------------------------------------------------------------ networks, one call
@pytest.mark.parametrize("which", ["tiny", "v0"])
def test_encoder_matches_oracle(which, request):
    gm = request.getfixturevalue("gpu_" + which); om = request.getfixturevalue("orc_" + which)
    d = gm.dims
    rng = np.random.RandomState(5)
    n = 3
    x = rng.uniform(-16, 8, size=(n, d.seg, d.mel)).astype(np.float32)
    h = rng.uniform(-0.5, 0.5, size=(n, d.n_layers, d.d_model)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(n, d.n_layers, d.hidden)).astype(np.float32)
    eout, h2, c2 = gm.run_encoder(x, h, c)
    for i in range(n):
        e0, h0, c0 = om.encoder(x[i:i + 1], h[i][:, None, :], c[i][:, None, :])
        assert np.abs(eout[i] - e0.ravel()).max() < 1e-4
        assert np.abs(h2[i] - h0[:, 0, :]).max() < 1e-4
        assert np.abs(c2[i] - c0[:, 0, :]).max() < 1e-4


def test_wide_conv_front_end_matches_oracle(gpu_v0, orc_v0):
    """64 rows through the encoder entry (x handed over directly): the launch is large enough for the one-workgroup-per-chunk form
    (>= 48 chunks); a few of its rows against the CPU oracle's encoder, tolerance as in tests/test_gpu_parity.py."""
    d = gpu_v0.dims
    rng = np.random.RandomState(55)
    n = 64
    x = rng.uniform(-16, 8, size=(n, d.seg, d.mel)).astype(np.float32)
    h = rng.uniform(-0.5, 0.5, size=(n, d.n_layers, d.d_model)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(n, d.n_layers, d.hidden)).astype(np.float32)
    eout, h2, c2 = gpu_v0.run_encoder(x, h, c)
    for i in (0, 31, 63):
        e0, h0, c0 = orc_v0.encoder(x[i:i + 1], h[i][:, None, :], c[i][:, None, :])
        assert np.abs(eout[i] - e0.ravel()).max() < 1e-4
        assert np.abs(h2[i] - h0[:, 0, :]).max() < 1e-4
        assert np.abs(c2[i] - c0[:, 0, :]).max() < 1e-4


@pytest.mark.parametrize("which", ["tiny", "v0"])
def test_decoder_and_joiner_match_oracle(which, request):
    gm = request.getfixturevalue("gpu_" + which); om = request.getfixturevalue("orc_" + which)
    d = gm.dims
    rng = np.random.RandomState(6)
    ctx = rng.randint(0, d.vocab, size=(5, 2)).astype(np.int64)
    dout = gm.run_decoder(ctx)
    for i in range(5):
        assert np.abs(dout[i] - om.decoder(ctx[i]).ravel()).max() < 1e-5
    e = rng.uniform(-2, 2, size=(5, d.joiner)).astype(np.float32)
    lg = gm.run_joiner(e, dout)
    for i in range(5):
        assert np.abs(lg[i] - om.joiner(e[i].reshape(1, 1, -1), dout[i].reshape(1, 1, -1)).ravel()).max() < 1e-4


def test_encoder_batch_invariant_bitwise(gpu_tiny):
    """A row's result does not depend on the batch it is in (fixed reduction order)."""
    d = gpu_tiny.dims
    rng = np.random.RandomState(7)
    n = 37
    x = rng.uniform(-16, 8, size=(n, d.seg, d.mel)).astype(np.float32)
    h = rng.uniform(-0.5, 0.5, size=(n, d.n_layers, d.d_model)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(n, d.n_layers, d.hidden)).astype(np.float32)
    e_all, h_all, c_all = gpu_tiny.run_encoder(x, h, c)
    for i in (0, 16, 36):
        e1, h1, c1 = gpu_tiny.run_encoder(x[i:i + 1], h[i:i + 1], c[i:i + 1])
        assert np.array_equal(e1[0], e_all[i]) and np.array_equal(h1[0], h_all[i]) and np.array_equal(c1[0], c_all[i])


# ------------------------------------------------------------------ full sessions
def run_oracle(om, pcm, chunk):
    from oracle import orc_py as O
    s = O.Session(om, trace_logits=12000)
    for i in range(0, pcm.size, chunk):
        s.feed(pcm[i:i + chunk])
    s.flush()
    ev = [(t, [(om.token(i).encode(), lp, fl, ms) for (i, lp, fl, ms) in toks]) for t, toks in s.events]
    lg = s.logits().copy(); n = s.chunks()
    s.close()
    return ev, lg, n


def run_gpu(gm, pcm, chunk, asynchronous=False):
    import april_asr_amd as A
    ev = []
    s = A.Session(gm, lambda t, toks: ev.append((t, toks)), asynchronous=asynchronous, no_rt=asynchronous, raw_events=True)
    s.trace_logits(12000)
    for i in range(0, pcm.size, chunk):
        s.feed_pcm16(pcm[i:i + chunk])
        if asynchronous:
            s.drain()
    s.flush()
    if asynchronous:
        s.drain()
    lg = s.traced_logits().copy(); n = s.chunks()
    s.close()
    return ev, lg, n


def assert_same_transcript(want, got, tol=1e-3):
    assert len(want) == len(got), "callback count %d vs %d" % (len(want), len(got))
    for k, ((t0, k0), (t1, k1)) in enumerate(zip(want, got)):
        assert t0 == t1, "callback %d type %d vs %d" % (k, t0, t1)
        assert len(k0) == len(k1), "callback %d token count" % k
        for a, b in zip(k0, k1):
            assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3], "callback %d token %r vs %r" % (k, a, b)
            assert abs(a[1] - b[1]) < tol


@pytest.mark.parametrize("chunk", [1600, 512, 16000 * 6])
def test_session_transcript_tiny(gpu_tiny, orc_tiny, chunk):
    pcm = np.concatenate([speech_like_pcm(3.0, seed=1), np.zeros(16000 * 3, np.int16)])
    want, lg0, n0 = run_oracle(orc_tiny, pcm, 1600)
    got, lg1, n1 = run_gpu(gpu_tiny, pcm, chunk)
    assert n0 == n1
    assert lg0.shape == lg1.shape and np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)


def test_session_transcript_tiny_async(gpu_tiny, orc_tiny):
    from oracle import orc_py as O
    pcm = O.lcg_pcm16_fast(16000 * 3, seed=4)
    want, lg0, _ = run_oracle(orc_tiny, pcm, 1600)
    got, lg1, _ = run_gpu(gpu_tiny, pcm, 1600, asynchronous=True)
    assert lg0.shape == lg1.shape and np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)


def test_session_transcript_v0(gpu_v0, orc_v0):
    """aprilv0 dimensions, 4 s of audio + flush (the oracle needs ~0.1 s per chunk)."""
    pcm = speech_like_pcm(4.0, seed=3, silence=(1.0, 1.6))
    want, lg0, n0 = run_oracle(orc_v0, pcm, 1600)
    got, lg1, n1 = run_gpu(gpu_v0, pcm, 1600)
    assert n0 == n1
    assert lg0.shape == lg1.shape and np.abs(lg0 - lg1).max() < 1e-3, np.abs(lg0 - lg1).max()
    assert_same_transcript(want, got)


@pytest.mark.parametrize("which,secs", [("tiny", 6.0), ("v0", 3.0)])
def test_layer_major_equals_streaming(which, secs, request):
    """a long feed (the whole input at once) takes the layer-major schedule -- conv front end,
    the input half of every gate GEMM, feed-forward blocks and encoder_proj once over all chunks, only the recurrent half
    per time step.  Same chains in the same order: every logit and every callback equals the 100 ms-feed run BIT FOR BIT."""
    gm = request.getfixturevalue("gpu_" + which)
    pcm = np.concatenate([speech_like_pcm(secs / 2, seed=11), np.zeros(16000, np.int16), speech_like_pcm(secs / 2, seed=12)])
    before = gm.stats().lm_chunks
    ev_s, lg_s, n_s = run_gpu(gm, pcm, 1600)
    assert gm.stats().lm_chunks - before <= 28                 # 100 ms feeds stay below the layer-major threshold (the flush tail may not)
    mid = gm.stats().lm_chunks
    ev_o, lg_o, n_o = run_gpu(gm, pcm, pcm.size)
    st = gm.stats()
    assert st.lm_chunks - mid >= n_o - 28 - 8 and st.replay_mismatch == 0    # everything but (parts of) the flush tail went layer-major
    assert n_s == n_o
    assert np.array_equal(lg_s, lg_o)
    assert ev_s == ev_o
    # (a long feed runs as a wavefront over blocks of time steps -- the same launch of all layers is ONE z-batched launch,
    # Engine::run_lm_wavefront -- traced or not; untraced the search additionally runs as one captured graph per block)
    import april_asr_amd as A
    ev_p = []
    s = A.Session(gm, lambda t, toks: ev_p.append((t, toks)), raw_events=True)
    s.feed_pcm16(pcm); s.flush()
    n_p = s.chunks()
    s.close()
    assert n_p == n_o and ev_p == ev_o
    assert gm.stats().replay_mismatch == 0


def test_layer_major_ragged_group(gpu_tiny):
    """Several sessions fed whole inputs of different lengths in one call: grouped layer-major steps (T = the shortest
    backlog of the group), the rest in further steps; each session equals itself streamed alone, bit for bit."""
    import april_asr_amd as A
    from oracle import orc_py as O
    lens = [16000 * 3, 16000 * 2 + 777, 16000 * 4, 9000, 16000 * 3]
    pcms = [O.lcg_pcm16_fast(n, seed=600 + i) for i, n in enumerate(lens)]
    evs = [[] for _ in lens]
    sess = [A.Session(gpu_tiny, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(len(lens))]
    for s in sess:
        s.trace_logits(1200)
    grp = A.SessionGroup(sess)
    before = gpu_tiny.stats().lm_chunks
    grp.feed(pcms)
    grp.flush()
    assert gpu_tiny.stats().lm_chunks > before
    for i, s in enumerate(sess):
        ev1, lg1, _ = run_gpu(gpu_tiny, pcms[i], 1600)
        assert np.array_equal(lg1, s.traced_logits()), i
        assert ev1 == evs[i], i
    for s in sess:
        s.close()


def test_many_sessions_equal_single(gpu_tiny):
    """64 sessions fed together (different inputs) == each one alone, bit for bit (logits and callbacks)."""
    import april_asr_amd as A
    from oracle import orc_py as O
    n = 64
    pcms = [O.lcg_pcm16_fast(16000 * 2, seed=100 + i) for i in range(n)]
    evs = [[] for _ in range(n)]
    sess = [A.Session(gpu_tiny, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]
    for s in sess:
        s.trace_logits(400)
    grp = A.SessionGroup(sess)
    for o in range(0, 32000, 1600):
        grp.feed([p[o:o + 1600] for p in pcms])
    grp.flush()
    st = gpu_tiny.stats()
    assert st.max_batch_seen == n
    for i in (0, 17, 63):
        ev1, lg1, _ = run_gpu(gpu_tiny, pcms[i], 1600)
        assert np.array_equal(lg1, sess[i].traced_logits())
        assert ev1 == evs[i]
    for s in sess:
        s.close()


# ------------------------------------------------------------------ more session behaviour
def test_feed_after_flush_and_double_flush(gpu_tiny, orc_tiny):
    """flush is idempotent until new audio arrives; feeding after a flush continues the same stream
    (reference src/april_session.c:510,548-550)."""
    import april_asr_amd as A
    from oracle import orc_py as O
    a = speech_like_pcm(2.0, seed=21); b = speech_like_pcm(1.5, seed=22)
    so = O.Session(orc_tiny, trace_logits=4000)
    so.feed(a); so.flush(); so.flush(); so.feed(b); so.flush()
    want = [(t, [(orc_tiny.token(i).encode(), lp, fl, ms) for (i, lp, fl, ms) in toks]) for t, toks in so.events]
    ev = []
    sg = A.Session(gpu_tiny, lambda t, toks: ev.append((t, toks)), raw_events=True)
    sg.trace_logits(4000)
    sg.feed_pcm16(a); sg.flush(); sg.flush(); sg.feed_pcm16(b.tobytes()); sg.flush()
    assert sg.chunks() == so.chunks()
    assert np.abs(sg.traced_logits() - so.logits()).max() < 1e-3
    assert_same_transcript(want, ev)
    sg.close(); so.close()


def test_async_overflow_reports_cant_keep_up(gpu_tiny):
    """More than 48000 queued samples in an asynchronous session: the push is dropped and CANT_KEEP_UP is delivered
    on the calling thread (reference src/april_session.c:482-492, src/audio_provider.c:31)."""
    import threading
    import april_asr_amd as A
    seen = []
    s = A.Session(gpu_tiny, lambda t, toks: seen.append((int(t), threading.get_ident())), asynchronous=True, no_rt=True, raw_events=True)
    big = np.zeros(48001, np.int16)
    s.feed_pcm16(big)
    assert (3, threading.get_ident()) in seen
    s.feed_pcm16(np.zeros(1600, np.int16))
    s.flush(); s.drain()
    assert s.get_rt_speedup() == 1.0
    s.close()


def test_async_overflow_boundary(gpu_tiny):
    """The reference's ring refuses a push that would make it hold MAX_AUDIO = 48000 samples or more
    (src/audio_provider.c:31,61: `(available + count) >= MAX_AUDIO`): 47999 samples into an empty ring are accepted,
    48000 are not."""
    import april_asr_amd as A
    for n, rejected in ((47999, False), (48000, True)):
        seen = []
        s = A.Session(gpu_tiny, lambda t, toks: seen.append(int(t)), asynchronous=True, no_rt=True, raw_events=True)
        s.feed_pcm16(np.zeros(n, np.int16))
        assert (3 in seen) == rejected, (n, seen)
        s.drain()
        frames = (n - 512) // 160 + 1                       # fbank.c:195-236: 512-sample frames every 160 samples
        assert s.chunks() == (0 if rejected else (frames - 9) // 4 + 1)
        s.close()


def test_rt_speedup_is_reported(gpu_tiny):
    """ASYNC_RT sessions report the reference's measure (EMA of processing time x 1.1 / audio time per chunk,
    src/april_session.c:95-97,456-462): positive, and far below 1 on a GPU that keeps up; other sessions report 1.0."""
    import april_asr_amd as A
    from oracle import orc_py as O
    s = A.Session(gpu_tiny, lambda t, toks: None, asynchronous=True, no_rt=False, raw_events=True)
    pcm = O.lcg_pcm16_fast(16000 * 2, seed=77)
    for o in range(0, pcm.size, 1600):
        s.feed_pcm16(pcm[o:o + 1600]); s.drain()
    v = s.get_rt_speedup()
    assert 0.0 < v < 1.0, v
    s.close()


def test_device_and_host_contexts_agree(gpu_tiny):
    """The device keeps the token context that steers the decoder; the host derives its own from the same records when it
    builds the callbacks.  After audio that emits tokens and crosses the 2200 ms silence reset they are identical, and the
    replay never disagreed with a device decision."""
    import april_asr_amd as A
    pcm = np.concatenate([speech_like_pcm(3.0, seed=1), np.zeros(16000 * 3, np.int16), speech_like_pcm(1.0, seed=2)])
    ev = []
    s = A.Session(gpu_tiny, lambda t, toks: ev.append((t, toks)), raw_events=True)
    for o in range(0, pcm.size, 1600):
        s.feed_pcm16(pcm[o:o + 1600])
        h, d = s.contexts()
        if not (h[0] == d[0] and h[1] == d[1]):
            s.close()
            raise AssertionError((o, h, d))
    s.flush()
    h, d = s.contexts()
    try:
        # flushed: nothing active; the context is reset unless it already started with blank (april_session.c:297 tests
        # element 0 only), on both sides alike
        assert h[0] == d[0] and h[1] == d[1] and d[0] == 0 and d[2] == -1, (h, d)
        assert any(t == 1 for t, _ in ev)
        assert gpu_tiny.stats().replay_mismatch == 0
    finally:
        s.close()


def test_async_handler_runs_on_library_thread(gpu_tiny):
    import threading
    import april_asr_amd as A
    from oracle import orc_py as O
    tids = set()
    s = A.Session(gpu_tiny, lambda t, toks: tids.add(threading.get_ident()), asynchronous=True, no_rt=True, raw_events=True)
    s.feed_pcm16(O.lcg_pcm16_fast(16000 * 2, seed=31)); s.flush(); s.drain()
    assert tids and threading.get_ident() not in tids
    s.close()


def test_concurrent_sync_callers(gpu_tiny):
    """Different sessions of one model fed from different threads (the reference allows this, concepts.md:41-45):
    every session's callbacks equal the single-threaded run."""
    import threading
    import april_asr_amd as A
    from oracle import orc_py as O
    n = 8
    pcms = [O.lcg_pcm16_fast(16000 * 2, seed=300 + i) for i in range(n)]
    want = [run_gpu(gpu_tiny, p, 1600)[0] for p in pcms]
    got = [[] for _ in range(n)]

    def worker(k):
        s = A.Session(gpu_tiny, lambda t, toks: got[k].append((t, toks)), raw_events=True)
        for o in range(0, pcms[k].size, 1600):
            s.feed_pcm16(pcms[k][o:o + 1600])
        s.flush(); s.close()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    assert got == want


def test_model_from_device_blob(gpu_tiny, tiny_model):
    """The multi-GPU load path on one GPU: export the packed blob, move it to the device the way the RCCL broadcast
    delivers it, build a second model from the device pointer, same results bit for bit."""
    import torch
    import april_asr_amd as A
    from oracle import orc_py as O
    blob = torch.from_numpy(gpu_tiny.export_blob()).cuda()
    torch.cuda.synchronize()
    m2 = A.Model.from_blob(None, device_ptr=blob.data_ptr(), size=blob.numel())
    assert m2.get_name() == gpu_tiny.get_name() and m2.dims.param_count == gpu_tiny.dims.param_count
    pcm = O.lcg_pcm16_fast(16000 * 2, seed=41)
    e1, l1, _ = run_gpu(gpu_tiny, pcm, 1600)
    e2, l2, _ = run_gpu(m2, pcm, 1600)
    assert e1 == e2 and np.array_equal(l1, l2)
    m2.close()


def test_model_broadcast_in_library(gpu_tiny, tiny_model):
    """The weight broadcast is a library call (RCCL): with one rank it degenerates to a communicator of one, which still
    exercises the linkage, the metadata / weight broadcasts and the bookkeeping; rank 0 gets its own model back."""
    import april_asr_amd as A
    with open("/proc/self/maps") as f:
        assert "librccl" in f.read()
    ident = A.Model.broadcast_id()
    assert len(ident) == 128
    m = A.Model.broadcast(gpu_tiny, 0, 1, ident)
    assert m is gpu_tiny
    li = m.load_info()
    assert li.used_rccl == 1 and li.ranks == 1 and li.broadcast_bytes > 0 and li.broadcast_ms >= 0.0


@pytest.mark.parametrize("which", ["tiny", "v0"])
def test_feed_wavefront_equals_chunk_by_chunk(which, request):
    """The 2..3 chunk steps of a 100 ms feed run as ONE wavefront over the layers (z-batched launches, Engine::lm_step mode 1).
    A model loaded with APRIL_WAVE_MIN_CHUNKS=0 steps the same feeds chunk by chunk (round-1 form): every logit and every
    callback of the two are equal BIT FOR BIT, alone and in a batch of different sessions."""
    import os
    import april_asr_amd as A
    gm = request.getfixturevalue("gpu_" + which)
    path = request.getfixturevalue(("tiny" if which == "tiny" else "v0") + "_model")["path"]
    old = os.environ.get("APRIL_WAVE_MIN_CHUNKS")
    os.environ["APRIL_WAVE_MIN_CHUNKS"] = "0"
    try:
        plain = A.Model(path)
    finally:
        if old is None:
            del os.environ["APRIL_WAVE_MIN_CHUNKS"]
        else:
            os.environ["APRIL_WAVE_MIN_CHUNKS"] = old
    try:
        pcm = speech_like_pcm(3.0, seed=21, silence=(1.2, 1.7))
        w0 = gm.stats().wave_chunks
        ev_w, lg_w, n_w = run_gpu(gm, pcm, 1600)
        assert gm.stats().wave_chunks - w0 >= n_w // 2                  # most chunks went through the wavefront
        ev_p, lg_p, n_p = run_gpu(plain, pcm, 1600)
        assert plain.stats().wave_chunks == 0
        assert n_w == n_p and np.array_equal(lg_w, lg_p) and ev_w == ev_p
        # a batch of different sessions (different feed sizes, so different chunk counts per feed)
        n = 5
        pcms = [speech_like_pcm(2.0, seed=300 + i) for i in range(n)]
        outs = []
        for m in (gm, plain):
            evs = [[] for _ in range(n)]
            sess = [A.Session(m, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]
            for s_ in sess:
                s_.trace_logits(4000)
            grp = A.SessionGroup(sess)
            step = [1600, 1600, 3200, 800, 2400]
            pos = [0] * n
            while any(pos[i] < pcms[i].size for i in range(n)):
                grp.feed([pcms[i][pos[i]:pos[i] + step[i]] for i in range(n)])
                pos = [pos[i] + step[i] for i in range(n)]
            grp.flush()
            outs.append(([s_.traced_logits().copy() for s_ in sess], evs))
            for s_ in sess:
                s_.close()
        for i in range(n):
            assert np.array_equal(outs[0][0][i], outs[1][0][i]), i
            assert outs[0][1][i] == outs[1][1][i], i
        assert gm.stats().replay_mismatch == 0 and plain.stats().replay_mismatch == 0
    finally:
        plain.close()


def test_many_batch_shapes(gpu_tiny):
    """Sessions join one per feed, so every feed has a batch shape (sessions x chunks) never seen before: the launch chains of
    one-off shapes go out eagerly (graphs are captured at the second use), more than 64 feed-wavefront plans are built (the
    plan cache is emptied once on the way), and every session still equals itself streamed alone, bit for bit."""
    import april_asr_amd as A
    from oracle import orc_py as O
    n = 72
    pcms = [O.lcg_pcm16_fast(1600 * (n + 6), seed=900 + i) for i in range(n)]
    evs = [[] for _ in range(n)]
    sess, pos = [], []
    for step in range(n + 6):
        if step < n:
            sess.append(A.Session(gpu_tiny, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(step), raw_events=True))
            pos.append(0)
            if step in (0, 35, 71):
                sess[-1].trace_logits(1200)
        grp = A.SessionGroup(sess)
        grp.feed([pcms[i][pos[i]:pos[i] + 1600] for i in range(len(sess))])
        pos = [p + 1600 for p in pos]
    A.SessionGroup(sess).flush()
    assert gpu_tiny.stats().replay_mismatch == 0
    for i in (0, 35, 71):
        fed = pcms[i][:pos[i]]
        ev1, lg1, _ = run_gpu(gpu_tiny, fed, 1600)
        assert np.array_equal(lg1, sess[i].traced_logits()), i
        assert ev1 == evs[i], i
    for s_ in sess:
        s_.close()


def test_decoder_table_equals_decoder_network(tiny_model, gpu_tiny):
    """The decoder output of EVERY 2-token context is computed once at load (Engine::build_dec_table) and the joiner reads the
    row of a session's context.  A model loaded with the table disabled runs the decoder network per context change as in
    round 1: decoder outputs, every joiner logit and every callback of the two are equal BIT FOR BIT."""
    import os
    import april_asr_amd as A
    from oracle import orc_py as O
    old = os.environ.get("APRIL_DEC_TABLE_MB")
    os.environ["APRIL_DEC_TABLE_MB"] = "0"
    try:
        direct = A.Model(tiny_model["path"])
    finally:
        if old is None:
            del os.environ["APRIL_DEC_TABLE_MB"]
        else:
            os.environ["APRIL_DEC_TABLE_MB"] = old
    try:
        V = int(gpu_tiny.dims.vocab)
        rng = np.random.RandomState(5)
        ctx = rng.randint(0, V, size=(300, 2)).astype(np.int64)
        ctx[:4] = [[0, 0], [V - 1, V - 1], [0, V - 1], [V - 1, 0]]
        assert np.array_equal(gpu_tiny.run_decoder(ctx), direct.run_decoder(ctx))
        pcm = O.lcg_pcm16_fast(16000 * 4, seed=77)
        ev_t, lg_t, n_t = run_gpu(gpu_tiny, pcm, 1600)
        ev_d, lg_d, n_d = run_gpu(direct, pcm, 1600)
        assert n_t == n_d and np.array_equal(lg_t, lg_d) and ev_t == ev_d and len(ev_t) > 0
        ev_t2, lg_t2, _ = run_gpu(gpu_tiny, pcm, pcm.size)          # long feed (wavefront) on both
        ev_d2, lg_d2, _ = run_gpu(direct, pcm, pcm.size)
        assert np.array_equal(lg_t2, lg_d2) and ev_t2 == ev_d2 and np.array_equal(lg_t2, lg_t)
        assert direct.stats().replay_mismatch == 0
    finally:
        direct.close()


def test_two_engines_on_one_device(tiny_model):
    """APRIL_GPU_DEVICES=0,0: two engines (stream + stepping thread + slot arrays each) on one GPU, the second one's
    weights are a device-to-device copy of the first (the multi-device load path without a second GPU); sessions are
    spread over both and every transcript equals the single-engine run."""
    import os
    import pickle
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import april_asr_amd as A\n"
        "from april_asr_amd import synth_model as SM\n"
        "m = A.Model(%r)\n"
        "n = 6; pcms = [SM.lcg_pcm16(16000, seed=800 + i) for i in range(n)]\n"
        "evs = [[] for _ in range(n)]\n"
        "ss = [A.Session(m, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]\n"
        "g = A.SessionGroup(ss)\n"
        "for o in range(0, 16000, 1600): g.feed([p[o:o + 1600] for p in pcms])\n"
        "g.flush()\n"
        "import pickle; pickle.dump((evs, int(m.dims.n_devices), [int(m.stats(i).chunks) for i in range(int(m.dims.n_devices))]), open(sys.argv[1], 'wb'))\n"
        "for s in ss: s.close()\n"
        "m.close()\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), tiny_model["path"]))
    outs = []
    for devs in ("0", "0,0"):
        out = os.path.join(os.path.dirname(tiny_model["path"]), "lanes_%d.pkl" % len(devs))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, APRIL_GPU_DEVICES=devs))
        outs.append(pickle.load(open(out, "rb")))
    assert outs[0][1] == 1 and outs[1][1] == 2
    assert all(c > 0 for c in outs[1][2])                      # both engines stepped sessions
    assert outs[0][0] == outs[1][0] and any(len(e) for e in outs[0][0])


def test_sessions_above_max_batch(tiny_model):
    """More ready sessions than APRIL_MAX_BATCH: the step is split into sub-batches, results unchanged."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import april_asr_amd as A\n"
        "from april_asr_amd import synth_model as SM\n"
        "m = A.Model(%r)\n"
        "n = 40; pcms = [SM.lcg_pcm16(16000, seed=500 + i) for i in range(n)]\n"
        "evs = [[] for _ in range(n)]\n"
        "ss = [A.Session(m, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]\n"
        "g = A.SessionGroup(ss)\n"
        "for o in range(0, 16000, 1600): g.feed([p[o:o + 1600] for p in pcms])\n"
        "g.flush()\n"
        "import pickle; pickle.dump(evs, open(sys.argv[1], 'wb'))\n"
        "for s in ss: s.close()\n"
        "m.close()\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), tiny_model["path"]))
    outs = []
    for mb in ("16", "2048"):
        out = os.path.join(os.path.dirname(tiny_model["path"]), "evs_%s.pkl" % mb)
        env = dict(os.environ, APRIL_MAX_BATCH=mb)
        subprocess.check_call([sys.executable, "-c", code, out], env=env)
        import pickle
        outs.append(pickle.load(open(out, "rb")))
    assert outs[0] == outs[1] and any(len(e) for e in outs[0])


def test_config3_256_sessions_aprilv0(gpu_v0, orc_v0):
    """BASELINE configs[2]: 256 concurrent streaming sessions at aprilv0 dimensions in 100 ms feeds on one GPU (the workload
    bench.py times).  The batch runs untraced, i.e. on the captured launch chains with the full-K GEMM schedule; session 0
    against the CPU oracle (token-exact, log-probabilities within 1e-3, and every logit of the same session stepped alone
    within 1e-3), sessions 1, 128 and 255 against themselves stepped alone: identical callbacks including the
    log-probability bits (batch invariance across the two GEMM schedules)."""
    import april_asr_amd as A
    from oracle import orc_py as O
    n, secs = 256, 4.0
    pcms = [speech_like_pcm(secs, seed=50, silence=(1.2, 1.7))] + [O.lcg_pcm16_fast(int(16000 * secs), seed=900 + i) for i in range(1, n)]
    watch = (0, 1, 128, 255)
    evs = {i: [] for i in watch}
    counts = np.zeros(6, np.uint64)
    sess = [A.Session(gpu_v0, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) if i in evs
            else A.Session(gpu_v0, None, counters=counts) for i in range(n)]
    grp = A.SessionGroup(sess)
    for o in range(0, int(16000 * secs), 1600):
        grp.feed([p[o:o + 1600] for p in pcms])
    grp.flush()
    st = gpu_v0.stats()
    assert st.max_batch_seen == n and st.replay_mismatch == 0
    want, lg0, n0 = run_oracle(orc_v0, pcms[0], 1600)
    assert sess[0].chunks() == n0
    assert_same_transcript(want, evs[0])
    for i in watch:
        ev1, lg1, _ = run_gpu(gpu_v0, pcms[i], 1600)
        assert ev1 == evs[i], i
        if i == 0:
            assert lg1.shape == lg0.shape and np.abs(lg1 - lg0).max() < 1e-3, np.abs(lg1 - lg0).max()
    for s in sess:
        s.close()


def test_session_60s_aprilv0(gpu_v0, orc_v0):
    """BASELINE configs[1]: aprilv0 dimensions, one session, 60 s of synthetic 16 kHz PCM16 in 100 ms feeds + flush,
    token-exact against the CPU oracle, every logit within 1e-3."""
    pcm = np.concatenate([speech_like_pcm(20.0, seed=5, silence=(8.0, 11.5)), speech_like_pcm(40.0, seed=6, silence=(30.0, 33.0))])
    want, lg0, n0 = run_oracle(orc_v0, pcm, 1600)
    got, lg1, n1 = run_gpu(gpu_v0, pcm, 1600)
    assert n0 == n1 == 1498 + 28
    assert lg0.shape == lg1.shape
    assert np.abs(lg0 - lg1).max() < 1e-3, np.abs(lg0 - lg1).max()
    assert_same_transcript(want, got)
    assert {t for t, _ in got} >= {1, 2, 4}
    # the same minute handed over in ONE aas_feed_pcm16 call + flush (the path bench.py's offline_single_session_60s times:
    # one pass of the layer-major wavefront at aprilv0 dims, block length 29, 8192-frame ring): every logit and every callback
    # against the oracle, and bit for bit against the 100 ms feeds
    got1, lg2, n2 = run_gpu(gpu_v0, pcm, pcm.size)
    assert n2 == n0 and lg2.shape == lg0.shape
    assert np.abs(lg0 - lg2).max() < 1e-3, np.abs(lg0 - lg2).max()
    assert_same_transcript(want, got1)
    assert np.array_equal(lg1.view(np.uint32), lg2.view(np.uint32)), "one 60 s feed differs from 100 ms feeds"
    assert got1 == got
    # ... and untraced (captured graphs, the wavefront's own search graphs): the callbacks again
    import april_asr_amd as A
    ev = []
    s = A.Session(gpu_v0, lambda t, toks: ev.append((t, toks)), raw_events=True)
    lm0 = gpu_v0.stats().lm_chunks
    s.feed_pcm16(pcm); s.flush()
    assert s.chunks() == n0 and ev == got
    assert gpu_v0.stats().lm_chunks - lm0 >= 1400, "the one-call minute did not take the layer-major path"
    s.close()


def test_odd_dimensions_model(medium_model):
    """3 layers, d=192, hidden=320, ffn=448, vocab=131 (padded to 160 on the device), 24 conv-2 channels
    (im2col K padded 216 -> 256): network calls and a full session against the oracle."""
    import april_asr_amd as A
    from oracle import orc_py as O
    gm = A.Model(medium_model["path"]); om = O.Model(medium_model["path"])
    d = gm.dims
    assert (d.n_layers, d.d_model, d.hidden, d.ffn, d.vocab) == (3, 192, 320, 448, 131)
    rng = np.random.RandomState(9)
    x = rng.uniform(-16, 8, size=(2, d.seg, d.mel)).astype(np.float32)
    h = rng.uniform(-0.5, 0.5, size=(2, d.n_layers, d.d_model)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(2, d.n_layers, d.hidden)).astype(np.float32)
    eout, h2, c2 = gm.run_encoder(x, h, c)
    for i in range(2):
        e0, h0, c0 = om.encoder(x[i:i + 1], h[i][:, None, :], c[i][:, None, :])
        assert np.abs(eout[i] - e0.ravel()).max() < 1e-4 and np.abs(h2[i] - h0[:, 0, :]).max() < 1e-4 and np.abs(c2[i] - c0[:, 0, :]).max() < 1e-4
    pcm = speech_like_pcm(3.0, seed=8, silence=(1.0, 1.4))
    want, lg0, n0 = run_oracle(om, pcm, 1600)
    got, lg1, n1 = run_gpu(gm, pcm, 1600)
    assert n0 == n1 and lg0.shape == lg1.shape and lg1.shape[1] == 131
    assert np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)
    gm.close(); om.close()


def test_widths_padded_at_load_match_the_oracle(narrow_model):
    """d 144, cell 208, ffn 304, joiner 80, 48 conv-3 channels: multiples of 16, not of 64.  The loader rounds every width up to 64 with
    zero weights (csrc/model_loader.cc pad_host_model; the BasicNorm mean keeps the file's 144) -- the encoder call on the real
    columns and a full session (logits, transcript) against the oracle, which runs the file's widths; the padded state columns stay 0."""
    import april_asr_amd as A
    from oracle import orc_py as O
    gm = A.Model(narrow_model["path"]); om = O.Model(narrow_model["path"])
    d = gm.dims
    assert (d.n_layers, d.d_model, d.hidden, d.ffn, d.joiner, d.vocab, d.d_model_file) == (2, 192, 256, 320, 128, 60, 144)
    D0, H0 = 144, 208
    rng = np.random.RandomState(19)
    x = rng.uniform(-16, 8, size=(2, d.seg, d.mel)).astype(np.float32)
    h = np.zeros((2, d.n_layers, d.d_model), np.float32); c = np.zeros((2, d.n_layers, d.hidden), np.float32)
    h[:, :, :D0] = rng.uniform(-0.5, 0.5, size=(2, d.n_layers, D0)); c[:, :, :H0] = rng.uniform(-1, 1, size=(2, d.n_layers, H0))
    eout, h2, c2 = gm.run_encoder(x, h, c)
    assert not h2[:, :, D0:].any() and np.abs(c2[:, :, H0:]).max() < 1e-6 and not eout[:, 80:].any()
    for i in range(2):
        e0, h0, c0 = om.encoder(x[i:i + 1], np.ascontiguousarray(h[i][:, None, :D0]), np.ascontiguousarray(c[i][:, None, :H0]))
        assert np.abs(eout[i][:80] - e0.ravel()).max() < 1e-4
        assert np.abs(h2[i][:, :D0] - h0[:, 0, :]).max() < 1e-4 and np.abs(c2[i][:, :H0] - c0[:, 0, :]).max() < 1e-4
    pcm = speech_like_pcm(3.0, seed=18, silence=(1.0, 1.4))
    want, lg0, n0 = run_oracle(om, pcm, 1600)
    got, lg1, n1 = run_gpu(gm, pcm, 1600)
    assert n0 == n1 and lg0.shape == lg1.shape and lg1.shape[1] == 60
    assert np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)
    gm.close(); om.close()


def test_any_width_is_padded_at_load(model_dir, built):
    """d 100, cell 150, ffn 210, joiner 70, 20 conv-3 channels -- not even multiples of 16: padded to 128 / 192 / 256 / 128 / 64 at load; a full
    session (logits within 1e-3, transcript) against the oracle, which runs the file's widths"""
    import april_asr_amd as A
    from april_asr_amd import synth_model as SM
    from oracle import orc_py as O
    p = str(model_dir / "odd100.april")
    SM.write_model(p, SM.ODD_DIMS, seed=9)
    gm = A.Model(p); om = O.Model(p)
    d = gm.dims
    assert (d.d_model, d.hidden, d.ffn, d.joiner, d.vocab, d.d_model_file) == (128, 192, 256, 128, 45, 100)
    pcm = speech_like_pcm(3.0, seed=28, silence=(1.0, 1.4))
    want, lg0, n0 = run_oracle(om, pcm, 1600)
    got, lg1, n1 = run_gpu(gm, pcm, 1600)
    assert n0 == n1 and lg0.shape == lg1.shape and lg1.shape[1] == 45
    assert np.abs(lg0 - lg1).max() < 1e-3
    assert_same_transcript(want, got)
    gm.close(); om.close()


def test_config5_larger_encoder_512_sessions(large_model):
    """BASELINE configs[4] shape (fp32 here): the larger encoder (16 x {768, 1536, 3072}), 512 concurrent sessions in
    100 ms feeds on one GPU.  Session 0 against the CPU oracle (token-exact, logits within 1e-3); sessions 1 and 511
    against themselves stepped alone (bit-identical)."""
    import april_asr_amd as A
    from oracle import orc_py as O
    gm = A.Model(large_model["path"]); om = O.Model(large_model["path"])
    d = gm.dims
    assert (d.n_layers, d.d_model, d.hidden, d.ffn, d.joiner) == (16, 768, 1536, 3072, 768)
    n, secs = 512, 0.6
    pcms = [speech_like_pcm(secs, seed=40)] + [O.lcg_pcm16_fast(int(16000 * secs), seed=300 + i) for i in range(1, n)]
    evs = [[] for _ in range(n)]
    sess = [A.Session(gm, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]
    for i in (0, 1, n - 1):
        sess[i].trace_logits(200)
    grp = A.SessionGroup(sess)
    for o in range(0, int(16000 * secs), 1600):
        grp.feed([p[o:o + 1600] for p in pcms])
    grp.flush()
    assert gm.stats().max_batch_seen == n
    want, lg0, n0 = run_oracle(om, pcms[0], 1600)
    assert sess[0].chunks() == n0
    lg = sess[0].traced_logits()
    assert lg.shape == lg0.shape and np.abs(lg - lg0).max() < 1e-3, np.abs(lg - lg0).max()
    assert_same_transcript(want, evs[0])
    for i in (1, n - 1):
        ev1, lg1, _ = run_gpu(gm, pcms[i], 1600)
        assert np.array_equal(lg1, sess[i].traced_logits()) and ev1 == evs[i]
    for s in sess:
        s.close()
    gm.close(); om.close()


def test_config4_2048_sessions_aprilv0_dims(v0_model):
    """BASELINE configs[3]: 2048 concurrent sessions at aprilv0 dimensions (12 x {512, 1024, 2048}) -- the whole node's session count
    on the ONE GPU of a test box, in 100 ms feeds (feeds of 2 / 3 chunk steps = 4096 / 6144 rows per wavefront, the GM_TILE schedules of
    the gates / projection / FFN GEMMs).  Session 0 against the CPU oracle (token-exact, logits within 1e-3); sessions 1, 1000 and 2047
    against themselves stepped alone (bit-identical: batch invariance across every GEMM schedule the row counts select)."""
    import april_asr_amd as A
    from oracle import orc_py as O
    gm = A.Model(v0_model["path"]); om = O.Model(v0_model["path"])
    d = gm.dims
    assert (d.n_layers, d.d_model, d.hidden, d.ffn) == (12, 512, 1024, 2048)
    n, secs = 2048, 0.6
    pcms = [speech_like_pcm(secs, seed=41)] + [O.lcg_pcm16_fast(int(16000 * secs), seed=900 + i) for i in range(1, n)]
    evs = [[] for _ in range(n)]
    sess = [A.Session(gm, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]
    probe = (0, 1, 1000, n - 1)
    for i in probe:
        sess[i].trace_logits(200)
    grp = A.SessionGroup(sess)
    for o in range(0, int(16000 * secs), 1600):
        grp.feed([p[o:o + 1600] for p in pcms])
    grp.flush()
    assert gm.stats().max_batch_seen == n and gm.stats().replay_mismatch == 0
    want, lg0, n0 = run_oracle(om, pcms[0], 1600)
    assert sess[0].chunks() == n0
    lg = sess[0].traced_logits()
    assert lg.shape == lg0.shape and np.abs(lg - lg0).max() < 1e-3, np.abs(lg - lg0).max()
    assert_same_transcript(want, evs[0])
    for i in probe[1:]:
        ev1, lg1, _ = run_gpu(gm, pcms[i], 1600)
        assert np.array_equal(lg1, sess[i].traced_logits()) and ev1 == evs[i]
    for s in sess:
        s.close()
    gm.close(); om.close()


def test_2048_sessions_one_gpu(gpu_tiny):
    """BASELINE configs[3]'s session count on ONE GPU (tiny dimensions so that it runs in seconds): 2048 concurrent
    sessions advance in shared steps of 2048 rows; sampled sessions are bit-identical to themselves stepped alone (size-independent property:
    batch invariance)."""
    import april_asr_amd as A
    from oracle import orc_py as O
    n = 2048
    pcms = [O.lcg_pcm16_fast(8000, seed=7000 + i) for i in range(n)]
    counts = np.zeros(6, np.uint64)
    evs = {i: [] for i in (0, 1, 777, 2047)}
    sess = []
    for i in range(n):
        if i in evs:
            sess.append(A.Session(gpu_tiny, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True))
            sess[-1].trace_logits(120)
        else:
            sess.append(A.Session(gpu_tiny, None, counters=counts))
    grp = A.SessionGroup(sess)
    for o in range(0, 8000, 1600):
        grp.feed([p[o:o + 1600] for p in pcms])
    grp.flush()
    assert gpu_tiny.stats().max_batch_seen == n
    assert int(counts[0]) > 0
    for i in evs:
        ev1, lg1, _ = run_gpu(gpu_tiny, pcms[i], 1600)
        assert np.array_equal(lg1, sess[i].traced_logits()) and ev1 == evs[i]
    for s in sess:
        s.close()


@pytest.mark.parametrize("which", ["tiny", "medium", "v0"])
def test_networks_match_torch_fp32(which, request):
    """The HIP kernels against a plain PyTorch fp32 statement of the network (tests/torch_ref.py: torch.nn.LSTM with
    proj_size, conv2d, linear), independent of the oracle's ONNX interpreter.  Tolerance 1e-4 per network call."""
    import april_asr_amd as A
    from april_asr_amd import synth_model as SM
    import torch_ref as TR
    mdl = request.getfixturevalue({"tiny": "tiny_model", "medium": "medium_model", "v0": "v0_model"}[which])
    dims = mdl["dims"]
    w = mdl["weights"] if mdl["weights"] is not None else SM.make_weights(dims)      # v0 fixture: default seed, not kept in memory
    gm = A.Model(mdl["path"])
    d = gm.dims
    rng = np.random.RandomState(21)
    n = 3
    x = rng.uniform(-16, 8, size=(n, d.seg, d.mel)).astype(np.float32)
    h = rng.uniform(-0.5, 0.5, size=(n, d.n_layers, d.d_model)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(n, d.n_layers, d.hidden)).astype(np.float32)
    eout, h2, c2 = gm.run_encoder(x, h, c)
    ctx = rng.randint(0, d.vocab, size=(n, d.context)).astype(np.int64)
    dout = gm.run_decoder(ctx)
    e = rng.uniform(-2, 2, size=(n, d.joiner)).astype(np.float32)
    lg = gm.run_joiner(e, dout)
    for i in range(n):
        e1, h1, c1 = TR.encoder(w, dims, x[i], h[i], c[i])
        assert np.abs(eout[i] - e1).max() < 1e-4 and np.abs(h2[i] - h1).max() < 1e-4 and np.abs(c2[i] - c1).max() < 1e-4
        assert np.abs(dout[i] - TR.decoder(w, dims, ctx[i])).max() < 1e-4
        assert np.abs(lg[i] - TR.joiner(w, dims, e[i], dout[i])).max() < 1e-4
    gm.close()


def test_session_churn_while_stepping(gpu_tiny):
    """Sessions are created, fed and freed on one thread while another thread steps a group whose size keeps changing
    (every new batch size captures a new hipGraph on the engine's stream; aas_free zeroes slots through the same stream).
    The watched session's transcript and logits must equal the same session run alone."""
    import threading
    import april_asr_amd as A
    from oracle import orc_py as O
    pcm = O.lcg_pcm16_fast(16000 * 2, seed=4242)
    want_ev, want_lg, _ = run_gpu(gpu_tiny, pcm, 1600)
    stop = threading.Event()
    errors = []

    def churn():
        rng = np.random.RandomState(5)
        try:
            while not stop.is_set():
                ss = [A.Session(gpu_tiny, lambda t, toks: None) for _ in range(int(rng.randint(1, 6)))]
                for s in ss:
                    s.feed_pcm16(O.lcg_pcm16_fast(1600 * int(rng.randint(1, 4)), seed=int(rng.randint(1 << 20))))
                for s in ss:
                    s.close()
        except Exception as e:                      # pragma: no cover
            errors.append(e)

    th = threading.Thread(target=churn)
    th.start()
    try:
        ev = []
        watched = A.Session(gpu_tiny, lambda t, toks: ev.append((t, toks)), raw_events=True)
        watched.trace_logits(400)
        extras = []
        for o in range(0, pcm.size, 1600):
            if (o // 1600) % 3 == 0:                 # the group changes size every few feeds
                extras.append(A.Session(gpu_tiny, lambda t, toks: None))
            if (o // 1600) % 5 == 4 and extras:
                extras.pop().close()
            grp = A.SessionGroup([watched] + extras)
            grp.feed([pcm[o:o + 1600]] + [O.lcg_pcm16_fast(1600, seed=o + k) for k in range(len(extras))])
        watched.flush()
    finally:
        stop.set(); th.join()
    assert not errors
    assert ev == want_ev and np.array_equal(watched.traced_logits(), want_lg)
    watched.close()
    for s in extras:
        s.close()
