#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own C sources compiled into
oracle/_ref/libaprilref.so (oracle/Makefile `ref` target).  Run in the build container, where
/root/reference exists; the resulting vectors are data (inputs + expected outputs) and travel
with the repo, the reference sources do not.

  fbank_lcg.npz     : seeded LCG PCM16 (SURVEY.md Appendix E recipe) -> every 9x80 chunk the
                      reference fbank produces for 1 s of audio fed in 3200-sample segments, the
                      flush-phase chunks, chunk counts for 10 s (248 + 9/28), the sum of all
                      feed-phase values, window and mel tables.
  fbank_frames.npz  : 40 single frames (random / extreme / silent PCM) -> 80 log-mel values each.
  fbank_nonpow2.npz : models with round_pow2 = 0 (the FFT length is the frame length): per geometry 24 single frames -> log-mel
                      (400 = 4 4 5 5, 320 = 4 4 4 5, 480 = 2 4 4 3 5, 200 = 2 4 5 5 at 8 kHz, where the lowest of the 80 mel filters fall between FFT bins;
                      882 = 2 3 3 7 7, 1102 = 2 19 29, 441 = 3 3 7 7, 220 = 4 5 11: pocketfft's generic pass, lengths that are not multiples of 4), and for the 400-point
                      one the whole online fbank on 1 s of the LCG recipe (feed chunks + both flush phases).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def to_wave(pcm):
    return pcm.astype(np.float32) / np.float32(32768.0)


# (name, RefFbank keywords, FFT length) of the round_pow2 = 0 geometries
NONPOW2 = [("n400", dict(round_pow2=0), 400), ("n320", dict(round_pow2=0, len_ms=20), 320), ("n480", dict(round_pow2=0, len_ms=30), 480),
           ("n200", dict(round_pow2=0, rate=8000), 200),
           # lengths with other factors (pocketfft's generic pass) and not multiples of 4 (the other two twiddle constructions):
           # 882 = 2 3 3 7 7 (44.1 kHz / 20 ms), 1102 = 2 19 29 (44.1 kHz / 25 ms), 441 = 3 3 7 7 (odd), 220 = 4 5 11
           ("n882", dict(round_pow2=0, rate=44100, len_ms=20), 882), ("n1102", dict(round_pow2=0, rate=44100), 1102),
           ("n441", dict(round_pow2=0, rate=44100, len_ms=10), 441), ("n220", dict(round_pow2=0, rate=22050, len_ms=10), 220)]


def run_ref(pcm, seg=3200, **kw):
    fb = O.RefFbank(**kw)
    feed = []
    w = to_wave(pcm)
    for i in range(0, w.size, seg):
        fb.accept(w[i:i + seg])
        feed += fb.pull_all()
    ph1 = []
    while fb.flush():
        ph1 += fb.pull_all()
    fb.accept_zeros(3200); fb.accept_zeros(3200)
    ph2 = []
    while fb.flush():
        ph2 += fb.pull_all()
    return np.array(feed), np.array(ph1), np.array(ph2)


def main():
    assert O.ref_available(), "needs /root/reference (build container)"
    pcm10 = O.lcg_pcm16_fast(160000)
    feed10, p1, p2 = run_ref(pcm10)
    pcm1 = pcm10[:16000]
    feed1, q1, q2 = run_ref(pcm1)
    R = O.ref()
    win = np.zeros(512, np.float32); R.generate_povey_window(win.ctypes.data, 512)
    mel = np.zeros((80, 256), np.float32); R.generate_banks(mel.ctypes.data, 80, 256, 512, 16000, 20, 0)
    np.savez_compressed(os.path.join(HERE, "fbank_lcg.npz"), seed=12345, feed_1s=feed1, flush1_1s=q1, flush2_1s=q2,
                        n_feed_10s=len(feed10), n_flush1_10s=len(p1), n_flush_total_10s=len(p1) + len(p2),
                        sum_feed_10s=feed10.astype(np.float64).sum(), first_chunk_10s=feed10[0], last_flush_chunk_10s=p2[-1],
                        window=win, mel=mel)
    rng = np.random.RandomState(11)
    frames = np.concatenate([rng.randint(-32768, 32768, size=(24, 512)), rng.randint(-200, 200, size=(8, 512)),
                             np.zeros((2, 512)), np.full((2, 512), -32768), np.full((2, 512), 32767),
                             np.tile(np.array([32767, -32768]), (2, 256))]).astype(np.int16)
    out = []
    for f in frames:
        fb = O.RefFbank()
        fb.accept(np.concatenate([to_wave(f), np.zeros(8 * 160, np.float32)]))
        out.append(fb.pull_all()[0][0])
    np.savez_compressed(os.path.join(HERE, "fbank_frames.npz"), pcm=frames, logmel=np.array(out))
    out = {}
    rng = np.random.RandomState(23)
    for name, kw, n in NONPOW2:
        shift = (kw.get("rate", 16000) // 100)
        frames = np.concatenate([rng.randint(-32768, 32768, size=(14, n)), rng.randint(-200, 200, size=(4, n)), np.zeros((1, n)),
                                 np.full((1, n), -32768), np.full((1, n), 32767), np.resize(np.array([32767, -32768]), n)[None],
                                 O.lcg_pcm16_fast(2 * n, seed=77).reshape(2, n)]).astype(np.int16)
        rows = []
        for f in frames:
            fb = O.RefFbank(**kw)
            fb.accept(np.concatenate([to_wave(f), np.zeros(8 * shift, np.float32)]))
            rows.append(fb.pull_all()[0][0])
        out[name + "_pcm"] = frames
        out[name + "_logmel"] = np.array(rows)
    f400, a400, b400 = run_ref(pcm1, 3200, round_pow2=0)
    out.update(seed=12345, n400_feed_1s=f400, n400_flush1_1s=a400, n400_flush2_1s=b400)
    np.savez_compressed(os.path.join(HERE, "fbank_nonpow2.npz"), **out)
    print("wrote goldens:", feed1.shape, len(feed10), len(p1), len(p1) + len(p2), feed10.astype(np.float64).sum(), "| nonpow2", f400.shape, a400.shape, b400.shape)


if __name__ == "__main__":
    main()
