"""BASELINE configs[0] equivalent: the ./main-style C++ program (examples/main.cpp) built only against
include/april_api.h, fed a PCM file, prints the reference's "- partial" / "@ final" lines; the text must
equal what the oracle's transcript renders to."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import speech_like_pcm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def render(events, token_text):
    lines = []
    for typ, toks in events:
        if typ == 4:
            lines.append("")
        elif typ in (1, 2):
            lines.append(("@ " if typ == 2 else "- ") + "".join(token_text(t[0]) for t in toks))
    return lines


@pytest.mark.parametrize("container", ["raw", "wav"])
def test_main_cli_matches_oracle(tiny_model, tmp_path, container):
    from oracle import orc_py as O
    exe = str(tmp_path / "main")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "main.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "april_asr_amd"), "-laprilasr", "-Wl,-rpath," + os.path.join(ROOT, "april_asr_amd"), "-o", exe])
    pcm = np.concatenate([speech_like_pcm(3.0, seed=12), np.zeros(16000 * 3, np.int16), speech_like_pcm(1.0, seed=13)])
    path = str(tmp_path / ("audio." + container))
    with open(path, "wb") as f:
        if container == "wav":
            data = pcm.tobytes()
            f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                    + b"data" + struct.pack("<I", len(data)))
            f.write(data)
        else:
            f.write(pcm.tobytes())
    out = subprocess.run([exe, path, tiny_model["path"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    got = out.stdout.decode().split("\n")[:-1]
    om = O.Model(tiny_model["path"])
    s = O.Session(om)
    for o in range(0, pcm.size, 1600):
        s.feed(pcm[o:o + 1600])
    s.flush()
    want = render(s.events, om.token)
    assert got == want and len(got) > 3 and all(l == "" or l[:2] in ("- ", "@ ") for l in got)
    s.close(); om.close()


def test_main_cli_aprilv0_10s_wav(v0_model, tmp_path):
    """BASELINE configs[0] at its own size: an aprilv0-dimension model, one session, a 10 s 16 kHz mono PCM16 WAV file through
    the ./main-style program (only include/april_api.h and the 12 reference symbols); the printed lines are the oracle's
    transcript of the same samples."""
    from oracle import orc_py as O
    exe = str(tmp_path / "main")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "main.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "april_asr_amd"), "-laprilasr", "-Wl,-rpath," + os.path.join(ROOT, "april_asr_amd"), "-o", exe])
    pcm = np.concatenate([speech_like_pcm(4.0, seed=21, silence=(1.5, 1.9)), np.zeros(16000 * 3, np.int16), speech_like_pcm(3.0, seed=22)])
    assert pcm.size == 160000
    path = str(tmp_path / "ten_seconds.wav")
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                + b"data" + struct.pack("<I", len(data)))
        f.write(data)
    out = subprocess.run([exe, path, v0_model["path"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    got = out.stdout.decode().split("\n")[:-1]
    om = O.Model(v0_model["path"])
    s = O.Session(om)
    for o in range(0, pcm.size, 1600):
        s.feed(pcm[o:o + 1600])
    s.flush()
    want = render(s.events, om.token)
    assert got == want and len(got) > 3 and any(l.startswith("@ ") for l in got)
    s.close(); om.close()


def _stamp(ms):
    """hh:mm:ss,mmm as the reference's example_srt.cpp computes it (units peeled off while the remainder EXCEEDS a unit)."""
    out = []
    for unit in (3600 * 1000, 60 * 1000, 1000):
        n = (ms - 1) // unit if ms > unit else 0
        ms -= n * unit
        out.append(n)
    return "%02d:%02d:%02d,%03d" % (out[0], out[1], out[2], ms)


def test_srt_cli_matches_oracle(v0_model, tmp_path):
    """the ./srt-style client (examples/srt.cpp, reference example_srt.cpp:57-129): one cue per
    token of every FINAL result, timed by AprilToken.time_ms; the whole file goes in ONE feed (layer-major schedule).  Expected
    cues are rendered from the oracle's transcript."""
    from oracle import orc_py as O
    exe = str(tmp_path / "srt")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "srt.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "april_asr_amd"), "-laprilasr", "-Wl,-rpath," + os.path.join(ROOT, "april_asr_amd"), "-o", exe])
    tiny_model = v0_model          # aprilv0 dimensions: this audio yields FINAL results with several tokens there
    pcm = np.concatenate([speech_like_pcm(4.0, seed=12), np.zeros(16000 * 3, np.int16)])
    path = str(tmp_path / "audio.raw")
    pcm.tofile(path)
    out = subprocess.run([exe, path, tiny_model["path"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    om = O.Model(tiny_model["path"])
    s = O.Session(om)
    for o in range(0, pcm.size, 1600):
        s.feed(pcm[o:o + 1600])
    s.flush()
    want, cue = [], 0
    for typ, toks in s.events:
        if typ != 2:
            continue
        text = ""
        for t, (tid, _lp, _fl, ms) in enumerate(toks):
            end = toks[t + 1][3] if t + 1 < len(toks) else ms + 2000
            cue += 1
            text += om.token(tid)
            want += [str(cue), "%s --> %s" % (_stamp(ms), _stamp(end)), text, ""]
    got = out.stdout.decode().split("\n")[:-1]
    assert got == want and cue >= 3


def test_serve_many_pipelined_equals_lockstep(v0_model, tmp_path):
    """examples/serve_many.cpp: 48 streams from one thread through the engine ABI's group feeds; the pipelined feed (depth 2) and
    the lock-step feed print the same per-stream results (callbacks, FINAL results, their tokens and the last FINAL text)."""
    exe = str(tmp_path / "serve_many")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "serve_many.cpp"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "april_asr_amd"), "-laprilasr", "-Wl,-rpath," + os.path.join(ROOT, "april_asr_amd"), "-o", exe])
    pcm = np.concatenate([speech_like_pcm(3.0, seed=31, silence=(1.0, 1.3)), np.zeros(16000 * 3, np.int16), speech_like_pcm(2.0, seed=32)])
    path = str(tmp_path / "audio.raw")
    pcm.tofile(path)
    outs = []
    for mode in ("pipelined", "lockstep"):
        r = subprocess.run([exe, v0_model["path"], path, "48", mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env=dict(os.environ, APRIL_MAX_SESSIONS="256", APRIL_MAX_BATCH="1024"))
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        outs.append(r.stdout.decode().splitlines())
    assert len(outs[0]) == 48 and outs[0] == outs[1]
    assert sum(int(l.split()[2]) for l in outs[0]) > 0, "no FINAL result in any stream"
