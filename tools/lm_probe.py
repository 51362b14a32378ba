#!/usr/bin/env python3
"""One session, one long feed (layer-major path), untraced; prints progress markers.  usage: lm_probe.py [tiny|v0] [seconds]"""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import april_asr_amd as A
from april_asr_amd import synth_model as SM
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
path = "/tmp/lm_probe_%s.april" % which
if not os.path.exists(path):
    SM.write_model(path, SM.TINY_DIMS if which == "tiny" else SM.APRILV0_DIMS)
m = A.Model(path); print("model ok", flush=True)
pcm = SM.lcg_pcm16(int(16000 * secs), seed=5)
prev = None
for rep in range(int(os.environ.get("LM_PROBE_REPS", "2"))):
    ev = []
    s = A.Session(m, lambda t, toks: ev.append(t), raw_events=True)
    a = time.perf_counter(); s.feed_pcm16(pcm); b = time.perf_counter()
    print("rep %d feed ok: %d chunks, %d callbacks, %.2f ms (%.1f us/chunk)" % (rep, s.chunks(), len(ev), (b - a) * 1e3, (b - a) * 1e6 / max(1, s.chunks())), flush=True)
    s.flush(); print("flush ok", flush=True)
    s.close(); print("close ok", flush=True)
    st = m.stats(); hm = list(st.host_ms)
    print("rep %d host_ms (this rep)" % rep, [round(x - (prev[i] if prev else 0.0), 1) for i, x in enumerate(hm)], flush=True)
    prev = hm
st = m.stats(); print("lm_chunks", st.lm_chunks, "mismatch", st.replay_mismatch, "host_ms", [round(x, 1) for x in st.host_ms], flush=True)
m.close(); print("model closed", flush=True)
