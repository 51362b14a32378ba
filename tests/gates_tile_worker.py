"""Worker for tests/test_gpu_gates_tile.py: streams a few sessions through the C ABI and prints a digest of every logit and
callback.  The parent runs it under APRIL_GATES_TILE=0 (hand-scheduled K-split gates kernel) and =1 (GM_TILE gates kernel)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import april_asr_amd as A  # noqa: E402
from april_asr_amd import synth_model as SM  # noqa: E402


def main():
    path, nsess, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    m = A.Model(path)
    events = [[] for _ in range(nsess)]
    sess = []
    for i in range(nsess):
        s = A.Session(m, (lambda t, toks, i=i: events[i].append((int(t), [(x[0], float(x[1]), int(x[2]), int(x[3])) for x in toks]))), raw_events=True)
        s.trace_logits(3 * (steps * 3 + 8))
        sess.append(s)
    grp = A.SessionGroup(sess)
    pcm = [SM.lcg_pcm16(1600 * steps, seed=777 + i) for i in range(nsess)]
    grp.plan(pcm, 1600)
    for k in range(steps):
        grp.feed_planned(k)
    h = hashlib.sha256()
    for i, s in enumerate(sess):
        h.update(np.ascontiguousarray(s.traced_logits()).tobytes())
        h.update(repr(events[i]).encode())
    st = m.stats()
    print("DIGEST", h.hexdigest(), int(st.chunks), int(st.replay_mismatch), flush=True)
    for s in sess:
        s.close()
    m.close()


if __name__ == "__main__":
    main()
