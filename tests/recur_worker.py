"""Worker for tests/test_gpu_recur_kernels.py: a few sessions hand over whole recordings in one call (layer-major schedule:
the recurrent GEMMs of every time step), then stream some 100 ms feeds (the projection at a handful of rows); prints a digest
of every logit and callback.  The parent runs it with APRIL_RECUR_KERNELS=0 (general GEMM kernels) and =1 (kernels_recur.hip)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import april_asr_amd as A  # noqa: E402
from april_asr_amd import synth_model as SM  # noqa: E402


def main():
    path, nsess, secs = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    m = A.Model(path)
    events = [[] for _ in range(nsess)]
    sess = []
    for i in range(nsess):
        s = A.Session(m, (lambda t, toks, i=i: events[i].append((int(t), [(x[0], float(x[1]), int(x[2]), int(x[3])) for x in toks]))), raw_events=True)
        s.trace_logits(4000)
        sess.append(s)
    grp = A.SessionGroup(sess)
    # whole recordings of different lengths in one call (ragged group: several layer-major steps), then streamed feeds
    grp.feed([SM.lcg_pcm16(int(16000 * secs) + 977 * i, seed=4100 + i) for i in range(nsess)])
    for k in range(5):
        grp.feed([SM.lcg_pcm16(1600, seed=4200 + 10 * k + i) for i in range(nsess)])
    grp.flush()
    h = hashlib.sha256()
    for i, s in enumerate(sess):
        h.update(np.ascontiguousarray(s.traced_logits()).tobytes())
        h.update(repr(events[i]).encode())
    st = m.stats()
    print("DIGEST", h.hexdigest(), int(st.chunks), int(st.lm_chunks), int(st.replay_mismatch), flush=True)
    for s in sess:
        s.close()
    m.close()


if __name__ == "__main__":
    main()
