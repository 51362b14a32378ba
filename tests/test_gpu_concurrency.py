"""Many client threads on one model, the way a server uses the C API: every thread owns one synchronous session and feeds it 100 ms
at a time (aas_feed_pcm16 blocks until that feed is processed), while another thread keeps creating, feeding and freeing short-lived
sessions.  The stepping thread then sees flights of changing shapes, launches flight k + 1 while flight k is in the air (other
clients' sessions: csrc/session.cc Scheduler::loop), splits them over the three streams, resets freed slots in batches -- and every
session must still produce exactly the callbacks it produces alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def transcript(model, pcm, step=1600):
    import april_asr_amd as A
    ev = []
    s = A.Session(model, lambda t, toks: ev.append((t, toks)), raw_events=True)
    for o in range(0, pcm.size, step):
        s.feed_pcm16(pcm[o:o + step])
    s.flush()
    s.close()
    return ev


@pytest.mark.parametrize("which", ["medium", "v0"])
def test_client_threads_and_session_churn(which, request):
    import april_asr_amd as A
    from oracle import orc_py as O
    path = request.getfixturevalue(which + "_model")["path"]
    m = A.Model(path)
    n_threads, secs = 12, 2.0
    pcms = [O.lcg_pcm16_fast(int(16000 * secs), seed=5000 + i) for i in range(n_threads)]
    want = [transcript(m, p) for p in pcms]
    got = [None] * n_threads
    errors = []
    stop = threading.Event()

    def client(i):
        try:
            got[i] = transcript(m, pcms[i], step=1600 if i % 3 else 800)
        except Exception as e:                      # noqa: BLE001
            errors.append(repr(e))

    def churn():
        k = 0
        try:
            while not stop.is_set():
                p = O.lcg_pcm16_fast(4800, seed=9000 + k)
                transcript(m, p)
                k += 1
        except Exception as e:                      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_threads)]
    ch = threading.Thread(target=churn)
    ch.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    stop.set()
    ch.join(timeout=60)
    assert not errors, errors
    st = m.stats()
    assert st.replay_mismatch == 0
    for i in range(n_threads):
        assert got[i] == want[i], "session %d differs when fed beside other clients" % i
    assert st.max_batch_seen >= 2, "the clients never shared a step"
    m.close()
