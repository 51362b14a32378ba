"""Worker for the scheduler tests: streams `nsess` sessions for `steps` feeds of `feed` samples through the C ABI in one of the
ingest modes and prints a digest of every callback (token ids, log-probabilities bit for bit, flags, times) in arrival order per
session.  Modes: sync = aprilx_feed_many (the caller lends its buffers and blocks), pipe<d> = aprilx_feed_many_pipelined with
depth d, async = asynchronous sessions (handlers on the library thread) fed through aprilx_feed_many_pipelined depth 2.
usage: stream_worker.py model.april nsess steps mode [feed=1600] [flush=1]"""
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import april_asr_amd as A  # noqa: E402
from april_asr_amd import synth_model as SM  # noqa: E402


def main():
    path, nsess, steps, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    feed = int(sys.argv[5]) if len(sys.argv) > 5 else 1600
    flush = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    m = A.Model(path)
    events = [[] for _ in range(nsess)]

    def handler(i):
        return lambda t, toks: events[i].append((int(t), [(x[0], struct.pack("<f", float(x[1])), int(x[2]), int(x[3])) for x in toks]))

    sess = [A.Session(m, handler(i), raw_events=True, asynchronous=(mode == "async"), no_rt=(mode == "async")) for i in range(nsess)]
    grp = A.SessionGroup(sess)
    pcm = [SM.lcg_pcm16(feed * steps, seed=4242 + i) for i in range(nsess)]
    grp.plan(pcm, feed)
    prof = int(os.environ.get("APRIL_TEST_PROFILE", "0"))      # 2: the gates clock from the 5th feed on (aprilx_model_profile(model, 2))
    for k in range(steps):
        if prof and k == 4:
            grp.drain()
            m.profile(prof)
        if mode == "sync":
            grp.feed_planned(k)
        else:
            grp.feed_planned_pipelined(k, 2 if mode == "async" else int(mode[4:]))
    grp.drain()
    if prof:
        m.profile(0)
        sg = m.stats()
        print("GCLOCK", int(sg.gates_clock_launches), "%.4f" % float(sg.gates_clock_ms), int(sg.gates_clock_rows), flush=True)
    lat = m.feed_latencies(reset=True)          # hand-over -> delivery per tick (aprilx_model_feed_latency), before the flush
    if flush:
        grp.flush()
    h = hashlib.sha256()
    ntok = 0
    for i in range(nsess):
        h.update(repr(events[i]).encode())
        ntok += sum(len(t) for _, t in events[i])
    st = m.stats()
    per_engine = [int(m.stats(i).chunks) for i in range(16)]          # (device_index beyond the model's engines reads zeros)
    print("ENGINES", " ".join(str(c) for c in per_engine), flush=True)
    st.chunks = sum(per_engine); st.replay_mismatch = sum(int(m.stats(i).replay_mismatch) for i in range(16)); st.flights = sum(int(m.stats(i).flights) for i in range(16))
    print("DIGEST", h.hexdigest(), int(st.chunks), int(st.replay_mismatch), sum(len(e) for e in events), ntok, int(st.flights), flush=True)
    print("LATENCY", lat.size, "%.4f" % (float(lat.min()) if lat.size else -1.0), "%.4f" % (float(lat.max()) if lat.size else -1.0), flush=True)
    for s in sess:
        s.close()
    m.close()


if __name__ == "__main__":
    main()
