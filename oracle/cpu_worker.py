"""ORACLE (test infrastructure, NOT product code): one CPU worker of bench.py's multi-core cpu_baseline leg.
Loads the model with the plain-C oracle, waits for the start file, runs one session over `seconds` of LCG-noise PCM16 in
100 ms feeds, prints the wall time of the feeding loop.  usage: cpu_worker.py <model> <seconds> <seed> <start_file>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc_py as O          # noqa: E402

path, seconds, seed, start_file = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
om = O.Model(path)
sess = O.Session(om)
pcm = O.lcg_pcm16_fast(int(16000 * seconds), seed=seed)
print("ready", flush=True)
while not os.path.exists(start_file):
    time.sleep(0.01)
a = time.perf_counter()
for o in range(0, pcm.size, 1600):
    sess.feed(pcm[o:o + 1600])
b = time.perf_counter()
print("elapsed %.6f %d" % (b - a, sess.chunks()), flush=True)
