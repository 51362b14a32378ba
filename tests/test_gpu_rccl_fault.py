"""Fault injection for the in-library weight distribution (csrc/april_api.cc broadcast_local; reference load site
src/april_model.c:57-61).  A one-GPU box cannot run a real multi-device RCCL broadcast, but it can make RCCL FAIL for real:
APRIL_FAULT_RCCL=1 with APRIL_GPU_DEVICES=0,0 hands ncclCommInitAll two ranks on the same device, which RCCL refuses.  The load
must then (a) fall back to device copies, loudly, and serve the same transcripts, (b) report used_rccl = 0, (c) fail cleanly --
no hang, no crash, NULL model -- under APRIL_STRICT_RCCL=1."""
import os
import pickle
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, pickle
sys.path.insert(0, %r)
import april_asr_amd as A
from april_asr_amd import synth_model as SM
try:
    m = A.Model(sys.argv[1])
except Exception as e:
    pickle.dump(("failed", repr(e)), open(sys.argv[2], "wb")); sys.exit(0)
li = m.load_info()
n = 6; pcms = [SM.lcg_pcm16(16000, seed=800 + i) for i in range(n)]
evs = [[] for _ in range(n)]
ss = [A.Session(m, (lambda k: (lambda t, toks: evs[k].append((t, toks))))(i), raw_events=True) for i in range(n)]
g = A.SessionGroup(ss)
for o in range(0, 16000, 1600): g.feed([p[o:o + 1600] for p in pcms])
g.flush()
pickle.dump(("ok", evs, int(m.dims.n_devices), int(li.used_rccl), int(li.ranks), [int(m.stats(i).chunks) for i in range(int(m.dims.n_devices))]), open(sys.argv[2], "wb"))
for s in ss: s.close()
m.close()
''' % ROOT


def run(path, out, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CODE, path, out], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return pickle.load(open(out, "rb")), r.stderr.decode()


def test_rccl_failure_falls_back_to_device_copies(tiny_model, tmp_path):
    ref, _ = run(tiny_model["path"], str(tmp_path / "a.pkl"), APRIL_GPU_DEVICES="0")
    got, err = run(tiny_model["path"], str(tmp_path / "b.pkl"), APRIL_GPU_DEVICES="0,0", APRIL_FAULT_RCCL=1)
    assert ref[0] == "ok" and got[0] == "ok"
    assert got[2] == 2 and got[3] == 0 and got[4] == 2, got[2:5]         # two engines, no RCCL, two broadcast peers
    assert "falling back to peer copies" in err and "RCCL" in err          # loud
    assert all(c > 0 for c in got[5])
    assert got[1] == ref[1] and any(len(e) for e in ref[1])


def test_rccl_failure_is_fatal_when_strict(tiny_model, tmp_path):
    got, err = run(tiny_model["path"], str(tmp_path / "c.pkl"), APRIL_GPU_DEVICES="0,0", APRIL_FAULT_RCCL=1, APRIL_STRICT_RCCL=1)
    assert got[0] == "failed"
    assert "APRIL_STRICT_RCCL" in err
