#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
for v in "APRIL_LM_NSTREAMS=3" "APRIL_LM_NSTREAMS=2" "APRIL_LM_NSTREAMS=4" "APRIL_LM_NSTREAMS=6" "APRIL_LM_NSTREAMS=3 APRIL_LM_BLOCK=5" "APRIL_LM_NSTREAMS=3 APRIL_LM_BLOCK=10" "APRIL_LM_NSTREAMS=12"; do
  echo "=== v0 $v"; env $v timeout 200 python tools/lm_probe.py v0 60 2>&1 | grep -E "rep 1|host_ms"; echo "rc=$?"
done > gpurun_out/h_probe3.txt 2>&1
cat gpurun_out/h_probe3.txt
