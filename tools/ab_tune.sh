APRIL_FUSE_ROW_MAX=100000 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for b in 1 16 256 1024; do for f in 0 100000; do echo -n "rep=$rep B=$b FUSE_ROW_MAX=$f  "; APRIL_FUSE_ROW_MAX=$f timeout 300 python bench.py --steps 30 --warmup 5 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_latency_ms']['p50'])"; done; done; done
