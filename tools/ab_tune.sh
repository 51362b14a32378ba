for rep in 1 2 3; do for b in 256 1024; do for f in 100000 64; do echo -n "rep=$rep B=$b FUSE_HR_MIN=$f  "; APRIL_FUSE_HR_MIN=$f timeout 300 python bench.py --steps 30 --warmup 5 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_latency_ms'])"; done; done; done
