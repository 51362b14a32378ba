for rep in 1 2; do for t in 256 128 64; do for b in 1 256 1024; do echo -n "rep=$rep KZ_TARGET=$t B=$b  "; APRIL_KZ_TARGET=$t timeout 300 python bench.py --steps 30 --warmup 5 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['class_ms'])"; done; done; done
