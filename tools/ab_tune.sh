timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for b in 256 1024; do echo "B=$b"; timeout 300 python bench.py --steps 30 --warmup 5 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rtf'], d['roofline']['class_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
