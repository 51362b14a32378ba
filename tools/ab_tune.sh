timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for b in 256 1024 1536 2048; do echo "B=$b"; timeout 300 python bench.py --steps 20 --warmup 3 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 2 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rtf'], d['host_phase_ms_total'])"; done
