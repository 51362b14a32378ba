timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for b in 256 1024 1536; do echo "B=$b"; timeout 300 python bench.py --steps 30 --warmup 5 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rtf'], d['host_phase_ms_total'], d['step_latency_ms']['p50'], d['step_latency_ms']['p99'])"; done
