for lanes in 1 2 3; do for b in 256 1024 2048; do echo "LANES=$lanes B=$b"; APRIL_LANES=$lanes timeout 300 python bench.py --steps 20 --warmup 4 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rtf'], d['max_batch_seen'])"; done; done
