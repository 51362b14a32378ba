for t in 0 1 2; do echo "TUNE=$t"; APRIL_GEMM_TUNE=$t timeout 300 python bench.py --steps 20 --warmup 3 --sessions 256 --no-cpu-baseline --no-sweep --profile-steps 4 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rtf'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['class_ms'])"; done
