timeout 900 python -m pytest tests/test_gpu_f16.py -m gpu -x -q -s 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or batch_invariant or decoder_and_joiner" 2>&1 | tail -5
for prec in f16; do for b in 256 1024 2048; do echo "PREC=$prec B=$b"; timeout 300 python bench.py --precision $prec --steps 20 --warmup 4 --sessions $b --no-cpu-baseline --no-sweep --profile-steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['rtf'], d['dtype'], d['roofline']['class_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done; done
