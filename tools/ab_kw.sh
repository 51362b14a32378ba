# A/B of a GM_KW knob over session counts: tools/ab_kw.sh "<env assignment A>" "<env assignment B>" sessions...
A=$1; B=$2; shift 2
for b in "$@"; do
  for cfg in "$A" "$B"; do
    env $cfg python bench.py --sessions $b --steps 20 --warmup 6 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b$b $cfg', d['ms_per_step'])"
  done
done
