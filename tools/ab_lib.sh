# A/B of two builds of the library over session counts: tools/ab_lib.sh <libA.so> <libB.so> sessions...
# (APRIL_ASR_LIB selects the library april_asr_amd/_ffi.py loads)
A=$1; B=$2; shift 2
for b in "$@"; do
  for lib in "$A" "$B"; do
    APRIL_ASR_LIB=$lib python bench.py --sessions $b --steps 20 --warmup 6 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b$b $lib', d['ms_per_step'], 'lockstep', d.get('other_ingest',{}).get('ms_per_step'))"
  done
done
