cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "256 0" "256 1" "1 0" "1024 0"; do set -- $cfg
  rm -rf /tmp/tr; APRIL_NO_GRAPHS=$2 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --sessions $1 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --profile-steps 0 > /tmp/tr.log 2>&1
  echo "== sessions=$1 NO_GRAPHS=$2"; python tools/gap_summary.py $(ls /tmp/tr/*kernel_trace.csv /tmp/tr/*/*kernel_trace.csv 2>/dev/null | head -1)
done
