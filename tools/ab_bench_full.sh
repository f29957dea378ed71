#!/bin/bash
# like ab_bench.sh, but prints every per-conv time (ms):   tools/ab_bench_full.sh [-r ROUNDS] "ENV=..." ...
ROUNDS=2
if [ "$1" = "-r" ]; then ROUNDS=$2; shift 2; fi
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    env $v python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-pipelined --no-train 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); pc=d['stage_ms_per_step']['per_conv']; print('%-20s' % '$v'[-20:], d['value'], ' '.join('%.3f' % x for x in pc[:16]))"
  done
done
