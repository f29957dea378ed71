#!/bin/bash
for v in ${VARIANTS:-1 2}; do for a in 0 1 2 3 4 6 7; do
  NAFP_GEMM_VARIANT=$v NAFP_ABL=$a python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v abl $a |', d['value'], d['stage_ms_per_step']['per_conv'][:9])"
done; done
