#!/bin/bash
# on the GPU box: parity tests + per-conv ms for each GEMM staging variant
for v in ${VARIANTS:-1 2 3 4 5}; do
  export NAFP_GEMM_VARIANT=$v
  t=$(python -m pytest tests -m gpu -x -q 2>&1 | tail -1)
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v | $t |', d['value'], d['roofline']['achieved'], d['stage_ms_per_step']['per_conv'])"
done
