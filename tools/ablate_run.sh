#!/bin/bash
# run on the GPU box: per-conv ms for each ablation variant
for v in base nostore nogb noload nomfma nostore_noload; do
  NAFP_LIB=$PWD/neural-audio-fp_amd/_abl/libnafp_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['stage_ms_per_step']['conv_gemm x15'], d['stage_ms_per_step']['per_conv'])"
done
