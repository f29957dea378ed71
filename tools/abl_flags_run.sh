#!/bin/bash
# per-conv ms of the GEMM conv under the in-kernel ablation flags (NAFP_ABL: 1 = no DMA after the
# prologue, 2 = no epilogue, 4 = no MFMA/LDS reads, 8 = prologue + pipeline fill only); results are wrong by construction.
for v in 0 1 2 3 4 6 8; do
  NAFP_ABL=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-train 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('abl=$v', d['value'], d['stage_ms_per_step']['conv_gemm x15'], d['stage_ms_per_step']['per_conv'][1:8])"
done
