#!/bin/bash
# per-conv ms of the GEMM conv with the operand staging taken apart (NAFP_ABL bits: 1 = no DMA after the prologue,
# 512 = A lanes all out of range (DMA issued, LDS written with zeros, no memory traffic), 1024 = every tile stages the A rows
# of tiles 0..7 (same instructions, L2 hits), 2 = no epilogue); results are wrong by construction.
for v in 0 1 512 1024 2 3 514 1026 0; do
  NAFP_ABL=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-pipelined --no-train 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pc=d['stage_ms_per_step']['per_conv']; print('abl=%4d' % $v, d['value'], d['stage_ms_per_step']['conv_gemm x15'], [round(x,3) for x in pc[1:6]], round(sum(pc[6:10]),3), round(sum(pc[10:16]),3))"
done
