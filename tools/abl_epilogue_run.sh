#!/bin/bash
# per-conv ms of the GEMM conv with pieces of its epilogue removed (NAFP_ABL bits: 16 = no statistics reduction,
# 32 = no G/Hb/gamma loads, 64 = no stores, 128 = no exp; 2 = no epilogue at all); results are wrong by construction.
for v in 0 16 32 64 128 48 112 240 2 0; do
  NAFP_ABL=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-pipelined --no-train 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); pc=d['stage_ms_per_step']['per_conv']; print('abl=%3d' % $v, d['value'], d['stage_ms_per_step']['conv_gemm x15'], [round(x,3) for x in pc[1:6]], round(sum(pc[6:10]),3), round(sum(pc[10:16]),3))"
done
