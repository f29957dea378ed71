#!/bin/bash
# A/B of library variants selected by environment variables, interleaved in one run on one box (box-to-box spread is
# 5 %, run-to-run on one box < 1 %):   tools/ab_bench.sh [-r ROUNDS] "NAFP_X=1" "NAFP_X=2 NAFP_Y=3" ...
# prints, per variant and round: segments/s, ms of the 15 GEMM convs, then per-conv ms (conv0 | convs 1-5 | 6-9 | 10-15).
ROUNDS=2
if [ "$1" = "-r" ]; then ROUNDS=$2; shift 2; fi
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    env $v python bench.py --steps 15 --warmup 4 --repeats 3 --no-cpu-baseline --no-pipelined --no-train --no-e2e 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); pc=d['stage_ms_per_step']['per_conv']; print('%-28s' % '$v', d['value'], d['stage_ms_per_step']['conv_gemm x15'], round(pc[0],3), [round(x,3) for x in pc[1:6]], [round(x,3) for x in pc[6:10]], [round(x,3) for x in pc[10:16]], 'mel', d['stage_ms_per_step']['melspec(2 kernels)'])"
  done
done
