#!/bin/bash
# per-conv ms (convs 5..15) of the forward pass with every 128-row FULL launch forced to one (tile columns, split-K factor)
for plan in "0:0" "64:1" "64:2" "64:3" "64:4" "64:6" "64:8" "128:1" "128:2" "128:3" "128:4" "128:6" "128:8"; do
  NAFP_FWD_PLAN=$plan python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-train 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); pc=d['stage_ms_per_step']['per_conv']; print('%-6s' % '$plan', d['value'], ' '.join('%.3f' % x for x in pc[5:16]))"
done
