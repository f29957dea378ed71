#!/bin/bash
# Samples rocm-smi (socket power, sclk, mclk, temperature, perf level) every 0.2 s while a train probe / bench runs:
#   tools/power_probe.sh "python tools/train_probe.py 5120 lamb 12"   -> gpurun_out/power_probe.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt; : > $OUT
rocm-smi --showpower --showclocks --showtemp --showperflevel --showmaxpower 2>&1 | head -40 >> $OUT
( $1 > gpurun_out/power_probe_cmd.txt 2>&1 ) &
PID=$!
while kill -0 $PID 2>/dev/null; do
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk|fclk' | sed 's/GPU\[0\][ \t]*: //' | tr '\n' ';')" >> $OUT
  sleep 0.2
done
tail -3 gpurun_out/power_probe_cmd.txt >> $OUT
