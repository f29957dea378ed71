#!/bin/bash
# Tile plan of the transposed convs (DGRAD) per layer: forces (column width : split-K) for every DGRAD launch on 128-row tiles and
# prints the per-layer table of one train step.   gpurun -- bash tools/dgrad_plan_sweep.sh B
B=${1:-640}
bash tools/layer_table.sh $B "NAFP_DGRAD_KSTEPS_OLD=1" "NAFP_X=0" "NAFP_DGRAD_PLAN=128:1" "NAFP_DGRAD_PLAN=128:2" "NAFP_DGRAD_PLAN=128:3" "NAFP_DGRAD_PLAN=64:1" "NAFP_DGRAD_PLAN=64:2" 2>&1 | grep -E "^==|^  1[0-5]|^   9|^   [6-8]"
