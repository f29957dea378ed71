#!/bin/bash
# Whole per-layer table (forward / ln_bwd / wgrad / dgrad, us and TFLOP/s) of one train step per env setting.
#   tools/layer_table.sh B "ENV=.." ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
B=$1; shift
for v in "$@"; do
  O=/tmp/ltab_$$; rm -rf $O; mkdir -p $O
  env $v rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python tools/train_probe.py $B lamb 4 > $O/probe.txt 2> $O/err.txt
  F=$(find $O -name "*kernel_trace.csv" | head -1)
  echo "== $v :: $(grep 'train_step wall' $O/probe.txt) :: $(grep '^backward' $O/probe.txt)"
  python tools/train_layer_table.py $F $B
  if [ -n "$LAYER_TABLE_TIMELINE" ]; then python tools/step_timeline.py $F 0 | tail -45; fi
  rm -rf $O
done
