#!/bin/bash
# Sweep of the small-layer wgrad plan on the GPU box: per setting, the per-layer kernel times of one rank-640 train step
# (rocprofv3 kernel trace of tools/train_probe.py).   tools/wgrad_sweep.sh "NAFP_WGRAD_SP_FORCE=64:1" "NAFP_WGRAD_SP_EFF1=70" ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
B=${WGRAD_SWEEP_B:-640}
for v in "$@"; do
  O=/tmp/wsweep_$$; rm -rf $O; mkdir -p $O
  env $v rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python tools/train_probe.py $B lamb 4 > $O/probe.txt 2> $O/err.txt
  F=$(find $O -name "*kernel_trace.csv" | head -1)
  echo "== $v :: $(grep 'train_step wall' $O/probe.txt)"
  python tools/train_layer_table.py $F $B | awk '/^ +1[0-5] /{printf "L%s wg %s | ", $1, $3} END{print ""}'
  rm -rf $O
done
