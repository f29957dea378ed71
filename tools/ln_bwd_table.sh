#!/bin/bash
# Per-layer LayerNorm-backward kernel times (us) of one train step, per env setting.  tools/ln_bwd_table.sh [B] "ENV=.." ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
B=$1; shift
for v in "$@"; do
  O=/tmp/lnsweep_$$; rm -rf $O; mkdir -p $O
  env $v rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python tools/train_probe.py $B lamb 4 > $O/probe.txt 2> $O/err.txt
  F=$(find $O -name "*kernel_trace.csv" | head -1)
  echo "== $v :: $(grep 'train_step wall' $O/probe.txt) :: $(grep '^backward' $O/probe.txt)"
  python tools/train_layer_table.py $F $B | awk '/^ +[0-9]+ /{printf "L%s %s | ", $1, $2} END{print ""}'
  rm -rf $O
done
