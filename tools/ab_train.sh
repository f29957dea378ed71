#!/bin/bash
# A/B of library variants on the TRAIN step, interleaved on one box:  tools/ab_train.sh [-r ROUNDS] [-b "640 1280"] "NAFP_X=0" "NAFP_X=1" ...
# prints the whole-step wall time of tools/train_probe.py per variant, batch size and round.
ROUNDS=2; SIZES="640 1280 5120"
while [ "${1:0:1}" = "-" ]; do
  if [ "$1" = "-r" ]; then ROUNDS=$2; shift 2; elif [ "$1" = "-b" ]; then SIZES=$2; shift 2; else break; fi
done
for r in $(seq 1 $ROUNDS); do
  for b in $SIZES; do
    steps=20; [ $b -ge 1280 ] && steps=12; [ $b -ge 5120 ] && steps=4
    for v in "$@"; do
      printf '%-34s B=%-5s ' "$v" $b
      env $v python tools/train_probe.py $b lamb $steps 2>/dev/null | grep "train_step wall"
    done
  done
done
