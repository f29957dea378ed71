#!/bin/bash
# per-conv ms of the exact-split forward (NAFP_OPT_BF16X3 = 2) for a list of variant libraries built by tools/build_variant.sh:
#   tools/x6_abl_run.sh a1 a512 ...      ("base" = the in-tree library)
for r in 1 2; do
for v in base "$@"; do
  if [ "$v" = base ]; then L=""; else L="NAFP_LIB=neural-audio-fp_amd/_abl/libnafp_$v.so"; fi
  echo "== $v"; env $L X6_ONLY=1 python tools/x6_per_conv.py 2>&1 | tail -2
done
done
