#!/bin/bash
# FETCH_SIZE (x2, gfx950) summed over the wgrad / GEMM-conv launches of a train step at BSZ 1280:  tools/pmc_fetch_train.sh [ENV=..]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
rm -rf gpurun_out/pmc_tr; mkdir -p gpurun_out/pmc_tr
env "$@" rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_tr -o p -- python tools/train_probe.py 1280 adam 3 > /dev/null 2>&1
python - <<PY
import csv
from collections import defaultdict
rows=list(csv.DictReader(open("gpurun_out/pmc_tr/p_counter_collection.csv")))
d=defaultdict(lambda:[0,0.0])
for r in rows:
    if r["Counter_Name"]=="FETCH_SIZE":
        n=r["Kernel_Name"].split("(")[0].replace("nafp::","").replace("void ","")
        d[n][0]+=1; d[n][1]+=float(r["Counter_Value"])*2*1024/1e9
# totals are per PROCESS (train_probe: warm-up + timed steps), not per step
for n,(c,g) in sorted(d.items(), key=lambda kv:-kv[1][1])[:8]: print("%-40s launches %4d  fetch x2 %.1f GB total" % (n[:40], c, g))
PY
