#!/bin/bash
# HBM-side traffic per launch (PMC FETCH_SIZE x 2, gfx950 correction) of the probe's kernels, tap-major and channel-major K order
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
for b in x6_gemm_probe x6_gemm_probe_k1; do
  O=gpurun_out/probe_pmc/$b; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O" -o p -- tools/probes/$b > /dev/null 2> "$O/err.txt"
  python - "$b" "$O" <<'PY'
import csv, glob, sys
from collections import defaultdict
b, root = sys.argv[1], sys.argv[2]
tot = defaultdict(lambda: [0, 0.0])
for f in glob.glob(f'{root}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE': continue
        k = r['Kernel_Name'][:70]
        tot[k][0] += 1; tot[k][1] += float(r['Counter_Value']) * 1024 * 2.0 / 1e6
print('==', b)
for k, (n, f) in tot.items(): print(f'  {k:70s} launches {n:3d} fetch {f / n:8.1f} MB')
PY
done
