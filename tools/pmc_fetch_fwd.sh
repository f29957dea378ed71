#!/bin/bash
# HBM-side traffic of the forward GEMM convs per launch for library variants selected by environment variables:
#   tools/pmc_fetch_fwd.sh "NAFP_XCDMAP=0" "NAFP_XCDMAP=1" ...
# One `rocprofv3 --pmc FETCH_SIZE` and one `--pmc WRITE_SIZE` pass per variant (they cannot share a pass); FETCH_SIZE is
# doubled (gfx950 counts a 16 B/lane stream at half, MI355X_MICROARCH.md, HBM).  Prints MB per launch by grid size.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
for v in "$@"; do
  tag=$(echo "$v" | tr ' =' '__')
  for c in FETCH_SIZE WRITE_SIZE; do
    O=gpurun_out/pmc_fwd/$tag/$c; rm -rf "$O"; mkdir -p "$O"
    env $v rocprofv3 --pmc $c --output-format csv -d "$O" -o p -- python bench.py --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e > /dev/null 2> "$O/err.txt"
  done
  python - "$v" "gpurun_out/pmc_fwd/$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
v, root = sys.argv[1], sys.argv[2]
tot = defaultdict(lambda: [0, 0.0, 0.0])
for c, scale in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
    for f in glob.glob(f'{root}/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c or 'conv_gemm' not in r['Kernel_Name'] and 'splitk_finish' not in r['Kernel_Name']:
                continue
            k = (r['Kernel_Name'].split('(')[0].replace('nafp::', '').replace('void ', ''), int(r['Grid_Size']))
            tot[k][0 if c == 'FETCH_SIZE' else 0] += (1 if c == 'FETCH_SIZE' else 0)
            tot[k][1 if c == 'FETCH_SIZE' else 2] += float(r['Counter_Value']) * 1024 * scale / 1e6
print(f'== {v}')
sf = sw = 0.0; n_fwd = 0
for (name, grid), (n, f, w) in sorted(tot.items(), key=lambda kv: -kv[0][1]):
    n = max(n, 1)
    print(f'  {name[:44]:44s} grid {grid:8d} launches {n:3d}  fetch {f / n:8.1f} MB  write {w / n:8.1f} MB')
    sf += f; sw += w
    if n_fwd == 0: n_fwd = n          # launches of the largest grid = conv1 = forwards in the run
print(f'  per forward (all GEMM-conv kernels): fetch {sf / max(n_fwd, 1):.1f} MB + write {sw / max(n_fwd, 1):.1f} MB = {(sf + sw) / max(n_fwd, 1) / 15:.1f} MB per conv launch (15 per forward)')
PY
done
