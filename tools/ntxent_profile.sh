# NT-Xent: parity tests, tools/ntxent_bench.py at the BSZ-5120 shapes and per-kernel durations from a rocprofv3 trace (run on the GPU box: gpurun -- bash tools/ntxent_profile.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ntxent.py -x -q -m gpu 2>&1 | tail -3
python tools/ntxent_bench.py 2560 8; python tools/ntxent_bench.py 640 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nt_prof -o nt -- python tools/ntxent_bench.py 2560 8 > /dev/null 2>&1
python - <<'PY'
import csv,glob
rows=list(csv.DictReader(open('gpurun_out/nt_prof/nt_kernel_trace.csv')))
import collections
d=collections.defaultdict(list)
for r in rows:
    if 'ntxent' in r['Kernel_Name']: d[(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(k, len(v), round(sum(v)/len(v),1))
PY
