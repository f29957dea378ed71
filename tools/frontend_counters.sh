#!/bin/bash
# The front end (melspec_r16_kernel) under rocprofv3: kernel time from a trace, then two SQ counter passes (issue mix, LDS conflicts).
#   gpurun -- bash tools/frontend_counters.sh [ENV=..]      prints one line per quantity
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=/tmp/fe_$$; rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pipelined --no-train --no-e2e --fullscale-rows 0"
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- $CMD > /dev/null 2> $O/e0
env "$@" rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/s1 -o p -- $CMD > /dev/null 2> $O/e1
env "$@" rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/s2 -o p -- $CMD > /dev/null 2> $O/e2
python - $O "$*" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
st = [r for r in csv.DictReader(open(glob.glob(O + '/tr/**/*kernel_stats.csv', recursive=True)[0])) if 'melspec_r16' in r['Name']]
print(f"[{tag}] melspec_r16_kernel: {float(st[0]['AverageNs']) / 1e3:.1f} us avg over {st[0]['Calls']} launches")
for d in ('s1', 's2'):
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob(O + f'/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'melspec_r16' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value'])
                if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n += 1
    if not n: continue
    for k, v in sorted(acc.items()): print(f'  {k:28s} {v / n:14.0f} per launch')
    wc = acc['SQ_WAVE_CYCLES']
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
        if k in acc: print(f'  {k} / SQ_WAVE_CYCLES = {acc[k] / wc:.3f}')
    if 'SQ_LDS_BANK_CONFLICT' in acc and 'SQ_LDS_IDX_ACTIVE' in acc: print(f"  conflicts / LDS active = {acc['SQ_LDS_BANK_CONFLICT'] / acc['SQ_LDS_IDX_ACTIVE']:.3f}")
PY
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
a = {}
for d in ('s1', 's2'):
    for f in glob.glob(O + f'/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'melspec_r16' in r['Kernel_Name']:
                a.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
if 'SQ_LDS_BANK_CONFLICT' in a and 'SQ_LDS_IDX_ACTIVE' in a:
    print(f"  LDS bank-conflict cycles / LDS active cycles = {sum(a['SQ_LDS_BANK_CONFLICT']) / len(a['SQ_LDS_BANK_CONFLICT']) / (sum(a['SQ_LDS_IDX_ACTIVE']) / len(a['SQ_LDS_IDX_ACTIVE'])):.3f}")
PY
rm -rf $O
