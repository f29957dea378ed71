#!/bin/bash
# MFMA-busy share and sustained clock of the train step's kernels (SQ counters, one pass).  tools/pmc_train_sq.sh [B] [adam|lamb]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
B=${1:-5120}; OPT=${2:-lamb}
O=/tmp/pmc_tr_$$; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O -o p -- python tools/train_probe.py $B $OPT 2 > $O/probe.txt 2> $O/err.txt
python - "$O" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('nafp::', '').replace('void ', '').split('(')[0][:34]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        agg[k]['_dur'] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[k] += 1
print(f'{"kernel":36s} {"n":>4s} {"ms":>8s} {"GHz":>5s} {"MFMA busy %":>11s} {"waves/CU":>8s} {"WAIT_ANY %":>10s} {"WAIT_INST %":>11s}')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['_dur'])[:14]:
    cyc = a['GRBM_GUI_ACTIVE'] / 8.0
    print(f'{k:36s} {cnt[k]:4d} {a["_dur"] / 1e6:8.2f} {cyc / a["_dur"]:5.2f} {a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) * 100:11.1f} '
          f'{a["SQ_WAVE_CYCLES"] * 4 / (cyc * 256):8.1f} {a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1) * 100:10.1f} {a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1) * 100:11.1f}')
PY
rm -rf $O
