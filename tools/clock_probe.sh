cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3; do
rm -rf gpurun_out/clk_$a
NAFP_ABL=$a rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/clk_$a -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv
from collections import defaultdict
d=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open('gpurun_out/clk_$a/p_counter_collection.csv')):
    if 'conv_gemm' in r['Kernel_Name'] and int(r['Grid_Size'])==2621440:
        d[r['Counter_Name']]['v'].append(float(r['Counter_Value'])); d[r['Counter_Name']]['t'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
g=d['GRBM_GUI_ACTIVE']; m=d['SQ_VALU_MFMA_BUSY_CYCLES']
cyc=sum(g['v'])/len(g['v'])/8; dur=sum(g['t'])/len(g['t'])
print('abl $a conv1: dur us %.1f clock GHz %.3f mfma busy %.1f%%' % (dur/1e3, cyc/dur, sum(m['v'])/len(m['v'])/(1024*cyc)*100))
PY
done
