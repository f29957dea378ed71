set -x
L=neural-audio-fp_amd/_abl/libnafp_nopk.so
echo "== determinism, default library"; OPT=2 python tools/x6_determinism_check.py 2>&1 | grep "results differ"
echo "== determinism, no packed f32"; NAFP_LIB=$L OPT=2 python tools/x6_determinism_check.py 2>&1 | grep "results differ"
echo "== check2 default"; OPT=2 python tools/x6_determinism_check2.py 2>&1 | tail -4
echo "== check2 nopk"; NAFP_LIB=$L OPT=2 python tools/x6_determinism_check2.py 2>&1 | tail -4
for r in 1 2; do
echo "== bench default $r"; python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-train --no-e2e > gpurun_out/ab_default_$r.json 2>/dev/null
echo "== bench nopk $r"; NAFP_LIB=$L python bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-train --no-e2e > gpurun_out/ab_nopk_$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ('default_1','nopk_1','default_2','nopk_2'):
    d=json.loads(open(f'gpurun_out/ab_{n}.json').read().strip().splitlines()[-1])
    x=d['bf16x6_f32_equivalent_experimental']
    print(n, 'f32', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'stages', d.get('stage_ms_per_step'), 'pipelined', (d.get('pipelined') or {}).get('value'), 'x6', x['value'], x['ms_per_step'])
PY
