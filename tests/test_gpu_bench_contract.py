"""GPU: bench.py prints ONE JSON line that carries every field of the driver's contract (metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config,
`roofline` and `cpu_baseline`), consistent with itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2',
                        '--train-steps', '2', '--train-bsz', '256'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['unit'] == 'segments/s' and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32'
    assert d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 640 * 4 / (d['ms_per_step'] * 4 / 1e3)) < 1e-3 * d['value']              # value = units / time
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 157.3
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 and 0.2 < rf['frac'] < 1.0
    assert abs(rf['achieved'] - rf['flops_per_launch_avg'] / (rf['ms_per_launch_avg'] * 1e-3) / 1e12) < 0.02 * rf['achieved']
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['unit'] == 'segments/s' and cb['value'] > 0 and cb['cores'] >= 1
    tr = d['train']
    assert tr['global_batch'] == 256 and tr['unit'] == 'steps/s' and tr['value'] > 0 and tr['scaling'] == 'strong'
