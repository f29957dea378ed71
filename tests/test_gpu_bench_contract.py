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
                        '--repeats', '3', '--train-steps', '2', '--train-bsz', '256'], cwd=ROOT, capture_output=True, text=True,
                       timeout=1500)
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
    sp = d['spread']
    assert d['repeats'] == 3 and len(sp['ms_per_step_all']) == 3 and sp['min'] <= d['ms_per_step'] <= sp['max']
    assert d['ms_per_step'] == sorted(sp['ms_per_step_all'])[1]                                       # the median region
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
    assert max(r['cores'] for r in cb['process_sweep']) > cb['host_cores'] - cb['processes'], cb      # the filled host was timed
    assert cb['value'] == max(r['value'] for r in cb['process_sweep'])
    tr = d['train']
    assert tr['global_batch'] == 256 and tr['unit'] == 'steps/s' and tr['value'] > 0 and tr['scaling'] == 'strong'
    t2 = d['train_1280']
    assert t2['optimizer'] == 'Adam' and t2['global_batch'] == 256 and t2['value'] > 0
    # the second half of the metric is timed like the first: >= 3 regions, the median reported, all of them listed
    for obj in (tr, t2, d['train_rank640']):
        assert 'error' not in obj, obj
        sp = obj['spread']
        assert obj['repeats'] >= 3 and len(sp['ms_per_step_all']) == obj['repeats'] and sp['min'] <= obj['ms_per_step'] <= sp['max']
        assert obj['ms_per_step'] == sorted(sp['ms_per_step_all'])[len(sp['ms_per_step_all']) // 2]
    r6 = d['train_rank640']
    assert r6['global_batch'] == 640 and r6['collectives']['backend'] == 'nccl' and r6['collectives']['world_size'] == 1
    assert abs(r6['exposed_comm_ms'] - (r6['ms_per_step'] - r6['no_process_group_ms_per_step'])) < 2e-3
    # the exact-split object: its own roofline against the bf16 matrix peak (executed FLOPs = 6 x algorithmic), never the fp32 peak
    x6 = d['bf16x6_f32_equivalent_experimental']
    r6x = x6['roofline']
    assert r6x['bound'] == 'mfma' and r6x['peak'] == 2500.0 and 0.05 < r6x['frac'] < 1.0
    assert abs(r6x['frac'] - r6x['achieved'] / r6x['peak']) < 1e-3
    assert abs(r6x['executed_flops_per_step'] - 6.0 * r6x['algorithmic_f32_flops_per_step']) < 1.0
    assert abs(r6x['achieved'] - r6x['executed_flops_per_step'] / (r6x['gemm_span_ms_per_step'] * 1e-3) / 1e12) < 0.02 * r6x['achieved']
    assert abs(r6x['f32_equivalent_TFLOP/s'] * 6.0 - r6x['achieved']) < 0.02 * r6x['achieved'] and len(r6x['per_conv_ms']) == 17
    assert x6['min_cosine_vs_f32_path'] > 1 - 1e-6 and x6['value'] > 0
    tx = d['train_x6_experimental']
    assert 'error' not in tx, tx
    assert tx['global_batch'] == 256 and tx['value'] > 0 and tx['vs_f32_step'] > 0 and tx['rank640']['ms_per_step'] > 0 and 'exact 3-way bf16 split' in tx['dtype']
    assert x6['pipelined']['streams'] == 4 and x6['pipelined']['value'] > 0 and x6['pipelined']['bit_identical_to_single_stream'] is True
    e = d['e2e_generate']
    assert e['clips_100']['segments'] == 5900 and e['clips_600']['segments'] == 35400
    assert e['clips_100']['value'] > 0 and e['clips_600']['value'] > 0 and e['clips_600']['ingest_only_segments_per_s'] > 0


@pytest.mark.parametrize('form', ['bare', 'torchrun'])
def test_bench_two_ranks_on_one_gpu_prints_train_with_collective_timings(form):
    """The two N > 1 forms: 'bare' = `python bench.py --gpus 2 ...` with no WORLD_SIZE (what the driver's BENCH/SCALE
    command line looks like): bench.py starts the ranks itself as a child `torch.distributed.run` and relays rank 0's
    line; 'torchrun' = already under `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`.
    Here 2 ranks share cuda:0 over gloo (NAFP_BENCH_BACKEND / NAFP_BENCH_ONE_GPU: RCCL refuses two ranks per GPU):
    ONE JSON line from rank 0, whole-job value, and the `train` object with per-collective timings."""
    env = dict(os.environ, NAFP_BENCH_BACKEND='gloo', NAFP_BENCH_ONE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    port = 29800 + (os.getpid() % 100)
    launcher = [] if form == 'bare' else ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                                          '--master-addr', '127.0.0.1', '--master-port', str(port)]
    r = subprocess.run([sys.executable] + launcher + [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                                                      '--repeats', '2', '--train-steps', '2', '--train-bsz', '128'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and 'cpu_baseline' not in d
    assert abs(d['value'] - 2 * 640 * 3 / (d['ms_per_step'] * 3 / 1e3)) < 1e-3 * d['value']           # whole-job throughput
    tr = d['train']
    assert tr['n_gpus'] == 2 and tr['global_batch'] == 128 and tr['per_gpu_batch'] == 64 and tr['scaling'] == 'strong'
    c = tr['collectives']
    assert c['backend'] == 'gloo' and c['world_size'] == 2
    assert c['all_gather(emb)_ms'] > 0 and c['reduce_scatter(d emb)_ms'] > 0
    assert len(c['all_reduce(grad pieces)_ms']) == 4 and all(x > 0 for x in c['all_reduce(grad pieces)_ms'])
    assert abs(sum(c['grad_piece_MB']) - 67.76) < 0.1
    # what the collectives leave exposed: the step minus rank 0's compute-only step of the same run
    assert abs(tr['exposed_comm_ms'] - (tr['ms_per_step'] - tr['no_process_group_ms_per_step'])) < 2e-3
