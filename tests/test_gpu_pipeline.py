"""GPU: the reference's whole workflow through the command line -- `run.py train`, `run.py generate`,
`run.py evaluate` (run.py:13-162 of the reference) -- on a synthetic dataset tree in the reference's
directory layout."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wav(path, pcm):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, 'w') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.writeframes(np.clip(pcm, -32768, 32767).astype('<i2').tobytes())


def _music(rng, n):
    t = np.arange(n) / 8000.0
    x = rng.integers(-1200, 1200, size=n).astype(np.float64)
    for f in rng.uniform(250, 3800, size=4):
        x += 5000 * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) * (0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(0.2, 2.0) * t))
    return x


def test_train_generate_evaluate_cli(tmp_path):
    rng = np.random.default_rng(2026)
    ds = str(tmp_path / 'dataset') + '/'
    for i in range(6):
        _wav(f'{ds}music/train-10k-30s/a/{i}.wav', _music(rng, 80000))
    for i in range(2):
        _wav(f'{ds}aug/bg/tr/{i}.wav', rng.integers(-4000, 4000, size=40000))
        _wav(f'{ds}aug/ir/tr/{i}.wav', 12000 * rng.normal(size=800) * np.exp(-np.arange(800) / 50.0))
    for i in range(5):
        _wav(f'{ds}music/test-dummy-db-100k-full/x/{i}.wav', _music(rng, 160000))
    for i in range(4):
        x = _music(rng, 160000)
        _wav(f'{ds}music/test-query-db-500-30s/db/x/{i}.wav', x)
        _wav(f'{ds}music/test-query-db-500-30s/query/x/{i}.wav', x + rng.normal(size=len(x)) * 1500)
    work = tmp_path / 'work'
    (work / 'config').mkdir(parents=True)
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
    cfg['DIR'].update({'SOURCE_ROOT_DIR': ds + 'music/', 'BG_ROOT_DIR': ds + 'aug/bg/', 'IR_ROOT_DIR': ds + 'aug/ir/',
                       'OUTPUT_ROOT_DIR': str(work) + '/logs/emb/', 'LOG_ROOT_DIR': str(work) + '/logs/'})
    cfg['BSZ'].update({'TR_BATCH_SZ': 32, 'TR_N_ANCHOR': 16})
    yaml.safe_dump(cfg, open(work / 'config' / 'tiny.yaml', 'w'))
    env = dict(os.environ, PYTHONPATH=ROOT)

    def run(*args, inp=None):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py')] + list(args), cwd=work, env=env, input=inp,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r.stdout

    out = run('train', 'EXP', '-c', 'tiny', '--max_epoch', '2')
    assert 'epoch 2: tr_loss' in out and (work / 'logs' / 'checkpoint' / 'EXP' / 'ckpt-2.pt').exists()
    out = run('generate', 'EXP', '-c', 'tiny')                    # latest checkpoint, default sources
    emb = work / 'logs' / 'emb' / 'EXP' / '2'
    n_db = 4 * 39
    assert tuple(np.load(emb / 'db_shape.npy')) == (n_db, 128) and tuple(np.load(emb / 'query_shape.npy')) == (n_db, 128)
    assert tuple(np.load(emb / 'dummy_db_shape.npy')) == (5 * 39, 128)
    ids = np.arange(0, n_db - 3)
    np.save(work / 'ids.npy', ids)
    out = run('evaluate', 'EXP', '2', '-c', 'tiny', '-t', str(work / 'ids.npy'), '--test_seq_len', '1 3')
    raw = np.load(emb / 'raw_score.npy')
    assert raw.shape == (len(ids), 8) and np.array_equal(np.load(emb / 'test_ids.npy'), ids)
    import json
    used = json.load(open(emb / 'index_used.json'))              # default -i ivfpq is served by the exact search: recorded next to the scores
    assert used['index_type_requested'].lower() == 'ivfpq' and used['substituted'] is True and used['index_type_used'].startswith('L2')
    top1_exact = raw[:, :2].mean(0)
    print(out[-600:], top1_exact)
    assert top1_exact[1] > 0.8 and top1_exact[1] >= top1_exact[0] - 0.05       # noisy copies are found; longer queries do not hurt
    # generate refuses to overwrite dummy_db.mm without confirmation (generate.py:55-58 of the reference)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'generate', 'EXP', '-c', 'tiny'], cwd=work, env=env, input='n\n',
                       capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 or 'overwrite' in (r.stdout + r.stderr).lower()
