"""GPU: nafp_augment_rows (window -> bg mix -> IR -> normalise, one launch) vs the oracle's restatement
of load_audio / bg_mix_batch / ir_aug_batch (oracle/augment.py) on the same windows and the same draws."""
import wave

import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu


def _write_wav(path, pcm, fs=8000):
    with wave.open(path, 'w') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs)
        w.writeframes(np.asarray(pcm).astype('<i2').tobytes())


def _oracle_rows(arena, rows, T):
    out = np.zeros((len(rows), T))
    for i, r in enumerate(rows):
        x = A.window(arena[r['ev_off']:r['ev_off'] + r['ev_valid']], 0, T)
        if r['mix']:
            nz = np.zeros(T)
            for key in ('nz', 'nz2'):
                if r[key + '_off'] >= 0:
                    nz += A.window(arena[r[key + '_off']:r[key + '_off'] + r[key + '_valid']], 0, T)
            x = A.bg_mix_rows(x[None], nz[None], [float(r['snr_db'])], [float(r['amp'])])[0]
        if r['ir_off'] >= 0:
            ir = arena[r['ir_off']:r['ir_off'] + r['ir_len']].astype(np.float64) / 2 ** 15
            x = A.ir_aug_rows(x[None], [ir])[0]
        out[i] = x
    return out


@pytest.fixture()
def corpus(tmp_path):
    rng = np.random.default_rng(7)
    t = np.arange(120000) / 8000.0

    def music(n):
        return (rng.integers(-2000, 2000, size=n) + 9000 * np.sin(2 * np.pi * rng.uniform(200, 3000) * t[:n])).astype(int)
    mk = lambda sub, arrs: [(_write_wav(str(tmp_path / f'{sub}{i}.wav'), a), str(tmp_path / f'{sub}{i}.wav'))[1] for i, a in enumerate(arrs)]
    ir = lambda n: (12000 * rng.normal(size=n) * np.exp(-np.arange(n) / 80.0)).astype(int)
    return {'ev': mk('ev', [music(120000), music(50000), np.zeros(30000, int), music(8000)]),
            'bg': mk('bg', [rng.integers(-6000, 6000, size=40000), np.zeros(9000, int)]),
            'ir': mk('ir', [ir(300), ir(2000), ir(601), np.zeros(50, int)]), 'sp': mk('sp', [rng.integers(-3000, 3000, size=20000)])}


@pytest.mark.parametrize('speech', [False, True])
def test_augmented_batch_matches_oracle(nafp, corpus, speech):
    from neural_audio_fp_amd.model.utils.dataloader_keras import genUnbalSequence
    ds = genUnbalSequence(corpus['ev'], bsz=48, n_anchor=8, shuffle=True, random_offset_anchor=True,
                          bg_mix_parameter=[True, corpus['bg'], (0, 10)], ir_mix_parameter=[True, corpus['ir']],
                          speech_mix_parameter=[speech, corpus['sp'], (3, 7)], seed=11)
    arena = ds.arena.host()
    worst = 0.0
    for idx in range(min(len(ds), 6)):
        rows = ds.plan(idx)
        Xa, Xp = ds[idx]                                          # same draws: plan() is a pure function of (seed, epoch, idx)
        assert Xa.shape == (8, 1, 8000) and Xp.shape == (40, 1, 8000) and Xa.dtype == torch.float32
        got = torch.cat([Xa, Xp])[:, 0].cpu().numpy()
        want = _oracle_rows(arena, rows, 8000)
        assert np.array_equal(got[:8], want[:8].astype(np.float32))          # anchors: exact int16 / 2^15
        worst = max(worst, np.abs(got - want).max())
        assert np.abs(got - want).max() < 2e-5
        # silent event / silent background / silent IR rows took the reference's special branches
    print('worst |augment - oracle|', worst)
    assert np.isfinite(got).all()


def test_special_rows(nafp, corpus):
    """silent event, silent background, silent impulse response, window past the end of a file."""
    from neural_audio_fp_amd import _lib
    from neural_audio_fp_amd.model.utils.dataloader_keras import genUnbalSequence
    ds = genUnbalSequence(corpus['ev'], bsz=4, n_anchor=2, bg_mix_parameter=[True, corpus['bg'], (0, 10)],
                          ir_mix_parameter=[True, corpus['ir']])
    arena = ds.arena.host()
    rows = np.zeros(5, dtype=_lib.AUG_ROW_DTYPE)
    rows['nz2_off'] = -1; rows['amp'] = [0.5, 0.25, 1.0, 0.7, 1.0]; rows['snr_db'] = 4.0; rows['mix'] = 1
    ev, bg, ir = ds.ev.start, ds.bg.start, ds.ir.start
    rows['ev_off'] = [ev[2], ev[0] + 123, ev[0] + 77, ev[1] + 46001, ev[3]]      # silent event | ... | 3999 valid samples | whole 1-s file
    rows['ev_valid'] = [8000, 8000, 8000, 3999, 8000]
    rows['nz_off'] = [bg[0], bg[1], bg[0] + 16000, bg[0] + 5, bg[1]]              # | silent bg | ...
    rows['nz_valid'] = [8000, 8000, 8000, 8000, 1000]
    rows['ir_off'] = [ir[0], ir[1], ir[3], ir[2], -1]                              # | | silent IR | 600 of 601 taps | none
    rows['ir_len'] = [300, 600, 50, 600, 0]
    got = ds.run(rows)[:, 0].cpu().numpy()
    want = _oracle_rows(arena, rows, 8000)
    assert np.abs(got - want).max() < 2e-5
    assert np.abs(got[2]).max() == 0.0                        # circular convolution with a silent IR is silence
    assert abs(np.abs(got[0]).max() - 1.0) < 1e-6 and abs(np.abs(got[4]).max() - 1.0) < 1e-6


def test_dataset2wav_writes_one_augmented_clip_per_source(nafp, cfg, tmp_path):
    import copy, importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('dataset2wav', os.path.join(root, 'tools', 'dataset2wav.py'))
    d2w = importlib.util.module_from_spec(spec); spec.loader.exec_module(d2w)
    rng = np.random.default_rng(5)
    base = str(tmp_path) + '/'
    t = np.arange(48000) / 8000.0
    for i in range(2):
        os.makedirs(base + 'music/q/db/x', exist_ok=True)
        _write_wav(base + f'music/q/db/x/{i}.wav', (rng.integers(-500, 500, size=48000) + 8000 * np.sin(2 * np.pi * (400 + 300 * i) * t)).astype(int))
    os.makedirs(base + 'bg/ts'); os.makedirs(base + 'ir/ts')
    _write_wav(base + 'bg/ts/0.wav', rng.integers(-3000, 3000, size=30000))
    _write_wav(base + 'ir/ts/0.wav', (15000 * np.exp(-np.arange(300) / 25.0) * rng.normal(size=300)).astype(int))
    c = copy.deepcopy(cfg)
    c['DIR'].update({'SOURCE_ROOT_DIR': base + 'music/', 'BG_ROOT_DIR': base + 'bg/', 'IR_ROOT_DIR': base + 'ir/'})
    files = d2w.synthesize(c, 'q/db', base + 'out', snr=(10, 10), interval=1, clip_sec=6)
    assert [f.split('/')[-2:] for f in files] == [['x', '0.wav'], ['x', '1.wav']]
    for f in files:
        with wave.open(f) as w:
            assert (w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()) == (8000, 1, 2, 48000)
            x = np.frombuffer(w.readframes(48000), dtype='<i2').astype(np.float64) / 32767
        pieces = np.abs(x.reshape(6, 8000)).max(1)
        assert np.all(pieces > 0.99) and np.all(pieces <= 1.0)     # every 1-s piece was max-normalised after the IR
