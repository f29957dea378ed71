"""CPU: host logic of the generate driver (row sharding, memmap layout, checkpoint
discovery, config surface) and its 2-process gloo run.  The embedding function is a
test stub here -- the product path has no CPU fallback (tests/test_abi.py)."""
import os
import wave

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_wav(path, pcm, fs=8000):
    with wave.open(path, 'w') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs)
        w.writeframes(pcm.astype('<i2').tobytes())


def _stub_embed(chunk, group):
    """deterministic per-segment 'fingerprint' (test stub only)."""
    x = chunk[:, 0, :].astype(np.float64)
    return np.stack([x[:, :128].sum(1) + k for k in range(128)], 1).astype(np.float32)


def _make_files(d, n_files=5, seed=0):
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(n_files):
        p = os.path.join(d, f'{i:03d}.wav')
        _write_wav(p, rng.integers(-8192, 8192, size=int(rng.integers(6000, 40000))))
        paths.append(p)
    return paths


def test_shard_rows_partitions_on_group_boundaries():
    from neural_audio_fp_amd.model.generate import shard_rows
    for n in (1, 124, 125, 126, 5900, 7777):
        for g in (1, 125, 640):
            for world in (1, 2, 3, 8):
                rs = [shard_rows(n, g, r, world) for r in range(world)]
                assert rs[0][0] == 0 and rs[-1][1] == n
                for (a0, a1), (b0, b1) in zip(rs, rs[1:]):
                    assert a1 == b0 and a0 <= a1
                assert all(a0 % g == 0 for a0, _ in rs)          # group maxima are rank-independent


def test_write_fingerprints_row_order_and_ragged_tail(tmp_path):
    from neural_audio_fp_amd.model.generate import write_fingerprints
    from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource
    src = SegmentSource(_make_files(str(tmp_path)), bsz=4)
    arr = np.zeros((src.n_samples, 128), np.float32)
    write_fingerprints(src, _stub_embed, arr, group=4, launch_rows=12)
    want = np.concatenate([_stub_embed(src[i][0], 4) for i in range(len(src))])
    assert np.array_equal(arr, want)


def _worker(rank, world, port, wav_dir, out_path):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    from neural_audio_fp_amd.model.generate import write_fingerprints
    from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource
    src = SegmentSource(sorted(os.path.join(wav_dir, f) for f in os.listdir(wav_dir)), bsz=3)
    shape = (src.n_samples, 128)
    if rank == 0:
        arr = np.memmap(out_path, dtype='float32', mode='w+', shape=shape)
    dist.barrier()
    if rank != 0:
        arr = np.memmap(out_path, dtype='float32', mode='r+', shape=shape)
    write_fingerprints(src, _stub_embed, arr, group=3, rank=rank, world=world, launch_rows=6)
    arr.flush()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_generate_writes_one_memmap(tmp_path):
    import torch.multiprocessing as mp
    from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource
    wav_dir = tmp_path / 'wav'; wav_dir.mkdir()
    paths = _make_files(str(wav_dir), n_files=6, seed=3)
    out = str(tmp_path / 'db.mm')
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, str(wav_dir), out), nprocs=2, join=True)
    src = SegmentSource(sorted(paths), bsz=3)
    got = np.memmap(out, dtype='float32', mode='r', shape=(src.n_samples, 128))
    want = np.concatenate([_stub_embed(src[i][0], 3) for i in range(len(src))])
    assert np.array_equal(np.asarray(got), want)


def test_checkpoint_discovery_and_errors(tmp_path):
    from neural_audio_fp_amd.model import generate as g

    class Fake:
        def __init__(self): self.sd = None
        def load_state_dict(self, sd): self.sd = sd
        def state_dict(self): return {'w': torch.ones(2)}
    root = str(tmp_path) + '/checkpoint/'
    with pytest.raises(FileNotFoundError):
        g.load_checkpoint(root, 'exp', None, Fake())
    g.save_checkpoint(root, 'exp', 3, Fake()); g.save_checkpoint(root, 'exp', 12, Fake())
    m = Fake()
    assert g.load_checkpoint(root, 'exp', None, m) == 12 and torch.equal(m.sd['w'], torch.ones(2))
    assert g.load_checkpoint(root, 'exp', '3', m) == 3
    with pytest.raises(FileNotFoundError):
        g.load_checkpoint(root, 'exp', 5, m)
    # retention like tf.train.CheckpointManager(max_to_keep=3, keep_checkpoint_every_n_hours=N)
    for i in (1, 2, 4, 5):
        g.save_checkpoint(root, 'keep', i, Fake())
    now = os.path.getmtime(root + 'keep/ckpt-5.pt')
    for i, age_h in ((1, 30), (2, 29.5), (4, 3)):
        os.utime(root + f'keep/ckpt-{i}.pt', (now - 3600 * age_h, now - 3600 * age_h))
    g.save_checkpoint(root, 'keep', 6, Fake()); g.save_checkpoint(root, 'keep', 7, Fake())
    removed = g.prune_checkpoints(root, 'keep', 3, keep_every_n_hours=1)
    assert [os.path.basename(f) for f in removed] == ['ckpt-2.pt']          # 1 preserved (first), 2 too close to it, 4 is 26 h later
    assert sorted(os.listdir(root + 'keep')) == ['ckpt-1.pt', 'ckpt-4.pt', 'ckpt-5.pt', 'ckpt-6.pt', 'ckpt-7.pt']
    assert len(g.prune_checkpoints(root, 'keep', 3, keep_every_n_hours=None)) == 2
    # weights converted from a TensorFlow checkpoint of the reference arrive as ckpt-N.npz (tools/convert_tf_checkpoint.py)
    np.savez(root + 'exp/ckpt-20.npz', w=np.full(2, 7.0, np.float32))
    assert g.load_checkpoint(root, 'exp', None, m) == 20 and np.array_equal(m.sd['w'], np.full(2, 7.0, np.float32))
    assert g.load_checkpoint(root, 'exp', 12, m) == 12 and torch.equal(m.sd['w'], torch.ones(2))


def test_config_surface_matches_reference_keys():
    keys = {'DIR': ['SOURCE_ROOT_DIR', 'BG_ROOT_DIR', 'IR_ROOT_DIR', 'SPEECH_ROOT_DIR', 'OUTPUT_ROOT_DIR', 'LOG_ROOT_DIR'],
            'DATA_SEL': ['TRAIN', 'TEST_DUMMY_DB', 'TEST_QUERY_DB', 'REDUCE_ITEMS_P'],
            'MODEL': ['FEAT', 'FS', 'DUR', 'HOP', 'STFT_WIN', 'STFT_HOP', 'F_MIN', 'F_MAX', 'N_MELS', 'EMB_SZ', 'BN'],
            'BSZ': ['TR_BATCH_SZ', 'TR_N_ANCHOR', 'VAL_BATCH_SZ', 'VAL_N_ANCHOR', 'TS_BATCH_SZ'],
            'TRAIN': ['MAX_EPOCH', 'OPTIMIZER', 'LR', 'LR_SCHEDULE', 'CHECKPOINT_KEEP_N_HOUR', 'TENSORBOARD', 'SAVE_IMG',
                      'MINI_TEST_IN_TRAIN'],
            'LOSS': ['LOSS_MODE', 'TAU', 'MARGIN'], 'SPEC_AUG': ['SPECAUG_CHAIN', 'SPECAUG_PROBS', 'SPECAUG_N_HOLES',
                                                               'SPECAUG_HOLE_FILL'],
            'DEVICE': ['CPU_N_WORKERS', 'CPU_MAX_QUEUE']}
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
    for sec, ks in keys.items():
        assert list(cfg[sec].keys()) == ks, sec
    assert cfg['BSZ']['TS_BATCH_SZ'] == 125 and cfg['LOSS']['TAU'] == 0.05 and cfg['MODEL']['N_MELS'] == 256
    lamb = yaml.safe_load(open(os.path.join(ROOT, 'config', '640_lamb.yaml')))
    assert lamb['BSZ']['TR_BATCH_SZ'] == 640 and lamb['TRAIN']['OPTIMIZER'] == 'LAMB'


def test_unknown_feat_raises_and_every_norm_string_is_a_model(nafp, cfg):
    """melspectrogram.py:114-131 raises on an unknown FEAT; MODEL.BN never raises in the reference: 'layer_norm1d', 'layer_norm2d',
    and any other string is BatchNormalization (nnfp.py:63-71).  (The models themselves: tests/test_gpu_norm_alternates.py.)"""
    import copy
    from neural_audio_fp_amd.model.fp import nnfp
    c = copy.deepcopy(cfg); c['MODEL']['FEAT'] = 'nope'
    with pytest.raises(NotImplementedError):
        nafp.get_melspec_layer(c)
    assert [nnfp.norm_kind(s) for s in ('layer_norm2d', 'layer_norm1d', 'batch_norm', 'bn', '')] == [0, 1, 2, 2, 2]
    assert len(nnfp.tensor_names('layer_norm1d')) == 68 and len(nnfp.tensor_names('batch_norm')) == 100
    assert nnfp.tensor_names('batch_norm')[:68] == nnfp.tensor_names() and nnfp.tensor_names('batch_norm')[69] == 'front_conv.0.BN_1x3.moving_variance'


def test_bench_flop_model_matches_survey():
    import bench
    m = bench.conv_effective_macs()
    assert sum(m) == 278888448 and m[0] == 1540096 and 2 * (sum(m) + 36864) == 557850624


def _scatter_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from neural_audio_fp_amd.model import trainer as T
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_a, d = 3, 8
    g = torch.Generator().manual_seed(100 + rank)
    d_a, d_b = torch.randn((world * n_a, d), generator=g), torch.randn((world * n_a, d), generator=g)
    loss, d_emb = T.scatter_embedding_gradients(dist, torch.tensor(0.5 + rank), d_a, d_b, n_a)
    torch.save({'loss': float(loss), 'd_emb': d_emb.clone(), 'd_a': d_a, 'd_b': d_b}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_reduce_scatter_of_embedding_gradients(tmp_path, world):
    """The exchange step of the data-parallel train step (NTxent_loss_tpu.py:57-87 backward): each rank gets
    the rank-summed gradient of ITS rows and the global loss from one reduce-scatter."""
    import torch.multiprocessing as mp
    port = 29100 + (os.getpid() % 400) + world
    mp.spawn(_scatter_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f'r{k}.pt', weights_only=True) for k in range(world)]
    tot_a, tot_b = sum(r['d_a'] for r in rs), sum(r['d_b'] for r in rs)
    for k, r in enumerate(rs):
        assert abs(r['loss'] - sum(0.5 + j for j in range(world))) < 1e-6
        want = torch.cat([tot_a[3 * k:3 * k + 3], tot_b[3 * k:3 * k + 3]])
        assert torch.allclose(r['d_emb'], want, atol=1e-6)
