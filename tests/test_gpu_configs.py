"""GPU: every BASELINE.json configuration exercised AT SIZE.

  configs[0]  config/default.yaml generate on 100 x 30-s clips, TS_BATCH_SZ = 125 -> 5,900 rows, three whole
              125-groups (first, a middle one, the ragged last) against the oracle end to end
  configs[2]  one whole contrastive train step at BSZ 1280, Adam
  configs[3]  one whole contrastive train step at BSZ 5120, LAMB (the per-rank shape 320 x 2560 of the sharded loss
              is in test_gpu_ntxent.py)
  configs[4]  full-scale generate: (i) ONE rank's full share of the 100 M-row / 8-rank job -- 12,500,000 rows through the
              product's writer into one 6.4 GB dummy_db.mm, whole 125-groups against the ORACLE, load_memmap_data, exact-search
              self hits; (ii) 2 ranks (gloo, one GPU) each writing its slice of ONE >= 1 M-row dummy_db.mm
(configs[1], generate at BSZ 640, is test_gpu_generate.py / test_gpu_parity_forward.py / bench.py.)

For the train steps: loss == the oracle's NT-Xent evaluated on the HIP embeddings; updated variables == the oracle's
Adam / LAMB rule applied to the HIP gradients; the gradients themselves against float64 autograd for 16 samples spread
over the batch (LayerNorm is per sample, so the parameter gradient is a sum over samples and a backward pass with
dL/d(emb) zeroed outside those 16 samples is their exact share of the step).
"""
import copy
import os
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

from oracle import melspec as o_mel, ntxent as o_nt, optim as o_opt, segments as o_seg, torch_ref
import _inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_fingerprints(x, w):
    """float64 front end (oracle/melspec.py) + the encoder on torch-CPU float64 kernels (oracle/torch_ref.py, held to
    the numpy restatement by tests/test_oracle_nnfp.py): fast enough for whole 125-groups."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    tf = torch_ref.TorchFingerprinter(w, dtype=torch.float64)
    with torch.no_grad():
        return tf(torch.from_numpy(o_mel.melspec_layer(x)).double()).numpy()


# ------------------------------------------------------------------------------------------------------------------
# configs[0]
# ------------------------------------------------------------------------------------------------------------------
def _write_clip(path, k, seconds=30, fs=8000):
    """clip k = seeded noise (uniform int16 in +-8192) + 3 sinusoids in 300..3900 Hz."""
    rng = np.random.default_rng(1000 + k)
    n = seconds * fs
    t = np.arange(n) / fs
    pcm = rng.integers(-8192, 8193, size=n).astype(np.float64)
    for f in rng.uniform(300, 3900, size=3):
        pcm += 4000 * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    with wave.open(path, 'w') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs)
        w.writeframes(np.clip(pcm, -32768, 32767).astype('<i2').tobytes())


def test_config0_generate_100_clips_ts_batch_125(nafp, cfg, tmp_path, arith):
    import yaml
    from neural_audio_fp_amd.model import generate as g
    src = tmp_path / 'clips'; src.mkdir()
    for k in range(100):
        _write_clip(str(src / f'{k:03d}.wav'), k)
    c = copy.deepcopy(cfg)
    assert c['BSZ']['TS_BATCH_SZ'] == 125
    c['DIR']['LOG_ROOT_DIR'] = str(tmp_path) + '/logs/'
    c['DIR']['OUTPUT_ROOT_DIR'] = str(tmp_path) + '/logs/emb/'
    (tmp_path / 'config').mkdir()
    with open(tmp_path / 'config' / 'c0.yaml', 'w') as f:
        yaml.safe_dump(c, f)
    w = _inputs.weights(seed=17)
    m_fp = nafp.get_fingerprinter(c)
    m_fp.set_weights(_inputs.weight_list(w))
    g.save_checkpoint(c['DIR']['LOG_ROOT_DIR'] + 'checkpoint/', 'c0', 1, m_fp)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'generate', 'c0', '1', '-c', 'c0', '-s', str(src)],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out_dir = c['DIR']['OUTPUT_ROOT_DIR'] + '/c0/1/'
    shape = np.load(out_dir + 'custom_source_shape.npy')
    assert shape.dtype == np.int64 and tuple(shape) == (5900, 128)              # 100 clips x 59 segments
    got = np.asarray(np.memmap(out_dir + 'custom_source.mm', dtype='float32', mode='r', shape=(5900, 128)))
    assert np.isfinite(got).all() and np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    paths = sorted(str(p) for p in src.glob('*.wav'))
    segs = o_seg.enumerate_segments(paths)
    assert len(segs) == 5900
    n_groups = -(-5900 // 125)                                                  # 47 full groups + one of 25
    for gi in (0, 23, n_groups - 1):
        rows = segs[gi * 125:(gi + 1) * 125]
        assert len(rows) == (125 if gi < n_groups - 1 else 5900 - 125 * (n_groups - 1))
        x = np.expand_dims(np.stack([o_seg.load_segment(fn, s) for fn, s in rows]), 1).astype(np.float32)
        want = _oracle_fingerprints(x, w)                                       # one group = one m_pre batch (melspectrogram.py:108)
        blk = got[gi * 125:gi * 125 + len(rows)]
        assert (1 - (blk * want).sum(1)).max() < 1e-5, gi                       # contract 1e-3
        assert np.abs(blk - want).max() < 1e-4, gi
    # ---- bit-reproducible generate ----
    # (a) the same command again: byte-identical custom_source.mm (the per-sample LayerNorm statistics are accumulated as
    #     64-bit fixed-point integers -- order-free -- and every other sum of the forward pass is taken in a fixed order)
    import hashlib
    first = open(out_dir + 'custom_source.mm', 'rb').read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'run.py'), 'generate', 'c0', '1', '-c', 'c0', '-s', str(src)],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    again = open(out_dir + 'custom_source.mm', 'rb').read()
    assert first == again, 'two runs of run.py generate differ'
    # (b) other launch sizes through the product's writer, each twice: 125 (one group per launch), 750, 1000 (8 groups; the
    #     last launch ragged): run-to-run identical bytes at every size.
    from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource
    m_pre = nafp.get_melspec_layer(c)
    source = SegmentSource(paths, bsz=125)
    by_size = {}
    for launch_rows in (125, 750, 1000):
        runs = []
        for rep in range(2):
            arr = np.zeros((5900, 128), np.float32)
            g.write_fingerprints(source, g.StreamedEmbedder(m_pre, m_fp), arr, 125, launch_rows=launch_rows)
            runs.append(arr)
        assert runs[0].tobytes() == runs[1].tobytes(), launch_rows
        by_size[launch_rows] = runs[0]
    assert by_size[750].tobytes() == first                                        # run.py's own launch size (6 groups: LAUNCH_SEGMENTS 640 rounded up)
    # [r5] ... and ACROSS launch sizes too: the inference forward plans every launch (tile shape, split-K factor) as at the
    # reference size of 640 segments whatever it holds (csrc/conv.hip fwd_plan_b()), the LayerNorm statistics are order-free
    # integers, so a segment's fingerprint no longer depends on what shares its launch
    for launch_rows in (125, 1000):
        assert by_size[launch_rows].tobytes() == first, launch_rows
    # (c) the committed hash of this output (seeded weights, seeded clips, this kernel generation): tests/golden/
    #     hip_generate_sha256.json.  A change of any forward kernel's summation order legitimately changes it -- then
    #     re-record it from gpurun_out/hip_generate_sha256.json of a run and say so in the commit.
    import json
    sha = hashlib.sha256(first).hexdigest()
    key = 'config0_custom_source_mm_sha256' + ('' if arith == 'f32' else '_x6')      # one record per arithmetic
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        rec_path = os.path.join(ROOT, 'gpurun_out', 'hip_generate_sha256.json')
        rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
        rec[key] = sha
        json.dump(rec, open(rec_path, 'w'))
    except (OSError, ValueError):
        pass
    want_sha = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'hip_generate_sha256.json'))).get(key)
    if want_sha:
        assert sha == want_sha, f'custom_source.mm of config 0 hashes to {sha}; tests/golden/hip_generate_sha256.json holds {want_sha}'


# ------------------------------------------------------------------------------------------------------------------
# configs[2], configs[3]
# ------------------------------------------------------------------------------------------------------------------
def _synthetic_pairs(n, seed):
    """Xa = seeded noise segments, Xp = Xa + noise at 5 dB SNR, resident on the device."""
    g = torch.Generator(device='cuda').manual_seed(seed)
    xa = 0.1 * torch.randn((n, 1, 8000), generator=g, device='cuda')
    xp = xa + 0.1 * 10.0 ** (-5.0 / 20.0) * torch.randn((n, 1, 8000), generator=g, device='cuda')
    return xa, xp


def _oracle_update(which, w0, g, lr, var_len):
    if which == 'adam':
        return o_opt.adam_step(w0, g, np.zeros_like(w0), np.zeros_like(w0), lr, 1)[0]
    if var_len == w0.size:
        return o_opt.lamb_step(w0, g, np.zeros_like(w0), np.zeros_like(w0), lr, 1)[0]
    wf, gf = w0.reshape(-1, var_len), g.reshape(-1, var_len)                      # stacked divide-and-encode variables
    return np.stack([o_opt.lamb_step(wf[k], gf[k], np.zeros(var_len), np.zeros(var_len), lr, 1)[0]
                     for k in range(wf.shape[0])]).reshape(w0.shape)


@pytest.mark.parametrize('bsz,which', [(1280, 'adam'), (5120, 'lamb')])
def test_whole_train_step_at_baseline_batch(nafp, cfg, bsz, which, observe, arith):
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import Adam, LAMB
    from neural_audio_fp_amd.model.fp.specaug_chain.specaug_chain import get_specaug_chain_layer
    n = bsz // 2
    xa, xp = _synthetic_pairs(n, seed=bsz)
    m_pre, m_specaug, m_fp = T.build_fp(cfg)
    assert m_fp.split_arithmetic == (2 if arith == 'x6' else 0)      # x6: forward_train and the transposed convs on the exact split
    w = _inputs.weights(seed=40 + (bsz % 7))
    m_fp.set_weights(_inputs.weight_list(w))
    # the step's own features: spec-augment with the same seeded draws on a second chain object
    m_specaug.rng = np.random.default_rng(bsz)
    twin = get_specaug_chain_layer(cfg); twin.rng = np.random.default_rng(bsz)
    feat = twin(m_pre(torch.cat([xa, xp], dim=0)))
    emb0 = m_fp(feat)                                                            # inference forward, variables before the step
    w0 = [v.detach().cpu().numpy().astype(np.float64) for v in m_fp.trainable_variables]
    lr = 1e-4 if which == 'adam' else 1e-3
    opt = Adam(learning_rate=lr) if which == 'adam' else LAMB(learning_rate=lr)
    bucket = T.GradientBucket(m_fp)
    loss_obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=cfg['LOSS']['TAU'])
    loss, _ = T.train_step((xa, xp), m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
    torch.cuda.synchronize()
    # (1) loss == oracle NT-Xent on the HIP embeddings (NTxent_loss_single_gpu.py:52-82)
    e = emb0.cpu().numpy()
    wl = o_nt.compute_loss(e[:n], e[n:], cfg['LOSS']['TAU'])[0]
    assert abs(float(loss) - wl) < 1e-4 * max(1.0, abs(wl)), (float(loss), wl)
    # (2) updated variables == the oracle's rule on the HIP gradients (trainer.py:47-48; lamb_optimizer.py:123-158)
    grads = [g.detach().cpu().numpy().astype(np.float64) for g in bucket.views]
    assert all(np.isfinite(g).all() for g in grads) and max(np.abs(g).max() for g in grads) > 0
    for i, (v, vl) in enumerate(zip(m_fp.trainable_variables, m_fp.variable_lengths())):
        want = _oracle_update(which, w0[i], grads[i], lr, vl)
        got = v.detach().cpu().numpy().astype(np.float64)
        # one step of size <= lr (Adam) / <= lr * |w| / |u| * |u| (LAMB): float32 arithmetic vs float64.  The error is the
        # rounding of the float32 variable itself (half an ulp of |w|) plus 1e-4 of the step
        observe('updated variable, excess over ulp(|w|)/2, in units of lr',
                max(0.0, np.abs(got - want).max() - 6e-8 * (np.abs(w0[i]).max() + 1e-30)) / lr, 1e-4)
    assert opt.iterations == 1
    # (3) the gradients: 16 samples spread over the batch (anchors and replicas, first / last tiles) vs float64 autograd
    d_a, d_b = o_nt.grad_embeddings(e[:n], e[n:], cfg['LOSS']['TAU'])
    d_full = np.concatenate([d_a, d_b]).astype(np.float32)
    idx = sorted({0, 1, 2, 3, n // 2, n - 1, n, n + 1, n + 5, bsz - 2, bsz - 1, 127, 128, 129, n + 127, n + 128})
    assert len(idx) == 16
    mask = np.zeros_like(d_full); mask[idx] = d_full[idx]
    m_chk = m_fp                                                                 # same handle and workspace (67 GB at 5120),
    m_chk.set_weights(_inputs.weight_list(w))                                    # back at the variables before the step
    emb_chk = m_chk.forward_train(feat)
    assert float((emb_chk - emb0).abs().max()) < 1e-5
    g_sub = [t.detach().cpu().numpy() for t in m_chk.backward(torch.from_numpy(mask).cuda())]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    tf = torch_ref.TorchFingerprinter(w, dtype=torch.float64, requires_grad=True)
    emb64 = tf(feat[idx].double().cpu())
    (emb64 * torch.from_numpy(d_full[idx]).double()).sum().backward()
    assert float((emb64.detach() - emb0[idx].double().cpu()).abs().max()) < 2e-5
    worst = 0.0
    for i, (gs, p) in enumerate(zip(g_sub, tf.params)):
        wg = p.grad.numpy()
        err = np.abs(gs - wg).max() / (np.abs(wg).max() + 1e-30)
        worst = max(worst, err)
    observe('gradient of 16 samples, rel. to the tensor max', worst, 1e-4)
    print(f'BSZ {bsz} {which}: loss {float(loss):.5f} (oracle {wl:.5f}); worst relative gradient error {worst:.2e}')


# ------------------------------------------------------------------------------------------------------------------
# configs[4] scaled
# ------------------------------------------------------------------------------------------------------------------
def test_config4_two_ranks_write_one_million_row_memmap(nafp, cfg, tmp_path):
    """Two processes (gloo; both on cuda:0 -- the box has one GPU) shard 1,048,000 rows on max-normalisation-group
    boundaries and write their slices of ONE dummy_db.mm (generate.py:131-188 format); the file then opens through
    eval_faiss.load_memmap_data and the exact index returns every probed row at its own id."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29900 + (os.getpid() % 90)
    n_rows = 1_048_000
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'tests', '_fullscale_worker.py'), str(tmp_path), str(n_rows)],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    from neural_audio_fp_amd.eval.eval_faiss import load_memmap_data
    db, shape = load_memmap_data(str(tmp_path) + '/', 'dummy_db')
    assert tuple(shape) == (n_rows, 128) and db.shape == (n_rows, 128)
    import json
    meta = json.load(open(tmp_path / 'ranks.json'))
    assert meta['0'][0] == 0 and meta['0'][1] == meta['1'][0] and meta['1'][1] == n_rows
    assert meta['0'][1] % 125 == 0                                              # split on a group boundary
    # no row left unwritten (norm 1), spot rows recomputed in this process agree with what the ranks wrote
    norms = np.linalg.norm(np.asarray(db[::997]), axis=1)
    assert np.abs(norms - 1).max() < 1e-5
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _fullscale_worker as FW
    m_pre = nafp.get_melspec_layer(cfg)
    m_fp = nafp.FingerPrinter(seed=FW.WEIGHT_SEED)
    for g0 in (0, meta['0'][1] - 125, meta['1'][0], n_rows - 125):             # first / last group of each rank
        x = FW.synth_rows(g0, 125, 'cuda')
        want = m_fp(m_pre(x, group_size=125)).cpu().numpy()
        # (the ranks ran launches of 625 rows, this is one of 125: tile shapes and split-K factors -- the fp32 summation
        # order -- depend on the launch size, so the two agree to rounding, not bit for bit)
        blk = np.asarray(db[g0:g0 + 125])
        assert np.abs(blk - want).max() < 3e-6, g0
        assert (1 - (blk * want).sum(1)).max() < 1e-6, g0                        # what a search sees: the cosine
    # search: every probed row comes back at its own id (eval_faiss.py:141-146, 209 with the exact index)
    from neural_audio_fp_amd.eval.eval_faiss import FlatL2Index
    index = FlatL2Index(128, capacity=n_rows)
    index.add(np.asarray(db))
    probe = np.unique(np.concatenate([np.arange(0, n_rows, 7919), [meta['0'][1] - 1, meta['0'][1], n_rows - 1]]))
    _, ids = index.search(np.asarray(db[probe]), 1)
    assert np.array_equal(ids[:, 0], probe)


def test_config4_one_rank_full_share_12_5_million_rows(nafp, cfg, tmp_path):
    """BASELINE.json configs[4] at ONE RANK'S FULL SIZE: 100 M segments sharded 8 ways = 12,500,000 rows = 6.4 GB of
    fingerprints per GPU (generate.py:131-188: one np.memmap float32 (n, 128) + <key>_shape.npy).  The rows go through
    the product's writer (`write_fingerprints_from_device_rows`: launches of 5 max-normalisation groups round-robin over
    4 HIP streams, pinned downloads, memmap stores) from the seeded on-the-fly audio of tests/_fullscale_worker.py -- no
    443 GB dataset exists on the box.  Checked: whole 125-groups against the float64 ORACLE (first, last, and the two
    groups either side of a 625-row launch boundary in the middle of the file), every sampled row has unit norm, the
    file opens through eval_faiss.load_memmap_data, and the exact index over all 12.5 M rows returns every probed row at
    its own id.  The sustained rate (audio synthesis, D2H, memmap stores and the final flush included) is printed and
    kept in gpurun_out/fullscale_r04.json.  NAFP_FULLSHARE_ROWS scales the test down for quick runs."""
    import json
    import shutil
    import time
    from neural_audio_fp_amd.model import generate as G
    from neural_audio_fp_amd.eval.eval_faiss import FlatL2Index, load_memmap_data
    import _fullscale_worker as FW
    n_rows = int(os.environ.get('NAFP_FULLSHARE_ROWS', 12_500_000))
    group, launch_groups = FW.GROUP, 5
    assert cfg['BSZ']['TS_BATCH_SZ'] == group and n_rows % group == 0
    need = n_rows * 128 * 4
    free = shutil.disk_usage(str(tmp_path)).free
    if free < need + (1 << 30):
        pytest.skip(f'{free / 1e9:.1f} GB free under {tmp_path}: the 12.5 M-row file needs {need / 1e9:.1f} GB')
    out_dir = str(tmp_path) + '/'
    w = _inputs.weights(seed=23)
    m_pre, m_fp = nafp.get_melspec_layer(cfg), nafp.get_fingerprinter(cfg)
    m_fp.set_weights(_inputs.weight_list(w))
    arr = np.memmap(out_dir + 'dummy_db.mm', dtype='float32', mode='w+', shape=(n_rows, 128))       # generate.py:157-161
    np.save(out_dir + 'dummy_db_shape.npy', (n_rows, 128))
    m_fp(m_pre(FW.synth_rows(0, launch_groups * group, 'cuda'), group_size=group, defer=True))      # warm-up: plans, workspaces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r0, r1 = G.write_fingerprints_from_device_rows(lambda a, n: FW.synth_rows(a, n, 'cuda'), n_rows, m_pre, m_fp, arr, group,
                                                   rank=0, world=1, launch_groups=launch_groups)
    t_write = time.perf_counter() - t0
    arr.flush()
    t_all = time.perf_counter() - t0
    assert (r0, r1) == (0, n_rows)
    del arr
    rate = n_rows / t_all
    rec = {'rows': n_rows, 'bytes': need, 'seconds_incl_flush': round(t_all, 2), 'seconds_before_flush': round(t_write, 2),
           'rows_per_s_incl_flush': round(rate, 1), 'launch_rows': launch_groups * group, 'streams': G.N_STREAMS,
           'what': 'one rank\'s share of BASELINE configs[4] (100 M rows / 8 ranks) on one MI355X: seeded on-device audio -> '
                   'log-mel -> encoder -> pinned D2H -> np.memmap stores -> flush'}
    print('fullscale:', json.dumps(rec))
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'fullscale_r04.json'), 'w') as f:
            json.dump(rec, f)
    except OSError:
        pass
    db, shape = load_memmap_data(out_dir, 'dummy_db')                               # eval_faiss.py:18-62
    assert tuple(shape) == (n_rows, 128) and db.shape == (n_rows, 128)
    norms = np.linalg.norm(np.asarray(db[::4999]), axis=1)                          # no row left unwritten
    assert np.abs(norms - 1).max() < 1e-5
    # whole groups against the oracle: first, last, and the groups either side of a launch boundary in mid-file
    launch = launch_groups * group
    mid = (n_rows // 2) // launch * launch
    worst_cos, worst_abs = 0.0, 0.0
    for g0 in sorted({0, mid - group, mid, n_rows - group}):
        x = FW.synth_rows(g0, group, 'cuda').cpu().numpy()
        want = _oracle_fingerprints(x, w)                                            # one group = one m_pre batch (melspectrogram.py:108)
        blk = np.asarray(db[g0:g0 + group])
        worst_cos = max(worst_cos, float((1 - (blk * want).sum(1)).max()))
        worst_abs = max(worst_abs, float(np.abs(blk - want).max()))
        assert (1 - (blk * want).sum(1)).max() < 1e-5, g0                           # contract 1e-3
        assert np.abs(blk - want).max() < 1e-4, g0
    print(f'fullscale: worst 1 - cos vs oracle {worst_cos:.2e}, worst |diff| {worst_abs:.2e}; {rate:.0f} rows/s incl. flush')
    # search: every probed row comes back at its own id out of ALL rows (eval_faiss.py:141-146, 209 with the exact index)
    index = FlatL2Index(128, capacity=n_rows)
    index.add(db)
    assert index.ntotal == n_rows
    probe = np.unique(np.concatenate([np.arange(0, n_rows, 104_729), [mid - 1, mid, n_rows - 1]]))
    _, ids = index.search(np.asarray(db[probe]), 1)
    assert np.array_equal(ids[:, 0], probe)
