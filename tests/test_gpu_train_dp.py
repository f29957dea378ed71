"""GPU: the data-parallel train step (all-gather of embeddings, summed
gradient w.r.t. the gathered arrays, one flat gradient all-reduce) equals the single-process step
on the concatenated batch, and the replicas stay bit-identical."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _inputs
import _dp_train_worker as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('world,n,arithmetic', [(2, 4, 'f32'), (2, 4, 'x6'), (4, 2, 'f32'), (8, 2, 'f32')])
def test_multi_rank_step_equals_single_process_step(nafp, tmp_path, monkeypatch, world, n, arithmetic):
    """`world` ranks x n anchors each: all-gather of the embeddings, reduce-scatter of d(emb), the flat gradient
    buffer all-reduced in NAFP_GRAD_GROUPS pieces on the communication stream behind the library's gradient-group
    events (trainer.GradientBucket.all_reduce).  (8, 2): the rank-offset labels and the reduce-scatter layout at the
    world size of BASELINE.json configs[3] (NTxent_loss_tpu.py:42-54, 57-87) -- eight processes share the one GPU over gloo."""
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import LAMB
    # (2, 4, 'x6'): the same step with forward_train and the transposed convs on the exact 3-way bf16 split (NAFP_BF16X3=2 in the environment of the
    # ranks and of this process): the distributed path under the option -- replicas bit-identical, equal to the single-process step
    if arithmetic == 'x6':
        monkeypatch.setenv('NAFP_BF16X3', '2')
    else:
        monkeypatch.delenv('NAFP_BF16X3', raising=False)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29600 + (os.getpid() % 300) + world + (7 if arithmetic == 'x6' else 0)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'tests', '_dp_train_worker.py'), str(tmp_path), str(n)],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rs = [torch.load(tmp_path / f'rank{k}.pt', weights_only=True) for k in range(world)]
    r0 = rs[0]
    for rk in rs[1:]:
        assert r0['losses'] == rk['losses']
        assert torch.equal(r0['grad0'], rk['grad0'])
        for a, b in zip(r0['params'], rk['params']):
            assert torch.equal(a, b)
    # single process on the concatenated batch: anchors of rank 0, 1, ... then the replicas likewise
    feats = [W.features(k, n) for k in range(world)]
    X = (torch.from_numpy(np.concatenate([f[0] for f in feats])).cuda(),
         torch.from_numpy(np.concatenate([f[1] for f in feats])).cuda())
    m_fp = nafp.FingerPrinter(seed=0)
    assert m_fp.split_arithmetic == (2 if arithmetic == 'x6' else 0)
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=31)))
    bucket = T.GradientBucket(m_fp)
    opt = LAMB(learning_rate=1e-3)
    loss_obj = nafp.NTxentLoss(n_org=world * n, n_rep=world * n, tau=0.05)
    loss, _ = T.train_step(X, W.Identity(), W.Identity(), m_fp, loss_obj, opt, bucket)
    assert abs(float(loss) - r0['losses'][0]) < 1e-5 * max(1.0, abs(float(loss)))
    g1 = bucket.flat.cpu()
    # NOTE the LN statistics are per sample, so splitting the batch changes nothing but summation order
    o = 0
    for v in m_fp.trainable_variables:
        a, b = g1[o:o + v.numel()], r0['grad0'][o:o + v.numel()]
        o += v.numel()
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-9
    loss2, _ = T.train_step(X, W.Identity(), W.Identity(), m_fp, loss_obj, opt, bucket)
    assert abs(float(loss2) - r0['losses'][1]) < 1e-3 * max(1.0, abs(float(loss2)))
    assert float(loss2) < float(loss)


def test_one_rank_rccl_group_runs_every_collective_of_the_step(nafp, tmp_path):
    """the distributed branch of `train_step` on the REAL backend.  A process group of one rank
    is legal for RCCL, so `all_gather_into_tensor`, `reduce_scatter_tensor` (the nccl path of `_reduce_scatter`, no
    fallback) and the 4 gradient-piece all-reduces on the communication stream all execute on RCCL; the step must equal
    the non-distributed step on the same batch."""
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import LAMB
    n = 6
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29400 + (os.getpid() % 150)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'tests', '_dp_train_worker.py'), str(tmp_path), str(n), 'nccl'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(tmp_path / 'rank0.pt', weights_only=True)
    assert got['backend'] == 'nccl' and got['world'] == 1
    fa, fp = W.features(0, n)
    X = (torch.from_numpy(fa).cuda(), torch.from_numpy(fp).cuda())
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=31)))
    bucket = T.GradientBucket(m_fp)
    opt = LAMB(learning_rate=1e-3)
    loss_obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    assert T._dist() is None                                        # this process: the plain single-device step
    loss, _ = T.train_step(X, W.Identity(), W.Identity(), m_fp, loss_obj, opt, bucket)
    assert abs(float(loss) - got['losses'][0]) < 1e-5 * max(1.0, abs(float(loss)))
    g1 = bucket.flat.cpu()
    o = 0
    for v in m_fp.trainable_variables:
        a, b = g1[o:o + v.numel()], got['grad0'][o:o + v.numel()]
        o += v.numel()
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-9
    loss2, _ = T.train_step(X, W.Identity(), W.Identity(), m_fp, loss_obj, opt, bucket)
    assert abs(float(loss2) - got['losses'][1]) < 1e-3 * max(1.0, abs(float(loss2)))


def test_weights_marked_dirty_then_four_streams(nafp):
    """the re-pack of the weights (nafp_encoder_set_weights: copies, packs, G/Hb launches into the handle's
    shared blob) is enqueued on ONE stream; forwards issued right afterwards on other streams must wait for it, and a
    later re-pack must wait for the forwards still reading the old blob."""
    rng = np.random.default_rng(3)
    feat = torch.from_numpy((-rng.uniform(0, 1.2, size=(4, 640, 256, 32, 1))).astype(np.float32)).cuda()
    m_fp = nafp.FingerPrinter(seed=1)
    w_a, w_b = _inputs.weight_list(_inputs.weights(seed=5)), _inputs.weight_list(_inputs.weights(seed=6))
    want = {}
    for tag, w in (('a', w_a), ('b', w_b)):
        m_fp.set_weights(w)
        want[tag] = [m_fp(feat[k]).clone() for k in range(4)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    for rep in range(6):
        tag, w = ('a', w_a) if rep % 2 == 0 else ('b', w_b)
        m_fp.set_weights(w)                      # dirty: the next forward re-packs on ITS stream
        outs = []
        for k in range(4):
            with torch.cuda.stream(streams[(k + rep) % 4]):
                outs.append(m_fp(feat[k]))
        torch.cuda.synchronize()
        for k in range(4):
            assert float((outs[k] - want[tag][k]).abs().max()) < 1e-6, (rep, k)


def test_pack_embedding_grads_kernel_equals_the_torch_packing(nafp):
    """nafp_pack_embedding_grads: send[r] = [d/d a rows of rank r | d/d b rows of rank r | loss * scale x 4]."""
    from neural_audio_fp_amd import _lib
    lib = _lib.load()
    world, n_a, d = 8, 320, 128
    g = torch.Generator(device='cuda').manual_seed(1)
    da = torch.randn((world * n_a, d), generator=g, device='cuda')
    db = torch.randn((world * n_a, d), generator=g, device='cuda')
    loss = torch.tensor([3.25], device='cuda')
    send = torch.full((world, 2 * n_a * d + 4), float('nan'), device='cuda')
    _lib.check(lib.nafp_pack_embedding_grads(_lib.ptr(da), _lib.ptr(db), _lib.ptr(loss), 0.5, world, n_a, d, _lib.ptr(send),
                                             _lib.current_stream()), 'pack_embedding_grads')
    want = torch.cat([da.reshape(world, -1), db.reshape(world, -1), torch.full((world, 4), 1.625, device='cuda')], dim=1)
    assert torch.equal(send, want)
