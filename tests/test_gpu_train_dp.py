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


def test_two_rank_step_equals_single_process_step(nafp, tmp_path):
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import LAMB
    n = 4
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29600 + (os.getpid() % 300)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'tests', '_dp_train_worker.py'), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0 = torch.load(tmp_path / 'rank0.pt', weights_only=True)
    r1 = torch.load(tmp_path / 'rank1.pt', weights_only=True)
    assert r0['losses'] == r1['losses']
    assert torch.equal(r0['grad0'], r1['grad0'])
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)
    # single process on the concatenated batch: anchors of rank 0 then rank 1, replicas likewise
    (fa0, fp0), (fa1, fp1) = W.features(0, n), W.features(1, n)
    X = (torch.from_numpy(np.concatenate([fa0, fa1])).cuda(), torch.from_numpy(np.concatenate([fp0, fp1])).cuda())
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=31)))
    bucket = T.GradientBucket(m_fp)
    opt = LAMB(learning_rate=1e-3)
    loss_obj = nafp.NTxentLoss(n_org=2 * n, n_rep=2 * n, tau=0.05)
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
