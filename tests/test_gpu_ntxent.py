"""GPU parity: fused NT-Xent kernel vs the oracle (NTxent_loss_single_gpu.py:52-82)."""
import numpy as np
import pytest
import torch

from oracle import ntxent as o_nt

pytestmark = pytest.mark.gpu


def _emb(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 128))
    b = a + 0.3 * rng.normal(size=(n, 128))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return a.astype(np.float32), b.astype(np.float32)


@pytest.mark.parametrize('n', [3, 60, 97, 320])
def test_ntxent_loss_and_sim(nafp, n):
    a, b = _emb(n, n)
    obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    loss, sim, labels = obj.compute_loss(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    wl, wsim, wlab = o_nt.compute_loss(a, b, tau=0.05)
    assert abs(float(loss) - wl) < 1e-4 * max(1.0, abs(wl))       # fp32 LSE over <= 639 terms
    assert sim.shape == (n, 2 * n - 1)
    assert np.abs(sim.cpu().numpy() - wsim).max() < 1e-4           # logits are in [-20, 20]
    assert np.array_equal(labels.cpu().numpy(), wlab)
