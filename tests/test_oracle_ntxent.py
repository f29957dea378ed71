"""CPU: the NT-Xent oracle (NTxent_loss_single_gpu.py:29-82, NTxent_loss_tpu.py:42-137)."""
import numpy as np
import torch

from oracle import ntxent as o_nt, torch_ref
import _inputs


def test_loss_vs_cross_entropy_formulation():
    for n in (2, 3, 17, 60):
        a, b = _inputs.unit_pairs(n, seed=n)
        loss, sim, labels = o_nt.compute_loss(a, b, tau=0.05)
        assert abs(loss - float(torch_ref.ntxent(a, b, 0.05))) < 1e-10
        assert sim.shape == labels.shape == (n, 2 * n - 1)
        assert np.array_equal(labels.argmax(1), np.arange(n)) and labels.sum() == n
        # sim = [ab | aa without diagonal]
        assert np.allclose(sim[:, :n], a.astype(np.float64) @ b.astype(np.float64).T / 0.05)
        assert np.allclose(sim[0, n:], (a.astype(np.float64) @ a.astype(np.float64).T / 0.05)[0, 1:])


def test_drop_diag():
    x = np.arange(9.).reshape(3, 3)
    assert np.array_equal(o_nt.drop_diag(x), [[1, 2], [3, 5], [6, 7]])


def test_analytic_gradient_vs_autograd():
    a, b = _inputs.unit_pairs(7, seed=1)
    ta = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tb = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    torch_ref.ntxent(ta, tb, 0.05).backward()
    ga, gb = o_nt.grad_embeddings(a, b, 0.05)
    assert np.abs(ga - ta.grad.numpy()).max() < 1e-10 and np.abs(gb - tb.grad.numpy()).max() < 1e-10


def test_replica_form_reproduces_single_device_loss():
    # R replicas each holding n_a anchors: the mean over all per-row losses of loss_fn equals
    # compute_loss on the concatenated batch (masking with -1e9 == dropping the diagonal).
    R, n_a = 4, 5
    a, b = _inputs.unit_pairs(R * n_a, seed=9)
    total = 0.0
    for r in range(R):
        loc = np.concatenate([a[r * n_a:(r + 1) * n_a], b[r * n_a:(r + 1) * n_a]])
        total += o_nt.replica_loss_fn(loc, a, b, rank=r, tau=0.05).sum()
    single = o_nt.compute_loss(a, b, 0.05)[0]
    assert abs(total / (R * n_a) - single) < 1e-9


def test_golden_ntxent(golden):
    for n in (5, 60):
        a, b = _inputs.unit_pairs(n, seed=100 + n)
        assert abs(o_nt.compute_loss(a, b, 0.05)[0] - golden[f'ntxent_loss_n{n}'][0]) < 1e-10
