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


def test_ntxent_sharded_rows_reproduce_replica_loss(nafp):
    """One replica of NTxent_loss_tpu.py:90-137: local rows against the gathered columns,
    labels/diagonal offset by rank*n_a.  Sum over ranks / N == the single-device loss."""
    import ctypes
    from neural_audio_fp_amd import _lib
    lib = _lib.load()
    R, n_a = 4, 40
    a, b = _emb(R * n_a, 77)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    total = 0.0
    for r in range(R):
        la, lb = ta[r * n_a:(r + 1) * n_a].contiguous(), tb[r * n_a:(r + 1) * n_a].contiguous()
        out = torch.empty(1, device='cuda')
        need = int(lib.nafp_ntxent_workspace_bytes(n_a, R * n_a))
        ws = torch.empty(need, dtype=torch.uint8, device='cuda')
        _lib.check(lib.nafp_ntxent_forward(_lib.ptr(la), _lib.ptr(lb), _lib.ptr(ta), _lib.ptr(tb), n_a, R * n_a,
                                           r * n_a, 128, 0.05, _lib.ptr(out), None, None, None, _lib.ptr(ws), need,
                                           _lib.current_stream()), 'ntxent sharded')
        want = o_nt.replica_loss_fn(np.concatenate([a[r * n_a:(r + 1) * n_a], b[r * n_a:(r + 1) * n_a]]), a, b, r, 0.05)
        assert abs(float(out) - want.sum()) < 1e-4 * max(1.0, abs(want.sum()))
        total += float(out)
    assert abs(total / (R * n_a) - o_nt.compute_loss(a, b, 0.05)[0]) < 1e-4


def test_ntxent_argument_errors(nafp):
    obj = nafp.NTxentLoss(n_org=4, n_rep=4, tau=0.05)
    a = torch.zeros(4, 128, device='cuda')
    with pytest.raises(ValueError):
        obj.compute_loss(a, torch.zeros(5, 128, device='cuda'))
    with pytest.raises(nafp._lib.NafpError):
        obj.compute_loss(torch.zeros(4, 128), torch.zeros(4, 128))          # CPU tensors: no fallback
