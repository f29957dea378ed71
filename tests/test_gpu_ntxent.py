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


@pytest.mark.parametrize('n', [5, 60, 97])
def test_ntxent_gradient_vs_oracle(nafp, n, golden):
    # hard pairs (noise 1.5): with the easy pairs of the loss tests the softmax saturates and the
    # gradient is ~1e-7, i.e. rounding noise
    a, b = __import__('_inputs').unit_pairs(n, seed=100 + n, noise=1.5)
    obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    loss, da, db = obj.loss_and_grad(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    wa, wb = o_nt.grad_embeddings(a, b, tau=0.05)
    scale = max(np.abs(wa).max(), np.abs(wb).max())
    # fp32 softmax: (p - 1) near saturation carries ~1e-7 absolute error, times 1/tau = 20 -> 1e-6 floor,
    # plus 2e-4 relative to the largest entry
    assert np.abs(da.cpu().numpy() - wa).max() < 2e-4 * scale + 1e-6
    assert np.abs(db.cpu().numpy() - wb).max() < 2e-4 * scale + 1e-6
    assert abs(float(loss) - o_nt.compute_loss(a, b, 0.05)[0]) < 1e-4
    if n == 5:
        assert np.abs(da.cpu().numpy() - golden['ntxent_grad_a_n5']).max() < 2e-4 * scale + 1e-6
    # the autograd route gives the same numbers
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    l2, _, _ = obj.compute_loss(ta, tb, return_sim=False)
    l2.backward()
    assert torch.equal(ta.grad, da) and torch.equal(tb.grad, db)


def test_ntxent_sharded_gradients_sum_to_full(nafp):
    from neural_audio_fp_amd import _lib
    from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import _ntxent_call
    lib = _lib.load()
    R, n_a = 3, 24
    a, b = __import__('_inputs').unit_pairs(R * n_a, seed=5, noise=1.5)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    tot_a, tot_b = torch.zeros_like(ta), torch.zeros_like(tb)
    for r in range(R):
        sl = slice(r * n_a, (r + 1) * n_a)
        _, _, da, db = _ntxent_call(lib, ta[sl].contiguous(), tb[sl].contiguous(), ta, tb, r * n_a, 0.05, False, True)
        tot_a += da; tot_b += db
    wa, wb = o_nt.grad_embeddings(a, b, 0.05)
    scale = max(np.abs(wa).max(), np.abs(wb).max())
    assert np.abs(tot_a.cpu().numpy() - wa).max() < 2e-4 * scale + 1e-6
    assert np.abs(tot_b.cpu().numpy() - wb).max() < 2e-4 * scale + 1e-6


# ---- BASELINE.json configs[2] (N = 640) and configs[3] (N = 2560, sharded 320 x 2560) at size -----------------------
def _hard(n, seed, d=128):
    return __import__('_inputs').unit_pairs(n, seed=seed, d=d, noise=1.5)


@pytest.mark.parametrize('n', [640, 2560])
def test_ntxent_at_baseline_sizes_loss_sim_and_gradient(nafp, n):
    """NTxent_loss_single_gpu.py:52-82 at the anchor counts of BSZ 1280 and BSZ 5120: loss, the (N, 2N-1) sim_mtx
    and both gradients vs the float64 oracle."""
    a, b = _hard(n, 7000 + n)
    obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    loss, sim, _ = obj.compute_loss(ta, tb)
    wl, wsim, _ = o_nt.compute_loss(a, b, tau=0.05)
    assert abs(float(loss) - wl) < 1e-4 * max(1.0, abs(wl))
    assert sim.shape == (n, 2 * n - 1) and np.abs(sim.cpu().numpy() - wsim).max() < 1e-4
    l2, da, db = obj.loss_and_grad(ta, tb)
    wa, wb = o_nt.grad_embeddings(a, b, tau=0.05)
    scale = max(np.abs(wa).max(), np.abs(wb).max())
    assert abs(float(l2) - wl) < 1e-4 * max(1.0, abs(wl))
    assert np.abs(da.cpu().numpy() - wa).max() < 2e-4 * scale + 1e-6
    assert np.abs(db.cpu().numpy() - wb).max() < 2e-4 * scale + 1e-6


def test_ntxent_sharded_320_of_2560_ranks_0_and_7(nafp):
    """configs[3]: global batch 5120 over 8 ranks = 320 local anchors against 2560 gathered columns.  Per-rank loss
    rows vs NTxent_loss_tpu.py:90-137 (`replica_loss_fn`) for the first and the last rank; the gradient w.r.t. the
    gathered arrays, summed over all 8 ranks, vs the analytic gradient of the single-device loss."""
    from neural_audio_fp_amd import _lib
    from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import _ntxent_call
    lib = _lib.load()
    R, n_a = 8, 320
    a, b = _hard(R * n_a, 99)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    tot_a, tot_b, total = torch.zeros_like(ta), torch.zeros_like(tb), 0.0
    for r in range(R):
        sl = slice(r * n_a, (r + 1) * n_a)
        loss_sum, _, da, db = _ntxent_call(lib, ta[sl].contiguous(), tb[sl].contiguous(), ta, tb, r * n_a, 0.05, False, True)
        tot_a += da; tot_b += db; total += float(loss_sum)
        if r in (0, R - 1):
            want = o_nt.replica_loss_fn(np.concatenate([a[sl], b[sl]]), a, b, r, 0.05)
            assert abs(float(loss_sum) - want.sum()) < 1e-4 * max(1.0, abs(want.sum()))
    wl = o_nt.compute_loss(a, b, 0.05)[0]
    assert abs(total / (R * n_a) - wl) < 1e-4 * max(1.0, abs(wl))
    wa, wb = o_nt.grad_embeddings(a, b, 0.05)
    scale = max(np.abs(wa).max(), np.abs(wb).max())
    assert np.abs(tot_a.cpu().numpy() - wa).max() < 2e-4 * scale + 1e-6
    assert np.abs(tot_b.cpu().numpy() - wb).max() < 2e-4 * scale + 1e-6


@pytest.mark.parametrize('d', [64, 256])
@pytest.mark.parametrize('n', [5, 97, 320])
def test_ntxent_other_embedding_widths(nafp, n, d):
    """MODEL.EMB_SZ 64 / 256 (nnfp.py:250): loss, sim_mtx, gradients, and one sharded rank."""
    from neural_audio_fp_amd import _lib
    from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import _ntxent_call
    a, b = _hard(n, 31 * n + d, d=d)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    loss, sim, _ = obj.compute_loss(ta, tb)
    wl, wsim, _ = o_nt.compute_loss(a, b, tau=0.05)
    assert abs(float(loss) - wl) < 1e-4 * max(1.0, abs(wl)) and np.abs(sim.cpu().numpy() - wsim).max() < 1e-4
    _, da, db = obj.loss_and_grad(ta, tb)
    wa, wb = o_nt.grad_embeddings(a, b, tau=0.05)
    scale = max(np.abs(wa).max(), np.abs(wb).max())
    assert np.abs(da.cpu().numpy() - wa).max() < 2e-4 * scale + 1e-6
    assert np.abs(db.cpu().numpy() - wb).max() < 2e-4 * scale + 1e-6
    if n == 97:                      # ragged split: rank 1 of 2 holds rows [48, 97)
        lo = 48
        ls, _, _, _ = _ntxent_call(_lib.load(), ta[lo:].contiguous(), tb[lo:].contiguous(), ta, tb, lo, 0.05, False, False)
        # replica_loss_fn needs equal shards; take the rows of the single-device loss instead
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        la = np.concatenate([a64 @ b64.T, a64 @ a64.T], 1) / 0.05; lb = np.concatenate([b64 @ a64.T, b64 @ b64.T], 1) / 0.05
        idx = np.arange(n)
        la[idx, n + idx] = -np.inf; lb[idx, n + idx] = -np.inf
        rows = (o_nt._lse(la) - la[idx, idx]) + (o_nt._lse(lb) - lb[idx, idx])
        assert abs(float(ls) - rows[lo:].sum()) < 1e-4 * max(1.0, abs(rows[lo:].sum()))
