"""MODEL.BN = 'layer_norm1d' / 'batch_norm' (nnfp.py:63-71, 250): the two normalisation alternates of the ConvLayers.

The default 'layer_norm2d' is folded into the GEMM epilogues; the alternates run on the same kernels with identity statistics and
their own positional images (csrc/norm.hip).  Checked here through the C ABI against the float64 restatement of the reference graph
(oracle/nnfp.py `norm=`; keras LayerNormalization(axis=-1), and BatchNormalization as the reference calls it: no `training`
argument anywhere, hence the moving statistics' affine map in the train step too) and, for the gradients, float64 autograd over
oracle/torch_ref.py."""
import numpy as np
import pytest
import torch

from oracle import nnfp as o_nnfp, torch_ref
import _inputs

pytestmark = pytest.mark.gpu

NORMS = ['layer_norm1d', 'batch_norm']


def _weights(norm, seed=7):
    return o_nnfp.convert_norm(_inputs.weights(seed=seed), norm, seed=seed + 1)


def _model(nafp, norm, w):
    m = nafp.FingerPrinter(seed=0, norm=norm)
    m.set_weights(_inputs.weight_list(w))
    return m


@pytest.mark.parametrize('norm', NORMS)
def test_tensor_table(nafp, norm):
    """gamma / beta of shape (C,); batch_norm: 32 non-trainable moving statistics behind the 68 trainable tensors; keras initial values."""
    m = nafp.FingerPrinter(seed=0, norm=norm)
    names = nafp.model.fp.nnfp.tensor_names(norm)
    assert len(m.trainable_variables) == 68 and len(names) == (100 if norm == 'batch_norm' else 68)
    assert tuple(m.trainable_variables[2].shape) == (128,) and tuple(m.trainable_variables[4 * 15 + 3].shape) == (1024,)
    assert bool((m.trainable_variables[2] == 1).all()) and bool((m.trainable_variables[3] == 0).all())
    nt = m.non_trainable_variables
    if norm == 'batch_norm':
        assert len(nt) == 32 and names[68] == 'front_conv.0.BN_1x3.moving_mean' and names[99] == 'front_conv.7.BN_3x1.moving_variance'
        assert bool((nt[0] == 0).all()) and bool((nt[1] == 1).all()) and tuple(nt[31].shape) == (1024,)
    else:
        assert nt == []
    assert set(m.state_dict()) == set(names)
    # any string other than the two layer norms is batch normalisation (the `else` of nnfp.py:69-71)
    assert nafp.model.fp.nnfp.norm_kind('whatever') == 2 and nafp.model.fp.nnfp.norm_kind('layer_norm2d') == 0


@pytest.mark.parametrize('B', [1, 9, 130])
@pytest.mark.parametrize('norm', NORMS)
def test_forward_matches_oracle(nafp, norm, B, observe):
    """B = 130: the launch plans of the bench sizes (256-row tiles, split-K with both finishes); 1 / 9: ragged tiles."""
    w = _weights(norm)
    m = _model(nafp, norm, w)
    rng = np.random.default_rng(B)
    feat = (-rng.uniform(0, 1.2, size=(B, 256, 32, 1))).astype(np.float32)
    ft = torch.from_numpy(feat).cuda()
    emb = m(ft).cpu().numpy()
    nb = min(B, 9)
    sel = np.linspace(0, B - 1, nb).astype(int)
    want = o_nnfp.fingerprinter(feat[sel], w, norm=norm)
    observe(f'{norm} fingerprint component', np.abs(emb[sel] - want).max(), 2e-5)
    flat = m.front_conv(ft).cpu().numpy()
    want_flat = o_nnfp.front_conv(feat[sel], w, norm=norm)
    observe(f'{norm} flatten output, rel. to its max', np.abs(flat[sel] - want_flat).max() / np.abs(want_flat).max(), 2e-5)
    # the training forward is the same map, and a second launch the same bits
    assert float((m.forward_train(ft).cpu() - torch.from_numpy(emb)).abs().max()) < 2e-6
    assert np.array_equal(m(ft).cpu().numpy(), emb)


@pytest.mark.parametrize('norm', NORMS)
def test_the_deferred_front_end_path(nafp, cfg, norm):
    """m_fp(m_pre(x, defer=True)): conv0 finishes the log-mel tail as it loads -- same result as the two-call form."""
    w = _weights(norm)
    m = _model(nafp, norm, w)
    m_pre = nafp.get_melspec_layer(cfg)
    x = torch.from_numpy(_inputs.audio(6, seed=2)).cuda()
    a = m(m_pre(x, defer=True))
    b = m(m_pre(x))
    assert float((a - b).abs().max()) < 1e-6


@pytest.mark.parametrize('B', [2, 5])
@pytest.mark.parametrize('norm', NORMS)
def test_backward_matches_autograd(nafp, norm, B, observe):
    w = _weights(norm, seed=12)
    rng = np.random.default_rng(40 + B)
    feat = (-rng.uniform(0, 1.2, size=(B, 256, 32, 1))).astype(np.float32)
    d_emb = rng.normal(size=(B, 128)).astype(np.float32)
    m = _model(nafp, norm, w)
    emb = m.forward_train(torch.from_numpy(feat).cuda())
    grads = [g.cpu().numpy() for g in m.backward(torch.from_numpy(d_emb).cuda())]
    tf = torch_ref.TorchFingerprinter(w, dtype=torch.float64, requires_grad=True, norm=norm)
    want_emb = tf(torch.tensor(feat, dtype=torch.float64))
    (want_emb * torch.tensor(d_emb, dtype=torch.float64)).sum().backward()
    want = [p.grad.numpy() for p in tf.params]
    assert np.abs(emb.cpu().numpy() - want_emb.detach().numpy()).max() < 2e-5
    assert len(grads) == len(want) == 68
    names = nafp.model.fp.nnfp.tensor_names(norm)
    worst, worst_name = 0.0, ''
    for i, (g, wg) in enumerate(zip(grads, want)):
        assert g.shape == wg.shape, names[i]
        err = np.abs(g - wg).max() / (np.abs(wg).max() + 1e-12)
        if err > worst:
            worst, worst_name = err, names[i]
    observe(f'{norm} gradient, rel. to the tensor max ({worst_name})', worst, 1e-4)
    # run to run: the same forward bits; a second backward agrees to rounding (float atomics in the parameter sums)
    g2 = [g.cpu().numpy() for g in m.backward(torch.from_numpy(d_emb).cuda())]
    assert all(np.abs(a - b).max() <= 1e-5 * (np.abs(a).max() + 1e-12) for a, b in zip(grads, g2))


@pytest.mark.parametrize('norm', NORMS)
def test_backward_at_a_bench_size_is_additive_over_the_batch(nafp, norm):
    """B = 640 (the launch plans of the bench sizes: side stream, slabs, 256-row tiles): the parameter gradient for a given
    dL/d(emb) is a sum over samples -- also with batch_norm, whose statistics are the moving ones, not the batch's."""
    B, Q = 640, 4
    g = torch.Generator(device='cuda').manual_seed(3)
    feat = -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')
    d_emb = torch.randn((B, 128), generator=g, device='cuda')
    m = _model(nafp, norm, _weights(norm, seed=5))
    emb = m.forward_train(feat)
    full = [t.clone() for t in m.backward(d_emb)]
    parts, n = None, B // Q
    for k in range(Q):
        e = m.forward_train(feat[k * n:(k + 1) * n])
        assert float((e - emb[k * n:(k + 1) * n]).abs().max()) < 2e-5
        gk = m.backward(d_emb[k * n:(k + 1) * n])
        parts = [t.clone() for t in gk] if parts is None else [a + b for a, b in zip(parts, gk)]
    for i, (a, b) in enumerate(zip(full, parts)):
        assert float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20) < 2e-4, i


@pytest.mark.parametrize('norm', NORMS)
def test_train_step_descends(nafp, cfg, norm):
    """trainer.train_step with MODEL.BN set: the loss of a fixed batch goes down, the moving statistics never move."""
    import copy
    from neural_audio_fp_amd.model import trainer as T
    c = copy.deepcopy(cfg)
    c['MODEL']['BN'] = norm
    c['BSZ']['TR_BATCH_SZ'] = 16
    from neural_audio_fp_amd.model.fp.specaug_chain.specaug_chain import get_specaug_chain_layer
    m_pre, m_aug, m_fp = nafp.get_melspec_layer(c), get_specaug_chain_layer(c), nafp.get_fingerprinter(c, trainable=True)
    assert m_fp.norm == norm
    m_aug.bypass = True
    opt = T.make_optimizer(c, 100)
    loss_obj = nafp.NTxentLoss(8, 8, 0.05)
    x = _inputs.audio(16, seed=9)
    Xa, Xp = torch.from_numpy(x[:8]).cuda(), torch.from_numpy(x[8:] * 0.8 + 0.05 * x[:8]).cuda()
    before = [v.clone() for v in m_fp.non_trainable_variables]
    losses = [float(T.train_step((Xa, Xp), m_pre, m_aug, m_fp, loss_obj, opt)[0]) for _ in range(12)]
    assert losses[-1] < losses[0], losses
    assert all(torch.equal(a, b) for a, b in zip(before, m_fp.non_trainable_variables))


@pytest.mark.parametrize('norm', NORMS)
def test_checkpoint_round_trip(nafp, cfg, norm, tmp_path):
    """save_checkpoint / load_checkpoint (generate.py:26-52) carry every tensor of the alternates -- for batch_norm the moving
    statistics too -- and a model restored from the file gives the same fingerprints, bit for bit."""
    import copy
    from neural_audio_fp_amd.model import generate as g
    c = copy.deepcopy(cfg)
    c['MODEL']['BN'] = norm
    w = _weights(norm, seed=21)
    m = nafp.get_fingerprinter(c)
    m.set_weights(_inputs.weight_list(w))
    feat = -1.2 * torch.rand((7, 256, 32, 1), generator=torch.Generator(device='cuda').manual_seed(1), device='cuda')
    emb = m(feat).clone()
    root = str(tmp_path) + '/checkpoint/'
    g.save_checkpoint(root, 'exp', 5, m)
    m2 = nafp.get_fingerprinter(c)
    assert float((m2(feat) - emb).abs().max()) > 1e-3                     # (freshly initialised: another model)
    assert g.load_checkpoint(root, 'exp', None, m2) == 5
    assert torch.equal(m2(feat), emb)
    # a checkpoint of another normalisation is refused by name / shape, not silently half-loaded
    other = copy.deepcopy(cfg)
    m3 = nafp.get_fingerprinter(other)                                     # layer_norm2d
    with pytest.raises((KeyError, ValueError)):
        g.load_checkpoint(root, 'exp', 5, m3)


@pytest.mark.parametrize('B', [9, 130])
@pytest.mark.parametrize('norm', NORMS)
def test_a_non_finite_sample_gives_a_nan_row_and_nothing_else(nafp, norm, B):
    """As with the default normalisation (tests/test_gpu_nonfinite.py): the identity statistics the alternates hand to every
    consumer keep the sample's poison flag, so a NaN / Inf sample is a NaN fingerprint -- not finite garbage behind the packed ELU
    of the next epilogue -- and the other rows of the launch are bit-equal to the clean launch."""
    m = _model(nafp, norm, _weights(norm))
    g = torch.Generator(device='cuda').manual_seed(B)
    feat = -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')
    clean = m(feat).clone()
    assert bool(torch.isfinite(clean).all())
    bad = sorted({1, B // 2, B - 1})
    dirty = feat.clone()
    dirty[bad[0], 37, 5, 0] = float('nan')
    dirty[bad[1], 200, 31, 0] = float('inf')
    dirty[bad[2]] = float('nan')
    ok = torch.ones(B, dtype=torch.bool, device='cuda')
    ok[bad] = False
    emb = m(dirty)
    assert bool(torch.isnan(emb[~ok]).all()), emb[~ok]
    assert torch.equal(emb[ok], clean[ok])
    emb_t = m.forward_train(dirty)
    assert bool(torch.isnan(emb_t[~ok]).all()) and bool(torch.isfinite(emb_t[ok]).all())
    assert torch.equal(m(feat), clean)


@pytest.mark.parametrize('emb_sz', [64, 256])
@pytest.mark.parametrize('norm', NORMS)
def test_other_geometries(nafp, norm, emb_sz, observe):
    """The (256, 63, 1) input of nnfp.py:266-268 (odd frame counts: symmetric SAME padding, ragged position tiles, other row counts for
    the row pass) and MODEL.EMB_SZ 64 / 256, forward and backward, with the alternates."""
    rng = np.random.default_rng(emb_sz)
    B = 3
    feat = (-rng.uniform(0, 1.2, size=(B, 256, 63, 1))).astype(np.float32)
    w = o_nnfp.convert_norm(o_nnfp.init_weights(seed=6, input_shape=(256, 63, 1), emb_sz=emb_sz, randomize_affine=True), norm, seed=3)
    m = nafp.FingerPrinter(input_shape=(256, 63, 1), emb_sz=emb_sz, norm=norm)
    m.set_weights(_inputs.weight_list(w))
    ft = torch.from_numpy(feat).cuda()
    emb = m(ft).cpu().numpy()
    want = o_nnfp.fingerprinter(feat, w, norm=norm)
    assert emb.shape == (B, emb_sz)
    observe(f'{norm}, 63 frames, EMB_SZ {emb_sz}: fingerprint component', np.abs(emb - want).max(), 2e-5)
    d_emb = rng.normal(size=(B, emb_sz)).astype(np.float32)
    m.forward_train(ft)
    grads = [g.cpu().numpy() for g in m.backward(torch.from_numpy(d_emb).cuda())]
    tf = torch_ref.TorchFingerprinter(w, input_shape=(256, 63, 1), dtype=torch.float64, requires_grad=True, norm=norm)
    (tf(torch.tensor(feat, dtype=torch.float64)) * torch.tensor(d_emb, dtype=torch.float64)).sum().backward()
    worst = max(np.abs(g - p.grad.numpy()).max() / (np.abs(p.grad.numpy()).max() + 1e-12) for g, p in zip(grads, tf.params))
    observe(f'{norm}, 63 frames, EMB_SZ {emb_sz}: gradient, rel. to the tensor max', worst, 1e-4)


def test_batch_norm_with_large_finite_activations_stays_finite(nafp, observe):
    """Round-5 ADVICE: inference-mode batch normalisation normalises nothing per sample, so a checkpoint with a large scale (or a small
    moving variance) drives the activations into the thousands.  keras gives finite rows for finite activations; the statistics these
    kernels still accumulate (they only carry the NaN poison of a sample here) must not poison a sample because a partial sum left the
    fixed-point range -- a finite partial beyond the range is dropped (nafp_common.h stat_add, range_is_benign)."""
    w = _weights('batch_norm')
    w['ln0.gamma'] = (w['ln0.gamma'] * 30000.0).astype(np.float32)       # (C,): the batch-norm scale of layer 0
    m = _model(nafp, 'batch_norm', w)
    rng = np.random.default_rng(5)
    feat = (-rng.uniform(0, 1.2, size=(9, 256, 32, 1))).astype(np.float32)
    flat = m.front_conv(torch.from_numpy(feat).cuda()).cpu().numpy()
    emb = m(torch.from_numpy(feat).cuda()).cpu().numpy()
    taps = []
    want_flat = o_nnfp.front_conv(feat, w, dtype=np.float64, taps=taps, norm='batch_norm')
    want = o_nnfp.fingerprinter(feat, w, dtype=np.float64, norm='batch_norm')
    rms = max(float(np.sqrt((t ** 2).mean(axis=(1, 2, 3))).max()) for t in taps)
    assert rms > 3000.0, f'the case must drive some layer beyond the fixed-point range of the statistics (largest per-sample RMS {rms:.3g})'
    assert np.isfinite(flat).all() and np.isfinite(emb).all()
    observe('|d flat| / max |flat|', np.abs(flat - want_flat).max() / np.abs(want_flat).max(), 1e-5)
    observe('|d emb|', np.abs(emb - want).max(), 1e-5)
    # a NaN sample is still a NaN row and nothing else
    feat[4, 10, 3, 0] = np.nan
    emb2 = m(torch.from_numpy(feat).cuda()).cpu().numpy()
    assert np.isnan(emb2[4]).all() and np.array_equal(np.delete(emb2, 4, 0), np.delete(emb, 4, 0))
