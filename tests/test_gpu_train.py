"""GPU: the train step (trainer.py:33-50) on the HIP path -- one-step parameter parity against the
float64 restatement (melspec -> encoder -> NT-Xent -> autograd -> keras Adam), descent on a fixed
batch for Adam and LAMB, and the epoch loop with checkpoint resume."""
import copy

import numpy as np
import pytest
import torch

from oracle import optim as o_optim
from oracle import torch_ref
import _inputs

pytestmark = pytest.mark.gpu


class _NoAug:
    bypass = False

    def __call__(self, x):
        return x


def _pairs(n, seed):
    xa = _inputs.audio(n, seed=seed)
    rng = np.random.default_rng(seed + 1)
    xp = (xa + 0.05 * rng.normal(size=xa.shape)).astype(np.float32)
    return xa, xp


def test_one_adam_step_matches_float64_reference(nafp, cfg):
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import Adam
    from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import NTxentLoss
    n = 6
    xa, xp = _pairs(n, 5)
    w = _inputs.weights(seed=21)
    m_pre = nafp.get_melspec_layer(cfg)
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(_inputs.weight_list(w))
    lr = 1e-3
    opt = Adam(learning_rate=lr)
    loss, _ = T.train_step((torch.from_numpy(xa).cuda(), torch.from_numpy(xp).cuda()), m_pre, _NoAug(), m_fp,
                           NTxentLoss(n_org=n, n_rep=n, tau=0.05), opt)
    # reference: the same step in float64 (feature extraction in float32 like the layer itself)
    feat = torch_ref.melspec_layer(np.concatenate([xa, xp]))
    tf = torch_ref.TorchFingerprinter(w, dtype=torch.float64, requires_grad=True)
    emb = tf(feat.double())
    want_loss = torch_ref.ntxent(emb[:n], emb[n:], tau=0.05)
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) < 1e-4 * max(1.0, abs(float(want_loss)))
    worst = 0.0
    for p, new in zip(tf.params, m_fp.trainable_variables):
        w0, g = p.detach().numpy(), p.grad.numpy()
        want, _, _ = o_optim.adam_step(w0, g, np.zeros_like(w0), np.zeros_like(w0), lr, 1)
        got = new.cpu().numpy().astype(np.float64)
        # the first Adam step is lr*g/(|g|+eps'): compare where the gradient is not at rounding level
        sel = np.abs(g) > 1e-4 * (np.abs(g).max() + 1e-30)
        if sel.any():
            err = np.abs(got - want)[sel].max() / lr
            worst = max(worst, err)
            assert err < 2e-2, err
        assert np.abs(got - w0).max() <= lr * 1.0001 + 1e-7
    print('worst update error / lr', worst)
    assert opt.iterations == 1


@pytest.mark.parametrize('which', ['adam', 'lamb'])
def test_loss_decreases_on_fixed_batch(nafp, cfg, which):
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import Adam, LAMB
    from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import NTxentLoss
    n = 16
    xa, xp = _pairs(n, 9)
    X = (torch.from_numpy(xa).cuda(), torch.from_numpy(xp).cuda())
    m_pre, m_specaug, m_fp = T.build_fp(cfg)
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=4, randomize_affine=False)))
    opt = Adam(learning_rate=1e-4) if which == 'adam' else LAMB(learning_rate=1e-3)
    loss_obj = NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    losses = [float(T.train_step(X, m_pre, _NoAug(), m_fp, loss_obj, opt)[0]) for _ in range(12)]
    print(which, losses)
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0] - 0.05
    # validation step sees the trained variables (set_weights after mark_dirty) and agrees with the train loss scale
    vloss, sim = T.val_step(X, m_pre, m_fp, loss_obj)
    assert sim.shape == (n, 2 * n - 1) and float(vloss) < losses[0]
    # augmented step runs too
    l_aug, _ = T.train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt)
    assert np.isfinite(float(l_aug))


def test_test_step_shapes(nafp, cfg):
    from neural_audio_fp_amd.model import trainer as T
    m_pre, _, m_fp = T.build_fp(cfg)
    xa, xp = _pairs(3, 2)
    f, f2, gf = T.test_step((torch.from_numpy(xa).cuda(), torch.from_numpy(xp).cuda()), m_pre, m_fp)
    assert f.shape == (6, m_fp.flat_dim) and f2.shape == f.shape and gf.shape == (6, 128)
    assert torch.allclose(gf.norm(dim=1), torch.ones(6, device='cuda'), atol=1e-5)
    assert torch.allclose(f2.norm(dim=1), torch.ones(6, device='cuda'), atol=1e-5)
    emb = m_fp(m_pre(torch.from_numpy(np.concatenate([xa, xp])).cuda()))
    assert torch.allclose(emb, gf, atol=1e-5)


def test_trainer_epochs_and_resume(nafp, cfg, tmp_path):
    from neural_audio_fp_amd.model import trainer as T
    c = copy.deepcopy(cfg)
    c['BSZ']['TR_BATCH_SZ'], c['BSZ']['TR_N_ANCHOR'] = 16, 8
    c['DIR']['LOG_ROOT_DIR'] = str(tmp_path) + '/'
    c['TRAIN']['MAX_EPOCH'] = 2
    hist = T.trainer(c, 'unit', train_batches=T.synthetic_batches(c, 3), steps_per_epoch=3)
    assert len(hist) == 2 and all(np.isfinite(hist))
    assert (tmp_path / 'checkpoint' / 'unit' / 'ckpt-2.pt').exists()
    c['TRAIN']['MAX_EPOCH'] = 3
    hist2 = T.trainer(c, 'unit', train_batches=T.synthetic_batches(c, 3), steps_per_epoch=3)
    assert len(hist2) == 1                         # resumed at epoch 3
    ck = torch.load(tmp_path / 'checkpoint' / 'unit' / 'ckpt-3.pt', weights_only=True)
    assert ck['optimizer']['iterations'] == 9 and len(ck['optimizer']['m']) == 68
    with pytest.raises(NotImplementedError):
        c['TRAIN']['OPTIMIZER'] = 'SGD'
        T.trainer(c, 'unit2', train_batches=T.synthetic_batches(c, 1), steps_per_epoch=1)


def test_trainer_on_a_dataset_directory(nafp, cfg, tmp_path):
    """`trainer(cfg, name)` with no batch source: the reference's directory layout (music/train-10k-30s,
    aug/bg/tr, aug/ir/tr) through the device-side loader + augmentation + train step."""
    import wave
    from neural_audio_fp_amd.model import trainer as T
    rng = np.random.default_rng(31)
    t = np.arange(80000) / 8000.0

    def wav(path, pcm):
        path.parent.mkdir(parents=True, exist_ok=True)
        with wave.open(str(path), 'w') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
            w.writeframes(np.asarray(pcm).astype('<i2').tobytes())
    root = tmp_path / 'ds'
    for i in range(6):
        wav(root / 'music' / 'train-10k-30s' / 'a' / f'{i}.wav',
            rng.integers(-1500, 1500, size=80000) + 8000 * np.sin(2 * np.pi * (300 + 400 * i) * t))
    for i in range(2):
        wav(root / 'aug' / 'bg' / 'tr' / f'{i}.wav', rng.integers(-5000, 5000, size=30000))
        wav(root / 'aug' / 'ir' / 'tr' / f'{i}.wav', 12000 * rng.normal(size=900) * np.exp(-np.arange(900) / 60.0))
    for i in range(3):
        wav(root / 'music' / 'val-query-db-500-30s' / 'v' / f'{i}.wav',
            rng.integers(-1500, 1500, size=80000) + 8000 * np.sin(2 * np.pi * (350 + 300 * i) * t))
    c = copy.deepcopy(cfg)
    c['BSZ']['VAL_BATCH_SZ'], c['BSZ']['VAL_N_ANCHOR'] = 16, 8
    c['DIR'].update({'SOURCE_ROOT_DIR': str(root / 'music') + '/', 'BG_ROOT_DIR': str(root / 'aug' / 'bg') + '/',
                     'IR_ROOT_DIR': str(root / 'aug' / 'ir') + '/', 'LOG_ROOT_DIR': str(tmp_path) + '/logs/'})
    c['BSZ']['TR_BATCH_SZ'], c['BSZ']['TR_N_ANCHOR'] = 32, 16
    c['TRAIN']['MAX_EPOCH'] = 2
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        hist = T.trainer(c, 'dirs')
    assert buf.getvalue().count('val_loss:') == 2
    n_seg = 6 * 19                                             # 10-s clips: 19 segments each
    assert len(hist) == 2 and all(np.isfinite(hist))
    ck = torch.load(tmp_path / 'logs' / 'checkpoint' / 'dirs' / 'ckpt-2.pt', weights_only=True)
    assert ck['optimizer']['iterations'] == 2 * (n_seg // 16)
    assert hist[1] < hist[0]


@pytest.mark.parametrize('emb_sz', [64, 256])
def test_train_step_with_other_embedding_widths(nafp, cfg, emb_sz):
    """MODEL.EMB_SZ 64 / 256 (nnfp.py:250) through setup() / train_step(): encoder tail, NT-Xent and both backward
    passes at that width; the loss of the first step equals the oracle's NT-Xent on the step's embeddings and a few
    steps on a fixed batch descend."""
    from neural_audio_fp_amd.model import trainer as T
    from oracle import ntxent as o_nt
    c = copy.deepcopy(cfg)
    c['MODEL']['EMB_SZ'] = emb_sz
    c['BSZ']['TR_BATCH_SZ'], c['BSZ']['TR_N_ANCHOR'] = 32, 16
    c['TRAIN']['OPTIMIZER'], c['TRAIN']['LR'] = 'Adam', 1e-4
    m_pre, m_specaug, m_fp, opt, loss_obj, bucket = T.setup(c, 100)
    assert m_fp.emb_sz == emb_sz
    xa, xp = _pairs(16, 31)
    X = (torch.from_numpy(xa).cuda(), torch.from_numpy(xp).cuda())
    emb0 = m_fp(m_pre(torch.cat(X, 0))).cpu().numpy()
    assert emb0.shape == (32, emb_sz)
    losses = [float(T.train_step(X, m_pre, _NoAug(), m_fp, loss_obj, opt, bucket)[0]) for _ in range(8)]
    want = o_nt.compute_loss(emb0[:16], emb0[16:], c['LOSS']['TAU'])[0]
    assert abs(losses[0] - want) < 1e-4 * max(1.0, abs(want))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_prefetched_weight_repack_equals_the_lazy_one(nafp, cfg, monkeypatch):
    """`train_step` starts the re-pack of the updated weights on a stream of the handle's own right after the optimizer
    (`FingerPrinter.prefetch_weights`); the next passes wait INSIDE the library for the part of it they read (the training forward:
    plain copies -> conv0, layer 1's share -> conv1, everything -> conv2; any other pass: everything).  Five steps with the
    prefetch and five with the lazy re-pack at the next forward (same seeded data, no spec-augment) must give the same
    losses and variables to the rounding of the backward pass's atomics -- and a forward issued on ANOTHER stream right
    after a step must see the new weights, not the old blob."""
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import LAMB
    from neural_audio_fp_amd.model.fp.NTxent_loss_single_gpu import NTxentLoss
    n = 16
    xa, xp = _pairs(n, 9)
    X = (torch.from_numpy(xa).cuda(), torch.from_numpy(xp).cuda())
    m_pre = nafp.get_melspec_layer(cfg)
    feat = m_pre(torch.cat(X, 0))
    runs = {}
    for prefetch in (True, False):
        monkeypatch.setattr(T, '_PREFETCH_WEIGHTS', prefetch)
        m_fp = nafp.FingerPrinter(seed=0)
        m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=22)))
        opt = LAMB(learning_rate=1e-3)
        loss_obj = NTxentLoss(n_org=n, n_rep=n, tau=0.05)
        losses = []
        for step in range(5):
            loss, _ = T.train_step(X, m_pre, _NoAug(), m_fp, loss_obj, opt)
            losses.append(float(loss))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                     # straight after the step, on another stream
            emb_side = m_fp(feat).clone()
        side.synchronize()
        torch.cuda.synchronize()
        emb_main = m_fp(feat)
        assert torch.equal(emb_side, emb_main)            # the side-stream forward waited for the re-pack
        runs[prefetch] = (losses, [v.detach().clone() for v in m_fp.trainable_variables], emb_main.clone())
    la, lb = runs[True][0], runs[False][0]
    assert lb[-1] < lb[0]
    assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-4 * max(1.0, abs(lb[0]))
    for a, b in zip(runs[True][1], runs[False][1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
    assert float((runs[True][2] - runs[False][2]).abs().max()) < 1e-4
