"""Consumers of the REFERENCE-PRODUCED fixtures that tools/gen_golden_tf.py writes inside the reference's TensorFlow
environment (tests/golden/hotpath_tf_v1.npz, tests/golden/tf_ckpt_tiny/, tests/golden/tf_ckpt_tiny_expected.npz).

The build image has no TensorFlow, so the files do not exist here and every test below SKIPS; the moment a holder of
a TF environment runs the recipe and commits the data, they pin
  * the oracle (CPU)      -- log-mel, flat, fingerprint, NT-Xent loss / sim_mtx / gradients, train-step gradients,
                             Adam / LAMB updates against what TensorFlow / kapre computed,
  * the HIP path (-m gpu) -- the same quantities through the C ABI,
  * model/utils/tf_checkpoint.py -- against a bundle TensorFlow itself wrote.
Tolerances: the fixture is float32 TensorFlow arithmetic, the oracle float64: |d log-mel| < 1e-4 (log10 of fp32 FFT
magnitudes), cosine >= 1 - 1e-3 per fingerprint (the north-star contract) and |d emb| < 1e-4, loss 1e-4 relative,
gradients 1e-3 of each tensor's largest entry.
"""
import hashlib
import os

import numpy as np
import pytest

import _inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.environ.get('NAFP_TF_GOLDEN_DIR') or os.path.join(ROOT, 'tests', 'golden')      # (the env override lets tests/_tf_fixture_standin.py dry-run these consumers)
HOT = os.path.join(GOLD, 'hotpath_tf_v1.npz')
CKPT = os.path.join(GOLD, 'tf_ckpt_tiny', 'ckpt-1')
CKPT_EXPECTED = os.path.join(GOLD, 'tf_ckpt_tiny_expected.npz')

needs_hot = pytest.mark.skipif(not os.path.exists(HOT), reason='tests/golden/hotpath_tf_v1.npz absent: run tools/gen_golden_tf.py '
                               'inside the reference\'s TensorFlow environment (none in this image)')
needs_ckpt = pytest.mark.skipif(not (os.path.exists(CKPT + '.index') and os.path.exists(CKPT_EXPECTED)),
                                reason='tests/golden/tf_ckpt_tiny absent: run tools/gen_golden_tf.py inside the reference\'s '
                                       'TensorFlow environment (none in this image)')

# library tensor name (nnfp.tensor_names()) -> oracle weight name
def _oracle_key(name):
    if name.startswith('div_enc.'):
        return {'div_enc.fc1.kernel': 'div.w1', 'div_enc.fc1.bias': 'div.b1', 'div_enc.fc2.kernel': 'div.w2', 'div_enc.fc2.bias': 'div.b2'}[name]
    _, blk, layer, kind = name.split('.')
    j = 2 * int(blk) + (1 if layer.endswith('3x1') else 0)
    return (f'conv{j}.' if layer.startswith('conv') else f'ln{j}.') + kind


def _sha(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode()); h.update(np.ascontiguousarray(w[k], dtype='<f4').tobytes())
    return h.hexdigest()


@pytest.fixture(scope='module')
def tfg():
    z = dict(np.load(HOT))
    w = _inputs.weights(seed=3)
    if _sha(w) != str(z['weights_sha256']):
        pytest.skip('the numpy that wrote the fixture drew other seeded weights than this numpy does (Generator stream): '
                    'regenerate the fixture with a matching numpy')
    assert np.array_equal(z['audio_seed11'], _inputs.audio(4, seed=11))
    return z, w


# ---------------------------------------------------------------- oracle vs TensorFlow (CPU) ----
@needs_hot
def test_oracle_front_end_matches_tensorflow(tfg):
    from oracle import melspec as o_mel
    z, _ = tfg
    x = z['audio_seed11']
    assert np.abs(o_mel.melspec_layer(x, dtype=np.float64) - z['mel_seed11']).max() < 1e-4          # melspectrogram.py:102-112
    assert np.abs(o_mel.melspec_layer(x[:2], dtype=np.float64) - z['mel_seed11_first2']).max() < 1e-4


@needs_hot
def test_oracle_encoder_matches_tensorflow(tfg):
    from oracle import nnfp as o_nnfp
    z, w = tfg
    feat = z['mel_seed11'].astype(np.float64)
    flat = o_nnfp.front_conv(feat, w, dtype=np.float64)
    emb = o_nnfp.l2_normalize(o_nnfp.div_enc(flat, w))
    assert np.abs(flat - z['flat_seed11_w3']).max() < 1e-3
    assert (1 - (emb * z['emb_seed11_w3']).sum(1)).max() < 1e-5                                       # contract 1e-3
    assert np.abs(emb - z['emb_seed11_w3']).max() < 1e-4


@needs_hot
def test_oracle_ntxent_matches_tensorflow(tfg):
    from oracle import ntxent as o_nt
    z, _ = tfg
    for tag in ('n5', 'n60', 'n5_hard'):
        a, b = z[f'ntxent_a_{tag}'], z[f'ntxent_b_{tag}']
        loss, sim, _ = o_nt.compute_loss(a, b, 0.05)
        assert abs(loss - float(z[f'ntxent_loss_{tag}'][0])) < 1e-4 * max(1.0, abs(loss)), tag
        if f'ntxent_sim_{tag}' in z:
            assert np.abs(sim - z[f'ntxent_sim_{tag}']).max() < 1e-4
            ga, gb = o_nt.grad_embeddings(a, b, 0.05)
            for got, want in ((ga, z[f'ntxent_grad_a_{tag}']), (gb, z[f'ntxent_grad_b_{tag}'])):
                assert np.abs(got - want).max() < 1e-3 * (np.abs(want).max() + 1e-12)


@needs_hot
def test_oracle_train_step_matches_tensorflow(tfg):
    """tape.gradient of trainer.py:43-47 and one Adam / LAMB step vs float64 autograd over oracle/torch_ref.py and
    oracle/optim.py."""
    import torch
    from oracle import ntxent as o_nt, optim as o_opt, torch_ref
    z, w = tfg
    tf_ = torch_ref.TorchFingerprinter(w, dtype=torch.float64, requires_grad=True)
    emb = tf_(torch.from_numpy(z['mel_seed11']).double())
    e = emb.detach().numpy()
    assert abs(o_nt.compute_loss(e[:2], e[2:], 0.05)[0] - float(z['train_loss'][0])) < 1e-4
    d_a, d_b = o_nt.grad_embeddings(e[:2], e[2:], 0.05)
    (emb * torch.from_numpy(np.concatenate([d_a, d_b]))).sum().backward()
    names = __import__('neural_audio_fp_amd').model.fp.nnfp.tensor_names()
    for name, p in zip(names, tf_.params):
        g = p.grad.numpy()
        assert abs(np.sqrt((g ** 2).sum()) - float(z['gradnorm.' + name][0])) < 1e-3 * float(z['gradnorm.' + name][0]) + 1e-9, name
        if 'grad.' + name in z:
            want = z['grad.' + name]
            assert np.abs(g.reshape(want.shape) - want).max() < 1e-3 * (np.abs(want).max() + 1e-12), name
            if name.endswith('bias'):
                w0 = w[_oracle_key(name)].astype(np.float64).reshape(want.shape)
                g64 = want.astype(np.float64)
                got = o_opt.adam_step(w0, g64, np.zeros_like(w0), np.zeros_like(w0), 1e-4, 1)[0]
                assert np.abs(got - z['adam_step1.' + name]).max() < 2e-7, name
                if w0.ndim == 1:                                                 # one keras variable = one LAMB trust ratio
                    got = o_opt.lamb_step(w0, g64, np.zeros_like(w0), np.zeros_like(w0), 1e-3, 1)[0]
                    assert np.abs(got - z['lamb_step1.' + name]).max() < 2e-6, name


# ---------------------------------------------------------------- TensorBundle reader vs a TF-written bundle ----
@needs_ckpt
def test_reader_reads_a_bundle_written_by_tensorflow():
    from neural_audio_fp_amd.model.utils import tf_checkpoint as tfc
    from neural_audio_fp_amd.model.fp.nnfp import tensor_names
    exp = dict(np.load(CKPT_EXPECTED))
    emb_sz = int(exp.pop('emb_sz')[0])
    names = tensor_names()
    assert sorted(exp) == sorted(names)
    sd = tfc.state_dict_from_tf_checkpoint(CKPT, names, [exp[n].shape for n in names], emb_sz)
    for n in names:
        assert np.array_equal(sd[n], exp[n]), n                                  # bit for bit: the reader copies bytes


# ---------------------------------------------------------------- HIP path vs TensorFlow (GPU) ----
@needs_hot
@pytest.mark.gpu
def test_hip_path_matches_tensorflow(nafp, cfg, tfg):
    import torch
    z, w = tfg
    m_pre, m_fp = nafp.get_melspec_layer(cfg), nafp.get_fingerprinter(cfg)
    m_fp.set_weights(_inputs.weight_list(w))
    x = torch.from_numpy(z['audio_seed11']).cuda()
    mel = m_pre(x)
    assert np.abs(mel.cpu().numpy() - z['mel_seed11']).max() < 1e-4
    emb = m_fp(mel).cpu().numpy()
    assert (1 - (emb * z['emb_seed11_w3']).sum(1)).max() < 1e-5                                       # contract 1e-3
    assert np.abs(emb - z['emb_seed11_w3']).max() < 1e-4
    assert np.abs(m_fp.front_conv(torch.from_numpy(z['mel_seed11']).cuda()).cpu().numpy() - z['flat_seed11_w3']).max() < 1e-3
    for tag in ('n5', 'n60', 'n5_hard'):
        a, b = torch.from_numpy(z[f'ntxent_a_{tag}']).cuda(), torch.from_numpy(z[f'ntxent_b_{tag}']).cuda()
        loss, sim, _ = nafp.NTxentLoss(len(a), len(b), 0.05).compute_loss(a, b)
        assert abs(float(loss) - float(z[f'ntxent_loss_{tag}'][0])) < 1e-4 * max(1.0, abs(float(loss))), tag
        if f'ntxent_sim_{tag}' in z:
            assert np.abs(sim.cpu().numpy() - z[f'ntxent_sim_{tag}']).max() < 1e-3


# ---------------------------------------------------------------- the consumers themselves ----
@pytest.mark.skipif(bool(os.environ.get('NAFP_TF_GOLDEN_DIR')), reason='already inside the dry run')
def test_consumers_execute_on_an_oracle_written_stand_in(tmp_path):
    """Keeps the consumers above from rotting while no TF-written fixture exists: tests/_tf_fixture_standin.py writes a
    file with the recipe's keys and shapes FROM THE ORACLE into a scratch directory and the CPU consumers must run green
    on it (oracle vs oracle: proves the plumbing, nothing about parity)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_tf_fixture_standin.py'), str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_golden_tf.py'), '-q', '-m', 'not gpu',
                        '-p', 'no:cacheprovider'], env=dict(os.environ, NAFP_TF_GOLDEN_DIR=str(tmp_path)),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and '4 passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
