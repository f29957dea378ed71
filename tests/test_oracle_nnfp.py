"""CPU: the encoder oracle against known answers and independent formulations
(model/fp/nnfp.py:20-231)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nnfp as o_nnfp, torch_ref
import _inputs


def test_param_count_known_answers():
    # the one numeric known-answer the reference holds for this path: nnfp.py:271
    assert o_nnfp.count_params((256, 63, 1))['total'] == 19224576
    c = o_nnfp.count_params((256, 32, 1))
    assert c == {'conv': 14606208, 'ln': 2291712, 'divenc': 41088, 'total': 16939008}


def test_geometry_matches_survey_appendix_a():
    geo = o_nnfp.conv_geometry()
    outs = [g['out'] for g in geo]
    assert outs == [(256, 16, 128), (128, 16, 128), (128, 8, 128), (64, 8, 128), (64, 4, 256), (32, 4, 256),
                    (32, 2, 256), (16, 2, 256), (16, 2, 512), (8, 2, 512), (8, 1, 512), (4, 1, 512),
                    (4, 1, 1024), (2, 1, 1024), (2, 1, 1024), (1, 1, 1024)]
    pads = [g['pad'] for g in geo]
    # TF SAME: stride 2 on an even size pads 0 before / 1 after; stride 1 or size 1 pads 1/1
    assert pads[0] == (0, 1) and pads[1] == (0, 1) and pads[8] == (1, 1) and pads[12] == (1, 1) and pads[14] == (1, 1)
    assert o_nnfp.same_pad(32, 3, 2) == (16, 0, 1) and o_nnfp.same_pad(1, 3, 2) == (1, 1, 1)
    assert o_nnfp.same_pad(63, 3, 2) == (32, 1, 1)


def test_conv_same_vs_torch_conv2d():
    rng = np.random.default_rng(0)
    for axis, stride, shape in [('T', (1, 2), (2, 6, 8, 5)), ('F', (2, 1), (2, 8, 3, 5)), ('T', (1, 1), (1, 4, 2, 3)),
                                ('T', (1, 2), (1, 4, 1, 3)), ('F', (2, 1), (1, 1, 1, 3))]:
        x = rng.normal(size=shape)
        cin, cout = shape[3], 7
        k = rng.normal(size=((1, 3) if axis == 'T' else (3, 1)) + (cin, cout))
        b = rng.normal(size=cout)
        n_in = shape[2] if axis == 'T' else shape[1]
        s = stride[1] if axis == 'T' else stride[0]
        _, pb, pa = o_nnfp.same_pad(n_in, 3, s)
        got = o_nnfp.conv_same(x, k, b, axis, stride, (pb, pa))
        xt = torch.from_numpy(x).permute(0, 3, 1, 2)
        xt = F.pad(xt, (pb, pa, 0, 0) if axis == 'T' else (0, 0, pb, pa))
        want = F.conv2d(xt, torch.from_numpy(k).permute(3, 2, 0, 1), torch.from_numpy(b), stride=stride)
        assert np.abs(got - want.permute(0, 2, 3, 1).numpy()).max() < 1e-12


def test_layer_norm_vs_torch():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, 4, 5, 6)); g = rng.normal(size=(4, 5, 6)); b = rng.normal(size=(4, 5, 6))
    want = F.layer_norm(torch.from_numpy(x), (4, 5, 6), torch.from_numpy(g), torch.from_numpy(b), eps=1e-3)
    assert np.abs(o_nnfp.layer_norm(x, g, b) - want.numpy()).max() < 1e-12


def test_elu_and_l2():
    x = np.array([-3., -1e-8, 0., 2.])
    assert np.allclose(o_nnfp.elu(x), F.elu(torch.from_numpy(x)).numpy())
    v = np.array([[3., 4.], [0., 0.]])
    out = o_nnfp.l2_normalize(v)
    assert np.allclose(out[0], [0.6, 0.8]) and np.all(out[1] == 0)      # eps 1e-12 guards the zero row


def test_fingerprinter_vs_torch_formulation():
    rng = np.random.default_rng(2)
    feat = -rng.uniform(0, 1.2, size=(3, 256, 32, 1))
    w = _inputs.weights(seed=5)
    flat = o_nnfp.front_conv(feat, w)
    emb = o_nnfp.fingerprinter(feat, w)
    tf = torch_ref.TorchFingerprinter(w)
    ft = torch.from_numpy(feat.astype(np.float32))
    assert np.abs(flat - tf.front_conv(ft).numpy()).max() < 2e-5       # torch runs in float32
    assert np.abs(emb - tf(ft).numpy()).max() < 2e-6
    assert np.allclose(np.linalg.norm(emb, axis=1), 1.0)
    emb32 = o_nnfp.fingerprinter(feat.astype(np.float32), w, dtype=np.float32)
    assert np.abs(emb32 - emb).max() < 5e-6                            # rounding bound f32 vs f64


def test_div_enc_slices_are_contiguous_channel_groups():
    w = _inputs.weights(seed=6)
    x = np.zeros((1, 1024)); x[0, 8 * 5:8 * 6] = 1.0                   # only slice q=5 sees non-zero input
    y = o_nnfp.div_enc(x, w)
    y0 = o_nnfp.div_enc(np.zeros((1, 1024)), w)
    changed = np.nonzero(np.abs(y - y0)[0] > 1e-12)[0]
    assert list(changed) == [5]


def test_two_second_input_builds():
    # nnfp.py:266-268 builds the model on (256,63,1) too
    w = o_nnfp.init_weights(seed=1, input_shape=(256, 63, 1))
    assert sum(v.size for v in w.values()) == 19224576
    feat = -np.random.default_rng(3).uniform(0, 1, size=(1, 256, 63, 1))
    assert o_nnfp.fingerprinter(feat, w).shape == (1, 128)


def test_golden_encoder(golden):
    x = _inputs.audio(4, seed=11)
    from oracle import melspec as o_mel
    feat = o_mel.melspec_layer(x)
    w = _inputs.weights(seed=3)
    taps = []
    flat = o_nnfp.front_conv(feat, w, taps=taps)
    assert np.abs(flat - golden['flat_seed11_w3']).max() < 1e-5
    assert np.abs(o_nnfp.l2_normalize(o_nnfp.div_enc(flat, w)) - golden['emb_seed11_w3']).max() < 1e-6
    assert np.allclose([t.mean() for t in taps], golden['ln_out_mean'], atol=1e-9)
    assert np.allclose([np.abs(t).mean() for t in taps], golden['ln_out_absmean'], atol=1e-9)


def test_non_finite_element_poisons_exactly_its_own_sample():
    """What the HIP path is held to in tests/test_gpu_nonfinite.py: LayerNormalization over (F, T, C) (nnfp.py:73-79) hands a
    NaN or an Inf of one sample to that sample's whole fingerprint, and to no other sample of the batch."""
    import _inputs
    rng = np.random.default_rng(5)
    feat = (-rng.uniform(0, 1.2, size=(3, 256, 32, 1))).astype(np.float32)
    w = _inputs.weights(seed=7)
    clean = o_nnfp.fingerprinter(feat, w)
    for bad in (np.nan, np.inf, -np.inf):
        dirty = feat.copy()
        dirty[1, 17, 5, 0] = bad
        with np.errstate(all='ignore'):
            got = o_nnfp.fingerprinter(dirty, w)
        assert np.isnan(got[1]).all()
        assert np.array_equal(got[[0, 2]], clean[[0, 2]])


@pytest.mark.parametrize('norm', ['layer_norm1d', 'batch_norm'])
def test_norm_alternates_numpy_vs_torch(norm):
    """MODEL.BN alternates (nnfp.py:63-71): the numpy restatement against torch's own layer_norm / the affine map, float64."""
    import torch
    from oracle import torch_ref
    w = o_nnfp.convert_norm(o_nnfp.init_weights(seed=2, randomize_affine=True), norm, seed=5)
    rng = np.random.default_rng(0)
    feat = -rng.uniform(0, 1.2, size=(2, 256, 32, 1))
    a = o_nnfp.fingerprinter(feat, w, norm=norm)
    b = torch_ref.TorchFingerprinter(w, dtype=torch.float64, norm=norm)(torch.tensor(feat)).numpy()
    assert np.abs(a - b).max() < 1e-10
    assert np.abs(a - o_nnfp.fingerprinter(feat, o_nnfp.init_weights(seed=2, randomize_affine=True))).max() > 1e-3   # (a different model)
    if norm == 'batch_norm':
        # as keras initialises it (moving mean 0, variance 1, gamma 1, beta 0) the layer is a division by sqrt(1.001)
        w0 = o_nnfp.convert_norm(o_nnfp.init_weights(seed=2), norm, randomize=False)
        x = rng.normal(size=(1, 4, 2, 128))
        assert np.allclose(o_nnfp.apply_norm(x, w0, 0, norm, np.float64), x / np.sqrt(1.001), rtol=0, atol=1e-15)
    else:
        x = rng.normal(size=(3, 4, 2, 128))
        y = o_nnfp.layer_norm1d(x, np.ones(128), np.zeros(128))
        assert np.abs(y.mean(-1)).max() < 1e-12 and np.abs((y ** 2).mean(-1) - x.var(-1) / (x.var(-1) + 1e-3)).max() < 1e-12
