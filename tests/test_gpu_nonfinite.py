"""Non-finite inputs fail like the reference.

keras `LayerNormalization` propagates a NaN / Inf of a sample to that sample's whole output (model/fp/nnfp.py:73-79, 223-231):
a NaN segment or a diverged parameter set gives NaN fingerprints, not plausible-looking finite ones.  The library's LayerNorm
statistics are 64-bit fixed-point sums (bit-reproducibility, csrc/nafp_common.h) and its GEMM epilogues use
max(t, exp(min(t, 0)) - 1) for the ELU (IEEE maxNum drops a NaN operand), so without the sticky per-sample poison flag a NaN
sample came back as finite garbage one layer later.  Checked here through the C ABI:

  * one poisoned sample -> its fingerprint row is NaN; every other row of the SAME launch is bit-equal to the clean launch;
  * activations out of the fixed-point range (weights scaled by 1e20) -> NaN rows, never a wrapped sum;
  * a NaN anywhere in the parameter set -> NaN rows (flag raised by set_weights, read by the tail);
  * the training forward and `front_conv` / `div_enc` behave the same way.
"""
import numpy as np
import pytest
import torch

import _inputs

pytestmark = pytest.mark.gpu


def _model(nafp, seed=7):
    m = nafp.FingerPrinter(seed=0)
    m.set_weights(_inputs.weight_list(_inputs.weights(seed=seed)))
    return m


def _feat(B, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')


@pytest.mark.parametrize('B', [9, 130, 640])
@pytest.mark.parametrize('what', ['nan_element', 'inf_element', 'nan_sample'])
def test_one_non_finite_sample_poisons_its_own_row_only(nafp, B, what):
    """B = 9: 128-row tiles with ragged sample groups; 130 / 640: the launch plans of the bench sizes (256-row tiles, 64-column
    tiles, split-K with the finish kernel and with the in-kernel finish)."""
    m = _model(nafp)
    feat = _feat(B, 100 + B)
    clean = m(feat).clone()
    assert bool(torch.isfinite(clean).all())
    bad_rows = sorted({1, B // 2, B - 1})
    dirty = feat.clone()
    for k, b in enumerate(bad_rows):
        if what == 'nan_sample':
            dirty[b] = float('nan')
        else:
            dirty[b, (37 * k + 5) % 256, (11 * k + 3) % 32, 0] = float('nan') if what == 'nan_element' else float('inf')
    emb = m(dirty)
    ok = torch.ones(B, dtype=torch.bool, device='cuda')
    ok[bad_rows] = False
    assert bool(torch.isnan(emb[~ok]).all()), emb[~ok]
    assert torch.equal(emb[ok], clean[ok])                     # same launch plan, independent rows: bit for bit
    if B <= 16:
        # ... and that is what the float64 restatement of the reference graph gives: LayerNormalization over (F, T, C) hands a NaN /
        # Inf to the whole sample and to nothing else (nnfp.py:73-79)
        from oracle import nnfp as o_nnfp
        with np.errstate(all='ignore'):
            want = o_nnfp.fingerprinter(dirty.cpu().numpy(), _inputs.weights(seed=7))
        assert np.isnan(want[bad_rows]).all() and np.isfinite(np.delete(want, bad_rows, axis=0)).all()
        assert float(np.abs(np.delete(emb.cpu().numpy(), bad_rows, axis=0) - np.delete(want, bad_rows, axis=0)).max()) < 5e-6
    # the flatten output and the training forward see the same poison
    flat = m.front_conv(dirty)
    assert bool(torch.isnan(flat[~ok]).all()) and bool(torch.isfinite(flat[ok]).all())
    emb_t = m.forward_train(dirty)
    assert bool(torch.isnan(emb_t[~ok]).all()) and bool(torch.isfinite(emb_t[ok]).all())
    # and the handle is not left poisoned: the next clean launch is the clean result again
    assert torch.equal(m(feat), clean)


def test_nan_segment_through_the_front_end_and_the_deferred_path(nafp, cfg):
    """A NaN in the AUDIO of one segment: its log-mel is NaN, the -80 dB clamp keeps it (max_keep_nan), conv0's statistics poison
    the sample.  The other segments of the max-normalisation group are unchanged: the group maximum (an atomic max over finite
    values) ignores the NaN segment -- where TensorFlow's reduce_max would hand the NaN to the whole device batch."""
    m_pre = nafp.get_melspec_layer(cfg)
    m = _model(nafp)
    x = torch.from_numpy(_inputs.audio(12, seed=5)).cuda()
    clean = m(m_pre(x, defer=True)).clone()
    x2 = x.clone()
    x2[4, 0, 1234] = float('nan')
    for defer in (True, False):
        emb = m(m_pre(x2, defer=defer))
        assert bool(torch.isnan(emb[4]).all())
        keep = [b for b in range(12) if b != 4]
        assert bool(torch.isfinite(emb[keep]).all())
        assert float((emb[keep] - clean[keep]).abs().max()) < 1e-6


def test_activations_beyond_the_fixed_point_range_come_back_nan(nafp):
    """Every conv kernel scaled by 1e20: the activations overflow float32 within two layers (keras: inf - inf = NaN inside
    LayerNormalization).  The fixed-point statistics must not wrap into a plausible mean / variance."""
    w = _inputs.weights(seed=7)
    for j in range(16):
        w[f'conv{j}.kernel'] = (w[f'conv{j}.kernel'].astype(np.float64) * 1e20).astype(np.float32)
    m = nafp.FingerPrinter(seed=0)
    m.set_weights(_inputs.weight_list(w))
    emb = m(_feat(33, 3))
    assert bool(torch.isnan(emb).all())
    # a moderately large scale stays finite and correct in direction: nothing is poisoned below the documented range
    w = _inputs.weights(seed=7)
    w['conv0.kernel'] = w['conv0.kernel'] * 50.0
    m.set_weights(_inputs.weight_list(w))
    assert bool(torch.isfinite(m(_feat(33, 3))).all())


@pytest.mark.parametrize('tensor', [0, 4 * 7, 4 * 15 + 1, 4 * 3 + 2, 4 * 9 + 3, 64, 67])
def test_a_nan_parameter_gives_nan_fingerprints(nafp, tensor):
    """conv0's kernel, a packed GEMM kernel, a bias, a LayerNorm scale / offset, the divide-and-encode weights: set_weights
    raises the handle's flag, the tail (and `div_enc`) write NaN; the next finite parameter set clears it."""
    w = _inputs.weight_list(_inputs.weights(seed=7))
    bad = [np.array(a, dtype=np.float32, copy=True) for a in w]
    bad[tensor].reshape(-1)[bad[tensor].size // 3] = np.nan
    m = nafp.FingerPrinter(seed=0)
    m.set_weights(bad)
    feat = _feat(6, 8)
    assert bool(torch.isnan(m(feat)).all())
    assert bool(torch.isnan(m.div_enc(torch.zeros((6, m.flat_dim), device='cuda'))).all())
    m.set_weights(w)
    assert bool(torch.isfinite(m(feat)).all())


def test_non_finite_samples_with_the_exact_split_option(nafp):
    """NAFP_OPT_BF16X3 = 2 changes the K-loop only; the statistics, the poison flag and the epilogues are the f32 path's: a NaN / Inf
    sample is a NaN row there too, the others are bit-equal to the clean launch of the same option."""
    m = _model(nafp)
    m.set_option(3, 2)
    B = 130
    feat = _feat(B, 77)
    clean = m(feat).clone()
    dirty = feat.clone()
    dirty[3, 10, 10, 0] = float('nan')
    dirty[64, 255, 31, 0] = float('-inf')
    emb = m(dirty)
    ok = torch.ones(B, dtype=torch.bool, device='cuda')
    ok[[3, 64]] = False
    assert bool(torch.isnan(emb[~ok]).all()) and torch.equal(emb[ok], clean[ok])
    m.set_option(3, 0)
