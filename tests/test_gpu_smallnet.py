"""NAFP_OPT_SMALLNET: the small layers (convs 10-15 + nothing else) as ONE persistent launch with an in-order work queue
(csrc/conv.hip, smallnet_kernel).  Off by default -- measured slower than the per-layer launches, DESIGN.md 4.2 [r5] -- but
built to be safe to ship: the tests hold it to the same oracle, to run-to-run bit equality, and to the property the design
argument promises: several launches resident at once (four generate streams, ragged batches) neither hang nor interfere.
Reference path: model/fp/nnfp.py:193-197, 210-218."""
import numpy as np
import pytest
import torch

from oracle import nnfp as o_nnfp
import _inputs

pytestmark = pytest.mark.gpu


def _model(nafp, persistent, seed=4):
    m = nafp.FingerPrinter(seed=0)
    m.set_option(5, 1 if persistent else 0)
    m.set_weights(_inputs.weight_list(_inputs.weights(seed=seed)))
    return m


def _feat(B, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return -1.2 * torch.rand((B, 256, 32, 1), generator=g, device='cuda')


@pytest.mark.parametrize('B', [1, 9, 130, 257, 640])
def test_persistent_launch_matches_the_per_layer_launches_and_the_oracle(nafp, B, observe):
    """B = 1 / 9: one ragged 128-sample group; 130 / 257: a full group plus a nearly empty one; 640: the bench batch."""
    feat = _feat(B, 300 + B)
    ref = _model(nafp, False)(feat)
    m = _model(nafp, True)
    got = m(feat)
    assert bool(torch.isfinite(got).all())
    observe('persistent vs per-layer launches, |d emb|', float((got - ref).abs().max()), 5e-6)
    assert torch.equal(m(feat), got)                              # run to run: bit for bit
    flat = m.front_conv(feat)
    observe('persistent vs per-layer launches, |d flat|', float((flat - _model(nafp, False).front_conv(feat)).abs().max()), 5e-5)
    if B <= 9:
        w = _inputs.weights(seed=4)
        want = o_nnfp.fingerprinter(feat.cpu().numpy(), w)
        observe('persistent launch vs oracle, |d emb|', float(np.abs(got.cpu().numpy() - want).max()), 5e-6)


@pytest.mark.timeout(300)
def test_four_streams_with_ragged_batches_finish_and_equal_the_single_stream_result(nafp):
    """Four generate streams share the chip: each launch's workgroups claim their own items in order, so no launch can be
    starved into a deadlock by another one's resident workgroups -- the run ENDS (pytest-timeout / gpurun's limit would catch a
    hang), no launch gave up a wait (no NaN row), and every result is bit-equal to the same batch alone on one stream."""
    m = _model(nafp, True)
    sizes = [640, 125, 333, 1, 640, 64, 513, 250]
    feats = [_feat(b, 900 + i) for i, b in enumerate(sizes)]
    alone = [m(f).clone() for f in feats]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    for rep in range(3):
        outs = [None] * len(feats)
        for i, f in enumerate(feats):
            with torch.cuda.stream(streams[i % 4]):
                outs[i] = m(f)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(outs, alone)):
            assert bool(torch.isfinite(a).all()), (rep, i)
            assert torch.equal(a, b), (rep, i, sizes[i])


def test_persistent_launch_in_the_training_forward_and_non_finite_samples(nafp):
    """forward_train takes the same launch (z only); a poisoned sample stays poisoned across the in-launch layers."""
    B = 160
    feat = _feat(B, 77)
    d_emb = torch.randn((B, 128), device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
    ref_m, m = _model(nafp, False), _model(nafp, True)
    e0, e1 = ref_m.forward_train(feat), m.forward_train(feat)
    assert float((e0 - e1).abs().max()) < 5e-6
    g0 = [t.clone() for t in ref_m.backward(d_emb)]
    g1 = m.backward(d_emb)
    for i, (a, b) in enumerate(zip(g1, g0)):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12, i
    dirty = feat.clone()
    dirty[17, 3, 5, 0] = float('nan')
    emb = m(dirty)
    assert bool(torch.isnan(emb[17]).all())
    keep = [b for b in range(B) if b != 17]
    assert torch.equal(emb[keep], m(feat)[keep])
