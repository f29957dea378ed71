"""Spec-augment: hole distribution of the host mirror (CPU) and the masking kernel vs the oracle
(GPU) with injected rectangles (the TF random stream cannot be matched: SURVEY 8 a6)."""
import numpy as np
import pytest
import torch

from oracle import specaug as o_sa


def test_hole_draws_follow_reference_ranges():
    from neural_audio_fp_amd.model.fp.specaug_chain.specaug_chain import draw_holes
    rng = np.random.default_rng(0)
    H, W = 256, 32
    ws, hs = set(), set()
    for _ in range(3000):
        (f0, f1, t0, t1), = draw_holes('cutout', H, W, 1, rng)
        assert 0 <= f0 < f1 <= H - 1 and 0 <= t0 < t1 <= W - 1
        ws.add(t1 - t0); hs.add(f1 - f0)
    # width in [W//10, int(W/2.5)) = [3, 12) -> inclusive span 2*(w//2) in {2,...,10}; height in [25, 102)
    assert max(ws) <= 10 and max(hs) <= 2 * (101 // 2)
    for _ in range(500):
        (f0, f1, t0, t1), = draw_holes('horizontal', H, W, 3, rng)        # always exactly one full-width band
        assert (t0, t1) == (0, W - 1) and 2 <= f1 - f0 <= 2 * (19 // 2)   # clipped at the borders
        (f0, f1, t0, t1), = draw_holes('vertical', H, W, 1, rng)
        assert (f0, f1) == (0, H - 1) and 2 <= t1 - t0 <= 14


def test_oracle_mask_is_inclusive_and_respects_activation():
    x = np.ones((3, 8, 8, 1))
    y = o_sa.apply_holes(x, [(1, 2, 3, 5)], active=[True, False, True], fill=0.0)
    assert y[0, 1:3, 3:6].sum() == 0 and y[0].sum() == 64 - 6
    assert y[1].sum() == 64 and y[2].sum() == 64 - 6


@pytest.mark.gpu
def test_kernel_matches_oracle_with_injected_rects(nafp, cfg):
    from neural_audio_fp_amd.model.fp.specaug_chain.specaug_chain import get_specaug_chain_layer
    m = get_specaug_chain_layer(cfg)
    assert m.bypass is False and m.chain_config == ['cutout', 'horizontal'] and m.hole_fill == 'zeros'
    rng = np.random.default_rng(1)
    x = rng.normal(size=(5, 256, 32, 1)).astype(np.float32)
    rects = [(10, 80, 3, 9), (200, 212, 0, 31), (0, 0, 31, 31)]
    act = np.array([1, 0, 1, 1, 0], np.uint8)
    got = m.apply_rects(torch.from_numpy(x.copy()).cuda(), rects, torch.from_numpy(act).cuda(), fill=-0.5)
    assert np.array_equal(got.cpu().numpy(), o_sa.apply_holes(x, rects, act, -0.5).astype(np.float32))
    got2 = m.apply_rects(torch.from_numpy(x.copy()).cuda(), rects[:1], None, fill=0.0)
    assert np.array_equal(got2.cpu().numpy(), o_sa.apply_holes(x, rects[:1], None, 0.0).astype(np.float32))
    # the layer call: every sample gets the same holes, everything outside them is untouched
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    holes = (y == 0) & (x != 0)
    assert holes.any() and np.array_equal(holes[0], holes[4]) and np.array_equal(y[~holes], x[~holes])
    m.bypass = True
    xt = torch.from_numpy(x).cuda()
    assert m(xt) is xt
