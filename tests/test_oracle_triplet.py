"""CPU: the online-triplet oracle (oracle/triplet.py = online_triplet_loss.py:98-239): numpy vs the torch
formulation, mask structure, hand-checked cases."""
import numpy as np
import torch

from oracle import triplet as T


def _unit(n, d, seed):
    x = np.random.default_rng(seed).normal(size=(n, d))
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def test_masks_follow_the_reference_layout():
    ap, an = T.masks(3, 2)
    assert ap.shape == an.shape == (3, 9)
    assert ap[1].tolist() == [0, 0, 1, 1, 0, 0, 0, 0, 0]
    assert an[1].tolist() == [1, 1, 0, 0, 1, 1, 1, 0, 1]          # own replicas and the anchor itself are not negatives
    assert (ap.sum(1) == 2).all() and (an.sum(1) == 9 - 3).all()


def test_numpy_and_torch_formulations_agree():
    a, p = _unit(5, 16, 0), _unit(15, 16, 1)
    for mode, margin in (('semi-hard', 0.4), ('all', 0.0), ('all', 0.3), ('all-balanced', 0.2), ('hardest', 0.1)):
        l, d, act = T.compute_loss(a, p, mode, margin)
        lt = T.torch_loss(torch.tensor(a), torch.tensor(p), mode, margin)
        assert abs(l - float(lt)) < 1e-12 and d.shape == (5, 20) and act == float(l > 0)
    # identical anchor and replicas: distances of the positive pairs are sqrt(EPS), of the anchor with itself too
    d = T.pairwise_dist(a, np.repeat(a, 3, axis=0))
    assert np.allclose(d[np.arange(5), 15 + np.arange(5)], np.sqrt(T.EPS), atol=1e-7)


def test_balanced_and_hardest_hand_checked():
    """'all-balanced' = per-anchor mean positive minus mean negative distance; 'hardest' as the reference writes it:
    the min runs over the MASKED matrix (online_triplet_loss.py:225), so the hardest-negative term is 0."""
    a, p = _unit(4, 8, 2), _unit(8, 8, 3)
    d = T.pairwise_dist(a, p)
    ap, an = T.masks(4, 2)
    want = np.mean([max(d[i][ap[i] > 0].mean() - d[i][an[i] > 0].mean() + 0.3, 0.) for i in range(4)])
    assert abs(T.compute_loss(a, p, 'all-balanced', 0.3)[0] - want) < 1e-12
    want_h = np.mean([d[i][ap[i] > 0].max() + 0.1 for i in range(4)])
    assert abs(T.compute_loss(a, p, 'hardest', 0.1)[0] - want_h) < 1e-12
