"""Oracle (TEST INFRASTRUCTURE): online triplet loss of the now-playing baseline.  PARITY UNPINNED
(TensorFlow absent); the arithmetic restated here is plain matrix algebra.

Follows model/fp/online_triplet_loss.py:34-244: masks `_get_anchor_positive_mask_v2` /
`_get_anchor_negative_mask_v2` (:98-121), `_pairwise_distances_v2_fast` (:185-196: d = sqrt(2(1-a.p)
* [2(1-a.p) > 0] + 1e-9)), and `compute_loss` (:199-239) in all four modes -- 'semi-hard' (training), 'all'
(validation), 'all-balanced', 'hardest' -- with use_anc_as_pos=True (the positives matrix is [emb_pos ; emb_anchor]).

'hardest' is restated AS WRITTEN: `tf.reduce_min(pairwise_dist * self.an_mask, axis=1)` (:225) runs over the masked
matrix, whose entries at the anchor's replicas and at the anchor itself are 0 while every distance is > 0, so the
reference's hardest-negative distance is identically 0."""
import numpy as np

EPS = 1e-9


def masks(n_anchor, n_pos_per_anchor):
    n_pos = n_anchor * n_pos_per_anchor
    ap = np.zeros((n_anchor, n_pos + n_anchor))
    for a in range(n_anchor):
        ap[a, a * n_pos_per_anchor:(a + 1) * n_pos_per_anchor] = 1
    an = 1 - np.concatenate([ap[:, :n_pos], np.eye(n_anchor)], axis=1)
    return ap, an


def pairwise_dist(emb_anc, emb_pos):
    cols = np.concatenate([emb_pos, emb_anc], 0)
    d = 2.0 * (1 - emb_anc @ cols.T)
    return np.sqrt(d * (d > 0) + EPS)


def compute_loss(emb_anc, emb_pos, mode='semi-hard', margin=0.5):
    """(loss, pairwise_dist, num_active_triplets) (online_triplet_loss.py:199-239)."""
    emb_anc, emb_pos = np.asarray(emb_anc, np.float64), np.asarray(emb_pos, np.float64)
    nA = len(emb_anc)
    ap, an = masks(nA, len(emb_pos) // nA)
    d = pairwise_dist(emb_anc, emb_pos)
    ap_d = d * ap
    if mode == 'all':
        loss = np.maximum(ap_d - d * an + margin, 0.).mean()
    elif mode == 'semi-hard':
        hardest = ap_d.max(axis=1, keepdims=True) * np.ones((1, d.shape[1]))
        loss = np.maximum((hardest - d + margin) * an, 0.).mean()
    elif mode == 'all-balanced':                                            # :215-222
        ap_m = ap_d.sum(axis=1) / ap.sum(axis=1)
        an_m = (d * an).sum(axis=1) / an.sum(axis=1)
        loss = np.maximum(ap_m - an_m + margin, 0.).mean()
    elif mode == 'hardest':                                                 # :223-227
        loss = np.maximum(ap_d.max(axis=1) - (d * an).min(axis=1) + margin, 0.).mean()
    else:
        raise NotImplementedError(mode)
    return loss, d, float(loss > 0)


def torch_loss(emb_anc, emb_pos, mode='semi-hard', margin=0.5):
    """the same graph in torch float64 (autograd reference for the gradients)."""
    import torch
    nA = emb_anc.shape[0]
    ap, an = (torch.as_tensor(m) for m in masks(nA, emb_pos.shape[0] // nA))
    cols = torch.cat([emb_pos, emb_anc], 0)
    d = 2.0 * (1 - emb_anc @ cols.T)
    d = torch.sqrt(d * (d > 0) + EPS)
    if mode == 'all':
        return torch.clamp(d * ap - d * an + margin, min=0.).mean()
    if mode == 'all-balanced':
        return torch.clamp((d * ap).sum(1) / ap.sum(1) - (d * an).sum(1) / an.sum(1) + margin, min=0.).mean()
    if mode == 'hardest':
        return torch.clamp((d * ap).max(dim=1).values - (d * an).min(dim=1).values + margin, min=0.).mean()
    hardest = (d * ap).max(dim=1, keepdim=True).values
    return torch.clamp((hardest - d + margin) * an, min=0.).mean()
