"""Oracle (TEST INFRASTRUCTURE): the in-training mini search test.  PARITY UNPINNED (TensorFlow absent;
the arithmetic is plain numpy here).

Follows model/utils/mini_search_subroutines.py:28-220: `pairwise_distances_for_eval` (squared L2 clipped at
0, or the dot product), `conv_eye_func` (sum over the length-s diagonal = 'valid' convolution with eye(s)),
`mini_search_eval` (argsort per query start, rank of the ground-truth id, top-1/3/10 accuracy in %, mean rank).
"""
import numpy as np


def pairwise(query, db, mode='argmin'):
    """query (nQ, d), db (nD, d) -> (nQ, nD)."""
    q, x = np.asarray(query, np.float64), np.asarray(db, np.float64)
    dot = q @ x.T
    if mode == 'argmax':
        return dot
    return np.maximum((q * q).sum(1)[:, None] + (x * x).sum(1)[None, :] - 2.0 * dot, 0.0)


def conv_eye(m, s):
    nq, nd = m.shape
    out = np.zeros((nq - s + 1, nd - s + 1))
    for i in range(s):
        out += m[i:i + nq - s + 1, i:i + nd - s + 1]
    return out


def mini_search_eval(query, db, scopes=(1, 3, 5, 9, 11, 19), mode='argmin', gt_id_offset=0):
    """query (nQ, nAug, d).  Returns ((top1, top3, top10) in %, mean_rank), arrays over the scopes."""
    query = np.asarray(query)
    n_augs = query.shape[1]
    top = np.zeros((3, len(scopes))); mean_rank = np.zeros(len(scopes))
    mats = [pairwise(query[:, a], db, mode) for a in range(n_augs)]
    for i, s in enumerate(scopes):
        conv = np.stack([conv_eye(m, s) for m in mats])                  # (n_augs, n_targets, n_db')
        order = np.argsort(conv, axis=2, kind='stable')
        if mode == 'argmax':
            order = order[:, :, ::-1]
        n_targets = conv.shape[1]
        ranks = np.zeros((n_augs, n_targets))
        for t in range(n_targets):
            for a in range(n_augs):
                ranks[a, t] = np.where(order[a, t] == t + gt_id_offset)[0][0]
        mean_rank[i] = ranks.mean()
        for j, k in enumerate((1, 3, 10)):
            top[j, i] = 100.0 * (ranks < k).mean()
    return (top[0], top[1], top[2]), mean_rank
