"""Oracle (TEST INFRASTRUCTURE): segment/sequence search and its evaluation.  PARITY UNPINNED
(see oracle/__init__.py): faiss is absent from the reference tree and from this image.

Follows eval/eval_faiss.py:115-289 with the exact index (`get_index('L2')` = faiss.IndexFlatL2,
eval/utils/get_index_faiss.py:57-62):
  * index = [dummy_db ; db]; ground truth of query row i is i + len(dummy_db)      (:118-138, :188)
  * per test id and sequence length sl: q = query[t : t+sl]; top-k_probe search per segment;
    candidate start ids = I[offset] - offset, >= 0, unique                          (:206-219)
  * score(c) = mean_i q[i] . index[c+i] over the rows that exist (np.diag of a possibly
    non-square product when c+sl runs past the end)                                 (:221-230)
  * pred_ids = candidates[argsort(-score)[:10]]; top1 exact / near (+-1), top3, top10 (:232-246)
Third-party semantics restated: IndexFlatL2.search returns the k smallest squared L2 distances
|q|^2 + |x|^2 - 2 q.x.  Ties (duplicate vectors) have no specified order in faiss; here the smaller
id wins, in the search and in the final ranking -- the HIP path uses the same rule.
"""
import numpy as np


def flat_l2_search(q, index, k, dtype=np.float64):
    """(distances, ids) of the k nearest index rows per query row, nearest first."""
    q = np.asarray(q, dtype=dtype)
    x = np.asarray(index, dtype=dtype)
    d = (q * q).sum(1)[:, None] + (x * x).sum(1)[None, :] - 2.0 * q @ x.T
    k = min(k, x.shape[0])
    order = np.lexsort((np.broadcast_to(np.arange(x.shape[0]), d.shape), d), axis=1)[:, :k]
    return np.take_along_axis(d, order, 1), order.astype(np.int64)


def sequence_candidates(I):
    """Offset compensation + unique (eval_faiss.py:213-219)."""
    I = np.asarray(I).copy()
    for offset in range(len(I)):
        I[offset, :] -= offset
    return np.unique(I[np.where(I >= 0)])


def sequence_score(q, index, cid, dtype=np.float64):
    sl = len(q)
    rows = np.asarray(index[cid:cid + sl], dtype=dtype)
    return float(np.mean(np.diag(np.dot(np.asarray(q, dtype=dtype), rows.T))))


def rank_candidates(candidates, scores, n=10):
    order = np.lexsort((candidates, -np.asarray(scores)))        # score descending, then smaller id
    return np.asarray(candidates)[order[:n]]


def evaluate(query, db, dummy_db, test_ids, test_seq_len=(1, 3, 5, 9, 11, 19), k_probe=20, dtype=np.float64):
    """Returns (top1_exact, top1_near, top3_exact, top10_exact), each (n_test, len(test_seq_len)) int,
    and the predicted ids (n_test, n_len, 10) padded with -1."""
    index = np.concatenate([np.asarray(dummy_db), np.asarray(db)], 0)
    n_dummy = len(dummy_db)
    n_test, n_len = len(test_ids), len(test_seq_len)
    out = [np.zeros((n_test, n_len), int) for _ in range(4)]
    preds = -np.ones((n_test, n_len, 10), np.int64)
    for ti, t in enumerate(test_ids):
        gt = t + n_dummy
        for si, sl in enumerate(test_seq_len):
            q = query[t:t + sl]
            _, I = flat_l2_search(q, index, k_probe, dtype)
            cand = sequence_candidates(I)
            scores = [sequence_score(q, index, c, dtype) for c in cand]
            p = rank_candidates(cand, scores)
            preds[ti, si, :len(p)] = p
            out[0][ti, si] = int(gt == p[0])
            out[1][ti, si] = int(p[0] in (gt - 1, gt, gt + 1))
            out[2][ti, si] = int(gt in p[:3])
            out[3][ti, si] = int(gt in p[:10])
    return out[0], out[1], out[2], out[3], preds
