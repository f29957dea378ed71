"""CPU: the search/eval oracle (oracle/search.py) against brute force and hand-made cases
(eval/eval_faiss.py:199-246)."""
import numpy as np

from oracle import search as S


def _unit(n, d, seed):
    x = np.random.default_rng(seed).normal(size=(n, d))
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def test_flat_l2_matches_brute_force_and_breaks_ties_by_id():
    x = _unit(300, 16, 0)
    q = _unit(7, 16, 1)
    D, I = S.flat_l2_search(q, x, 5)
    for i in range(7):
        d = ((x.astype(np.float64) - q[i].astype(np.float64)) ** 2).sum(1)
        assert list(I[i]) == list(np.argsort(d, kind='stable')[:5])
        assert np.allclose(D[i], np.sort(d)[:5], atol=1e-12)
    x2 = np.concatenate([x[:10], x[:10]])          # duplicates: the smaller id first
    _, I2 = S.flat_l2_search(x[:3], x2, 2)
    assert [list(r) for r in I2] == [[0, 10], [1, 11], [2, 12]]


def test_sequence_pipeline_finds_the_planted_sequence():
    rng = np.random.default_rng(3)
    dummy = _unit(400, 32, 4)
    db = _unit(200, 32, 5)
    query = db + 0.05 * rng.normal(size=db.shape).astype(np.float32)
    query /= np.linalg.norm(query, axis=1, keepdims=True)
    t1e, t1n, t3, t10, preds = S.evaluate(query, db, dummy, np.array([0, 17, 150]), (1, 3, 5), k_probe=5)
    assert t1e.all() and t1n.all() and t3.all() and t10.all()
    assert list(preds[:, 0, 0]) == [400, 417, 550]
    # offset compensation: the hits of segment `offset` vote for start id - offset
    I = np.array([[10, 50], [11, 3], [0, 12]])
    assert list(S.sequence_candidates(I)) == [2, 10, 50]
    # a candidate that runs past the end of the index is scored on the rows that exist
    idx = _unit(10, 8, 6)
    q = idx[7:10]
    assert abs(S.sequence_score(q[:3], idx, 8) - np.mean([q[0] @ idx[8], q[1] @ idx[9]])) < 1e-12
