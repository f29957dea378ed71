"""Oracle (TEST INFRASTRUCTURE): NT-Xent loss.  PARITY UNPINNED (see oracle/__init__.py).

Single-device form follows model/fp/NTxent_loss_single_gpu.py:29-82; the
multi-replica form follows model/fp/NTxent_loss_tpu.py:42-137 (a file nothing in
the reference imports; it is the specification of the sharded loss).

Third-party semantics restated: `tf.compat.v1.losses.softmax_cross_entropy`
(default reduction) = mean over rows of -log softmax(logits)[label];
`tf.nn.softmax_cross_entropy_with_logits` = the per-row vector, no reduction.
"""
import numpy as np


def _lse(x, axis=1):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def drop_diag(x):
    """NTxentLoss.drop_diag (NTxent_loss_single_gpu.py:46-49): (N,N) -> (N,N-1)."""
    n = x.shape[0]
    mask = ~np.eye(n, dtype=bool)
    return x[mask].reshape(n, n - 1)


def compute_loss(emb_org, emb_rep, tau=0.05, dtype=np.float64):
    """NTxentLoss.compute_loss (NTxent_loss_single_gpu.py:52-82).

    Returns (loss, sim_mtx (N,2N-1) = [ab | aa without diagonal], labels one-hot (N,2N-1)).
    """
    ha = np.asarray(emb_org, dtype=dtype)
    hb = np.asarray(emb_rep, dtype=dtype)
    n = ha.shape[0]
    assert hb.shape[0] == n
    aa = drop_diag(ha @ ha.T / tau)
    bb = drop_diag(hb @ hb.T / tau)
    ab = ha @ hb.T / tau
    ba = hb @ ha.T / tau
    la = np.concatenate([ab, aa], 1)
    lb = np.concatenate([ba, bb], 1)
    idx = np.arange(n)
    loss_a = (_lse(la) - la[idx, idx]).mean()
    loss_b = (_lse(lb) - lb[idx, idx]).mean()
    labels = np.zeros((n, 2 * n - 1), dtype)
    labels[idx, idx] = 1
    return loss_a + loss_b, la, labels


def grad_embeddings(emb_org, emb_rep, tau=0.05, dtype=np.float64):
    """Analytic d(loss)/d(emb_org), d(loss)/d(emb_rep) of compute_loss."""
    ha = np.asarray(emb_org, dtype=dtype)
    hb = np.asarray(emb_rep, dtype=dtype)
    n = ha.shape[0]
    z = np.concatenate([ha, hb], 0)
    s = z @ z.T / tau
    np.fill_diagonal(s, -np.inf)
    p = np.exp(s - _lse(s)[:, None])
    partner = np.concatenate([np.arange(n) + n, np.arange(n)])
    p[np.arange(2 * n), partner] -= 1.0
    g = p / n                                   # dLoss/dS (row r = its own CE, mean over n)
    dz = (g + g.T) @ z / tau
    return dz[:n], dz[n:]


def replica_loss_fn(emb_local, ha_large, hb_large, rank, tau=0.05, large_num=1e9,
                    dtype=np.float64):
    """NTxentLoss.loss_fn for one replica (NTxent_loss_tpu.py:90-137).

    emb_local = [ha; hb] of this replica (2*n_a rows); ha_large/hb_large are the
    cross-replica concatenations (R*n_a rows).  Returns the per-row loss vector
    (n_a,), no reduction, exactly as the reference returns it.
    """
    emb_local = np.asarray(emb_local, dtype=dtype)
    n_a = emb_local.shape[0] // 2
    ha, hb = emb_local[:n_a], emb_local[n_a:]
    ha_large = np.asarray(ha_large, dtype=dtype)
    hb_large = np.asarray(hb_large, dtype=dtype)
    big = ha_large.shape[0]
    lab = np.arange(n_a) + rank * n_a                      # NTxent_loss_tpu.py:50
    diag = np.zeros((n_a, big), dtype)
    diag[np.arange(n_a), lab] = 1
    aa = ha @ ha_large.T / tau - diag * large_num           # :116-117
    bb = hb @ hb_large.T / tau - diag * large_num           # :118-119
    ab = ha @ hb_large.T / tau
    ba = hb @ ha_large.T / tau
    la = np.concatenate([ab, aa], 1)
    lb = np.concatenate([ba, bb], 1)
    r = np.arange(n_a)
    return (_lse(la) - la[r, lab]) + (_lse(lb) - lb[r, lab])
