"""In-memory mini search test of the training loop, on the HIP library.

Host mirror of model/utils/mini_search_subroutines.py:122-260 of the reference: `mini_search_eval(query, db,
scopes, mode, display, gt_id_offset)` -> ((top1, top3, top10) in %, mean_rank) and
`mini_search_validation(ds, m_pre, m_fp, mode, scopes, max_n_samples)` -> (accs_by_scope, scopes, key_strs) over
the three embeddings f(.), L2(f(.)), g(f(.)) of `test_step` (trainer.py:67-77).  The pairwise score matrix and the
rank of the ground truth under the eye(scope) diagonal sum come from `nafp_minisearch_scores` /
`nafp_minisearch_ranks`; no (n_aug, nQ, nD) matrix is argsorted -- the rank of the ground-truth id is counted
directly (equal sums keep argsort's order).
"""
import numpy as np
import torch

from ... import _lib


def mini_search_eval(query, db, scopes=[1, 3, 5, 9, 11, 19], mode='argmin', display=True, gt_id_offset=0):
    if mode.lower() not in ('argmin', 'argmax'):
        raise NotImplementedError(mode)
    m = 0 if mode == 'argmin' else 1
    lib = _lib.load()
    query = _lib.require_cuda(torch.as_tensor(query), 'query').float()
    db = _lib.require_cuda(torch.as_tensor(db), 'db').float().contiguous()
    nQ, n_augs, d = query.shape
    nD = db.shape[0]
    n_scopes = len(scopes)
    ranks = [[] for _ in scopes]
    with torch.cuda.device(db.device):
        for a in range(n_augs):
            qa = query[:, a].contiguous()
            scores = torch.empty((nQ, nD), dtype=torch.float32, device=db.device)
            _lib.check(lib.nafp_minisearch_scores(_lib.ptr(qa), _lib.ptr(db), nQ, nD, d, m, _lib.ptr(scores), _lib.current_stream()),
                       'minisearch_scores')
            for i, s in enumerate(scopes):
                r = torch.empty((nQ - s + 1,), dtype=torch.int32, device=db.device)
                _lib.check(lib.nafp_minisearch_ranks(_lib.ptr(scores), nQ, nD, int(s), m, int(gt_id_offset), _lib.ptr(r),
                                                     _lib.current_stream()), 'minisearch_ranks')
                ranks[i].append(r)
    mean_rank = np.zeros(n_scopes)
    top1_acc, top3_acc, top10_acc = np.zeros(n_scopes), np.zeros(n_scopes), np.zeros(n_scopes)
    for i in range(n_scopes):
        r = torch.stack(ranks[i]).cpu().numpy()                       # (n_augs, n_targets)
        mean_rank[i] = r.mean()
        top1_acc[i], top3_acc[i], top10_acc[i] = 100. * (r < 1).mean(), 100. * (r < 3).mean(), 100. * (r < 10).mean()
    if display:
        color_cyan, color_def = '\033[36m', '\033[0m'
        line_int, line_float = '{:^6}\t' * n_scopes, '{:>4.2f}\t' * n_scopes
        print(color_cyan + 'Scope:\t', line_int.format(*scopes), color_def)
        print(color_cyan + 'T1acc:\t' + color_def, line_float.format(*top1_acc))
        print(color_cyan + 'mRank:\t' + color_def, line_float.format(*mean_rank))
    return (top1_acc, top3_acc, top10_acc), mean_rank


def mini_search_validation(ds, m_pre, m_fp, mode='argmin', scopes=[1, 3, 5, 9, 11, 19], max_n_samples=3000):
    """mini_search_subroutines.py:222-260: anchors of the validation batches form the DB, their replicas the queries."""
    from ..trainer import test_step
    key_strs = ['f', 'L2(f)', 'g(f)']
    m_fp.trainable = False
    bsz = ds.bsz
    n_anchor = bsz // 2
    n_iter = min(len(ds), max_n_samples // bsz)
    db, query = {k: [] for k in key_strs}, {k: [] for k in key_strs}
    for i in range(n_iter):
        X = ds[i]
        if len(X[0]) + len(X[1]) != bsz:
            continue
        for k, e in zip(key_strs, test_step(X, m_pre, m_fp)):
            db[k].append(e[:n_anchor]); query[k].append(e[n_anchor:])
    accs_by_scope = dict()
    for k in key_strs:
        print(f'======= mini-search-validation: \033[31m{mode} \033[33m{k} \033[0m=======' + '\033[0m')
        q, x = torch.cat(query[k]), torch.cat(db[k])
        usable = [s for s in scopes if s <= min(len(q), len(x))]
        accs, _ = mini_search_eval(q[:, None, :], x, usable, mode, display=True)
        accs_by_scope[k] = accs
    return accs_by_scope, scopes, key_strs
