"""Host mirror of the reference's online triplet loss (now-playing baseline), backed by libnafp.

Mirrors `OnlineTripletLoss` (model/fp/online_triplet_loss.py:34-244): constructor arguments, the
`(loss, pairwise_dist, num_active_triplets)` return of `compute_loss(emb_anchor, emb_pos)`, modes
'semi-hard' (training) and 'all' (validation) as trainer.py:160-169 uses them, plus 'all-balanced' and 'hardest'
(online_triplet_loss.py:215-227; 'hardest' keeps the reference's min over the MASKED distance matrix, i.e. a
hardest-negative distance of 0), use_anc_as_pos=True.  Squared distances / use_anc_as_pos=False raise NotImplementedError.
`num_active_triplets` is, as in the reference, the count of positive entries of the SCALAR loss (0 or 1,
online_triplet_loss.py:238).  `loss_and_grad` returns the gradients the tape would derive.
"""
import torch

from ... import _lib


_MODES = {'semi-hard': 0, 'all': 1, 'all-balanced': 2, 'hardest': 3}


class OnlineTripletLoss:
    def __init__(self, bsz=int(), n_anchor=int(), n_pos_per_anchor=int(), use_anc_as_pos=True, mode='semi-hard', margin=.5):
        if not use_anc_as_pos:
            raise NotImplementedError('use_anc_as_pos=False')
        if mode not in _MODES:
            raise NotImplementedError(mode)
        self.bsz, self.n_anchor = bsz, n_anchor
        self.n_pos_per_anchor = n_pos_per_anchor if n_pos_per_anchor else int((bsz - n_anchor) / n_anchor)
        self.use_anc_as_pos, self.mode, self.margin = use_anc_as_pos, mode, margin
        self._lib = _lib.load()

    def _call(self, emb_anchor, emb_pos, want_dist, want_grad):
        a = _lib.require_cuda(torch.as_tensor(emb_anchor), 'emb_anchor').detach().float().contiguous()
        p = _lib.require_cuda(torch.as_tensor(emb_pos), 'emb_pos').detach().float().contiguous()
        nA, d = a.shape
        nP = p.shape[0]
        if nA != self.n_anchor or nP != nA * self.n_pos_per_anchor or p.shape[1] != d:
            raise ValueError(f'expected anchors ({self.n_anchor},{d}) and positives ({self.n_anchor * self.n_pos_per_anchor},{d}); '
                             f'got {tuple(a.shape)}, {tuple(p.shape)}')
        dev = a.device
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        dist = torch.empty((nA, nP + nA), dtype=torch.float32, device=dev) if want_dist else None
        da = torch.empty_like(a) if want_grad else None
        dp = torch.empty_like(p) if want_grad else None
        need = int(self._lib.nafp_triplet_workspace_bytes(nA, nP))
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.nafp_triplet_forward(_lib.ptr(a), _lib.ptr(p), nA, nP, d, _MODES[self.mode],
                                                      float(self.margin), _lib.ptr(loss), _lib.ptr(dist), _lib.ptr(da), _lib.ptr(dp),
                                                      _lib.ptr(ws), need, _lib.current_stream()), 'triplet_forward')
        return loss[0], dist, da, dp

    def compute_loss(self, emb_anchor, emb_pos, squared=False):
        if squared:
            raise NotImplementedError('squared distances')
        loss, dist, _, _ = self._call(emb_anchor, emb_pos, True, False)
        return loss, dist, (loss > 0).float()

    def loss_and_grad(self, emb_anchor, emb_pos):
        loss, _, da, dp = self._call(emb_anchor, emb_pos, False, True)
        return loss, da, dp
