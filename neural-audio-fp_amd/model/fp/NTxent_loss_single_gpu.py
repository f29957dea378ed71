"""Host mirror of the reference's NT-Xent loss, backed by libnafp's fused HIP kernel.

Mirrors `NTxentLoss` of the reference (model/fp/NTxent_loss_single_gpu.py:29-82):
`NTxentLoss(n_org, n_rep, tau).compute_loss(emb_org, emb_rep) -> (loss, sim_mtx
(N, 2N-1), labels (N, 2N-1))`.  Like the reference it requires n_org == n_rep
(the reference's drop_diag/labels only line up in that case).  Tensors are torch
CUDA tensors; no CPU path.
"""
import torch

from ... import _lib


class NTxentLoss:
    def __init__(self, n_org=int(), n_rep=int(), tau=0.05, **kwargs):
        self.n_org, self.n_rep, self.tau = n_org, n_rep, tau
        self._labels = None
        self._lib = _lib.load()

    @property
    def labels(self):
        """tf.one_hot(tf.range(n_org), n_org*2-1) (NTxent_loss_single_gpu.py:42)."""
        if self._labels is None:
            lab = torch.zeros((self.n_org, 2 * self.n_org - 1), dtype=torch.float32, device='cuda')
            idx = torch.arange(self.n_org, device='cuda')
            lab[idx, idx] = 1.0
            self._labels = lab
        return self._labels

    def compute_loss(self, emb_org, emb_rep, return_sim=True):
        emb_org = _lib.require_cuda(torch.as_tensor(emb_org), 'emb_org').float().contiguous()
        emb_rep = _lib.require_cuda(torch.as_tensor(emb_rep), 'emb_rep').float().contiguous()
        n, d = emb_org.shape
        if emb_rep.shape != (n, d) or n != self.n_org or self.n_org != self.n_rep:
            raise ValueError(f'expected emb_org, emb_rep of shape ({self.n_org},{d}); '
                             f'got {tuple(emb_org.shape)}, {tuple(emb_rep.shape)}')
        dev = emb_org.device
        loss_sum = torch.empty((1,), dtype=torch.float32, device=dev)
        sim = torch.empty((n, 2 * n - 1), dtype=torch.float32, device=dev) if return_sim else None
        need = int(self._lib.nafp_ntxent_workspace_bytes(n, n))
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.nafp_ntxent_forward(
                _lib.ptr(emb_org), _lib.ptr(emb_rep), _lib.ptr(emb_org), _lib.ptr(emb_rep),
                n, n, 0, d, float(self.tau), _lib.ptr(loss_sum), _lib.ptr(sim), None, None,
                _lib.ptr(ws), need, _lib.current_stream()), 'ntxent_forward')
        loss = loss_sum[0] / n          # mean-CE(a) + mean-CE(b): NTxent_loss_single_gpu.py:78-82
        return loss, sim, self.labels
